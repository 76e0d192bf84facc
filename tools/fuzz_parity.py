#!/usr/bin/env python3
"""Developer tool (GPU box): randomised parity run.  Every seed draws a FASTQ dialect (title grammar, read length law,
base / quality alphabets, ambiguity density, final newline, virtual rank count), builds a file of it, compresses it with
the HIP engine and with the oracle, and compares the containers byte for byte.  Inputs the oracle rejects must be
rejected by the engine too.  A failing seed is replayed with --first-seed S --seeds 1.

  python3 tools/fuzz_parity.py [--seconds 240] [--first-seed 0] [--max-mb 24] [--cpu-only]

--cpu-only checks the generator against the oracle alone (oracle output must decode to its input when the
separators are regular); it is what runs where no GPU is present."""
from __future__ import annotations

import argparse
import os
import random
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402

from oracle import binding as orc  # noqa: E402

SEPS = b" ._,=:/-#"
TEXT = b"ABCDEFGHIJKLMNOPQRSTUVWXYZabcdefghijklmnopqrstuvwxyz0123456789+*;!?()[]{}<>~^&%$'\"\\|`"
IUPAC = b"RYKMSWBDHV"


def draw_dialect(rnd: random.Random, max_mb: float):
    d = {}
    nf = rnd.choice([1, 1, 2, 3, 4, 5, 6, 8, 10, 12])
    fields = []
    for _ in range(nf):
        kind = rnd.choice(["const", "inc", "inc", "rand", "rand", "text", "vtext", "mixed", "zeropad", "fixnum"])
        f = dict(kind=kind)
        if kind == "const":
            f["tok"] = bytes(rnd.choice(TEXT) for _ in range(rnd.randint(0, 12)))
        elif kind == "inc":
            f["start"] = rnd.choice([0, 1, 9, 99, 12345, 99999, 10 ** 6 - 50, rnd.randint(0, 10 ** 9), rnd.randint(0, 10 ** 9),
                                 2 ** 31 - 200000])
            f["step"] = rnd.choice([1, 1, 1, 2, 7, 1000])
            f["every"] = rnd.choice([1, 1, 2, 33, 1000])
        elif kind == "rand":
            f["hi"] = rnd.choice([1, 2, 9, 10, 255, 256, 65535, 65536, 10 ** 6, 10 ** 6, 2 ** 29, 2 ** 29, 2 ** 29, 2 ** 31 - 1])
            f["lo"] = rnd.choice([0, 0, f["hi"] // 2])
        elif kind in ("text", "vtext"):
            f["alpha"] = bytes(rnd.sample(list(TEXT), rnd.randint(1, 20)))
            lo = rnd.randint(0 if kind == "vtext" else 1, 6)
            f["len"] = (lo, lo if kind == "text" else lo + rnd.randint(1, 10))
        elif kind == "mixed":
            f["p_text"] = rnd.choice([0.001, 0.05, 0.5])
        elif kind == "zeropad":
            f["width"] = rnd.randint(2, 8)
        elif kind == "fixnum":
            f["val"] = rnd.choice([0, 1, 7, 36, 100, 4294967295, 2147483647])
        fields.append(f)
    d["fields"] = fields
    d["seps"] = [rnd.choice(SEPS) for _ in range(nf - 1)]
    d["vary_sep"] = rnd.random() < 0.1 and nf > 1          # lossy for the format, still has to agree
    law = rnd.choice(["fixed", "fixed", "uniform", "uniform", "bimodal", "tiny", "long"])
    if law == "fixed":
        L = rnd.choice([1, 15, 16, 17, 31, 32, 33, 36, 50, 64, 76, 100, 101, 127, 128, 150, 151, 250, 300])
        d["len"] = (L, L)
    elif law == "uniform":
        lo = rnd.choice([0, 1, 20, 36, 90])
        d["len"] = (lo, lo + rnd.choice([1, 10, 64, 264, 500]))
    elif law == "bimodal":
        d["len"] = (rnd.randint(20, 60), rnd.randint(200, 700))
    elif law == "tiny":
        d["len"] = (0, rnd.randint(1, 20))
    else:
        lo = rnd.choice([700, 1000, 1020, 1500, 3000])
        d["len"] = (lo, lo + rnd.choice([0, 10, 300]))
    d["law"] = law
    d["p_n"] = rnd.choice([0, 0, 0.001, 0.01, 0.05, 0.3])
    d["p_iupac"] = rnd.choice([0, 0, 0, 0.001, 0.02])
    d["p_lower"] = rnd.choice([0, 0, 0, 0.01])
    d["bases"] = rnd.choice([b"ACGT", b"ACGT", b"ACGT", b"ACG", b"AC", b"A", b"ACGTU"])
    nq = rnd.choice([1, 2, 4, 8, 20, 41, 42, 64, 65, 94])
    q_lo = rnd.choice([33, 33, 35, 59, 64])
    pool = list(range(q_lo, 127))
    d["quals"] = bytes(sorted(rnd.sample(pool, min(nq, len(pool)))))
    d["q_skew"] = rnd.choice([0, 1, 3])
    d["n_qual"] = rnd.choice(["low", "low", "any", "two"])
    d["p_all_n"] = rnd.choice([0, 0, 0.01])
    d["final_newline"] = rnd.random() > 0.15
    # how the file ends: complete, or with the damage real files show (both sides must agree on accept / reject)
    d["tail"] = rnd.choice(["ok"] * 12 + ["blank", "blank2", "cut_quality", "cut_sequence", "cut_title", "cut_plus", "junk"])
    d["P"] = rnd.choice([2, 2, 3, 4, 5, 7, 8, 8, 13, 16])
    mb = rnd.choice([0.02, 0.1, 0.5, 1.0, 3.0, 9.0, 17.0, max_mb])
    d["bytes"] = int(min(mb, max_mb) * (1 << 20))
    return d


def field_bytes(f, i, rnd):
    k = f["kind"]
    if k == "const":
        return f["tok"]
    if k == "inc":
        return b"%d" % (f["start"] + (i // f["every"]) * f["step"])
    if k == "rand":
        return b"%d" % rnd.randint(f["lo"], f["hi"])
    if k in ("text", "vtext"):
        n = rnd.randint(*f["len"])
        a = f["alpha"]
        return bytes(a[rnd.randrange(len(a))] for _ in range(n))
    if k == "mixed":
        return b"%d" % rnd.randint(0, 99999) if rnd.random() > f["p_text"] else b"x%dy" % rnd.randint(0, 99)
    if k == "zeropad":
        return b"%0*d" % (f["width"], rnd.randint(0, 10 ** f["width"] - 1))
    return b"%d" % f["val"]


def build_file(d, seed):
    rnd = random.Random(seed * 7919 + 1)
    nrng = np.random.default_rng(seed)
    lo, hi = d["len"]
    quals = np.frombuffer(d["quals"], dtype=np.uint8)
    w = (np.arange(len(quals), dtype=np.float64) + 1.0) ** d["q_skew"]
    w /= w.sum()
    bases = np.frombuffer(d["bases"], dtype=np.uint8)
    iupac = np.frombuffer(IUPAC, dtype=np.uint8)
    out = []
    size = 0
    i = 0
    while size < d["bytes"]:
        t = bytearray(b"@")
        for j, f in enumerate(d["fields"]):
            if j:
                t.append(rnd.choice(SEPS) if d["vary_sep"] and rnd.random() < 0.2 else d["seps"][j - 1])
            t += field_bytes(f, i, rnd)
        if d["law"] == "bimodal":
            L = lo if rnd.random() < 0.5 else hi
        else:
            L = rnd.randint(lo, hi)
        s = bases[nrng.integers(0, len(bases), L)]
        q = quals[nrng.choice(len(quals), L, p=w)]
        if L:
            if d["p_all_n"] and rnd.random() < d["p_all_n"]:
                mask = np.ones(L, dtype=bool)
            elif d["p_n"]:
                mask = nrng.random(L) < d["p_n"]
            else:
                mask = None
            if mask is not None and mask.any():
                s = s.copy()
                q = q.copy()
                s[mask] = ord("N")
                if d["n_qual"] == "low":
                    q[mask] = quals[0]
                elif d["n_qual"] == "two":
                    q[mask] = quals[nrng.integers(0, min(2, len(quals)), int(mask.sum()))]
            if d["p_iupac"]:
                m2 = nrng.random(L) < d["p_iupac"]
                if m2.any():
                    s = s.copy()
                    s[m2] = iupac[nrng.integers(0, len(iupac), int(m2.sum()))]
            if d["p_lower"]:
                m3 = nrng.random(L) < d["p_lower"]
                if m3.any():
                    s = s.copy()
                    s[m3] |= 0x20
        rec = bytes(t) + b"\n" + s.tobytes() + b"\n+\n" + q.tobytes() + b"\n"
        out.append(rec)
        size += len(rec)
        i += 1
    data = b"".join(out)
    tail = d.get("tail", "ok")
    last = out[-1]
    if tail == "ok":
        if not d["final_newline"]:
            data = data[:-1]
    elif tail == "blank":
        data += b"\n"
    elif tail == "blank2":
        data += b"\n\n"
    elif tail == "junk":
        data += b"xyz\n"
    else:
        nl = [k for k, c in enumerate(last) if c == 10]          # title, sequence, plus, quality newlines
        keep = {"cut_title": max(1, nl[0] // 2), "cut_sequence": (nl[0] + nl[1] + 1) // 2,
                "cut_plus": nl[1] + 2, "cut_quality": (nl[2] + nl[3] + 1) // 2}[tail]
        data = data[: len(data) - len(last) + keep]
    return data


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--max-mb", type=float, default=24)
    ap.add_argument("--cpu-only", action="store_true")
    ap.add_argument("--reference", action="store_true",
                    help="build container only: compare the oracle with oracle/_ref/phyNGSC (mpiexec -np P) instead")
    ap.add_argument("--seeds", type=int, default=10 ** 9)
    a = ap.parse_args()

    eng = None
    if a.reference:
        import subprocess
        import tempfile
        from phyngsc_amd import ngsc_format as nf
        ref_bin = os.path.join(ROOT, "oracle", "_ref", "phyNGSC")
        mpiexec = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")
        assert os.path.exists(ref_bin), "run oracle/build_ref.sh first"
        tmp = tempfile.mkdtemp(prefix="ngsc_fuzz_")
        a.cpu_only = True
    if not a.cpu_only:
        import torch
        from phyngsc_amd import api
        assert torch.cuda.is_available(), "needs the MI355X (or --cpu-only)"
        eng = api.Engine(0)
    t_end = time.time() + a.seconds
    seed = a.first_seed
    n_ok = n_rej = n_bad = n_skip = n_crash = n_limit = 0
    total = 0
    last = time.time()
    while time.time() < t_end and seed < a.first_seed + a.seeds:
        rnd = random.Random(seed)
        d = draw_dialect(rnd, a.max_mb)
        data = build_file(d, seed)
        total += len(data)
        max_rec = max(len(x) for x in data.split(b"\n@")) + 2
        try:
            ref, _ = orc.compress_file(data, d["P"], threads=8)
            o_rc = 0
        except orc.OracleError as e:
            ref, o_rc = None, e.rc
        verdict = "ok"
        if a.reference and (o_rc or max_rec > 480 or d["P"] > 8 or d["tail"] != "ok"):
            n_skip += 1          # outside the reference's defined behaviour or too slow here
            seed += 1
            continue
        if a.reference:
            fq, outp = os.path.join(tmp, "in.fastq"), os.path.join(tmp, "out.ngsc")
            with open(fq, "wb") as f:
                f.write(data)
            if os.path.exists(outp):
                os.remove(outp)
            r = subprocess.run([mpiexec, "-np", str(d["P"]), ref_bin, fq, outp, "1"], capture_output=True, text=True,
                               timeout=300)
            r_ok = r.returncode == 0 and os.path.exists(outp) and "WARNING" not in r.stdout
            if not r_ok and r.returncode in (11, 139, -11):
                # the reference segfaults on inputs the oracle takes: known cases are MakeFooter's log2(0) when every
                # rank region starts on a record; counted, not compared
                n_crash += 1
                seed += 1
                continue
            if r_ok != (o_rc == 0):
                verdict = f"reference ok={r_ok} (rc {r.returncode}: {r.stdout[-200:]!r}), oracle rc {o_rc}"
            elif r_ok:
                with open(outp, "rb") as f:
                    same, msg = nf.equivalent(f.read(), ref)
                if not same:
                    verdict = "reference and oracle differ: " + msg
        elif eng is None:
            if o_rc == 0 and not d["vary_sep"]:
                try:
                    back = orc.decompress(ref)
                except orc.OracleError as e:
                    back = None
                    verdict = f"oracle decoder rc {e.rc}"
                whole = data if d["tail"] != "ok" or d["final_newline"] else data + b"\n"
                if back is not None and d["tail"] == "ok" and back != data:
                    verdict = "oracle round trip differs"
                elif back is not None and not whole.startswith(back[:-1]):    # damaged tail: the records before it
                    verdict = "oracle round trip is not a prefix of the input"
        else:
            arr = np.frombuffer(data, dtype=np.uint8)
            d_in = torch.from_numpy(arr.copy()).cuda()
            try:
                res = eng.compress_device(d_in, arr.size, d["P"])
                got = api.assemble_container(res, d["P"], arr.size)
                g_rc = 0
            except api.NgscError as e:
                got, g_rc = None, e.code
            if o_rc == 0 and g_rc == -8 and len(data) // d["P"] < 4 * max_rec:
                # documented engine limit: rank regions of about one record's size (two ranks own the same record)
                n_limit += 1
                seed += 1
                continue
            if (o_rc == 0) != (g_rc == 0):
                verdict = f"oracle rc {o_rc}, engine rc {g_rc}"
            elif o_rc == 0 and got != ref:
                k = next((x for x in range(min(len(got), len(ref))) if got[x] != ref[x]), min(len(got), len(ref)))
                verdict = f"containers differ at byte {k} (sizes {len(got)} / {len(ref)})"
        if verdict == "ok":
            if o_rc:
                n_rej += 1
            else:
                n_ok += 1
        else:
            n_bad += 1
            print(f"MISMATCH seed {seed}: {verdict}\n  dialect {d}", flush=True)
        if time.time() - last > 20:
            last = time.time()
            print(f"[fuzz] seed {seed}: {n_ok} equal, {n_rej} rejected by both, {n_bad} mismatches, {total / 1e6:.0f} MB",
                  flush=True)
        seed += 1
    print(f"fuzz: seeds {a.first_seed}..{seed - 1}: {n_ok} equal, {n_rej} rejected by both, {n_bad} mismatches, "
          f"{n_skip} skipped, {n_crash} reference crashes, {n_limit} engine limits, {total / 1e6:.0f} MB of FASTQ", flush=True)
    if eng is not None:
        eng.close()
    sys.exit(1 if n_bad else 0)


if __name__ == "__main__":
    main()
