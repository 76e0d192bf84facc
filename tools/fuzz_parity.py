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

from tests.dialects import build_file, draw_dialect  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=240)
    ap.add_argument("--first-seed", type=int, default=0)
    ap.add_argument("--max-mb", type=float, default=24)
    ap.add_argument("--cpu-only", action="store_true")
    ap.add_argument("--reference", action="store_true",
                    help="build container only: compare the oracle with oracle/_ref/phyNGSC (mpiexec -np P) instead")
    ap.add_argument("--seeds", type=int, default=10 ** 9)
    ap.add_argument("--twice", action="store_true", help="compress every input twice in a row on the context (the second call is a predicted step)")
    ap.add_argument("--tiny", action="store_true",
                    help="files of 20..200 KB split over 8..24 ranks: rank regions of a record's size, records owned by two ranks")
    ap.add_argument("--driver", action="store_true",
                    help="also push every accepted file through ngsc_compress_file_multi (two contexts, small jobs)")
    ap.add_argument("--decode", action="store_true",
                    help="also read every container back with the GPU decoder (ngsc_decompress_file) and compare with the input")
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
        if a.driver or a.decode:
            import tempfile
            eng2 = api.Engine(0)
            dtmp = tempfile.mkdtemp(prefix="ngsc_fuzz_drv_")
    t_end = time.time() + a.seconds
    seed = a.first_seed
    n_ok = n_rej = n_bad = n_skip = n_crash = n_limit = 0
    total = 0
    last = time.time()
    while time.time() < t_end and seed < a.first_seed + a.seeds:
        rnd = random.Random(seed)
        d = draw_dialect(rnd, a.max_mb)
        if a.tiny:
            d["bytes"] = rnd.choice([20_000, 60_000, 200_000])
            d["P"] = rnd.choice([8, 12, 16, 20, 24])
        data = build_file(d, seed)
        total += len(data)
        _ln = data.split(b"\n")          # (quality lines may start with '@': count lines, do not look for "\n@")
        max_rec = max((sum(len(x) + 1 for x in _ln[k:k + 4]) for k in range(0, len(_ln) - 3, 4)), default=0)
        try:
            # (against the reference binary: the reference's own container also where it does not hold its input - numeric title
            # values beyond 32 bits, stored modulo 2^32; against the engine such input is refused by both sides, rc -5)
            ref, _ = orc.compress_file(data, d["P"], threads=8, as_reference=bool(a.reference))
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
                    rbytes = f.read()
                try:
                    same, msg = nf.equivalent(rbytes, ref)
                except AssertionError as e:
                    same, msg = False, f"the reference's output does not parse ({e}); oracle {len(ref)} B, reference {len(rbytes)} B"
                if not same:
                    verdict = "reference and oracle differ: " + msg
        elif eng is None:
            n_file = data.count(b"\n+\n") if o_rc == 0 else 0       # (an upper bound is enough to spot duplicates)
            if o_rc == 0 and not d["vary_sep"]:
                try:
                    back = orc.decompress(ref)
                except orc.OracleError as e:
                    back = None
                    verdict = f"oracle decoder rc {e.rc}"
                whole = data if d["tail"] != "ok" or d["final_newline"] else data + b"\n"
                if back is not None and back.count(b"\n+\n") > n_file:
                    pass      # a record owned by two ranks is stored twice (phyNGSC.cpp:295-300): no round trip to expect
                elif back is not None and d["tail"] == "ok" and back != data:
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
                if a.twice:
                    # the same job again: the context PREDICTS this call's shape from the first (no sync in the middle of the step, the
                    # histogram / tree kernels launched as the first call's flags say) - it has to write the same bytes
                    res2 = eng.compress_device(d_in, arr.size, d["P"])
                    if api.assemble_container(res2, d["P"], arr.size) != got:
                        got = b"the predicted second call differs from the first"
            except api.NgscError as e:
                got, g_rc = None, e.code
            if o_rc == 0 and g_rc == -8 and max((len(x) + 1 for x in _ln[0::4]), default=0) > 512:
                n_limit += 1     # a documented engine limit: a sub-block's first title of more than 512 bytes
                seed += 1
                continue
            if (o_rc == 0) != (g_rc == 0):
                verdict = f"oracle rc {o_rc}, engine rc {g_rc}"
            elif o_rc == 0 and got != ref:
                k = next((x for x in range(min(len(got), len(ref))) if got[x] != ref[x]), min(len(got), len(ref)))
                verdict = f"containers differ at byte {k} (sizes {len(got)} / {len(ref)})"
            elif o_rc == 0 and a.driver:
                # the file driver: jobs of a few ranks each, one or two contexts
                fq, outp = os.path.join(dtmp, "in.fastq"), os.path.join(dtmp, "out.ngsc")
                with open(fq, "wb") as f:
                    f.write(data)
                os.environ["NGSC_JOB_BYTES"] = str(rnd.choice([1, 300_000, 2_000_000, 9_000_000]))
                engines = [eng, eng2] if d["P"] % 2 == 0 and rnd.random() < 0.5 else [eng]
                try:
                    api.compress_file_multi(engines, fq, outp, d["P"])
                    with open(outp, "rb") as f:
                        if f.read() != ref:
                            verdict = f"file driver output differs (NGSC_JOB_BYTES={os.environ['NGSC_JOB_BYTES']}, {len(engines)} contexts)"
                except api.NgscError as e:
                    verdict = f"file driver rc {e.code} (NGSC_JOB_BYTES={os.environ['NGSC_JOB_BYTES']})"
                del os.environ["NGSC_JOB_BYTES"]
        if verdict == "ok" and eng is not None and a.decode and o_rc == 0 and not d["vary_sep"] and d["tail"] == "ok":
            # the way back: the ORACLE's container (so that a decoder fault is not hidden by a matching encoder fault)
            # through the container reader and the GPU decoder; colour space is stored lossily and refused
            ng, back = os.path.join(dtmp, "d.ngsc"), os.path.join(dtmp, "d.fastq")
            with open(ng, "wb") as f:
                f.write(ref)
            try:
                eng.decompress_file(ng, back)
                with open(back, "rb") as f:
                    got = f.read()
                if got != data:
                    verdict = "GPU decoder: decoded file differs from the input"
            except api.NgscError as e:
                if e.code != -4:
                    verdict = f"GPU decoder rc {e.code}"
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
