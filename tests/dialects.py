"""Random FASTQ dialects for the randomised parity runs (tools/fuzz_parity.py, tests/test_gpu_parity.py,
tests/test_oracle_roundtrip.py): a seed draws a title grammar, read length law, base / quality alphabets, ambiguity
density, the way the file ends and the virtual rank count, and builds a file of it.  Test infrastructure."""
from __future__ import annotations

import random

import numpy as np

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
    # drawn last, so that every earlier seed keeps its dialect: now and then a title of more than 64 fields
    if rnd.random() < 0.05:
        extra = rnd.choice([64, 70, 100, 100, 120, 120, 200]) - nf   # (200: a first title beyond the engine's 512 bytes)
        for _ in range(max(extra, 0)):
            kind = rnd.choice(["const", "inc", "rand", "text", "fixnum"])
            f = dict(kind=kind)
            if kind == "const":
                f["tok"] = bytes(rnd.choice(TEXT) for _ in range(rnd.randint(0, 3)))
            elif kind == "inc":
                f["start"], f["step"], f["every"] = rnd.randint(0, 10 ** 6), 1, rnd.choice([1, 33])
            elif kind == "rand":
                f["hi"] = rnd.choice([1, 9, 255, 65536])
                f["lo"] = 0
            elif kind == "text":
                f["alpha"] = bytes(rnd.sample(list(TEXT), rnd.randint(1, 4)))
                f["len"] = (1, 2)
            else:
                f["val"] = rnd.choice([0, 7, 100])
            d["fields"].append(f)
            d["seps"].append(rnd.choice(SEPS))
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


def to_json(x):
    """A dialect as JSON-able data (bytes -> {"__bytes__": latin-1 text}); from_json() is the inverse."""
    if isinstance(x, (bytes, bytearray)):
        return {"__bytes__": bytes(x).decode("latin-1")}
    if isinstance(x, dict):
        return {k: to_json(v) for k, v in x.items()}
    if isinstance(x, (list, tuple)):
        return [to_json(v) for v in x]
    return x


def from_json(x):
    if isinstance(x, dict):
        if set(x) == {"__bytes__"}:
            return x["__bytes__"].encode("latin-1")
        return {k: from_json(v) for k, v in x.items()}
    if isinstance(x, list):
        return [from_json(v) for v in x]
    return x


def manifest_input(man):
    """The input a golden manifest (tests/golden/manifest_*.json) was made from: generator-made, or a stored dialect."""
    import hashlib
    from phyngsc_amd import fastq_gen
    if "dialect" in man:
        data = np.frombuffer(build_file(from_json(man["dialect"]), man["seed"]), dtype=np.uint8)
        assert hashlib.sha256(data.tobytes()).hexdigest() == man["fastq_sha256"], "tests/dialects.py no longer rebuilds this input"
    else:
        data = fastq_gen.generate(fastq_gen.GenConfig(**man["generator"]), target_bytes=man["target_bytes"])
    assert data.size == man["fastq_bytes"]
    return data
