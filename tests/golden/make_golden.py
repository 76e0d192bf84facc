#!/usr/bin/env python3
"""Generates the golden vectors under tests/golden/ by running the UNMODIFIED reference binary.

Run in the build container only (needs oracle/_ref/phyNGSC from oracle/build_ref.sh and MPICH's
mpiexec).  For every case it writes

  small/<name>.fastq          the crafted input (committed whole, <= 256 KB each)
  small/<name>.p<P>.ngsc      the reference's output, threads=1 (the StoreTree static-buffer race,
                              huffman.cpp:193-194, makes threads>1 unsafe)
  manifest_<name>.json        for generator-made inputs (10 MB .. 1 GB): generator parameters +
                              per-(WRID,k) block sha256 + footer scalars of the reference's output

Fixtures are data only: inputs and expected outputs.  The reference itself never travels.
"""
from __future__ import annotations

import json
import os
import random
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)

from phyngsc_amd import fastq_gen, ngsc_format  # noqa: E402

REF = os.path.join(ROOT, "oracle", "_ref", "phyNGSC")
MPIEXEC = os.environ.get("MPIEXEC", "/opt/conda/bin/mpiexec")
SMALL = os.path.join(HERE, "small")


def run_ref(fastq_path: str, P: int, out_path: str) -> bool:
    if os.path.exists(out_path):
        os.remove(out_path)
    r = subprocess.run([MPIEXEC, "-np", str(P), REF, fastq_path, out_path, "1"], capture_output=True, text=True,
                       timeout=3600)
    ok = r.returncode == 0 and os.path.exists(out_path) and "WARNING" not in r.stdout
    if not ok:
        print(f"   reference failed rc={r.returncode}: {r.stdout[-300:]} {r.stderr[-300:]}")
    return ok


# ------------------------------------------------------------------------------------------------
# crafted small inputs
# ------------------------------------------------------------------------------------------------
def rec(title: str, seq: str, qual: str) -> str:
    assert len(seq) == len(qual)
    return f"{title}\n{seq}\n+\n{qual}\n"


def rand_seq(rng, n, alphabet="ACGT"):
    return "".join(rng.choice(alphabet) for _ in range(n))


def rand_qual(rng, n, lo=33, hi=73, p_top=0.7):
    return "".join(chr(hi) if rng.random() < p_top else chr(rng.randint(lo, hi)) for _ in range(n))


def case_err_style(rng, n, L=36, start=1):
    return "".join(rec(f"@ERR005195.{start + i} BGI-FC30BFTAAXX_5_1_000:{rng.randint(1, 2000)}/2",
                       rand_seq(rng, L), rand_qual(rng, L)) for i in range(n))


def case_n_transfer(rng, n, L=50, mode="transfer"):
    out = []
    amb = "NYRWSKMDVHBXU.-"
    for i in range(n):
        seq = list(rand_seq(rng, L))
        qual = list(rand_qual(rng, L))
        k = rng.choice([0, 0, 1, 2, 5])
        for _ in range(k):
            p = rng.randrange(L)
            seq[p] = rng.choice(amb) if mode != "nonly" else "N"
            if mode == "transfer" or (mode == "mixed" and rng.random() < 0.7) or mode == "nonly":
                qual[p] = chr(rng.randint(33, 40))
            else:
                qual[p] = chr(rng.randint(41, 73))
        out.append(rec(f"@SRR1.{i + 1} L:{rng.randint(1, 8)}:{rng.randint(1000, 9999)}", "".join(seq), "".join(qual)))
    return "".join(out)


def case_text_titles(rng, n, L=40, long_field=False):
    words = ["alpha", "beta", "gamma", "delta", "epsilon", "zeta", "eta", "theta", "iota", "kappa", "la", "m"]
    out = []
    for i in range(n):
        w = rng.choice(words)
        tag = "".join(rng.choice("ACGTN") for _ in range(rng.randint(4, 9)))
        extra = ""
        if long_field:
            ln = rng.choice([3, 60, 127, 129, 130, 200, 255])
            extra = " " + "".join(rng.choice("abcdefghijklmnopqrstuvwxyz0123456789") for _ in range(ln))
        out.append(rec(f"@read{w}{extra} sample={tag}:{rng.randint(0, 3)}#{i % 7}", rand_seq(rng, L), rand_qual(rng, L)))
    return "".join(out)


def case_numeric_fields(rng, n, L=36):
    out = []
    a = 100
    b = 5_000_000
    for i in range(n):
        a += rng.randint(1, 5)          # small positive deltas -> delta coding + Huffman over deltas
        b += rng.randint(-3000, 9000)   # wide deltas (> 512 distinct) -> plain bit packing
        lane1 = rng.randint(1, 8)       # small value range, field index 1 (< last vector realloc): no tree
        lane2 = rng.randint(1, 8)       # same at index 6 (kept num_values): value coding + Huffman
        big = rng.randint(0, 1_000_000_000)
        out.append(rec(f"@run:{lane1}:{a}:{b}:{big}:{i // 40}:{lane2}", rand_seq(rng, L), rand_qual(rng, L)))
    return "".join(out)


def case_leading_zero_and_switch(rng, n, L=30):
    out = []
    for i in range(n):
        z = f"{rng.randint(0, 999):03d}"                      # leading zeros -> text (utils.h:107-114)
        sw = str(rng.randint(10, 99)) if i < n // 2 else rng.choice(["ab", "cd", "12"])  # numeric then text
        empty = "" if i % 3 else "x"                          # sometimes-empty field
        out.append(rec(f"@id_{z}_{sw}..{empty},{i}", rand_seq(rng, L), rand_qual(rng, L)))
    return "".join(out)


def case_group_constant(rng, n, L=36):
    """Text field constant inside 32-record groups, numeric field with constant delta in groups."""
    out = []
    v = 1000
    for i in range(n):
        g = i // 32
        tile = f"T{g % 5}x"
        step = 1 if (g % 3) else rng.randint(1, 2)
        v += step
        same = 77 if (g % 2) else rng.randint(70, 80)
        out.append(rec(f"@{tile}:{v}:{same}:{rng.choice(['aa', 'ab', 'b'])}", rand_seq(rng, L), rand_qual(rng, L)))
    return "".join(out)


def case_const_titles(rng, n, L=36):
    return "".join(rec("@same title:1/2", rand_seq(rng, L), rand_qual(rng, L)) for _ in range(n))


def case_varlen(rng, n, lo=20, hi=120):
    out = []
    for i in range(n):
        L = rng.randint(lo, hi)
        out.append(rec(f"@v.{i + 1} len={L}", rand_seq(rng, L), rand_qual(rng, L, 33, 126, 0.2)))
    return "".join(out)


def case_alphabets(rng, n, L, seq_alpha, q_lo, q_hi):
    return "".join(rec(f"@a.{i + 1}", rand_seq(rng, L, seq_alpha), rand_qual(rng, L, q_lo, q_hi, 0.5)) for i in range(n))


def small_cases():
    cases = []

    def add(name, text, ranks):
        assert len(text) <= 256 * 1024, (name, len(text))
        cases.append((name, text, ranks))

    rng = random.Random(20261003)
    add("err_8rec", case_err_style(rng, 8), [2])
    add("err_62rec", case_err_style(rng, 62), [2])
    add("err_64rec", case_err_style(rng, 64), [2])
    add("err_66rec", case_err_style(rng, 66), [2])
    add("err_200rec_digits", case_err_style(rng, 200, start=950), [2, 3])
    add("err_2000rec", case_err_style(rng, 2000), [2, 4, 8])
    add("n_transfer", case_n_transfer(rng, 600, mode="transfer"), [2])
    add("n_keep", case_n_transfer(rng, 600, mode="keep"), [2])
    add("n_mixed", case_n_transfer(rng, 900, mode="mixed"), [2, 3])
    add("n_only", case_n_transfer(rng, 500, mode="nonly"), [2])
    add("title_text", case_text_titles(rng, 700), [2, 4])
    add("title_long", case_text_titles(rng, 300, long_field=True), [2])
    add("title_numeric", case_numeric_fields(rng, 1500), [2, 3])
    add("title_zero_switch", case_leading_zero_and_switch(rng, 500), [2])
    add("title_group_const", case_group_constant(rng, 1000), [2, 4])
    add("title_const", case_const_titles(rng, 301), [2])
    add("varlen", case_varlen(rng, 800), [2, 5])
    add("lowercase", case_alphabets(rng, 300, 40, "ACGTacgtn", 33, 73), [2])
    add("dna_3sym", case_alphabets(rng, 300, 40, "ACG", 33, 73), [2])
    add("dna_1sym", case_alphabets(rng, 300, 40, "A", 33, 73), [2])
    add("qual_1sym", case_alphabets(rng, 300, 40, "ACGT", 73, 73), [2])
    add("qual_2sym", case_alphabets(rng, 300, 40, "ACGT", 72, 73), [2])
    add("qual_wide", case_alphabets(rng, 1200, 60, "ACGT", 33, 126), [2])
    # later additions draw from their own generator so that the cases above stay byte-identical
    rng2 = random.Random(20261004)
    add("no_final_newline", case_err_style(rng2, 1500)[:-1], [2, 3])
    add("empty_reads", "".join(rec(f"@e.{i + 1}", rand_seq(rng2, L), rand_qual(rng2, L))
                               for i in range(2500) for L in [0 if i % 53 == 7 else 1 + (i * 31) % 50]), [2, 3])
    add("iupac_odd_quals", "".join(rec(f"@q.{i + 1}", rand_seq(rng2, 36, "ACGTACGTACGTRYKMSWN" if i % 3 == 0 else "ACGT"),
                                       "".join(rng2.choice("IJKabcxyz~") for _ in range(36))) for i in range(2000)), [2])
    # inputs the randomised runs (tools/fuzz_parity.py) found the first engine / oracle versions wrong or too strict on
    rng3 = random.Random(20261005)
    el = [rec(f"@el.{i + 1}", rand_seq(rng3, i % 5), rand_qual(rng3, i % 5)) for i in range(1500)] + [rec("@el.1501", "", "")]
    add("empty_last_quality", "".join(el)[:-1], [2, 3])            # ends with "+\n" and an unterminated empty line
    add("const_numeric_beyond_int32", "".join(rec(f"@c{i % 7} 4294967295 x:{i} 99999999999", rand_seq(rng3, 36), rand_qual(rng3, 36))
                                              for i in range(1500)), [2])
    add("twelve_numeric_fields", "".join(rec("@M" + ":".join(str((i * (k + 3) + k * k) % (10 ** (1 + k % 6))) for k in range(12)),
                                             rand_seq(rng3, 36), rand_qual(rng3, 36)) for i in range(1500)), [2])
    add("many_text_fields", "".join(rec("@T" + " ".join("".join(rng3.choice("abcdefgh") for _ in range(1 + (k + rng3.randrange(3)) % 4))
                                                        for k in range(8)), rand_seq(rng3, 50), rand_qual(rng3, 50))
                                    for i in range(1800)), [10])   # 10 sub-blocks x 8 varying text fields
    # colour space (SOLiD csfastq): primer base + transitions 0..3, '.' = unknown (phyNGSC.cpp:473-547)
    rng4 = random.Random(20261006)

    def cs_read(L, dots=0.0, primer="T"):
        return primer + "".join("." if rng4.random() < dots else rng4.choice("0123") for _ in range(L - 1))

    def cs_qual(L, first="!"):
        return first + "".join(chr(rng4.randint(34, 70)) for _ in range(L - 1))
    add("cs_constant", "".join(rec(f"@solid_{i}_F3", cs_read(36), cs_qual(36)) for i in range(1500)), [2, 3])
    add("cs_breaks", "".join(rec(f"@solid_{i}_F3", cs_read(36), cs_qual(36, "!" if i < 400 or i % 5 else "5"))
                             for i in range(1500)), [2])           # the 401st read ends "delta constant" (rank 0), rank 1 starts anew
    add("cs_dots", "".join(rec(f"@solid_{i}_F3", cs_read(50, dots=0.02), cs_qual(50)) for i in range(1200)), [2])
    add("cs_primers", "".join(rec(f"@solid_{i}_F3", cs_read(30 + i % 9, primer="TGAC"[i % 4]), cs_qual(30 + i % 9))
                              for i in range(1500)), [2])
    add("cs_no_primer", "".join(rec(f"@solid_{i}_F3", cs_read(36, dots=0.005, primer=rng4.choice("0123")), cs_qual(36))
                                for i in range(1500)), [2])
    add("digit_first_base", "".join(rec(f"@d.{i}", ("2" if i % 700 == 0 else "A") + rand_seq(rng4, 35), rand_qual(rng4, 36))
                                    for i in range(1400)), [2])    # base space whose first read starts with a digit: flags only
    # round 3: titles on what were engine limits (a varying text field whose FIRST record is longer than 256 bytes; more than 64
    # separators) - both inside the reference's 500-byte record domain
    rng5 = random.Random(20261007)

    def long_text(i):
        n = 300 if i == 0 else rng5.randint(180, 320)
        return "".join(rng5.choice("ACGTacgt-_") if k % 7 else rng5.choice("xyz") for k in range(n)).replace("-", "q").replace("_", "w")
    add("title_text_first_300", "".join(rec(f"@L{i % 3} {long_text(i)} {i}", rand_seq(rng5, 40), rand_qual(rng5, 40)) for i in range(420)), [2])
    add("title_fields_80", "".join(rec(f"@F{i}:" + ":".join(str((i * (k + 1) + k) % (3 + k % 7)) if k % 5 else "c" for k in range(79)),
                                       rand_seq(rng5, 36), rand_qual(rng5, 36)) for i in range(900)), [2, 3])
    # round 4: what only the randomised runs had caught (7241c91): a quality alphabet reaching past 'Q' - the raw bytes 82..89 would
    # land on the table columns of a transferred N's quality - together with N bases that are transferred (quality '!'..'(') and N /
    # IUPAC bases that keep their place (any other quality), at 36 bases (one tile, short last chunk) and 100 bases (windowed tables)
    rng6 = random.Random(20261008)

    def wide_q_with_n(n, L):
        out = []
        for i in range(n):
            seq = list(rand_seq(rng6, L))
            qual = [chr(rng6.choice([73, 73, 73, 80, 81, 82, 83, 85, 88, 89, 90, 95, 104, 126]) if rng6.random() < 0.8 else rng6.randint(41, 126)) for _ in range(L)]
            kind = i % 4            # 0: no N, 1: transferred N, 2: N that keeps its place, 3: both kinds of record mixed with IUPAC
            for _ in range(0 if kind == 0 else rng6.choice([1, 1, 2, 5])):
                p = rng6.randrange(L)
                seq[p] = "N" if kind != 3 else rng6.choice("NNRYK")
                qual[p] = chr(rng6.randint(33, 40)) if kind == 1 or (kind == 3 and rng6.random() < 0.5) else chr(rng6.choice([82, 85, 89, 41, 73]))
            out.append(rec(f"@w.{i + 1}", "".join(seq), "".join(qual)))
        return "".join(out)
    add("qual_past_Q_with_N_36", wide_q_with_n(2400, 36), [2, 3])
    add("qual_past_Q_with_N_100", wide_q_with_n(1000, 100), [2])
    # round 4, last session: the wide title kernels (more than 64 separators) pinned by the reference binary beyond 80 fields - 130
    # fields (numeric ones that vary, text ones that vary in content and in length, constants) and 200 one- and two-character fields,
    # both inside the reference's 500-byte record domain
    rng7 = random.Random(20261009)

    def wide_title(i, nf):
        f = []
        for k in range(nf - 1):
            if k % 9 == 0:
                f.append(str((i * (k + 3) + k) % (5 + k % 11)))                                  # numeric, varying
            elif k % 13 == 5:
                f.append("".join(rng7.choice("xyz") for _ in range(1 + (i + k) % 2)))            # text, varying length
            elif k % 7 == 3:
                f.append(rng7.choice("ab"))                                                      # text, varying content
            else:
                f.append("c")                                                                    # constant
        return f"@W{i}:" + ":".join(f)
    add("title_fields_130", "".join(rec(wide_title(i, 130), rand_seq(rng7, 36), rand_qual(rng7, 36)) for i in range(700)), [2, 3])
    add("title_fields_200", "".join(rec(wide_title(i, 200), rand_seq(rng7, 30), rand_qual(rng7, 30)) for i in range(500)), [2])
    return cases


# generator-made inputs: (name, GenConfig, target bytes, ranks)
GEN_CASES = [
    ("cfg1_10MiB", fastq_gen.CFG1, 10 << 20, [2, 8]),
    ("cfg2_100MB", fastq_gen.CFG2, 100_000_000, [2, 8]),
    ("cfg3_150bp_100MB", fastq_gen.CFG3, 100_000_000, [4, 8]),
    ("cfg5_mixed_60MB", fastq_gen.CFG5, 60_000_000, [8]),
    ("cfg5_36_150_60MB", fastq_gen.CFG5_150, 60_000_000, [8]),
    ("cfg5_36_300_60MB", fastq_gen.CFG5_300, 60_000_000, [8]),   # records beyond the overlap (Q7): kept because the reference's run is deterministic here and equals the oracle
    ("cfg2_1GB", fastq_gen.CFG2, 1_000_000_000, [8]),
    ("cfg2fixed_256MB", fastq_gen.CFG2_FIXED, 256 << 20, [8]),
]


# inputs of the randomised runs kept as manifests: (name, seed of tests/dialects.py, max MB the dialect was drawn with)
DIALECT_CASES = [
    ("dialect_shared_record", 11040, 24.0),     # 17 MB, P = 2: rank 0's last window holds one record, the next is rank 1's first
]


def main():
    only = set(sys.argv[1:])
    os.makedirs(SMALL, exist_ok=True)
    if not os.path.exists(REF):
        sys.exit("oracle/_ref/phyNGSC missing: run oracle/build_ref.sh first")
    index = {"small": {}, "generated": {}}
    for name, text, ranks in small_cases():
        if only and name not in only and "small" not in only:
            continue
        path = os.path.join(SMALL, name + ".fastq")
        with open(path, "w", newline="") as f:
            f.write(text)
        good = []
        for P in ranks:
            outp = os.path.join(SMALL, f"{name}.p{P}.ngsc")
            print(f"[small] {name} P={P}")
            if run_ref(path, P, outp):
                good.append(P)
            elif os.path.exists(outp):
                os.remove(outp)
        index["small"][name] = good
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        for name, cfg, target, ranks in GEN_CASES:
            if only and name not in only and "generated" not in only:
                continue
            data = fastq_gen.generate(cfg, target_bytes=target)
            path = os.path.join(tmp, name + ".fastq")
            data.tofile(path)
            man = {"generator": cfg.__dict__, "target_bytes": target, "fastq_bytes": int(data.size), "ranks": {}}
            for P in ranks:
                outp = os.path.join(tmp, f"{name}.p{P}.ngsc")
                print(f"[gen] {name} ({data.size} B) P={P}")
                if run_ref(path, P, outp):
                    with open(outp, "rb") as f:
                        man["ranks"][str(P)] = ngsc_format.manifest(f.read())
                    os.remove(outp)
            with open(os.path.join(HERE, f"manifest_{name}.json"), "w") as f:
                json.dump(man, f, indent=1, sort_keys=True)
            os.remove(path)
    with tempfile.TemporaryDirectory(dir="/tmp") as tmp:
        import hashlib
        import random as _random
        from tests import dialects
        for name, seed, max_mb in DIALECT_CASES:
            if only and name not in only and "dialects" not in only:
                continue
            d = dialects.draw_dialect(_random.Random(seed), max_mb)
            data = dialects.build_file(d, seed)
            path = os.path.join(tmp, name + ".fastq")
            with open(path, "wb") as f:
                f.write(data)
            man = {"dialect": dialects.to_json(d), "seed": seed, "fastq_bytes": len(data),
                   "fastq_sha256": hashlib.sha256(data).hexdigest(), "ranks": {}}
            P = d["P"]
            outp = os.path.join(tmp, f"{name}.p{P}.ngsc")
            print(f"[dialect] {name} ({len(data)} B) P={P}")
            if run_ref(path, P, outp):
                with open(outp, "rb") as f:
                    man["ranks"][str(P)] = ngsc_format.manifest(f.read())
            with open(os.path.join(HERE, f"manifest_{name}.json"), "w") as f:
                json.dump(man, f, indent=1, sort_keys=True)
    if index["small"]:
        idx_path = os.path.join(HERE, "index.json")
        merged = {}
        if only and "small" not in only and os.path.exists(idx_path):   # a partial run keeps the other cases' entries
            with open(idx_path) as f:
                merged = json.load(f)
        merged.update(index["small"])
        with open(idx_path, "w") as f:
            json.dump(merged, f, indent=1, sort_keys=True)


if __name__ == "__main__":
    main()
