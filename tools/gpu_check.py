#!/usr/bin/env python3
"""Developer tool (GPU box): runs the HIP engine on fixtures and localises any mismatch against the
golden reference output / the oracle, sub-block by sub-block and stream by stream."""
from __future__ import annotations

import glob
import os
import re
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

from oracle import binding as orc  # noqa: E402
from phyngsc_amd import api, fastq_gen, ngsc_format as nf  # noqa: E402


def split_streams(sub: bytes):
    """info|title|quality|DNA lengths cannot be read from the bytes alone; return the raw bytes."""
    return sub


def compare_case(eng, name, data: bytes, P: int, golden: bytes | None, verbose=True):
    arr = np.frombuffer(data, dtype=np.uint8)
    d_in = torch.from_numpy(arr.copy()).cuda()
    t0 = time.time()
    try:
        res = eng.compress_device(d_in, arr.size, P)
    except Exception as e:  # noqa: BLE001
        print(f"FAIL {name} P={P}: engine raised {e}")
        return False
    dt = time.time() - t0
    out = api.assemble_container(res, P, arr.size)
    ref = golden
    if ref is None:
        ref, _ = orc.compress_file(data, P, threads=8)
    ok, msg = nf.equivalent(ref, out)
    if ok:
        if verbose:
            print(f"OK   {name} P={P} ({len(res.subblocks)} sub-blocks, {dt*1e3:.1f} ms)")
        return True
    print(f"FAIL {name} P={P}: {msg}")
    # localise: compare per-rank sub-block streams
    try:
        rs_ref = nf.rank_streams(ref)
        host = res.out[:res.total_bytes].cpu().numpy().tobytes()
        for rk in res.ranks:
            for k in range(rk.n_subblocks):
                sb = res.subblocks[rk.first_subblock + k]
                mine = host[sb.out_off:sb.out_off + sb.size]
                theirs = rs_ref[rk.rank][k] if k < len(rs_ref.get(rk.rank, [])) else b""
                if mine == theirs:
                    continue
                n = min(len(mine), len(theirs))
                first = next((i for i in range(n) if mine[i] != theirs[i]), n)
                bounds = [("info", sb.info_len), ("title", sb.title_len), ("quality", sb.qual_len), ("dna", sb.dna_len)]
                acc = 0
                where = "?"
                for nm, ln in bounds:
                    if first < acc + ln:
                        where = f"{nm}+{first - acc}"
                        break
                    acc += ln
                print(f"     rank {rk.rank} sub-block {k}: len {len(mine)} vs ref {len(theirs)}; first diff at byte {first} ({where}); "
                      f"lens info={sb.info_len} title={sb.title_len} qual={sb.qual_len} dna={sb.dna_len} n_rec={sb.n_records} "
                      f"nsym={sb.nsym} nqual={sb.nqual} flags={sb.flags} nf={sb.n_fields} maxq={sb.max_qlen}")
                lo = max(0, first - 8)
                print(f"       mine  : {mine[lo:first + 24].hex()}")
                print(f"       theirs: {theirs[lo:first + 24].hex()}")
                try:   # what the two streams decode to
                    ta, tb = orc.decode_subblock(mine)[0], orc.decode_subblock(theirs)[0]
                    m2 = min(len(ta), len(tb))
                    fd = next((i for i in range(m2) if ta[i] != tb[i]), m2)
                    a0 = ta.rfind(b"\n@", 0, fd) + 1
                    print(f"       decoded texts differ at byte {fd} of {len(ta)} / {len(tb)}; record {ta[:fd].count(bytes([10])) // 4}")
                    print(f"       mine  : {ta[a0:fd + 60]!r}")
                    print(f"       theirs: {tb[a0:fd + 60]!r}")
                except Exception as e:  # noqa: BLE001
                    print(f"       (decode: {e})")
                break
    except Exception as e:  # noqa: BLE001
        print(f"     (could not localise: {e})")
    return False


def main():
    eng = api.Engine(0)
    bad = 0
    which = sys.argv[1:] or ["small"]
    if "small" in which:
        for ng in sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "small", "*.ngsc"))):
            m = re.match(r"(.*)\.p(\d+)\.ngsc$", ng)
            with open(m.group(1) + ".fastq", "rb") as f:
                data = f.read()
            with open(ng, "rb") as f:
                gold = f.read()
            bad += not compare_case(eng, os.path.basename(m.group(1)), data, int(m.group(2)), gold)
    if "gen" in which:
        for nm, cfg, size, P in [("cfg1", fastq_gen.CFG1, 10 << 20, 2), ("cfg1", fastq_gen.CFG1, 10 << 20, 8),
                                 ("cfg3", fastq_gen.CFG3, 30_000_000, 4), ("cfg5", fastq_gen.CFG5, 30_000_000, 8),
                                 ("cfg2", fastq_gen.CFG2, 100_000_000, 8)]:
            data = fastq_gen.generate(cfg, target_bytes=size).tobytes()
            bad += not compare_case(eng, f"{nm}_{size}", data, P, None)
    for w in which:
        if w.startswith("fuzz:"):      # replay a seed of tools/fuzz_parity.py
            import random
            sys.path.insert(0, os.path.join(ROOT, "tools"))
            import fuzz_parity as fz
            seed = int(w[5:])
            d = fz.draw_dialect(random.Random(seed), 24)
            print("dialect", d)
            bad += not compare_case(eng, w, fz.build_file(d, seed), d["P"], None)
    print("FAILED" if bad else "ALL OK", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
