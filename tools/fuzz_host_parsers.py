#!/usr/bin/env python3
"""Hostile inputs for the host-side entry points of libngsc_hip.so that read bytes somebody else wrote (no GPU needed):
ngsc_read_footer, ngsc_read_header, ngsc_rows_to_offsets, ngsc_plan_blocks.  Meant to run against a build of the host files with
AddressSanitizer + UBSan (tools/asan_host.sh builds one and runs this and the C-ABI tests of the CPU suite against it): a
finding is the sanitizer's report (the process dies), a return code is never one.  Also checks that what the parsers DO accept
is consistent: a footer that parses gives counts inside the capacities that were handed in, a header's lengths fit its fields.

    python tools/fuzz_host_parsers.py [--n 20000] [--seed 1]
"""
from __future__ import annotations

import argparse
import ctypes
import glob
import os
import random
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from phyngsc_amd import _lib  # noqa: E402


def mutate(rng: random.Random, b: bytes) -> bytes:
    a = bytearray(b)
    how = rng.randrange(6)
    if how == 0 and a:      # bit flips
        for _ in range(rng.randrange(1, 9)):
            i = rng.randrange(len(a))
            a[i] ^= 1 << rng.randrange(8)
    elif how == 1 and a:    # a byte run overwritten
        i = rng.randrange(len(a))
        for j in range(i, min(len(a), i + rng.randrange(1, 16))):
            a[j] = rng.randrange(256)
    elif how == 2:          # cut short
        a = a[:rng.randrange(len(a) + 1)]
    elif how == 3:          # all ones / all zeros in front (maximal field widths and counts)
        n = rng.randrange(1, 12)
        a[:n] = bytes([rng.choice([0, 255])]) * n
    elif how == 4:          # noise of the same length
        a = bytearray(rng.randrange(256) for _ in range(len(a)))
    else:                   # grown by noise
        a += bytes(rng.randrange(256) for _ in range(rng.randrange(1, 64)))
    return bytes(a)


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=20000)
    ap.add_argument("--seed", type=int, default=1)
    args = ap.parse_args()
    rng = random.Random(args.seed)
    L = _lib.lib()
    print(f"library: {_lib.LIB_PATH}")
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "small", "*.ngsc")))
    assert files, "no committed containers"
    seeds_f, seeds_h = [], []
    for p in files:
        with open(p, "rb") as f:
            d = f.read()
        flen = (d[-2] << 8) | d[-1]
        seeds_f.append(d[len(d) - 2 - flen:len(d) - 2])
        seeds_h.append((d[:320], 1 << rng.randrange(1, 7)))
    n_ok = {"footer": 0, "header": 0, "rows": 0, "plan": 0}
    info = _lib.FooterInfo()
    for it in range(args.n):
        # ---- footer ----
        fb = mutate(rng, rng.choice(seeds_f))
        buf = (ctypes.c_uint8 * max(len(fb), 1)).from_buffer_copy(fb.ljust(1, b"\0"))
        rc = L.ngsc_read_footer(buf, len(fb), ctypes.byref(info), None, None, 0, None, 0)
        if rc == 0:
            capr, capb = rng.choice([info.P, max(info.P - 1, 0), info.P + 3]), rng.choice([info.n_blocks, max(info.n_blocks - 1, 0), info.n_blocks + 3])
            ov = (ctypes.c_int32 * max(capr, 1))()
            lbs = (ctypes.c_uint32 * max(capr, 1))()
            cbo = (ctypes.c_uint32 * max(capb, 1))()
            rc2 = L.ngsc_read_footer(buf, len(fb), ctypes.byref(info), ov, lbs, capr, cbo, capb)
            if rc2 == 0:
                n_ok["footer"] += 1
                assert capr >= info.P and capb >= info.n_blocks
                assert all(x < info.P for x in list(cbo)[:info.n_blocks]), "block owner beyond P accepted"
            else:
                assert rc2 == -10 and (capr < info.P or capb < info.n_blocks), rc2
        # ---- block header ----
        hb, P = rng.choice(seeds_h)
        hb = mutate(rng, hb)
        P = rng.choice([P, 2, 3, 8, 64, 1000, 32767, 1 << 20])
        buf = (ctypes.c_uint8 * max(len(hb), 1)).from_buffer_copy(hb.ljust(1, b"\0"))
        got = _lib.BlockHeaderInfo()
        if L.ngsc_read_header(buf, len(hb), P, ctypes.byref(got)) == 0:
            n_ok["header"] += 1
            assert got.wrid < P and got.n_subblocks <= 63 and got.header_len < 4096 and got.bcss < 4
        # ---- rows -> offsets (what the team's all-gather hands every worker) ----
        Pr = rng.choice([1, 2, 3, 8, 64])
        row_len = rng.choice([4, 5, 9, 20])
        rows = (ctypes.c_uint64 * (Pr * row_len))()
        for i in range(Pr * row_len):
            rows[i] = rng.choice([0, 1, rng.randrange(0, row_len), rng.randrange(1 << 23), (1 << 64) - 1, (1 << 64) - 3, 1 << 40])
        offs = (ctypes.c_uint64 * (Pr * (row_len - 3)))()
        capc = rng.choice([0, 1, Pr * (row_len - 3)])
        cbo = (ctypes.c_uint32 * max(capc, 1))()
        nb, end = ctypes.c_uint32(), ctypes.c_uint64()
        fs = rng.choice([0, 1, 1000, 10 << 20, 5 << 30, (1 << 64) - 1])
        rc = L.ngsc_rows_to_offsets(rows, Pr, row_len, fs, offs, cbo if rng.random() < 0.7 else None, capc, ctypes.byref(nb), ctypes.byref(end))
        if rc == 0:
            n_ok["rows"] += 1
            assert nb.value <= Pr * (row_len - 3)
        # ---- block planner (sub-block sizes come from the device; a planner must survive any) ----
        nsb = rng.randrange(0, 200)
        sizes = (ctypes.c_uint32 * max(nsb, 1))()
        for i in range(nsb):
            sizes[i] = rng.choice([0, 1, rng.randrange(1 << 16), rng.randrange(1 << 23), (8 << 20) - rng.randrange(64), 8 << 20, (1 << 32) - 1])
        cap = rng.choice([0, 1, 4, 64, 400])
        blocks = (_lib.Block * max(cap, 1))()
        nbk = ctypes.c_uint32()
        Pp = rng.choice([2, 3, 8, 64, 32767])
        rc = L.ngsc_plan_blocks(Pp, rng.randrange(Pp), sizes, nsb, blocks, cap, ctypes.byref(nbk))
        if rc == 0:
            n_ok["plan"] += 1
            assert nbk.value <= cap
            for b in list(blocks)[:nbk.value]:
                assert b.header_len <= 288 and b.size == b.header_len + b.payload and b.size <= (8 << 20), (b.size, b.header_len, b.payload)
        if (it + 1) % 5000 == 0:
            print(f"{it + 1} rounds, accepted: {n_ok}", flush=True)
    print(f"done: {args.n} rounds of each entry point, accepted {n_ok}, no sanitizer report, no inconsistent acceptance")
    return 0


if __name__ == "__main__":
    sys.exit(main())
