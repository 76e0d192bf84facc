"""CPU tests of the C-ABI library: it loads, exports every symbol include/ngsc.h declares, fails loudly
without a GPU, and its host-side container functions (block split, MakeHeader, MakeFooter) reproduce
the reference's bytes.  No GPU compute is attempted here."""
import ctypes
import glob
import os
import re

import numpy as np
import pytest

from oracle import binding as orc
from phyngsc_amd import _lib, api, ngsc_format as nf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")


def test_header_symbols_are_exported():
    with open(os.path.join(ROOT, "include", "ngsc.h")) as f:
        text = f.read()
    declared = set(re.findall(r"\b(ngsc_[a-z_0-9]+)\s*\(", text))
    assert declared, "no declarations parsed"
    L = _lib.lib()
    for name in sorted(declared):
        assert hasattr(L, name), f"{name} declared in include/ngsc.h but not exported"
    assert declared == set(_lib.EXPORTS)


def test_no_gpu_fails_loudly():
    """Without a HIP device the product path raises; it never computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(api.NgscError) as ei:
        api.Engine(0)
    assert ei.value.code == -9


def test_strerror():
    L = _lib.lib()
    assert b"FASTQ" in L.ngsc_strerror(-1)
    assert L.ngsc_strerror(0) == b"ok"


NGSC = sorted(glob.glob(os.path.join(GOLDEN, "small", "*.ngsc")))


@pytest.mark.parametrize("ngsc_path", NGSC, ids=[os.path.basename(p) for p in NGSC])
def test_container_from_reference_streams(ngsc_path):
    """Rebuild the reference's file from its own sub-block streams with ngsc_plan_blocks /
    ngsc_make_header / ngsc_make_footer: every block and the footer scalars must match."""
    with open(ngsc_path, "rb") as f:
        ref = f.read()
    foot = nf.read_footer(ref)
    P = foot.PS
    streams = nf.rank_streams(ref)
    rank_blocks, parts, cbo = [], [], []
    for r in range(P):
        rank_blocks.append(api.plan_blocks(P, r, [len(s) for s in streams[r]]))
    joined = {r: b"".join(streams[r]) for r in range(P)}
    for r, k in api.round_robin_order([len(b) for b in rank_blocks]):
        b = rank_blocks[r][k]
        parts.append(bytes(b.header[:b.header_len]) + joined[r][b.stream_off:b.stream_off + b.payload])
        cbo.append(r)
    lb = [bl[-1].size for bl in rank_blocks]
    footer = api.make_footer(P, foot.FS, foot.SS, [0] + foot.OV, cbo, lb)
    ok, msg = nf.equivalent(ref, b"".join(parts) + footer)
    assert ok, msg


def test_block_split_large_streams():
    """8 MiB block split with sub-blocks crossing block boundaries (LSBS/FSBS), against the oracle's
    whole-file output for a 100 MB input (its per-rank streams are taken from the oracle's own file)."""
    from phyngsc_amd import fastq_gen
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=100_000_000)
    ref, st = orc.compress_file(data, 2, threads=2)
    assert st.n_blocks >= 4
    streams = nf.rank_streams(ref)
    foot = nf.read_footer(ref)
    out, cbo, rb = [], [], []
    for r in range(2):
        rb.append(api.plan_blocks(2, r, [len(s) for s in streams[r]]))
    joined = {r: b"".join(streams[r]) for r in range(2)}
    for r, k in api.round_robin_order([len(b) for b in rb]):
        b = rb[r][k]
        out.append(bytes(b.header[:b.header_len]) + joined[r][b.stream_off:b.stream_off + b.payload])
        cbo.append(r)
    footer = api.make_footer(2, foot.FS, foot.SS, [0] + foot.OV, cbo, [bl[-1].size for bl in rb])
    assert b"".join(out) + footer == ref


def test_block_split_refuses_a_sub_block_longer_than_two_blocks():
    """BCSS marks a sub-block that continues over ONE block border (tasks.cpp:1179-1200).  One whose remainder would not fit
    the next block either - nothing the engine produces, 8 MiB of FASTQ text never compress to more than a block, but the
    planner is a public entry point - is refused instead of planned as a last block beyond 8 MiB (found by
    tools/fuzz_host_parsers.py)."""
    B = _lib.BLOCK_SIZE
    for sizes in ([(1 << 32) - 1], [1000, 3 * B], [B - 100, 2 * B + 5, 7]):
        with pytest.raises(api.NgscError) as e:
            api.plan_blocks(8, 1, sizes)
        assert e.value.code == -8   # NGSC_ERR_UNSUPPORTED
    # ... while a remainder that just fits is still planned, every block within 8 MiB
    blocks = api.plan_blocks(8, 1, [B - 100, B - 200, 5])
    assert all(b.size <= B and b.size == b.header_len + b.payload for b in blocks)
    assert sum(b.payload for b in blocks) == 2 * B - 295


def test_make_header_matches_oracle():
    rng = np.random.default_rng(5)
    L = _lib.lib()
    for _ in range(50):
        nosb = int(rng.integers(1, 20))
        sbol = rng.integers(1, 8 << 20, nosb).astype(np.uint32)
        beso = int(sbol.max()).bit_length()
        bewr = int(rng.integers(1, 6))
        wrid = int(rng.integers(0, 1 << bewr))
        bcss = int(rng.integers(0, 4))
        bhs = (bewr + 25 + beso * nosb + 7) // 8
        want = orc.make_header(wrid, bewr, bhs, beso, bcss, sbol)
        buf = (ctypes.c_uint8 * 512)()
        arr = (ctypes.c_uint32 * nosb)(*[int(x) for x in sbol])
        n = L.ngsc_make_header(buf, wrid, bewr, bhs, beso, bcss, arr, nosb)
        assert bytes(buf[:n]) == want


def test_make_footer_matches_oracle():
    rng = np.random.default_rng(6)
    for _ in range(50):
        P = int(rng.integers(2, 17))
        nb = int(rng.integers(P, 4 * P))
        cbo = rng.integers(0, P, nb)
        ov = [0] + [int(x) for x in rng.integers(1, 600, P - 1)]
        lb = [int(x) for x in rng.integers(1000, 8 << 20, P)]
        fs = int(rng.integers(1 << 20, 1 << 40))
        want = orc.make_footer(P, fs, 3 * nb, ov, cbo, lb)
        assert api.make_footer(P, fs, 3 * nb, ov, [int(x) for x in cbo], lb) == want


@pytest.mark.parametrize("ngsc_path", sorted(glob.glob(os.path.join(GOLDEN, "small", "*.ngsc"))), ids=os.path.basename)
def test_read_footer_and_headers_of_reference_outputs(ngsc_path):
    """The reading side of the container (ngsc_read_footer = ReadFooter tasks.cpp:1203-1292, ngsc_read_header = the block header
    of tasks.cpp:1179-1200 read back) on the committed outputs of the reference binary: same scalars, overlaps, block order,
    last-block sizes and sub-block lengths as the independent Python reader of phyngsc_amd/ngsc_format.py."""
    L = _lib.lib()
    with open(ngsc_path, "rb") as f:
        data = f.read()
    want = nf.read_footer(data)
    flen = (data[-2] << 8) | data[-1]
    foot = (ctypes.c_uint8 * flen).from_buffer_copy(data[len(data) - 2 - flen:len(data) - 2])
    info = _lib.FooterInfo()
    assert L.ngsc_read_footer(foot, flen, ctypes.byref(info), None, None, 0, None, 0) == 0
    assert (info.P, info.n_blocks, info.n_subblocks, info.lbes, info.fastq_size) == (want.PS, want.BS, want.SS, want.LBES, want.FS)
    ov = (ctypes.c_int32 * info.P)()
    lbs = (ctypes.c_uint32 * info.P)()
    cbo = (ctypes.c_uint32 * info.n_blocks)()
    assert L.ngsc_read_footer(foot, flen, ctypes.byref(info), ov, lbs, info.P, cbo, info.n_blocks) == 0
    assert list(ov)[1:] == list(want.OV) and list(cbo) == list(want.CBO)
    if not want.LBES:
        assert list(lbs) == list(want.LBS)
    assert L.ngsc_read_footer(foot, flen, ctypes.byref(info), ov, lbs, info.P - 1, cbo, info.n_blocks) == -10   # NGSC_ERR_CAPACITY
    assert L.ngsc_read_footer(foot, max(flen // 2, 1), ctypes.byref(info), None, None, 0, None, 0) != 0 or flen < 4
    _, blocks = nf.split_blocks(data)
    for w, blk in blocks:
        hdr = nf.read_block_header(blk, 0, want.PS)
        got = _lib.BlockHeaderInfo()
        buf = (ctypes.c_uint8 * len(blk)).from_buffer_copy(blk)
        assert L.ngsc_read_header(buf, len(blk), want.PS, ctypes.byref(got)) == 0
        assert (got.wrid, got.header_len, got.n_subblocks, got.bcss) == (w, hdr.BHS, hdr.NOSB, hdr.BCSS)
        assert list(got.sb_len)[:hdr.NOSB] == list(hdr.SBOL)


def test_hostile_inputs_to_the_host_side_readers():
    """tools/fuzz_host_parsers.py (the tool that runs under AddressSanitizer + UBSan in tools/asan_host.sh), a short run in the
    ordinary build: damaged footers, block headers, block-size rows and sub-block size lists are refused or read consistently
    (counts inside the capacities handed in, block owners below P, planned blocks within 8 MiB) and nothing crashes."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "fuzz_host_parsers.py"), "--n", "600", "--seed", "3"],
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "no inconsistent acceptance" in r.stdout
