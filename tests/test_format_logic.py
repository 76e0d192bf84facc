"""CPU tests of the per-thread format logic shared with the HIP kernels (ngsc_common.h) against the
oracle: Huffman build + blob, title stream (header + 32-record groups), sub-block cut plan.
The harness feeds the same NGSC_HD helpers the kernels call; see tests/harness/harness.cpp."""
import ctypes
import glob
import os
import re
import struct

import numpy as np
import pytest

from oracle import binding as orc
from phyngsc_amd import ngsc_format as nf
from tests import harness

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FASTQS = sorted(glob.glob(os.path.join(GOLDEN, "small", "*.fastq")))


def _h_huffman(freq):
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    packed = np.zeros(f.size, dtype=np.uint32)
    blob = np.zeros(1024, dtype=np.uint8)
    mx = ctypes.c_uint32()
    bl = harness.lib().h_huffman(f.ctypes.data, f.size, packed.ctypes.data, blob.ctypes.data, ctypes.byref(mx))
    return packed, bytes(blob[:bl]), mx.value


@pytest.mark.parametrize("seed", range(40))
def test_huffman_matches_oracle(seed):
    rng = np.random.default_rng(seed)
    n = int(rng.choice([1, 2, 3, 4, 5, 17, 41, 64, 100, 256, 300, 512]))
    kind = seed % 4
    if kind == 0:
        freq = rng.integers(0, 1000, n)
    elif kind == 1:
        freq = (rng.integers(0, 3, n) == 0) * rng.integers(1, 50, n)       # many zeros
    elif kind == 2:
        freq = np.minimum(2 ** rng.integers(0, 17, n), 60000)               # skewed, many ties
    else:
        freq = np.zeros(n, dtype=np.int64)
        freq[rng.integers(0, n)] = 5                                        # a single present symbol
    o = orc.huffman(freq)
    packed, blob, mx = _h_huffman(freq)
    assert blob == o.blob
    assert np.array_equal(packed >> 5, o.code[:n]) and np.array_equal(packed & 31, o.length[:n])
    assert mx == int(o.length[:n].max())


@pytest.mark.parametrize("fastq", FASTQS, ids=[os.path.basename(p)[:-6] for p in FASTQS])
def test_title_stream_matches_oracle(fastq):
    """Whole file as one sub-block window: harness title stream == oracle title stream."""
    data = np.fromfile(fastq, dtype=np.uint8)
    te, se, _ = orc.record_index(data, wsize=data.size, overlap=0)
    sub, inf = orc.subblock(data, wsize=data.size, overlap=0)
    want = sub[inf.info_len:inf.info_len + inf.title_len]
    out = np.zeros(len(want) + 4096, dtype=np.uint8)
    n = harness.lib().h_title_stream(data.ctypes.data, data.size, te.ctypes.data, se.ctypes.data, te.size, 0,
                                     out.ctypes.data, out.size)
    assert n == len(want), f"length {n} vs {len(want)}"
    got = bytes(out[:n])
    if got != want:
        first = next(i for i in range(n) if got[i] != want[i])
        raise AssertionError(f"title stream differs at byte {first}: {got[first:first+16].hex()} vs {want[first:first+16].hex()}")


NGSC = sorted(glob.glob(os.path.join(GOLDEN, "small", "*.ngsc")))


@pytest.mark.parametrize("ngsc_path", NGSC, ids=[os.path.basename(p) for p in NGSC])
def test_cut_plan_matches_reference(ngsc_path):
    """plan_rank() reproduces the reference's sub-block cuts: records per sub-block per rank + footer OV."""
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    data = np.fromfile(m.group(1) + ".fastq", dtype=np.uint8)
    P = int(m.group(2))
    with open(ngsc_path, "rb") as f:
        ref = f.read()
    streams = nf.rank_streams(ref)
    want = [(r, struct.unpack(">I", sb[:4])[0]) for r in sorted(streams) for sb in streams[r]]
    out = np.zeros(4 * 4096, dtype=np.uint32)
    nc = harness.lib().h_plan_file(data.ctypes.data, data.size, P, out.ctypes.data, 4096)
    assert nc == len(want)
    got = [(int(out[4 * i]), int(out[4 * i + 2])) for i in range(nc)]
    assert got == want
    ov = {int(out[4 * i]): int(np.int32(out[4 * i + 3])) for i in range(nc)}
    assert [ov[r] for r in range(1, P)] == nf.read_footer(ref).OV
