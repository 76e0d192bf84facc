"""Parity tests proper (-m gpu): the HIP engine, called through the C ABI of libngsc_hip.so, against
 (1) the committed outputs of the unmodified reference (tests/golden/small/*.ngsc),
 (2) reference manifests of generator-made inputs up to BASELINE's 1 GB configuration,
 (3) the oracle on further seeded inputs,
 (4) size-independent properties: rank-split invariance, repeatability, record conservation,
 plus the error behaviour of the boundary.  Bar: bit-exact (integer / byte work)."""
import ctypes
import glob
import json
import functools
import os
import re
import struct
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

torch = pytest.importorskip("torch")

from oracle import binding as orc  # noqa: E402
from phyngsc_amd import _lib, api, fastq_gen, ngsc_format as nf  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")
SMALL = sorted(glob.glob(os.path.join(GOLDEN, "small", "*.ngsc")))


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    e = api.Engine(0)
    yield e
    e.close()


def _gpu(eng, data, P):
    arr = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else data
    d_in = torch.from_numpy(np.ascontiguousarray(arr)).cuda()
    res = eng.compress_device(d_in, arr.size, P)
    return res, api.assemble_container(res, P, arr.size)


def test_native_library_is_loaded():
    L = _lib.lib()
    assert L.ngsc_version() == 100
    with open("/proc/self/maps") as f:
        assert "libngsc_hip.so" in f.read()


# (no committed reference output is beyond the engine any more: titles with more than 64 separators - title_fields_80 - run on the
# wide title kernels since round 4)
SMALL_OK = SMALL


@pytest.mark.parametrize("ngsc_path", SMALL, ids=[os.path.basename(p) for p in SMALL])
def test_reference_goldens(eng, ngsc_path):
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    with open(m.group(1) + ".fastq", "rb") as f:
        data = f.read()
    with open(ngsc_path, "rb") as f:
        ref = f.read()
    _, out = _gpu(eng, data, int(m.group(2)))
    ok, msg = nf.equivalent(ref, out)
    assert ok, msg


def _manifest_params():
    res = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "manifest_*.json"))):
        with open(p) as f:
            man = json.load(f)
        for P in sorted(man["ranks"], key=int):
            res.append(pytest.param(p, int(P), id=f"{os.path.basename(p)[9:-5]}-P{P}"))
    return res


@pytest.mark.parametrize("man_path,P", _manifest_params())
def test_reference_manifests(eng, man_path, P):
    """Generator-made inputs (10 MiB .. 1 GB): per-(WRID,k) block hashes + footer of the reference."""
    with open(man_path) as f:
        man = json.load(f)
    from tests.dialects import manifest_input
    data = manifest_input(man)
    _, out = _gpu(eng, data, P)
    got = nf.manifest(out)
    assert got["footer"] == man["ranks"][str(P)]["footer"]
    assert got["blocks"] == man["ranks"][str(P)]["blocks"]


ORACLE_CASES = [
    # (name, config, bytes, P)
    ("36bp_P3", fastq_gen.GenConfig(seed=11, style=0), 20_000_000, 3),
    ("150bp_N_P5", fastq_gen.GenConfig(seed=12, style=1, len_min=150, len_max=150, qual_model=1, n_rate_ppm=5000,
                                       n_lowq_pct=50), 40_000_000, 5),
    ("mixed_36_200_P8", fastq_gen.GenConfig(seed=13, style=2, len_min=36, len_max=200, qual_model=2, n_rate_ppm=3000,
                                            n_lowq_pct=90), 50_000_000, 8),
    ("short_20_60_P2", fastq_gen.GenConfig(seed=14, style=1, len_min=20, len_max=60, qual_model=0), 30_000_000, 2),
    ("long_300_P8", fastq_gen.GenConfig(seed=15, style=0, len_min=100, len_max=220, qual_model=2, n_rate_ppm=100000,
                                        n_lowq_pct=100), 20_000_000, 8),
]


@pytest.mark.parametrize("name,cfg,size,P", ORACLE_CASES, ids=[c[0] for c in ORACLE_CASES])
def test_against_oracle(eng, name, cfg, size, P):
    data = fastq_gen.generate(cfg, target_bytes=size)
    _, out = _gpu(eng, data, P)
    want, _ = orc.compress_file(data, P, threads=8)
    assert out == want


@pytest.mark.parametrize("name,cfg,size,P", ORACLE_CASES[:4], ids=[c[0] for c in ORACLE_CASES[:4]])
def test_no_dense_copies_against_oracle(eng, monkeypatch, name, cfg, size, P):
    """The default kernels without the chunk-major copies of bases and qualities (NGSC_SQ_DENSE=0: measure and pack read
    the file themselves, which is also what jobs whose copies would not fit do)."""
    data = fastq_gen.generate(cfg, target_bytes=size // 2)
    want, _ = orc.compress_file(data, P, threads=8)
    monkeypatch.setenv("NGSC_SQ_DENSE", "0")
    _, out = _gpu(eng, data, P)
    assert out == want, "NGSC_SQ_DENSE=0"


PATH_CASES = ["err_2000rec.p4", "n_mixed.p3", "n_keep.p2", "title_text.p4", "title_numeric.p3", "title_long.p2",
              "varlen.p5", "qual_wide.p2", "dna_1sym.p2", "qual_past_Q_with_N_36.p3", "qual_past_Q_with_N_100.p2"]


@pytest.mark.parametrize("sq_path", ["reg", "dense", "wave"])
def test_kernel_paths_agree(eng, monkeypatch, sq_path):
    """The sequence/quality kernels exist in two forms: register-streaming (one lane per record, the default for
    records up to 2 KB; "dense" = with the chunk-major copies the statistics kernel leaves for measure and pack, "reg" =
    measure and pack read the file themselves) and wave-per-record (records above 2 KB; forced here with
    NGSC_SQ_PATH=wave): every form must reproduce the reference."""
    if sq_path == "dense":
        monkeypatch.setenv("NGSC_SQ_DENSE", "1")
    elif sq_path == "reg":
        monkeypatch.setenv("NGSC_SQ_DENSE", "0")
    else:
        monkeypatch.setenv("NGSC_SQ_PATH", "wave")
    for name in PATH_CASES:
        stem, P = name.rsplit(".p", 1)
        with open(os.path.join(GOLDEN, "small", stem + ".fastq"), "rb") as f:
            data = f.read()
        with open(os.path.join(GOLDEN, "small", name + ".ngsc"), "rb") as f:
            ref = f.read()
        _, out = _gpu(eng, data, int(P))
        ok, msg = nf.equivalent(ref, out)
        assert ok, f"{name} (sq_path={sq_path}): {msg}"


SCHEDULE_SWITCHES = [("NGSC_OVERLAP", "0"), ("NGSC_NO_SPEC", "1"), ("NGSC_SQ_RECORDS_PER_WG", "256"), ("NGSC_SQ_RECORDS_PER_WG", "512")]


@pytest.mark.parametrize("switch", SCHEDULE_SWITCHES, ids=[f"{k}={v}" for k, v in SCHEDULE_SWITCHES])
def test_schedule_switches_agree(eng, monkeypatch, switch):
    """The forms of a step that are not a kernel choice by input: the serial schedule (what a per-kernel timing pass runs), every call
    reading the job's shape back (NGSC_NO_SPEC=1: the first call's form; later calls predict it and enqueue the step in one go), and
    other records-per-workgroup numbers of the sequence/quality kernels (small inputs otherwise never walk several tiles per workgroup)
    - each reproduces the reference on committed goldens and the oracle on generated 36-base, 150-base and mixed-length inputs (three
    calls each on one context: the later ones run on the first one's shape, arena and output hints)."""
    monkeypatch.setenv(*switch)
    for name in PATH_CASES:
        stem, P = name.rsplit(".p", 1)
        with open(os.path.join(GOLDEN, "small", stem + ".fastq"), "rb") as f:
            data = f.read()
        with open(os.path.join(GOLDEN, "small", name + ".ngsc"), "rb") as f:
            ref = f.read()
        _, out = _gpu(eng, data, int(P))
        ok, msg = nf.equivalent(ref, out)
        assert ok, f"{name} ({switch}): {msg}"
    for cfg, size, P in ((fastq_gen.CFG2, 24_000_000, 4), (fastq_gen.CFG3, 30_000_000, 3), (fastq_gen.CFG5_300, 20_000_000, 5)):
        data = fastq_gen.generate(cfg, target_bytes=size)
        want, _ = orc.compress_file(data, P, threads=8)
        for _ in range(3):
            _, out = _gpu(eng, data, P)
            assert out == want, (switch, cfg)


@pytest.mark.parametrize("title_path", ["reg", "swar"])
def test_title_paths_agree(eng, monkeypatch, title_path):
    """Titles of at most 64 bytes take the word-parallel title kernels (separator/digit masks from v_perm nibble
    tables, field-wise SWAR), longer ones the byte-serial kernels; NGSC_TITLE_PATH=reg forces the latter.  Both
    must reproduce the reference on every committed golden and on generated inputs."""
    monkeypatch.setenv("NGSC_TITLE_PATH", title_path)
    for ngsc_path in SMALL_OK:
        m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
        with open(m.group(1) + ".fastq", "rb") as f:
            data = f.read()
        with open(ngsc_path, "rb") as f:
            ref = f.read()
        _, out = _gpu(eng, data, int(m.group(2)))
        ok, msg = nf.equivalent(ref, out)
        assert ok, f"{os.path.basename(ngsc_path)} (title path {title_path}): {msg}"
    for cfg, size, P in ((fastq_gen.CFG2, 30_000_000, 4), (fastq_gen.CFG5, 30_000_000, 3), (fastq_gen.CFG1, 8 << 20, 2)):
        data = fastq_gen.generate(cfg, target_bytes=size)
        _, out = _gpu(eng, data, P)
        want, _ = orc.compress_file(data, P, threads=4)
        assert out == want, f"cfg seed {cfg.seed} (title path {title_path})"


def _byte_class_fastq() -> bytes:
    """Every byte value inside a title, with the separator count kept at three per title (a changing count is an
    error): separators split `1c2z`, everything else sits in the middle of `1c2` (numeric only for digits)."""
    seps = b" ._,=:/-#"
    recs = []
    for rep in range(3):
        for c in range(256):
            if c == 10:
                continue
            ch = bytes([c])
            title = b"@x." + (b"1" + ch + b"2z" if ch in seps else b"1" + ch + b"2.z")
            recs.append(title + b"\nACGTACGT\n+\nIIIIIIII\n")
    return b"".join(recs)


def test_title_separator_and_digit_classes(eng):
    """The nibble-table classifier must agree with the separator list " ._,=:/-#\\n" (phyNGSC.cpp:208) and with
    utils::is_num for all byte values (CR, NUL, high-bit bytes, ...)."""
    data = _byte_class_fastq()
    _, out = _gpu(eng, data, 2)
    want, _ = orc.compress_file(data, 2)
    assert out == want      # (no round trip here: the format keeps ONE separator byte per field position)


@pytest.mark.parametrize("lo,hi,size", [(300, 400, 40_000_000), (1200, 1500, 30_000_000)])
def test_long_records(eng, lo, hi, size):
    """300-400 bp reads (~850 B records: many 16-byte chunks per lane in the register-streaming kernels) and
    1200-1500 bp reads (~3 KB records: the wave-per-record kernels).  Both exceed the reference's 500-byte
    overlap, where it reads past its buffer (Q7), so the oracle, which sees the real bytes, is the comparison."""
    cfg = fastq_gen.GenConfig(seed=31, style=1, len_min=lo, len_max=hi, qual_model=2, n_rate_ppm=2000, n_lowq_pct=70)
    data = fastq_gen.generate(cfg, target_bytes=size)
    _, out = _gpu(eng, data, 4)
    want, _ = orc.compress_file(data, 4, threads=4)
    assert out == want
    assert orc.decompress(out) == data.tobytes()


def test_rank_split_invariance_and_repeatability(eng):
    """Compressing ranks [0,4) and [4,8) in separate calls (as two processes would) gives the same
    sub-block bytes as one whole-file call; a second identical call gives identical bytes."""
    data = fastq_gen.generate(fastq_gen.GenConfig(seed=21, style=0), target_bytes=120_000_000)
    P = 8
    d_in = torch.from_numpy(data).cuda()
    whole = eng.compress_device(d_in, data.size, P)
    whole_host = whole.out[:whole.total_bytes].cpu().numpy().copy()
    again = eng.compress_device(d_in, data.size, P)
    assert np.array_equal(again.out[:again.total_bytes].cpu().numpy(), whole_host)
    parts = []
    for lo, hi in ((0, 4), (4, 8)):
        b0, b1 = api.job_byte_range(data.size, P, lo, hi)
        # 16-byte aligned start for the device buffer
        b0a = b0 & ~15
        part = torch.from_numpy(data[b0a:b1].copy()).cuda()
        r = eng.compress_device(part, data.size, P, rank_lo=lo, rank_hi=hi, buf_off=b0a)
        parts.append(r.out[:r.total_bytes].cpu().numpy().copy())
        assert [rk.rank for rk in r.ranks] == list(range(lo, hi))
    assert np.array_equal(np.concatenate(parts), whole_host)
    # record conservation: every record of the file is in exactly one sub-block
    n_rec = int((data == 10).sum()) // 4
    assert sum(sb.n_records for sb in whole.subblocks) == n_rec


def test_many_ranks_in_one_call(eng):
    """Calls of more ranks than one workgroup of the first kernel takes (k_sync_first: a wave per rank, up to 64 ranks in the
    workgroup that also clears the step's small state; more: a grid of them behind a clearing launch of its own) and than one round
    of k_init_sb's cut slots: 65, 100 and 300 ranks over a file of 36-base reads, each twice (the second call is a predicted step)."""
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=12_000_000)
    for P in (64, 65, 100, 300):
        want, _ = orc.compress_file(data, P, threads=8)
        for _ in range(2):
            _, out = _gpu(eng, data, P)
            assert out == want, P


def test_rank_walked_in_parts(eng):
    """ngsc_compress_rank_part: a rank's region streamed through the engine in several buffers (the reference takes a
    region of any size through 8 MiB windows, phyNGSC.cpp:113-128, 168, 745-755; one engine job stays below 4 GiB).
    The sub-blocks of the parts, one after the other, are the sub-blocks of the whole-rank call."""
    data = fastq_gen.generate(fastq_gen.GenConfig(seed=23, style=1, len_min=60, len_max=110, qual_model=2, n_rate_ppm=1500,
                                                  n_lowq_pct=70), target_bytes=150_000_000)
    P = 3
    d_all = torch.from_numpy(data).cuda()
    whole = eng.compress_device(d_all, data.size, P)
    whole_host = whole.out[:whole.total_bytes].cpu().numpy().copy()
    for part_bytes in (20_000_000, 33_000_000):
        got = []
        for rank in range(P):
            st = _lib.RankPart()
            pos = api.region_bounds(data.size, P, rank)[0]
            sizes, ov = [], None
            guard = 0
            while not st.done:
                lo = pos & ~15
                hi = min(data.size, lo + part_bytes)
                part = torch.from_numpy(data[lo:hi].copy()).cuda()
                r = eng.compress_rank_part(part, data.size, P, rank, st, buf_off=lo)
                got.append(r.out[:r.total_bytes].cpu().numpy().copy())
                sizes += [sb.size for sb in r.subblocks]
                pos = st.next_pos
                guard += 1
                assert guard < 100
            rk = whole.ranks[rank]
            assert sizes == [whole.subblocks[rk.first_subblock + k].size for k in range(rk.n_subblocks)]
            assert st.overlap_used == rk.overlap_used
        assert np.array_equal(np.concatenate(got), whole_host), part_bytes


def test_full_size_properties_1gb(eng):
    """BASELINE configs[1] size (1 GB, 36 bp): structure checks that do not need the reference -
    sizes add up, every sub-block header carries its record count, total records conserved."""
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=1_000_000_000)
    d_in = torch.from_numpy(data).cuda()
    res = eng.compress_device(d_in, data.size, 8)
    host = res.out[:res.total_bytes].cpu().numpy()
    off = 0
    for sb in res.subblocks:
        assert sb.out_off == off
        n, maxq = struct.unpack(">II", host[off:off + 8].tobytes())
        assert n == sb.n_records and maxq == 36
        off += sb.size
    assert off == res.total_bytes
    assert sum(sb.n_records for sb in res.subblocks) == int((data == 10).sum()) // 4
    assert sum(sb.n_bases for sb in res.subblocks) == 36 * (int((data == 10).sum()) // 4)
    # encode -> decode round trip at full size: every sub-block decodes (oracle/ngsc_decode.c) to exactly the
    # bytes of the file it covers, in order, and to exactly its own length
    from concurrent.futures import ThreadPoolExecutor
    spans = [(sb.out_off, sb.size) for sb in res.subblocks]

    def dec(span):
        text, used, _ = orc.decode_subblock(host[span[0]:span[0] + span[1]])
        assert used == span[1]
        return text
    pos = 0
    with ThreadPoolExecutor(8) as ex:
        for text in ex.map(dec, spans):
            assert text == data[pos:pos + len(text)].tobytes(), f"round trip differs near byte {pos}"
            pos += len(text)
    assert pos == data.size


# ---- error behaviour of the boundary -----------------------------------------------------------------
def _expect_error(eng, data: bytes, P: int, code: int):
    arr = np.frombuffer(data, dtype=np.uint8)
    d_in = torch.from_numpy(arr.copy()).cuda()
    with pytest.raises(api.NgscError) as ei:
        eng.compress_device(d_in, arr.size, P)
    assert ei.value.code == code, ei.value


def _fq(n, title=lambda i: f"@r.{i}", seq="ACGTACGTAC", qual="IIIIIIIIII", plus="+"):
    return "".join(f"{title(i)}\n{seq}\n{plus}\n{qual}\n" for i in range(n)).encode()


def test_error_plus_line_repeats_title(eng):
    # README.md:129-151: '+title' records are invalid input for the reference; we report NGSC_ERR_FORMAT
    _expect_error(eng, _fq(400, plus="+r"), 2, -1)


def test_error_quality_length_mismatch(eng):
    _expect_error(eng, _fq(400, qual="IIIIIIIII"), 2, -1)


def test_error_last_record_without_quality_line(eng):
    # the file stops after the '+' of its last record (or after the read): the reference would read the quality past the
    # end of its buffer; oracle and engine refuse.  Two bare blank lines behind the last record are not such a record.
    good = _fq(400)
    for tail in (b"@r.400\nACGTACGTAC\n+", b"@r.400\nACGTACGTAC\n+\n", b"@r.400\nACGTACGTAC\n"):
        with pytest.raises(orc.OracleError):
            orc.compress_file(good + tail, 2)
        _expect_error(eng, good + tail, 2, -1)


def test_error_field_count_changes(eng):
    _expect_error(eng, _fq(400, title=lambda i: f"@r.{i}" if i % 7 else f"@r.{i}.x"), 2, -2)


def test_colorspace_reads(eng):
    """SOLiD colour-space reads (phyNGSC.cpp:473-547) take the GPU path too: decoded to bases on the device, stored as
    the reference stores them (the committed cs_* goldens are its outputs).  What the reference cannot decode - a byte
    that is not a colour - is refused."""
    data = _fq(400, seq="T012301230", qual="IIIIIIIIII")
    _, out = _gpu(eng, data, 2)
    want, _ = orc.compress_file(data, 2)
    assert out == want
    _expect_error(eng, _fq(400, seq="T0123A1230", qual="IIIIIIIIII"), 2, -1)


def test_titles_of_more_than_64_fields(eng):
    """The reference has no field limit (phyNGSC.cpp:342-423); a title of more than 64 separators takes the wide title kernels
    (found out by a first run that ends in the context's retry).  66 fields with varying / group-constant / constant numbers and a
    text field, 200 single-digit fields, and - on the same context afterwards - an ordinary file again; the way back through the GPU
    decoder as well.  (tests/golden/small/title_fields_80.* is the reference binary's own output for 80 fields.)"""
    def title66(i):
        return "@" + ":".join(("t" if (i // 50) % 2 else "uu") if k == 7 else str((i * (k + 1)) % 10 if k % 3 == 0 else ((i // 32) % 7 if k % 3 == 1 else k % 10))
                              for k in range(66))
    def title100(i):   # seventy varying numbers, then text fields whose first letter never changes (round-4 fuzz: seeds 880040 ...: the GPU
        # decoder's fast path for a text field's first positions was taken for fields beyond the 64 it knows them of)
        return "@" + ":".join(str((i * (k + 1)) % 10) if k < 70 else ("q" + "abc"[(i * k) % 3] if k < 80 else "z") for k in range(100))
    for data, P in ((_fq(3000, title=title66), 2),
                    (_fq(1500, title=title100), 2),
                    (_fq(1200, title=lambda i: "@" + "_".join(str((i >> (k % 11)) & 1) for k in range(200))), 3),
                    (_fq(900, title=lambda i: f"@plain.{i} x"), 2)):
        _, out = _gpu(eng, data, P)
        want, _ = orc.compress_file(data, P)
        assert out == want
        assert api.decompress_bytes(out, engine=eng) == data


def test_predicted_step_meets_what_its_probes_did_not_expect(eng):
    """A predicted step (the context's second job of a kind onwards) does not launch the title's histogram and tree kernels and the
    512-symbol tree instance when the context's last job gave them nothing to do; a job of the same size that does - a text field that
    varies, a Huffman-coded number, a quality alphabet of more than 64 symbols - raises ERR_SHAPE in the finalisation kernels and the
    step runs again with everything (engine.cpp: big_lds / hist_lds / ttree_lds).  Same bytes as the oracle either way, in any order."""
    n = 6000
    quiet = "".join(f"@ab.{i:06d} k:100\nACGTACGTACGTACGTACGT\n+\n{'I' * 20}\n" for i in range(n)).encode()
    wide_q = "".join(chr(40 + (i * 7 + j * 3) % 70) for i in range(n) for j in range(20))
    busy = "".join(f"@{'ab' if (i // 3) % 2 else 'cd'}.{i:06d} k:{100 + (i * i) % 50:03d}\nACGTACGTACGTACGTACGT\n+\n{wide_q[20 * i:20 * i + 20]}\n"
                   for i in range(n)).encode()
    assert len(quiet) == len(busy)
    want_quiet, _ = orc.compress_file(quiet, 4)
    want_busy, _ = orc.compress_file(busy, 4)
    for data, want in ((quiet, want_quiet), (quiet, want_quiet), (quiet, want_quiet), (busy, want_busy), (busy, want_busy),
                       (quiet, want_quiet), (busy, want_busy), (quiet, want_quiet), (quiet, want_quiet)):
        _, out = _gpu(eng, data, 4)
        assert out == want


def test_predicted_step_whose_output_outgrows_what_was_cleared(eng):
    """A predicted step has its output cleared early as far as the context's last job's reached (+ 1/64) and leaves the rest to the layout
    kernel: a little more (up to 1 MiB) is cleared by that kernel's last wave in passing, much more is a missed prediction (ERR_SHAPE on
    the device, the step runs again).  Jobs of one shape whose qualities - and so whose sub-blocks - grow a little, then a lot, and
    shrink again; the output buffer is full of ones before every call (the pack kernels OR their shared words into it)."""
    n = 60000
    rnd = np.random.default_rng(12)
    seqs = rnd.choice(np.frombuffer(b"ACGT", dtype=np.uint8), size=(n, 100))
    noise = rnd.choice(np.frombuffer(b"#+5:>CFI", dtype=np.uint8), size=(n, 100))

    def job(every):   # every k-th record with noisy qualities (0: none)
        q = np.full((n, 100), ord("I"), dtype=np.uint8)
        if every:
            q[::every] = noise[::every]
        return b"".join(b"@run7.%06d\n%s\n+\n%s\n" % (i, seqs[i].tobytes(), q[i].tobytes()) for i in range(n))

    quiet, some, all_ = job(0), job(16), job(1)
    assert len(quiet) == len(some) == len(all_)
    want = {id(d): orc.compress_file(d, 4)[0] for d in (quiet, some, all_)}
    out = torch.empty(len(quiet), dtype=torch.uint8, device="cuda")
    sizes = []
    for data in (quiet, quiet, some, quiet, all_, all_, quiet, some, some):
        out.fill_(0xFF)
        d_in = torch.from_numpy(np.frombuffer(data, dtype=np.uint8).copy()).cuda()
        res = eng.compress_device(d_in, len(data), 4, out=out)
        sizes.append(res.total_bytes)
        assert api.assemble_container(res, 4, len(data)) == want[id(data)]
    # the third job outgrew the early clearing by less than the layout kernel's limit, the fifth by more
    assert sizes[1] + sizes[1] // 64 < sizes[2] <= sizes[1] + sizes[1] // 64 + (1 << 20) - 64, sizes
    assert sizes[4] > sizes[3] + sizes[3] // 64 + (1 << 20), sizes


def test_error_numeric_overflow(eng):
    _expect_error(eng, _fq(400, title=lambda i: f"@r:{3_000_000_000 + i}"), 2, -5)


def test_error_output_capacity(eng):
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=2 << 20)
    d_in = torch.from_numpy(data).cuda()
    small = torch.empty(4096, dtype=torch.uint8, device="cuda")
    with pytest.raises(api.NgscError) as ei:
        eng.compress_device(d_in, data.size, 2, out=small)
    assert ei.value.code == -10
    # the engine still works afterwards and reports the size that was needed
    res = eng.compress_device(d_in, data.size, 2)
    assert res.total_bytes > 4096


def test_error_bad_arguments(eng):
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=1 << 20)
    d_in = torch.from_numpy(data).cuda()
    for kwargs in (dict(P=1), dict(P=4, rank_lo=3, rank_hi=2), dict(P=4, rank_lo=0, rank_hi=5)):
        with pytest.raises(api.NgscError) as ei:
            eng.compress_device(d_in, data.size, **kwargs)
        assert ei.value.code == -7


def test_cli_file_to_file(tmp_path):
    """`ngsc <in.fastq> <out.ngsc>`: the command line the reference exposes (phyNGSC.cpp:61-108)."""
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=12_000_000)
    inp, outp = tmp_path / "in.fastq", tmp_path / "out.ngsc"
    data.tofile(inp)
    exe = os.path.join(ROOT, "phyngsc_amd", "bin", "ngsc")
    r = subprocess.run([exe, str(inp), str(outp), "1", "--ranks", "4"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    want, _ = orc.compress_file(data, 4, threads=4)
    assert outp.read_bytes() == want
    # usage error -> exit code 1 like the reference
    assert subprocess.run([exe], capture_output=True).returncode == 1
    assert subprocess.run([exe, "/nonexistent.fastq", str(outp)], capture_output=True).returncode == 2


def test_multi_context_file_driver(tmp_path, monkeypatch):
    """ngsc_compress_file_multi: several contexts, one host thread each, write the same file as one context -- the
    single-process multi-GPU driver (phyNGSC.cpp:113-116 rank split).  The contexts sit on distinct devices when the
    node has more than one (hipSetDevice per thread, per-device streams, pinned staging seen by every device);
    on a one-GPU box they share device 0."""
    data = fastq_gen.generate(fastq_gen.CFG5, target_bytes=30_000_000)
    inp = tmp_path / "in.fastq"
    data.tofile(inp)
    want, _ = orc.compress_file(data, 8, threads=8)
    ndev = torch.cuda.device_count()
    dev = lambda i: i % ndev   # noqa: E731
    e0, e1 = api.Engine(dev(0)), api.Engine(dev(1))
    assert [e0.device, e1.device] == [0, 1 if ndev > 1 else 0]
    for engines, name in (([e0], "one"), ([e0, e1], "two"), ([e0, e1, api.Engine(dev(2)), api.Engine(dev(3))], "four")):
        outp = tmp_path / f"{name}.ngsc"
        api.compress_file_multi(engines, str(inp), str(outp), 8)
        assert outp.read_bytes() == want, name
    # files larger than one engine job (4 GiB) are walked in jobs of consecutive ranks: force tiny jobs here
    monkeypatch.setenv("NGSC_JOB_BYTES", str(9_000_000))   # a rank is 3.75 MB: two ranks per job
    for engines, name in (([e0], "one_jobs"), ([e0, e1], "two_jobs")):
        outp = tmp_path / f"{name}.ngsc"
        api.compress_file_multi(engines, str(inp), str(outp), 8)
        assert outp.read_bytes() == want, name
    monkeypatch.delenv("NGSC_JOB_BYTES")
    with pytest.raises(api.NgscError) as ei:      # 8 ranks cannot be dealt to 3 contexts
        api.compress_file_multi([e0, e1, api.Engine(0)], str(inp), str(tmp_path / "x.ngsc"), 8)
    assert ei.value.code == -7
    exe = os.path.join(ROOT, "phyngsc_amd", "bin", "ngsc")
    outp = tmp_path / "cli.ngsc"
    r = subprocess.run([exe, str(inp), str(outp), "--ranks", "8", "--devices", "0,0"], capture_output=True, text=True,
                       timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert outp.read_bytes() == want
    assert subprocess.run([exe, str(inp), str(outp), "--ranks", "8", "--devices", "0,0,0"],
                          capture_output=True).returncode == 1


@pytest.mark.timeout(300)
def test_file_driver_worker_allocation_failure_is_an_error_not_a_hang(tmp_path, monkeypatch):
    """A worker whose staging buffers cannot be allocated (forced here) must end the whole job with an error code: its
    reader / writer threads never start, so nothing would end its final wait, and the other workers would wait for its
    groups' turn for ever."""
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=12_000_000)
    inp = tmp_path / "in.fastq"
    data.tofile(inp)
    e0, e1 = api.Engine(0), api.Engine(0)
    monkeypatch.setenv("NGSC_TEST_FAIL_ALLOC", "0")   # (a file of this size is one group of the block order: worker 0's)
    for engines in ([e0], [e0, e1]):
        with pytest.raises(api.NgscError) as ei:
            api.compress_file_multi(engines, str(inp), str(tmp_path / "x.ngsc"), 8)
        assert ei.value.code in (-6, -11)
    monkeypatch.delenv("NGSC_TEST_FAIL_ALLOC")
    api.compress_file_multi([e0, e1], str(inp), str(tmp_path / "ok.ngsc"), 8)
    want, _ = orc.compress_file(data, 8, threads=8)
    assert (tmp_path / "ok.ngsc").read_bytes() == want


@pytest.mark.timeout(300)
def test_exchange_member_failure_is_an_error_not_a_hang(tmp_path, monkeypatch):
    """The multi-GPU driver agrees on block offsets through the team's collective (RCCL over the contexts' communicator when
    they sit on distinct GPUs, host memory otherwise).  A member that fails in front of the collective (forced here:
    NGSC_TEST_FAIL_MEMBER) votes its error in host memory first, so nobody enters the collective and every worker - and the call
    - returns the code instead of waiting for the missing rank for ever; the bare collective behaves the same."""
    data = fastq_gen.generate(fastq_gen.CFG2, target_bytes=12_000_000)
    inp = tmp_path / "in.fastq"
    data.tofile(inp)
    ndev = torch.cuda.device_count()
    e0, e1 = api.Engine(0), api.Engine(1 % ndev)
    monkeypatch.setenv("NGSC_TEST_FAIL_MEMBER", "1")
    with pytest.raises(api.NgscError) as ei:
        api.compress_file_multi([e0, e1], str(inp), str(tmp_path / "x.ngsc"), 8)
    assert ei.value.code == -6
    team = api.Team([e0, e1])
    with pytest.raises(api.NgscError) as ei:
        team.allgather_u64(np.array([[1, 2], [3, 4]], dtype=np.uint64))
    assert ei.value.code == -6
    monkeypatch.delenv("NGSC_TEST_FAIL_MEMBER")
    assert team.allgather_u64(np.array([[1, 2], [3, 4]], dtype=np.uint64)).tolist() == [[1, 2, 3, 4], [1, 2, 3, 4]]
    team.close()
    api.compress_file_multi([e0, e1], str(inp), str(tmp_path / "ok.ngsc"), 8)
    want, _ = orc.compress_file(data, 8, threads=8)
    assert (tmp_path / "ok.ngsc").read_bytes() == want


def test_file_driver_parts_groups_and_streamed_writes(tmp_path, monkeypatch):
    """The file driver beyond one group (ngsc.h: NGSC_ORDER_GROUP_BYTES): (a) a rank larger than a job is walked in parts
    (ngsc_compress_rank_part) - inside a multi-rank group its stream is collected, (b) as a group of its own (region above
    512 MiB) its blocks are written as they fill; (c) groups dealt to two contexts in turn are written at their places.
    Every file is byte-identical to the oracle's, which orders the blocks by the same group rule."""
    ndev = torch.cuda.device_count()
    data = fastq_gen.generate(fastq_gen.CFG3, target_bytes=60_000_000)
    inp = tmp_path / "a.fastq"
    data.tofile(inp)
    want, _ = orc.compress_file(data, 2, threads=2)
    monkeypatch.setenv("NGSC_JOB_BYTES", str(20_000_000))   # a rank is 30 MB: two parts each
    outp = tmp_path / "a.ngsc"
    api.compress_file_multi([api.Engine(0)], str(inp), str(outp), 2)
    assert outp.read_bytes() == want
    monkeypatch.delenv("NGSC_JOB_BYTES")
    del data, want
    big = fastq_gen.generate(fastq_gen.CFG2, target_bytes=1_300_000_000)
    inp = tmp_path / "b.fastq"
    big.tofile(inp)
    for P, engines in ((2, [api.Engine(0)]), (4, [api.Engine(0), api.Engine(1 % ndev)])):
        want, _ = orc.compress_file(big, P, threads=8)
        foot = nf.read_footer(want)
        if P == 2:   # two groups of one rank: all of rank 0's blocks, then all of rank 1's
            assert foot.CBO == sorted(foot.CBO) and foot.CBO[0] == 0 and foot.CBO[-1] == 1
        outp = tmp_path / f"b{P}.ngsc"
        api.compress_file_multi(engines, str(inp), str(outp), P)
        with open(outp, "rb") as f:
            got = f.read()
        assert got == want, P
        del got, want
        outp.unlink()


REVERSIBLE = [p for p in SMALL_OK if not os.path.basename(p).startswith("cs_")]   # (colour space is stored lossily: phyNGSC.cpp:531)


@pytest.mark.parametrize("ngsc_path", REVERSIBLE[::3], ids=[os.path.basename(p) for p in REVERSIBLE[::3]])
def test_round_trip_goldens(eng, ngsc_path):
    """GPU output -> CPU decoder -> the original FASTQ (decoder pinned on the reference's own outputs in
    tests/test_oracle_roundtrip.py)."""
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    with open(m.group(1) + ".fastq", "rb") as f:
        data = f.read()
    _, out = _gpu(eng, data, int(m.group(2)))
    assert orc.decompress(out) == data


@pytest.mark.parametrize("cfg,size,P", [(fastq_gen.CFG3, 60_000_000, 8), (fastq_gen.CFG5, 40_000_000, 6),
                                        (fastq_gen.CFG5_300, 40_000_000, 8)])
def test_round_trip_generated(eng, cfg, size, P):
    data = fastq_gen.generate(cfg, target_bytes=size)
    _, out = _gpu(eng, data, P)
    assert orc.decompress(out) == data.tobytes()


def test_cfg5_as_stated_36_300bp(eng):
    """BASELINE configs[4] as stated: 36-300 bp mix, skewed quality alphabet, 0.2 % N.  Against the oracle at several
    rank counts (the P = 8 run is also pinned by the reference manifest manifest_cfg5_36_300_60MB.json)."""
    data = fastq_gen.generate(fastq_gen.CFG5_300, target_bytes=80_000_000)
    for P in (2, 8, 16):
        _, out = _gpu(eng, data, P)
        want, _ = orc.compress_file(data, P, threads=8)
        assert out == want, P


def test_team_in_process_rccl_and_host_exchange(eng):
    """The single-process multi-GPU host (ngsc_team_*): a team over ONE context has a one-rank RCCL communicator created in
    this process (ncclCommInitAll) and its all-gather carries the block rows; a team whose contexts share a device cannot
    have one (RCCL wants a GPU per rank) and says so - the rows then meet in host memory.  Either way every member
    ends up with the rows the block planner gives for the engine's own result, and the offsets follow from them."""
    data = fastq_gen.generate(fastq_gen.CFG5, target_bytes=24_000_000)
    P = 8
    d_in = torch.from_numpy(data).cuda()
    res = eng.compress_device(d_in, data.size, P)
    sizes = [[b.size for b in api.plan_blocks(P, rk.rank, [res.subblocks[rk.first_subblock + k].size for k in range(rk.n_subblocks)])]
             for rk in res.ranks]
    meta = [(rk.n_subblocks, rk.overlap_used) for rk in res.ranks]
    want_rows = api.rows_from_sizes(sizes, meta, 8)
    ref_out = res.out[:res.total_bytes].cpu().numpy().copy()
    # one member: RCCL with one rank
    t1 = api.Team([eng])
    assert t1.rccl_ranks == 1, t1.describe
    assert "ncclCommInitAll" in t1.describe
    out1 = torch.empty_like(res.out)
    rows, ms = t1.compress([(d_in, 0, 0, P, out1)], data.size, P, 8)
    assert (rows == want_rows).all() and len(ms) == 1
    assert (out1[:res.total_bytes].cpu().numpy() == ref_out).all()
    send = np.arange(5, dtype=np.uint64).reshape(1, 5) + 7
    assert (t1.allgather_u64(send) == send).all()
    t1.close()
    # two members on the same device: no RCCL, host exchange, same rows; each member compresses its half of the ranks
    e2 = api.Engine(0)
    t2 = api.Team([eng, e2])
    assert t2.rccl_ranks == 0 and "share a device" in t2.describe
    lo0, hi0 = api.job_byte_range(data.size, P, 0, 4)
    lo1, hi1 = api.job_byte_range(data.size, P, 4, 8)
    lo1 &= ~15
    a_in, b_in = torch.from_numpy(data[lo0:hi0].copy()).cuda(), torch.from_numpy(data[lo1:hi1].copy()).cuda()
    a_out, b_out = torch.empty_like(res.out), torch.empty_like(res.out)
    rows2, ms2 = t2.compress([(a_in, lo0, 0, 4, a_out), (b_in, lo1, 4, 8, b_out)], data.size, P, 8)
    assert (rows2 == want_rows).all() and len(ms2) == 2
    send = np.arange(6, dtype=np.uint64).reshape(2, 3)
    got = t2.allgather_u64(send)
    assert (got[0] == send.reshape(-1)).all() and (got[1] == send.reshape(-1)).all()
    table, cbo, end = api.rows_to_offsets(rows2, data.size)
    assert (table, cbo, end) == api.file_offsets(sizes, api.ranks_per_group(data.size, P))
    with pytest.raises(api.NgscError):      # row too short for the rank's blocks
        t2.compress([(a_in, lo0, 0, 4, a_out), (b_in, lo1, 4, 8, b_out)], data.size, P, 4 if max(len(s) for s in sizes) > 1 else 3)
    t2.close()


def test_bench_multi_gpu_modes_start_without_a_launcher():
    """`python bench.py --gpus N` needs no external launcher: one process, a team of N contexts on N host threads inside the
    library (here in rehearsal form when the box has a single GPU: all contexts on device 0, where RCCL cannot give every
    rank a GPU and the line must say rccl_ranks 0)."""
    import sys
    env = dict(os.environ)
    if torch.cuda.device_count() < 2:
        env["NGSC_BENCH_REHEARSE"] = "1"
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--size-mb", "64", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["value"] > 0 and line["config"]["virtual_ranks"] == 16
    assert "team of 2" in line["config"]["launch"]
    if torch.cuda.device_count() < 2:
        assert line["config"]["rccl_ranks"] == 0 and "share a device" in line["config"]["collective"]
    else:
        assert line["config"]["rccl_ranks"] == 2, line["config"]["collective"]


@pytest.mark.parametrize("ngsc_path", REVERSIBLE, ids=[os.path.basename(p) for p in REVERSIBLE])
def test_gpu_decoder_on_reference_outputs(eng, ngsc_path):
    """The GPU decoder (ngsc_decompress_device) turns the committed outputs of the UNMODIFIED reference back into the
    FASTQ files they were made from - the same pin the CPU decoder of oracle/ has."""
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    with open(m.group(1) + ".fastq", "rb") as f:
        want = f.read()
    with open(ngsc_path, "rb") as f:
        got = api.decompress_bytes(f.read(), engine=eng)
    assert got == want


@pytest.mark.parametrize("cfg,size,P", [(fastq_gen.CFG2, 120_000_000, 8), (fastq_gen.CFG3, 60_000_000, 4), (fastq_gen.CFG5_300, 40_000_000, 6)])
def test_gpu_round_trip(eng, cfg, size, P):
    """compress on the GPU -> decompress on the GPU -> the input; sub-block by sub-block the GPU decoder agrees with the
    CPU decoder of oracle/."""
    data = fastq_gen.generate(cfg, target_bytes=size)
    res, out = _gpu(eng, data, P)
    assert api.decompress_bytes(out, engine=eng) == data.tobytes()
    offs = [sb.out_off for sb in res.subblocks]
    lens = [sb.size for sb in res.subblocks]
    text, toff, nrec = eng.decompress_device(res.out, offs, lens)
    host = text.cpu().numpy()
    sub = res.out[:res.total_bytes].cpu().numpy()
    assert nrec == [sb.n_records for sb in res.subblocks]
    for k in (0, len(offs) // 2, len(offs) - 1):
        want, used, _ = orc.decode_subblock(sub[offs[k]:offs[k] + lens[k]])
        assert used == lens[k] and host[toff[k]:toff[k + 1]].tobytes() == want


@pytest.mark.parametrize("ngsc_path", REVERSIBLE, ids=[os.path.basename(p) for p in REVERSIBLE])
def test_decompress_file_on_reference_outputs(eng, ngsc_path, tmp_path):
    """`ngsc -d` (ngsc_decompress_file: footer, block headers and sub-block pieces read back in C++) on the committed
    outputs of the unmodified reference - its block order is arrival order - gives the FASTQ files they were made from,
    to the byte (also for the file whose last line has no newline)."""
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    with open(m.group(1) + ".fastq", "rb") as f:
        want = f.read()
    outp = tmp_path / "back.fastq"
    eng.decompress_file(ngsc_path, str(outp))
    assert outp.read_bytes() == want


def test_decompress_file_shared_records_and_cli(eng, tmp_path):
    """(a) A record two ranks both stored (test_record_shared_by_two_ranks) comes back once: every rank's text is placed
    by the footer's sync distances.  (b) The command line both ways on a file of several groups' worth of blocks with
    sub-blocks that continue over block borders; a container of the wrong size is refused with exit code 4."""
    import random
    rnd = random.Random(5)
    recs = []
    for i in range(40):
        L = rnd.randint(36, 536)
        recs.append(b"@%d\n%s\n+\n%s\n" % (10 ** 9 + rnd.randrange(10 ** 9), bytes(rnd.choice(b"ACGT") for _ in range(L)),
                                            bytes(rnd.choice(b"(=") for _ in range(L))))
    data = b"".join(recs)
    for P in (8, 12, 16, 20, 24):
        _, out = _gpu(eng, data, P)
        (tmp_path / "s.ngsc").write_bytes(out)
        eng.decompress_file(str(tmp_path / "s.ngsc"), str(tmp_path / "s.fastq"))
        assert (tmp_path / "s.fastq").read_bytes() == data, P
    big = fastq_gen.generate(fastq_gen.CFG5_300, target_bytes=150_000_000)
    inp, mid, back = tmp_path / "a.fastq", tmp_path / "a.ngsc", tmp_path / "a.back"
    big.tofile(inp)
    exe = os.path.join(ROOT, "phyngsc_amd", "bin", "ngsc")
    assert subprocess.run([exe, str(inp), str(mid), "--ranks", "6"], capture_output=True).returncode == 0
    r = subprocess.run([exe, "-d", str(mid), str(back)], capture_output=True)
    assert r.returncode == 0, r.stderr
    assert back.read_bytes() == big.tobytes()
    # the same file in batches of three sub-blocks: reader, decoder and writer threads with two buffers between each pair
    back.unlink()
    r = subprocess.run([exe, "-d", str(mid), str(back)], capture_output=True, env=dict(os.environ, NGSC_DEC_BATCH="3", NGSC_VERBOSE="1"))
    assert r.returncode == 0, r.stderr
    assert b"batches" in r.stderr and b" in 1 batches" not in r.stderr, r.stderr
    assert back.read_bytes() == big.tobytes()
    raw = mid.read_bytes()
    mid.write_bytes(raw[:len(raw) // 2])          # (the format has no checksums: what can be refused is a broken structure)
    assert subprocess.run([exe, "-d", str(mid), str(back)], capture_output=True).returncode == 4
    mid.write_bytes(raw[:8 << 20] + raw[(8 << 20) + 5:])
    assert subprocess.run([exe, "-d", str(mid), str(back)], capture_output=True).returncode == 4


@pytest.mark.parametrize("switch", [("NGSC_DEC_DNA_PAR", "0"), ("NGSC_DEC_SPEC", "0")], ids=["serial_bases", "serial_qualities"])
def test_gpu_decoder_switches_agree(eng, monkeypatch, switch):
    """The decoder's serial forms - what a stream the parallel kernels give up on falls back to - in place of the parallel ones:
    Huffman-coded bases decoded by one wave, the quality stream decoded serially.  Each must give the input back (150-base reads
    with N bases: Huffman-coded bases, equally long reads; mixed lengths: no speculation anyway)."""
    monkeypatch.setenv(*switch)
    for cfg, size, P in ((fastq_gen.CFG3, 30_000_000, 3), (fastq_gen.CFG5_300, 12_000_000, 2), (fastq_gen.CFG2, 20_000_000, 2)):
        data = fastq_gen.generate(cfg, target_bytes=size)
        _, out = _gpu(eng, data, P)
        assert api.decompress_bytes(out, engine=eng) == data.tobytes(), (switch, cfg)


def test_gpu_decoder_refuses_garbage(eng):
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=2_000_000)
    res, _ = _gpu(eng, data, 2)
    sub = res.out[:res.total_bytes].clone()
    sub[40:4000] = 0xA5          # title header / body damaged
    with pytest.raises(api.NgscError):
        eng.decompress_device(sub, [res.subblocks[0].out_off], [res.subblocks[0].size])


@functools.lru_cache(maxsize=None)
def _edge_cases():   # (built once: every parametrised case used to rebuild all of them, two seconds each time)
    from tests.edge_cases import build
    return build()


@pytest.mark.parametrize("name", sorted(_edge_cases()))
def test_edge_cases(eng, name):
    """Inputs on the decision boundaries of the kernels (title length 64/65, empty and very short reads, 0..7 and
    all-N reads, quality bytes outside the LDS columns, IUPAC codes, records around 2 KB, lowercase bases):
    bit-exact against the oracle, and the output decodes to the input."""
    data, P = _edge_cases()[name]
    _, out = _gpu(eng, data, P)
    want, _ = orc.compress_file(data, P)
    assert out == want
    assert orc.decompress(out) == data


@pytest.mark.parametrize("seed", range(0, 64))
def test_random_dialects(eng, seed):
    """Random FASTQ dialects (tests/dialects.py; the longer runs are tools/fuzz_parity.py): the engine and the oracle
    produce the same container or both refuse the input."""
    import random
    from tests.dialects import build_file, draw_dialect
    d = draw_dialect(random.Random(seed), 3.0)
    data = build_file(d, seed)
    try:
        want, _ = orc.compress_file(data, d["P"], threads=8)
    except orc.OracleError:
        want = None
    try:
        _, out = _gpu(eng, data, d["P"])
    except api.NgscError as e:
        assert want is None, (seed, e)
        return
    assert want is not None, f"seed {seed}: the oracle refuses what the engine took"
    assert out == want, seed


def test_record_shared_by_two_ranks(eng):
    """A sub-block always takes its second record (phyNGSC.cpp:295-300): when a rank's last window holds one record
    before the overlap, that second record is also the first record of the next rank.  Both sub-blocks encode it, each
    against its own first record (the engine indexes the sub-block-dependent per-record arrays by record + sub-block)."""
    import random
    rnd = random.Random(5)
    recs = []
    for i in range(40):
        L = rnd.randint(36, 536)
        recs.append(b"@%d\n%s\n+\n%s\n" % (10 ** 9 + rnd.randrange(10 ** 9), bytes(rnd.choice(b"ACGT") for _ in range(L)),
                                            bytes(rnd.choice(b"(=") for _ in range(L))))
    data = b"".join(recs)
    shared = 0
    for P in (8, 12, 16, 20, 24):
        res, out = _gpu(eng, data, P)
        want, _ = orc.compress_file(data, P)
        assert out == want, P
        is_shared = sum(sb.n_records for sb in res.subblocks) - len(recs) > 0
        shared += is_shared
        # both public decode entry points place a rank's text at its stretch of the file: the shared record comes out once
        assert api.decompress_bytes(out, engine=eng) == data, (P, is_shared)
    assert shared, "no rank split of this input shares a record: the test input lost its point"


def test_numeric_title_values_beyond_32_bits_are_refused(eng):
    """utils::to_num is uint32 arithmetic (utils.h:117-125): 8888888888 becomes 298954296, 21474836480 becomes 0, and the
    reference goes on with those - its container holds the values modulo 2^32, not its input (checked against the reference
    binary; the oracle restates that and counts such values).  The engine refuses a VARYING numeric field with digits beyond
    int32 (NGSC_ERR_RANGE): whether the wrapped value lands below 2^31 or not, it never writes a container that loses data.
    A constant field of that size is stored once as text and is fine."""
    import random
    rnd = random.Random(4)
    def make(vals):
        return b"".join(b"@r.%d x %s y 99999999999\n%s\n+\n%s\n" % (i, vals[i % len(vals)], bytes(rnd.choice(b"ACGT") for _ in range(36)),
                                                                    bytes(rnd.choice(b"5?IJ") for _ in range(36))) for i in range(3000))
    fine = make([b"888888", b"12", b"1000000000"])      # (values and delta range within int32)
    _, out = _gpu(eng, fine, 2)
    want, _ = orc.compress_file(fine, 2)
    assert out == want
    for vals in ([b"8888888888", b"888888", b"12"], [b"21474836480", b"12"], [b"88888888888888", b"7"], [b"9999999999", b"1410000000"]):
        lossy = make(vals)
        ref_like, _ = orc.compress_file(lossy, 2, as_reference=True)     # what the reference writes ...
        assert orc.decompress(ref_like) != lossy                          # ... does not hold its input
        with pytest.raises(orc.OracleError) as ei:
            orc.compress_file(lossy, 2)
        assert ei.value.rc == -5
        _expect_error(eng, lossy, 2, -5)
    bad = make([b"3000000000", b"12"])      # wraps to nothing: a negative int32 for the reference, which then loops
    with pytest.raises(orc.OracleError):
        orc.compress_file(bad, 2, as_reference=True)
    _expect_error(eng, bad, 2, -5)
