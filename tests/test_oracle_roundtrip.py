"""Round trip: the decoder of oracle/ngsc_decode.c (test infrastructure) turns .ngsc files back into the
FASTQ text they were made from.  Run on the committed outputs of the UNMODIFIED reference binary this pins
the decoder to the reference's format (the reference itself has no decompressor, only unused Fetch*
fragments, tasks.cpp:625-1101); run on oracle output it is the encode -> decode property of SURVEY 8f-1."""
import glob
import os
import re

import numpy as np
import pytest

from oracle import binding as orc
from phyngsc_amd import fastq_gen, ngsc_format as nf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
# Colour-space files (cs_*) are not reversible: the reference overwrites the quality line with the decoded bases or the
# colours (phyNGSC.cpp:531) and stores bases instead of transitions, so their streams do not hold the input any more.
SMALL = sorted(p for p in glob.glob(os.path.join(GOLDEN, "small", "*.ngsc")) if not os.path.basename(p).startswith("cs_"))


@pytest.mark.parametrize("ngsc_path", SMALL, ids=[os.path.basename(p) for p in SMALL])
def test_reference_output_decodes_to_its_input(ngsc_path):
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    with open(m.group(1) + ".fastq", "rb") as f:
        want = f.read()
    with open(ngsc_path, "rb") as f:
        got = orc.decompress(f.read())
    assert got == want


@pytest.mark.parametrize("name,cfg,size,P", [
    ("cfg1", fastq_gen.CFG1, 6 << 20, 2),
    ("cfg2", fastq_gen.CFG2, 20_000_000, 8),
    ("cfg3_150bp", fastq_gen.CFG3, 20_000_000, 3),
    ("cfg5_mixed", fastq_gen.CFG5, 20_000_000, 5),
])
def test_oracle_round_trip(name, cfg, size, P):
    data = fastq_gen.generate(cfg, target_bytes=size)
    out, _ = orc.compress_file(data, P, threads=4)
    assert orc.decompress(out) == data.tobytes()


def test_every_subblock_is_self_delimiting():
    """The decoder consumes exactly the bytes the block headers (SBOL) attribute to each sub-block, and the
    record counts add up to the file's."""
    data = fastq_gen.generate(fastq_gen.CFG5, target_bytes=12_000_000)
    out, st = orc.compress_file(data, 4, threads=4)
    n = 0
    k = 0
    for w, subs in sorted(nf.rank_streams(out).items()):
        for sb in subs:
            text, used, nrec = orc.decode_subblock(sb)
            assert used == len(sb)
            assert text.count(b"\n") == 4 * nrec
            n += nrec
            k += 1
    assert k == st.n_subblocks
    assert n == int((data == 10).sum()) // 4


def test_truncated_or_corrupt_subblock_is_rejected():
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=200_000)
    sb, _ = orc.subblock(data)
    orc.decode_subblock(sb)
    for cut in (3, 17, len(sb) // 2, len(sb) - 1):
        with pytest.raises(orc.OracleError):
            orc.decode_subblock(sb[:cut])
    bad = bytearray(sb)
    bad[0:4] = b"\xff\xff\xff\xff"   # absurd record count
    with pytest.raises(orc.OracleError):
        orc.decode_subblock(bytes(bad))


def test_edge_cases_round_trip_on_the_oracle():
    """The crafted boundary inputs of tests/edge_cases.py (used by the GPU parity tests) are valid for the oracle and
    survive encode -> decode."""
    from tests.edge_cases import build
    for name, (data, P) in build().items():
        out, _ = orc.compress_file(data, P)
        assert orc.decompress(out) == data, name


@pytest.mark.parametrize("seed", range(0, 48))
def test_random_dialects_round_trip_on_the_oracle(seed):
    """Random FASTQ dialects (tests/dialects.py): whatever the oracle accepts decodes back to the input - to its
    records before the damage when the file's tail is damaged."""
    import random
    from tests.dialects import build_file, draw_dialect
    d = draw_dialect(random.Random(seed), 0.5)
    data = build_file(d, seed)
    try:
        out, _ = orc.compress_file(data, d["P"])
    except orc.OracleError as e:
        assert e.rc in (-1, -2, -5), (seed, e.rc)     # format / field count / numeric range
        return
    if d["vary_sep"]:
        return          # one separator per field position is kept: lossy by design of the format
    back = orc.decompress(out)
    if d["tail"] == "ok":
        assert back == data, seed
    else:
        assert data.startswith(back[:-1]), seed
