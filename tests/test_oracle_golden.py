"""Pins the CPU oracle (oracle/ngsc_oracle.c) to the UNMODIFIED reference.

tests/golden/small/*.ngsc are outputs of the reference binary (oracle/_ref/phyNGSC, threads=1) on
the committed inputs; tests/golden/manifest_*.json are per-(WRID,k) block hashes of its output on
generator-made inputs (tests/golden/make_golden.py).  Equality is the SURVEY 5b definition: every
(WRID,k) block byte-identical, all footer scalars identical, block order valid.
"""
import glob
import json
import os
import re

import pytest

from oracle import binding as orc
from phyngsc_amd import fastq_gen, ngsc_format as nf

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SMALL = sorted(glob.glob(os.path.join(GOLDEN, "small", "*.ngsc")))


def test_small_goldens_present():
    assert len(SMALL) >= 25


@pytest.mark.parametrize("ngsc_path", SMALL, ids=[os.path.basename(p) for p in SMALL])
def test_oracle_matches_reference_small(ngsc_path):
    m = re.match(r"(.*)\.p(\d+)\.ngsc$", ngsc_path)
    fastq, P = m.group(1) + ".fastq", int(m.group(2))
    with open(fastq, "rb") as f:
        data = f.read()
    with open(ngsc_path, "rb") as f:
        ref = f.read()
    out, st = orc.compress_file(data, P, as_reference=True)
    ok, msg = nf.equivalent(ref, out)
    assert ok, msg
    # sub-block count and footer are also what the reference reports
    assert nf.read_footer(ref).SS == st.n_subblocks


def _manifests(max_bytes):
    res = []
    for p in sorted(glob.glob(os.path.join(GOLDEN, "manifest_*.json"))):
        with open(p) as f:
            man = json.load(f)
        if man["fastq_bytes"] <= max_bytes:
            for P in sorted(man["ranks"], key=int):
                res.append(pytest.param(p, int(P), id=f"{os.path.basename(p)[9:-5]}-P{P}"))
    return res


@pytest.mark.parametrize("man_path,P", _manifests(64 << 20))
def test_oracle_matches_reference_manifest(man_path, P):
    """Generator-made inputs (<= 64 MB here; the bigger manifests are used by the -m gpu suite)."""
    with open(man_path) as f:
        man = json.load(f)
    from tests.dialects import manifest_input
    data = manifest_input(man)
    out, _ = orc.compress_file(data, P, threads=4, as_reference=True)
    assert nf.manifest(out) == man["ranks"][str(P)]


def test_oracle_thread_invariance():
    data = fastq_gen.generate(fastq_gen.CFG1, target_bytes=3 << 20)
    a, _ = orc.compress_file(data, 4, threads=1)
    b, _ = orc.compress_file(data, 4, threads=4)
    assert a == b
