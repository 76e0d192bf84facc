"""N>1 path on CPU: world_size-2 gloo processes, each owning half of the virtual ranks, run the same
host code the multi-GPU path runs after compression (block split -> all-gather of block sizes ->
deterministic offsets -> pwrite -> footer).  The per-rank sub-block streams come from the reference's
golden file, standing in for the GPU output, so the test needs no GPU and checks the result against
the reference block for block."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from phyngsc_amd import api, distributed as ngd, ngsc_format as nf

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = os.path.join(ROOT, "tests", "golden", "small")


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, ngsc_path, out_path):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        with open(ngsc_path, "rb") as f:
            ref = f.read()
        foot = nf.read_footer(ref)
        P = foot.PS
        streams = nf.rank_streams(ref)
        lo, hi = api.ranks_of_process(P, world, rank)
        ov = [0] + foot.OV
        local = [ngd.LocalRank(r, [len(s) for s in streams[r]], ov[r],
                               np.frombuffer(b"".join(streams[r]), dtype=np.uint8)) for r in range(lo, hi)]
        info = ngd.write_container(out_path, P, foot.FS, local, dist=dist)
        assert info["n_blocks"] == foot.BS
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("name", ["err_2000rec.p4", "err_2000rec.p8", "title_text.p4", "n_mixed.p2"])
def test_two_process_container(tmp_path, name):
    ngsc_path = os.path.join(SMALL, name + ".ngsc")
    out_path = str(tmp_path / "out.ngsc")
    mp.spawn(_worker, args=(2, _free_port(), ngsc_path, out_path), nprocs=2, join=True)
    with open(ngsc_path, "rb") as f:
        ref = f.read()
    with open(out_path, "rb") as f:
        got = f.read()
    ok, msg = nf.equivalent(ref, got)
    assert ok, msg
    # deterministic order: round k = k-th block of every rank
    foot = nf.read_footer(got)
    per = {}
    for w in foot.CBO:
        per[w] = per.get(w, 0) + 1
    assert foot.CBO == [r for r, _ in api.round_robin_order([per.get(r, 0) for r in range(foot.PS)])]


def test_rank_partition():
    assert api.ranks_of_process(8, 2, 1) == (4, 8)
    assert api.ranks_of_process(16, 8, 3) == (6, 8)
    with pytest.raises(AssertionError):
        api.ranks_of_process(6, 4, 0)
    lo, hi = api.job_byte_range(1000_000, 8, 2, 4)
    assert lo == 2 * 125000 and hi == min(1000_000, 4 * 125000 + 500 + (1 << 16))


def test_rows_to_offsets_matches_python_order():
    """ngsc_rows_to_offsets (the prefix sum behind the team's RCCL all-gather) against api.file_offsets, on random block
    lists and file sizes that give one group, several groups and one rank per group."""
    import random
    rng = random.Random(7)
    for _ in range(200):
        P = rng.choice([2, 3, 8, 16, 64])
        fs = rng.choice([1000, 10 << 20, 600 << 20, 5 << 30, 50 << 30])
        sizes = [[rng.randrange(1, 8 << 20) for _ in range(rng.randrange(0, 6))] for _ in range(P)]
        meta = [(rng.randrange(0, 50), rng.randrange(-3, 500)) for _ in range(P)]
        rows = api.rows_from_sizes(sizes, meta, 3 + 6)
        assert api.sizes_from_rows(rows) == (sizes, meta)
        assert api.rows_to_offsets(rows, fs) == api.file_offsets(sizes, api.ranks_per_group(fs, P))
