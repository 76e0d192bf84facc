"""Multi-process container assembly: one process per GPU, virtual ranks sharded across processes.

The compression itself needs no communication: sub-blocks of different virtual ranks are independent.
The only exchange is where each block lands in the output file.  The reference gets that from a shared
MPI file pointer (arrival order) plus MPI_Gather / MPI_Gatherv of block counts and timestamps
(phyNGSC.cpp:875, 925, 964, 994).  Here every process all-gathers the block sizes of its ranks (a few
hundred bytes, one collective - RCCL over xGMI with the ``nccl`` backend, ``gloo`` in CPU tests),
derives the same deterministic block order and file offsets, ``pwrite``s its own blocks, and process 0
appends the footer.
"""
from __future__ import annotations

import os
from dataclasses import dataclass
from typing import List, Sequence

import numpy as np

from . import api


@dataclass
class LocalRank:
    """What a process knows about one of its virtual ranks after compression."""
    wrid: int
    subblock_sizes: List[int]
    overlap_used: int
    stream: np.ndarray  # uint8: the rank's sub-block bytes back to back (host copy)


def write_container(path: str, P: int, fastq_size: int, local: Sequence[LocalRank], dist=None, device=None) -> dict:
    """Writes this process' blocks (and, on process 0, the footer) of the shared output file.

    ``dist`` is ``torch.distributed`` (initialised) or None for a single process.  Returns a summary
    {'n_blocks', 'file_bytes', 'cbo'} identical on every process.
    """
    plans = [api.plan_blocks(P, lr.wrid, lr.subblock_sizes) for lr in local]
    sizes_local = [[b.size for b in pl] for pl in plans]
    meta_local = [(len(lr.subblock_sizes), lr.overlap_used) for lr in local]
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        sizes, meta = api.exchange_block_sizes(sizes_local, meta_local, dist, device=device)
        rank0 = dist.get_rank() == 0
    else:
        sizes, meta = sizes_local, meta_local
        rank0 = True
    assert len(sizes) == P, f"gathered {len(sizes)} ranks, expected {P}"
    # offsets: the library's prefix sum over the deterministic block order (ngsc_rows_to_offsets), the same function the
    # one-process team uses on the rows its RCCL all-gather delivers
    row_len = max((len(s) for s in sizes), default=0) + 3
    offsets, cbo, end = api.rows_to_offsets(api.rows_from_sizes(sizes, meta, row_len), fastq_size)
    assert (offsets, cbo, end) == api.file_offsets(sizes, api.ranks_per_group(fastq_size, P))
    if rank0:
        # create / truncate before anyone writes
        with open(path, "wb"):
            pass
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    fd = os.open(path, os.O_WRONLY)
    try:
        for lr, pl in zip(local, plans):
            for k, b in enumerate(pl):
                off = offsets[(lr.wrid, k)]
                os.pwrite(fd, bytes(b.header[:b.header_len]), off)
                os.pwrite(fd, lr.stream[b.stream_off:b.stream_off + b.payload].tobytes(), off + b.header_len)
        if rank0:
            lb = [s[-1] if s else 0 for s in sizes]
            footer = api.make_footer(P, fastq_size, sum(m[0] for m in meta), [m[1] for m in meta], cbo, lb)
            os.pwrite(fd, footer, end)
            end += len(footer)
    finally:
        os.close(fd)
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    return {"n_blocks": len(cbo), "file_bytes": end if rank0 else None, "cbo": cbo}


def local_ranks_from_result(res: api.DeviceResult) -> List[LocalRank]:
    """Host copies of every rank's stream out of a DeviceResult (one D2H copy)."""
    host = res.out[:res.total_bytes].cpu().numpy()
    out = []
    for rk in res.ranks:
        sizes = [res.subblocks[rk.first_subblock + k].size for k in range(rk.n_subblocks)]
        out.append(LocalRank(rk.rank, sizes, rk.overlap_used, host[rk.stream_off:rk.stream_off + rk.stream_bytes]))
    return out
