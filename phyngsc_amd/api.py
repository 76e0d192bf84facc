"""Host-side mirror of the reference's interface for the sub-block compression path.

The reference is a compiled MPI program: ``mpiexec -np P ./phyNGSC in.fastq out.ngsc T``
(phyNGSC.cpp:61-108).  Its per-sub-block work (tasks.h:20-42) and container functions
(MakeHeader / MakeFooter, tasks.h:61-65) run here behind the C ABI of ``libngsc_hip.so``; this
module only moves buffers (torch tensors in HBM), splits virtual ranks over processes and assembles
the container.  Names follow the reference: working regions (ranks), sub-blocks, blocks, footer.

There is no CPU fallback: every compute entry point needs the HIP library and a GPU.
"""
from __future__ import annotations

import ctypes
from dataclasses import dataclass
from typing import List, Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Block, KernelTime, NgscError, Rank, RankPart, SubBlock

WINDOW = 1 << 23   # READ_BUFFER_SIZE, defs.h:20
OVERLAP = 500      # phyNGSC.cpp:48


def region_bounds(file_size: int, P: int, rank: int) -> Tuple[int, int]:
    """[wr_start, wr_end] of a virtual rank (phyNGSC.cpp:113-116), both inclusive."""
    region = file_size // P
    start = rank * region
    end = start + region + OVERLAP - 1 if rank != P - 1 else file_size - 1
    return start, end


def job_byte_range(file_size: int, P: int, rank_lo: int, rank_hi: int, slack: int = 1 << 16) -> Tuple[int, int]:
    """Byte range [lo, hi) of the file a process must hold to compress ranks [rank_lo, rank_hi)."""
    lo = region_bounds(file_size, P, rank_lo)[0]
    hi = min(file_size, region_bounds(file_size, P, rank_hi - 1)[1] + 1 + slack)
    return lo, hi


@dataclass
class DeviceResult:
    """Outcome of one ngsc_compress_device() call; the sub-block bytes stay in HBM (``out``)."""
    total_bytes: int
    subblocks: List[SubBlock]
    ranks: List[Rank]
    out: object  # torch.Tensor (uint8, cuda)


class Engine:
    """One GPU context (ngsc_ctx): grow-only HBM workspace + stream."""

    def __init__(self, device: int = 0):
        self._L = _lib.lib()
        h = ctypes.c_void_p()
        rc = self._L.ngsc_ctx_create(device, ctypes.byref(h))
        if rc:
            raise NgscError(rc, "no HIP device: this engine has no CPU fallback")
        self._h = h
        self.device = device

    def close(self):
        if getattr(self, "_h", None):
            self._L.ngsc_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- profiling -------------------------------------------------------------------------------
    def set_profiling(self, on: bool):
        self._L.ngsc_set_profiling(self._h, int(on))

    def profile_only(self, kernel_name):
        """Time just this kernel (None: all of them): every timed launch costs two event records (~10 us)."""
        self._L.ngsc_profile_only(self._h, kernel_name.encode() if kernel_name else None)

    def reset_kernel_times(self):
        self._L.ngsc_reset_kernel_times(self._h)

    def kernel_times(self) -> dict:
        n = self._L.ngsc_get_kernel_times(self._h, None, 0)
        arr = (KernelTime * max(n, 1))()
        n = self._L.ngsc_get_kernel_times(self._h, arr, n)
        return {arr[i].name.decode(): (arr[i].ms, arr[i].launches) for i in range(n)}

    # ---- the hot path ----------------------------------------------------------------------------
    def compress_device(self, d_in, file_size: int, P: int, rank_lo: int = 0, rank_hi: Optional[int] = None,
                        buf_off: int = 0, out=None, stream: Optional[int] = None) -> DeviceResult:
        """Compresses virtual ranks [rank_lo, rank_hi) of a P-rank run.

        d_in: uint8 CUDA tensor holding file bytes [buf_off, buf_off + d_in.numel()).
        out:  uint8 CUDA tensor for the sub-block bytes (allocated when None).
        """
        import torch
        if rank_hi is None:
            rank_hi = P
        assert d_in.is_cuda and d_in.dtype == torch.uint8 and d_in.is_contiguous()
        if out is None:
            out = torch.empty(int(d_in.numel() * 0.6) + (16 << 20), dtype=torch.uint8, device=d_in.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        rc = self._L.ngsc_compress_device(self._h, d_in.data_ptr(), d_in.numel(), buf_off, file_size, P, rank_lo,
                                          rank_hi, out.data_ptr(), out.numel(), ctypes.c_void_p(stream))
        if rc:
            raise NgscError(rc, self._L.ngsc_last_hip_error(self._h).decode())
        nsb = self._L.ngsc_result_num_subblocks(self._h)
        nrk = self._L.ngsc_result_num_ranks(self._h)
        sp, rp = self._L.ngsc_result_subblocks(self._h), self._L.ngsc_result_ranks(self._h)
        # one memmove per table (the context reuses its result storage on the next call)
        sbs = (SubBlock * nsb)()
        rks = (Rank * nrk)()
        if nsb:
            ctypes.memmove(sbs, sp, ctypes.sizeof(sbs))
        if nrk:
            ctypes.memmove(rks, rp, ctypes.sizeof(rks))
        return DeviceResult(int(self._L.ngsc_result_bytes(self._h)), sbs, rks, out)

    def compress_rank_part(self, d_in, file_size: int, P: int, rank: int, state: RankPart, buf_off: int, out=None,
                           stream: Optional[int] = None) -> DeviceResult:
        """One part of a rank that is walked in several calls (ngsc.h: ngsc_compress_rank_part): every window whose
        records are complete in d_in; ``state`` (zero-initialised before the first part) is updated in place."""
        import torch
        if out is None:
            out = torch.empty(int(d_in.numel() * 0.6) + (16 << 20), dtype=torch.uint8, device=d_in.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_in.device).cuda_stream
        rc = self._L.ngsc_compress_rank_part(self._h, d_in.data_ptr(), d_in.numel(), buf_off, file_size, P, rank,
                                             ctypes.byref(state), out.data_ptr(), out.numel(), ctypes.c_void_p(stream))
        if rc:
            raise NgscError(rc, self._L.ngsc_last_hip_error(self._h).decode())
        nsb = self._L.ngsc_result_num_subblocks(self._h)
        nrk = self._L.ngsc_result_num_ranks(self._h)
        sbs, rks = (SubBlock * nsb)(), (Rank * nrk)()
        if nsb:
            ctypes.memmove(sbs, self._L.ngsc_result_subblocks(self._h), ctypes.sizeof(sbs))
        if nrk:
            ctypes.memmove(rks, self._L.ngsc_result_ranks(self._h), ctypes.sizeof(rks))
        return DeviceResult(int(self._L.ngsc_result_bytes(self._h)), sbs, rks, out)

    def decompress_device(self, d_sub, offsets: Sequence[int], lengths: Sequence[int], text_cap: Optional[int] = None,
                          stream: Optional[int] = None):
        """The inverse path (ngsc.h: ngsc_decompress_device): sub-blocks in HBM -> FASTQ text in HBM.
        Returns (text tensor, text offsets per sub-block + total, records per sub-block)."""
        import torch
        n = len(offsets)
        offs = (ctypes.c_uint64 * n)(*offsets)
        lens = (ctypes.c_uint32 * n)(*lengths)
        toff = (ctypes.c_uint64 * (n + 1))()
        nrec = (ctypes.c_uint32 * n)()
        if text_cap is None:
            text_cap = n * (9 << 20)
        text = torch.empty(text_cap, dtype=torch.uint8, device=d_sub.device)
        if stream is None:
            stream = torch.cuda.current_stream(d_sub.device).cuda_stream
        rc = self._L.ngsc_decompress_device(self._h, d_sub.data_ptr(), offs, lens, n, text.data_ptr(), text_cap, toff, nrec,
                                            ctypes.c_void_p(stream))
        if rc:
            raise NgscError(rc, self._L.ngsc_last_hip_error(self._h).decode())
        return text[:toff[n]], list(toff), list(nrec)

    def compress_file(self, in_path: str, out_path: str, P: int = 8) -> Tuple[float, float, float]:
        """``<in.fastq> <out.ngsc>`` on this GPU; returns (read, H2D+device+D2H, plan+write) seconds."""
        sec = (ctypes.c_double * 3)()
        rc = self._L.ngsc_compress_file(self._h, in_path.encode(), out_path.encode(), P, sec)
        if rc:
            raise NgscError(rc, self._L.ngsc_last_hip_error(self._h).decode())
        return sec[0], sec[1], sec[2]


    def decompress_file(self, in_path: str, out_path: str) -> Tuple[float, float, float]:
        """``<in.ngsc> <out.fastq>`` (ngsc.h: ngsc_decompress_file); returns (read, H2D+decode+D2H, write) seconds."""
        sec = (ctypes.c_double * 3)()
        rc = self._L.ngsc_decompress_file(self._h, in_path.encode(), out_path.encode(), sec)
        if rc:
            raise NgscError(rc, self._L.ngsc_last_hip_error(self._h).decode())
        return sec[0], sec[1], sec[2]


def compress_file_multi(engines: Sequence["Engine"], in_path: str, out_path: str, P: int = 8) -> Tuple[float, float, float]:
    """One process, several contexts (one per GPU): ranks are dealt in runs of P/len(engines); the file is
    byte-identical to the single-context one (ngsc.h: ngsc_compress_file_multi)."""
    L = _lib.lib()
    hs = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
    sec = (ctypes.c_double * 3)()
    rc = L.ngsc_compress_file_multi(hs, len(engines), in_path.encode(), out_path.encode(), P, sec)
    if rc:
        raise NgscError(rc, "; ".join(L.ngsc_last_hip_error(e._h).decode() for e in engines))
    return sec[0], sec[1], sec[2]


class Team:
    """One process, several GPUs (ngsc.h: ngsc_team_*): one host thread per engine inside the library plus an RCCL communicator
    created in-process (ncclCommInitAll).  ``compress`` runs one step on every GPU at once and all-gathers the per-rank block
    rows; ``rccl_ranks`` is what the communicator reports (0: the rows met in host memory, ``describe`` says why)."""

    def __init__(self, engines: Sequence["Engine"]):
        self._L = _lib.lib()
        self.engines = list(engines)
        hs = (ctypes.c_void_p * len(engines))(*[e._h for e in engines])
        h = ctypes.c_void_p()
        rc = self._L.ngsc_team_create(hs, len(engines), ctypes.byref(h))
        if rc:
            raise NgscError(rc)
        self._h = h
        self.n = len(engines)
        self.rccl_ranks = int(self._L.ngsc_team_rccl_ranks(h))
        self.describe = self._L.ngsc_team_describe(h).decode()

    def close(self):
        if getattr(self, "_h", None):
            self._L.ngsc_team_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def compress(self, shards: Sequence[tuple], file_size: int, P: int, row_len: int):
        """shards[i] = (d_in tensor, buf_off, rank_lo, rank_hi, d_out tensor) on engine i's device.  Returns (rows, ms per member):
        rows[r] = [n_blocks, n_subblocks, overlap_used, block sizes ...] for every virtual rank r, as every GPU holds them after
        the all-gather."""
        arr = (_lib.Shard * self.n)()
        for i, (d_in, buf_off, lo, hi, d_out) in enumerate(shards):
            arr[i].d_buf, arr[i].buf_len, arr[i].buf_off = d_in.data_ptr(), d_in.numel(), buf_off
            arr[i].rank_lo, arr[i].rank_hi, arr[i].d_out, arr[i].out_cap = lo, hi, d_out.data_ptr(), d_out.numel()
        rows = (ctypes.c_uint64 * (P * row_len))()
        ms = (ctypes.c_double * self.n)()
        rc = self._L.ngsc_team_compress(self._h, arr, file_size, P, rows, row_len, ms)
        if rc:
            raise NgscError(rc, "; ".join(self._L.ngsc_last_hip_error(e._h).decode() for e in self.engines))
        return np.frombuffer(rows, dtype=np.uint64).reshape(P, row_len).copy(), list(ms)

    def exchange_ms(self):
        """Host time of every member in the exchange of the last step (includes waiting for the slowest member's compression)."""
        ms = (ctypes.c_double * self.n)()
        self._L.ngsc_team_exchange_ms(self._h, ms)
        return list(ms)

    def allgather_u64(self, send: np.ndarray) -> np.ndarray:
        """The bare collective: send[i] (count words) from member i; returns recv[i] = what member i received (n x count)."""
        send = np.ascontiguousarray(send, dtype=np.uint64)
        assert send.shape[0] == self.n
        count = int(send.size // self.n)
        recv = np.zeros((self.n, self.n * count), dtype=np.uint64)
        rc = self._L.ngsc_team_allgather_u64(self._h, send.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), count,
                                             recv.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)))
        if rc:
            raise NgscError(rc)
        return recv


def rows_from_sizes(sizes: Sequence[Sequence[int]], meta: Sequence[Tuple[int, int]], row_len: int) -> np.ndarray:
    """The exchange's row layout (ngsc.h): [n_blocks, n_subblocks, overlap_used, sizes ...] per virtual rank."""
    rows = np.zeros((len(sizes), row_len), dtype=np.uint64)
    for r, (s, m) in enumerate(zip(sizes, meta)):
        assert len(s) + 3 <= row_len
        rows[r, 0], rows[r, 1], rows[r, 2] = len(s), m[0], np.int64(m[1]).astype(np.uint64)
        rows[r, 3:3 + len(s)] = s
    return rows


def sizes_from_rows(rows: np.ndarray):
    """Inverse of rows_from_sizes: (sizes per rank, (n_subblocks, overlap_used) per rank)."""
    sizes = [[int(x) for x in row[3:3 + int(row[0])]] for row in rows]
    meta = [(int(row[1]), int(np.uint64(row[2]).astype(np.int64))) for row in rows]
    return sizes, meta


def rows_to_offsets(rows: np.ndarray, file_size: int):
    """ngsc_rows_to_offsets: {(rank, k): file offset}, cbo, end - the library's own prefix sum over the block order."""
    L = _lib.lib()
    rows = np.ascontiguousarray(rows, dtype=np.uint64)
    P, row_len = rows.shape
    offs = np.zeros((P, row_len - 3), dtype=np.uint64)
    cap = int(rows[:, 0].sum()) + 1
    cbo = (ctypes.c_uint32 * cap)()
    nb, end = ctypes.c_uint32(), ctypes.c_uint64()
    rc = L.ngsc_rows_to_offsets(rows.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), P, row_len, file_size,
                                offs.ctypes.data_as(ctypes.POINTER(ctypes.c_uint64)), cbo, cap, ctypes.byref(nb), ctypes.byref(end))
    if rc:
        raise NgscError(rc)
    table = {(r, k): int(offs[r, k]) for r in range(P) for k in range(int(rows[r, 0]))}
    return table, list(cbo[:nb.value]), int(end.value)


def _copy_struct(s):
    c = type(s)()
    ctypes.memmove(ctypes.byref(c), ctypes.byref(s), ctypes.sizeof(s))
    return c


# ---- container (host) -----------------------------------------------------------------------------
def plan_blocks(P: int, wrid: int, sb_sizes: Sequence[int]) -> List[Block]:
    """8 MiB block split of one rank's sub-block stream (phyNGSC.cpp:842-928)."""
    L = _lib.lib()
    n = len(sb_sizes)
    sizes = (ctypes.c_uint32 * max(n, 1))(*sb_sizes)
    cap = sum(sb_sizes) // (_lib.BLOCK_SIZE - 4096) + 4
    blocks = (Block * cap)()
    nb = ctypes.c_uint32()
    rc = L.ngsc_plan_blocks(P, wrid, sizes, n, blocks, cap, ctypes.byref(nb))
    if rc:
        raise NgscError(rc)
    return [_copy_struct(blocks[i]) for i in range(nb.value)]


def make_footer(P: int, fastq_size: int, n_subblocks: int, overlaps: Sequence[int], cbo: Sequence[int],
                lb_sizes: Sequence[int]) -> bytes:
    """MakeFooter (tasks.cpp:1104-1176); cbo = WRID of every block in physical order."""
    L = _lib.lib()
    ov = (ctypes.c_int32 * P)(*overlaps)
    cb = (ctypes.c_uint32 * max(len(cbo), 1))(*cbo)
    lb = (ctypes.c_uint32 * P)(*lb_sizes)
    out = (ctypes.c_uint8 * (256 + 4 * len(cbo) + 8 * P))()
    n = L.ngsc_make_footer(out, P, fastq_size, len(cbo), n_subblocks, ov, cb, lb)
    return bytes(out[:n])


ORDER_GROUP_BYTES = 512 << 20   # NGSC_ORDER_GROUP_BYTES (ngsc.h): part of the definition of the block order


def ranks_per_group(file_size: int, P: int) -> int:
    """Consecutive ranks whose regions make up at most 512 MiB of input form a group of the block order."""
    region = file_size // P
    return max(1, min(P, ORDER_GROUP_BYTES // region)) if region else P


def round_robin_order(blocks_per_rank: Sequence[int], ranks_per_grp: Optional[int] = None) -> List[Tuple[int, int]]:
    """Deterministic block order: rank groups in turn (``ranks_per_group``; one group when None); inside a group
    round k = k-th block of each of its ranks, ranks ascending."""
    P = len(blocks_per_rank)
    rpg = ranks_per_grp or max(P, 1)
    order = []
    for g0 in range(0, P, rpg):
        grp = range(g0, min(P, g0 + rpg))
        for k in range(max((blocks_per_rank[r] for r in grp), default=0)):
            for r in grp:
                if k < blocks_per_rank[r]:
                    order.append((r, k))
    return order


def assemble_container(res: DeviceResult, P: int, fastq_size: int) -> bytes:
    """Single-process assembly of a complete .ngsc file from a whole-file DeviceResult."""
    host = res.out[:res.total_bytes].cpu().numpy()
    assert len(res.ranks) == P
    rank_blocks = []
    for rk in res.ranks:
        sizes = [res.subblocks[rk.first_subblock + k].size for k in range(rk.n_subblocks)]
        rank_blocks.append(plan_blocks(P, rk.rank, sizes))
    parts = []
    cbo = []
    for r, k in round_robin_order([len(b) for b in rank_blocks], ranks_per_group(fastq_size, P)):
        b = rank_blocks[r][k]
        base = res.ranks[r].stream_off + b.stream_off
        parts.append(bytes(b.header[:b.header_len]))
        parts.append(host[base:base + b.payload].tobytes())
        cbo.append(r)
    lb = [bl[-1].size if bl else 0 for bl in rank_blocks]
    footer = make_footer(P, fastq_size, sum(rk.n_subblocks for rk in res.ranks),
                         [rk.overlap_used for rk in res.ranks], cbo, lb)
    return b"".join(parts) + footer


def compress_bytes(data, P: int = 8, engine: Optional[Engine] = None, device: int = 0) -> bytes:
    """FASTQ bytes -> .ngsc bytes through the GPU (upload, compress, assemble)."""
    import torch
    eng = engine or Engine(device)
    arr = np.frombuffer(data, dtype=np.uint8) if isinstance(data, (bytes, bytearray)) else np.ascontiguousarray(data)
    d_in = torch.from_numpy(arr.copy()).to(f"cuda:{eng.device}")
    res = eng.compress_device(d_in, arr.size, P)
    torch.cuda.synchronize()
    return assemble_container(res, P, arr.size)


def decompress_bytes(ngsc: bytes, engine: Optional[Engine] = None, device: int = 0) -> bytes:
    """.ngsc container -> FASTQ text through the GPU decoder: the sub-blocks of every rank, ranks ascending (each rank's
    records are a contiguous run of the file, phyNGSC.cpp:113-166); the container is taken apart on the host."""
    import torch
    from . import ngsc_format
    eng = engine or Engine(device)
    streams = ngsc_format.rank_streams(bytes(ngsc))
    ranks = sorted(streams)
    subs = [sb for w in ranks for sb in streams[w]]
    owner = [w for w in ranks for _ in streams[w]]
    offs, pos = [], 0
    for sb in subs:
        offs.append(pos)
        pos += (len(sb) + 15) & ~15
    buf = np.zeros(pos + 16, dtype=np.uint8)
    for o, sb in zip(offs, subs):
        buf[o:o + len(sb)] = np.frombuffer(sb, dtype=np.uint8)
    d_sub = torch.from_numpy(buf).to(f"cuda:{eng.device}")
    text, toff, _ = eng.decompress_device(d_sub, offs, [len(sb) for sb in subs])
    out = text.cpu().numpy().tobytes()
    # Rank w's text belongs at s_w = w * (FS / P) + OV[w] (the first-record sync distance the footer keeps, phyNGSC.cpp:131-160)
    # up to s_(w+1), exactly as ngsc_decompress_file places it: a record that two ranks both stored (phyNGSC.cpp:295-300) comes
    # out once, and a file whose last line had no newline gets its exact size back.
    f = ngsc_format.read_footer(bytes(ngsc))
    P, fs = f.PS, f.FS
    ov = [0] + list(f.OV)
    s_at = [min(fs, w * (fs // P) + ov[w]) for w in range(P)] + [fs]
    res = bytearray(fs)
    filled = [0] * P
    toff = list(toff)
    for k, w in enumerate(owner):
        piece = out[toff[k]:toff[k + 1]]
        room = s_at[w + 1] - s_at[w] - filled[w]
        n = min(len(piece), max(room, 0))
        res[s_at[w] + filled[w]:s_at[w] + filled[w] + n] = piece[:n]
        filled[w] += n
    for w in range(P):
        if s_at[w + 1] < s_at[w] or filled[w] != s_at[w + 1] - s_at[w]:
            raise NgscError(-1, f"rank {w} decodes to {filled[w]} bytes, its stretch of the file has {s_at[w + 1] - s_at[w]}")
    return bytes(res)


# ---- multi-process (one process per GPU) ------------------------------------------------------------
def ranks_of_process(P: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous virtual ranks [lo, hi) owned by process ``rank`` of ``world`` (P % world == 0)."""
    assert P % world == 0, "virtual rank count must be a multiple of the process count"
    per = P // world
    return rank * per, (rank + 1) * per


def exchange_block_sizes(local_sizes: List[List[int]], local_meta: List[Tuple[int, int]], dist, device=None, group=None):
    """The only cross-GPU step, one-process-per-GPU form: all-gather of the per-rank rows
    [n_blocks, n_subblocks, overlap_used, block sizes ...] (the row layout of ngsc.h: ngsc_team_compress does the same
    exchange with RCCL inside one process).

    Replaces MPI_Gather / MPI_Gatherv of phyNGSC.cpp:964, 994 and the arrival-order shared file
    pointer: every process learns every rank's block sizes and derives the same file offsets.
    local_sizes: for each local virtual rank the sizes of its blocks; local_meta likewise.
    Returns (sizes per global rank, meta per global rank) in WRID order.
    """
    import torch
    world = dist.get_world_size(group)
    maxb = max((len(s) for s in local_sizes), default=0)
    t = torch.tensor([maxb], dtype=torch.int64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    row_len = int(t.item()) + 3
    rows = rows_from_sizes(local_sizes, local_meta, row_len)
    payload = torch.from_numpy(rows.view(np.int64)).to(device) if device is not None else torch.from_numpy(rows.view(np.int64).copy())
    gathered = [torch.zeros_like(payload) for _ in range(world)]
    dist.all_gather(gathered, payload, group=group)
    all_rows = np.concatenate([g.cpu().numpy().view(np.uint64) for g in gathered], axis=0)
    return sizes_from_rows(all_rows)


def file_offsets(sizes_per_rank: Sequence[Sequence[int]], ranks_per_grp: Optional[int] = None):
    """Exclusive prefix sum over the deterministic block order -> {(rank, k): file offset}, cbo, end."""
    off = 0
    table, cbo = {}, []
    for r, k in round_robin_order([len(s) for s in sizes_per_rank], ranks_per_grp):
        table[(r, k)] = off
        off += sizes_per_rank[r][k]
        cbo.append(r)
    return table, cbo, off
