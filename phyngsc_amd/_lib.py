"""ctypes loader for libngsc_hip.so (the C ABI declared in include/ngsc.h).

The HIP extension is the product: if it is missing this module raises, it never falls back to a CPU
implementation.  Import torch first when you use both, so that the library binds to the HIP runtime
already loaded by torch (same SONAME) and torch tensors' device pointers are valid for it.
"""
from __future__ import annotations

import ctypes
import os

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("NGSC_LIB_PATH") or os.path.join(_PKG, "libngsc_hip.so")   # (NGSC_LIB_PATH: another build, tools/ab_inproc.py)

EXPORTS = [
    "ngsc_version", "ngsc_strerror", "ngsc_ctx_create", "ngsc_ctx_destroy", "ngsc_last_hip_error",
    "ngsc_set_profiling", "ngsc_reset_kernel_times", "ngsc_get_kernel_times", "ngsc_compress_device",
    "ngsc_result_bytes", "ngsc_result_num_subblocks", "ngsc_result_num_ranks", "ngsc_result_subblocks",
    "ngsc_result_ranks", "ngsc_plan_blocks", "ngsc_make_header", "ngsc_make_footer", "ngsc_compress_file",
    "ngsc_compress_file_multi", "ngsc_ctx_device", "ngsc_ctx_trim", "ngsc_profile_only", "ngsc_compress_rank_part", "ngsc_decompress_device",
    "ngsc_decompress_file", "ngsc_read_footer", "ngsc_read_header",
    "ngsc_team_create", "ngsc_team_destroy", "ngsc_team_size", "ngsc_team_rccl_ranks", "ngsc_team_describe",
    "ngsc_team_compress", "ngsc_team_allgather_u64", "ngsc_team_member_allgather", "ngsc_team_abort", "ngsc_team_exchange_ms", "ngsc_rows_to_offsets",
]

BLOCK_SIZE = 8 << 20


class SubBlock(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_uint32), ("n_records", ctypes.c_uint32), ("win_pos", ctypes.c_uint64),
                ("wsize", ctypes.c_uint64), ("out_off", ctypes.c_uint64), ("info_len", ctypes.c_uint32),
                ("title_len", ctypes.c_uint32), ("qual_len", ctypes.c_uint32), ("dna_len", ctypes.c_uint32),
                ("max_qlen", ctypes.c_uint32), ("max_slen", ctypes.c_uint32), ("nsym", ctypes.c_uint32),
                ("nqual", ctypes.c_uint32), ("flags", ctypes.c_uint32), ("n_fields", ctypes.c_uint32),
                ("n_bases", ctypes.c_uint32), ("reserved", ctypes.c_uint32)]

    @property
    def size(self) -> int:
        return self.info_len + self.title_len + self.qual_len + self.dna_len


class Rank(ctypes.Structure):
    _fields_ = [("rank", ctypes.c_uint32), ("n_subblocks", ctypes.c_uint32), ("first_subblock", ctypes.c_uint32),
                ("overlap_used", ctypes.c_int32), ("stream_off", ctypes.c_uint64), ("stream_bytes", ctypes.c_uint64)]


class RankPart(ctypes.Structure):
    """ngsc_rank_part_t: where a rank's loop stands between the calls of ngsc_compress_rank_part()."""
    _fields_ = [("next_pos", ctypes.c_uint64), ("bytes_read", ctypes.c_uint64), ("bsize", ctypes.c_uint64),
                ("overlap", ctypes.c_uint32), ("done", ctypes.c_uint32), ("overlap_used", ctypes.c_int32),
                ("resume", ctypes.c_uint32)]


class FooterInfo(ctypes.Structure):
    """ngsc_footer_t"""
    _fields_ = [("P", ctypes.c_uint32), ("n_blocks", ctypes.c_uint32), ("n_subblocks", ctypes.c_uint32), ("lbes", ctypes.c_uint32),
                ("fastq_size", ctypes.c_uint64)]


class BlockHeaderInfo(ctypes.Structure):
    """ngsc_block_header_t"""
    _fields_ = [("wrid", ctypes.c_uint32), ("header_len", ctypes.c_uint32), ("n_subblocks", ctypes.c_uint32), ("bcss", ctypes.c_uint32),
                ("sb_len", ctypes.c_uint32 * 63)]


class Block(ctypes.Structure):
    _fields_ = [("wrid", ctypes.c_uint32), ("size", ctypes.c_uint32), ("header_len", ctypes.c_uint32),
                ("payload", ctypes.c_uint32), ("stream_off", ctypes.c_uint64), ("header", ctypes.c_uint8 * 288)]


class Shard(ctypes.Structure):
    """ngsc_shard_t: one team member's share of a step."""
    _fields_ = [("d_buf", ctypes.c_void_p), ("buf_len", ctypes.c_uint64), ("buf_off", ctypes.c_uint64),
                ("rank_lo", ctypes.c_uint32), ("rank_hi", ctypes.c_uint32), ("d_out", ctypes.c_void_p), ("out_cap", ctypes.c_uint64)]


class KernelTime(ctypes.Structure):
    _fields_ = [("name", ctypes.c_char_p), ("ms", ctypes.c_double), ("launches", ctypes.c_uint32)]


class NgscError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        msg = lib().ngsc_strerror(code).decode()
        super().__init__(f"ngsc error {code}: {msg}{(' - ' + detail) if detail else ''}")
        self.code = code


_lib = None


def lib():
    """Loads libngsc_hip.so; raises if the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
    L = ctypes.CDLL(LIB_PATH)
    vp, u32, u64 = ctypes.c_void_p, ctypes.c_uint32, ctypes.c_uint64
    L.ngsc_version.restype = ctypes.c_int
    L.ngsc_strerror.restype = ctypes.c_char_p
    L.ngsc_strerror.argtypes = [ctypes.c_int]
    L.ngsc_ctx_create.restype = ctypes.c_int
    L.ngsc_ctx_create.argtypes = [ctypes.c_int, ctypes.POINTER(vp)]
    L.ngsc_ctx_destroy.restype = None
    L.ngsc_ctx_destroy.argtypes = [vp]
    L.ngsc_last_hip_error.restype = ctypes.c_char_p
    L.ngsc_last_hip_error.argtypes = [vp]
    L.ngsc_set_profiling.argtypes = [vp, ctypes.c_int]
    L.ngsc_reset_kernel_times.argtypes = [vp]
    L.ngsc_profile_only.argtypes = [vp, ctypes.c_char_p]
    L.ngsc_get_kernel_times.restype = u32
    L.ngsc_get_kernel_times.argtypes = [vp, ctypes.POINTER(KernelTime), u32]
    L.ngsc_compress_device.restype = ctypes.c_int
    L.ngsc_compress_device.argtypes = [vp, vp, u64, u64, u64, u32, u32, u32, vp, u64, vp]
    L.ngsc_compress_rank_part.restype = ctypes.c_int
    L.ngsc_compress_rank_part.argtypes = [vp, vp, u64, u64, u64, u32, u32, ctypes.POINTER(RankPart), vp, u64, vp]
    L.ngsc_decompress_device.restype = ctypes.c_int
    L.ngsc_decompress_device.argtypes = [vp, vp, ctypes.POINTER(u64), ctypes.POINTER(u32), u32, vp, u64, ctypes.POINTER(u64),
                                         ctypes.POINTER(u32), vp]
    L.ngsc_read_footer.restype = ctypes.c_int
    L.ngsc_read_footer.argtypes = [vp, u64, ctypes.POINTER(FooterInfo), ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(u32), u32,
                                   ctypes.POINTER(u32), u32]
    L.ngsc_read_header.restype = ctypes.c_int
    L.ngsc_read_header.argtypes = [vp, u64, u32, ctypes.POINTER(BlockHeaderInfo)]
    L.ngsc_decompress_file.restype = ctypes.c_int
    L.ngsc_decompress_file.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, ctypes.POINTER(ctypes.c_double)]
    L.ngsc_result_bytes.restype = u64
    L.ngsc_result_bytes.argtypes = [vp]
    L.ngsc_result_num_subblocks.restype = u32
    L.ngsc_result_num_subblocks.argtypes = [vp]
    L.ngsc_result_num_ranks.restype = u32
    L.ngsc_result_num_ranks.argtypes = [vp]
    L.ngsc_result_subblocks.restype = ctypes.POINTER(SubBlock)
    L.ngsc_result_subblocks.argtypes = [vp]
    L.ngsc_result_ranks.restype = ctypes.POINTER(Rank)
    L.ngsc_result_ranks.argtypes = [vp]
    L.ngsc_plan_blocks.restype = ctypes.c_int
    L.ngsc_plan_blocks.argtypes = [u32, u32, ctypes.POINTER(u32), u32, ctypes.POINTER(Block), u32, ctypes.POINTER(u32)]
    L.ngsc_make_header.restype = u32
    L.ngsc_make_header.argtypes = [vp, u32, u32, u32, u32, u32, ctypes.POINTER(u32), u32]
    L.ngsc_make_footer.restype = u32
    L.ngsc_make_footer.argtypes = [vp, u32, u64, u32, u32, ctypes.POINTER(ctypes.c_int32), ctypes.POINTER(u32),
                                   ctypes.POINTER(u32)]
    L.ngsc_compress_file.restype = ctypes.c_int
    L.ngsc_compress_file.argtypes = [vp, ctypes.c_char_p, ctypes.c_char_p, u32, ctypes.POINTER(ctypes.c_double)]
    L.ngsc_compress_file_multi.restype = ctypes.c_int
    L.ngsc_compress_file_multi.argtypes = [ctypes.POINTER(vp), u32, ctypes.c_char_p, ctypes.c_char_p, u32,
                                           ctypes.POINTER(ctypes.c_double)]
    L.ngsc_ctx_device.restype = ctypes.c_int
    L.ngsc_ctx_device.argtypes = [vp]
    L.ngsc_team_create.restype = ctypes.c_int
    L.ngsc_team_create.argtypes = [ctypes.POINTER(vp), u32, ctypes.POINTER(vp)]
    L.ngsc_team_destroy.restype = None
    L.ngsc_team_destroy.argtypes = [vp]
    L.ngsc_team_size.restype = u32
    L.ngsc_team_size.argtypes = [vp]
    L.ngsc_team_rccl_ranks.restype = u32
    L.ngsc_team_rccl_ranks.argtypes = [vp]
    L.ngsc_team_describe.restype = ctypes.c_char_p
    L.ngsc_team_describe.argtypes = [vp]
    L.ngsc_team_compress.restype = ctypes.c_int
    L.ngsc_team_compress.argtypes = [vp, ctypes.POINTER(Shard), u64, u32, ctypes.POINTER(u64), u32, ctypes.POINTER(ctypes.c_double)]
    if hasattr(L, "ngsc_team_exchange_ms"):   # (an older build loaded through NGSC_LIB_PATH, tools/ab_inproc.py, may lack it)
        L.ngsc_team_exchange_ms.restype = None
        L.ngsc_team_exchange_ms.argtypes = [vp, ctypes.POINTER(ctypes.c_double)]
    L.ngsc_team_allgather_u64.restype = ctypes.c_int
    L.ngsc_team_allgather_u64.argtypes = [vp, ctypes.POINTER(u64), u32, ctypes.POINTER(u64)]
    L.ngsc_rows_to_offsets.restype = ctypes.c_int
    L.ngsc_rows_to_offsets.argtypes = [ctypes.POINTER(u64), u32, u32, u64, ctypes.POINTER(u64), ctypes.POINTER(u32), u32,
                                       ctypes.POINTER(u32), ctypes.POINTER(u64)]
    _lib = L
    return L
