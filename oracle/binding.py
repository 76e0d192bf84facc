"""ctypes binding of oracle/libngsc_oracle.so (the CPU restatement of the reference).

TEST INFRASTRUCTURE.  Only tests/, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline``
leg may import this module; the product path (phyngsc_amd) never does.
"""
from __future__ import annotations

import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libngsc_oracle.so")
_SRC = [os.path.join(_HERE, "ngsc_oracle.c"), os.path.join(_HERE, "ngsc_decode.c"), os.path.join(_HERE, "ngsc_oracle.h")]

MAX_RECORDS = 100000
WINDOW = 1 << 23
OVERLAP = 500


class SbInfo(ctypes.Structure):
    _fields_ = [("n_records", ctypes.c_uint32), ("max_qlen", ctypes.c_uint32), ("max_slen", ctypes.c_uint32),
                ("nsym", ctypes.c_uint32), ("nqual", ctypes.c_uint32), ("flags", ctypes.c_uint32),
                ("n_fields", ctypes.c_uint32), ("info_len", ctypes.c_uint32), ("title_len", ctypes.c_uint32),
                ("qual_len", ctypes.c_uint32), ("dna_len", ctypes.c_uint32), ("next_offset", ctypes.c_uint64)]


class FileStats(ctypes.Structure):
    _fields_ = [("n_blocks", ctypes.c_uint32), ("n_subblocks", ctypes.c_uint32), ("out_bytes", ctypes.c_uint64),
                ("seconds", ctypes.c_double)]


class OracleError(RuntimeError):
    def __init__(self, rc: int):
        super().__init__(f"oracle error {rc}")
        self.rc = rc


_lib = None
_libc = ctypes.CDLL(None)
_libc.free.argtypes = [ctypes.c_void_p]


def build(force: bool = False) -> str:
    stale = not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(s) for s in _SRC)
    if force or stale:
        subprocess.check_call(["gcc", "-O2", "-fopenmp", "-shared", "-fPIC", "-o", _SO, _SRC[0], _SRC[1], "-lm"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        u8p = ctypes.c_void_p
        L.ngsc_o_compress_file.restype = ctypes.c_int
        L.ngsc_o_compress_file.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_int,
                                           ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                           ctypes.POINTER(FileStats)]
        L.ngsc_o_subblock.restype = ctypes.c_int
        L.ngsc_o_subblock.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.POINTER(ctypes.c_uint64),
                                      ctypes.POINTER(SbInfo)]
        L.ngsc_o_record_index.restype = ctypes.c_int
        L.ngsc_o_record_index.argtypes = [u8p, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint64, ctypes.c_uint32,
                                          u8p, u8p, ctypes.POINTER(ctypes.c_uint32), ctypes.POINTER(ctypes.c_uint64)]
        L.ngsc_o_first_record.restype = ctypes.c_int64
        L.ngsc_o_first_record.argtypes = [u8p, ctypes.c_uint64]
        L.ngsc_o_huffman.restype = ctypes.c_int
        L.ngsc_o_huffman.argtypes = [u8p, ctypes.c_uint32, u8p, u8p, ctypes.POINTER(ctypes.c_uint32)]
        L.ngsc_o_make_header.restype = ctypes.c_uint32
        L.ngsc_o_make_header.argtypes = [u8p, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32, ctypes.c_uint32,
                                         ctypes.c_uint32, u8p, ctypes.c_uint32]
        L.ngsc_o_make_footer.restype = ctypes.c_uint32
        L.ngsc_o_make_footer.argtypes = [u8p, ctypes.c_uint32, ctypes.c_uint64, ctypes.c_uint32, ctypes.c_uint32,
                                         u8p, u8p, u8p]
        L.ngsc_o_decode_subblock.restype = ctypes.c_int
        L.ngsc_o_decode_subblock.argtypes = [u8p, ctypes.c_uint64, ctypes.POINTER(ctypes.c_void_p),
                                             ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64),
                                             ctypes.POINTER(ctypes.c_uint32)]
        L.ngsc_o_lossy_count.restype = ctypes.c_uint64
        L.ngsc_o_lossy_count.argtypes = [ctypes.c_int]
        _lib = L
    return _lib


def _as_u8(data) -> np.ndarray:
    if isinstance(data, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(data), dtype=np.uint8)
    return np.ascontiguousarray(data, dtype=np.uint8)


def compress_file(data, P: int, threads: int = 1, as_reference: bool = False):
    """Returns (ngsc bytes, FileStats).

    A varying numeric title field whose digits exceed 32 bits is stored modulo 2^32 by the reference (utils.h:117-125): its
    container does not hold its input.  The HIP engine refuses such input (NGSC_ERR_RANGE), and so does this function
    (OracleError(-5)) unless ``as_reference`` asks for the reference's own container."""
    a = _as_u8(data)
    out = ctypes.c_void_p()
    n = ctypes.c_uint64()
    st = FileStats()
    lib().ngsc_o_lossy_count(1)
    rc = lib().ngsc_o_compress_file(a.ctypes.data, a.size, P, threads, ctypes.byref(out), ctypes.byref(n),
                                    ctypes.byref(st))
    if rc:
        raise OracleError(rc)
    if lib().ngsc_o_lossy_count(1) and not as_reference:
        _libc.free(out)
        raise OracleError(-5)
    if n.value < (1 << 31):
        res = ctypes.string_at(out.value, n.value)
    else:   # (string_at takes a C int) multi-GB outputs come back as a numpy array
        res = np.ctypeslib.as_array((ctypes.c_ubyte * n.value).from_address(out.value)).copy()
    _libc.free(out)
    return res, st


def subblock(window, wsize: int | None = None, rec_start: int = 0, overlap: int = OVERLAP):
    """One sub-block: returns (bytes info|title|quality|DNA, SbInfo)."""
    a = _as_u8(window)
    if wsize is None:
        wsize = a.size
    out = ctypes.c_void_p()
    n = ctypes.c_uint64()
    inf = SbInfo()
    rc = lib().ngsc_o_subblock(a.ctypes.data, a.size, wsize, rec_start, overlap, ctypes.byref(out), ctypes.byref(n),
                               ctypes.byref(inf))
    if rc:
        raise OracleError(rc)
    res = ctypes.string_at(out.value, n.value)
    _libc.free(out)
    return res, inf


def record_index(window, wsize: int | None = None, rec_start: int = 0, overlap: int = OVERLAP):
    a = _as_u8(window)
    if wsize is None:
        wsize = a.size
    te = np.zeros(MAX_RECORDS + 2, dtype=np.uint32)
    se = np.zeros(MAX_RECORDS + 2, dtype=np.uint32)
    n = ctypes.c_uint32()
    nxt = ctypes.c_uint64()
    rc = lib().ngsc_o_record_index(a.ctypes.data, a.size, wsize, rec_start, overlap, te.ctypes.data, se.ctypes.data,
                                   ctypes.byref(n), ctypes.byref(nxt))
    if rc:
        raise OracleError(rc)
    return te[:n.value].copy(), se[:n.value].copy(), nxt.value


@dataclass
class Huffman:
    root: int
    code: np.ndarray  # uint32[2n-1]
    length: np.ndarray  # uint32[2n-1]
    blob: bytes


def huffman(freq) -> Huffman:
    f = np.ascontiguousarray(freq, dtype=np.uint32)
    n = f.size
    codes = np.zeros((2 * n, 2), dtype=np.uint32)
    blob = np.zeros(64 + 4 * n, dtype=np.uint8)
    bl = ctypes.c_uint32()
    root = lib().ngsc_o_huffman(f.ctypes.data, n, codes.ctypes.data, blob.ctypes.data, ctypes.byref(bl))
    return Huffman(root, codes[:2 * n - 1, 0].copy(), codes[:2 * n - 1, 1].copy(), bytes(blob[:bl.value]))


def make_header(wrid, bewr, bhs, beso, bcss, sbol) -> bytes:
    s = np.ascontiguousarray(sbol, dtype=np.uint32)
    out = np.zeros(64 + 4 * s.size, dtype=np.uint8)
    n = lib().ngsc_o_make_header(out.ctypes.data, wrid, bewr, bhs, beso, bcss, s.ctypes.data, s.size)
    return bytes(out[:n])


def make_footer(P, fastq_size, n_subblocks, overlaps, cbo, lb_sizes) -> bytes:
    ov = np.ascontiguousarray(overlaps, dtype=np.int32)
    cb = np.ascontiguousarray(cbo, dtype=np.uint32)
    lb = np.ascontiguousarray(lb_sizes, dtype=np.uint32)
    out = np.zeros(256 + 8 * (cb.size + P), dtype=np.uint8)
    n = lib().ngsc_o_make_footer(out.ctypes.data, P, fastq_size, cb.size, n_subblocks, ov.ctypes.data,
                                 cb.ctypes.data, lb.ctypes.data)
    return bytes(out[:n])


def decode_subblock(sb):
    """Inverse of subblock(): returns (FASTQ bytes, bytes consumed, n_records)."""
    a = _as_u8(sb)
    out = ctypes.c_void_p()
    n = ctypes.c_uint64()
    used = ctypes.c_uint64()
    nrec = ctypes.c_uint32()
    rc = lib().ngsc_o_decode_subblock(a.ctypes.data, a.size, ctypes.byref(out), ctypes.byref(n), ctypes.byref(used),
                                      ctypes.byref(nrec))
    if rc:
        raise OracleError(rc)
    res = ctypes.string_at(out.value, n.value)
    _libc.free(out)
    return res, used.value, nrec.value


def decompress(ngsc) -> bytes:
    """.ngsc container -> FASTQ text: the sub-blocks of every rank, ranks ascending (each rank's records are a
    contiguous run of the file, phyNGSC.cpp:113-166).  Every sub-block must decode to exactly its SBOL length."""
    from phyngsc_amd import ngsc_format
    streams = ngsc_format.rank_streams(bytes(ngsc))
    parts = []
    for w in sorted(streams):
        for sb in streams[w]:
            text, used, _ = decode_subblock(sb)
            if used != len(sb):
                raise OracleError(-1)
            parts.append(text)
    text = b"".join(parts)
    # the footer records the FASTQ size: a file whose last line had no newline decodes with one too many
    fs = ngsc_format.read_footer(bytes(ngsc)).FS
    return text[:fs] if len(text) == fs + 1 and text.endswith(b"\n") else text
