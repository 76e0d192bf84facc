"""Container-level reader for ``.ngsc`` files (blocks, block headers, footer).

This is the parity judge's view of a file: the reference appends blocks in
arrival order (``MPI_File_write_shared``, phyNGSC.cpp:875/925), so two runs of the reference differ
in block ORDER while every ``(WRID, k)`` block -- the k-th block of working region WRID -- is byte
identical.  :func:`canonical` therefore maps a file to ``{(WRID, k): bytes}`` + footer scalars, and
:func:`equivalent` compares two files under that definition.

Format sources: block header tasks.cpp:1179-1200 / structures.h:310-333, footer
tasks.cpp:1104-1176 (reader counterpart tasks.cpp:1203-1292).
"""
from __future__ import annotations

import hashlib
import math
from dataclasses import dataclass, field
from typing import Dict, List, Tuple

BLOCK_SIZE = 8 << 20  # WRITE_BUFFER_SIZE, defs.h:21


class BitReader:
    """MSB-first bit reader over bytes (mirror of BitStream::GetBits, bit_stream.h:280-433)."""

    def __init__(self, data: bytes, pos: int = 0):
        self.data = data
        self.bitpos = pos * 8

    def bits(self, n: int) -> int:
        v = 0
        for _ in range(n):
            byte = self.data[self.bitpos >> 3]
            v = (v << 1) | ((byte >> (7 - (self.bitpos & 7))) & 1)
            self.bitpos += 1
        return v

    def align(self) -> None:
        self.bitpos = (self.bitpos + 7) & ~7

    @property
    def bytepos(self) -> int:
        return (self.bitpos + 7) >> 3


def _ceil_log2(x: int) -> int:
    return max(0, math.ceil(math.log2(x))) if x > 1 else 0


@dataclass
class Footer:
    BEPS: int = 0
    BEFS: int = 0
    BEBS: int = 0
    BESS: int = 0
    BELB: int = 0
    BEOV: int = 0
    LBES: int = 0
    PS: int = 0
    FS: int = 0
    BS: int = 0
    SS: int = 0
    OV: List[int] = field(default_factory=list)
    CBO: List[int] = field(default_factory=list)
    LBS: List[int] = field(default_factory=list)
    footer_len: int = 0  # bytes before the trailing u16

    def scalars(self) -> dict:
        """Everything except CBO (block order is run dependent in the reference)."""
        return dict(BEPS=self.BEPS, BEFS=self.BEFS, BEBS=self.BEBS, BESS=self.BESS, BELB=self.BELB,
                    BEOV=self.BEOV, LBES=self.LBES, PS=self.PS, FS=self.FS, BS=self.BS, SS=self.SS,
                    OV=list(self.OV), LBS=list(self.LBS))


@dataclass
class BlockHeader:
    WRID: int
    BHS: int
    NOSB: int
    BESO: int
    BCSS: int
    SBOL: List[int]


def read_footer(data: bytes) -> Footer:
    flen = (data[-2] << 8) | data[-1]
    start = len(data) - 2 - flen
    r = BitReader(data, start)
    f = Footer(footer_len=flen)
    f.BEPS, f.BEFS, f.BEBS, f.BESS = r.bits(4), r.bits(6), r.bits(4), r.bits(4)
    f.BELB, f.BEOV, f.LBES = r.bits(5), r.bits(4), r.bits(1)
    f.PS = r.bits(f.BEPS)
    if f.BEFS > 32:
        hi = r.bits(f.BEFS - 32)
        f.FS = (hi << 32) | r.bits(32)
    else:
        f.FS = r.bits(f.BEFS)
    f.BS = r.bits(f.BEBS)
    f.SS = r.bits(f.BESS)
    f.OV = [r.bits(f.BEOV) for _ in range(f.PS - 1)]
    cb = _ceil_log2(f.PS)
    f.CBO = [r.bits(cb) for _ in range(f.BS)]
    if f.LBES == 0:
        f.LBS = [r.bits(f.BELB) for _ in range(f.PS)]
    return f


def read_block_header(data: bytes, pos: int, n_ranks: int) -> BlockHeader:
    r = BitReader(data, pos)
    bewr = _ceil_log2(n_ranks)
    wrid = r.bits(bewr)
    bhs, nosb, beso, bcss = r.bits(12), r.bits(6), r.bits(5), r.bits(2)
    sbol = [r.bits(beso) for _ in range(nosb)]
    return BlockHeader(wrid, bhs, nosb, beso, bcss, sbol)


def split_blocks(data: bytes) -> Tuple[Footer, List[Tuple[int, bytes]]]:
    """Returns (footer, [(WRID, block bytes) in physical order]).

    Block sizes: 8 MiB except the last block of each rank (tasks.cpp:1280-1292).  With LBES=1 all
    last blocks share one size, which is what is left after the full blocks.
    """
    f = read_footer(data)
    body_len = len(data) - 2 - f.footer_len
    last_index: Dict[int, int] = {}
    for j, w in enumerate(f.CBO):
        last_index[w] = j
    sizes = [BLOCK_SIZE] * f.BS
    if f.LBES == 0:
        for w, j in last_index.items():
            sizes[j] = f.LBS[w]
    else:
        n_last = len(last_index)
        n_full = f.BS - n_last
        rest = body_len - n_full * BLOCK_SIZE
        assert n_last and rest % n_last == 0, "inconsistent LBES footer"
        for w, j in last_index.items():
            sizes[j] = rest // n_last
    assert sum(sizes) == body_len, f"block sizes {sum(sizes)} != body {body_len}"
    out = []
    pos = 0
    for j, w in enumerate(f.CBO):
        blk = data[pos:pos + sizes[j]]
        hdr = read_block_header(blk, 0, f.PS)
        assert hdr.WRID == w, f"block {j}: header WRID {hdr.WRID} != CBO {w}"
        out.append((w, blk))
        pos += sizes[j]
    return f, out


def canonical(data: bytes) -> Tuple[dict, Dict[Tuple[int, int], bytes]]:
    f, blocks = split_blocks(data)
    counter: Dict[int, int] = {}
    table: Dict[Tuple[int, int], bytes] = {}
    for w, blk in blocks:
        k = counter.get(w, 0)
        counter[w] = k + 1
        table[(w, k)] = blk
    return f.scalars(), table


def manifest(data: bytes) -> dict:
    """JSON-able fingerprint: footer scalars + sha256 of every (WRID, k) block."""
    scal, table = canonical(data)
    return {
        "footer": scal,
        "blocks": {f"{w}.{k}": {"len": len(b), "sha256": hashlib.sha256(b).hexdigest()}
                   for (w, k), b in sorted(table.items())},
    }


def rank_streams(data: bytes) -> Dict[int, List[bytes]]:
    """Per-rank list of sub-block byte strings, re-joined across split blocks (LSBS/FSBS)."""
    f, blocks = split_blocks(data)
    per_rank: Dict[int, List[bytes]] = {}
    carry: Dict[int, bytes] = {}
    for w, blk in blocks:
        hdr = read_block_header(blk, 0, f.PS)
        pos = hdr.BHS
        subs = per_rank.setdefault(w, [])
        for i, ln in enumerate(hdr.SBOL):
            piece = blk[pos:pos + ln]
            pos += ln
            first_cont = (i == 0) and (hdr.BCSS & 2)
            last_cont = (i == hdr.NOSB - 1) and (hdr.BCSS & 1)
            if first_cont:
                piece = carry.pop(w, b"") + piece
            if last_cont:
                carry[w] = piece
            else:
                subs.append(piece)
    assert not carry, "dangling split sub-block"
    return per_rank


def equivalent(a: bytes, b: bytes) -> Tuple[bool, str]:
    """SURVEY 5b definition of bit-exact: same (WRID,k) blocks, same footer scalars, valid CBO."""
    sa, ta = canonical(a)
    sb, tb = canonical(b)
    if sa != sb:
        diff = {k: (sa[k], sb[k]) for k in sa if sa[k] != sb.get(k)}
        return False, f"footer scalars differ: {diff}"
    if set(ta) != set(tb):
        return False, f"block sets differ: {sorted(set(ta) ^ set(tb))}"
    for key in sorted(ta):
        if ta[key] != tb[key]:
            x, y = ta[key], tb[key]
            n = min(len(x), len(y))
            first = next((i for i in range(n) if x[i] != y[i]), n)
            return False, f"block {key} differs at byte {first} (len {len(x)} vs {len(y)})"
    return True, "ok"
