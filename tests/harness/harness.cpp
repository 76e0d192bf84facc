// tests/harness/harness.cpp - CPU unit-test harness for the per-thread format logic in
// phyngsc_amd/csrc/ngsc_common.h (TEST INFRASTRUCTURE, compiled with g++, never part of the product).
//
// The HIP kernels gather statistics with wave intrinsics and LDS; what they *decide and emit* from
// those statistics lives in NGSC_HD helpers (finalize_field, title_header, title_group,
// huffman_sequential, plan_rank, block/header layout).  This harness feeds those helpers with
// statistics computed by plain loops so that the format logic can be checked against the oracle on a
// machine without a GPU.  It is not a CPU implementation of the engine and nothing in phyngsc_amd
// loads it.
#include "../../phyngsc_amd/csrc/ngsc_common.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace ngsc;

namespace {

static void stat_max(uint32_t &p, uint32_t v) { if (v > p) p = v; }

struct Tree {
    std::vector<uint32_t> packed;  // code<<5|len per symbol
    std::vector<uint8_t> blob;
    uint32_t max_len = 0;
};

static Tree build_tree(const uint32_t *freq, uint32_t n)
{
    std::vector<uint64_t> keys(n);
    for (uint32_t i = 0; i < n; ++i) keys[i] = ((uint64_t)freq[i] << 32) | i;
    std::sort(keys.begin(), keys.end());
    std::vector<uint32_t> ifreq(n + 1), code(2 * n + 2);
    std::vector<uint16_t> lc(2 * n + 2), rc(2 * n + 2), stack(2 * n + 4);
    std::vector<uint8_t> len(2 * n + 2), blob(BLOB_SLOT_512 + 64);
    Tree t;
    uint32_t bl = huffman_sequential(keys.data(), n, ifreq.data(), lc.data(), rc.data(), code.data(), len.data(),
                                     stack.data(), blob.data(), &t.max_len);
    t.blob.assign(blob.begin(), blob.begin() + bl);
    t.packed.resize(n);
    for (uint32_t i = 0; i < n; ++i) t.packed[i] = (code[i] << 5) | len[i];
    return t;
}

}  // namespace

extern "C" {

// Huffman: codes (code<<5|len) + blob for a frequency vector. Returns blob length; *max_len longest code.
uint32_t h_huffman(const uint32_t *freq, uint32_t n, uint32_t *packed, uint8_t *blob, uint32_t *max_len)
{
    Tree t = build_tree(freq, n);
    memcpy(packed, t.packed.data(), 4 * n);
    memcpy(blob, t.blob.data(), t.blob.size());
    *max_len = t.max_len;
    return (uint32_t)t.blob.size();
}

// Title stream of one sub-block whose records are (te[i], se[i]) inside buf, first record starting at ls0.
// Returns the stream length written to out (capacity cap), or a negative error code.
int64_t h_title_stream(const uint8_t *buf, uint32_t buf_len, const uint32_t *te, const uint32_t *se, uint32_t n,
                       uint32_t ls0, uint8_t *out, uint64_t cap)
{
    std::vector<uint32_t> ls(n + 1);
    ls[0] = ls0;
    for (uint32_t i = 0; i < n; ++i) ls[i + 1] = 2 * se[i] - te[i] + 3;
    const uint32_t nf = count_fields(buf, ls[0], te[0]);
    if (nf > WIDE_FIELDS) return ERR_UNSUPPORTED;
    // (as the engine: the field arrays of a job with more than MAX_FIELDS separators are WIDE_FIELDS wide, and its title groups
    // are walked with multi-word field masks)
    const bool wide = nf > MAX_FIELDS;
    const uint32_t fcap = wide ? WIDE_FIELDS : MAX_FIELDS;

    // ---- statistics (what k_title_stats accumulates) ----
    std::vector<FieldStat> st(fcap);
    memset(st.data(), 0, sizeof(FieldStat) * fcap);
    std::vector<uint32_t> prev(nf, 0);
    std::vector<uint8_t> prevnum(nf, 0);
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t pc = ls[r], p0 = ls[0], k = 0;
        while (pc <= te[r]) {
            uint32_t ec = field_end(buf, pc);
            if (k < nf) {
                uint32_t e0 = field_end(buf, p0);
                uint32_t len = ec - pc, len0 = e0 - p0, v = 0;
                bool ovf = false;
                bool isnum = parse_num(buf, pc, ec, v, ovf);
                FieldStat &s = st[k];
                stat_max(s.max_len, len);
                stat_max(s.inv_min_len, ~len);
                uint32_t m = std::min(len, len0);
                bool diff = false;
                for (uint32_t p = 0; p < m; ++p)
                    if (buf[pc + p] != buf[p0 + p]) { diff = true; if (p < 512) s.not_ham[p >> 5] |= 1u << (p & 31); }
                if (len != len0) { s.not_len_const = 1; s.not_const = 1; }
                else if (diff) s.not_const = 1;
                if (!isnum) s.not_num = 1;
                else {
                    if (ovf) s.ovf = 1;
                    stat_max(s.max_v, bias((int32_t)v));
                    stat_max(s.inv_min_v, ~bias((int32_t)v));
                    if (r >= 1 && prevnum[k]) {
                        int32_t d = (int32_t)(v - prev[k]);
                        stat_max(s.max_d, bias(d));
                        stat_max(s.inv_min_d, ~bias(d));
                    }
                }
                prev[k] = v;
                prevnum[k] = isnum;
                p0 = e0 + 1;
            }
            ++k;
            pc = ec + 1;
        }
        if (k != nf) return ERR_FIELDS;
    }

    // ---- arena for one sub-block ----
    Arena A;
    memset(&A, 0, sizeof A);
    SbState sb;
    memset(&sb, 0, sizeof sb);
    sb.rec_lo = 0; sb.n_rec = n; sb.nf = nf;
    std::vector<FieldDesc> fd(fcap);
    std::vector<uint32_t> nhist((size_t)fcap * MAX_NUM_HUF, 0), ncodes((size_t)fcap * MAX_NUM_HUF, 0);
    std::vector<uint8_t> nblob((size_t)fcap * BLOB_SLOT_512, 0);
    std::vector<uint32_t> nblob_len(fcap, 0);
    uint32_t n_slots = 0;
    for (uint32_t f = 0; f < nf; ++f) {
        uint32_t p = ls[0], e = 0;
        for (uint32_t k = 0; k <= f; ++k) { e = field_end(buf, p); if (k < f) p = e + 1; }
        memset(&fd[f], 0, sizeof(FieldDesc));
        int rc = finalize_field(st[f], p, e - p, buf[e], n, f, nf, fd[f]);
        if (rc) return rc;
        if (fd[f].kind == KIND_TEXT) fd[f].slot = n_slots++;
    }
    std::vector<uint32_t> chist((size_t)std::max(n_slots, 1u) * TEXT_TREES * 256, 0), tcodes(chist.size(), 0);
    std::vector<uint8_t> tblob((size_t)std::max(n_slots, 1u) * TEXT_TREES * BLOB_SLOT_256, 0);
    std::vector<uint32_t> tblob_len((size_t)std::max(n_slots, 1u) * TEXT_TREES, 0);
    // ---- histograms (what k_title_hist accumulates) ----
    std::fill(prev.begin(), prev.end(), 0);
    for (uint32_t r = 0; r < n; ++r) {
        uint32_t pc = ls[r];
        for (uint32_t f = 0; f < nf; ++f) {
            uint32_t ec = field_end(buf, pc);
            const FieldDesc &d = fd[f];
            if (d.kind == KIND_NUM && d.has_huff) {
                uint32_t v = 0;
                bool o;
                parse_num(buf, pc, ec, v, o);
                uint32_t *nh = nhist.data() + (size_t)f * MAX_NUM_HUF;
                if (!d.delta_coding) { uint32_t idx = v - (uint32_t)d.min_v; if (idx < d.n_sym) nh[idx] += r == 0 ? 2 : 1; }
                else if (r >= 1) { uint32_t idx = v - prev[f] - (uint32_t)d.min_d; if (idx < d.n_sym) nh[idx] += 1; }
                prev[f] = v;
            } else if (d.kind == KIND_TEXT) {
                uint32_t *ch = chist.data() + (size_t)d.slot * TEXT_TREES * 256;
                for (uint32_t x = 0; x < ec - pc; ++x) ch[(x < MAX_STAT_LEN ? x : MAX_STAT_LEN) * 256 + buf[pc + x]]++;
            }
            pc = ec + 1;
        }
    }
    // ---- trees ----
    for (uint32_t f = 0; f < nf; ++f) {
        const FieldDesc &d = fd[f];
        if (d.kind == KIND_NUM && d.has_huff) {
            Tree t = build_tree(nhist.data() + (size_t)f * MAX_NUM_HUF, d.n_sym);
            if (t.max_len > MAX_CODE_LEN) return ERR_UNSUPPORTED;
            memcpy(ncodes.data() + (size_t)f * MAX_NUM_HUF, t.packed.data(), 4 * d.n_sym);
            memcpy(nblob.data() + (size_t)f * BLOB_SLOT_512, t.blob.data(), t.blob.size());
            nblob_len[f] = (uint32_t)t.blob.size();
        } else if (d.kind == KIND_TEXT) {
            for (uint32_t j = 0; j <= MAX_STAT_LEN; ++j) {
                if (!text_tree_needed(d, j)) continue;
                size_t tix = (size_t)d.slot * TEXT_TREES + j;
                Tree t = build_tree(chist.data() + tix * 256, 256);
                if (t.max_len > MAX_CODE_LEN) return ERR_UNSUPPORTED;
                memcpy(tcodes.data() + tix * 256, t.packed.data(), 4 * 256);
                memcpy(tblob.data() + tix * BLOB_SLOT_256, t.blob.data(), t.blob.size());
                tblob_len[tix] = (uint32_t)t.blob.size();
            }
        }
    }
    A.buf = buf; A.buf_len = buf_len; A.te = te; A.se = se; A.ls = ls.data();
    A.sb = &sb; A.fdesc = fd.data();
    A.ncodes = ncodes.data(); A.nblob = nblob.data(); A.nblob_len = nblob_len.data();
    A.tcodes = tcodes.data(); A.tblob = tblob.data(); A.tblob_len = tblob_len.data();
    A.n_sb = 1;
    A.fcap = fcap;
    // ---- measure, then write (the two passes the kernels make) ----
    BitCounter hc;
    title_header(A, 0, hc);
    uint64_t total = hc.nbytes();
    std::vector<uint64_t> goff;
    for (uint32_t g0 = 0; g0 < n; g0 += GROUP) {
        BitCounter bc;
        if (wide) title_group<WideMask>(A, 0, g0, std::min(g0 + GROUP, n), bc);
        else title_group(A, 0, g0, std::min(g0 + GROUP, n), bc);
        goff.push_back(total);
        total += bc.nbytes();
    }
    if (total > cap) return ERR_CAPACITY;
    ByteWriter hw(out);
    title_header(A, 0, hw);
    if (hw.nbytes() != hc.nbytes()) return -100;
    uint32_t gi = 0;
    for (uint32_t g0 = 0; g0 < n; g0 += GROUP, ++gi) {
        ByteWriter w(out + goff[gi]);
        if (wide) title_group<WideMask>(A, 0, g0, std::min(g0 + GROUP, n), w);
        else title_group(A, 0, g0, std::min(g0 + GROUP, n), w);
        uint64_t expect = (gi + 1 < goff.size() ? goff[gi + 1] : total) - goff[gi];
        if (w.nbytes() != expect) return -101;
    }
    return (int64_t)total;
}

// Cut plan of every rank of a whole file held in buf (record table built by a plain scan).
// out: for each cut 4 values (rank, rec_lo, n_rec, overlap_used of its rank). Returns number of cuts or <0.
int64_t h_plan_file(const uint8_t *buf, uint64_t file_size, uint32_t P, uint32_t *out, uint32_t cap)
{
    std::vector<uint32_t> te, se, ls;
    ls.push_back(0);
    uint32_t line = 0;
    // a newline is assumed just past the file's last byte (kernels_index.hip: k_nl_count, at_file_end)
    const uint64_t scan_end = file_size + (file_size ? 1 : 0);
    for (uint64_t i = 0; i < scan_end; ++i) {
        if (i < file_size && buf[i] != '\n') continue;
        uint32_t k = line & 3;
        if (k == 0) te.push_back((uint32_t)i);
        else if (k == 1) se.push_back((uint32_t)i);
        else if (k == 3) ls.push_back((uint32_t)i + 1);
        ++line;
    }
    uint32_t n = line / 4;
    te.resize(n + 1); se.resize(n + 1); ls.resize(n + 2);
    uint32_t nc = 0;
    for (uint32_t r = 0; r < P; ++r) {
        RankGeom g = rank_geometry(file_size, P, r);
        uint64_t first = 0;
        if (r) {
            // phyNGSC.cpp:131-156
            uint64_t i = (uint64_t)g.wr_start;
            const uint64_t lim = std::min<uint64_t>(file_size, (uint64_t)g.wr_start + (uint64_t)g.bsize);   // the rank's first window
            while (i < lim && buf[i] != '@') ++i;
            uint64_t at = i;
            while (i < lim && buf[i] != '\n') ++i;
            if (i >= lim) return ERR_FORMAT;
            first = (i + 1 < lim && buf[i + 1] == '@') ? i + 1 : at;
        }
        std::vector<Cut> cuts(4096);
        RankPlan pl;
        int rc = plan_rank(file_size, P, r, 0, te.data(), ls.data(), n, first, cuts.data(), 4096, &pl);
        if (rc) return rc;
        for (uint32_t k = 0; k < pl.n_cuts; ++k) {
            if (nc >= cap) return ERR_CAPACITY;
            out[4 * nc + 0] = r; out[4 * nc + 1] = cuts[k].rec_lo; out[4 * nc + 2] = cuts[k].n_rec;
            out[4 * nc + 3] = (uint32_t)pl.overlap_used;
            ++nc;
        }
    }
    return nc;
}

}  // extern "C"
