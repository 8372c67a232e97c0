// Host-side decision loops of the post-processing, as native code behind the C ABI.
//
// The reference's duplicate filters are sequential greedy loops over a few dozen masks per call (inference.py:1451-1459,
// 2640-2671): every decision depends on the ones before it, so they stay on the host -- but as ONE native call per batch of
// tiles over the integer tables the device reduced (pixel counts, boxes, the pair-intersection matrix of
// demia_mask_pair_matrix), not as interpreted loops per tile.  Same integer counts, same float64 division and comparison as
// the reference's scalar code; nothing here touches the GPU.
#include <algorithm>
#include <cstdint>
#include <numeric>
#include <vector>

#include "common.h"

namespace {

inline double iou_of(int64_t inter, int64_t a, int64_t b) {
    const int64_t uni = a + b - inter;
    return uni > 0 ? (double)inter / (double)uni : 0.0;     // `inter / union if union > 0 else 0` (inference.py:422-435)
}

}  // namespace

// inference.py:1451-1459 for S segments: mask p of a segment is kept unless IoU(p, q) > thr for a q kept before it.
// inter [n, ld]: row i, column j - row_first[i] = |mask_i & mask_j| for j > i (demia_mask_pair_matrix); seg_first / seg_len:
// the masks first .. first + len - 1 of segment s take part (len may be shorter than the segment the matrix was built for).
extern "C" int demia_host_greedy_keep(const int32_t* inter, int ld, const int32_t* row_first, const int64_t* area,
                                      const int32_t* seg_first, const int32_t* seg_len, int S, double thr, uint8_t* keep) {
    DEMIA_REQUIRE(inter && row_first && area && seg_first && seg_len && keep && ld > 0 && S >= 0, "args");
    std::vector<int> kept;
    for (int s = 0; s < S; ++s) {
        const int f = seg_first[s], n = seg_len[s];
        kept.clear();
        for (int p = f; p < f + n; ++p) {
            bool dup = false;
            for (int q : kept) {                                  // q < p
                const int col = p - row_first[q];
                const int64_t it = col < ld ? inter[(long)q * ld + col] : 0;
                if (iou_of(it, area[p], area[q]) > thr) { dup = true; break; }
            }
            keep[p] = dup ? 0 : 1;
            if (!dup) kept.push_back(p);
        }
    }
    return DEMIA_OK;
}

// Step 2 of deduplicate_masks_smart (inference.py:2640-2671) for T tiles, bug for bug: descending stable score
// order (`np.argsort(scores)[::-1]`: of equal scores the HIGHER index first), `others = sorted_indices[idx + 1:]` sliced by
// the MASK INDEX, and the box pre-filter that stores (y_min, y_max, x_min, x_max) but reads (y_min, x_min, y_max, x_max).
// items: global mask indices of all tiles, tile t = items[tile_off[t] .. tile_off[t + 1]); scores / classes per item;
// bbox [n_all][4] = y0, x0, y1, x1 and area [n_all] per GLOBAL mask; inter / ld / row_first as above (both masks of a pair in
// the same matrix segment).  keep_out: per tile the LOCAL positions kept, in the order they are kept; keep_cnt [T].
extern "C" int demia_host_dedup_smart(const int32_t* inter, int ld, const int32_t* row_first, const int64_t* area,
                                      const int64_t* bbox, const int32_t* items, const double* scores, const int32_t* classes,
                                      const int32_t* tile_off, int T, double thr, int32_t* keep_out, int32_t* keep_cnt) {
    DEMIA_REQUIRE(inter && row_first && area && bbox && items && scores && classes && tile_off && keep_out && keep_cnt && ld > 0, "args");
    std::vector<int> order, pos_after;
    std::vector<uint8_t> hit, removed;
    for (int t = 0; t < T; ++t) {
        const int o = tile_off[t], n = tile_off[t + 1] - o;
        keep_cnt[t] = 0;
        if (n <= 0) continue;
        order.resize(n);
        std::iota(order.begin(), order.end(), 0);
        std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return scores[o + a] < scores[o + b]; });
        std::reverse(order.begin(), order.end());
        hit.assign((size_t)n * n, 0);
        for (int i = 0; i < n; ++i) {
            const int gi = items[o + i];
            const int64_t* bi = bbox + (long)gi * 4;            // stored as (y_min, y_max, x_min, x_max) = bi[0], bi[2], bi[1], bi[3]
            const int64_t si[4] = {bi[0], bi[2], bi[1], bi[3]};
            for (int j = 0; j < n; ++j) {
                if (classes[o + i] != classes[o + j]) continue;
                const int gj = items[o + j];
                const int64_t* bj = bbox + (long)gj * 4;
                const int64_t sj[4] = {bj[0], bj[2], bj[1], bj[3]};
                // ... read as (y_min, x_min, y_max, x_max): the literal test of inference.py:2685-2694
                if (si[3] < sj[1] || sj[3] < si[1] || si[2] < sj[0] || sj[2] < si[0]) continue;
                int64_t it;
                if (gi == gj) it = area[gi];
                else {
                    const int lo = gi < gj ? gi : gj, hi = gi < gj ? gj : gi;
                    const int col = hi - row_first[lo];
                    it = (col >= 0 && col < ld) ? inter[(long)lo * ld + col] : 0;
                }
                if (it > 0 && iou_of(it, area[gi], area[gj]) > thr) hit[(size_t)i * n + j] = 1;
            }
        }
        // rank[m] = position of mask m in `order`; "after[p]" = the masks at positions > p
        pos_after.assign(n, 0);
        for (int p = 0; p < n; ++p) pos_after[order[p]] = p;
        removed.assign(n, 0);
        int32_t* out = keep_out + o;
        int cnt = 0;
        for (int p = 0; p < n; ++p) {
            const int idx = order[p];
            if (removed[idx]) continue;
            out[cnt++] = idx;
            // `others = sorted_indices[idx + 1:]`: idx (a mask index) used as a POSITION
            for (int j = 0; j < n; ++j)
                if (hit[(size_t)idx * n + j] && pos_after[j] > idx) removed[j] = 1;
        }
        keep_cnt[t] = cnt;
    }
    return DEMIA_OK;
}

// ---------------------------------------------------------------------------------------------------------------------
// The float columns of measurements_results.csv as text.  The reference writes its rows with Python's csv.writer, which
// formats a float with repr(): the shortest digit string that round-trips (David Gay's mode 0 = what std::to_chars yields),
// laid out by CPython's rule -- scientific notation iff the decimal point position `decpt` (value = 0.d1d2... x 10^decpt) is
// <= -4 or > 16, exponent with sign and at least two digits, otherwise positional with a ".0" appended to integers.  2700
// rows x 12 floats per 48-tile step are 21 ms of interpreter time that way; here one native call writes the comma-joined
// floats of every row (rows separated by '\n'), the host code adds the string columns around them.
#include <charconv>
#include <cmath>
#include <cstring>

namespace {

// repr(x) into `out`; returns the number of characters written (at most 24 + sign)
inline int py_float_repr(double x, char* out) {
    if (std::isnan(x)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(x)) { if (x < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char* o = out;
    if (std::signbit(x)) { *o++ = '-'; x = -x; }
    if (x == 0.0) { memcpy(o, "0.0", 3); return (int)(o - out) + 3; }
    char sci[40];
    const auto r = std::to_chars(sci, sci + sizeof(sci), x, std::chars_format::scientific);     // d[.ddd]e[+-]XX, shortest digits
    const char* e = sci;
    while (e < r.ptr && *e != 'e') ++e;
    char digits[24];
    int nd = 0;
    for (const char* c = sci; c < e; ++c)
        if (*c != '.') digits[nd++] = *c;
    int ex = 0;
    {
        const char* c = e + 1;
        const bool neg = *c == '-';
        if (*c == '+' || *c == '-') ++c;
        for (; c < r.ptr; ++c) ex = ex * 10 + (*c - '0');
        if (neg) ex = -ex;
    }
    const int decpt = ex + 1;
    if (decpt <= -4 || decpt > 16) {
        *o++ = digits[0];
        if (nd > 1) { *o++ = '.'; memcpy(o, digits + 1, nd - 1); o += nd - 1; }
        *o++ = 'e';
        int xe = decpt - 1;
        *o++ = xe < 0 ? '-' : '+';
        if (xe < 0) xe = -xe;
        char tmp[8];
        int nt = 0;
        do { tmp[nt++] = (char)('0' + xe % 10); xe /= 10; } while (xe);
        if (nt < 2) tmp[nt++] = '0';
        while (nt) *o++ = tmp[--nt];
    } else if (decpt <= 0) {
        *o++ = '0'; *o++ = '.';
        for (int i = 0; i < -decpt; ++i) *o++ = '0';
        memcpy(o, digits, nd); o += nd;
    } else if (decpt >= nd) {
        memcpy(o, digits, nd); o += nd;
        for (int i = 0; i < decpt - nd; ++i) *o++ = '0';
        *o++ = '.'; *o++ = '0';
    } else {
        memcpy(o, digits, decpt); o += decpt;
        *o++ = '.';
        memcpy(o, digits + decpt, nd - decpt); o += nd - decpt;
    }
    return (int)(o - out);
}

}  // namespace

// vals [rows][cols] f64 -> out: per row the repr() of its floats joined by ',', rows separated by '\n' (no trailing one).
// Returns the number of bytes written, or -1 if `cap` is too small (needs at most rows * cols * 26 bytes).
extern "C" int64_t demia_host_repr_rows(const double* vals, int64_t rows, int cols, char* out, int64_t cap) {
    if (!vals || !out || rows < 0 || cols <= 0) return -1;
    int64_t pos = 0;
    for (int64_t r = 0; r < rows; ++r) {
        if (pos + (int64_t)cols * 26 + 1 > cap) return -1;
        if (r) out[pos++] = '\n';
        for (int c = 0; c < cols; ++c) {
            if (c) out[pos++] = ',';
            pos += py_float_repr(vals[r * cols + c], out + pos);
        }
    }
    return pos;
}

// a16: rle_encoding (mask_utils.py:17-35) of M masks as the TEXT the RLE CSV holds (`" ".join(map(str, runs))`,
// inference.py:917-925), from the bbox-cropped packed words the instance tables use (demia_mask_crop_pack: mask m = rows x word
// columns, row-major, at payload[offsets[m]]; bit x & 31 of word x >> 5).  Runs are 1-based (start, length) pairs over the
// COLUMN-major flattening of an H-row frame: pixel (y, x) has index x * H + y, so a run ends at the bottom of a column
// unless the next column starts with a set pixel at row 0 (then the reference's scan glues them).  Only the box is visited.
// text_off [M + 1]: mask m's text = out[text_off[m] .. text_off[m + 1]) (empty for an empty mask).  Returns the bytes
// written, or -(bytes needed) if `cap` is too small.
extern "C" int64_t demia_host_rle_text(const uint32_t* payload, const int32_t* bbox, const int64_t* offsets, int64_t M, int H,
                                       char* out, int64_t cap, int64_t* text_off) {
    if (!payload || !bbox || !offsets || !out || !text_off || M < 0 || H <= 0) return 0;
    int64_t pos = 0;
    bool fits = true;
    char tmp[24];
    auto put = [&](int64_t v, bool first) {
        int k = 0;
        do { tmp[k++] = (char)('0' + v % 10); v /= 10; } while (v);
        const int need = k + (first ? 0 : 1);
        if (fits && pos + need <= cap) {
            if (!first) out[pos++] = ' ';
            while (k) out[pos++] = tmp[--k];
        } else {
            fits = false;
            pos += need;
        }
    };
    for (int64_t m = 0; m < M; ++m) {
        text_off[m] = pos;
        const int y0 = bbox[4 * m], x0 = bbox[4 * m + 1], y1 = bbox[4 * m + 2], x1 = bbox[4 * m + 3];
        if (y0 < 0) continue;
        const int c0 = x0 >> 5, cols = (x1 >> 5) - c0 + 1;
        const uint32_t* w = payload + offsets[m];
        int64_t start = -1, len = 0;
        bool first = true;
        for (int x = x0; x <= x1; ++x) {
            const uint32_t* col = w + ((x >> 5) - c0);
            const uint32_t bit = 1u << (x & 31);
            for (int y = y0; y <= y1; ++y) {
                if (!(col[(int64_t)(y - y0) * cols] & bit)) continue;
                const int64_t f = (int64_t)x * H + y;
                if (start >= 0 && f == start + len) { ++len; continue; }
                if (start >= 0) { put(start + 1, first); first = false; put(len, false); }
                start = f;
                len = 1;
            }
        }
        if (start >= 0) { put(start + 1, first); put(len, false); }
    }
    text_off[M] = pos;
    return fits ? pos : -pos;
}
