// Per-mask REGION machinery shared by the packed-mask kernels (maskops.hip, contours.hip).
//
// An instance mask of a 2048 x 2048 tile is a 512 KiB bit plane, but its set pixels live in a bounding box that is
// typically ~70 rows x 3 words.  Every per-mask stage therefore works on the bbox region only (+ a margin for
// dilation / the background ring of the hole fill): one workgroup per mask stages the region in LDS, runs the
// whole stage sequence there and writes the region back IN PLACE -- the frame outside the region is never read or
// written.  Regions that do not fit in LDS use the mask itself and a scratch plane in HBM/L2 with the same code.
//
// Flood fills (scipy binary_fill_holes = 4-connected background flood from outside; skimage label = 8-connected
// foreground flood) are done by SWEEPS: each wave owns a band of rows and walks it down and then up, one row per
// step, lane = one 32-pixel word of the row.  Inside a row the front runs through whole words at once
// (carry-propagate trick, fill_runs) and across words by lane shuffles, so a sweep moves a front any distance
// vertically and horizontally; a few rounds (until no wave changes a word) replace the hundreds of
// one-row-per-iteration passes of a Jacobi flood, which is what bounded the latency of the largest mask.
#pragma once
#include "common.h"

namespace mreg {

// every maximal run of 1-bits of p that contains a bit of g, completely (g must be a subset of p).
// q + s0 ripples a carry from each run start through the seed-free low part of the run; the same on the
// bit-reversed words gives the seed-free high part; a run is dropped iff the two parts are the whole run.
__device__ __forceinline__ uint32_t fill_runs(uint32_t g, uint32_t p) {
    const uint32_t q = p & ~g;
    const uint32_t s0 = p & ~(p << 1) & q;
    const uint32_t low = ((q + s0) ^ q) & q;
    const uint32_t pr = __brev(p), qr = pr & ~__brev(g);
    const uint32_t s0r = pr & ~(pr << 1) & qr;
    const uint32_t high = __brev(((qr + s0r) ^ qr) & qr);
    return p & ~(low & high);
}

// bits of word wx (pixels wx*32 .. wx*32+31) that lie in [x0, x1]
__device__ __forceinline__ uint32_t span_mask(int wx, int x0, int x1) {
    const int lo = max(x0 - wx * 32, 0), hi = min(x1 - wx * 32, 31);
    if (lo > hi) return 0u;
    const uint32_t upto_hi = hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u);
    return upto_hi & ~((1u << lo) - 1u);
}

struct Reg {
    uint32_t* A;        // current bits of the region
    uint32_t* B;        // second buffer (flood result / morphology output)
    int stride;         // words per row of A and B
    int rh, rw;         // region rows, words per row
    int ry0, wx0;       // frame row / word column of the region origin
    int H, W, wpr;      // frame
};

// Region of a (superset) bbox grown by `e` pixels, clipped to the frame.
__device__ __forceinline__ void region_of(int y0, int x0, int y1, int x1, int e, int H, int W, int& ry0, int& wx0, int& rh, int& rw) {
    ry0 = max(y0 - e, 0);
    const int ry1 = min(y1 + e, H - 1);
    wx0 = max(x0 - e, 0) >> 5;
    const int wx1 = min(x1 + e, W - 1) >> 5;
    rh = ry1 - ry0 + 1;
    rw = wx1 - wx0 + 1;
}

// R <- every bit of pass = P ^ inv that is connected to a seed bit of R (R must be a subset of pass on entry).
// EIGHT = false: 4-connected; true: 8-connected.  All threads of the block must call it.
template <bool EIGHT>
__device__ void flood(const uint32_t* P, uint32_t inv, uint32_t* R, int stride, int rh, int rw, int* s_changed) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int rpb = (rh + nw - 1) / nw;
    const int yb0 = min(wave * rpb, rh), yb1 = min(yb0 + rpb, rh);     // this wave's band of rows
    const int max_rounds = 2 * (rh + 32 * rw) + 8;
    for (int round = 0; round < max_rounds; ++round) {
        if (threadIdx.x == 0) *s_changed = 0;
        __syncthreads();
        bool ch = false;
        if (rw <= 64) {
            // The usual region (at most 64 words = 2048 pixels wide): lane = word, and the row the sweep has just left is the
            // value the lane computed a step ago -- it stays in a register (its left / right neighbours for the diagonal
            // links come by lane shuffle), the next row's words are requested before this row's front is worked out, and
            // no fence is needed between row steps: a step reads nothing another lane wrote during this sweep.  (The general
            // loop below re-read the left row from LDS behind a workgroup fence: three dependent LDS round trips per row.)
            const int lx = lane;
            const bool act = lx < rw;
            const int nrows = yb1 - yb0;
            for (int dir = 0; dir < 2 && nrows > 0; ++dir) {
                const int step = dir == 0 ? 1 : -1;
                int y = dir == 0 ? yb0 : yb1 - 1;
                const int ya0 = y - step;                                  // the neighbouring band's row (or outside the region)
                uint32_t prev = 0u;
                if (act && (unsigned)ya0 < (unsigned)rh) prev = R[ya0 * stride + lx];
                uint32_t pass_n = 0u, r_n = 0u;
                if (act) { pass_n = P[y * stride + lx] ^ inv; r_n = R[y * stride + lx]; }
                for (int k = 0; k < nrows; ++k, y += step) {
                    const uint32_t pass = pass_n, r = r_n;
                    if (k + 1 < nrows && act) { pass_n = P[(y + step) * stride + lx] ^ inv; r_n = R[(y + step) * stride + lx]; }
                    uint32_t vt = prev;
                    if (EIGHT) {
                        const uint32_t pl = __shfl_up(prev, 1, 64), pr = __shfl_down(prev, 1, 64);
                        vt |= (prev << 1) | (prev >> 1);
                        if (lx > 0) vt |= pl >> 31;
                        if (lx < rw - 1) vt |= pr << 31;
                    }
                    uint32_t v = (r | vt) & pass;
                    for (;;) {
                        v = fill_runs(v, pass);
                        uint32_t l = __shfl_up(v, 1, 64), rr = __shfl_down(v, 1, 64);
                        if (lane == 0) l = 0u;
                        if (lane == 63) rr = 0u;
                        const uint32_t nv = v | (((l >> 31) | (rr << 31)) & pass);
                        const bool grow = nv != v;
                        v = nv;
                        if (!__any(grow)) break;
                    }
                    if (act && v != r) { R[y * stride + lx] = v; ch = true; }
                    prev = v;
                }
            }
        } else
        for (int dir = 0; dir < 2; ++dir) {
            for (int k = 0; k < yb1 - yb0; ++k) {
                const int y = dir == 0 ? yb0 + k : yb1 - 1 - k;
                const int ya = dir == 0 ? y - 1 : y + 1;               // the row this sweep has just left
                const bool has_adj = (unsigned)ya < (unsigned)rh;
                for (int cx = 0; cx < rw; cx += 64) {
                    const int lx = cx + lane;
                    const bool act = lx < rw;
                    const int o = y * stride + lx, oa = ya * stride + lx;
                    uint32_t pass = 0u, r = 0u, vt = 0u, edge_l = 0u, edge_r = 0u;
                    if (act) {
                        pass = P[o] ^ inv;
                        r = R[o];
                        if (has_adj) {
                            const uint32_t a = R[oa];
                            vt = a;
                            if (EIGHT) {
                                vt |= (a << 1) | (a >> 1);
                                if (lx > 0) vt |= R[oa - 1] >> 31;
                                if (lx < rw - 1) vt |= R[oa + 1] << 31;
                            }
                        }
                        // neighbours across a 64-word chunk boundary come from memory
                        if (lane == 0 && lx > 0) edge_l = R[o - 1];
                        if (lane == 63 && lx < rw - 1) edge_r = R[o + 1];
                    }
                    uint32_t v = (r | vt) & pass;
                    for (;;) {
                        v = fill_runs(v, pass);
                        uint32_t l = __shfl_up(v, 1, 64), rr = __shfl_down(v, 1, 64);
                        if (lane == 0) l = edge_l;
                        if (lane == 63) rr = edge_r;
                        const uint32_t nv = v | (((l >> 31) | (rr << 31)) & pass);
                        const bool grow = nv != v;
                        v = nv;
                        if (!__any(grow)) break;
                    }
                    if (act && v != r) { R[o] = v; ch = true; }
                }
                // the next row step reads what this one wrote (other lanes' words included)
                __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup");
            }
        }
        if (ch) *s_changed = 1;
        __syncthreads();
        if (!*s_changed) break;
        __syncthreads();
    }
}

// B <- the OUTSIDE background of A: background pixels 4-connected to beyond the box (cy0, cx0, cy1, cx1) -- any box
// that contains every set pixel of A -- or lying on the image frame.
__device__ void outside_background(const Reg& g, int cy0, int cx0, int cy1, int cx1, int* s_changed) {
    const int tid = threadIdx.x, nt = blockDim.x, n = g.rh * g.rw;
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const int y = g.ry0 + ly, wx = g.wx0 + lx;
        const uint32_t mk = g.A[ly * g.stride + lx];
        uint32_t seed;
        if (y < cy0 || y > cy1 || y == 0 || y == g.H - 1) seed = 0xFFFFFFFFu;
        else {
            seed = ~span_mask(wx, cx0, cx1);
            if (wx == 0) seed |= 1u;
            if (wx == g.wpr - 1) seed |= 0xFFFFFFFFu << ((g.W - 1) & 31);   // pixel W-1 and the padding bits beyond it
        }
        g.B[ly * g.stride + lx] = ~mk & seed;
    }
    __syncthreads();
    flood<false>(g.A, 0xFFFFFFFFu, g.B, g.stride, g.rh, g.rw, s_changed);
}

// scipy.ndimage.binary_fill_holes on the region: A <- A | (background not 4-connected to the outside).  Uses B.
__device__ void fill_holes(const Reg& g, int cy0, int cx0, int cy1, int cx1, int* s_changed) {
    const int tid = threadIdx.x, nt = blockDim.x, n = g.rh * g.rw;
    outside_background(g, cy0, cx0, cy1, cx1, s_changed);
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const uint32_t valid = (g.wx0 + lx == g.wpr - 1 && (g.W & 31)) ? ((1u << (g.W & 31)) - 1u) : 0xFFFFFFFFu;
        g.A[ly * g.stride + lx] = ~g.B[ly * g.stride + lx] & valid;
    }
    __syncthreads();
}

// With B = outside_background(A): does A enclose any background?  s_flag is a shared int.
__device__ bool has_holes(const Reg& g, int* s_flag) {
    const int tid = threadIdx.x, nt = blockDim.x, n = g.rh * g.rw;
    if (tid == 0) *s_flag = 0;
    __syncthreads();
    bool h = false;
    for (int i = tid; i < n; i += nt) {
        const int o = (i / g.rw) * g.stride + i % g.rw;
        if (~(g.A[o] | g.B[o])) h = true;            // padding bits beyond W are seeds of B, so they never count
    }
    if (h) *s_flag = 1;
    __syncthreads();
    const bool res = *s_flag != 0;
    __syncthreads();
    return res;
}

// 4 x the Euler number of A for 8-connectivity (components - holes), by Gray's bit-quad counts over every 2 x 2
// window of the zero-padded region: 4E = n(Q1) - n(Q3) - 2 n(QD).  A popcount reduction: no propagation at all.
__device__ int euler8_x4(const Reg& g, int* s_acc) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int cols = g.rw + 1, n = (g.rh + 1) * cols;
    if (tid == 0) *s_acc = 0;
    __syncthreads();
    int acc = 0;
    for (int i = tid; i < n; i += nt) {
        const int uy = i / cols - 1, lx = i - (uy + 1) * cols;     // window = rows uy, uy+1; right pixels in word lx
        uint32_t U = 0u, Up = 0u, D = 0u, Dp = 0u;
        if (uy >= 0) {
            if (lx < g.rw) U = g.A[uy * g.stride + lx];
            if (lx > 0) Up = g.A[uy * g.stride + lx - 1];
        }
        if (uy + 1 < g.rh) {
            if (lx < g.rw) D = g.A[(uy + 1) * g.stride + lx];
            if (lx > 0) Dp = g.A[(uy + 1) * g.stride + lx - 1];
        }
        const uint32_t a = (U << 1) | (Up >> 31), b = U, c = (D << 1) | (Dp >> 31), d = D;
        const uint32_t odd = a ^ b ^ c ^ d;
        const uint32_t two = (a & b) | (a & c) | (a & d) | (b & c) | (b & d) | (c & d);
        const uint32_t qd = (a & d & ~b & ~c) | (b & c & ~a & ~d);
        acc += __popc(odd & ~two) - __popc(odd & two) - 2 * __popc(qd);
    }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_down(acc, o, 64);
    if ((tid & 63) == 0 && acc) atomicAdd(s_acc, acc);
    __syncthreads();
    const int res = *s_acc;
    __syncthreads();
    return res;
}

// skimage erosion / dilation with the 3x3 cross; 'reflect' border = the frame edge replicates, everything else
// outside the region is background (the region carries a margin of zeros).  B <- op(A); caller swaps.
template <bool DILATE>
__device__ void morph_cross(const Reg& g) {
    const int tid = threadIdx.x, nt = blockDim.x, n = g.rh * g.rw;
    const int last = (g.W - 1) & 31;
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const int y = g.ry0 + ly, wx = g.wx0 + lx;
        const int o = ly * g.stride + lx;
        const uint32_t c = g.A[o];
        const uint32_t up = y == 0 ? c : (ly == 0 ? 0u : g.A[o - g.stride]);
        const uint32_t dn = y == g.H - 1 ? c : (ly == g.rh - 1 ? 0u : g.A[o + g.stride]);
        const uint32_t lbit = wx == 0 ? (c & 1u) : (lx == 0 ? 0u : g.A[o - 1] >> 31);
        const uint32_t left = (c << 1) | lbit;
        uint32_t right, valid = 0xFFFFFFFFu;
        if (wx == g.wpr - 1) { right = (c >> 1) | (((c >> last) & 1u) << last); valid = last == 31 ? 0xFFFFFFFFu : ((2u << last) - 1u); }
        else right = (c >> 1) | (lx == g.rw - 1 ? 0u : (g.A[o + 1] & 1u) << 31);
        g.B[o] = (DILATE ? (c | up | dn | left | right) : (c & up & dn & left & right)) & valid;
    }
    __syncthreads();
}

// skimage.measure.label(A).max() > 1 (8-connected).  A mask without holes has exactly E8 components, and the hole
// test is the cheap kind of flood (seeded all around the box, every wave busy from the first round); only a mask
// WITH holes pays for the flood of one component from its first pixel (a single travelling front).
// Uses B; (cy0, cx0, cy1, cx1) as in outside_background; s_first is a shared int.
__device__ bool more_than_one_component(const Reg& g, int cy0, int cx0, int cy1, int cx1, int* s_first, int* s_changed) {
    const int tid = threadIdx.x, nt = blockDim.x, n = g.rh * g.rw;
    outside_background(g, cy0, cx0, cy1, cx1, s_changed);
    if (!has_holes(g, s_first)) return euler8_x4(g, s_first) > 4;
    if (tid == 0) *s_first = 0x7FFFFFFF;
    __syncthreads();
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        if (g.A[ly * g.stride + lx]) { atomicMin(s_first, i); break; }     // i ascends per thread: its first hit is its minimum
    }
    __syncthreads();
    const int first = *s_first;
    __syncthreads();
    if (first == 0x7FFFFFFF) return false;                                    // empty mask: no component at all
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const uint32_t mk = g.A[ly * g.stride + lx];
        g.B[ly * g.stride + lx] = i == first ? (mk & (0u - mk)) : 0u;
    }
    __syncthreads();
    flood<true>(g.A, 0u, g.B, g.stride, g.rh, g.rw, s_changed);
    if (tid == 0) *s_changed = 0;
    __syncthreads();
    bool diff = false;
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        if (g.A[ly * g.stride + lx] != g.B[ly * g.stride + lx]) diff = true;
    }
    if (diff) *s_changed = 1;
    __syncthreads();
    const bool res = *s_changed != 0;
    __syncthreads();
    return res;
}

}  // namespace mreg
