// Per-instance contour extraction and morphometrics on bit-packed masks.
//
//   trace    cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE)   inference.py:1164, 2605
//            + cv2.contourArea (shoelace, exact integer) + cv2.arcLength (per-segment f32 sqrt,
//            accumulated in double)                                        inference.py:1175, 2607; measurements.py:134-135
//   measure  measurements.py:114-233: minAreaRect (convex hull + f32 rotating calipers) -> boxPoints
//            -> int truncation -> imutils order_points -> mid-point distances; fitEllipse
//            (algebraic least squares); the 12 derived values of the CSV row.
//
// Parallel decomposition (order/compare work, latency-bound).  The usual instance mask -- one component, no holes
// (hole test + Euler number, maskregion.h) -- has ONE border: up to 64 walkers start on the flanks of evenly spaced
// rows and each walks to the next walker's start state; stitched in cycle order from the raster-first pixel this is
// OpenCV's sequential walk point for point.  Any other mask: a border can only start at a pixel whose west,
// north-west, north and north-east neighbours are background; every such pixel that also has TRUE outside background
// to its west (not a hole) is traced by one lane with Suzuki-Abe's 8-neighbour rule, and the trace is kept iff it
// never meets a pixel that precedes its start in raster order -- i.e. iff the start is the component's first pixel,
// which is where OpenCV's raster scan starts the same border; valid traces are walked again to emit their points.
// Measurements: one wave per contour (rank sort + Jacobi least squares over the lanes).
// Compiled with -ffp-contract=off: the f32 calipers follow OpenCV's operation order.
#include "common.h"
#include "maskregion.h"

namespace {

// Direction codes of OpenCV's tracer: 0 E, 1 NE, 2 N, 3 NW, 4 W, 5 SW, 6 S, 7 SE (y grows downwards).
// dx + 1 / dy + 1 packed 2 bits per code -- no table in memory on the critical path of the walk.
__device__ __forceinline__ int dir_dx(int s) { return (int)((0x901Au >> (2 * s)) & 3u) - 1; }
__device__ __forceinline__ int dir_dy(int s) { return (int)((0xA901u >> (2 * s)) & 3u) - 1; }

// Bit image of one mask restricted to its region [ry0, ry0+rh) x word columns [wx0, wx0+rw): everything
// outside the region is background by construction.  `p` points at the region's first word, `stride` = words per
// row.  PAD = true: the LDS copy carries a zero row above / below and a zero word left / right of the region, so
// no probe of a walk needs a bounds check; PAD = false (the mask itself in HBM/L2): probes are clipped.
struct BitImg {
    const uint32_t* p;
    int stride, ry0, wx0, rh, rw;
};

// pixels x-1, x, x+1 of row y as bits 0..2
template <bool PAD>
__device__ __forceinline__ uint32_t row3(const BitImg& im, int x, int y) {
    const int ly = y - im.ry0;
    const int lk = ((x - 1) >> 5) - im.wx0;              // word of pixel x-1 (arithmetic shift: -1 for x = 0)
    const uint32_t* row = im.p + ly * im.stride;
    uint32_t lo, hi;
    if (PAD) { lo = row[lk]; hi = row[lk + 1]; }
    else {
        if ((unsigned)ly >= (unsigned)im.rh) return 0u;
        lo = (unsigned)lk < (unsigned)im.rw ? row[lk] : 0u;
        hi = (unsigned)(lk + 1) < (unsigned)im.rw ? row[lk + 1] : 0u;
    }
    return __funnelshift_r(lo, hi, (x - 1) & 31) & 7u;
}

// the 8 neighbours of (x, y): bit s = pixel at (x + dx[s], y + dy[s])
template <bool PAD>
__device__ __forceinline__ uint32_t nbr8(const BitImg& im, int x, int y) {
    const uint32_t n = row3<PAD>(im, x, y - 1), c = row3<PAD>(im, x, y), d = row3<PAD>(im, x, y + 1);
    return ((c >> 2) & 1u) | (((n >> 2) & 1u) << 1) | (((n >> 1) & 1u) << 2) | ((n & 1u) << 3) |
           ((c & 1u) << 4) | ((d & 1u) << 5) | (((d >> 1) & 1u) << 6) | (((d >> 2) & 1u) << 7);
}

// first set neighbour looking clockwise from direction `from` (exclusive): from-1, from-2, ..., from-7; -1 if none
__device__ __forceinline__ int first_clockwise(uint32_t n8, int from) {
    for (int k = 1; k < 8; ++k) {
        const int s = (from - k) & 7;
        if ((n8 >> s) & 1u) return s;
    }
    return -1;
}

struct TraceResult {
    int valid;
    int npts;
    long area2;      // signed shoelace sum
    double perimeter;
};

// Suzuki-Abe border following from the start pixel (sx, sy), as cv::findContours walks an outer border.
// Points of CHAIN_APPROX_SIMPLE are counted always and written to pts[2*i], pts[2*i+1] while i < cap.
template <bool PAD>
__device__ TraceResult trace_border(const BitImg& im, int sx, int sy, int* pts, int cap) {
    TraceResult r;
    r.valid = 1; r.npts = 0; r.area2 = 0; r.perimeter = 0.0;
    uint32_t n8 = nbr8<PAD>(im, sx, sy);
    int s = first_clockwise(n8, 4);                       // the W pixel is background: look clockwise from there
    if (s < 0) {  // single pixel
        if (cap > 0) { pts[0] = sx; pts[1] = sy; }
        r.npts = 1;
        return r;
    }
    const int i1x = sx + dir_dx(s), i1y = sy + dir_dy(s);
    int x = sx, y = sy;
    int prev_s = s ^ 4;
    int fx = 0, fy = 0, lx = 0, ly = 0;  // first / last emitted point
    const long max_steps = 4L * ((long)im.rh * im.rw * 32 + 4);
    for (long step = 0; step < max_steps; ++step) {
        // first set neighbour in the order s+1, s+2, ... (counter-clockwise)
        const uint32_t rot = ((n8 | (n8 << 8)) >> ((s + 1) & 7)) & 0xFFu;
        s = (s + 1 + (__ffs((int)rot) - 1)) & 7;
        const int qx = x + dir_dx(s), qy = y + dir_dy(s);
        if (s != prev_s) {
            if (r.npts < cap) { pts[2 * r.npts] = x; pts[2 * r.npts + 1] = y; }
            if (r.npts == 0) { fx = x; fy = y; }
            else {
                const float ddx = (float)(x - lx), ddy = (float)(y - ly);
                r.perimeter += (double)sqrtf(ddx * ddx + ddy * ddy);
            }
            lx = x; ly = y;
            ++r.npts;
            prev_s = s;
        }
        r.area2 += (long)x * qy - (long)y * qx;
        if (qy < sy || (qy == sy && qx < sx)) { r.valid = 0; return r; }
        const bool done = (qx == sx && qy == sy && x == i1x && y == i1y);
        x = qx; y = qy;
        if (done) break;
        s = (s + 4) & 7;
        n8 = nbr8<PAD>(im, x, y);
    }
    if (r.npts > 1) {
        const float ddx = (float)(fx - lx), ddy = (float)(fy - ly);
        r.perimeter += (double)sqrtf(ddx * ddx + ddy * ddy);
    }
    return r;
}

// ---- one border, many walkers ---------------------------------------------------------------------------------
// A walk is a cycle in the space of states (pixel, direction to the previous pixel), and the start rule of the
// tracer ("from a background 4-neighbour look clockwise for the first set pixel") puts ANY border pixel with that
// background neighbour on the cycle of the border between its component and that background.  A mask that is one
// component without holes has a single border, so walkers started on the left and right flanks of evenly spaced
// rows all sit on the cycle of the raster-first pixel; each walks until it reaches the next walker's start state and
// the segments, taken in cycle order from the raster-first pixel, are exactly the sequential walk (the vertex
// test at a seam sees the true previous move: start state (p, s) means the walk arrived by the move s ^ 4).
struct Walkers {
    int sx[64], sy[64], ss[64], nxt[64], np[64], fx[64], fy[64], lx[64], ly[64], off[64];
    long long a2[64];
    double per[64];
    int K, fail, total, base;
};

template <bool EMIT>
__device__ void walk_segment(const BitImg& im, const uint32_t* startmap, Walkers& w, int k, int* out) {
    int x = w.sx[k], y = w.sy[k], s = w.ss[k];
    int prev_s = s ^ 4;
    int np = 0, fx = 0, fy = 0, lx = 0, ly = 0, nxt = -1;
    long long a2 = 0;
    double per = 0.0;
    uint32_t n8 = nbr8<true>(im, x, y);
    const long max_steps = 4L * ((long)im.rh * im.rw * 32 + 4);
    for (long step = 0; step < max_steps; ++step) {
        const uint32_t rot = ((n8 | (n8 << 8)) >> ((s + 1) & 7)) & 0xFFu;
        s = (s + 1 + (__ffs((int)rot) - 1)) & 7;
        const int qx = x + dir_dx(s), qy = y + dir_dy(s);
        if (s != prev_s) {
            if (EMIT) { out[2 * np] = x; out[2 * np + 1] = y; }
            else {
                if (np == 0) { fx = x; fy = y; }
                else {
                    const float ddx = (float)(x - lx), ddy = (float)(y - ly);
                    per += (double)sqrtf(ddx * ddx + ddy * ddy);
                }
                lx = x; ly = y;
            }
            ++np;
            prev_s = s;
        }
        if (!EMIT) a2 += (long long)x * qy - (long long)y * qx;
        x = qx; y = qy;
        s = (s + 4) & 7;                                   // state at q: direction back to where we came from
        const uint32_t sm = startmap[(y - im.ry0) * im.stride + ((x >> 5) - im.wx0)];
        if ((sm >> (x & 31)) & 1u) {
            for (int j = 0; j < w.K; ++j)
                if (w.sx[j] == x && w.sy[j] == y && w.ss[j] == s) { nxt = j; break; }
            if (nxt >= 0) break;
        }
        n8 = nbr8<true>(im, x, y);
    }
    if (!EMIT) {
        w.nxt[k] = nxt; w.np[k] = np; w.fx[k] = fx; w.fy[k] = fy; w.lx[k] = lx; w.ly[k] = ly; w.a2[k] = a2; w.per[k] = per;
        if (nxt < 0) w.fail = 1;
    }
}

// LDS words per region buffer (mask bits with a zero ring; outside flood, later the walkers' start map) and
// candidate slots.  As in maskops.hip the masks are split over a SMALL variant (10 KiB of LDS: fits beside a
// convolution's workgroups on the other stream) and a large one; a small-region mask with too many candidates is
// handed on through count[m] = -1.
constexpr int TRACE_WORDS_SMALL = 1024, CAND_SMALL = 512;
constexpr int TRACE_WORDS = 8192, CAND_MAX = 3072;

struct ContourP {
    const uint32_t* masks;
    uint32_t* scratch;
    const int* bbox;
    int M, H, W, C;          // C = max contours per mask
    int max_points;
    int* count;              // [M]
    int* info;               // [M, C, 4] sx, sy, npts, offset
    double* red;             // [M, C, 2] area, perimeter
    int* points;             // [max_points, 2]
    int* counters;           // [0] = points used, [1] = error flags, [2] = masks whose walkers fell back, [3] = masks walked by walkers
};

// One workgroup per mask.  (1) region (+1 ring) -> LDS, (2) flood the outside background (what RETR_EXTERNAL
// needs to tell an outer border from a component sitting in a hole), hole test and Euler number, (3a) the usual
// mask -- one component, no holes -- has one border: up to 64 walkers share it (above); (3b) otherwise every
// start candidate (W, NW, N, NE background, W pixel outside background) is walked by one lane and kept iff the walk
// never meets a pixel that precedes its start in raster order, i.e. iff it is where OpenCV's raster scan starts
// that border; valid ones are walked a second time to emit their points.
template <int TW, int CM>
__device__ void contour_trace_one(const ContourP& p, const int m, int* __restrict__ worklist) {
    constexpr bool SMALL = TW != TRACE_WORDS;
    extern __shared__ __attribute__((aligned(16))) uint32_t smem[];
    uint32_t* regA = smem;                              // TW
    uint32_t* regB = smem + TW;                         // TW
    int* cand = reinterpret_cast<int*>(smem + 2 * TW);  // CM
    __shared__ int ncand, ncont, s_changed, first_cand;
    __shared__ Walkers wk;
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wpr = (p.W + 31) >> 5;
    const int y0 = p.bbox[m * 4 + 0], x0 = p.bbox[m * 4 + 1], y1 = p.bbox[m * 4 + 2], x1 = p.bbox[m * 4 + 3];
    if (y0 < 0) { if (tid == 0 && SMALL) p.count[m] = 0; return; }
    mreg::Reg g;
    g.H = p.H; g.W = p.W; g.wpr = wpr;
    mreg::region_of(y0, x0, y1, x1, 1, p.H, p.W, g.ry0, g.wx0, g.rh, g.rw);
    const int ps = g.rw + 2, pn = (g.rh + 2) * ps;      // padded LDS image
    if (SMALL ? pn > TW : (pn <= TRACE_WORDS_SMALL && p.count[m] != -1)) {           // the other variant's mask
        if (SMALL && worklist && tid == 0) worklist[2 + atomicAdd(&worklist[0], 1)] = m;
        return;
    }
    if (tid == 0) { ncand = 0; ncont = 0; first_cand = 0x7FFFFFFF; wk.K = 0; wk.fail = 0; }
    const int n = g.rh * g.rw;
    const bool use_lds = pn <= TW;
    const uint32_t* home = p.masks + (long)m * p.H * wpr + (long)g.ry0 * wpr + g.wx0;
    if (use_lds) {
        g.A = regA + ps + 1; g.B = regB + ps + 1; g.stride = ps;
        for (int i = tid; i < pn; i += nt) {
            const int py = i / ps, px = i - py * ps;
            const bool in = py >= 1 && py <= g.rh && px >= 1 && px <= g.rw;
            regA[i] = in ? home[(long)(py - 1) * wpr + px - 1] : 0u;
        }
    } else {
        g.A = const_cast<uint32_t*>(home); g.B = p.scratch + (long)m * p.H * wpr + (long)g.ry0 * wpr + g.wx0; g.stride = wpr;
    }
    __syncthreads();
    // ---- outside background R (4-connected from beyond the bbox / the image frame) -> g.B -------------------
    mreg::outside_background(g, y0, x0, y1, x1, &s_changed);
    // one component without holes (the usual instance mask): one border
    const bool simple = !mreg::has_holes(g, &s_changed) && mreg::euler8_x4(g, &s_changed) == 4;
    // ---- candidates: W, NW, N, NE background and the W pixel is OUTSIDE background (or off the frame) ------
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const int y = g.ry0 + ly, wx = g.wx0 + lx;
        const int o = ly * g.stride + lx;
        const uint32_t mm = g.A[o];
        if (!mm) continue;
        const uint32_t ml = (mm << 1) | (lx > 0 ? g.A[o - 1] >> 31 : 0u);
        const uint32_t out_l = (g.B[o] << 1) | (lx > 0 ? g.B[o - 1] >> 31 : (wx == 0 ? 1u : 0u));
        uint32_t u = 0u, ul = 0u, ur = 0u;
        if (ly > 0) {
            u = g.A[o - g.stride];
            ul = (u << 1) | (lx > 0 ? g.A[o - g.stride - 1] >> 31 : 0u);
            ur = (u >> 1) | (lx < g.rw - 1 ? g.A[o - g.stride + 1] << 31 : 0u);
        }
        uint32_t c = mm & ~ml & out_l & ~u & ~ul & ~ur;
        while (c) {
            const int b = __ffs((int)c) - 1;
            c &= c - 1u;
            const int code = y * p.W + wx * 32 + b;
            if (!simple) {                                    // (a simple mask only needs its raster-first pixel)
                const int slot = atomicAdd(&ncand, 1);
                if (slot < CM) cand[slot] = code;
            }
            atomicMin(&first_cand, code);
        }
    }
    __syncthreads();
    if (ncand > CM) {
        if (tid == 0) {
            if (SMALL) {                                      // more candidates than this variant holds: the large one takes it
                p.count[m] = -1;
                if (worklist) worklist[2 + atomicAdd(&worklist[0], 1)] = m;
            }
            else { atomicOr(&p.counters[1], 1); p.count[m] = 0; }
        }
        return;
    }
    const BitImg im{g.A, g.stride, g.ry0, g.wx0, g.rh, g.rw};
    bool walkers_done = false;
    if (simple && use_lds) {
        // ---------------- (3a) one border, up to 64 walkers (wave 0; lane k = walker k) --------------------------
        const int Fy = first_cand / p.W, Fx = first_cand - Fy * p.W;
        for (int i = tid; i < n; i += nt) g.B[(i / g.rw) * g.stride + i % g.rw] = 0u;      // start map
        __syncthreads();
        const int y_last = min(y1, g.ry0 + g.rh - 1);
        const int nrows = y_last - Fy + 1;
        const int K = min(64, nrows);
        if (tid < 64) {
            const int k = tid;
            wk.sx[k] = -1; wk.sy[k] = -1; wk.ss[k] = -1; wk.nxt[k] = -1; wk.np[k] = 0;
            if (k < K) {
                const int yk = Fy + (int)(((long)k * nrows) / K);
                const uint32_t* row = g.A + (yk - g.ry0) * g.stride;
                int xs = -1;
                if ((k & 1) == 0) { for (int lx = 0; lx < g.rw; ++lx) if (row[lx]) { xs = (g.wx0 + lx) * 32 + __ffs((int)row[lx]) - 1; break; } }
                else { for (int lx = g.rw - 1; lx >= 0; --lx) if (row[lx]) { xs = (g.wx0 + lx) * 32 + 31 - __clz((int)row[lx]); break; } }
                if (xs >= 0) {
                    const int s0 = first_clockwise(nbr8<true>(im, xs, yk), (k & 1) ? 0 : 4);
                    if (s0 >= 0 || k == 0) { wk.sx[k] = xs; wk.sy[k] = yk; wk.ss[k] = s0; }
                }
            }
        }
        __syncthreads();
        if (tid == 0) wk.K = K;
        const bool single_pixel = wk.ss[0] < 0;               // the whole mask is one pixel
        if (tid < 64 && !single_pixel && wk.ss[tid] >= 0)
            atomicOr(&g.B[(wk.sy[tid] - g.ry0) * g.stride + ((wk.sx[tid] >> 5) - g.wx0)], 1u << (wk.sx[tid] & 31));
        __syncthreads();
        if (tid < 64 && !single_pixel && wk.ss[tid] >= 0) walk_segment<false>(im, g.B, wk, tid, nullptr);
        __syncthreads();
        if (tid == 0) {
            if (single_pixel) { wk.total = 1; wk.fail = 0; }
            else if (!wk.fail) {
                // stitch in cycle order from the raster-first pixel
                int active = 0;
                for (int k = 0; k < K; ++k) active += wk.ss[k] >= 0;
                int k = 0, seen = 0, total = 0;
                double per = 0.0;
                long long a2 = 0;
                int havev = 0, vx0 = 0, vy0 = 0, vlx = 0, vly = 0;
                do {
                    wk.off[k] = total;
                    total += wk.np[k];
                    a2 += wk.a2[k];
                    if (wk.np[k] > 0) {
                        if (havev) {
                            const float ddx = (float)(wk.fx[k] - vlx), ddy = (float)(wk.fy[k] - vly);
                            per += (double)sqrtf(ddx * ddx + ddy * ddy);
                        } else { havev = 1; vx0 = wk.fx[k]; vy0 = wk.fy[k]; }
                        per += wk.per[k];
                        vlx = wk.lx[k]; vly = wk.ly[k];
                    }
                    k = wk.nxt[k];
                    ++seen;
                } while (k != 0 && k >= 0 && seen <= K);
                if (k != 0 || seen != active) wk.fail = 1;
                else {
                    if (total > 1) {
                        const float ddx = (float)(vx0 - vlx), ddy = (float)(vy0 - vly);
                        per += (double)sqrtf(ddx * ddx + ddy * ddy);
                    }
                    wk.total = total; wk.a2[0] = a2; wk.per[0] = per;
                }
            }
            if (!wk.fail) {
                const int total = wk.total;
                ncont = 1;
                const int off = atomicAdd(&p.counters[0], total + 4);
                int* inf = p.info + (long)m * p.C * 4;
                inf[0] = Fx; inf[1] = Fy; inf[2] = total; inf[3] = off;
                double* rd = p.red + (long)m * p.C * 2;
                const long long a2 = single_pixel ? 0 : wk.a2[0];
                rd[0] = (double)(a2 < 0 ? -a2 : a2) * 0.5;
                rd[1] = single_pixel ? 0.0 : wk.per[0];
                wk.base = off;
                atomicAdd(&p.counters[3], 1);
                if (off + total + 4 > p.max_points) { atomicOr(&p.counters[1], 4); inf[2] = 0; wk.base = -1; }
                else if (single_pixel) { p.points[2L * off] = Fx; p.points[2L * off + 1] = Fy; }
            }
        }
        __syncthreads();
        if (!wk.fail) {
            if (tid < 64 && !single_pixel && wk.ss[tid] >= 0 && wk.base >= 0)
                walk_segment<true>(im, g.B, wk, tid, p.points + 2L * (wk.base + wk.off[tid]));
            walkers_done = true;
        } else {
            // never expected; the sequential walk below is always right
            __syncthreads();
            if (tid == 0) { cand[0] = first_cand; ncand = 1; ncont = 0; atomicAdd(&p.counters[2], 1); }
            __syncthreads();
        }
    } else if (simple) {
        __syncthreads();
        if (tid == 0) { cand[0] = first_cand; ncand = 1; }    // HBM-resident region: one sequential walk
        __syncthreads();
    }
    if (!walkers_done) {
        // ---------------- (3b) one lane per candidate ---------------------------------------------------------
        for (int c = tid; c < ncand; c += nt) {
            const int code = cand[c];
            const int sy = code / p.W, sx = code - sy * p.W;
            const TraceResult r = use_lds ? trace_border<true>(im, sx, sy, nullptr, 0) : trace_border<false>(im, sx, sy, nullptr, 0);
            if (!r.valid) continue;
            const int slot = atomicAdd(&ncont, 1);
            if (slot >= p.C) { atomicOr(&p.counters[1], 2); continue; }
            // each contour reserves 4 spare slots: the measurement kernel carves its scratch pools by this
            // offset, and the hull stacks need up to n + 3 entries
            const int off = atomicAdd(&p.counters[0], r.npts + 4);
            int* inf = p.info + ((long)m * p.C + slot) * 4;
            inf[0] = sx; inf[1] = sy; inf[2] = r.npts; inf[3] = off;
            double* rd = p.red + ((long)m * p.C + slot) * 2;
            rd[0] = (double)(r.area2 < 0 ? -r.area2 : r.area2) * 0.5;
            rd[1] = r.perimeter;
            if (off + r.npts + 4 > p.max_points) { atomicOr(&p.counters[1], 4); inf[2] = 0; continue; }
            if (use_lds) trace_border<true>(im, sx, sy, p.points + 2L * off, r.npts);
            else trace_border<false>(im, sx, sy, p.points + 2L * off, r.npts);
        }
    }
    __syncthreads();
    if (tid == 0) p.count[m] = ncont < p.C ? ncont : p.C;
}

// One workgroup per mask; `worklist` (optional, SMALL variant): the masks it leaves to the large variant are appended to
// worklist[2 ..] (count in worklist[0]).
template <int TW, int CM>
__global__ __launch_bounds__(256) void contour_trace_kernel(const ContourP p, int* __restrict__ worklist) {
    contour_trace_one<TW, CM>(p, blockIdx.x, worklist);
}

// The large variant over that list: a fixed grid, every workgroup takes the next listed mask until the list is empty
// (as mask_program_list_kernel in maskops.hip: one 78-KiB workgroup per MASK mostly found out that its mask was small).
template <int TW, int CM>
__global__ __launch_bounds__(256) void contour_trace_list_kernel(const ContourP p, int* __restrict__ worklist) {
    __shared__ int s_next;
    const int count = worklist[0];
    for (;;) {
        if (threadIdx.x == 0) s_next = atomicAdd(&worklist[1], 1);
        __syncthreads();
        const int i = s_next;
        if (i >= count) break;                               // (block-uniform; every workgroup gets here)
        contour_trace_one<TW, CM>(p, worklist[2 + i], nullptr);
        __syncthreads();
    }
}

// =============================================================================================
// measurements: one wave per contour
// =============================================================================================
__device__ __forceinline__ bool pt_less(const int* pts, int a, int b) {
    const int ax = pts[2 * a], ay = pts[2 * a + 1], bx = pts[2 * b], by = pts[2 * b + 1];
    if (ax != bx) return ax < bx;
    if (ay != by) return ay < by;
    return a < b;
}

__device__ void heapsort_idx(const int* pts, int* idx, int n) {
    for (int i = 0; i < n; ++i) idx[i] = i;
    auto sift = [&](int start, int end) {
        int root = start;
        for (;;) {
            int child = 2 * root + 1;
            if (child > end) break;
            if (child + 1 <= end && pt_less(pts, idx[child], idx[child + 1])) ++child;
            if (pt_less(pts, idx[root], idx[child])) { const int t = idx[root]; idx[root] = idx[child]; idx[child] = t; root = child; }
            else break;
        }
    };
    for (int s = (n - 2) / 2; s >= 0; --s) sift(s, n - 1);
    for (int e = n - 1; e > 0; --e) {
        const int t = idx[e]; idx[e] = idx[0]; idx[0] = t;
        sift(0, e - 1);
    }
}

__device__ __forceinline__ int sgn(long v) { return (v > 0) - (v < 0); }

// cv::Sklansky_ on the sorted order; writes the stack, returns its size
__device__ int sklansky(const int* pts, const int* order, int start, int end, int* stack, int nsign, int sign2) {
    const int incr = end > start ? 1 : -1;
    int pprev = start, pcur = pprev + incr, pnext = pcur + incr;
    int size = 3;
#define PX(i) pts[2 * order[i]]
#define PY(i) pts[2 * order[i] + 1]
    if (start == end || (PX(start) == PX(end) && PY(start) == PY(end))) { stack[0] = start; return 1; }
    stack[0] = pprev; stack[1] = pcur; stack[2] = pnext;
    end += incr;
    while (pnext != end) {
        const int cury = PY(pcur), nexty = PY(pnext);
        const int by = nexty - cury;
        if (sgn(by) != nsign) {
            const int ax = PX(pcur) - PX(pprev), bx = PX(pnext) - PX(pcur), ay = cury - PY(pprev);
            const long conv = (long)ay * bx - (long)ax * by;
            if (sgn(conv) == sign2 && (ax != 0 || ay != 0)) {
                pprev = pcur; pcur = pnext; pnext += incr;
                stack[size++] = pnext;
            } else if (pprev == start) {
                pcur = pnext; stack[1] = pcur; pnext += incr; stack[2] = pnext;
            } else {
                stack[size - 2] = pnext; pcur = pprev; pprev = stack[size - 4]; --size;
            }
        } else {
            pnext += incr;
            stack[size - 1] = pnext;
        }
    }
#undef PX
#undef PY
    return --size;
}

// cv::convexHull(points, clockwise=false) -> indices into pts; returns count.  work: 3*n+4 ints
// `order` must hold the point indices sorted by (x, y, index)
__device__ int convex_hull_idx(const int* pts, int n, const int* order, int* stack, int* hull) {
    int miny = 0, maxy = 0;
    for (int i = 1; i < n; ++i) {
        const int y = pts[2 * order[i] + 1];
        if (pts[2 * order[miny] + 1] > y) miny = i;
        if (pts[2 * order[maxy] + 1] < y) maxy = i;
    }
    int nout = 0;
    if (pts[2 * order[0]] == pts[2 * order[n - 1]] && pts[2 * order[0] + 1] == pts[2 * order[n - 1] + 1]) {
        hull[nout++] = order[0];
        return nout;
    }
    int* tl = stack;
    int tlc = sklansky(pts, order, 0, maxy, tl, -1, 1);
    int* tr = stack + tlc;
    int trc = sklansky(pts, order, n - 1, maxy, tr, -1, -1);
    { int* t = tl; tl = tr; tr = t; const int c = tlc; tlc = trc; trc = c; }  // !clockwise
    for (int i = 0; i < tlc - 1; ++i) hull[nout++] = order[tl[i]];
    for (int i = trc - 1; i > 0; --i) hull[nout++] = order[tr[i]];
    const int stop_idx = trc > 2 ? tr[1] : (tlc > 2 ? tl[tlc - 2] : -1);
    // the stacks are reused below, but stop_idx is a position in `order`, still valid
    int* bl = stack;
    int blc = sklansky(pts, order, 0, miny, bl, 1, -1);
    int* br = stack + blc;
    int brc = sklansky(pts, order, n - 1, miny, br, 1, 1);
    if (stop_idx >= 0) {
        const int check_idx = blc > 2 ? bl[1] : (blc + brc > 2 ? br[2 - blc] : -1);
        if (check_idx == stop_idx || (check_idx >= 0 && pts[2 * order[check_idx]] == pts[2 * order[stop_idx]] &&
                                      pts[2 * order[check_idx] + 1] == pts[2 * order[stop_idx] + 1])) {
            blc = blc < 2 ? blc : 2;
            brc = brc < 2 ? brc : 2;
        }
    }
    for (int i = 0; i < blc - 1; ++i) hull[nout++] = order[bl[i]];
    for (int i = brc - 1; i > 0; --i) hull[nout++] = order[br[i]];
    if (nout >= 3) {
        int min_i = 0, max_i = 0, lt = 0;
        for (int i = 1; i < nout; ++i) {
            const int v = hull[i];
            lt += hull[i - 1] < v;
            if (lt > 1 && lt <= i - 2) break;
            if (v < hull[min_i]) min_i = i;
            if (v > hull[max_i]) max_i = i;
        }
        const int mm = max_i > min_i ? max_i - min_i : min_i - max_i;
        if ((mm == 1 || mm == nout - 1) && (lt <= 1 || lt >= nout - 2)) {
            const int ascending = (max_i + 1) % nout == min_i;
            const int i0 = ascending ? min_i : max_i;
            if (i0 > 0) {
                int j = i0, i = 0;
                for (; i < nout; ++i) {
                    const int cur = stack[i] = hull[j];
                    const int nj = j + 1 < nout ? j + 1 : 0;
                    if (i < nout - 1 && (ascending != (cur < hull[nj]))) break;
                    j = nj;
                }
                if (i == nout) for (int q = 0; q < nout; ++q) hull[q] = stack[q];
            }
        }
    }
    return nout;
}

// cv::rotatingCalipers(..., CALIPERS_MINAREARECT), float32; hp = hull points (float x, y), work floats: 3*n
__device__ void rotating_calipers(const float* hp, int n, float* vect, float* inv_len, float out[6]) {
    int left = 0, bottom = 0, right = 0, top = 0;
    float p0x = hp[0], p0y = hp[1];
    float left_x = p0x, right_x = p0x, top_y = p0y, bottom_y = p0y;
    for (int i = 0; i < n; ++i) {
        if (p0x < left_x) { left_x = p0x; left = i; }
        if (p0x > right_x) { right_x = p0x; right = i; }
        if (p0y > top_y) { top_y = p0y; top = i; }
        if (p0y < bottom_y) { bottom_y = p0y; bottom = i; }
        const int nx = (i + 1 < n) ? i + 1 : 0;
        const float qx = hp[2 * nx], qy = hp[2 * nx + 1];
        const double dx = (double)qx - (double)p0x, dy = (double)qy - (double)p0y;
        vect[2 * i] = (float)dx; vect[2 * i + 1] = (float)dy;
        inv_len[i] = (float)(1.0 / sqrt(dx * dx + dy * dy));
        p0x = qx; p0y = qy;
    }
    float orientation = 0.f;
    {
        double ax = vect[2 * (n - 1)], ay = vect[2 * (n - 1) + 1];
        for (int i = 0; i < n; ++i) {
            const double bx = vect[2 * i], by = vect[2 * i + 1];
            const double conv = ax * by - ay * bx;
            if (conv != 0) { orientation = conv > 0 ? 1.f : -1.f; break; }
            ax = bx; ay = by;
        }
    }
    float base_a = orientation, base_b = 0.f;
    int seq[4] = {bottom, right, top, left};
    float minarea = 3.402823466e+38f;
    int b_left = 0, b_bottom = 0;
    float b_a = 0.f, b_w = 0.f, b_b = 0.f, b_h = 0.f;
    for (int k = 0; k < n; ++k) {
        float dp[4];
        dp[0] = +base_a * vect[2 * seq[0]] + base_b * vect[2 * seq[0] + 1];
        dp[1] = -base_b * vect[2 * seq[1]] + base_a * vect[2 * seq[1] + 1];
        dp[2] = -base_a * vect[2 * seq[2]] - base_b * vect[2 * seq[2] + 1];
        dp[3] = +base_b * vect[2 * seq[3]] - base_a * vect[2 * seq[3] + 1];
        float maxcos = dp[0] * inv_len[seq[0]];
        int main_e = 0;
        for (int i = 1; i < 4; ++i) {
            const float c = dp[i] * inv_len[seq[i]];
            if (c > maxcos) { main_e = i; maxcos = c; }
        }
        {
            const int pi = seq[main_e];
            const float lead_x = vect[2 * pi] * inv_len[pi], lead_y = vect[2 * pi + 1] * inv_len[pi];
            if (main_e == 0) { base_a = lead_x; base_b = lead_y; }
            else if (main_e == 1) { base_a = lead_y; base_b = -lead_x; }
            else if (main_e == 2) { base_a = -lead_x; base_b = -lead_y; }
            else { base_a = -lead_y; base_b = lead_x; }
        }
        seq[main_e] = (seq[main_e] + 1 == n) ? 0 : seq[main_e] + 1;
        float dx = hp[2 * seq[1]] - hp[2 * seq[3]], dy = hp[2 * seq[1] + 1] - hp[2 * seq[3] + 1];
        const float width = dx * base_a + dy * base_b;
        dx = hp[2 * seq[2]] - hp[2 * seq[0]]; dy = hp[2 * seq[2] + 1] - hp[2 * seq[0] + 1];
        const float height = -dx * base_b + dy * base_a;
        const float area = width * height;
        if (area <= minarea) {
            minarea = area; b_left = seq[3]; b_a = base_a; b_w = width; b_b = base_b; b_h = height; b_bottom = seq[0];
        }
    }
    const float A1 = b_a, B1 = b_b, A2 = -b_b, B2 = b_a;
    const float C1 = A1 * hp[2 * b_left] + hp[2 * b_left + 1] * B1;
    const float C2 = A2 * hp[2 * b_bottom] + hp[2 * b_bottom + 1] * B2;
    const float idet = 1.f / (A1 * B2 - A2 * B1);
    out[0] = (C1 * B2 - C2 * B1) * idet;
    out[1] = (A1 * C2 - A2 * C1) * idet;
    out[2] = A1 * b_w; out[3] = B1 * b_w;
    out[4] = A2 * b_h; out[5] = B2 * b_h;
}

// Least squares through a one-sided Jacobi (Hestenes) SVD, as cv::SVD::compute + cv::SVD::backSubst do:
// A is n x K (row-major, overwritten), x = V diag(1/w) U^T b with singular values <= 2*DBL_EPSILON*sum(w)
// treated as zero.  wmin/wmax return the extreme singular values.
template <int K>
__device__ void svd_lstsq(double* A, int n, const double* bvec, double bconst, double x[K], double* wmax, double* wmin) {
    double V[K][K];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool changed = false;
        for (int p = 0; p < K - 1; ++p)
            for (int q = p + 1; q < K; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = 0; i < n; ++i) { const double u = A[i * K + p], v = A[i * K + q]; a += u * u; b += v * v; g += u * v; }
                if (fabs(g) <= 2.220446049250313e-16 * sqrt(a * b)) continue;
                changed = true;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = 0; i < n; ++i) {
                    const double u = A[i * K + p], v = A[i * K + q];
                    A[i * K + p] = c * u - sn * v; A[i * K + q] = sn * u + c * v;
                }
                for (int i = 0; i < K; ++i) {
                    const double u = V[i][p], v = V[i][q];
                    V[i][p] = c * u - sn * v; V[i][q] = sn * u + c * v;
                }
            }
        if (!changed) break;
    }
    double w[K], utb[K], sum = 0;
    *wmax = 0; *wmin = 1e300;
    for (int j = 0; j < K; ++j) {
        double nn = 0, d = 0;
        for (int i = 0; i < n; ++i) { const double u = A[i * K + j]; nn += u * u; d += u * (bvec ? bvec[i] : bconst); }
        w[j] = sqrt(nn); utb[j] = d; sum += w[j];
        *wmax = fmax(*wmax, w[j]); *wmin = fmin(*wmin, w[j]);
    }
    const double thr = sum * 2.0 * 2.220446049250313e-16;
    for (int i = 0; i < K; ++i) x[i] = 0;
    for (int j = 0; j < K; ++j) {
        if (w[j] <= thr) continue;
        const double f = utb[j] / (w[j] * w[j]);   // (u_j . b) / w_j, with u_j = a_j / w_j
        for (int i = 0; i < K; ++i) x[i] += f * V[i][j];
    }
}

__device__ __forceinline__ double wave_sum(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// svd_lstsq with the n rows of A spread over the 64 lanes of one wave (row i on lane i % 64): the column dot
// products are lane-partial sums + a butterfly, every lane ends up with the same rotation and the same solution.
template <int K>
__device__ void svd_lstsq_wave(double* A, int n, double bconst, double x[K], double* wmax, double* wmin) {
    const int lane = threadIdx.x & 63;
    double V[K][K];
    for (int i = 0; i < K; ++i) for (int j = 0; j < K; ++j) V[i][j] = i == j ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 40; ++sweep) {
        bool changed = false;
        for (int p = 0; p < K - 1; ++p)
            for (int q = p + 1; q < K; ++q) {
                double a = 0, b = 0, g = 0;
                for (int i = lane; i < n; i += 64) { const double u = A[i * K + p], v = A[i * K + q]; a += u * u; b += v * v; g += u * v; }
                a = wave_sum(a); b = wave_sum(b); g = wave_sum(g);
                if (fabs(g) <= 2.220446049250313e-16 * sqrt(a * b)) continue;
                changed = true;
                const double zeta = (b - a) / (2.0 * g);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), sn = c * t;
                for (int i = lane; i < n; i += 64) {
                    const double u = A[i * K + p], v = A[i * K + q];
                    A[i * K + p] = c * u - sn * v; A[i * K + q] = sn * u + c * v;
                }
                for (int i = 0; i < K; ++i) {
                    const double u = V[i][p], v = V[i][q];
                    V[i][p] = c * u - sn * v; V[i][q] = sn * u + c * v;
                }
            }
        if (!changed) break;
    }
    double w[K], utb[K], sum = 0;
    *wmax = 0; *wmin = 1e300;
    for (int j = 0; j < K; ++j) {
        double nn = 0, d = 0;
        for (int i = lane; i < n; i += 64) { const double u = A[i * K + j]; nn += u * u; d += u * bconst; }
        nn = wave_sum(nn); d = wave_sum(d);
        w[j] = sqrt(nn); utb[j] = d; sum += w[j];
        *wmax = fmax(*wmax, w[j]); *wmin = fmin(*wmin, w[j]);
    }
    const double thr = sum * 2.0 * 2.220446049250313e-16;
    for (int i = 0; i < K; ++i) x[i] = 0;
    for (int j = 0; j < K; ++j) {
        if (w[j] <= thr) continue;
        const double f = utb[j] / (w[j] * w[j]);
        for (int i = 0; i < K; ++i) x[i] += f * V[i][j];
    }
}

// cv::fitEllipse (fitEllipseNoDirect) by one wave: returns width <= height (same values on every lane).
// Ad: scratch of 5 * n doubles; sh: 4 doubles of LDS.  The f32 centroid sums keep their sequential order (lane 0).
__device__ void fit_ellipse_wave(const int* pts, int n, double* Ad, double* sh, float* bw, float* bh) {
    const int lane = threadIdx.x & 63;
    if (lane == 0) {
        float csx = 0.f, csy = 0.f;
        for (int i = 0; i < n; ++i) { csx += (float)pts[2 * i]; csy += (float)pts[2 * i + 1]; }
        const float cx0 = csx / (float)n, cy0 = csy / (float)n;
        double s0 = 0;
        for (int i = 0; i < n; ++i) s0 += fabs((double)((float)pts[2 * i] - cx0)) + fabs((double)((float)pts[2 * i + 1] - cy0));
        sh[0] = (double)cx0; sh[1] = (double)cy0; sh[2] = s0;
    }
    __syncthreads();
    const float cx = (float)sh[0], cy = (float)sh[1];
    const double s = sh[2];
    const double eps32 = 1.1920928955078125e-07;
    const double scale = 100.0 / (s > eps32 ? s : eps32);
    double gfp[5], wmax, wmin;
    float eps = 0.f;
    int attempt = 0;
    auto point = [&](int i, double& px, double& py) {
        float fxp = (float)pts[2 * i], fyp = (float)pts[2 * i + 1];
        if (attempt) { fxp = fxp + (float)((i & 1) * 2 - 1) * eps; fyp = fyp + (float)((i & 2) - 1) * eps; }
        px = (double)(fxp - cx) * scale; py = (double)(fyp - cy) * scale;
    };
    for (;; ++attempt) {
        for (int i = lane; i < n; i += 64) {
            double px, py;
            point(i, px, py);
            Ad[i * 5 + 0] = -px * px; Ad[i * 5 + 1] = -py * py; Ad[i * 5 + 2] = -px * py; Ad[i * 5 + 3] = px; Ad[i * 5 + 4] = py;
        }
        __syncthreads();
        svd_lstsq_wave<5>(Ad, n, 10000.0, gfp, &wmax, &wmin);
        __syncthreads();
        if (attempt == 1 || !(wmax * eps32 > wmin)) break;
        eps = (float)(s / (n * 2) * 1e-3);
    }
    double rp[5];
    {
        double m2[4] = {2 * gfp[0], gfp[2], gfp[2], 2 * gfp[1]}, b2[2] = {gfp[3], gfp[4]}, c2[2], a, b;
        svd_lstsq<2>(m2, 2, b2, 0.0, c2, &a, &b);
        rp[0] = c2[0]; rp[1] = c2[1];
    }
    double g[3];
    for (int i = lane; i < n; i += 64) {
        double px, py;
        point(i, px, py);
        Ad[i * 3 + 0] = (px - rp[0]) * (px - rp[0]); Ad[i * 3 + 1] = (py - rp[1]) * (py - rp[1]); Ad[i * 3 + 2] = (px - rp[0]) * (py - rp[1]);
    }
    __syncthreads();
    svd_lstsq_wave<3>(Ad, n, 1.0, g, &wmax, &wmin);
    __syncthreads();
    rp[4] = -0.5 * atan2(g[2], g[1] - g[0]);
    double t;
    if (fabs(g[2]) > 1e-8) t = g[2] / sin(-2.0 * rp[4]);
    else t = g[1] - g[0];
    rp[2] = fabs(g[0] + g[1] - t);
    if (rp[2] > 1e-8) rp[2] = sqrt(2.0 / rp[2]);
    rp[3] = fabs(g[0] + g[1] + t);
    if (rp[3] > 1e-8) rp[3] = sqrt(2.0 / rp[3]);
    float w = (float)(rp[2] * 2 / scale), h = (float)(rp[3] * 2 / scale);
    if (w > h) { const float tt = w; w = h; h = tt; }
    *bw = w; *bh = h;
}

struct MeasureP {
    const int* select;    // [M] or NULL: masks with select[m] == 0 are skipped
    const int* count;     // [M]
    const int* info;      // [M, C, 4]
    const double* red;    // [M, C, 2]
    const int* points;
    int M, C, max_points;
    int* work_i;          // [4 * max_points]
    float* work_f;        // [5 * max_points]
    double* work_d;       // [5 * max_points]
    double um_pix;
    double* out;          // [M, out_c, 12]
    int out_c;
};

constexpr int MEAS_NL = 256;      // contours up to this many points keep all their working arrays in LDS
constexpr int MEAS_CG = 4;        // blocks per mask (contours c, c + 4, ... of the mask)

// One WAVE per contour.  The O(n^2) / O(n * sweeps) parts -- ordering the points for the hull scan, the two
// least-squares solves of the ellipse fit -- use the 64 lanes; the order-dependent f32 arithmetic of the hull scans
// and the rotating calipers stays on lane 0, exactly as OpenCV sequences it.
__global__ __launch_bounds__(64) void contour_measure_kernel(const MeasureP p) {
    // LDS: points | order | union { hull phase: stack, hull, hp, vect, inv_len ; fit phase: A (5 n doubles) }
    __shared__ __attribute__((aligned(16))) char lds[MEAS_NL * 8 + MEAS_NL * 4 + MEAS_NL * 40 + 64];
    __shared__ double sh[4];
    __shared__ float shf[8];
    const int m = blockIdx.x, lane = threadIdx.x;
    if (p.select && !p.select[m]) return;
    const int cnt = min(p.count[m], p.out_c);
    for (int c = blockIdx.y; c < cnt; c += MEAS_CG) {
        const long t = (long)m * p.C + c;
        const int* inf = p.info + t * 4;
        const int n = inf[2], off = inf[3];
        double* o = p.out + ((long)m * p.out_c + c) * 12;
        const double area = p.red[2 * t], perimeter = p.red[2 * t + 1];
        const int* gpts = p.points + 2L * off;
        const bool in_lds = n <= MEAS_NL;
        // scratch: LDS, or carved per contour by its point offset (contour t owns slots [off, off + n + 4) of the
        // point pool, hence disjoint ranges of every pool below)
        int* lpts = reinterpret_cast<int*>(lds);
        int* order = in_lds ? reinterpret_cast<int*>(lds + MEAS_NL * 8) : p.work_i + off;
        char* un = lds + MEAS_NL * 12;
        int* stack = in_lds ? reinterpret_cast<int*>(un) : p.work_i + 1L * p.max_points + 2L * off;          // <= n + 3, 2n + 8 available
        int* hull = in_lds ? reinterpret_cast<int*>(un) + 2 * MEAS_NL + 8 : p.work_i + 3L * p.max_points + off;
        float* hp = in_lds ? reinterpret_cast<float*>(un) + 3 * MEAS_NL + 8 : p.work_f + 2L * off;
        float* vect = in_lds ? hp + 2 * MEAS_NL : p.work_f + 2L * p.max_points + 2L * off;
        float* inv_len = in_lds ? vect + 2 * MEAS_NL : p.work_f + 4L * p.max_points + off;
        double* Ad = in_lds ? reinterpret_cast<double*>(un) : p.work_d + 5L * off;
        __syncthreads();                                         // the previous contour of this block is done with LDS
        if (in_lds) for (int i = lane; i < 2 * n; i += 64) lpts[i] = gpts[i];
        const int* pts = in_lds ? lpts : gpts;
        __syncthreads();
        // ---- order the points by (x, y, index) for the hull scan: rank sort over the lanes ----------------------
        if (n <= 4096) {
            for (int i = lane; i < n; i += 64) {
                const int xi = pts[2 * i], yi = pts[2 * i + 1];
                int r = 0;
                for (int j = 0; j < n; ++j) {
                    const int xj = pts[2 * j], yj = pts[2 * j + 1];
                    r += (xj < xi) || (xj == xi && (yj < yi || (yj == yi && j < i)));
                }
                order[r] = i;
            }
        } else if (lane == 0) heapsort_idx(pts, order, n);
        __syncthreads();
        // ---- minAreaRect -> boxPoints -> int -> order_points -> dA, dB (lane 0) ---------------------------------
        if (lane == 0) {
            float rcx = 0.f, rcy = 0.f, rw = 0.f, rh = 0.f, rang = 0.f;
            int hn = 0;
            if (n > 0) {
                hn = convex_hull_idx(pts, n, order, stack, hull);
                for (int i = 0; i < hn; ++i) { hp[2 * i] = (float)pts[2 * hull[i]]; hp[2 * i + 1] = (float)pts[2 * hull[i] + 1]; }
            }
            if (hn > 2) {
                float q[6];
                rotating_calipers(hp, hn, vect, inv_len, q);
                rcx = q[0] + (q[2] + q[4]) * 0.5f;
                rcy = q[1] + (q[3] + q[5]) * 0.5f;
                rw = (float)sqrt((double)q[2] * q[2] + (double)q[3] * q[3]);
                rh = (float)sqrt((double)q[4] * q[4] + (double)q[5] * q[5]);
                rang = (float)atan2((double)q[3], (double)q[2]);
            } else if (hn == 2) {
                rcx = (hp[0] + hp[2]) * 0.5f; rcy = (hp[1] + hp[3]) * 0.5f;
                const double dx = (double)hp[2] - (double)hp[0], dy = (double)hp[3] - (double)hp[1];
                rw = (float)sqrt(dx * dx + dy * dy); rh = 0.f;
                rang = (float)atan2(dy, dx);
            } else if (hn == 1) { rcx = hp[0]; rcy = hp[1]; }
            shf[0] = rcx; shf[1] = rcy; shf[2] = rw; shf[3] = rh; shf[4] = rang;
        }
        __syncthreads();                                         // hull-phase arrays are dead from here: Ad may reuse them
        double maj = 0, mnr = 0, ecc = 0;
        if (n >= 5) {
            float w, h;
            fit_ellipse_wave(pts, n, Ad, sh, &w, &h);
            maj = (double)w; mnr = (double)h;
            const double ea = (maj > mnr ? maj : mnr) / 2.0, eb = (maj > mnr ? mnr : maj) / 2.0;
            ecc = ea != 0 ? sqrt(1.0 - (eb * eb) / (ea * ea)) : 0.0;
        }
        if (lane != 0) continue;
        const float rcx = shf[0], rcy = shf[1], rw = shf[2], rh = shf[3];
        float rang = shf[4];
        rang = (float)((double)rang * 180.0 / 3.141592653589793238462643383279502884);
        // RotatedRect::points
        const double ar = (double)rang * 3.141592653589793238462643383279502884 / 180.0;
        const float b = (float)cos(ar) * 0.5f, a = (float)sin(ar) * 0.5f;
        float bx[4], by[4];
        bx[0] = rcx - a * rh - b * rw; by[0] = rcy + b * rh - a * rw;
        bx[1] = rcx + a * rh - b * rw; by[1] = rcy - b * rh - a * rw;
        bx[2] = 2.f * rcx - bx[0];     by[2] = 2.f * rcy - by[0];
        bx[3] = 2.f * rcx - bx[1];     by[3] = 2.f * rcy - by[1];
        int ix[4], iy[4];
        for (int i = 0; i < 4; ++i) { ix[i] = (int)bx[i]; iy[i] = (int)by[i]; }  // np.array(box, dtype="int") truncates
        // imutils.perspective.order_points: argsort by x (numpy quicksort on 4 items == insertion sort, stable)
        int ord[4] = {0, 1, 2, 3};
        for (int i = 1; i < 4; ++i) { int j = i; while (j > 0 && ix[ord[j]] < ix[ord[j - 1]]) { const int tt = ord[j]; ord[j] = ord[j - 1]; ord[j - 1] = tt; --j; } }
        int l0 = ord[0], l1 = ord[1], r0 = ord[2], r1 = ord[3];
        if (iy[l1] < iy[l0]) { const int tt = l0; l0 = l1; l1 = tt; }     // stable: swap only when strictly smaller
        const int tl = l0, bl = l1;
        const double d0 = sqrt((double)(ix[r0] - ix[tl]) * (ix[r0] - ix[tl]) + (double)(iy[r0] - iy[tl]) * (iy[r0] - iy[tl]));
        const double d1 = sqrt((double)(ix[r1] - ix[tl]) * (ix[r1] - ix[tl]) + (double)(iy[r1] - iy[tl]) * (iy[r1] - iy[tl]));
        // np.argsort(D)[::-1]: ascending stable then reversed -> on ties the later index comes first
        int br, tr;
        if (d1 >= d0) { br = r1; tr = r0; } else { br = r0; tr = r1; }
        const double tltrx = (ix[tl] + ix[tr]) * 0.5, tltry = (iy[tl] + iy[tr]) * 0.5;
        const double blbrx = (ix[bl] + ix[br]) * 0.5, blbry = (iy[bl] + iy[br]) * 0.5;
        const double tlblx = (ix[tl] + ix[bl]) * 0.5, tlbly = (iy[tl] + iy[bl]) * 0.5;
        const double trbrx = (ix[tr] + ix[br]) * 0.5, trbry = (iy[tr] + iy[br]) * 0.5;
        const double dA = sqrt((tltrx - blbrx) * (tltrx - blbrx) + (tltry - blbry) * (tltry - blbry));
        const double dB = sqrt((tlblx - trbrx) * (tlblx - trbrx) + (tlbly - trbry) * (tlbly - trbry));
        const double um = p.um_pix;
        const double PI = 3.141592653589793;
        const double dmax = fmax(dA, dB), dmin = fmin(dA, dB);
        const double aspect = (dA != 0 && dB != 0) ? dmax / dmin : 0.0;
        o[0] = maj * um;                                   // major_axis_length
        o[1] = mnr * um;                                   // minor_axis_length
        o[2] = ecc;                                        // eccentricity
        o[3] = dmin * um;                                  // Length
        o[4] = dmax * um;                                  // Width
        o[5] = sqrt(4 * area / PI) * um;                   // CircularED
        o[6] = aspect;                                     // Aspect_Ratio
        o[7] = perimeter != 0 ? 4 * PI * (area / (perimeter * perimeter)) * um : 0.0;  // Circularity
        o[8] = perimeter * um;                             // Chords
        o[9] = dmax * um;                                  // Feret_diam
        o[10] = aspect != 0 ? 1.0 / aspect : 0.0;          // Roundness
        o[11] = perimeter != 0 ? (2 * sqrt(PI * area)) / perimeter * um : 0.0;        // Sphericity
    }
}

}  // namespace

extern "C" int64_t demia_contour_work_ints(int M, int C, int max_points) { return 4L * max_points + 8L * M * C + 16; }
extern "C" int64_t demia_contour_work_floats(int M, int C, int max_points) { return 5L * max_points + 16L * M * C + 16; }
extern "C" int64_t demia_contour_work_doubles(int M, int C, int max_points) { return 5L * max_points + 8L * M * C + 16; }

extern "C" int demia_mask_contours_wl(const uint32_t* masks, uint32_t* scratch, const int32_t* bbox, int M, int H, int W, int C,
                                      int max_points, int32_t* count, int32_t* info, double* red, int32_t* points,
                                      int32_t* counters, int32_t* worklist, void* stream) {
    DEMIA_REQUIRE(masks && scratch && bbox && count && info && red && points && counters && W > 0, "args");
    DEMIA_REQUIRE((long)H * W < (1L << 31) && C > 0 && max_points > 0, "sizes");
    if (M == 0) return DEMIA_OK;
    hipStream_t st = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(counters, 0, 4 * sizeof(int32_t), st);
    if (e == hipSuccess && worklist) e = hipMemsetAsync(worklist, 0, 2 * sizeof(int32_t), st);
    if (e != hipSuccess) { demia_set_error("hipMemsetAsync: %s", hipGetErrorString(e)); return DEMIA_ELAUNCH; }
    constexpr int smem = (2 * TRACE_WORDS + CAND_MAX) * 4, smem_small = (2 * TRACE_WORDS_SMALL + CAND_SMALL) * 4;
    auto k_small = contour_trace_kernel<TRACE_WORDS_SMALL, CAND_SMALL>;
    auto k_large = contour_trace_kernel<TRACE_WORDS, CAND_MAX>;
    auto k_list = contour_trace_list_kernel<TRACE_WORDS, CAND_MAX>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_large), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_list), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    ContourP p{masks, scratch, bbox, M, H, W, C, max_points, count, info, red, points, counters};
    hipLaunchKernelGGL(k_small, dim3(M), dim3(256), smem_small, st, p, worklist);
    DEMIA_CHECK_LAUNCH("contour_trace_kernel<small>");
    if (worklist) {
        hipLaunchKernelGGL(k_list, dim3(M < 512 ? M : 512), dim3(256), smem, st, p, worklist);     // two 78-KiB workgroups per CU
        DEMIA_CHECK_LAUNCH("contour_trace_list_kernel<large>");
    } else {
        hipLaunchKernelGGL(k_large, dim3(M), dim3(256), smem, st, p, (int*)nullptr);
        DEMIA_CHECK_LAUNCH("contour_trace_kernel<large>");
    }
    return DEMIA_OK;
}

extern "C" int demia_mask_contours(const uint32_t* masks, uint32_t* scratch, const int32_t* bbox, int M, int H, int W, int C,
                                   int max_points, int32_t* count, int32_t* info, double* red, int32_t* points,
                                   int32_t* counters, void* stream) {
    return demia_mask_contours_wl(masks, scratch, bbox, M, H, W, C, max_points, count, info, red, points, counters, nullptr, stream);
}

extern "C" int demia_contour_measure(const int32_t* select, const int32_t* count, const int32_t* info, const double* red,
                                     const int32_t* points, int M,
                                     int C, int max_points, int32_t* work_i, float* work_f, double* work_d, double um_pix,
                                     double* out, int out_c, void* stream) {
    DEMIA_REQUIRE(count && info && red && points && work_i && work_f && work_d && out && out_c > 0, "args");
    if (M * C == 0) return DEMIA_OK;
    MeasureP p{select, count, info, red, points, M, C, max_points, work_i, work_f, work_d, um_pix, out, out_c};
    hipLaunchKernelGGL(contour_measure_kernel, dim3(M, MEAS_CG), dim3(64), 0, (hipStream_t)stream, p);
    DEMIA_CHECK_LAUNCH("contour_measure_kernel");
    return DEMIA_OK;
}
