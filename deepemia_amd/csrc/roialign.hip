// ROIAlignV2 (aligned=True, sampling_ratio=0 -> adaptive grid) over the FPN pyramid, NHWC.
// HBM/L2 gather-bound: one workgroup per ROI (geometry once), one wavefront per output bin in turn, lanes span
// the channel axis (16 B per lane) so every bilinear tap is one coalesced row read; the adaptive sample grid is
// walked in registers.  -ffp-contract=off: coordinate arithmetic follows torchvision's op order.
//
// Replaces detectron2.modeling.poolers.ROIPooler.forward -> torchvision.ops.roi_align
// (v0.11.1 roi_align_kernel.cpp) incl. assign_boxes_to_levels, as run by
// StandardROIHeads._forward_box / _forward_mask under predictor(image)
// (reference src/functions/inference.py:1395).
#include "common.h"

namespace {

struct RoiP {
    const void* feat[4];
    int H[4], W[4];
    int N, R, C, P;
    const float* boxes;
    const int* count;
    void* out;
    const float* meta[4];   // P32 only: {amax, s} of each level, [groups][2]
    float* out_meta;        // P32 only, [groups][2]
    int groups;             // P32 only: 1, or N (one scale group per image)
    int single;             // P32 only: write a zero low plane (demia_roialign_desc.single)
    const int* order;       // launch order (demia_roi_order) or NULL
};

// Element access policies: bytes per pixel, a lane's byte offset inside a pixel (four channels per lane), load / store
// of those four channels.  P32 (conv_p32.hip): 4 high halves at p, 4 low halves 64 bytes on, value = (h + l) * inv_s; a
// lane's channels 4 l .. 4 l + 3 sit in group l / 8 at (l % 8) * 8 bytes; the buffer starts with a 128-byte zero header.
template <typename T> struct Acc;
template <> struct Acc<float> {
    static constexpr int HEADER = 0;
    static __device__ __forceinline__ long pix_bytes(int C) { return (long)C * 4; }
    static __device__ __forceinline__ int lane_off(int lane) { return lane * 16; }
    static __device__ __forceinline__ void load(const char* p, float, float v[4]) {
        const float4 t = *reinterpret_cast<const float4*>(p);
        v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
    }
    static __device__ __forceinline__ void store(char* p, float, const float v[4]) { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Acc<bf16_t> {
    static constexpr int HEADER = 0;
    static __device__ __forceinline__ long pix_bytes(int C) { return (long)C * 2; }
    static __device__ __forceinline__ int lane_off(int lane) { return lane * 8; }
    static __device__ __forceinline__ void load(const char* p, float, float v[4]) {
        const bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
        v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
    }
    static __device__ __forceinline__ void store(char* p, float, const float v[4]) {
        bf16x4 o;
        o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
        *reinterpret_cast<bf16x4*>(p) = o;
    }
};
struct P32Tag {};
template <> struct Acc<P32Tag> {
    static constexpr int HEADER = 128;
    static __device__ __forceinline__ long pix_bytes(int C) { return (long)C * 4; }
    static __device__ __forceinline__ int lane_off(int lane) { return (lane >> 3) * 128 + (lane & 7) * 8; }
    static __device__ __forceinline__ void load(const char* p, float inv_s, float v[4]) {
        const f16x4 h = *reinterpret_cast<const f16x4*>(p);
        const f16x4 l = *reinterpret_cast<const f16x4*>(p + 64);
#pragma unroll
        for (int c = 0; c < 4; ++c) v[c] = ((float)h[c] + (float)l[c]) * inv_s;
    }
    static __device__ __forceinline__ void store(char* p, float s, const float v[4]) {
        f16x4 h, l;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float y = v[c] * s;
            h[c] = (_Float16)y;
            l[c] = (_Float16)(y - (float)h[c]);
        }
        *reinterpret_cast<f16x4*>(p) = h;
        *reinterpret_cast<f16x4*>(p + 64) = l;
    }
};
// (Round 3 tried FULL 128-byte lines per memory instruction here, as in the conv epilogue: a lane takes 16 bytes = 8 channels
//  of ONE plane, sums that plane, and partner lanes exchange their plane sums by DPP at the end of a bin.  Same box, the boxes
//  of a real 48-tile forward: 7x7 2.43 ms against 2.19 ms for this version, 14x14 0.92 against 0.82 -- 11-13 % SLOWER.  The
//  kernel moves ~37 GB through L1 / L2 per 7x7 call (every bin re-reads the pixels it shares with its neighbours): it is
//  bound by bytes through the texture path, not by the number of requests, and the rewrite added a DPP exchange and a
//  second plane split per lane.  Reverted.)

// One workgroup per ROI: the box -> level / scale / bin geometry is worked out once, then the four waves walk the
// P x P bins (wave w takes bins w, w + 4, ...).  Lanes span the channel axis, four channels per lane, so every
// bilinear tap is one 16-byte load per lane = one coalesced 1 KiB row read per wave (C = 256).  Six waves per SIMD (78
// registers; left alone the compiler took 85 = five waves): 7x7 1716 -> 1598 us per 48-tile forward; eight waves spill.
template <typename T>
__global__ __launch_bounds__(256, 6) void roi_align_kernel(const RoiP p) {
    typedef Acc<T> A;
    constexpr bool P32 = A::HEADER != 0;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    // workgroup b takes ROI order[b] (demia_roi_order: spatial neighbours on the same XCD, close in time); NULL: ROI b
    const long roi = p.order ? (long)p.order[blockIdx.x] : (long)blockIdx.x;  // n*R + r
    const int PP = p.P * p.P;
    const int n = (int)(roi / p.R), r = (int)(roi % p.R);
    const bool lane_on = lane * 4 < p.C;
    const long pixb = A::pix_bytes(p.C);
    float s_out = 1.f;
    if (P32) {
        // bilinear taps and bin averages are convex combinations: the output shares the coarsest scale of the four levels
        // and their largest |x|
        const int gi = p.groups > 1 ? 2 * n : 0;
        s_out = fminf(fminf(p.meta[0][gi + 1], p.meta[1][gi + 1]), fminf(p.meta[2][gi + 1], p.meta[3][gi + 1]));
        if ((p.groups > 1 ? r == 0 : roi == 0) && threadIdx.x == 0) {
            p.out_meta[gi] = fmaxf(fmaxf(p.meta[0][gi], p.meta[1][gi]), fmaxf(p.meta[2][gi], p.meta[3][gi]));
            p.out_meta[gi + 1] = s_out;
        }
    }
    char* out0 = reinterpret_cast<char*>(p.out) + A::HEADER + roi * PP * pixb + A::lane_off(lane);
    if (r >= p.count[n]) {
        const float z[4] = {0.f, 0.f, 0.f, 0.f};
        if (lane_on) for (int bin = wave; bin < PP; bin += 4) A::store(out0 + (long)bin * pixb, s_out, z);
        return;
    }
    const float4 b = reinterpret_cast<const float4*>(p.boxes)[roi];
    // level assignment: floor(4 + log2(sqrt(area) / 224 + 1e-8)) clamped to [2, 5]
    const float area = (b.z - b.x) * (b.w - b.y);
    float lvf = floorf(4.0f + log2f(sqrtf(area) / 224.0f + 1e-8f));
    lvf = fminf(fmaxf(lvf, 2.0f), 5.0f);
    const int lv = (int)lvf - 2;
    const int H = p.H[lv], W = p.W[lv];
    const float scale = 1.0f / (float)(4 << lv);
    const char* feat = reinterpret_cast<const char*>(p.feat[lv]) + A::HEADER + (long)n * H * W * pixb + A::lane_off(lane);
    const float inv_s = P32 ? 1.0f / p.meta[lv][(p.groups > 1 ? 2 * n : 0) + 1] : 1.f;

    const float rsw = b.x * scale - 0.5f, rsh = b.y * scale - 0.5f;
    const float rew = b.z * scale - 0.5f, reh = b.w * scale - 0.5f;
    const float rw = rew - rsw, rh = reh - rsh;
    const float bin_h = rh / (float)p.P, bin_w = rw / (float)p.P;
    const int gh = (int)ceilf(rh / (float)p.P), gw = (int)ceilf(rw / (float)p.P);
    const float cnt = (float)max(gh * gw, 1);

    // Separable form (the common case: at most 8 rows and 8 columns of feature pixels under one bin).  A bin's value is
    // sum_samples sum_taps w f with w = (hy | ly) (hx | lx) and a sample dropped when its y OR its x is out of range, i.e.
    // sum_y sum_x Wy[y] Wx[x] f[y][x] with Wy / Wx the per-row / per-column sums of the samples' weights: every feature
    // pixel under the bin is loaded ONCE instead of once per sample that touches it ((g + 1)^2 instead of 4 g^2 loads for a
    // g x g sample grid -- the kernel was bound by L1 bandwidth).  The tables are built once per ROI by the first threads.
    __shared__ float s_w[2][14 * 8];
    __shared__ int s_lo[2][14], s_n[2][14];
    const bool tables = p.P <= 14 && gh >= 1 && gw >= 1 && gh <= 7 && gw <= 7;        // (block-uniform)
    if (tables) {
        if (threadIdx.x < 2 * p.P) {
            const int dim = threadIdx.x / p.P, b = threadIdx.x - dim * p.P;          // dim 0: rows (y), 1: columns (x)
            const int g = dim == 0 ? gh : gw, size = dim == 0 ? H : W;
            const float start = dim == 0 ? rsh : rsw, bsz = dim == 0 ? bin_h : bin_w;
            float w8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
            int lo0 = 0, hi_last = -1;
            bool any = false;
            for (int pass = 0; pass < 2; ++pass) {                                   // pass 0: the range; pass 1: the weights
                for (int i = 0; i < g; ++i) {
                    float v = start + (float)b * bsz + ((float)i + 0.5f) * bsz / (float)g;
                    if ((v < -1.0f) || (v > (float)size)) continue;
                    if (v <= 0.f) v = 0.f;
                    int lo = (int)v, hi;
                    if (lo >= size - 1) { lo = hi = size - 1; v = (float)lo; } else { hi = lo + 1; }
                    const float l = v - (float)lo, h = 1.0f - l;
                    if (pass == 0) {
                        if (!any) { lo0 = lo; any = true; }
                        hi_last = hi;
                    } else {
#pragma unroll
                        for (int k = 0; k < 8; ++k) w8[k] += (lo - lo0 == k ? h : 0.f) + (hi - lo0 == k ? l : 0.f);
                    }
                }
            }
            const int n = any ? hi_last - lo0 + 1 : 0;
            s_lo[dim][b] = lo0;
            s_n[dim][b] = n;                                                         // n <= g + 1 <= 8
#pragma unroll
            for (int k = 0; k < 8; ++k) s_w[dim][b * 8 + k] = w8[k];
        }
        __syncthreads();
        if (!lane_on) return;
        // A wave walks a PAIR of neighbouring bin columns top to bottom.  Neighbouring bins share feature pixels: down a column
        // the last row of bin ph is the first of bin ph + 1 (bins less than a pixel high share both), and a row's x-combination
        // t = sum_x Wx[x] f[y][x] does not depend on ph -- the two rows combined last are kept, so a column reads g rows per
        // bin instead of g + 1; across the pair the last pixel column of the left bin is the first of the right one (adjacent
        // samples are at most a pixel apart: the union of the two ranges has no gap) -- a row's pixels are loaded once for
        // both, each column taking its own weights (0 outside its range: t + 0 v = t).  Every value sees the same operations in
        // the same order as in the bin-by-bin walk (same bits: checksums of both outputs on the boxes of a 48-tile forward,
        // scripts/gpu_roi_ab.py): 7x7 2190 -> 1825 (row cache) -> 1717 us, 14x14 820 -> 596 -> 531 us.
        for (int pa = 2 * wave; pa < p.P; pa += 8) {
            const bool two = pa + 1 < p.P;
            const int xa0 = s_lo[1][pa], nxa = s_n[1][pa];
            const int xb0 = two ? s_lo[1][pa + 1] : 0, nxb = two ? s_n[1][pa + 1] : 0;
            const int ux0 = nxa > 0 ? (nxb > 0 ? min(xa0, xb0) : xa0) : xb0;
            const int ux1 = max(nxa > 0 ? xa0 + nxa : 0, nxb > 0 ? xb0 + nxb : 0);       // (both empty: no pixel is read)
            const int nu = (nxa > 0 || nxb > 0) ? ux1 - ux0 : 0;
            int ya = -1, yb = -1;
            float ta[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}}, tb[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            for (int ph = 0; ph < p.P; ++ph) {
                const int y0 = s_lo[0][ph], ny = s_n[0][ph];
                float acc[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                for (int yy = 0; yy < ny; ++yy) {
                    const float wy = s_w[0][ph * 8 + yy];
                    const int y = y0 + yy;
                    if (y == ya) {                         // (wave-uniform branches)
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int c = 0; c < 4; ++c) { const float u = ta[k][c]; ta[k][c] = tb[k][c]; tb[k][c] = u; }
                        ya = yb;
                        yb = y;
                    } else if (y != yb) {
                        const char* rowp = feat + ((long)y * W + ux0) * pixb;
                        float t[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                        for (int xx = 0; xx < nu; ++xx) {
                            const int ia = ux0 + xx - xa0, ib = ux0 + xx - xb0;
                            const float wa = (unsigned)ia < (unsigned)nxa ? s_w[1][pa * 8 + ia] : 0.f;
                            const float wb = (unsigned)ib < (unsigned)nxb ? s_w[1][(pa + 1) * 8 + ib] : 0.f;
                            float v[4];
                            A::load(rowp + (long)xx * pixb, inv_s, v);
#pragma unroll
                            for (int c = 0; c < 4; ++c) {
                                t[0][c] += wa * v[c];
                                t[1][c] += wb * v[c];
                            }
                        }
#pragma unroll
                        for (int k = 0; k < 2; ++k)
#pragma unroll
                            for (int c = 0; c < 4; ++c) { ta[k][c] = tb[k][c]; tb[k][c] = t[k][c]; }
                        ya = yb;
                        yb = y;
                    }
#pragma unroll
                    for (int k = 0; k < 2; ++k)
#pragma unroll
                        for (int c = 0; c < 4; ++c) acc[k][c] += wy * tb[k][c];
                }
#pragma unroll
                for (int k = 0; k < 2; ++k) {
                    if (k == 1 && !two) break;
                    float o[4];
#pragma unroll
                    for (int c = 0; c < 4; ++c) o[c] = acc[k][c] / cnt;
                    const long bin = (long)ph * p.P + pa + k;
                    A::store(out0 + bin * pixb, s_out, o);
                    if (P32 && p.single) *reinterpret_cast<uint2*>(out0 + bin * pixb + 64) = make_uint2(0u, 0u);
                }
            }
        }
        return;
    }
    if (!lane_on) return;

    for (int bin = wave; bin < PP; bin += 4) {
        const int ph = bin / p.P, pw = bin - ph * p.P;
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int iy = 0; iy < gh; ++iy) {
            float y = rsh + (float)ph * bin_h + ((float)iy + 0.5f) * bin_h / (float)gh;
            const bool oy = (y < -1.0f) || (y > (float)H);
            if (y <= 0.f) y = 0.f;
            int yl = (int)y, yh;
            if (yl >= H - 1) { yl = yh = H - 1; y = (float)yl; } else { yh = yl + 1; }
            const float ly = y - (float)yl, hy = 1.0f - ly;
            for (int ix = 0; ix < gw; ++ix) {
                float x = rsw + (float)pw * bin_w + ((float)ix + 0.5f) * bin_w / (float)gw;
                const bool ox = (x < -1.0f) || (x > (float)W);
                if (oy || ox) continue;
                if (x <= 0.f) x = 0.f;
                int xl = (int)x, xh;
                if (xl >= W - 1) { xl = xh = W - 1; x = (float)xl; } else { xh = xl + 1; }
                const float lx = x - (float)xl, hx = 1.0f - lx;
                const float w1 = hy * hx, w2 = hy * lx, w3 = ly * hx, w4 = ly * lx;
                float v1[4], v2[4], v3[4], v4[4];
                A::load(feat + ((long)yl * W + xl) * pixb, inv_s, v1);
                A::load(feat + ((long)yl * W + xh) * pixb, inv_s, v2);
                A::load(feat + ((long)yh * W + xl) * pixb, inv_s, v3);
                A::load(feat + ((long)yh * W + xh) * pixb, inv_s, v4);
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float v = w1 * v1[c] + w2 * v2[c] + w3 * v3[c] + w4 * v4[c];
                    acc[c] += v;
                }
            }
        }
        float o[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o[c] = acc[c] / cnt;
        A::store(out0 + (long)bin * pixb, s_out, o);
        if (P32 && p.single) *reinterpret_cast<uint2*>(out0 + (long)bin * pixb + 64) = make_uint2(0u, 0u);
    }
}

// Launch order of an image's ROIs: sorted by (pyramid level, row band, column) of the box centre, so that workgroups that run
// at about the same time read overlapping pixels of one level -- proposals arrive in score order, i.e. scattered over the
// frame and the levels, and every ROI pulls its whole footprint (1-3 MB at its level) through L1 / L2.  Workgroups go to the
// eight XCDs round robin by index, each XCD with an L2 of its own: a sorted run of R / 8 ROIs is dealt to ONE XCD (slot j of
// the image = (s % (R / 8)) * 8 + s / (R / 8) for sorted rank s; plain sorted order when 8 does not divide R).  The OUTPUT row
// of a ROI does not move: results are bit-identical to the unordered launch.  One workgroup per image, bitonic sort in LDS.
__global__ __launch_bounds__(1024) void roi_order_kernel(const float* __restrict__ boxes, const int* __restrict__ count, int R,
                                                         int* __restrict__ order) {
    __shared__ unsigned long long key[1024];
    const int n = blockIdx.x, t = threadIdx.x;
    unsigned long long k = ~0ull;                         // padding beyond R sorts last
    if (t < R) {
        const float4 b = reinterpret_cast<const float4*>(boxes)[(long)n * R + t];
        if (t < count[n]) {
            const float area = (b.z - b.x) * (b.w - b.y);
            float lvf = floorf(4.0f + log2f(sqrtf(fmaxf(area, 0.f)) / 224.0f + 1e-8f));
            lvf = fminf(fmaxf(lvf, 2.0f), 5.0f);
            const int lv = (int)lvf - 2;
            const float scale = 1.0f / (float)(4 << lv);
            const int cy = min(max((int)((b.y + b.w) * 0.5f * scale), 0), 4095);
            const int cx = min(max((int)((b.x + b.z) * 0.5f * scale), 0), 4095);
            k = ((unsigned long long)lv << 40) | ((unsigned long long)(cy >> 3) << 28) | ((unsigned long long)cx << 12) | (unsigned)t;
        } else {
            k = (1ull << 62) | (unsigned)t;               // unused slots of the table: after every real ROI, in index order
        }
    }
    key[t] = k;
    __syncthreads();
    for (int size = 2; size <= 1024; size <<= 1) {
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int partner = t ^ stride;
            if (partner > t) {
                const bool up = (t & size) == 0;
                const unsigned long long a = key[t], c = key[partner];
                if ((a > c) == up) { key[t] = c; key[partner] = a; }
            }
            __syncthreads();
        }
    }
    if (t < R) {
        const int s = t, per = R >> 3;
        const int slot = (R & 7) == 0 ? (s % per) * 8 + s / per : s;
        order[(long)n * R + slot] = n * R + (int)(key[s] & 0xfffu);
    }
}

}  // namespace

extern "C" int demia_roi_order(const float* boxes, const int32_t* count, int N, int R, int32_t* order, void* stream) {
    DEMIA_REQUIRE(boxes && count && order && N >= 0, "null pointer");
    DEMIA_REQUIRE(R > 0 && R <= 1024, "R must be in 1..1024");
    if (N == 0) return DEMIA_OK;
    hipLaunchKernelGGL(roi_order_kernel, dim3(N), dim3(1024), 0, (hipStream_t)stream, boxes, count, R, order);
    DEMIA_CHECK_LAUNCH("roi_order_kernel");
    return DEMIA_OK;
}

extern "C" int demia_roi_align(const demia_roialign_desc* d, void* stream) {
    DEMIA_REQUIRE(d && d->boxes && d->count && d->out, "null pointer");
    DEMIA_REQUIRE(d->C % 4 == 0 && d->C <= 256, "C must be a multiple of 4, at most 256");
    DEMIA_REQUIRE(d->P > 0, "P");
    RoiP p;
    for (int l = 0; l < 4; ++l) {
        DEMIA_REQUIRE(d->feat[l], "feat pointer");
        p.feat[l] = d->feat[l]; p.H[l] = d->H[l]; p.W[l] = d->W[l];
    }
    p.N = d->N; p.R = d->R; p.C = d->C; p.P = d->P; p.boxes = d->boxes; p.count = d->count; p.out = d->out;
    for (int l = 0; l < 4; ++l) p.meta[l] = d->meta[l];
    p.out_meta = d->out_meta;
    p.groups = d->groups > 1 ? d->groups : 1;
    p.single = d->single;
    p.order = d->order;
    DEMIA_REQUIRE(p.groups == 1 || p.groups == d->N, "scale groups: one per image");
    if (d->dtype == DEMIA_P32) {
        DEMIA_REQUIRE(d->C % 32 == 0 && d->out_meta && d->meta[0] && d->meta[1] && d->meta[2] && d->meta[3], "P32 needs C % 32 == 0 and the meta pointers");
    }
    const long total = (long)d->N * d->R;
    if (total == 0 || d->P == 0) return DEMIA_OK;
    const int grid = (int)total;
    if (d->dtype == DEMIA_P32)
        hipLaunchKernelGGL(roi_align_kernel<P32Tag>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else if (d->dtype == DEMIA_BF16)
        hipLaunchKernelGGL(roi_align_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(roi_align_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    DEMIA_CHECK_LAUNCH("roi_align_kernel");
    return DEMIA_OK;
}
