// Packed-mask morphology for the post-processing stages the reference runs on dense numpy
// masks on the host (HBM-bound bit work: 32 pixels per u32 word, no GEMM reshaping):
//
//   fill holes     scipy.ndimage.binary_fill_holes          mask_utils.py:75, inference.py:193,1780
//   erode / dilate skimage erosion / dilation, 3x3 cross,    mask_utils.py:76, inference.py:196-198,
//                  'reflect' border (the frame never erodes)  1786-1796
//   overlap prefix `overlap += mask; mask[overlap > 1] = 0`   mask_utils.py:77-78
//   components > 1 skimage.measure.label(mask).max() > 1     mask_utils.py:79-81 (8-connected)
//   column counts  np.sum(masks, axis=(0, 1))                 mask_utils.py:62
//   pair counts    np.count_nonzero(m1 & m2)                  inference.py:431, 2710; spatial_constraints.py:143,186
//   place tile     cv2.resize(INTER_NEAREST) + paste at (x, y) offset, edge test   inference.py:2399-2420, 2522-2549
//
// Layout: [M, H, W/32] u32, bit (x & 31) of word (x >> 5).  Flood fills work on the mask's
// bounding box only (holes and components cannot leave it) with a Kogge-Stone occluded fill
// inside each word, so one iteration moves a front a whole word horizontally and one row
// vertically; one workgroup per mask iterates to a fixed point.
#include "common.h"

namespace {

__device__ __forceinline__ uint32_t fill_up(uint32_t g, uint32_t p) {  // towards higher bits, through p
    g |= p & (g << 1); p &= p << 1;
    g |= p & (g << 2); p &= p << 2;
    g |= p & (g << 4); p &= p << 4;
    g |= p & (g << 8); p &= p << 8;
    g |= p & (g << 16);
    return g;
}
__device__ __forceinline__ uint32_t fill_down(uint32_t g, uint32_t p) {
    g |= p & (g >> 1); p &= p >> 1;
    g |= p & (g >> 2); p &= p >> 2;
    g |= p & (g >> 4); p &= p >> 4;
    g |= p & (g >> 8); p &= p >> 8;
    g |= p & (g >> 16);
    return g;
}

// bits of word wx (pixels wx*32 .. wx*32+31) that lie in [x0, x1]
__device__ __forceinline__ uint32_t span_mask(int wx, int x0, int x1) {
    const int lo = max(x0 - wx * 32, 0), hi = min(x1 - wx * 32, 31);
    if (lo > hi) return 0u;
    const uint32_t upto_hi = hi == 31 ? 0xFFFFFFFFu : ((1u << (hi + 1)) - 1u);
    return upto_hi & ~((1u << lo) - 1u);
}

// ---- fill holes: block per mask --------------------------------------------------------------
// R = background reachable from outside (4-connected).  The bbox region (+1 ring) of the mask and of R is
// staged in LDS when it fits (2 x 8192 words = 64 KiB: e.g. 256 rows x 1024 px), so the fixed-point iteration
// runs at LDS latency; larger regions iterate in place in HBM/L2.
constexpr int FILL_LDS_WORDS = 8192;

__device__ void flood_bg_iterate(const uint32_t* __restrict__ M, uint32_t* R, int stride, int rh, int rw, int max_iter, int* changed) {
    const int tid = threadIdx.x, nt = blockDim.x;
    for (int iter = 0; iter < max_iter; ++iter) {
        if (tid == 0) *changed = 0;
        __syncthreads();
        bool ch = false;
        for (int i = tid; i < rh * rw; i += nt) {
            const int ly = i / rw, lx = i - ly * rw;
            const int o = ly * stride + lx;
            const uint32_t bg = ~M[o];
            const uint32_t r = R[o];
            uint32_t n = r;
            if (ly > 0) n |= R[o - stride];
            if (ly < rh - 1) n |= R[o + stride];
            uint32_t lr = (r << 1) | (r >> 1);
            if (lx > 0) lr |= R[o - 1] >> 31;
            if (lx < rw - 1) lr |= R[o + 1] << 31;
            uint32_t c = (n | lr) & bg;
            c = fill_up(c, bg);
            c = fill_down(c, bg);
            if (c != r) { R[o] = c; ch = true; }
        }
        if (ch) *changed = 1;
        __syncthreads();
        if (!*changed) break;
        __syncthreads();
    }
}

__global__ __launch_bounds__(1024) void fill_holes_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out,
                                                          const int* __restrict__ bbox, int H, int W) {
    __shared__ int changed;
    __shared__ uint32_t lds[2 * FILL_LDS_WORDS];
    const long m = blockIdx.x;
    const int wpr = (W + 31) >> 5;
    const uint32_t* src = in + m * (long)H * wpr;
    uint32_t* dst = out + m * (long)H * wpr;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (y0 < 0) {  // empty mask
        for (int i = tid; i < H * wpr; i += nt) dst[i] = 0u;
        return;
    }
    const int ry0 = max(y0 - 1, 0), ry1 = min(y1 + 1, H - 1);
    const int wx0 = max((x0 - 1) >> 5, 0), wx1 = min((x1 + 1) >> 5, wpr - 1);
    const int rw = wx1 - wx0 + 1, rh = ry1 - ry0 + 1;
    const bool use_lds = rh * rw <= FILL_LDS_WORDS;
    // everything outside the bbox rows/words is copied (no holes there)
    for (int i = tid; i < H * wpr; i += nt) {
        const int y = i / wpr, wx = i - y * wpr;
        if (y < ry0 || y > ry1 || wx < wx0 || wx > wx1) dst[i] = src[i];
    }
    uint32_t* Mreg = use_lds ? lds : nullptr;
    uint32_t* Rreg = use_lds ? lds + FILL_LDS_WORDS : dst + (long)ry0 * wpr + wx0;
    const int stride = use_lds ? rw : wpr;
    // seeds: background outside the tight bbox, or on the image frame
    for (int i = tid; i < rh * rw; i += nt) {
        const int ly = i / rw, lx = i - ly * rw;
        const int y = ry0 + ly, wx = wx0 + lx;
        const uint32_t mk = src[(long)y * wpr + wx];
        uint32_t seed;
        if (y < y0 || y > y1 || y == 0 || y == H - 1) seed = 0xFFFFFFFFu;
        else {
            seed = ~span_mask(wx, x0, x1);
            if (wx == 0) seed |= 1u;
            if (wx == wpr - 1) seed |= 0xFFFFFFFFu << ((W - 1) & 31);   // pixel W-1 and the padding bits beyond it
        }
        if (use_lds) Mreg[ly * stride + lx] = mk;
        Rreg[ly * stride + lx] = ~mk & seed;
    }
    __syncthreads();
    const uint32_t* Mptr = use_lds ? Mreg : src + (long)ry0 * wpr + wx0;
    flood_bg_iterate(Mptr, Rreg, stride, rh, rw, 2 * (H + W), &changed);
    for (int i = tid; i < rh * rw; i += nt) {
        const int ly = i / rw, lx = i - ly * rw;
        const uint32_t valid = (wx0 + lx == wpr - 1 && (W & 31)) ? ((1u << (W & 31)) - 1u) : 0xFFFFFFFFu;
        dst[(long)(ry0 + ly) * wpr + wx0 + lx] = ~Rreg[ly * stride + lx] & valid;
    }
}

// ---- cross erosion / dilation, border = replicate ------------------------------------------------
template <bool DILATE>
__global__ void morph_cross_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ out, long nwords, int H, int W) {
    const int wpr = (W + 31) >> 5;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nwords; i += (long)gridDim.x * blockDim.x) {
        const int wx = (int)(i % wpr);
        const int y = (int)((i / wpr) % H);
        const uint32_t c = in[i];
        const uint32_t up = y > 0 ? in[i - wpr] : c;
        const uint32_t dn = y < H - 1 ? in[i + wpr] : c;
        const int last = (W - 1) & 31;                                           // bit of pixel W-1 in the last word
        const uint32_t lbit = wx > 0 ? (in[i - 1] >> 31) : (c & 1u);            // pixel x-1 of bit 0
        const uint32_t left = (c << 1) | lbit;     // bit b = pixel b-1
        uint32_t right, valid = 0xFFFFFFFFu;
        if (wx < wpr - 1) right = (c >> 1) | ((in[i + 1] & 1u) << 31);
        else { right = (c >> 1) | (((c >> last) & 1u) << last); valid = last == 31 ? 0xFFFFFFFFu : ((2u << last) - 1u); }
        out[i] = (DILATE ? (c | up | dn | left | right) : (c & up & dn & left & right)) & valid;
    }
}

// ---- sequential overlap removal across the masks of one call (score order) -----------------------
// seg (optional): segment id per mask, non-decreasing; the accumulator restarts at every segment
// boundary, so the masks of many (tile, class) calls are processed by ONE launch.
__global__ void overlap_prefix_kernel(uint32_t* __restrict__ masks, const int* __restrict__ seg, int M, long words_per_mask) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < words_per_mask; i += (long)gridDim.x * blockDim.x) {
        uint32_t seen = 0u;
        int cur = seg ? seg[0] : 0;
        for (int m = 0; m < M; ++m) {
            if (seg && seg[m] != cur) { cur = seg[m]; seen = 0u; }
            const uint32_t c = masks[m * words_per_mask + i];
            masks[m * words_per_mask + i] = c & ~seen;
            seen |= c;
        }
    }
}

// ---- more than one 8-connected component?  block per mask, `scratch` same shape as the masks -----
// Flood the component of the first set pixel (8-connected) and compare with the mask; the bbox region
// is staged in LDS when it fits, `scratch` is only touched by the out-of-LDS fallback.
__global__ __launch_bounds__(1024) void components_gt1_kernel(const uint32_t* __restrict__ in, uint32_t* __restrict__ scratch,
                                                              const int* __restrict__ bbox, int* __restrict__ flag,
                                                              int H, int W) {
    __shared__ int changed;
    __shared__ int first;
    __shared__ uint32_t lds[2 * FILL_LDS_WORDS];
    const long m = blockIdx.x;
    const int wpr = (W + 31) >> 5;
    const uint32_t* src = in + m * (long)H * wpr;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (y0 < 0) { if (tid == 0) flag[m] = 0; return; }
    const int wx0 = x0 >> 5, wx1 = x1 >> 5;
    const int rw = wx1 - wx0 + 1, rh = y1 - y0 + 1;
    const bool use_lds = rh * rw <= FILL_LDS_WORDS;
    const int stride = use_lds ? rw : wpr;
    uint32_t* R = use_lds ? lds + FILL_LDS_WORDS : scratch + m * (long)H * wpr + (long)y0 * wpr + wx0;
    const uint32_t* Mk = use_lds ? lds : src + (long)y0 * wpr + wx0;
    if (tid == 0) first = 0x7FFFFFFF;
    __syncthreads();
    for (int i = tid; i < rw; i += nt)
        if (src[(long)y0 * wpr + wx0 + i]) atomicMin(&first, i);
    __syncthreads();
    for (int i = tid; i < rh * rw; i += nt) {
        const int ly = i / rw, lx = i - ly * rw;
        const uint32_t mk = src[(long)(y0 + ly) * wpr + wx0 + lx];
        if (use_lds) lds[ly * stride + lx] = mk;
        R[ly * stride + lx] = (ly == 0 && lx == first) ? (mk & (0u - mk)) : 0u;
    }
    __syncthreads();
    for (int iter = 0; iter < 2 * (H + W); ++iter) {
        if (tid == 0) changed = 0;
        __syncthreads();
        bool ch = false;
        for (int i = tid; i < rh * rw; i += nt) {
            const int ly = i / rw, lx = i - ly * rw;
            const int o = ly * stride + lx;
            const uint32_t fg = Mk[o];
            const uint32_t r = R[o];
            uint32_t v = r, lcar = 0u, rcar = 0u;
            if (lx > 0) lcar = R[o - 1];
            if (lx < rw - 1) rcar = R[o + 1];
            if (ly > 0) { v |= R[o - stride]; if (lx > 0) lcar |= R[o - stride - 1]; if (lx < rw - 1) rcar |= R[o - stride + 1]; }
            if (ly < rh - 1) { v |= R[o + stride]; if (lx > 0) lcar |= R[o + stride - 1]; if (lx < rw - 1) rcar |= R[o + stride + 1]; }
            uint32_t c = v | (v << 1) | (v >> 1) | (lcar >> 31) | (rcar << 31);
            c &= fg;
            c = fill_up(c, fg);
            c = fill_down(c, fg);
            if (c != r) { R[o] = c; ch = true; }
        }
        if (ch) changed = 1;
        __syncthreads();
        if (!changed) break;
        __syncthreads();
    }
    if (tid == 0) changed = 0;
    __syncthreads();
    bool diff = false;
    for (int i = tid; i < rh * rw; i += nt) {
        const int ly = i / rw, lx = i - ly * rw;
        if (R[ly * stride + lx] != Mk[ly * stride + lx]) diff = true;
    }
    if (diff) changed = 1;
    __syncthreads();
    if (tid == 0) flag[m] = changed;
}

// ---- per-column pixel counts over all masks of a call (counts must be zeroed by the caller) ------
// seg (optional): segment id per mask; counts is [S, W], one row per segment.  One block per mask walks
// only the mask's bbox rows (bbox from demia_mask_area_bbox).
__global__ void column_counts_kernel(const uint32_t* __restrict__ masks, const int* __restrict__ seg, const int* __restrict__ bbox,
                                     int H, int W, int* __restrict__ counts) {
    const int m = blockIdx.y;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    if (y0 < 0) return;
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    if (x >= W || x < x0 || x > x1) return;
    const int wpr = (W + 31) >> 5, wx = x >> 5, b = x & 31;
    const uint32_t* src = masks + (long)m * H * wpr;
    int c = 0;
    for (int y = y0; y <= y1; ++y) c += (src[(long)y * wpr + wx] >> b) & 1u;
    if (c) atomicAdd(&counts[(long)(seg ? seg[m] : 0) * W + x], c);
}

// ---- |a & b| for a list of pairs: block per pair over the bbox intersection ----------------------
__global__ __launch_bounds__(256) void pair_intersections_kernel(const uint32_t* __restrict__ a, const uint32_t* __restrict__ b,
                                                                 const int* __restrict__ pi, const int* __restrict__ pj,
                                                                 const int* __restrict__ bbox_a, const int* __restrict__ bbox_b,
                                                                 int* __restrict__ out, int H, int W) {
    __shared__ int acc;
    const int p = blockIdx.x;
    const int i = pi[p], j = pj[p];
    const int wpr = (W + 31) >> 5;
    const int y0 = max(bbox_a[i * 4 + 0], bbox_b[j * 4 + 0]), y1 = min(bbox_a[i * 4 + 2], bbox_b[j * 4 + 2]);
    const int x0 = max(bbox_a[i * 4 + 1], bbox_b[j * 4 + 1]), x1 = min(bbox_a[i * 4 + 3], bbox_b[j * 4 + 3]);
    if (threadIdx.x == 0) acc = 0;
    __syncthreads();
    if (bbox_a[i * 4] >= 0 && bbox_b[j * 4] >= 0 && y0 <= y1 && x0 <= x1) {
        const int wx0 = x0 >> 5, rw = (x1 >> 5) - wx0 + 1, rh = y1 - y0 + 1;
        const uint32_t* ma = a + (long)i * H * wpr;
        const uint32_t* mb = b + (long)j * H * wpr;
        int c = 0;
        for (int t = threadIdx.x; t < rh * rw; t += blockDim.x) {
            const long o = (long)(y0 + t / rw) * wpr + wx0 + t % rw;
            c += __popc(ma[o] & mb[o]);
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if ((threadIdx.x & 63) == 0 && c) atomicAdd(&acc, c);
    }
    __syncthreads();
    if (threadIdx.x == 0) out[p] = acc;
}

// ---- tile mask -> global frame (nearest resize + offset paste), with the edge-band test ----------
// src [T, th, tw/32] (tile masks at network scale), dst [T, H, W/32]; tile t goes to (x_off[t], y_off[t]);
// the mask is first resized to (tile_h, tile_w) with cv2's INTER_NEAREST rule.
__global__ void place_tile_kernel(const uint32_t* __restrict__ src, uint32_t* __restrict__ dst, const int* __restrict__ x_off,
                                  const int* __restrict__ y_off, int T, int sh, int sw, int tile_h, int tile_w, int H, int W) {
    const int wpr = (W + 31) >> 5, swpr = (sw + 31) >> 5;
    const long total = (long)T * H * wpr;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int wx = (int)(i % wpr);
        const int y = (int)((i / wpr) % H);
        const int t = (int)(i / ((long)wpr * H));
        const int ty = y - y_off[t];
        uint32_t bits = 0u;
        if (ty >= 0 && ty < tile_h) {
            const int sy = min((int)floor((double)ty * (1.0 / ((double)tile_h / (double)sh))), sh - 1);
            const uint32_t* srow = src + ((long)t * sh + sy) * swpr;
            for (int bb = 0; bb < 32; ++bb) {
                if (wx * 32 + bb >= W) break;                      // padding bits of the last word stay 0
                const int tx = wx * 32 + bb - x_off[t];
                if (tx < 0 || tx >= tile_w) continue;
                const int sx = min((int)floor((double)tx * (1.0 / ((double)tile_w / (double)sw))), sw - 1);
                bits |= ((srow[sx >> 5] >> (sx & 31)) & 1u) << bb;
            }
        }
        dst[i] = bits;
    }
}

inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int demia_mask_fill_holes(const uint32_t* in, uint32_t* out, const int32_t* bbox, int64_t M, int H, int W,
                                     void* stream) {
    DEMIA_REQUIRE(in && out && bbox && in != out && W > 0, "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(fill_holes_kernel, dim3((int)M), dim3(1024), 0, (hipStream_t)stream, in, out, bbox, H, W);
    DEMIA_CHECK_LAUNCH("fill_holes_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_morph_cross(const uint32_t* in, uint32_t* out, int64_t M, int H, int W, int dilate, void* stream) {
    DEMIA_REQUIRE(in && out && in != out && W > 0, "args");
    const long n = (long)M * H * ((W + 31) / 32);
    if (n == 0) return DEMIA_OK;
    if (dilate)
        hipLaunchKernelGGL(morph_cross_kernel<true>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, H, W);
    else
        hipLaunchKernelGGL(morph_cross_kernel<false>, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, in, out, n, H, W);
    DEMIA_CHECK_LAUNCH("morph_cross_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_overlap_prefix(uint32_t* masks, const int32_t* seg, int64_t M, int H, int W, void* stream) {
    DEMIA_REQUIRE(masks && W > 0, "args");
    const long wpm = (long)H * ((W + 31) / 32);
    if (M == 0 || wpm == 0) return DEMIA_OK;
    hipLaunchKernelGGL(overlap_prefix_kernel, dim3(grid_for(wpm, 256)), dim3(256), 0, (hipStream_t)stream, masks, seg, (int)M, wpm);
    DEMIA_CHECK_LAUNCH("overlap_prefix_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_components_gt1(const uint32_t* in, uint32_t* scratch, const int32_t* bbox, int32_t* flag,
                                         int64_t M, int H, int W, void* stream) {
    DEMIA_REQUIRE(in && scratch && bbox && flag && in != scratch && W > 0, "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(components_gt1_kernel, dim3((int)M), dim3(1024), 0, (hipStream_t)stream, in, scratch, bbox, flag, H, W);
    DEMIA_CHECK_LAUNCH("components_gt1_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_column_counts(const uint32_t* masks, const int32_t* seg, const int32_t* bbox, int64_t M, int H, int W,
                                        int32_t* counts, void* stream) {
    DEMIA_REQUIRE(masks && counts && W > 0, "args");
    if (W == 0) return DEMIA_OK;
    DEMIA_REQUIRE(bbox, "bbox");
    if (M == 0) return DEMIA_OK;
    DEMIA_REQUIRE(M <= 65535, "M <= 65535");
    hipLaunchKernelGGL(column_counts_kernel, dim3(cdiv(W, 256), (int)M), dim3(256), 0, (hipStream_t)stream, masks, seg, bbox, H, W,
                       counts);
    DEMIA_CHECK_LAUNCH("column_counts_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_pair_intersections(const uint32_t* a, const uint32_t* b, const int32_t* pi, const int32_t* pj,
                                             const int32_t* bbox_a, const int32_t* bbox_b, int32_t* out, int64_t P,
                                             int H, int W, void* stream) {
    DEMIA_REQUIRE(a && b && pi && pj && bbox_a && bbox_b && out && W > 0, "args");
    if (P == 0) return DEMIA_OK;
    hipLaunchKernelGGL(pair_intersections_kernel, dim3((int)P), dim3(256), 0, (hipStream_t)stream, a, b, pi, pj, bbox_a, bbox_b,
                       out, H, W);
    DEMIA_CHECK_LAUNCH("pair_intersections_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_place_tiles(const uint32_t* src, uint32_t* dst, const int32_t* x_off, const int32_t* y_off, int64_t T,
                                      int src_h, int src_w, int tile_h, int tile_w, int H, int W, void* stream) {
    DEMIA_REQUIRE(src && dst && x_off && y_off && W > 0 && src_w > 0, "args");
    const long n = (long)T * H * ((W + 31) / 32);
    if (n == 0) return DEMIA_OK;
    hipLaunchKernelGGL(place_tile_kernel, dim3(grid_for(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, x_off, y_off,
                       (int)T, src_h, src_w, tile_h, tile_w, H, W);
    DEMIA_CHECK_LAUNCH("place_tile_kernel");
    return DEMIA_OK;
}
