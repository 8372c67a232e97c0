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
// Layout: [M, H, ceil(W/32)] u32, bit (x & 31) of word (x >> 5).  Fill / erode / dilate / component test run as
// stage PROGRAMS on the mask's bounding-box region, in place, one workgroup per mask (maskregion.h): the frame
// outside the box is never touched.  Every kernel that takes a bbox accepts a superset of the tight box.
#include "common.h"
#include "maskregion.h"

namespace {

// LDS words per region buffer.  Two launches per program: most masks are a few hundred words and run in the SMALL
// variant, whose 8 KiB of LDS fit beside the two resident workgroups of a convolution on the other stream (the next
// batch's network runs concurrently); the few large masks take the 64 KiB variant, regions beyond that go through HBM.
constexpr int REG_WORDS_SMALL = 1024;
constexpr int REG_WORDS = 8192;

// ---- per-mask stage programs on the bbox region, in place ----------------------------------------------------
// program = up to 8 stage codes, 4 bits each, executed low nibble first (DEMIA_MOP_*).  The region (bbox grown by
// one pixel per dilation + 1) is staged in LDS; the result is written back into the mask and its area / tight
// bbox are reduced on the way out, so a chain such as fill -> dilate -> erode costs one launch and touches
// ~1 KiB per mask instead of streaming the 512 KiB frame three times.
struct ProgP {
    uint32_t* masks;
    uint32_t* scratch;
    const int* bbox;
    const uint8_t* active;
    uint32_t program;
    int H, W;
    int* area;
    int* bbox_out;
    int* flag;
};

// One mask through the program; every thread of the block calls it (block-uniform control flow).  `lo` / RW as in the
// kernels below.  Returns true when mask m belongs to a LARGER variant than this one (n > RW and RW is not the largest).
template <int RW>
__device__ __forceinline__ bool run_mask_program(const ProgP& p, const long m, const int lo, uint32_t* lds, int* s_changed, int* s_first, int* s_area,
                                                 int* s_y0, int* s_y1, int* s_x0, int* s_x1) {
    const int tid = threadIdx.x, nt = blockDim.x;
    const int wpr = (p.W + 31) >> 5;
    int cy0 = p.bbox[m * 4 + 0], cx0 = p.bbox[m * 4 + 1], cy1 = p.bbox[m * 4 + 2], cx1 = p.bbox[m * 4 + 3];
    if (cy0 < 0) {                      // empty mask stays empty under every stage
        if (tid == 0 && lo == 0) {
            if (p.area) p.area[m] = 0;
            if (p.bbox_out) { p.bbox_out[m * 4 + 0] = -1; p.bbox_out[m * 4 + 1] = -1; p.bbox_out[m * 4 + 2] = -1; p.bbox_out[m * 4 + 3] = -1; }
            if (p.flag) p.flag[m] = 0;
        }
        return false;
    }
    int ndil = 0;
    for (uint32_t q = p.program; q; q >>= 4) ndil += (q & 15u) == DEMIA_MOP_DILATE;
    mreg::Reg g;
    g.H = p.H; g.W = p.W; g.wpr = wpr;
    mreg::region_of(cy0, cx0, cy1, cx1, ndil + 1, p.H, p.W, g.ry0, g.wx0, g.rh, g.rw);
    const int n = g.rh * g.rw;
    if (n <= lo) return false;                                // a smaller variant's mask
    if (n > RW && RW != REG_WORDS) return true;               // a larger variant's mask
    const bool use_lds = n <= RW;
    uint32_t* home = p.masks + m * (long)p.H * wpr + (long)g.ry0 * wpr + g.wx0;
    if (use_lds) {
        g.A = lds; g.B = lds + RW; g.stride = g.rw;
        for (int i = tid; i < n; i += nt) g.A[i] = home[(long)(i / g.rw) * wpr + i % g.rw];
    } else {
        g.A = home; g.B = p.scratch + m * (long)p.H * wpr + (long)g.ry0 * wpr + g.wx0; g.stride = wpr;
    }
    __syncthreads();
    int flagged = 0;
    for (uint32_t q = p.program; q; q >>= 4) {
        const uint32_t op = q & 15u;
        if (op == DEMIA_MOP_FILL) {
            mreg::fill_holes(g, cy0, cx0, cy1, cx1, s_changed);
        } else if (op == DEMIA_MOP_DILATE || op == DEMIA_MOP_ERODE) {
            if (op == DEMIA_MOP_DILATE) {
                mreg::morph_cross<true>(g);
                cy0 = max(cy0 - 1, 0); cx0 = max(cx0 - 1, 0); cy1 = min(cy1 + 1, p.H - 1); cx1 = min(cx1 + 1, p.W - 1);
            } else {
                mreg::morph_cross<false>(g);
            }
            uint32_t* t = g.A; g.A = g.B; g.B = t;
        } else if (op == DEMIA_MOP_DROP_MULTI || op == DEMIA_MOP_FLAG_MULTI) {
            const bool multi = mreg::more_than_one_component(g, cy0, cx0, cy1, cx1, s_first, s_changed);
            flagged |= multi;
            if (multi && op == DEMIA_MOP_DROP_MULTI) {
                for (int i = tid; i < n; i += nt) g.A[(i / g.rw) * g.stride + i % g.rw] = 0u;
                __syncthreads();
            }
        } else if (op == DEMIA_MOP_GATE) {
            if (!(p.active && p.active[m])) break;
        }
    }
    // ---- write back + area / tight bbox -------------------------------------------------------------------
    if (tid == 0) { *s_area = 0; *s_y0 = 1 << 30; *s_x0 = 1 << 30; *s_y1 = -1; *s_x1 = -1; }
    __syncthreads();
    int a = 0, y0 = 1 << 30, y1 = -1, x0 = 1 << 30, x1 = -1;
    const bool copy = g.A != home;
    for (int i = tid; i < n; i += nt) {
        const int ly = i / g.rw, lx = i - ly * g.rw;
        const uint32_t b = g.A[ly * g.stride + lx];
        if (copy) home[(long)ly * wpr + lx] = b;
        if (b) {
            const int y = g.ry0 + ly, xb = (g.wx0 + lx) << 5;
            a += __popc(b);
            y0 = min(y0, y); y1 = max(y1, y);
            x0 = min(x0, xb + __ffs((int)b) - 1);
            x1 = max(x1, xb + 31 - __clz((int)b));
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_down(a, o, 64);
        y0 = min(y0, __shfl_down(y0, o, 64)); x0 = min(x0, __shfl_down(x0, o, 64));
        y1 = max(y1, __shfl_down(y1, o, 64)); x1 = max(x1, __shfl_down(x1, o, 64));
    }
    if ((tid & 63) == 0 && a) {
        atomicAdd(s_area, a);
        atomicMin(s_y0, y0); atomicMin(s_x0, x0);
        atomicMax(s_y1, y1); atomicMax(s_x1, x1);
    }
    __syncthreads();
    if (tid == 0) {
        const bool e = *s_area == 0;
        if (p.area) p.area[m] = *s_area;
        if (p.bbox_out) {
            p.bbox_out[m * 4 + 0] = e ? -1 : *s_y0; p.bbox_out[m * 4 + 1] = e ? -1 : *s_x0;
            p.bbox_out[m * 4 + 2] = e ? -1 : *s_y1; p.bbox_out[m * 4 + 3] = e ? -1 : *s_x1;
        }
        if (p.flag) p.flag[m] = flagged;
    }
    return false;
}

// One workgroup per mask.  `worklist` (optional, the SMALL variant only): masks that need the large variant are appended to
// worklist[2 ..] (count in worklist[0]) instead of being found again by a second launch over all masks.
template <int RW, int NT>
__global__ __launch_bounds__(NT) void mask_program_kernel(const ProgP p, const int lo, int* __restrict__ worklist) {
    __shared__ uint32_t lds[2 * RW];
    __shared__ int s_changed, s_first, s_area, s_y0, s_y1, s_x0, s_x1;
    const long m = blockIdx.x;
    const bool larger = run_mask_program<RW>(p, m, lo, lds, &s_changed, &s_first, &s_area, &s_y0, &s_y1, &s_x0, &s_x1);
    if (larger && worklist && threadIdx.x == 0) worklist[2 + atomicAdd(&worklist[0], 1)] = (int)m;
}

// The large variant over a worklist: a fixed grid whose workgroups take the next listed mask until the list is empty (the
// list is complete: the small variant ran before this kernel on the same stream).  Three of four masks of a real batch are
// small -- a launch of one 64-KiB-LDS workgroup per MASK spent most of its workgroups finding that out, each waiting for a
// CU with 64 KiB to spare beside the convolution workgroups of the other stream.
template <int RW, int NT>
__global__ __launch_bounds__(NT) void mask_program_list_kernel(const ProgP p, const int lo, int* __restrict__ worklist) {
    __shared__ uint32_t lds[2 * RW];
    __shared__ int s_changed, s_first, s_area, s_y0, s_y1, s_x0, s_x1, s_next;
    const int count = worklist[0];
    for (;;) {
        if (threadIdx.x == 0) s_next = atomicAdd(&worklist[1], 1);
        __syncthreads();
        const int i = s_next;
        if (i >= count) break;                                 // (block-uniform; every workgroup gets here)
        run_mask_program<RW>(p, worklist[2 + i], lo, lds, &s_changed, &s_first, &s_area, &s_y0, &s_y1, &s_x0, &s_x1);
        __syncthreads();
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

// The same with bounding boxes (supersets allowed): one block per mask m clears, inside m's box, the pixels of
// every earlier mask j of its segment whose box meets it.  Blocks race on purpose: block j may already have
// removed from mask j what even earlier masks cover, but those pixels are still in the union of the earlier masks
// that block m reads (each pixel stays in the first mask that has it), so m loses exactly the same pixels.
__global__ __launch_bounds__(256) void overlap_prefix_bbox_kernel(uint32_t* masks, const int* __restrict__ seg,
                                                                  const int* __restrict__ bbox, int H, int W) {
    __shared__ int sb[256 * 4];
    __shared__ int sj[256];
    __shared__ int s_cnt, s_more;
    const int m = blockIdx.x, tid = threadIdx.x;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    if (y0 < 0) return;
    const int wpr = (W + 31) >> 5;
    const int s = seg ? seg[m] : 0;
    uint32_t* mine = masks + (long)m * H * wpr;
    const int wx0 = x0 >> 5, rw = (x1 >> 5) - wx0 + 1, rh = y1 - y0 + 1;
    for (int jbase = m - 1; jbase >= 0; jbase -= 256) {
        if (tid == 0) { s_cnt = 0; s_more = 1; }
        __syncthreads();
        const int j = jbase - tid;                       // earlier masks, 256 at a time
        if (j >= 0) {
            if (seg && seg[j] != s) s_more = 0;          // the segment starts inside this chunk
            else {
                const int jy0 = bbox[j * 4 + 0], jx0 = bbox[j * 4 + 1], jy1 = bbox[j * 4 + 2], jx1 = bbox[j * 4 + 3];
                if (jy0 >= 0 && jy0 <= y1 && jy1 >= y0 && jx0 <= x1 && jx1 >= x0) {
                    const int k = atomicAdd(&s_cnt, 1);
                    sb[k * 4 + 0] = jy0; sb[k * 4 + 1] = jx0; sb[k * 4 + 2] = jy1; sb[k * 4 + 3] = jx1;
                    sj[k] = j;
                }
            }
        }
        __syncthreads();
        const int cnt = s_cnt;
        // every word of m's box belongs to ONE thread for the whole kernel: no read-modify-write race inside the block
        for (int t = tid; t < rh * rw; t += 256) {
            const int y = y0 + t / rw, wx = wx0 + t % rw;
            const long o = (long)y * wpr + wx;
            uint32_t c = mine[o];
            if (!c) continue;
            const uint32_t c_in = c;
            for (int k = 0; k < cnt; ++k) {
                if (y < sb[k * 4 + 0] || y > sb[k * 4 + 2] || wx * 32 + 31 < sb[k * 4 + 1] || wx * 32 > sb[k * 4 + 3]) continue;
                c &= ~masks[(long)sj[k] * H * wpr + o];
            }
            if (c != c_in) mine[o] = c;
        }
        const int more = s_more;
        __syncthreads();
        if (!more) break;
    }
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

// ---- |a_i & a_j| for EVERY pair of a segment (the masks of one tile / one call): block per mask i, one wave per
// candidate j > i in turn; pairs whose boxes are disjoint (nearly all) cost four scalar loads.  Row i of `out`
// ([M, ld], pre-zeroed) receives |a_i & a_j| at column j - first[i] -- the upper triangle; the host mirrors it.
// `label` (optional): only pairs with equal labels are counted (same-class pairs of a cross-class set).
__global__ __launch_bounds__(256) void pair_matrix_kernel(const uint32_t* __restrict__ a, const int* __restrict__ bbox,
                                                          const int* __restrict__ first, const int* __restrict__ count,
                                                          const int* __restrict__ label, int* __restrict__ out, int ld, int H, int W) {
    const int i = blockIdx.x;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int f = first[i], end = f + count[i];
    const int4 bi = reinterpret_cast<const int4*>(bbox)[i];          // y0, x0, y1, x1
    if (bi.x < 0) return;
    const int li = label ? label[i] : 0;
    const int wpr = (W + 31) >> 5;
    const uint32_t* ma = a + (long)i * H * wpr;
    for (int j = i + 1 + wave; j < end; j += 4) {
        if (j - f >= ld) break;
        if (label && label[j] != li) continue;
        const int4 bj = reinterpret_cast<const int4*>(bbox)[j];
        const int y0 = max(bi.x, bj.x), y1 = min(bi.z, bj.z), x0 = max(bi.y, bj.y), x1 = min(bi.w, bj.w);
        if (bj.x < 0 || y0 > y1 || x0 > x1) continue;
        const int wx0 = x0 >> 5, rw = (x1 >> 5) - wx0 + 1, rh = y1 - y0 + 1;
        const uint32_t* mb = a + (long)j * H * wpr;
        int c = 0;
        for (int t = lane; t < rh * rw; t += 64) {
            const long o = (long)(y0 + t / rw) * wpr + wx0 + t % rw;
            c += __popc(ma[o] & mb[o]);
        }
        for (int o = 32; o > 0; o >>= 1) c += __shfl_down(c, o, 64);
        if (lane == 0) out[(long)i * ld + (j - f)] = c;
    }
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

// ---- instance tables (what crosses xGMI): the bbox rows x word columns of every mask, back to back --------------
// offsets = exclusive prefix sums of (rows * word columns) per mask; PACK: masks -> payload, else payload -> masks
// (the destination planes are zero everywhere else: the caller hands in a zeroed tensor).
template <bool PACK>
__global__ __launch_bounds__(256) void crop_copy_kernel(uint32_t* masks, const int* __restrict__ bbox, const long* __restrict__ offsets,
                                                        int H, int W, uint32_t* payload) {
    const int m = blockIdx.x;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    if (y0 < 0) return;
    const int wpr = (W + 31) >> 5;
    const int c0 = x0 >> 5, cols = (x1 >> 5) - c0 + 1, rows = y1 - y0 + 1;
    uint32_t* plane = masks + (long)m * H * wpr + (long)y0 * wpr + c0;
    uint32_t* seg = payload + offsets[m];
    for (int t = threadIdx.x; t < rows * cols; t += blockDim.x) {
        const int ry = t / cols, cx = t - ry * cols;
        if (PACK) seg[t] = plane[(long)ry * wpr + cx];
        else plane[(long)ry * wpr + cx] = seg[t];
    }
}

// dst[i] = src[index[i]] for masks that are zero outside their bbox: the destination plane is WRITTEN once (zeros outside the
// box, the source words inside it) and only the box is READ -- half the traffic of a plane-to-plane gather (512 KiB read +
// 512 KiB written per 2048^2 mask), and no torch index kernel between two stages of the mask path.
// grid (M, row chunks); a thread writes four words (16 bytes) at a time.
__global__ __launch_bounds__(256) void gather_regions_kernel(const uint32_t* __restrict__ src, const long* __restrict__ index,
                                                             const int* __restrict__ bbox, uint32_t* __restrict__ dst, int H, int wpr,
                                                             int rows_per_block) {
    const int m = blockIdx.x;
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    const int c0 = x0 >> 5, c1 = x1 >> 5;
    const uint32_t* sp = src + index[m] * (long)H * wpr;
    uint32_t* dp = dst + (long)m * H * wpr;
    const int q4 = (wpr + 3) >> 2;                          // 4-word groups per row
    const int r0 = blockIdx.y * rows_per_block, r1 = min(r0 + rows_per_block, H);
    const bool vec = (wpr & 3) == 0;
    for (int t = threadIdx.x; t < (r1 - r0) * q4; t += blockDim.x) {
        const int ry = r0 + t / q4, cq = (t % q4) * 4;
        const bool row_in = y0 >= 0 && ry >= y0 && ry <= y1;
        uint32_t w[4] = {0u, 0u, 0u, 0u};
        if (row_in && cq <= c1 && cq + 3 >= c0) {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cq + k >= c0 && cq + k <= c1 && cq + k < wpr) w[k] = sp[(long)ry * wpr + cq + k];
        }
        if (vec) {
            *reinterpret_cast<uint4*>(dp + (long)ry * wpr + cq) = make_uint4(w[0], w[1], w[2], w[3]);
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (cq + k < wpr) dp[(long)ry * wpr + cq + k] = w[k];
        }
    }
}

// The same gather into a POOL of planes that are known to be zero outside `prev` (the box the previous use of the slot could
// have set): only the union of the slot's old box and the new one is written -- a few hundred words instead of a 512-KiB
// plane -- and `prev` becomes the new box grown by `grow` pixels (what the in-place stages that follow may still set: a
// dilation reaches one pixel beyond its input).  One block per slot.
__global__ __launch_bounds__(256) void gather_regions_pooled_kernel(const uint32_t* __restrict__ src, const long* __restrict__ index,
                                                                    const int* __restrict__ bbox, uint32_t* __restrict__ dst,
                                                                    int* __restrict__ prev, int H, int W, int grow) {
    const int m = blockIdx.x;
    const int wpr = (W + 31) >> 5;
    const int4 nb = reinterpret_cast<const int4*>(bbox)[m];       // y0, x0, y1, x1 (inclusive), -1: empty
    const int4 pb = reinterpret_cast<const int4*>(prev)[m];
    __syncthreads();                                              // every thread has read prev before thread 0 rewrites it
    const bool has_n = nb.x >= 0, has_p = pb.x >= 0;
    if (threadIdx.x == 0) {
        int4 g = make_int4(-1, -1, -1, -1);
        if (has_n) g = make_int4(max(nb.x - grow, 0), max(nb.y - grow, 0), min(nb.z + grow, H - 1), min(nb.w + grow, W - 1));
        reinterpret_cast<int4*>(prev)[m] = g;
    }
    if (!has_n && !has_p) return;
    const int ry0 = min(has_n ? nb.x : 1 << 30, has_p ? pb.x : 1 << 30), ry1 = max(has_n ? nb.z : -1, has_p ? pb.z : -1);
    const int c0 = min(has_n ? nb.y >> 5 : 1 << 30, has_p ? pb.y >> 5 : 1 << 30), c1 = max(has_n ? nb.w >> 5 : -1, has_p ? pb.w >> 5 : -1);
    const int nc0 = nb.y >> 5, nc1 = nb.w >> 5;
    const uint32_t* sp = src + index[m] * (long)H * wpr;
    uint32_t* dp = dst + (long)m * H * wpr;
    const int cols = c1 - c0 + 1, rows = ry1 - ry0 + 1;
    for (int t = threadIdx.x; t < rows * cols; t += blockDim.x) {
        const int ry = ry0 + t / cols, cx = c0 + t % cols;
        const bool in = has_n && ry >= nb.x && ry <= nb.z && cx >= nc0 && cx <= nc1;
        dp[(long)ry * wpr + cx] = in ? sp[(long)ry * wpr + cx] : 0u;
    }
}

inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g > 32768 ? 32768 : (g < 1 ? 1 : g));
}

// 256-bin histogram of the gray image under one mask (measurements.py:197-205: cv2.cvtColor(BGR2GRAY) then
// np.histogram(gray[mask > 0], bins=256, range=(0, 255)): with integer data bin i holds the pixels of value i).
// One workgroup per mask over its bbox words; BGR -> gray is OpenCV's 8-bit fixed point (B 1868, G 9617, R 4899, >> 14).
__global__ __launch_bounds__(256) void gray_hist_kernel(const uint32_t* __restrict__ masks, const int* __restrict__ bbox,
                                                        const uint8_t* __restrict__ img, int channels, int H, int W,
                                                        int* __restrict__ hist) {
    __shared__ int s_h[256];
    const long m = blockIdx.x;
    const int wpr = (W + 31) >> 5;
    s_h[threadIdx.x] = 0;
    __syncthreads();
    const int y0 = bbox[m * 4 + 0], x0 = bbox[m * 4 + 1], y1 = bbox[m * 4 + 2], x1 = bbox[m * 4 + 3];
    if (y0 >= 0) {
        const int ry0 = max(y0, 0), rh = min(y1, H - 1) - ry0 + 1;
        const int wx0 = max(x0, 0) >> 5, rw = (min(x1, W - 1) >> 5) - wx0 + 1;
        const uint32_t* src = masks + m * (long)H * wpr;
        for (int i = threadIdx.x; i < rh * rw; i += blockDim.x) {
            const int ly = i / rw, lx = i - ly * rw;
            const int y = ry0 + ly;
            uint32_t b = src[(long)y * wpr + wx0 + lx];
            while (b) {
                const int bit = __ffs((int)b) - 1;
                b &= b - 1;
                const long px = (long)y * W + ((wx0 + lx) << 5) + bit;
                int g;
                if (channels == 3) {
                    const uint8_t* q = img + px * 3;
                    g = (q[0] * 1868 + q[1] * 9617 + q[2] * 4899 + (1 << 13)) >> 14;
                } else {
                    g = img[px];
                }
                atomicAdd(&s_h[g], 1);
            }
        }
    }
    __syncthreads();
    hist[m * 256 + threadIdx.x] = s_h[threadIdx.x];
}

}  // namespace

extern "C" int demia_mask_program_wl(uint32_t* masks, uint32_t* scratch, const int32_t* bbox, const uint8_t* active, uint32_t program,
                                     int64_t M, int H, int W, int32_t* area, int32_t* bbox_out, int32_t* flag, int32_t* worklist, void* stream) {
    DEMIA_REQUIRE(masks && scratch && bbox && masks != scratch && W > 0 && H > 0, "args");
    for (uint32_t q = program; q; q >>= 4) DEMIA_REQUIRE((q & 15u) <= DEMIA_MOP_GATE, "unknown stage code");
    if (M == 0) return DEMIA_OK;
    DEMIA_REQUIRE(M <= 0x7ffffff0L, "M");
    ProgP p{masks, scratch, bbox, active, program, H, W, area, bbox_out, flag};
    hipStream_t st = (hipStream_t)stream;
    if (worklist) {
        if (hipMemsetAsync(worklist, 0, 2 * sizeof(int32_t), st) != hipSuccess) { demia_set_error("demia_mask_program: hipMemsetAsync failed"); return DEMIA_ELAUNCH; }
    }
    hipLaunchKernelGGL((mask_program_kernel<REG_WORDS_SMALL, 256>), dim3((int)M), dim3(256), 0, st, p, 0, worklist);
    DEMIA_CHECK_LAUNCH("mask_program_kernel<small>");
    if (worklist) {
        const int grid = (int)(M < 512 ? M : 512);            // two 64-KiB workgroups per CU
        hipLaunchKernelGGL((mask_program_list_kernel<REG_WORDS, 512>), dim3(grid), dim3(512), 0, st, p, REG_WORDS_SMALL, worklist);
        DEMIA_CHECK_LAUNCH("mask_program_list_kernel<large>");
    } else {
        hipLaunchKernelGGL((mask_program_kernel<REG_WORDS, 512>), dim3((int)M), dim3(512), 0, st, p, REG_WORDS_SMALL, (int*)nullptr);
        DEMIA_CHECK_LAUNCH("mask_program_kernel<large>");
    }
    return DEMIA_OK;
}

extern "C" int demia_mask_program(uint32_t* masks, uint32_t* scratch, const int32_t* bbox, const uint8_t* active, uint32_t program,
                                  int64_t M, int H, int W, int32_t* area, int32_t* bbox_out, int32_t* flag, void* stream) {
    return demia_mask_program_wl(masks, scratch, bbox, active, program, M, H, W, area, bbox_out, flag, nullptr, stream);
}

extern "C" int demia_mask_overlap_prefix(uint32_t* masks, const int32_t* seg, const int32_t* bbox, int64_t M, int H, int W,
                                         void* stream) {
    DEMIA_REQUIRE(masks && W > 0, "args");
    const long wpm = (long)H * ((W + 31) / 32);
    if (M == 0 || wpm == 0) return DEMIA_OK;
    if (bbox) {
        hipLaunchKernelGGL(overlap_prefix_bbox_kernel, dim3((int)M), dim3(256), 0, (hipStream_t)stream, masks, seg, bbox, H, W);
        DEMIA_CHECK_LAUNCH("overlap_prefix_bbox_kernel");
        return DEMIA_OK;
    }
    hipLaunchKernelGGL(overlap_prefix_kernel, dim3(grid_for(wpm, 256)), dim3(256), 0, (hipStream_t)stream, masks, seg, (int)M, wpm);
    DEMIA_CHECK_LAUNCH("overlap_prefix_kernel");
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

extern "C" int demia_mask_pair_matrix(const uint32_t* masks, const int32_t* bbox, const int32_t* first, const int32_t* count,
                                      const int32_t* label, int32_t* out, int64_t M, int ld, int H, int W, void* stream) {
    DEMIA_REQUIRE(masks && bbox && first && count && out && W > 0 && ld > 0, "args");
    if (M == 0) return DEMIA_OK;
    DEMIA_REQUIRE(M <= 0x7fffffffL, "M");
    hipLaunchKernelGGL(pair_matrix_kernel, dim3((int)M), dim3(256), 0, (hipStream_t)stream, masks, bbox, first, count, label, out, ld,
                       H, W);
    DEMIA_CHECK_LAUNCH("pair_matrix_kernel");
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

extern "C" int demia_mask_crop_pack(const uint32_t* masks, const int32_t* bbox, const int64_t* offsets, int64_t M, int H, int W,
                                    uint32_t* payload, void* stream) {
    DEMIA_REQUIRE(masks && bbox && offsets && payload && W > 0, "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(crop_copy_kernel<true>, dim3((int)M), dim3(256), 0, (hipStream_t)stream, const_cast<uint32_t*>(masks), bbox,
                       reinterpret_cast<const long*>(offsets), H, W, payload);
    DEMIA_CHECK_LAUNCH("crop_copy_kernel<pack>");
    return DEMIA_OK;
}

extern "C" int demia_mask_crop_unpack(const uint32_t* payload, const int32_t* bbox, const int64_t* offsets, int64_t M, int H, int W,
                                      uint32_t* masks, void* stream) {
    DEMIA_REQUIRE(masks && bbox && offsets && payload && W > 0, "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(crop_copy_kernel<false>, dim3((int)M), dim3(256), 0, (hipStream_t)stream, masks, bbox,
                       reinterpret_cast<const long*>(offsets), H, W, const_cast<uint32_t*>(payload));
    DEMIA_CHECK_LAUNCH("crop_copy_kernel<unpack>");
    return DEMIA_OK;
}

extern "C" int demia_mask_gather_regions(const uint32_t* src, const int64_t* index, const int32_t* bbox, int64_t M, int H, int W,
                                         uint32_t* dst, void* stream) {
    DEMIA_REQUIRE(src && index && bbox && dst && W > 0 && H > 0, "args");
    if (M == 0) return DEMIA_OK;
    DEMIA_REQUIRE(M <= 0x7fffffffL, "M");
    const int wpr = (W + 31) >> 5;
    // ~16 KiB of destination per workgroup: enough blocks for a handful of masks, few enough for tens of thousands
    int rows_per_block = (4096 + wpr - 1) / wpr;
    if (rows_per_block > H) rows_per_block = H;
    const int chunks = (H + rows_per_block - 1) / rows_per_block;
    DEMIA_REQUIRE(chunks <= 65535, "H");
    hipLaunchKernelGGL(gather_regions_kernel, dim3((unsigned)M, (unsigned)chunks), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const long*>(index), bbox, dst, H, wpr, rows_per_block);
    DEMIA_CHECK_LAUNCH("gather_regions_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_gather_regions_pooled(const uint32_t* src, const int64_t* index, const int32_t* bbox, int64_t M, int H, int W,
                                                uint32_t* pool, int32_t* prev, int grow, void* stream) {
    DEMIA_REQUIRE(src && index && bbox && pool && prev && W > 0 && H > 0 && grow >= 0, "args");
    if (M == 0) return DEMIA_OK;
    DEMIA_REQUIRE(M <= 0x7fffffffL, "M");
    hipLaunchKernelGGL(gather_regions_pooled_kernel, dim3((unsigned)M), dim3(256), 0, (hipStream_t)stream, src,
                       reinterpret_cast<const long*>(index), bbox, pool, prev, H, W, grow);
    DEMIA_CHECK_LAUNCH("gather_regions_pooled_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_gray_histogram(const uint32_t* masks, const int32_t* bbox, const uint8_t* image, int channels,
                                         int64_t M, int H, int W, int32_t* hist, void* stream) {
    DEMIA_REQUIRE(masks && bbox && image && hist && W > 0 && (channels == 1 || channels == 3), "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(gray_hist_kernel, dim3((int)M), dim3(256), 0, (hipStream_t)stream, masks, bbox, image, channels, H, W, hist);
    DEMIA_CHECK_LAUNCH("gray_hist_kernel");
    return DEMIA_OK;
}
