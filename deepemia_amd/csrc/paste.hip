// Full-resolution mask paste -> bit-packed masks, plus the packed-mask utilities that keep
// N x H x W masks off PCIe (the reference copies them to the host three times per call,
// src/functions/inference.py:1401-1403).
//
// Replaces detectron2.layers.mask_ops.paste_masks_in_image / _do_paste_mask (v0.6) and
// detector_postprocess's box rescale + clip + nonempty filter.  -ffp-contract=off: the
// sampling-grid arithmetic follows torch's separate fp32 ops, then the grid_sample
// (bilinear, zeros padding, align_corners=False) formula.
//
// Output layout: one bit per pixel, 32 pixels per u32 word, bit (x & 31) of word (x >> 5);
// a 2048 x 2048 mask is 512 KiB instead of the reference's 4 MiB bool array.
#include "common.h"

namespace {

struct PasteP {
    const float* mask_prob;
    int ld;
    const float* det_boxes;
    const int* det_classes;
    const int* det_count;
    int N, D, img_h, img_w, out_h, out_w;
    float* out_boxes;
    uint8_t* valid;
    uint32_t* packed;
    int* out_bbox;
    const int* prev_bbox;
};

constexpr int PASTE_ROWS = 16;   // rows per block: 4 KiB of packed output at 2048 px (most blocks only write zeros)
constexpr int PASTE_COLS = 2048; // widest box whose per-column sampling table fits the workgroup's LDS (wider boxes: per pixel)

__global__ __launch_bounds__(256) void paste_kernel(const PasteP p) {
    __shared__ float sm[28 * 28];
    __shared__ float sbox[4];
    __shared__ int sflag;
    const int inst = blockIdx.y;  // n * D + i
    const int n = inst / p.D, i = inst - n * p.D;
    const int tid = threadIdx.x;
    const int wpr = (p.out_w + 31) >> 5;  // words per row
    if (tid == 0) {
        int ok = i < p.det_count[n];
        float4 b = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ok) {
            const float4 nb = reinterpret_cast<const float4*>(p.det_boxes)[inst];
            const float sx = (float)((double)p.out_w / (double)p.img_w);
            const float sy = (float)((double)p.out_h / (double)p.img_h);
            b.x = fminf(fmaxf(nb.x * sx, 0.f), (float)p.out_w);
            b.z = fminf(fmaxf(nb.z * sx, 0.f), (float)p.out_w);
            b.y = fminf(fmaxf(nb.y * sy, 0.f), (float)p.out_h);
            b.w = fminf(fmaxf(nb.w * sy, 0.f), (float)p.out_h);
            ok = ((b.z - b.x) > 0.f) && ((b.w - b.y) > 0.f);
        }
        sbox[0] = b.x; sbox[1] = b.y; sbox[2] = b.z; sbox[3] = b.w;
        sflag = ok;
        if (blockIdx.x == 0) {
            reinterpret_cast<float4*>(p.out_boxes)[inst] = b;
            p.valid[inst] = (uint8_t)ok;
            if (p.out_bbox) {     // the pixels the paste can set: y0, x0, y1, x1 inclusive (-1: none)
                int4 h = make_int4(-1, -1, -1, -1);
                if (ok) {
                    h.x = max((int)floorf(b.y) - 1, 0); h.y = max((int)floorf(b.x) - 1, 0);
                    h.z = min((int)ceilf(b.w) + 1, p.out_h) - 1; h.w = min((int)ceilf(b.z) + 1, p.out_w) - 1;
                }
                reinterpret_cast<int4*>(p.out_bbox)[inst] = h;
            }
        }
    }
    __syncthreads();
    const bool ok = sflag != 0;
    const float x0 = sbox[0], y0 = sbox[1], x1 = sbox[2], y1 = sbox[3];
    const int x0i = max((int)floorf(x0) - 1, 0), y0i = max((int)floorf(y0) - 1, 0);
    const int x1i = min((int)ceilf(x1) + 1, p.out_w), y1i = min((int)ceilf(y1) + 1, p.out_h);
    // Rows and word columns of this instance's plane that have to be written.  Whole plane (prev_bbox == NULL): all of
    // them, the plane's previous content is unknown.  INCREMENTAL (prev_bbox given): the plane is known to be zero outside
    // the box the PREVIOUS paste into this buffer could set, so only the union of that box and the new one is written --
    // a 2048 x 2048 plane is 512 KiB, a detection's box a few KiB: the whole-plane paste of a batch is mostly a memset.
    int ry0 = 0, ry1 = p.out_h, wlo = 0, whi = wpr - 1;
    if (p.prev_bbox) {
        const int4 pb = reinterpret_cast<const int4*>(p.prev_bbox)[inst];
        ry0 = 1 << 30; ry1 = -1; wlo = 1 << 30; whi = -1;
        if (pb.x >= 0) { ry0 = pb.x; ry1 = pb.z + 1; wlo = pb.y >> 5; whi = pb.w >> 5; }
        if (ok) { ry0 = min(ry0, y0i); ry1 = max(ry1, y1i); wlo = min(wlo, x0i >> 5); whi = max(whi, (x1i - 1) >> 5); }
        if (ry1 <= ry0) return;                      // nothing was pasted here and nothing is now (block-uniform)
    }
    const float invw = x1 - x0, invh = y1 - y0;
    // The column part of the sampling (grid coordinate, west / east taps and weights) is the same for every row of the
    // mask: worked out once per workgroup for the box's columns -- same operations in the same order as per pixel, so the
    // bits do not change -- instead of once per pixel (the division alone was a third of the per-pixel work).
    __shared__ float s_we[PASTE_COLS];
    __shared__ short s_xi[PASTE_COLS];
    const int ncol = x1i - x0i;
    const bool col_table = ncol <= PASTE_COLS;
    if (ok) {                                        // (block-uniform)
        const int cls = p.det_classes[inst];
        const float* mp = p.mask_prob + (long)inst * 196 * 4 * p.ld + cls;
        for (int e = tid; e < 784; e += 256) {
            const int yy = e / 28, xx = e - yy * 28;
            const int cell = (yy >> 1) * 14 + (xx >> 1), sub = (yy & 1) * 2 + (xx & 1);
            sm[e] = mp[(long)(cell * 4 + sub) * p.ld];
        }
        if (col_table) {
            for (int cix = tid; cix < ncol; cix += 256) {
                const int X = x0i + cix;
                float gx = ((float)X + 0.5f - x0) / invw * 2.0f - 1.0f;
                const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
                const float xw = floorf(ix);
                s_we[cix] = ix - xw;
                // clamped before narrowing: for a box clipped to a sliver |ix| can exceed what a short holds, and a wrapped
                // value could land on a valid tap; anything <= -2 or >= 29 has both taps outside the 28-wide mask either way
                s_xi[cix] = (short)(int)fminf(fmaxf(xw, -2.0f), 29.0f);
            }
        }
        __syncthreads();
    }
    // the bits of one 32-pixel word of the plane (row Y, word column wx): zero outside the box
    auto word_bits = [&](int Y, int wx) -> uint32_t {
        uint32_t bits = 0u;
        const int xa = wx << 5;
        if (ok && Y >= y0i && Y < y1i && xa < x1i && xa + 32 > x0i) {
            float gy = ((float)Y + 0.5f - y0) / invh * 2.0f - 1.0f;
            const float iy = ((gy + 1.0f) * 28.0f - 1.0f) / 2.0f;
            const float yn = floorf(iy);
            const float ns = iy - yn, ss = 1.0f - ns;  // weights: n (south part), s
            const int yi0 = (int)yn, yi1 = yi0 + 1;
            const bool vy0 = (unsigned)yi0 < 28u, vy1 = (unsigned)yi1 < 28u;
            const float* r0 = sm + (vy0 ? yi0 : 0) * 28;
            const float* r1 = sm + (vy1 ? yi1 : 0) * 28;
            for (int bx = 0; bx < 32; ++bx) {
                const int X = xa + bx;
                if (X < x0i || X >= x1i) continue;
                float we;
                int xi0;
                if (col_table) {
                    we = s_we[X - x0i];
                    xi0 = s_xi[X - x0i];
                } else {
                    float gx = ((float)X + 0.5f - x0) / invw * 2.0f - 1.0f;
                    const float ix = ((gx + 1.0f) * 28.0f - 1.0f) / 2.0f;
                    const float xw = floorf(ix);
                    we = ix - xw;
                    xi0 = (int)xw;
                }
                const float ww = 1.0f - we;
                const int xi1 = xi0 + 1;
                const bool vx0 = (unsigned)xi0 < 28u, vx1 = (unsigned)xi1 < 28u;
                const float nw_ = (vy0 && vx0) ? r0[xi0] : 0.f;
                const float ne = (vy0 && vx1) ? r0[xi1] : 0.f;
                const float sw = (vy1 && vx0) ? r1[xi0] : 0.f;
                const float se = (vy1 && vx1) ? r1[xi1] : 0.f;
                const float v = nw_ * (ss * ww) + ne * (ss * we) + sw * (ns * ww) + se * (ns * we);
                if (v >= 0.5f) bits |= 1u << bx;
            }
        }
        return bits;
    };
    uint32_t* plane = p.packed + (long)inst * p.out_h * wpr;
    if (p.prev_bbox) {
        // INCREMENTAL, two rectangles instead of their union: A = the words the new box touches (pasted), B = the words the
        // PREVIOUS paste into this plane could have set (cleared where A does not cover them).  When consecutive replays bring
        // different tiles the two boxes lie anywhere in the frame, and their union -- what this kernel wrote until round 5 --
        // is mostly zeros between them (0.87 ms per 48-tile step); the rectangles themselves are a few KiB per instance.
        const int4 pb = reinterpret_cast<const int4*>(p.prev_bbox)[inst];
        const int ar0 = ok ? y0i : 0, ar1 = ok ? y1i : 0, aw0 = ok ? (x0i >> 5) : 0, aw1 = ok ? ((x1i - 1) >> 5) : -1;
        const int br0 = pb.x >= 0 ? pb.x : 0, br1 = pb.x >= 0 ? pb.z + 1 : 0, bw0 = pb.x >= 0 ? (pb.y >> 5) : 0, bw1 = pb.x >= 0 ? (pb.w >> 5) : -1;
        const int anw = aw1 - aw0 + 1, bnw = bw1 - bw0 + 1;
        const int na = (ar1 - ar0) * anw, nb = (br1 - br0) * bnw;
        for (int t = blockIdx.x * 256 + tid; t < na + nb; t += gridDim.x * 256) {
            if (t < na) {
                const int ry = t / anw, wx = aw0 + (t - ry * anw);
                plane[(long)(ar0 + ry) * wpr + wx] = word_bits(ar0 + ry, wx);
            } else {
                const int u = t - na;
                const int ry = u / bnw, wx = bw0 + (u - ry * bnw);
                const int Y = br0 + ry;
                if (!(Y >= ar0 && Y < ar1 && wx >= aw0 && wx <= aw1)) plane[(long)Y * wpr + wx] = 0u;
            }
        }
        return;
    }
    const int c0 = ry0 / PASTE_ROWS, c1 = (ry1 - 1) / PASTE_ROWS;
    const int nw = whi - wlo + 1;
    for (int chunk = c0 + blockIdx.x; chunk <= c1; chunk += gridDim.x) {
        const int row0 = chunk * PASTE_ROWS;
        const int nrows = min(PASTE_ROWS, p.out_h - row0);
        uint32_t* dst = plane + (long)row0 * wpr;
        const bool live = ok && (row0 < y1i) && (row0 + nrows > y0i);
        if (!live && nw == wpr && (wpr & 3) == 0) {   // whole rows of zeros, 16-byte stores (rows and the plane base are 16-byte multiples)
            uint4* d4 = reinterpret_cast<uint4*>(dst);
            for (int w = tid; w < ((nrows * wpr) >> 2); w += 256) d4[w] = make_uint4(0u, 0u, 0u, 0u);
            continue;
        }
        for (int t = tid; t < nrows * nw; t += 256) {
            const int ry = t / nw, wx = wlo + (t - ry * nw);
            dst[ry * wpr + wx] = live ? word_bits(row0 + ry, wx) : 0u;
        }
    }
}

__global__ void unpack_kernel(const uint32_t* __restrict__ packed, uint8_t* __restrict__ out, long nwords, int W) {
    const int wpr = (W + 31) >> 5;
    const bool fast = (W & 31) == 0;
    for (long w = blockIdx.x * (long)blockDim.x + threadIdx.x; w < nwords; w += (long)gridDim.x * blockDim.x) {
        const uint32_t b = packed[w];
        if (fast) {
            uint32_t o[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const uint32_t nib = (b >> (4 * q)) & 15u;
                o[q] = (nib & 1u) | ((nib & 2u) << 7) | ((nib & 4u) << 14) | ((nib & 8u) << 21);
            }
            uint4* d = reinterpret_cast<uint4*>(out + w * 32);
            d[0] = make_uint4(o[0], o[1], o[2], o[3]);
            d[1] = make_uint4(o[4], o[5], o[6], o[7]);
        } else {   // rows are W bytes long: byte stores, the last word of a row is partial
            const long row = w / wpr;
            const int wx = (int)(w - row * wpr);
            const int n = min(32, W - wx * 32);
            uint8_t* d = out + row * W + wx * 32;
            for (int q = 0; q < n; ++q) d[q] = (uint8_t)((b >> q) & 1u);
        }
    }
}

// one block per mask: popcount + tight bbox (y0, x0, y1, x1 inclusive; -1 when empty).  hint (optional): a box
// known to contain every set pixel (e.g. the paste box) -- only that region is read.
__global__ __launch_bounds__(1024) void area_bbox_kernel(const uint32_t* __restrict__ packed, const int* __restrict__ hint,
                                                         int* __restrict__ area, int* __restrict__ bbox, int H, int W) {
    __shared__ int s_area, s_y0, s_y1, s_x0, s_x1;
    const long m = blockIdx.x;
    const int wpr = (W + 31) >> 5;
    const uint32_t* src = packed + m * (long)H * wpr;
    if (threadIdx.x == 0) { s_area = 0; s_y0 = 1 << 30; s_x0 = 1 << 30; s_y1 = -1; s_x1 = -1; }
    __syncthreads();
    int ry0 = 0, rh = H, wx0 = 0, rw = wpr;
    if (hint) {
        const int hy0 = hint[m * 4 + 0], hx0 = hint[m * 4 + 1], hy1 = hint[m * 4 + 2], hx1 = hint[m * 4 + 3];
        if (hy0 < 0) rh = 0;
        else {
            ry0 = max(hy0, 0); rh = min(hy1, H - 1) - ry0 + 1;
            wx0 = max(hx0, 0) >> 5; rw = (min(hx1, W - 1) >> 5) - wx0 + 1;
        }
    }
    int a = 0, y0 = 1 << 30, y1 = -1, x0 = 1 << 30, x1 = -1;
    for (int i = threadIdx.x; i < rh * rw; i += blockDim.x) {
        const int ly = i / rw, lx = i - ly * rw;
        const uint32_t b = src[(long)(ry0 + ly) * wpr + wx0 + lx];
        if (b) {
            const int y = ry0 + ly, xb = (wx0 + lx) << 5;
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
    if ((threadIdx.x & 63) == 0 && a) {
        atomicAdd(&s_area, a);
        atomicMin(&s_y0, y0); atomicMin(&s_x0, x0);
        atomicMax(&s_y1, y1); atomicMax(&s_x1, x1);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        area[m] = s_area;
        const bool e = s_area == 0;
        bbox[m * 4 + 0] = e ? -1 : s_y0; bbox[m * 4 + 1] = e ? -1 : s_x0;
        bbox[m * 4 + 2] = e ? -1 : s_y1; bbox[m * 4 + 3] = e ? -1 : s_x1;
    }
}

}  // namespace

extern "C" int demia_paste_masks(const demia_paste_desc* d, void* stream) {
    DEMIA_REQUIRE(d && d->mask_prob && d->det_boxes && d->det_classes && d->det_count && d->out_boxes && d->valid &&
                      d->packed, "null pointer");
    DEMIA_REQUIRE(d->out_w > 0 && d->out_h > 0, "output size");
    DEMIA_REQUIRE(d->N * d->D <= 65535, "N*D <= 65535");
    PasteP p;
    p.mask_prob = d->mask_prob; p.ld = d->ld; p.det_boxes = d->det_boxes; p.det_classes = d->det_classes;
    p.det_count = d->det_count; p.N = d->N; p.D = d->D; p.img_h = d->img_h; p.img_w = d->img_w;
    p.out_h = d->out_h; p.out_w = d->out_w; p.out_boxes = d->out_boxes; p.valid = d->valid; p.packed = d->packed;
    p.out_bbox = d->out_bbox;
    p.prev_bbox = d->prev_bbox;
    DEMIA_REQUIRE(!d->prev_bbox || (d->out_bbox && d->prev_bbox != d->out_bbox), "incremental paste: out_bbox set and distinct from prev_bbox");
    if (d->N * d->D == 0) return DEMIA_OK;
    // whole planes: one block per 16-row chunk; incremental: eight blocks share the chunks of an instance's two boxes
    const int gx = d->prev_bbox ? 2 : cdiv(d->out_h, PASTE_ROWS);     // incremental: two rectangles of a few hundred words per instance
    hipLaunchKernelGGL(paste_kernel, dim3(gx, d->N * d->D), dim3(256), 0, (hipStream_t)stream, p);
    DEMIA_CHECK_LAUNCH("paste_kernel");
    return DEMIA_OK;
}

extern "C" int demia_unpack_masks(const uint32_t* packed, uint8_t* out_bool, int64_t M, int H, int W, void* stream) {
    DEMIA_REQUIRE(packed && out_bool && W > 0, "args");
    const long nwords = (long)M * H * ((W + 31) / 32);
    if (nwords == 0) return DEMIA_OK;
    long g = (nwords + 255) / 256;
    if (g > 32768) g = 32768;
    hipLaunchKernelGGL(unpack_kernel, dim3((int)g), dim3(256), 0, (hipStream_t)stream, packed, out_bool, nwords, W);
    DEMIA_CHECK_LAUNCH("unpack_kernel");
    return DEMIA_OK;
}

extern "C" int demia_mask_area_bbox(const uint32_t* packed, const int32_t* hint, int32_t* area, int32_t* bbox, int64_t M, int H, int W,
                                    void* stream) {
    DEMIA_REQUIRE(packed && area && bbox && W > 0, "args");
    if (M == 0) return DEMIA_OK;
    hipLaunchKernelGGL(area_bbox_kernel, dim3((int)M), dim3(hint ? 256 : 1024), 0, (hipStream_t)stream, packed, hint, area, bbox, H, W);
    DEMIA_CHECK_LAUNCH("area_bbox_kernel");
    return DEMIA_OK;
}
