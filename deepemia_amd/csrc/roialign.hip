// ROIAlignV2 (aligned=True, sampling_ratio=0 -> adaptive grid) over the FPN pyramid, NHWC.
// HBM/L2 gather-bound: one wavefront per output bin, lanes span the channel axis so every
// bilinear tap is one coalesced 256..1024-byte row read; the adaptive sample grid is walked
// in registers.  -ffp-contract=off: coordinate arithmetic follows torchvision's op order.
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
};

template <typename T>
__global__ __launch_bounds__(256) void roi_align_kernel(const RoiP p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long bin = (long)blockIdx.x * 4 + wave;
    const int PP = p.P * p.P;
    const long total = (long)p.N * p.R * PP;
    if (bin >= total) return;
    const int pw = (int)(bin % p.P);
    const int ph = (int)((bin / p.P) % p.P);
    const long roi = bin / PP;  // n*R + r
    const int n = (int)(roi / p.R), r = (int)(roi % p.R);
    T* out = reinterpret_cast<T*>(p.out) + bin * p.C;
    const int cpl = p.C / 64;  // channels per lane (C = 256 -> 4)
    if (r >= p.count[n]) {
        for (int c = 0; c < cpl; ++c) out[lane * cpl + c] = from_f32<T>(0.f);
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
    const T* feat = reinterpret_cast<const T*>(p.feat[lv]) + (long)n * H * W * p.C;

    const float rsw = b.x * scale - 0.5f, rsh = b.y * scale - 0.5f;
    const float rew = b.z * scale - 0.5f, reh = b.w * scale - 0.5f;
    const float rw = rew - rsw, rh = reh - rsh;
    const float bin_h = rh / (float)p.P, bin_w = rw / (float)p.P;
    const int gh = (int)ceilf(rh / (float)p.P), gw = (int)ceilf(rw / (float)p.P);
    const float cnt = (float)max(gh * gw, 1);

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
            const T* p1 = feat + ((long)yl * W + xl) * p.C + lane * cpl;
            const T* p2 = feat + ((long)yl * W + xh) * p.C + lane * cpl;
            const T* p3 = feat + ((long)yh * W + xl) * p.C + lane * cpl;
            const T* p4 = feat + ((long)yh * W + xh) * p.C + lane * cpl;
            for (int c = 0; c < 4; ++c) {
                if (c < cpl) {
                    const float v = w1 * to_f32<T>(p1[c]) + w2 * to_f32<T>(p2[c]) + w3 * to_f32<T>(p3[c]) +
                                    w4 * to_f32<T>(p4[c]);
                    acc[c] += v;
                }
            }
        }
    }
    for (int c = 0; c < cpl && c < 4; ++c) out[lane * cpl + c] = from_f32<T>(acc[c] / cnt);
}

}  // namespace

extern "C" int demia_roi_align(const demia_roialign_desc* d, void* stream) {
    DEMIA_REQUIRE(d && d->boxes && d->count && d->out, "null pointer");
    DEMIA_REQUIRE(d->C % 64 == 0 && d->C <= 256, "C must be 64, 128, 192 or 256");
    DEMIA_REQUIRE(d->P > 0, "P");
    RoiP p;
    for (int l = 0; l < 4; ++l) {
        DEMIA_REQUIRE(d->feat[l], "feat pointer");
        p.feat[l] = d->feat[l]; p.H[l] = d->H[l]; p.W[l] = d->W[l];
    }
    p.N = d->N; p.R = d->R; p.C = d->C; p.P = d->P; p.boxes = d->boxes; p.count = d->count; p.out = d->out;
    const long total = (long)d->N * d->R * d->P * d->P;
    if (total == 0) return DEMIA_OK;
    const int grid = (int)((total + 3) / 4);
    if (d->dtype == DEMIA_BF16)
        hipLaunchKernelGGL(roi_align_kernel<bf16_t>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    else
        hipLaunchKernelGGL(roi_align_kernel<float>, dim3(grid), dim3(256), 0, (hipStream_t)stream, p);
    DEMIA_CHECK_LAUNCH("roi_align_kernel");
    return DEMIA_OK;
}
