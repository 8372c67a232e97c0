// Input side of the network: Pillow-exact resize, normalise+pad, 7x7 stem conv, max pools.
// All HBM-bound or tiny; written for coalesced NHWC access, no MFMA reshaping.
//
// Replaces (Detectron2 0.6, reached from reference src/functions/inference.py:1395):
//   DefaultPredictor.__call__: ResizeShortestEdge -> PIL Image.resize(BILINEAR)
//   GeneralizedRCNN.preprocess_image: (x - pixel_mean) / 1, ImageList pad to /32
//   BasicStem: conv7x7 s2 p3 + FrozenBN + ReLU + max_pool2d(3, 2, 1)
//   LastLevelMaxPool: max_pool2d(kernel 1, stride 2)
#include "common.h"

namespace {

__device__ __forceinline__ int clip8(int v) {
    v >>= 22;  // PRECISION_BITS = 32 - 8 - 2
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// horizontal pass: one thread per output pixel (3 channels)
__global__ void resize_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, long total, int W, int newW,
                                const int* __restrict__ xmin, const int* __restrict__ xsize,
                                const int* __restrict__ xk, int ks) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % newW);
        const long row = i / newW;  // n*H + y
        const uint8_t* s = src + (row * W + xmin[xx]) * 3;
        const int n = xsize[xx];
        const int* k = xk + (long)xx * ks;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int x = 0; x < n; ++x) {
            const int kv = k[x];
            a0 += (int)s[x * 3 + 0] * kv;
            a1 += (int)s[x * 3 + 1] * kv;
            a2 += (int)s[x * 3 + 2] * kv;
        }
        uint8_t* d = tmp + i * 3;
        d[0] = (uint8_t)clip8(a0);
        d[1] = (uint8_t)clip8(a1);
        d[2] = (uint8_t)clip8(a2);
    }
}

// The same pass with the source row staged in LDS (round 3): one workgroup per image row.  The row's 3 W bytes come in
// as 16-byte loads from the 16-byte-aligned address at or below its first byte (3-byte pixels at a per-thread stride of
// 3 * scale bytes made the plain version a byte-load kernel: 840 MB in 0.78 ms), the taps are read from LDS, and the 3 newW
// output bytes leave through LDS as 4-byte stores.  Same integer arithmetic, same bytes out.
__global__ __launch_bounds__(256) void resize_h_row_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ tmp, long src_bytes, int W,
                                                           int newW, const int* __restrict__ xmin, const int* __restrict__ xsize,
                                                           const int* __restrict__ xk, int ks) {
    extern __shared__ __attribute__((aligned(16))) uint8_t rowbuf[];      // [in: 3 W + 32 | out: 3 newW + 8]
    const long row = blockIdx.x;
    const long b0 = row * (long)W * 3;                                    // first byte of the row
    const int mis = (int)((reinterpret_cast<unsigned long long>(src) + (unsigned long long)b0) & 15ull);
    const long a0 = b0 - mis;                                             // 16-byte-aligned ADDRESS at or below it (>= the allocation: offsets < 0 only for a misaligned base, never read below byte 0 of a 16-aligned block)
    const int nbytes = W * 3 + mis;
    for (int o = threadIdx.x * 16; o < nbytes; o += 256 * 16) {
        if (a0 + o >= 0 && a0 + o + 16 <= src_bytes) {
            *reinterpret_cast<uint4*>(rowbuf + o) = *reinterpret_cast<const uint4*>(src + a0 + o);
        } else {
            for (int q = 0; q < 16; ++q)
                if (a0 + o + q >= 0 && a0 + o + q < src_bytes) rowbuf[o + q] = src[a0 + o + q];
        }
    }
    __syncthreads();
    const int in_sz = (W * 3 + 32 + 15) & ~15;
    uint8_t* outb = rowbuf + in_sz;
    for (int xx = threadIdx.x; xx < newW; xx += 256) {
        const uint8_t* s = rowbuf + mis + xmin[xx] * 3;
        const int n = xsize[xx];
        const int* k = xk + (long)xx * ks;
        int a0_ = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        for (int x = 0; x < n; ++x) {
            const int kv = k[x];
            a0_ += (int)s[x * 3 + 0] * kv;
            a1 += (int)s[x * 3 + 1] * kv;
            a2 += (int)s[x * 3 + 2] * kv;
        }
        outb[xx * 3 + 0] = (uint8_t)clip8(a0_);
        outb[xx * 3 + 1] = (uint8_t)clip8(a1);
        outb[xx * 3 + 2] = (uint8_t)clip8(a2);
    }
    __syncthreads();
    const long d0 = row * (long)newW * 3;
    const int ob = newW * 3;
    const int dmis = (int)((4 - (d0 & 3)) & 3);                           // bytes up to the first 4-byte boundary of the output row
    uint8_t* d = tmp + d0;
    if ((int)threadIdx.x < dmis && (int)threadIdx.x < ob) d[threadIdx.x] = outb[threadIdx.x];
    const int nw = (ob - dmis) >> 2;
    for (int w4 = threadIdx.x; w4 < nw; w4 += 256) {
        const uint8_t* q = outb + dmis + 4 * w4;
        *reinterpret_cast<uint32_t*>(d + dmis + 4 * w4) = (uint32_t)q[0] | ((uint32_t)q[1] << 8) | ((uint32_t)q[2] << 16) | ((uint32_t)q[3] << 24);
    }
    for (int t = dmis + 4 * nw + threadIdx.x; t < ob; t += 256) d[t] = outb[t];
}

// (Round 3 also tried keeping a thread's taps in registers, for 1, 2 or 16 rows per workgroup, so that the 19 KB coefficient
//  table is not re-read per row: same bytes out, 0.3 ms SLOWER for 48 tiles -- the table reads hit L1 and the dynamic tap loop
//  beats eight predicated taps; one short-lived workgroup per row it stays.)

// vertical pass + (x - mean) + write into the zero-bordered, 4-channel f32 stem input
__global__ void resize_v_norm_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ dst, long total, int H, int newW,
                                     int newH, int PH, int PW, const int* __restrict__ ymin,
                                     const int* __restrict__ ysize, const int* __restrict__ yk, int ks,
                                     float m0, float m1, float m2) {
    const int DW = PW + 8, DH = PH + 6;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int xx = (int)(i % newW);
        const long t = i / newW;
        const int yy = (int)(t % newH);
        const long n = t / newH;
        const uint8_t* s = tmp + ((n * H + ymin[yy]) * (long)newW + xx) * 3;
        const int cnt = ysize[yy];
        const int* k = yk + (long)yy * ks;
        int a0 = 1 << 21, a1 = 1 << 21, a2 = 1 << 21;
        const long rs = (long)newW * 3;
        for (int y = 0; y < cnt; ++y) {
            const int kv = k[y];
            a0 += (int)s[y * rs + 0] * kv;
            a1 += (int)s[y * rs + 1] * kv;
            a2 += (int)s[y * rs + 2] * kv;
        }
        float4 o;
        o.x = (float)clip8(a0) - m0;
        o.y = (float)clip8(a1) - m1;
        o.z = (float)clip8(a2) - m2;
        o.w = 0.f;
        *reinterpret_cast<float4*>(dst + ((n * DH + yy + 3) * (long)DW + xx + 3) * 4) = o;
    }
}

// The same with FOUR output pixels per thread: a tap row contributes 12 consecutive bytes = three aligned dwords (the plain
// version issued 3 one-byte loads per pixel and tap).  Needs newW % 4 == 0 (then every row of `tmp` starts on a dword).
__global__ void resize_v_norm4_kernel(const uint8_t* __restrict__ tmp, float* __restrict__ dst, long total4, int H, int newW,
                                      int newH, int PH, int PW, const int* __restrict__ ymin,
                                      const int* __restrict__ ysize, const int* __restrict__ yk, int ks,
                                      float m0, float m1, float m2) {
    const int DW = PW + 8, DH = PH + 6;
    const int q4 = newW >> 2;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total4; i += (long)gridDim.x * blockDim.x) {
        const int xq = (int)(i % q4);
        const long t = i / q4;
        const int yy = (int)(t % newH);
        const long n = t / newH;
        const uint32_t* s = reinterpret_cast<const uint32_t*>(tmp + ((n * H + ymin[yy]) * (long)newW + xq * 4) * 3);
        const int cnt = ysize[yy];
        const int* k = yk + (long)yy * ks;
        int acc[12];
#pragma unroll
        for (int c = 0; c < 12; ++c) acc[c] = 1 << 21;
        const long rs = (long)newW * 3 / 4;                                // dwords per row
        for (int y = 0; y < cnt; ++y) {
            const int kv = k[y];
            const uint32_t w0 = s[y * rs], w1 = s[y * rs + 1], w2 = s[y * rs + 2];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                acc[c] += (int)((w0 >> (8 * c)) & 255u) * kv;
                acc[4 + c] += (int)((w1 >> (8 * c)) & 255u) * kv;
                acc[8 + c] += (int)((w2 >> (8 * c)) & 255u) * kv;
            }
        }
        float* o = dst + ((n * DH + yy + 3) * (long)DW + xq * 4 + 3) * 4;
#pragma unroll
        for (int px = 0; px < 4; ++px)
            *reinterpret_cast<float4*>(o + 4 * px) = make_float4((float)clip8(acc[3 * px]) - m0, (float)clip8(acc[3 * px + 1]) - m1,
                                                                 (float)clip8(acc[3 * px + 2]) - m2, 0.f);
    }
}

// cv2.resize(INTER_LINEAR) on 8-bit 3-channel images, OpenCV's fixed-point path:
// horizontal: S[sx]*a0 + S[sx+1]*a1 (11-bit coefficients), vertical:
// (((b0 * (r0 >> 4)) >> 16) + ((b1 * (r1 >> 4)) >> 16) + 2) >> 2.  Tables come from the host.
__global__ void resize_linear_u8_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, long total, int H, int W,
                                        int oh, int ow, const int* __restrict__ xofs, const short* __restrict__ ialpha,
                                        const int* __restrict__ yofs, const short* __restrict__ ibeta) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int dx = (int)(i % ow);
        const long t = i / ow;
        const int dy = (int)(t % oh);
        const long n = t / oh;
        const int sx0 = xofs[2 * dx], sx1 = xofs[2 * dx + 1];
        const int a0 = ialpha[2 * dx], a1 = ialpha[2 * dx + 1];
        const int sy0 = yofs[2 * dy], sy1 = yofs[2 * dy + 1];
        const int b0 = ibeta[2 * dy], b1 = ibeta[2 * dy + 1];
        const uint8_t* r0 = src + ((n * H + sy0) * (long)W) * 3;
        const uint8_t* r1 = src + ((n * H + sy1) * (long)W) * 3;
        uint8_t* o = dst + i * 3;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int h0 = (int)r0[sx0 * 3 + c] * a0 + (int)r0[sx1 * 3 + c] * a1;
            const int h1 = (int)r1[sx0 * 3 + c] * a0 + (int)r1[sx1 * 3 + c] * a1;
            const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
            o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
        }
    }
}

// ---- stem: 7x7 s2 p3, 3(4) -> 64, + FrozenBN + ReLU ----------------------------------
// block = 256 threads = 16x16 output pixels; thread = 4 adjacent pixels x 16 channels.
// LDS: input patch 37 x 40 x 4 f32 and the whole 7x7x4x64 f32 filter bank.
constexpr int ST_TW = 16, ST_TH = 16;
constexpr int ST_IH = ST_TH * 2 + 5, ST_IW = 40;  // 37 rows, 37 cols padded to 40
template <typename TO>
__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ in, const float* __restrict__ w,
                                                       const float* __restrict__ scale, const float* __restrict__ bias,
                                                       TO* __restrict__ out, int PH, int PW) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* sIn = reinterpret_cast<float4*>(smem);                           // [37][40] float4
    float4* sW = reinterpret_cast<float4*>(smem + ST_IH * ST_IW * 16);       // [7*7*4][16] float4 (64 co)
    const int Ho = PH / 2, Wo = PW / 2;
    const int DW = PW + 8, DH = PH + 6;
    const int n = blockIdx.z;
    const int ho0 = blockIdx.y * ST_TH, wo0 = blockIdx.x * ST_TW;
    const int tid = threadIdx.x;
    // weights: [kh][kw(8)][c(4)][64] in global -> we only stage kw < 7
    for (int i = tid; i < 7 * 7 * 4 * 16; i += 256) {
        const int co4 = i & 15;
        const int c = (i >> 4) & 3;
        const int kw = (i >> 6) % 7;
        const int kh = (i >> 6) / 7;
        sW[i] = *reinterpret_cast<const float4*>(w + (((kh * 8 + kw) * 4 + c) * 64) + co4 * 4);
    }
    // input patch: rows 2*ho0 .. 2*ho0+36 (buffer coords: +3 border already included), cols 2*wo0 .. +36
    const float4* in4 = reinterpret_cast<const float4*>(in) + (long)n * DH * DW;
    for (int i = tid; i < ST_IH * ST_IW; i += 256) {
        const int r = i / ST_IW, c = i - r * ST_IW;
        const int gy = 2 * ho0 + r, gx = 2 * wo0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < DH && gx < DW) v = in4[(long)gy * DW + gx];
        sIn[i] = v;
    }
    __syncthreads();
    const int cg = tid & 3;
    const int q = tid >> 2;
    const int row = q >> 2;
    const int col0 = (q & 3) * 4;
    // accumulators as pairs: v_pk_fma_f32 does two f32 FMAs per lane and instruction (same fused arithmetic per element)
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[4][8];
#pragma unroll
    for (int p = 0; p < 4; ++p)
#pragma unroll
        for (int c = 0; c < 8; ++c) acc2[p][c] = f32x2{0.f, 0.f};
    for (int kh = 0; kh < 7; ++kh) {
        const float4* irow = sIn + (2 * row + kh) * ST_IW + 2 * col0;
#pragma unroll
        for (int kw = 0; kw < 7; ++kw) {
            float4 x[4];
#pragma unroll
            for (int p = 0; p < 4; ++p) x[p] = irow[2 * p + kw];
            const float4* wp = sW + ((kh * 7 + kw) * 4) * 16 + cg * 4;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                f32x2 wv[8];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float4 t = wp[c * 16 + j];
                    wv[2 * j] = f32x2{t.x, t.y};
                    wv[2 * j + 1] = f32x2{t.z, t.w};
                }
#pragma unroll
                for (int p = 0; p < 4; ++p) {
                    const float xs = c == 0 ? x[p].x : (c == 1 ? x[p].y : x[p].z);
                    const f32x2 xv = {xs, xs};
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc2[p][j] = __builtin_elementwise_fma(xv, wv[j], acc2[p][j]);
                }
            }
        }
    }
    const int ho = ho0 + row;
    if (ho >= Ho) return;
    float sc[16], bs[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        sc[j] = scale[cg * 16 + j];
        bs[j] = bias[cg * 16 + j];
    }
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int wo = wo0 + col0 + p;
        if (wo >= Wo) continue;
        TO* o = out + (((long)n * Ho + ho) * Wo + wo) * 64 + cg * 16;
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = acc2[p][j >> 1][j & 1] * sc[j] + bs[j];
            o[j] = from_f32<TO>(v > 0.f ? v : 0.f);
        }
    }
}

// maxpool 3x3 s2 p1 (padding never wins: implicit -inf), NHWC, 4 channels per thread
template <typename T>
__global__ void maxpool3x3s2_kernel(const T* __restrict__ in, T* __restrict__ out, long total, int H, int W, int C,
                                    int Ho, int Wo) {
    const int C4 = C / 4;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c4 = (int)(i % C4);
        long t = i / C4;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = 2 * ho - 1 + dy;
            if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int x = 2 * wo - 1 + dx;
                if ((unsigned)x >= (unsigned)W) continue;
                const T* p = in + ((n * H + y) * (long)W + x) * C + c4 * 4;
#pragma unroll
                for (int q = 0; q < 4; ++q) m[q] = fmaxf(m[q], to_f32<T>(p[q]));
            }
        }
        T* o = out + ((n * Ho + ho) * (long)Wo + wo) * C + c4 * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q] = from_f32<T>(m[q]);
    }
}

// ---- stem on the matrix pipe (the f16x2 path, round 3) ------------------------------------------------------------
// The same 7x7 s2 p3 convolution + FrozenBN + ReLU in the arithmetic of conv_p32.hip: both operands as two fp16 planes of
// x * s (s an exact power of two; 22 significand bits), a product = a_h b_l + a_l b_h + a_h b_h in the f32 accumulator of
// v_mfma_f32_16x16x32_f16 -- 3 MFMAs of 16 cycles for 16 pixels x 16 channels x 32 k instead of ~1000 packed VALU FMAs.
//   GEMM view: M = output pixels, N = 64 channels, K = 7 kernel rows x (8 pixels x 4 channels): one K-step of 32 is ONE
//   kernel row, k = kw * 4 + c; kw = 7 and c = 3 carry zero weights.  The zero-bordered 4-channel f32 input is split into
//   planes once per workgroup while it is staged in LDS ([row][40 px][4 ch] fp16 per plane); an A fragment (row = output
//   pixel wo, k-chunk q = pixels 2 wo + 2 q, + 1) is then 16 contiguous bytes, and the 16 pixels of an output row read 16
//   contiguous chunks: no gather, no conflicts.  Weights arrive as planes [2][7][64][32] fp16 of w * 2^e(co) (host:
//   engine.py::stem_weight_planes), rows padded to 80 bytes in LDS (conflict-free 16-byte fragment reads).
// block = 4 waves = 16 x 16 output pixels: wave w takes output rows 4 w .. 4 w + 3 (one 16-pixel M block each) x 64 channels.
constexpr int SM_PW = 40;                        // staged input pixels per row (2 * 16 + 7 = 39, padded)
constexpr int SM_ROWS = 37;                      // 2 * 16 + 5
__global__ __launch_bounds__(256, 2) void stem_mfma_kernel(const float* __restrict__ in, const _Float16* __restrict__ wpl,
                                                           const float* __restrict__ scale, const float* __restrict__ bias,
                                                           float* __restrict__ out, int PH, int PW, float s_in) {
    __shared__ __attribute__((aligned(16))) char sx[2 * SM_ROWS * SM_PW * 8];          // input planes h | l (23.7 KB: 4+ blocks per CU)
    const int Ho = PH / 2, Wo = PW / 2, DW = PW + 8, DH = PH + 6;
    const int n = blockIdx.z, ho0 = blockIdx.y * 16, wo0 = blockIdx.x * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = lane & 15, q = lane >> 4;
    // B fragments straight from global memory (56 KB of weight planes, the same for every workgroup: L2 / L1 resident),
    // one kernel row ahead in a second register set: lane (n = lane & 15, q) reads k = 8 q .. + 7 of row kh * 64 + nb * 16 + n
    const f16x8* wh = reinterpret_cast<const f16x8*>(wpl) + (lane & 15) * 4 + q;        // + (kh * 64 + nb * 16) * 4
    constexpr int WPLANE = 7 * 64 * 4;                                                 // f16x8 units per plane
    f16x8 bh[2][4], bl[2][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        bh[0][nb] = wh[(nb * 16) * 4];
        bl[0][nb] = wh[WPLANE + (nb * 16) * 4];
    }
    // input patch: rows 2 ho0 .. + 36, pixels 2 wo0 .. + 39 of the bordered image, split into planes
    const float4* in4 = reinterpret_cast<const float4*>(in) + (long)n * DH * DW;
    for (int i = tid; i < SM_ROWS * SM_PW; i += 256) {
        const int r = i / SM_PW, c = i - r * SM_PW;
        const int gy = 2 * ho0 + r, gx = 2 * wo0 + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy < DH && gx < DW) v = in4[(long)gy * DW + gx];
        const float y[4] = {v.x * s_in, v.y * s_in, v.z * s_in, v.w * s_in};
        f16x4 h, l;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            h[c4] = (_Float16)y[c4];
            l[c4] = (_Float16)(y[c4] - (float)h[c4]);
        }
        *reinterpret_cast<f16x4*>(sx + i * 8) = h;
        *reinterpret_cast<f16x4*>(sx + SM_ROWS * SM_PW * 8 + i * 8) = l;
    }
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_off = (2 * wo + 2 * q) * 8;                    // + (input row) * SM_PW * 8
    constexpr int XL = SM_ROWS * SM_PW * 8;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        const int cur = kh & 1, nxt = cur ^ 1;
        if (kh + 1 < 7) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                bh[nxt][nb] = wh[((kh + 1) * 64 + nb * 16) * 4];
                bl[nxt][nb] = wh[WPLANE + ((kh + 1) * 64 + nb * 16) * 4];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* xp = sx + (2 * (wave * 4 + i) + kh) * (SM_PW * 8) + a_off;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(xp);
            const f16x8 al = *reinterpret_cast<const f16x8*>(xp + XL);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4 c = acc[i][nb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[cur][nb], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[cur][nb], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[cur][nb], c, 0, 0, 0);
                acc[i][nb] = c;
            }
        }
    }
    // Epilogue through LDS so that every store instruction writes whole pixels: the 16 pixels x 64 channels of an M block
    // are 4 KiB CONTIGUOUS in the NHWC output.  C layout of 16x16: column (channel) = lane & 15, row (pixel) = 4 * (lane >> 4)
    // + r; each wave passes one M block at a time through its own 16 x 68-word piece of the (now free) input buffer, then
    // lane l stores 16 bytes of pixel 4 it + (l >> 4): four instructions of 1 KiB each.  (Straight from the accumulators a
    // store instruction wrote four 64-byte pieces: 1.39 ms for 48 tiles against 1.97 ms for the VALU stem.)
    __syncthreads();                                            // every wave is done reading the input planes
    float* ep = reinterpret_cast<float*>(sx) + wave * (16 * 68);
    float sc4[4], bs4[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        sc4[nb] = scale[nb * 16 + (lane & 15)];                 // scale already carries 1 / (s_in * 2^e(co))
        bs4[nb] = bias[nb * 16 + (lane & 15)];
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int ho = ho0 + wave * 4 + i;
#pragma unroll
        for (int nb = 0; nb < 4; ++nb)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float v = acc[i][nb][r] * sc4[nb] + bs4[nb];
                ep[(4 * (lane >> 4) + r) * 68 + nb * 16 + (lane & 15)] = v > 0.f ? v : 0.f;
            }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (ho < Ho) {
            float* orow = out + (((long)n * Ho + ho) * Wo + wo0) * 64;
#pragma unroll
            for (int it = 0; it < 4; ++it) {
                const int px = 4 * it + (lane >> 4);
                const float4 v = *reinterpret_cast<const float4*>(ep + px * 68 + (lane & 15) * 4);
                if (wo0 + px < Wo) *reinterpret_cast<float4*>(orow + px * 64 + (lane & 15) * 4) = v;
            }
        }
        __builtin_amdgcn_wave_barrier();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
}

// ---- stem + max pool in ONE kernel: 7x7 s2 conv + FrozenBN + ReLU + 3x3 s2 p1 max pool, P32 planes out ---------------
// The f32 stem output (2 GB per 48 tiles) is never written: a workgroup computes a 16 x 16 tile of conv outputs that
// starts ONE row / column before an even position (conv rows 2 pi0 - 1 .. 2 pi0 + 14), leaves them in LDS and pools the
// 7 x 7 outputs whose windows lie inside the tile.  Tiles therefore overlap by two conv rows / columns (1.31 x the conv
// work, which is the cheap part).  Conv positions outside the image count as 0: every real value is >= 0 after the ReLU,
// so a 0 never changes the maximum over the valid ones (torch pads the pool with -inf).
// The conv tile goes through LDS in two HALVES of 32 channels (= the two 128-byte groups of a P32 pixel): 34.8 KB instead of
// 69.6 KB per workgroup, THREE workgroups per CU instead of two (four would cap a wave at 128 registers: spills) -- a workgroup is a chain of staging, 336 MFMAs per wave,
// tile write and pooling, and only other workgroups hide it (1.15 -> see DESIGN §8 for the measured time).
constexpr int SP_CPX = 36;                       // words per conv pixel in LDS: 32 channels + 4 pad
constexpr int SP_CROW = 16 * SP_CPX;             // words per conv row in LDS
__global__ __launch_bounds__(256, 3) void stem_pool_mfma_kernel(const float* __restrict__ in, const _Float16* __restrict__ wpl,
                                                                const float* __restrict__ scale, const float* __restrict__ bias,
                                                                char* __restrict__ out, float* __restrict__ meta, int PH, int PW,
                                                                float s_in, float s_out, int groups, int single) {
    __shared__ __attribute__((aligned(16))) char smem[16 * SP_CROW * 4];               // half a conv tile (36.9 KB); first the input planes (23.7 KB)
    char* sx = smem;
    const int Ho = PH / 2, Wo = PW / 2, Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1, DW = PW + 8, DH = PH + 6;
    const int n = blockIdx.z, pi0 = blockIdx.y * 7, pj0 = blockIdx.x * 7;
    const int hb = 2 * pi0 - 1, wb = 2 * pj0 - 1;                // conv position of tile element (0, 0)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wo = lane & 15, q = lane >> 4;
    const f16x8* wh = reinterpret_cast<const f16x8*>(wpl) + (lane & 15) * 4 + q;
    constexpr int WPLANE = 7 * 64 * 4;
    f16x8 bh[2][4], bl[2][4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        bh[0][nb] = wh[(nb * 16) * 4];
        bl[0][nb] = wh[WPLANE + (nb * 16) * 4];
    }
    // input patch: bordered rows 2 hb .. + 36, pixels 2 wb .. + 39 (negative at the top / left tiles: zeros)
    const float4* in4 = reinterpret_cast<const float4*>(in) + (long)n * DH * DW;
    for (int i = tid; i < SM_ROWS * SM_PW; i += 256) {
        const int r = i / SM_PW, c = i - r * SM_PW;
        const int gy = 2 * hb + r, gx = 2 * wb + c;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (gy >= 0 && gx >= 0 && gy < DH && gx < DW) v = in4[(long)gy * DW + gx];
        const float y[4] = {v.x * s_in, v.y * s_in, v.z * s_in, v.w * s_in};
        f16x4 h, l;
#pragma unroll
        for (int c4 = 0; c4 < 4; ++c4) {
            h[c4] = (_Float16)y[c4];
            l[c4] = (_Float16)(y[c4] - (float)h[c4]);
        }
        *reinterpret_cast<f16x4*>(sx + i * 8) = h;
        *reinterpret_cast<f16x4*>(sx + SM_ROWS * SM_PW * 8 + i * 8) = l;
    }
    __syncthreads();
    f32x4 acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const int a_off = (2 * wo + 2 * q) * 8;
    constexpr int XL = SM_ROWS * SM_PW * 8;
#pragma unroll
    for (int kh = 0; kh < 7; ++kh) {
        const int cur = kh & 1, nxt = cur ^ 1;
        if (kh + 1 < 7) {
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                bh[nxt][nb] = wh[((kh + 1) * 64 + nb * 16) * 4];
                bl[nxt][nb] = wh[WPLANE + ((kh + 1) * 64 + nb * 16) * 4];
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const char* xp = sx + (2 * (wave * 4 + i) + kh) * (SM_PW * 8) + a_off;
            const f16x8 ah = *reinterpret_cast<const f16x8*>(xp);
            const f16x8 al = *reinterpret_cast<const f16x8*>(xp + XL);
#pragma unroll
            for (int nb = 0; nb < 4; ++nb) {
                f32x4 c = acc[i][nb];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[cur][nb], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[cur][nb], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[cur][nb], c, 0, 0, 0);
                acc[i][nb] = c;
            }
        }
    }
    float* cv = reinterpret_cast<float*>(smem);
    float sc4[4], bs4[4];
#pragma unroll
    for (int nb = 0; nb < 4; ++nb) {
        sc4[nb] = scale[nb * 16 + (lane & 15)];
        bs4[nb] = bias[nb * 16 + (lane & 15)];
    }
    float vmax = 0.f;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                        // the input planes / the other half's tile are dead
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int tr = wave * 4 + i, ho = hb + tr;
            const bool row_ok = ho >= 0 && ho < Ho;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int px = 4 * (lane >> 4) + r, wx = wb + px;
                const bool ok = row_ok && wx >= 0 && wx < Wo;
#pragma unroll
                for (int nh = 0; nh < 2; ++nh) {
                    const int nb = 2 * half + nh;
                    const float v = acc[i][nb][r] * sc4[nb] + bs4[nb];
                    cv[tr * SP_CROW + px * SP_CPX + nh * 16 + (lane & 15)] = (ok && v > 0.f) ? v : 0.f;
                }
            }
        }
        __syncthreads();
        for (int it = tid; it < 49 * 4; it += 256) {
            const int pp = it >> 2, cg = it & 3;
            const int pi = pp / 7, pj = pp - pi * 7;
            const int gi = pi0 + pi, gj = pj0 + pj;
            if (gi >= Hp || gj >= Wp) continue;
            float m[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float* p = cv + (2 * pi + dy) * SP_CROW + (2 * pj + dx) * SP_CPX + cg * 8;
                    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
                    m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
                    m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
                }
            f16x8 h, l;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                vmax = fmaxf(vmax, m[c]);
                const float y = m[c] * s_out;
                h[c] = (_Float16)y;
                l[c] = single ? (_Float16)0.f : (_Float16)(y - (float)h[c]);
            }
            char* o = out + 128 + (((long)n * Hp + gi) * Wp + gj) * 256L + half * 128 + cg * 16;
            *reinterpret_cast<f16x8*>(o) = h;
            *reinterpret_cast<f16x8*>(o + 64) = l;
        }
    }
    float* slot = meta + (groups > 1 ? 2 * n : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if (lane == 0 && vmax > *reinterpret_cast<volatile const float*>(slot))
        atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(vmax));
    if (blockIdx.x == 0 && blockIdx.y == 0 && tid == 0 && (groups > 1 || n == 0)) slot[1] = s_out;
}

// maxpool 3x3 s2 p1 from f32 into P32 planes (conv_p32.hip): 8 channels per thread, 16-byte stores per plane.
// blockIdx.y = image; meta is [groups][2] with groups = 1 or one group per image.
__global__ void maxpool3x3s2_p32_kernel(const float* __restrict__ in, char* __restrict__ out, float* __restrict__ meta, float s,
                                        long per_image, int H, int W, int C, int Ho, int Wo, int groups, int single) {
    const int C8 = C / 8;
    const long n = blockIdx.y;
    float vmax = 0.f;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < per_image; i += (long)gridDim.x * blockDim.x) {
        const int c8 = (int)(i % C8);
        long t = i / C8;
        const int wo = (int)(t % Wo);
        const int ho = (int)(t / Wo);
        float m[8];
#pragma unroll
        for (int q = 0; q < 8; ++q) m[q] = -INFINITY;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int y = 2 * ho - 1 + dy;
            if ((unsigned)y >= (unsigned)H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int x = 2 * wo - 1 + dx;
                if ((unsigned)x >= (unsigned)W) continue;
                const float4* p = reinterpret_cast<const float4*>(in + ((n * H + y) * (long)W + x) * C + c8 * 8);
                const float4 a = p[0], b = p[1];
                m[0] = fmaxf(m[0], a.x); m[1] = fmaxf(m[1], a.y); m[2] = fmaxf(m[2], a.z); m[3] = fmaxf(m[3], a.w);
                m[4] = fmaxf(m[4], b.x); m[5] = fmaxf(m[5], b.y); m[6] = fmaxf(m[6], b.z); m[7] = fmaxf(m[7], b.w);
            }
        }
        f16x8 h, l;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            vmax = fmaxf(vmax, fabsf(m[q]));
            const float y = m[q] * s;
            h[q] = (_Float16)y;
            l[q] = single ? (_Float16)0.f : (_Float16)(y - (float)h[q]);
        }
        char* o = out + 128 + ((n * Ho + ho) * (long)Wo + wo) * (C * 4L) + (c8 >> 2) * 128 + (c8 & 3) * 16;
        *reinterpret_cast<f16x8*>(o) = h;
        *reinterpret_cast<f16x8*>(o + 64) = l;
    }
    float* slot = meta + (groups > 1 ? 2 * n : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
    if ((threadIdx.x & 63) == 0 && vmax > *reinterpret_cast<volatile const float*>(slot))
        atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(vmax));
    if (blockIdx.x == 0 && threadIdx.x == 0 && (groups > 1 || n == 0)) slot[1] = s;
}

template <typename T>
__global__ void subsample2_kernel(const T* __restrict__ in, T* __restrict__ out, long total, int H, int W, int C, int Ho,
                                  int Wo) {
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        long t = i / C;
        const int wo = (int)(t % Wo);
        t /= Wo;
        const int ho = (int)(t % Ho);
        const long n = t / Ho;
        out[i] = in[((n * H + 2 * ho) * (long)W + 2 * wo) * C + c];
    }
}

inline int grid_for(long total, int block) {
    long g = (total + block - 1) / block;
    return (int)(g > 16384 ? 16384 : (g < 1 ? 1 : g));
}

}  // namespace

extern "C" int demia_resize_h_u8(const uint8_t* src, uint8_t* tmp, int N, int H, int W, int newW, const int32_t* xmin,
                                 const int32_t* xsize, const int32_t* xk, int ksx, void* stream) {
    DEMIA_REQUIRE(src && tmp && xmin && xsize && xk && ksx > 0, "args");
    const long total = (long)N * H * newW;
    if (total == 0) return DEMIA_OK;
    const long smem = (((long)W * 3 + 32 + 15) & ~15L) + (long)newW * 3 + 8;
    if (smem <= 60 * 1024 && (long)N * H <= 0x7fffffffL) {                // a row (in + out) fits the default LDS window
        hipLaunchKernelGGL(resize_h_row_kernel, dim3((unsigned)((long)N * H)), dim3(256), (size_t)smem, (hipStream_t)stream, src, tmp,
                           (long)N * H * W * 3, W, newW, xmin, xsize, xk, ksx);
        DEMIA_CHECK_LAUNCH("resize_h_row_kernel");
        return DEMIA_OK;
    }
    hipLaunchKernelGGL(resize_h_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, tmp, total, W,
                       newW, xmin, xsize, xk, ksx);
    DEMIA_CHECK_LAUNCH("resize_h_kernel");
    return DEMIA_OK;
}

extern "C" int demia_resize_v_norm(const uint8_t* tmp, void* dst, int N, int H, int newW, int newH, int PH, int PW,
                                   const int32_t* ymin, const int32_t* ysize, const int32_t* yk, int ksy,
                                   const float* mean3, int dtype, void* stream) {
    DEMIA_REQUIRE(tmp && dst && ymin && ysize && yk && mean3 && ksy > 0, "args");
    DEMIA_REQUIRE(dtype == DEMIA_F32, "stem input is always f32");
    DEMIA_REQUIRE(PH >= newH && PW >= newW && PH % 32 == 0 && PW % 32 == 0, "padded size");
    const long total = (long)N * newH * newW;
    if (total == 0) return DEMIA_OK;
    if ((newW & 3) == 0 && (reinterpret_cast<unsigned long long>(tmp) & 3ull) == 0 && !getenv("DEMIA_RESIZE_PLAIN")) {
        hipLaunchKernelGGL(resize_v_norm4_kernel, dim3(grid_for(total / 4, 256)), dim3(256), 0, (hipStream_t)stream, tmp,
                           (float*)dst, total / 4, H, newW, newH, PH, PW, ymin, ysize, yk, ksy, mean3[0], mean3[1], mean3[2]);
        DEMIA_CHECK_LAUNCH("resize_v_norm4_kernel");
        return DEMIA_OK;
    }
    hipLaunchKernelGGL(resize_v_norm_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, tmp,
                       (float*)dst, total, H, newW, newH, PH, PW, ymin, ysize, yk, ksy, mean3[0], mean3[1], mean3[2]);
    DEMIA_CHECK_LAUNCH("resize_v_norm_kernel");
    return DEMIA_OK;
}

extern "C" int demia_stem_conv(const void* in, const void* w, const float* scale, const float* bias, void* mid, int N,
                               int PH, int PW, int dtype, void* stream) {
    DEMIA_REQUIRE(in && w && scale && bias && mid, "args");
    DEMIA_REQUIRE(PH % 32 == 0 && PW % 32 == 0, "padded size");
    const int Ho = PH / 2, Wo = PW / 2;
    const int smem = ST_IH * ST_IW * 16 + 7 * 7 * 4 * 64 * 4;
    dim3 grid(cdiv(Wo, ST_TW), cdiv(Ho, ST_TH), N);
    if (dtype == DEMIA_BF16) {
        auto k = stem_conv_kernel<bf16_t>;
        static bool done = false;
        if (!done) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem); done = true; }
        hipLaunchKernelGGL(k, grid, dim3(256), smem, (hipStream_t)stream, (const float*)in, (const float*)w, scale, bias,
                           (bf16_t*)mid, PH, PW);
    } else {
        auto k = stem_conv_kernel<float>;
        static bool done = false;
        if (!done) { (void)hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, smem); done = true; }
        hipLaunchKernelGGL(k, grid, dim3(256), smem, (hipStream_t)stream, (const float*)in, (const float*)w, scale, bias,
                           (float*)mid, PH, PW);
    }
    DEMIA_CHECK_LAUNCH("stem_conv_kernel");
    return DEMIA_OK;
}

extern "C" int demia_stem_conv_mfma(const float* in, const void* w_planes, const float* scale, const float* bias, float* mid, int N,
                                    int PH, int PW, float s_in, void* stream) {
    DEMIA_REQUIRE(in && w_planes && scale && bias && mid && s_in > 0.f, "args");
    DEMIA_REQUIRE(PH % 32 == 0 && PW % 32 == 0, "padded size");
    DEMIA_REQUIRE(N <= 65535, "N");
    const int Ho = PH / 2, Wo = PW / 2;
    if ((long)N * Ho * Wo == 0) return DEMIA_OK;
    hipLaunchKernelGGL(stem_mfma_kernel, dim3(cdiv(Wo, 16), cdiv(Ho, 16), N), dim3(256), 0, (hipStream_t)stream, in,
                       reinterpret_cast<const _Float16*>(w_planes), scale, bias, mid, PH, PW, s_in);
    DEMIA_CHECK_LAUNCH("stem_mfma_kernel");
    return DEMIA_OK;
}

extern "C" int demia_stem_pool_mfma(const float* in, const void* w_planes, const float* scale, const float* bias, void* out,
                                    float* out_meta, int N, int PH, int PW, float s_in, float s_out, int groups, int single, void* stream) {
    DEMIA_REQUIRE(in && w_planes && scale && bias && out && out_meta && s_in > 0.f && s_out > 0.f, "args");
    DEMIA_REQUIRE(PH % 32 == 0 && PW % 32 == 0, "padded size");
    DEMIA_REQUIRE(groups <= 1 || groups == N, "scale groups: one per image");
    DEMIA_REQUIRE(N <= 65535, "N");
    const int Hp = PH / 4, Wp = PW / 4;
    if ((long)N * Hp * Wp == 0) return DEMIA_OK;
    hipLaunchKernelGGL(stem_pool_mfma_kernel, dim3(cdiv(Wp, 7), cdiv(Hp, 7), N), dim3(256), 0, (hipStream_t)stream, in,
                       reinterpret_cast<const _Float16*>(w_planes), scale, bias, (char*)out, out_meta, PH, PW, s_in, s_out, groups,
                       single);
    DEMIA_CHECK_LAUNCH("stem_pool_mfma_kernel");
    return DEMIA_OK;
}

extern "C" int demia_maxpool3x3s2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
    DEMIA_REQUIRE(in && out && C % 4 == 0, "args");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long total = (long)N * Ho * Wo * (C / 4);
    if (total == 0) return DEMIA_OK;
    if (dtype == DEMIA_BF16)
        hipLaunchKernelGGL(maxpool3x3s2_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)in, (bf16_t*)out, total, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(maxpool3x3s2_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, total, H, W, C, Ho, Wo);
    DEMIA_CHECK_LAUNCH("maxpool3x3s2_kernel");
    return DEMIA_OK;
}

extern "C" int demia_maxpool3x3s2_p32(const float* in, void* out, float* out_meta, float s, int N, int H, int W, int C, int groups,
                                      int single, void* stream) {
    DEMIA_REQUIRE(in && out && out_meta && C % 32 == 0 && s > 0.f, "args");
    DEMIA_REQUIRE(groups <= 1 || groups == N, "scale groups: one per image");
    DEMIA_REQUIRE(N <= 65535, "N");
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long per_image = (long)Ho * Wo * (C / 8);
    if (per_image * N == 0) return DEMIA_OK;
    const int gx = (int)((grid_for(per_image * N, 256) + N - 1) / N);
    hipLaunchKernelGGL(maxpool3x3s2_p32_kernel, dim3(gx, N), dim3(256), 0, (hipStream_t)stream, in, (char*)out, out_meta, s,
                       per_image, H, W, C, Ho, Wo, groups, single);
    DEMIA_CHECK_LAUNCH("maxpool3x3s2_p32_kernel");
    return DEMIA_OK;
}

extern "C" int demia_subsample2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream) {
    DEMIA_REQUIRE(in && out, "args");
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const long total = (long)N * Ho * Wo * C;
    if (total == 0) return DEMIA_OK;
    if (dtype == DEMIA_BF16)
        hipLaunchKernelGGL(subsample2_kernel<bf16_t>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const bf16_t*)in, (bf16_t*)out, total, H, W, C, Ho, Wo);
    else
        hipLaunchKernelGGL(subsample2_kernel<float>, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream,
                           (const float*)in, (float*)out, total, H, W, C, Ho, Wo);
    DEMIA_CHECK_LAUNCH("subsample2_kernel");
    return DEMIA_OK;
}

extern "C" int demia_resize_linear_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int out_h, int out_w,
                                      const int32_t* xofs, const int16_t* ialpha, const int32_t* yofs, const int16_t* ibeta,
                                      void* stream) {
    DEMIA_REQUIRE(src && dst && xofs && ialpha && yofs && ibeta, "args");
    const long total = (long)N * out_h * out_w;
    if (total == 0) return DEMIA_OK;
    hipLaunchKernelGGL(resize_linear_u8_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, total, H, W,
                       out_h, out_w, xofs, ialpha, yofs, ibeta);
    DEMIA_CHECK_LAUNCH("resize_linear_u8_kernel");
    return DEMIA_OK;
}
