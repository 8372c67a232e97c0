// Implicit-GEMM convolution on the CDNA4 matrix cores (gfx950).
//
//   out[m, co] = act( (sum_{kh,kw,ci} in[n, ho*s+kh-p, wo*s+kw-p, ci] * w[co, kh, kw, ci]) * scale[co]
//                     + bias[co] + residual[m, co] )          m = (n, ho, wo)
//
// GEMM view: M = N*Ho*Wo output pixels (A rows, gathered on the fly from NHWC),
// N = Cout (B rows = packed weights, K-contiguous), K = KH*KW*Cin walked in steps of
// 128 bytes of one (kh, kw) tap, so a K-step of one A row is ONE contiguous 128-B run
// of the input tensor (or zeros when the tap falls into the padding).
//
// Work decomposition for 64-wide wavefronts: 256 threads = 4 waves; each wave owns
// TM x TN MFMA tiles of 32x32 (v_mfma_f32_32x32x16_bf16, or the exact-f32
// v_mfma_f32_32x32x2_f32 in the parity build), accumulators stay in registers for
// the whole K loop.  A/B tiles are register-staged into a double-buffered LDS image
// with 144-byte rows (128 B + one 16-B pad => ds_read_b128 is bank-conflict free),
// global loads for step s+1 are issued before the MFMAs of step s and written to LDS
// after them (one barrier per K-step).  The epilogue goes through LDS so that
// scale/bias/residual/activation run on, and global memory sees, 16-byte rows.
//
// Reference call sites replaced: every Conv2d/Linear/ConvTranspose2d executed by
// Detectron2 0.6 `GeneralizedRCNN.inference` under `predictor(image)`
// (reference src/functions/inference.py:1395,1398,1507,1669; src/data/models.py:107).
#include "common.h"
#ifndef DEMIA_DEV
#define DEMIA_DEV 0       // 1 (`make DEV=1`): also the kernels of the non-default precisions -- f32x3, bf16x2, f16x2r (operands split in
                          // the K loop: rounds 1-2) and plain bf16.  The product build keeps the exact-f32 MFMA kernel only: it is the
                          // control arithmetic of the parity suite (`--precision f32`) and serves the CLI's f32 mode.
#endif
#ifndef F16_BK
#define F16_BK 32      // K-step of the f16x2 kernel: 32 keeps a stage at 41 KiB and the kernel at <= 168 registers -> THREE workgroups per CU
#endif

namespace {

struct ConvP {
    const void* in;
    const void* w;
    const float* scale;
    const float* bias;
    const void* residual;
    void* out;
    int N, H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad;
    int act, res_mode, out_ld;
    int M, HoWo, ntn, nwg, ksteps, csteps, vec_ok;
    const float* amax_in;   // f16x2: upper bound of |in| (device scalar) -> power-of-two operand scale
    float* amax_out;        // any dtype: atomic max of |out| is accumulated here when non-NULL
    int w_bytes;            // split kernels: size of the tiled weight planes (buffer descriptor range)
};

template <typename T> struct Mma;
template <> struct Mma<bf16_t> {
    // one 16-byte fragment = 8 bf16 = one 32x32x16 MFMA
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a),
                                                       *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // one 16-byte fragment = 4 f32 = four 32x32x2 MFMAs.  Lane half h owns k = 4h..4h+3 of
    // each 8-wide chunk (a permutation of the k order; the sum is over all of them).
    static __device__ __forceinline__ f32x16 run(const uint4& a, const uint4& b, f32x16 c) {
        const float* af = reinterpret_cast<const float*>(&a);
        const float* bf = reinterpret_cast<const float*>(&b);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[0], bf[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[1], bf[1], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[2], bf[2], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x2f32(af[3], bf[3], c, 0, 0, 0);
        return c;
    }
};

template <typename TO> __device__ __forceinline__ void store4(TO* p, const float v[4]);
template <> __device__ __forceinline__ void store4<float>(float* p, const float v[4]) {
    *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
}
template <> __device__ __forceinline__ void store4<bf16_t>(bf16_t* p, const float v[4]) {
    bf16x4 o;
    o[0] = (bf16_t)v[0]; o[1] = (bf16_t)v[1]; o[2] = (bf16_t)v[2]; o[3] = (bf16_t)v[3];
    *reinterpret_cast<bf16x4*>(p) = o;
}
template <typename TO> __device__ __forceinline__ void load4(const TO* p, float v[4]);
template <> __device__ __forceinline__ void load4<float>(const float* p, float v[4]) {
    float4 t = *reinterpret_cast<const float4*>(p);
    v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
}
template <> __device__ __forceinline__ void load4<bf16_t>(const bf16_t* p, float v[4]) {
    bf16x4 t = *reinterpret_cast<const bf16x4*>(p);
    v[0] = (float)t[0]; v[1] = (float)t[1]; v[2] = (float)t[2]; v[3] = (float)t[3];
}

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == DEMIA_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DEMIA_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

constexpr int ROWB = 144;  // bytes per LDS tile row: 128 B of K + 16 B pad

// Epilogue shared by the conv kernels: accumulators -> LDS (f32) -> scale/bias/residual/activation on 16-byte rows.
template <typename TO, int BM, int BN, int TM, int TN, int EH = 1>
__device__ __forceinline__ void conv_epilogue(const ConvP& p, char* smem, f32x16 (&acc)[TM][TN], int a_row0, int b_row0, int m0, int n0,
                                              float post = 1.0f) {
    constexpr int EROW = BN * 4 + 16;          // epilogue LDS row (f32) + pad
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    constexpr int G = BN / 4;          // 4-channel groups per row
    constexpr int RPP = 256 / G;       // rows per pass
    constexpr int NR = BM / RPP;       // passes
    const int g = tid % G;
    const int r0 = tid / G;
    const int co = n0 + g * 4;
    const TO* __restrict__ res = reinterpret_cast<const TO*>(p.residual);
    // The residual rows of this thread are requested BEFORE the accumulators go through LDS: a short-K layer
    // (1x1, K <= 256) is otherwise one exposed global round trip per pass.
    const bool res_pref = p.res_mode != DEMIA_RES_NONE && p.vec_ok && co < p.Cout;
    float rv[NR][4];
    if (res_pref) {
#pragma unroll
        for (int k = 0; k < NR; ++k) {
            const int m = m0 + r0 + k * RPP;
            const int mm = m < p.M ? m : p.M - 1;
            long ridx;
            if (p.res_mode == DEMIA_RES_SAME) {
                ridx = (long)mm * p.Cout + co;
            } else {
                const int n = mm / p.HoWo;
                const int rem = mm - n * p.HoWo;
                const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                const int Hr = (p.Ho + 1) >> 1, Wr = (p.Wo + 1) >> 1;
                ridx = (((long)n * Hr + (ho >> 1)) * Wr + (wo >> 1)) * p.Cout + co;
            }
            load4<TO>(res + ridx, rv[k]);
        }
    }
    // ---- epilogue: accumulators -> LDS (f32) -> 16-byte rows, in EH passes of BM / EH rows (EH = 2 keeps the LDS
    // footprint of a 128-row tile at 33 KiB, below its K-loop stage: the epilogue must not cost a workgroup per CU) ----
    // C layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    constexpr int HR = BM / EH, NRH = NR / EH;
    static_assert(NR % EH == 0 && HR % 32 == 0, "epilogue halves");
    float vmax = 0.f;
    TO* __restrict__ out = reinterpret_cast<TO*>(p.out);
    float sc[4], bs[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const bool ok = (co + q) < p.Cout;
        sc[q] = (p.scale && ok) ? p.scale[co + q] : 1.0f;
        bs[q] = (p.bias && ok) ? p.bias[co + q] : 0.0f;
    }
#pragma unroll
    for (int h = 0; h < EH; ++h) {
        __syncthreads();
        {
            float* e = reinterpret_cast<float*>(smem);
            constexpr int EF = EROW / 4;
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int rb = a_row0 + i * 32 - h * HR;          // wave-uniform
                if (rb >= 0 && rb < HR) {
#pragma unroll
                    for (int j = 0; j < TN; ++j)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int row = rb + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                            const int col = b_row0 + j * 32 + (lane & 31);
                            e[row * EF + col] = acc[i][j][r] * post;
                        }
                }
            }
        }
        __syncthreads();
        if (co < p.Cout) {
            const char* e = smem;
#pragma unroll
            for (int kk = 0; kk < NRH; ++kk) {
                const int k = h * NRH + kk;
                const int r = r0 + k * RPP;
                const int m = m0 + r;
                if (m < p.M) {
                    const float4 a4 = *reinterpret_cast<const float4*>(e + (r - h * HR) * EROW + g * 16);
                    float v[4] = {a4.x, a4.y, a4.z, a4.w};
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = v[q] * sc[q] + bs[q];
                    if (res_pref) {
#pragma unroll
                        for (int q = 0; q < 4; ++q) v[q] += rv[k][q];
                    } else if (p.res_mode != DEMIA_RES_NONE) {
                        long ridx;
                        if (p.res_mode == DEMIA_RES_SAME) {
                            ridx = (long)m * p.Cout + co;
                        } else {
                            const int n = m / p.HoWo;
                            const int rem = m - n * p.HoWo;
                            const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
                            const int Hr = (p.Ho + 1) >> 1, Wr = (p.Wo + 1) >> 1;
                            ridx = (((long)n * Hr + (ho >> 1)) * Wr + (wo >> 1)) * p.Cout + co;
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (co + q < p.Cout) v[q] += to_f32<TO>(res[ridx + q]);
                    }
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        v[q] = apply_act(v[q], p.act);
                        if (co + q < p.Cout) vmax = fmaxf(vmax, fabsf(v[q]));
                    }
                    TO* o = out + (long)m * p.out_ld + co;
                    if (p.vec_ok) {
                        store4<TO>(o, v);
                    } else {
#pragma unroll
                        for (int q = 0; q < 4; ++q)
                            if (co + q < p.Cout) o[q] = from_f32<TO>(v[q]);
                    }
                }
            }
        }
    }
    if (p.amax_out) {
        // |out| bound for the next layer's operand scale: wave max, one atomic per wave (non-negative floats order
        // like their bit patterns)
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, off));
        // the bound only grows, so a (possibly stale) read that already covers this wave's maximum makes the atomic
        // unnecessary: all but the first few waves of a launch skip it (40 000 atomics on one word cost more than a
        // short-K layer's whole launch)
        if (lane == 0 && vmax > *reinterpret_cast<volatile const float*>(p.amax_out))
            atomicMax(reinterpret_cast<unsigned int*>(p.amax_out), __float_as_uint(vmax));
    }
}

template <typename T, typename TO, int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_igemm_kernel(const ConvP p) {
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int EPV = 16 / (int)sizeof(T);   // elements per 16-B vector
    constexpr int BK = 128 / (int)sizeof(T);   // elements per K-step
    constexpr int AV = BM / 32;                // A vectors per thread per K-step
    constexpr int BV = BN / 32;                // B vectors per thread per K-step
    constexpr int STAGE = (BM + BN) * ROWB;
    constexpr int EROW = BN * 4 + 16;          // epilogue LDS row (f32) + pad
    static_assert(BM * EROW <= 2 * STAGE, "epilogue staging must fit in the tile buffers");

    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    const int swz = xcd_remap(blockIdx.x, p.nwg);
    const int tile_n = swz % p.ntn;
    const int tile_m = swz / p.ntn;
    const int m0 = tile_m * BM;
    const int n0 = tile_n * BN;

    const T* __restrict__ in = reinterpret_cast<const T*>(p.in);
    const T* __restrict__ wt = reinterpret_cast<const T*>(p.w);

    // ---- per-thread gather bookkeeping --------------------------------------------------
    const int chunk = tid & 7;
    const int lrow = tid >> 3;  // 0..31
    long a_base[AV];
    int a_hi0[AV], a_wi0[AV];
    bool a_vm[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int m = m0 + lrow + 32 * i;
        a_vm[i] = m < p.M;
        const int mm = a_vm[i] ? m : 0;
        const int n = mm / p.HoWo;
        const int rem = mm - n * p.HoWo;
        const int ho = rem / p.Wo;
        const int wo = rem - ho * p.Wo;
        a_hi0[i] = ho * p.stride - p.pad;
        a_wi0[i] = wo * p.stride - p.pad;
        a_base[i] = (((long)n * p.H + a_hi0[i]) * p.W + a_wi0[i]) * p.Cin + chunk * EPV;
    }
    const long Ktot = (long)p.KH * p.KW * p.Cin;
    long b_off[BV];
    bool b_vm[BV];
#pragma unroll
    for (int i = 0; i < BV; ++i) {
        const int co = n0 + lrow + 32 * i;
        b_vm[i] = co < p.CoutPad;
        b_off[i] = (long)(b_vm[i] ? co : 0) * Ktot + chunk * EPV;
    }

    uint4 ra[AV], rb[BV];
    int kh = 0, kw = 0, c0 = 0;  // tap / channel offset of the step being LOADED

    auto load_step = [&]() {
        const long tap = ((long)kh * p.W + kw) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int hi = a_hi0[i] + kh, wi = a_wi0[i] + kw;
            const bool ok = a_vm[i] && (unsigned)hi < (unsigned)p.H && (unsigned)wi < (unsigned)p.W;
            ra[i] = ok ? *reinterpret_cast<const uint4*>(in + a_base[i] + tap) : make_uint4(0, 0, 0, 0);
        }
        const long kof = ((long)kh * p.KW + kw) * p.Cin + c0;
#pragma unroll
        for (int i = 0; i < BV; ++i)
            rb[i] = b_vm[i] ? *reinterpret_cast<const uint4*>(wt + b_off[i] + kof) : make_uint4(0, 0, 0, 0);
        c0 += BK;
        if (c0 >= p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };
    auto store_step = [&](int buf) {
        char* sA = smem + buf * STAGE;
        char* sB = sA + BM * ROWB;
#pragma unroll
        for (int i = 0; i < AV; ++i)
            *reinterpret_cast<uint4*>(sA + (lrow + 32 * i) * ROWB + chunk * 16) = ra[i];
#pragma unroll
        for (int i = 0; i < BV; ++i)
            *reinterpret_cast<uint4*>(sB + (lrow + 32 * i) * ROWB + chunk * 16) = rb[i];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * ROWB + (lane >> 5) * 16;
    const int a_row0 = wm * TM * 32;
    const int b_row0 = wn * TN * 32;

    load_step();
    store_step(0);
    __syncthreads();

    for (int s = 0; s < p.ksteps; ++s) {
        const int buf = s & 1;
        const bool more = (s + 1) < p.ksteps;
        if (more) load_step();
        const char* sA = smem + buf * STAGE + a_row0 * ROWB + frag_off;
        const char* sB = smem + buf * STAGE + BM * ROWB + b_row0 * ROWB + frag_off;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            uint4 fa[TM], fb[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) fa[i] = *reinterpret_cast<const uint4*>(sA + i * 32 * ROWB + kk * 32);
#pragma unroll
            for (int j = 0; j < TN; ++j) fb[j] = *reinterpret_cast<const uint4*>(sB + j * 32 * ROWB + kk * 32);
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[i][j] = Mma<T>::run(fa[i], fb[j], acc[i][j]);
        }
        if (more) store_step(buf ^ 1);
        __syncthreads();
    }

    conv_epilogue<TO, BM, BN, TM, TN>(p, smem, acc, a_row0, b_row0, m0, n0);
}

// ---------------------------------------------------------------------------------------------------
// f32 convolution on the bf16 matrix pipe ("f32x3"): every f32 operand is split into three bf16 planes
//   x = x1 + x2 + x3,  x1 = bf16(x), x2 = bf16(x - x1), x3 = bf16(x - x1 - x2)      (3 x 8 = 24 mantissa bits)
// and a*b is taken as the six products with i + j <= 4 (a1b1, a1b2, a2b1, a1b3, a2b2, a3b1), each exact in the
// f32 accumulator of v_mfma_f32_32x32x16_bf16; the dropped terms are <= 3 * 2^-24 |ab|, the size of one f32
// rounding.  Six bf16 MFMAs (6 x 32 cycles per 32x32x16) replace eight f32 MFMAs (8 x 64 cycles), so the matrix
// pipe does the same f32 dot product 2.7x faster.  Activations stay f32 in HBM (nothing else changes): the A tile
// is split in registers on its way to LDS; weights are split and tiled once on the host
// ([CoutPad / 64][ksteps][3][64][32] bf16, see the B loads below).
// K-step = 32 elements; LDS rows are 64 B + 16 B pad (80 B: conflict-free ds_read_b128); 2 stages x 3 planes.

__device__ __forceinline__ void split3(float x, bf16_t& h, bf16_t& m, bf16_t& l) {
    h = (bf16_t)x;
    const float r1 = x - (float)h;
    m = (bf16_t)r1;
    const float r2 = r1 - (float)m;
    l = (bf16_t)r2;
}

template <bool F16>
__device__ __forceinline__ f32x16 mma_planes(const uint4& a, const uint4& b, f32x16 c) {
    if (F16) return __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const f16x8*>(&a), *reinterpret_cast<const f16x8*>(&b), c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(*reinterpret_cast<const bf16x8*>(&a), *reinterpret_cast<const bf16x8*>(&b), c, 0, 0, 0);
}

// F16 ("f16x2", NP = 2): the planes are fp16 -- x = x1 + x2 with x1 = half(x * s), x2 = half(x * s - x1): 2 x 11 = 22
// significand bits, representation error <= 2^-22 |x| and the dropped x2 * y2 <= 2^-22 |xy|, i.e. an f32-sized error
// from THREE MFMAs per product.  fp16 has 5 exponent bits, so both operands are brought into range by exact
// power-of-two scales: the weights per output channel on the host (folded into `scale`), the activations per tensor
// with s = 2^(13 - ilogb(amax_in)), amax_in being the |.| bound that the producing layer's epilogue accumulated
// (elements below 2^-16 of the tensor maximum have a denormal low plane: absolute error <= 2^-38 of the maximum).
// NP = 3: "f32x3" as above.  NP = 2 ("bf16x2"): two planes and the three products a1b1, a1b2, a2b1 -- 16 significand
// bits per operand, dropped terms <= 3 * 2^-16 |ab|: 256 x less exact than f32, 256 x more exact than plain bf16, at half
// the matrix work of f32x3 (an opt-in speed mode; the default stays f32x3).
template <typename TO, int WM, int WN, int TM, int TN, int NP = 3, bool F16 = false>
__global__ __launch_bounds__(256, F16 ? 3 : 2) void conv_igemm_split_kernel(const ConvP p) {
    static_assert(!F16 || NP == 2, "fp16 planes come in pairs");
    constexpr int BM = WM * TM * 32;
    constexpr int BN = WN * TN * 32;
    constexpr int BK = F16 ? F16_BK : 32;
    constexpr int ROWS = BK * 2 + 16;           // LDS row: BK 2-byte elements + 16 B pad (80 / 144 B: conflict-free ds_read_b128)
    constexpr bool RING = F16 && BK == 16;      // two LDS stages of K-step 16, ONE barrier per step (see the loop below)
    constexpr int EPT = F16 ? (BM * BK / 256 >= 8 ? 8 : 4) : 4;   // A elements per thread per row pass (f16x2: 16-byte LDS writes where the tile allows)
    constexpr int CH = BK / EPT;                // thread chunks per A row
    constexpr int RPP = 256 / CH;               // A rows per pass of the 256 threads
    constexpr int AV = BM / RPP;                // row passes per K-step (EPT / 4 f32 vectors each)
    constexpr int BVR = BK / 8;                 // 16-B plane vectors (8 elements) per B row
    constexpr int BVT = NP * BN * BVR / 256;    // ... of the B planes per thread per K-step
    constexpr int PLANE_A = BM * ROWS, PLANE_B = BN * ROWS;
    constexpr int STAGE_BYTES = NP * (BM + BN) * ROWS;
    static_assert((NP * BN * BVR) % 256 == 0, "B tile must divide over the 256 threads");

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int swz = xcd_remap(blockIdx.x, p.nwg);
    const int tile_n = swz % p.ntn, tile_m = swz / p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    float a_scale = 1.f, post = 1.f;
    if (F16 && p.amax_in) {
        const float am = *p.amax_in;
        if (am > 0.f && am < 3.0e38f) {
            int e = 13 - ilogbf(am);
            e = e < -60 ? -60 : (e > 60 ? 60 : e);
            a_scale = ldexpf(1.f, e);
            post = ldexpf(1.f, -e);
        }
    }
    // Operands are fetched with buffer loads: a 32-bit byte offset per lane, the hardware range check returns zeros for
    // OOB (= padding taps, rows beyond M, channels beyond CoutPad) -- no 64-bit address arithmetic and no exec-mask
    // branches in the K loop.  The A descriptor starts at the first image this tile touches, so offsets stay small
    // whatever the batch; bit t of a_mask says whether tap t = kh * KW + kw of that row lies inside the image.
    constexpr int OOB = (int)0x80000000;
    const int n_first = m0 / p.HoWo;
    const long img_elems = (long)p.H * p.W * p.Cin;
    long a_rec = ((long)p.N - n_first) * img_elems * 4;
    a_rec = a_rec > 0x7fffffffL ? 0x7fffffffL : a_rec;
    const __amdgpu_buffer_rsrc_t rsrc_a = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(reinterpret_cast<const float*>(p.in) + n_first * img_elems), 0, (int)a_rec, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_b = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, p.w_bytes, 0x00020000);
    const int chunk = tid % CH, lrow = tid / CH;
    int a_off[AV];
    unsigned a_mask[AV];
#pragma unroll
    for (int i = 0; i < AV; ++i) {
        const int m = m0 + lrow + RPP * i;
        const bool vm = m < p.M;
        const int mm = vm ? m : 0;
        const int n = mm / p.HoWo;
        const int rem = mm - n * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        a_off[i] = (int)(((((long)(n - n_first) * p.H + hi0) * p.W + wi0) * p.Cin + chunk * EPT) * 4);
        unsigned mk = 0;
        for (int t = 0; t < p.KH * p.KW; ++t) {
            const int th = t / p.KW, tw = t - th * p.KW;
            if (vm && (unsigned)(hi0 + th) < (unsigned)p.H && (unsigned)(wi0 + tw) < (unsigned)p.W) mk |= 1u << t;
        }
        a_mask[i] = mk;
    }
    // Weight planes arrive TILED: [CoutPad / 64][ksteps][NP][64 rows][BK k] 2-byte elements -- the 64 x BK piece of one
    // plane that a K-step needs is 4 KiB contiguous, so the B loads of a wave are whole 128-byte lines (row-major
    // [CoutPad][K] planes made every 16-lane group touch four half-used lines).
    int b_voff[BVT];
    int b_lds[BVT];
#pragma unroll
    for (int j = 0; j < BVT; ++j) {
        const int v = tid + 256 * j;
        const int plane = v / (BN * BVR), rem = v - plane * (BN * BVR);
        const int row = rem / BVR, c8 = rem % BVR;
        const int co = n0 + row;
        const int n64 = co >> 6;
        b_voff[j] = co < p.CoutPad ? (int)((((long)n64 * p.ksteps * NP + plane) * (64 * BK) + (co & 63) * BK + c8 * 8) * 2) : OOB;
        b_lds[j] = NP * PLANE_A + plane * PLANE_B + row * ROWS + c8 * 16;
    }
    int b_soff = 0;

    // ONE LDS stage (61 KiB for 128x128) so that TWO workgroups share a CU: while one is in its barrier / split /
    // store phase the other one's MFMAs keep the matrix pipe busy (a K-step is only 48 MFMAs, ~1.5k cycles, far
    // too short to hide an HBM round trip behind a single wave per SIMD).  The next tile waits in registers.
    uint4 ra[AV][EPT / 4], rb[BVT];
    int kh = 0, kw = 0, c0 = 0;
    auto load_step = [&]() {
        const int tap_b = ((kh * p.W + kw) * p.Cin + c0) * 4;
        const unsigned bit = 1u << (kh * p.KW + kw);
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const int vo = (a_mask[i] & bit) ? a_off[i] + tap_b : OOB;
#pragma unroll
            for (int v = 0; v < EPT / 4; ++v) {
                const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, vo + 16 * v, 0, 0);
                ra[i][v] = make_uint4(t[0], t[1], t[2], t[3]);
            }
        }
#pragma unroll
        for (int j = 0; j < BVT; ++j) {
            const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rsrc_b, b_voff[j], b_soff, 0);
            rb[j] = make_uint4(t[0], t[1], t[2], t[3]);
        }
        b_soff += NP * 64 * BK * 2;
        c0 += BK;
        if (c0 >= p.Cin) { c0 = 0; if (++kw == p.KW) { kw = 0; ++kh; } }
    };
    auto store_step = [&](int st) {
        char* sbase = smem + st * STAGE_BYTES;
#pragma unroll
        for (int i = 0; i < AV; ++i) {
            const float* f = reinterpret_cast<const float*>(&ra[i][0]);
            char* dst = sbase + (lrow + RPP * i) * ROWS + chunk * (EPT * 2);
            if (F16 && EPT == 8) {
                f16x8 h, l;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const float x = f[q] * a_scale;
                    h[q] = (_Float16)x;
                    l[q] = (_Float16)(x - (float)h[q]);
                }
                *reinterpret_cast<f16x8*>(dst) = h;
                *reinterpret_cast<f16x8*>(dst + PLANE_A) = l;
            } else if (F16) {
                f16x4 h, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float x = f[q] * a_scale;
                    h[q] = (_Float16)x;
                    l[q] = (_Float16)(x - (float)h[q]);
                }
                *reinterpret_cast<f16x4*>(dst) = h;
                *reinterpret_cast<f16x4*>(dst + PLANE_A) = l;
            } else {
                bf16x4 h, m, l;
#pragma unroll
                for (int q = 0; q < 4; ++q) { bf16_t a, b, c; split3(f[q], a, b, c); h[q] = a; m[q] = b; l[q] = c; }
                *reinterpret_cast<bf16x4*>(dst) = h;
                *reinterpret_cast<bf16x4*>(dst + PLANE_A) = m;
                if (NP == 3) *reinterpret_cast<bf16x4*>(dst + 2 * PLANE_A) = l;
            }
        }
#pragma unroll
        for (int j = 0; j < BVT; ++j) *reinterpret_cast<uint4*>(sbase + b_lds[j]) = rb[j];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * ROWS + (lane >> 5) * 16;
    const int a_row0 = wm * TM * 32, b_row0 = wn * TN * 32;
    const char* sA = smem + a_row0 * ROWS + frag_off;
    const char* sB = smem + NP * PLANE_A + b_row0 * ROWS + frag_off;

    auto mfma_step = [&](int st) {
        const char* cA = sA + st * STAGE_BYTES;
        const char* cB = sB + st * STAGE_BYTES;
#pragma unroll
        for (int kk = 0; kk < BK / 16; ++kk) {
            uint4 fa[NP][TM], fb[NP][TN];
#pragma unroll
            for (int q = 0; q < NP; ++q) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[q][i] = *reinterpret_cast<const uint4*>(cA + q * PLANE_A + i * 32 * ROWS + kk * 32);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[q][j] = *reinterpret_cast<const uint4*>(cB + q * PLANE_B + j * 32 * ROWS + kk * 32);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];
                    if (NP == 3) {
                        c = mma_planes<F16>(fa[0][i], fb[NP - 1][j], c);   // smallest terms first
                        c = mma_planes<F16>(fa[1][i], fb[1][j], c);
                        c = mma_planes<F16>(fa[NP - 1][i], fb[0][j], c);
                    }
                    c = mma_planes<F16>(fa[0][i], fb[1][j], c);
                    c = mma_planes<F16>(fa[1][i], fb[0][j], c);
                    c = mma_planes<F16>(fa[0][i], fb[0][j], c);
                    acc[i][j] = c;
                }
        }
    };
    if (RING) {
        // Two LDS stages of one MFMA K-step each.  Step s: write the registers (step s + 1) into the other stage, request
        // step s + 2, then this wave's fragment reads + MFMAs on the current stage -- ONE barrier per step, and no phase
        // in which every wave of the workgroup is outside its MFMAs: a wave that has written its share goes straight on
        // to the matrix pipe while its neighbours are still converting.  (The stage being written was last read before
        // the previous step's barrier; the stage being read was written before it.)
        load_step();
        store_step(0);
        if (p.ksteps > 1) load_step();
        __syncthreads();
        for (int s = 0; s < p.ksteps; ++s) {
            const int cur = s & 1;
            if (s + 1 < p.ksteps) store_step(cur ^ 1);
            if (s + 2 < p.ksteps) load_step();
            mfma_step(cur);
            __syncthreads();
        }
    } else {
        load_step();
        store_step(0);
        __syncthreads();
        for (int s = 0; s < p.ksteps; ++s) {
            const bool more = (s + 1) < p.ksteps;
            if (more) load_step();
            mfma_step(0);
            __syncthreads();                 // every wave is done reading this K-step
            if (more) store_step(0);
            __syncthreads();
        }
    }
    conv_epilogue<TO, BM, BN, TM, TN, (BM == 128 ? 2 : 1)>(p, smem, acc, a_row0, b_row0, m0, n0, post);
}

// 128 x 128 or 64 x 128 tiles for the split kernels: the workgroups of a CU share its matrix pipes, so a launch lasts as
// long as the CU with the most tiles -- balance = (tiles / 256) / ceil(tiles / 256).  The half tile re-reads the weight
// planes twice as often per FLOP and measures 7 % slower per tile at equal balance (164 vs 172 TFLOP/s on the large
// layers), so it is taken only where its better balance outweighs that (res4 / res5 at 16 tiles per batch).
inline bool prefer_half_tile(long blocks128, long blocks64, double half_tile_rate = 0.93) {
    auto balance = [](long b) { const double per_cu = (double)b / 256.0; return per_cu / (double)((b + 255) / 256); };
    if (blocks128 < 512) return true;          // fewer than two tiles per CU: fill the chip first
    return half_tile_rate * balance(blocks64) > balance(blocks128);
}

template <typename TO, int WM, int WN, int TM, int TN, int NP = 3, bool F16 = false>
int launch_split_cfg(ConvP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int stage = NP * (BM + BN) * ((F16 ? F16_BK : 32) * 2 + 16) * ((F16 && F16_BK == 16) ? 2 : 1), epi = (BM == 128 ? 64 : BM) * (BN * 4 + 16);
    constexpr int smem = stage > epi ? stage : epi;      // one tile stage, re-used by the epilogue
    p.ntn = cdiv(p.CoutPad, BN);
    p.nwg = p.ntn * cdiv(p.M, BM);
    auto k = conv_igemm_split_kernel<TO, WM, WN, TM, TN, NP, F16>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(p.nwg), dim3(256), smem, st, p);
    DEMIA_CHECK_LAUNCH("conv_igemm_split_kernel");
    return DEMIA_OK;
}

template <typename T, typename TO, int WM, int WN, int TM, int TN>
int launch_cfg(ConvP p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int smem = 2 * (BM + BN) * ROWB;
    p.ntn = cdiv(p.CoutPad, BN);
    const int ntm = cdiv(p.M, BM);
    p.nwg = p.ntn * ntm;
    auto k = conv_igemm_kernel<T, TO, WM, WN, TM, TN>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(p.nwg), dim3(256), smem, st, p);
    DEMIA_CHECK_LAUNCH("conv_igemm_kernel");
    return DEMIA_OK;
}

template <typename T, typename TO>
int launch_typed(ConvP p, int bn, hipStream_t st) {
    if (bn == 128) {
        // 128x128 tiles need >= 2 workgroups per CU to keep one MFMA pipe busy while the other wave waits on
        // LDS / global loads; mid-size layers (res4 / res5 at small batch) get 64x128 tiles to double the grid
        const long blocks128 = (long)cdiv(p.M, 128) * cdiv(p.CoutPad, 128);
        if (blocks128 < 3 * 256) return launch_cfg<T, TO, 2, 2, 1, 2>(p, st);
        return launch_cfg<T, TO, 2, 2, 2, 2>(p, st);
    }
    if (bn == 64) return launch_cfg<T, TO, 4, 1, 1, 2>(p, st);
    return launch_cfg<T, TO, 4, 1, 1, 1>(p, st);
}

}  // namespace

extern "C" int demia_conv_f16x2_kstep(void) { return F16_BK; }

extern "C" int demia_conv2d_nhwc(const demia_conv_desc* d, void* stream) {
    DEMIA_REQUIRE(d && d->in && d->w && d->out, "null pointer");
    DEMIA_REQUIRE(d->dtype == DEMIA_F32 || d->dtype == DEMIA_BF16 || d->dtype == DEMIA_F32X3 || d->dtype == DEMIA_BF16X2 ||
                  d->dtype == DEMIA_F16X2, "dtype");
    DEMIA_REQUIRE(d->out_dtype == DEMIA_F32 || d->out_dtype == DEMIA_BF16, "out_dtype");
    const int bk = d->dtype == DEMIA_BF16 ? 64 : (d->dtype == DEMIA_F16X2 ? F16_BK : 32);
    DEMIA_REQUIRE((d->dtype != DEMIA_F32X3 && d->dtype != DEMIA_BF16X2 && d->dtype != DEMIA_F16X2) ||
                  (d->CoutPad % 64 == 0 && d->out_dtype == DEMIA_F32), "f32x3 / bf16x2 / f16x2 need CoutPad % 64 == 0, f32 output");
    DEMIA_REQUIRE(d->Cin > 0 && d->Cin % bk == 0, "Cin must be a multiple of 64 (bf16, f16x2) / 32 (f32, f32x3, bf16x2)");
    DEMIA_REQUIRE(d->CoutPad >= d->Cout && d->CoutPad % 32 == 0, "CoutPad");
    DEMIA_REQUIRE(d->KH > 0 && d->KW > 0 && d->stride > 0 && d->pad >= 0, "kernel geometry");
    DEMIA_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1, "Ho");
    DEMIA_REQUIRE(d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1, "Wo");
    DEMIA_REQUIRE(d->res_mode == DEMIA_RES_NONE || d->residual, "residual pointer");
    DEMIA_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31), "M overflow");
    ConvP p;
    p.in = d->in; p.w = d->w; p.scale = d->scale; p.bias = d->bias; p.residual = d->residual; p.out = d->out;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo;
    p.Cout = d->Cout; p.CoutPad = d->CoutPad; p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.act = d->act; p.res_mode = d->res_mode;
    p.out_ld = d->out_ld > 0 ? d->out_ld : d->Cout;
    DEMIA_REQUIRE(p.out_ld >= d->Cout, "out_ld");
    p.M = d->N * d->Ho * d->Wo;
    p.HoWo = d->Ho * d->Wo;
    p.csteps = d->Cin / bk;
    p.ksteps = d->KH * d->KW * p.csteps;
    p.vec_ok = (d->Cout % 4 == 0 && p.out_ld % 4 == 0) ? 1 : 0;
    p.ntn = p.nwg = 0;
    if (p.M == 0) return DEMIA_OK;
    int bn = d->tile_hint;
    if (bn != 128 && bn != 64 && bn != 32) bn = d->CoutPad >= 128 ? 128 : (d->CoutPad >= 64 ? 64 : 32);
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    p.amax_in = d->amax_in; p.amax_out = d->amax_out;
    p.w_bytes = 0;
    if (d->dtype == DEMIA_F16X2 || d->dtype == DEMIA_BF16X2 || d->dtype == DEMIA_F32X3) {
        const long np = d->dtype == DEMIA_F32X3 ? 3 : 2;
        const long wb = np * d->CoutPad * (long)d->KH * d->KW * d->Cin * 2;
        DEMIA_REQUIRE(wb < (1L << 31), "weight planes must stay below 2 GiB");
        DEMIA_REQUIRE(d->KH * d->KW <= 32, "at most 32 taps");
        DEMIA_REQUIRE((long)d->H * d->W * d->Cin * 4 * 2 < (1L << 30), "one image must stay below 512 MiB");
        p.w_bytes = (int)wb;
    }
#if DEMIA_DEV
    if (d->dtype == DEMIA_F16X2) {
        if (d->CoutPad % 128 != 0) return launch_split_cfg<float, 4, 1, 1, 2, 2, true>(p, st);
        const long blocks128 = (long)cdiv(p.M, 128) * cdiv(p.CoutPad, 128);
        // f16x2: the 64 x 128 tile runs at 0.74 of the 128 x 128 tile's rate (233 vs 322 TFLOP/s at equal balance)
        if (prefer_half_tile(blocks128, (long)cdiv(p.M, 64) * cdiv(p.CoutPad, 128), 0.75))
            return launch_split_cfg<float, 2, 2, 1, 2, 2, true>(p, st);
        return launch_split_cfg<float, 2, 2, 2, 2, 2, true>(p, st);
    }
    if (d->dtype == DEMIA_BF16X2) {
        if (d->CoutPad % 128 != 0) return launch_split_cfg<float, 4, 1, 1, 2, 2>(p, st);
        const long blocks128 = (long)cdiv(p.M, 128) * cdiv(p.CoutPad, 128);
        if (prefer_half_tile(blocks128, (long)cdiv(p.M, 64) * cdiv(p.CoutPad, 128)))
            return launch_split_cfg<float, 2, 2, 1, 2, 2>(p, st);
        return launch_split_cfg<float, 2, 2, 2, 2, 2>(p, st);
    }
    if (d->dtype == DEMIA_F32X3) {
        if (d->CoutPad % 128 != 0) return launch_split_cfg<float, 4, 1, 1, 2>(p, st);            // 128 x 64
        const long blocks128 = (long)cdiv(p.M, 128) * cdiv(p.CoutPad, 128);
        if (prefer_half_tile(blocks128, (long)cdiv(p.M, 64) * cdiv(p.CoutPad, 128)))
            return launch_split_cfg<float, 2, 2, 1, 2>(p, st);                                    // 64 x 128
        return launch_split_cfg<float, 2, 2, 2, 2>(p, st);                                        // 128 x 128
    }
    if (d->dtype == DEMIA_BF16) {
        if (d->out_dtype == DEMIA_BF16) return launch_typed<bf16_t, bf16_t>(p, bn, st);
        return launch_typed<bf16_t, float>(p, bn, st);
    }
#else
    DEMIA_REQUIRE(d->dtype == DEMIA_F32, "product build: demia_conv2d_nhwc computes exact f32 only (f32x3 / bf16x2 / f16x2r / bf16 need `make DEV=1`)");
#endif
    DEMIA_REQUIRE(d->out_dtype == DEMIA_F32, "f32 input requires f32 output");
    return launch_typed<float, float>(p, bn, st);
}
