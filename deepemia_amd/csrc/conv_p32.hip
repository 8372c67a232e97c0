// f32-accurate convolution on the fp16 matrix pipe with BOTH operands pre-split in HBM ("P32" activations).
//
//   out[m, co] = act( (sum_k in[pixel(m, tap), ci] * w[co, tap, ci]) * scale[co] + bias[co] + residual[m, co] )
//
// Same arithmetic as the f16x2 mode of conv_mfma.hip -- every f32 value x travels as two fp16 planes of x * s
// (s an exact power of two), x * s = h + l, h = half(x s), l = half(x s - h): 22 significand bits; a product is the
// three MFMAs a_h b_l + a_l b_h + a_h b_h in the f32 accumulator of v_mfma_f32_32x32x16_f16, error <= 3 * 2^-22 |ab| --
// but the split happens ONCE, in the epilogue of the layer that produces the tensor, not in every consumer's K loop:
//
//   P32 activation buffer = 128 zero bytes, then pixels [M][C / 32][2][32] fp16: per pixel and 32-channel group one
//   128-byte line = 32 high halves, 32 low halves.  A K-step (32 channels of one tap) of one GEMM row is exactly one
//   line, so the A tile is a pure row gather.  Padding taps and rows beyond M gather the zero header.
//   meta = {amax, s} per tensor (device floats): `amax` is the measured max |x| (atomic max in the producer's
//   epilogue, zeroed by the host before the forward), `s` the power of two the planes were scaled with.  The producer
//   cannot know its own amax before it has written the tensor, so s comes from an a-priori bound instead:
//   |out| <= amax_in * max_co(|scale_co| * sum_k |w_co,k|) + max |bias| (+ amax_residual), s = 2^(14 - ilogb(bound)),
//   i.e. |x s| < 2^15.  The bound is loose by the usual sqrt(K)-ish factor, which only moves the point where the LOW
//   plane goes denormal (absolute error <= 2^-39 of the bound); it never compounds because every layer starts again
//   from the MEASURED amax of its input.
//
// Kernel shape (CDNA4, 64-wide waves): one workgroup of 8 waves (512 threads, <= 256 VGPRs) per CU computes a
// BM x BN tile, BM = WM * TM * 32, BN = WN * TN * 32 (256 x 256 for the big layers), accumulators in registers.
// Both operands go global -> LDS by LDS-DMA (buffer_load_dwordx4 ... lds, 1 KiB per wave instruction, no staging
// registers, no ds_write); the LDS image is linear (DMA writes lane-linear) and bank conflicts of the ds_read_b128
// fragment reads are removed by an XOR swizzle applied to the SOURCE address of the DMA and to the read address
// (chunk ^= (row >> 1) & 7).  Two LDS stages of one K-step (32 k, (BM + BN) * 128 B) each: the DMA of step t + 1 is
// in flight while the MFMAs of step t run; one barrier per K-step.  K is walked channel-group outer, tap inner, so
// the nine taps of a 3x3 layer re-read the same few lines from L2 while they are still there.
//
// Reference call sites replaced: every Conv2d / Linear / ConvTranspose2d executed by Detectron2 0.6
// `GeneralizedRCNN.inference` under `predictor(image)` (reference src/functions/inference.py:1395,1398,1507,1669;
// src/data/models.py:107).
#include <utility>
#include "common.h"
#ifndef P32_ABLATE
#define P32_ABLATE 0      // timing-only dev builds: 1 = no A DMA after the first two steps, 2 = no B DMA, 4 = no MFMAs,
                          // 32 = no residual loads, 64 = no global stores in the epilogue, 128 = one K-step only,
                          // 256 = sigmoid / nearest-2x paths compiled out (instruction counting), 512 = no epilogue passes (the MFMAs die with them),
                          // 2048 = no epilogue, accumulators kept alive (K loop + prologue only),
                          // 1024 = no fragment reads (scripts/build_variant.sh, scripts/gpu_ablate.sh)
#endif

#ifndef P32_HEAD_DIRECT
#define P32_HEAD_DIRECT 1  // 0: the fused-head layer through the general epilogue (A/B builds)
#endif
#ifndef P32_SINGLE
#define P32_SINGLE 0      // 1: the flagged single-plane build of this file (Makefile: conv_p32_single.o, entry point
                          // demia_conv2d_p32_single): fp16 operands in the high plane, ONE MFMA per product, zero low plane out
#endif
#ifndef P32_DEV_TILES
#define P32_DEV_TILES 0   // 1: also instantiate the experimental tiles / schedules reachable through tile hints only (ping-pong
                          // kernel, three LDS stages, 32x32x16 MFMAs, alternative wave grids): dev builds for same-box A/B
#endif
#ifndef P32_ST_AUX
#define P32_ST_AUX 0      // cache policy of the epilogue's plane stores (raw buffer `aux`: 2 = nt, streaming) -- A/B builds
#endif
#ifndef P32_RES_AUX
#define P32_RES_AUX 0     // ... of its residual loads
#endif
#ifndef P32_A_NT
#define P32_A_NT 0        // 1: the A-operand LDS-DMA of 1x1 layers carries `nt` (every line is read once per column tile)
#endif
#ifndef P32_FRAG_PIPE
#define P32_FRAG_PIPE 0   // 1: A fragments double-buffered in registers, reads of tile-row i + 1 pinned in front of the MFMAs of row i
#endif

namespace {

struct ConvQ {
    const void* in;
    const float* in_meta;
    const void* w;
    const float* scale;
    const float* bias;
    const void* res;
    const float* res_meta;
    void* out;
    float* out_meta;
    float wbound, bbound;
    int N, H, W, Cin, Ho, Wo, Cout, CoutPad, KH, KW, stride, pad;
    int act, res_mode, out_f32, out_ld;
    int M, HoWo, ntn, nwg, ksteps, taps;
    int kloop;              // K-steps the kernel walks: ksteps, or ksteps / 2 in the h-only single-plane kernel (HK, 64 k per step)
    unsigned in_bytes, w_bytes;
    unsigned out_bytes, res_bytes;   // planes epilogue (EPI_PLANES): sizes of the output / residual buffers incl. their headers
    const float* head_w;    // fused 1x1 head (head_n > 0): [head_n][256] f32 weights applied to every 256-channel block of the
    const float* head_b;    // activated output row, + bias, sigmoid -> head_out[(m * ntn + tile_n) * head_ld + j]; the P32
    float* head_out;        // output itself is then not written
    int head_n, head_ld, head_act;
    int groups, group_rows, row0;   // scale groups (images): in_meta / res_meta / out_meta are [groups][2]; output row m of this
                                    // call belongs to group (m + row0) / group_rows
    int no_hk;              // single-plane build only (A/B switch DEMIA_P32_NO_HK=1): the plain K-step of 32 with both planes moved
    int zero_low;           // single-plane build only: write the output's low plane as zeros (demia_conv_p32_desc.single == 2)
    int stagger_ticks;      // > 0: the second half of the first resident set of workgroups starts this many 10-ns ticks late
    int resident;           // workgroups resident at once (256 CUs x workgroups per CU) -- for the stagger
};

typedef __attribute__((address_space(3))) void lds_void;

// two 16-byte LDS reads at a compile-time offset (inline asm: issued where they stand, invisible to hipcc's waitcnt pass);
// WAIT >= 0: followed IN THE SAME asm block by s_waitcnt lgkmcnt(WAIT)
template <int OFF, int WAIT>
__device__ __forceinline__ void ds_read2_b128(f16x8& a, f16x8& b, unsigned addr_a, unsigned addr_b) {
    if constexpr (WAIT >= 0)
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4\n\ts_waitcnt lgkmcnt(%5)"
                     : "=&v"(a), "=&v"(b) : "v"(addr_a), "v"(addr_b), "n"(OFF), "n"(WAIT) : "memory");
    else
        asm volatile("ds_read_b128 %0, %2 offset:%4\n\tds_read_b128 %1, %3 offset:%4"
                     : "=&v"(a), "=&v"(b) : "v"(addr_a), "v"(addr_b), "n"(OFF) : "memory");
}

template <int... Is, class F>
__device__ __forceinline__ void static_for_impl(std::integer_sequence<int, Is...>, F&& f) { (f(std::integral_constant<int, Is>{}), ...); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) { static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f)); }
typedef int i32x4 __attribute__((ext_vector_type(4)));

// One LDS-DMA piece: 64 lanes x 16 B from `rsrc` at byte offset voff (per lane) + soff (scalar) to LDS bytes
// [lds_addr, lds_addr + 1024).  Inline asm on purpose: hipcc orders every later ds_read behind a builtin LDS-DMA with
// s_waitcnt vmcnt(0) (it cannot prove the two stages disjoint), which would serialise the DMA of step t + 1 with the
// MFMAs of step t.  The kernel counts these loads itself (s_waitcnt vmcnt(0) before the barrier that publishes a stage).
__device__ __forceinline__ void dma16(const i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ void dma16_nt(const i32x4 rsrc, unsigned lds_addr, unsigned voff, unsigned soff) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1\n\ts_nop 0\n\tbuffer_load_dwordx4 %2, %3, %4 offen nt lds\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "s"(lds_addr), "v"(voff), "s"(rsrc), "s"(soff) : "memory");
}
__device__ __forceinline__ i32x4 make_rsrc(const void* base, unsigned bytes) {
    const unsigned long long a = reinterpret_cast<unsigned long long>(base);
    i32x4 r;
    r[0] = __builtin_amdgcn_readfirstlane((int)(unsigned)a);
    r[1] = __builtin_amdgcn_readfirstlane((int)((unsigned)(a >> 32) & 0xffffu));      // stride 0
    r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
    r[3] = 0x00020000;
    return r;
}

__device__ __forceinline__ float apply_act_q(float v, int act) {
    if (act == DEMIA_ACT_RELU) return v > 0.f ? v : 0.f;
    if (act == DEMIA_ACT_SIGMOID) return 1.0f / (1.0f + expf(-v));
    return v;
}

// power of two that brings |x| <= bound below 2^15
__device__ __forceinline__ float plane_scale(float bound) {
    if (!(bound > 0.f) || !(bound < 3.0e38f)) return 1.f;
    int e = 14 - ilogbf(bound);
    e = e < -100 ? -100 : (e > 100 ? 100 : e);
    return ldexpf(1.f, e);
}

// accumulators of one 32-row tile-row (32x32 MFMA C layout: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5))
// into the epilogue's LDS image; `e` points at this wave's first row / column, EF floats per image row
template <int TN, int EF>
__device__ __forceinline__ void write_acc32(const f32x16 (&row)[TN], float* e, int lane) {
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
            e[((r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)) * EF + j * 32 + (lane & 31)] = row[j][r];
}
// the same for 16x16 MFMA tiles (C layout: col = lane & 15, row = 4 * (lane >> 4) + r): two tile-rows of 16, 2 TN tile-columns
template <int TN, int EF>
__device__ __forceinline__ void write_acc16(const f32x4 (&r0)[2 * TN], const f32x4 (&r1)[2 * TN], float* e, int lane) {
#pragma unroll
    for (int j = 0; j < 2 * TN; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            e[(4 * (lane >> 4) + r) * EF + j * 16 + (lane & 15)] = r0[j][r];
            e[(16 + 4 * (lane >> 4) + r) * EF + j * 16 + (lane & 15)] = r1[j][r];
        }
}

// Epilogues.  The guarded one (p32_epilogue: f32 outputs, fused head, odd channel counts): TM passes; in pass i every wave
// hands tile-row i of its accumulators to LDS, then the 512 threads walk the WM * 32 rows x BN columns in 8-channel
// groups: scale / bias / residual / activation, split into planes, 16-byte stores.  The planes one (p32_epilogue_planes,
// every ordinary P32 layer) does the same wave by wave without workgroup barriers, in straight-line code with hardware
// bounded buffer accesses of whole 128-byte lines.  All waves have passed a barrier after their last LDS read.
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }

// Scales are per GROUP of rows (one group per image, so that a tile's result does not depend on its batch neighbours).  A
// tile of <= 256 rows meets at most three groups (group_rows >= 128, checked by the host): g0, g0 + 1, g0 + 2, the second
// and third starting at local rows b1 and b2.  Everything here is block-uniform and is read at the START of the kernel
// (scalar loads, hidden behind the K loop; read in the epilogue they were nine dependent memory round trips per tile) and
// kept in scalar registers -- the 256 x 256 tile has no vector register to spare.
struct GroupScales {      // plain scalars, no arrays: the struct must dissolve into (scalar) registers
    int g0, b1, b2;
    float post0, post1, post2, resi0, resi1, resi2, sout0, sout1, sout2;
};

__device__ __forceinline__ void load_group_scale(const ConvQ& p, int gi, bool planes_out, float& post, float& resi, float& sout) {
    gi = min(gi, p.groups - 1);
    const float amax_in = p.in_meta[2 * gi], s_in = p.in_meta[2 * gi + 1];
    float amax_res = 0.f, s_res = 1.f;
    if (p.res_mode != DEMIA_RES_NONE) { amax_res = p.res_meta[2 * gi]; s_res = p.res_meta[2 * gi + 1]; }
    post = uniform(1.0f / s_in);
    resi = uniform(p.res_mode != DEMIA_RES_NONE ? 1.0f / s_res : 0.f);
    sout = uniform(planes_out ? plane_scale(amax_in * p.wbound + p.bbound + amax_res) : 1.f);
}

__device__ __forceinline__ GroupScales load_group_scales(const ConvQ& p, int m0, bool planes_out) {
    GroupScales gs;
    gs.g0 = (m0 + p.row0) / p.group_rows;
    gs.b1 = (gs.g0 + 1) * p.group_rows - p.row0;
    gs.b2 = gs.b1 + p.group_rows;
    load_group_scale(p, gs.g0, planes_out, gs.post0, gs.resi0, gs.sout0);
    load_group_scale(p, gs.g0 + 1, planes_out, gs.post1, gs.resi1, gs.sout1);
    load_group_scale(p, gs.g0 + 2, planes_out, gs.post2, gs.resi2, gs.sout2);
    return gs;
}

template <int WM, int WN, int TM, int TN, bool HEAD, typename WriteRow>
__device__ __forceinline__ void p32_epilogue(const ConvQ& p, const GroupScales& gs, char* smem, WriteRow&& write_tile_row, int wm, int wn, int m0, int n0) {
    constexpr int BN = WN * TN * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    constexpr int EROW = BN * 4 + 16;
    constexpr int GPR = BN / 8, RSTEP = 512 / GPR, ITEMS = WM * 32 / RSTEP;
    // (the launch sizes the LDS for max(two stages, this image))
    static_assert(512 % GPR == 0 && (WM * 32) % RSTEP == 0, "epilogue split");
    constexpr int BM_ = WM * TM * 32;
    static_assert(!HEAD || BN == 256, "the fused head needs a 256-wide tile");
    const bool planes_out = !HEAD && !p.out_f32;
    // (taken out of the struct through readfirstlane: selects between plain struct fields get rewritten into an indexed
    // load of the struct, which then has to live in scratch memory)
    const int g0 = __builtin_amdgcn_readfirstlane(gs.g0), b1 = __builtin_amdgcn_readfirstlane(gs.b1), b2 = __builtin_amdgcn_readfirstlane(gs.b2);
    const float post0 = uniform(gs.post0), post1 = uniform(gs.post1), post2 = uniform(gs.post2);
    const float resi0 = uniform(gs.resi0), resi1 = uniform(gs.resi1), resi2 = uniform(gs.resi2);
    const float sout0 = uniform(gs.sout0), sout1 = uniform(gs.sout1), sout2 = uniform(gs.sout2);
    if (planes_out && n0 == 0 && tid == 0) {
        // the tile (of the first column block) that holds a group's first row publishes the group's scale
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int first = d == 0 ? g0 * p.group_rows - p.row0 : (d == 1 ? b1 : b2);
            if (g0 + d < p.groups && first >= m0 && first < m0 + BM_ && first < p.M)
                p.out_meta[2 * (g0 + d) + 1] = d == 0 ? sout0 : (d == 1 ? sout1 : sout2);
        }
    }
    const int g = tid % GPR, r_first = tid / GPR;
    const int co = n0 + g * 8;
    float sc[8], bs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const bool ok = (co + q) < p.Cout;
        sc[q] = (p.scale && ok) ? p.scale[co + q] : 1.0f;
        bs[q] = (p.bias && ok) ? p.bias[co + q] : 0.0f;
    }
    float vmax3[3] = {0.f, 0.f, 0.f};
    // fused head (BN == 256 only): this thread's 8 columns of up to 4 head rows
    constexpr int HMAX = 4;
    float hw[HMAX][8];
    constexpr bool head_on = HEAD;      // its own instantiation: the head's 32 weight registers stay out of every other kernel
    if (head_on) {
#pragma unroll
        for (int j = 0; j < HMAX; ++j)
#pragma unroll
            for (int q = 0; q < 8; ++q) hw[j][q] = j < p.head_n ? p.head_w[j * 256 + g * 8 + q] : 0.f;
    }
    const int tile_n_ = n0 / BN;
    const bool res_on = p.res_mode != DEMIA_RES_NONE, sigmoid_on = p.act == DEMIA_ACT_SIGMOID;
    const float act_lo = p.act == DEMIA_ACT_RELU ? 0.f : -INFINITY;
    const char* resb = reinterpret_cast<const char*>(p.res) + 128;
    char* outb = reinterpret_cast<char*>(p.out) + 128;
    const long cbytes = (long)p.Cout * 4;                       // bytes per P32 pixel of the output / residual
    const int gofs = (co >> 5) * 128 + ((co & 31) >> 3) * 16;   // this thread's 8 channels inside a pixel (high plane)
    // output row of item k in pass i, and the residual pixel that goes with it
    auto row_of = [&](int i, int k) { const int lr = r_first + k * RSTEP; return m0 + (lr >> 5) * (TM * 32) + i * 32 + (lr & 31); };
    auto res_pix = [&](int m) -> long {
        if (p.res_mode == DEMIA_RES_SAME) return m;
        const int n = m / p.HoWo;
        const int rem = m - n * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int Hr = (p.Ho + 1) >> 1, Wr = (p.Wo + 1) >> 1;
        return ((long)n * Hr + (ho >> 1)) * Wr + (wo >> 1);
    };
    // The residual rows of a pass are requested ONE PASS AHEAD (all of them at once): a short-K layer is otherwise one
    // exposed HBM round trip per item -- sixteen in a row for a 256 x 256 tile.
    f16x8 rh[ITEMS], rl[ITEMS];
    const bool has_res = p.res_mode != DEMIA_RES_NONE && co < p.Cout;
    auto load_res = [&](int i) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int m = row_of(i, k);
            if (has_res && m < p.M && !(P32_ABLATE & 32)) {
                const char* rp = resb + res_pix(m) * cbytes + gofs;
                rh[k] = *reinterpret_cast<const f16x8*>(rp);
                rl[k] = *reinterpret_cast<const f16x8*>(rp + 64);
            }
        }
    };
    load_res(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        if (i > 0) __syncthreads();
        // this wave's accumulators of tile-row i -> image rows wm * 32 .. + 31, columns wn * TN * 32 .. (EROW bytes per row)
        write_tile_row(i, reinterpret_cast<float*>(smem) + (wm * 32) * (EROW / 4) + wn * TN * 32);
        __syncthreads();
        f16x8 ch[ITEMS], cl[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) { ch[k] = rh[k]; cl[k] = rl[k]; }
        if (i + 1 < TM) load_res(i + 1);
        if (co < p.Cout) {
#pragma unroll
            for (int k = 0; k < ITEMS; ++k) {
                const int lr = r_first + k * RSTEP;
                const int m = row_of(i, k);
                if (m >= p.M) continue;
                const float4 x0 = *reinterpret_cast<const float4*>(smem + lr * EROW + g * 32);
                const float4 x1 = *reinterpret_cast<const float4*>(smem + lr * EROW + g * 32 + 16);
                float v[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
                const int gd = (m >= b1) + (m >= b2);
                const float post = gd == 0 ? post0 : (gd == 1 ? post1 : post2);     // exact powers of two
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = (v[q] * post) * sc[q] + bs[q];
                if (res_on) {
                    const float res_inv = gd == 0 ? resi0 : (gd == 1 ? resi1 : resi2);
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] += ((float)ch[k][q] + (float)cl[k][q]) * res_inv;
                }
                // ReLU as a max against 0 / -inf: no per-element branch (one wave-uniform branch per ITEM at most -- the
                // per-element `switch (act)` this replaces cost three scalar branches and an inlined division per element
                // and made the epilogue the longest part of every short-K layer)
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = fmaxf(v[q], act_lo);
                if (sigmoid_on) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) v[q] = 1.0f / (1.0f + expf(-v[q]));
                }
                if (head_on) {
                    // 256 activated channels of this row live in the 32 lanes of this half-wave: 8 FMAs per head row, then
                    // a 5-step butterfly; lane 0 of the group adds the bias, applies the sigmoid and stores
                    float acc_h[HMAX];
#pragma unroll
                    for (int j = 0; j < HMAX; ++j) {
                        float a = 0.f;
#pragma unroll
                        for (int q = 0; q < 8; ++q) a = fmaf(v[q], hw[j][q], a);
#pragma unroll
                        for (int o = 16; o > 0; o >>= 1) a += __shfl_xor(a, o, 32);
                        acc_h[j] = a;
                    }
                    if (g == 0) {
                        float* o = p.head_out + ((long)m * p.ntn + tile_n_) * p.head_ld;
#pragma unroll
                        for (int j = 0; j < HMAX; ++j)
                            if (j < p.head_n) {
                                const float z = acc_h[j] + p.head_b[j];
                                o[j] = p.head_act == DEMIA_ACT_SIGMOID ? 1.0f / (1.0f + expf(-z)) : (p.head_act == DEMIA_ACT_RELU ? fmaxf(z, 0.f) : z);
                            }
                    }
                } else if (p.out_f32) {
                    float* o = reinterpret_cast<float*>(p.out) + (long)m * p.out_ld + co;
                    if (co + 8 <= p.Cout && (p.out_ld & 3) == 0) {
                        *reinterpret_cast<float4*>(o) = make_float4(v[0], v[1], v[2], v[3]);
                        *reinterpret_cast<float4*>(o + 4) = make_float4(v[4], v[5], v[6], v[7]);
                    } else {
#pragma unroll
                        for (int q = 0; q < 8; ++q)
                            if (co + q < p.Cout) o[q] = v[q];
                    }
                } else {
                    f16x8 h, l;
                    const float s_out = gd == 0 ? sout0 : (gd == 1 ? sout1 : sout2);
                    float vm = 0.f;
#pragma unroll
                    for (int q = 0; q < 8; ++q) {
                        vm = fmaxf(vm, fabsf(v[q]));
                        const float y = v[q] * s_out;
                        h[q] = (_Float16)y;
                        l[q] = (P32_SINGLE && p.zero_low) ? (_Float16)0.f : (_Float16)(y - (float)h[q]);
                    }
                    vmax3[0] = fmaxf(vmax3[0], gd == 0 ? vm : 0.f);
                    vmax3[1] = fmaxf(vmax3[1], gd == 1 ? vm : 0.f);
                    vmax3[2] = fmaxf(vmax3[2], gd == 2 ? vm : 0.f);
                    char* o = outb + (long)m * cbytes + gofs;
                    if (P32_ABLATE & 64) {
                        asm volatile("" :: "v"(h), "v"(l), "v"(o));
                    } else {
                        *reinterpret_cast<f16x8*>(o) = h;
                        *reinterpret_cast<f16x8*>(o + 64) = l;
                    }
                }
            }
        }
    }
    if (planes_out) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            if (d > 0 && (d == 1 ? b1 : b2) >= m0 + BM_) break;            // this tile has no rows of that group (block-uniform)
            if (g0 + d >= p.groups) break;
            float vmax = vmax3[d];
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
            // the bound only grows: a (possibly stale) read that already covers this wave's maximum makes the atomic unnecessary
            float* slot = p.out_meta + 2 * (g0 + d);
            if (lane == 0 && vmax > *reinterpret_cast<volatile const float*>(slot))
                atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(vmax));
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of a plain P32 layer (planes out, whole 8-channel groups, buffers below 4 GiB): STRAIGHT-LINE code.  Residual
// loads and plane stores are buffer instructions whose bounds the hardware checks (rows >= M: loads return 0, stores are
// dropped; a layer without residual loads from an empty buffer and scales by 0), so every pass issues the same number of
// memory instructions and the waits are counted: a pass waits for ITS residual lines only, not -- as with stores inside
// `if (m < M)` branches, where the count is unknown and the wait becomes vmcnt(0) -- for the previous pass's stores to be
// acknowledged by the L2.  That acknowledgement was the longest stall of every short-K (HBM-bound) layer.
struct NoHook { __device__ __forceinline__ void operator()() const {} };
// (`leader`: the one thread that publishes the scale of a group whose first row lies in this tile; `hook`: called after every
//  item -- the ping-pong kernel below puts its barriers there; the plain kernel passes nothing)
template <int WM, int WN, int TM, int TN, typename WriteRow, typename Hook = NoHook>
__device__ __forceinline__ void p32_epilogue_planes(const ConvQ& p, const GroupScales& gs, char* smem, WriteRow&& write_tile_row, int wm, int wn,
                                                    int m0, int n0, bool leader = (threadIdx.x == 0), Hook&& hook = NoHook()) {
    constexpr int BM_ = WM * TM * 32;
    const int tid = threadIdx.x, lane = tid & 63;
    // WAVE-LOCAL passes: a wave takes the 32 x (TN * 32) block of ITS OWN accumulators through ITS OWN piece of LDS and
    // finishes it -- no workgroup barrier anywhere in the epilogue (an LDS read of a wave sees that wave's earlier writes:
    // its LDS instructions execute in order), so the eight waves drift apart and one wave's memory waits are another's
    // arithmetic.  (With all 512 threads sharing one image every pass was a barrier, and the passes ran in lockstep.)
    constexpr int LDW = TN * 32 + 4;                         // floats per row of the wave's block (+4: conflict-free writes)
    constexpr int PPW = 8 / TN, RPI = 2 * PPW, ITEMS = 32 / RPI;   // row pairs / rows per memory instruction; items per pass
    static_assert(TN == 1 || TN == 2 || TN == 4, "TN");
    char* const wsm = smem + (tid >> 6) * (32 * LDW * 4);
    const int g0 = __builtin_amdgcn_readfirstlane(gs.g0), b1 = __builtin_amdgcn_readfirstlane(gs.b1), b2 = __builtin_amdgcn_readfirstlane(gs.b2);
    const float post0 = uniform(gs.post0), post1 = uniform(gs.post1), post2 = uniform(gs.post2);
    const float resi0 = uniform(gs.resi0), resi1 = uniform(gs.resi1), resi2 = uniform(gs.resi2);
    const float sout0 = uniform(gs.sout0), sout1 = uniform(gs.sout1), sout2 = uniform(gs.sout2);
    if (n0 == 0 && leader) {
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int first = d == 0 ? g0 * p.group_rows - p.row0 : (d == 1 ? b1 : b2);
            if (g0 + d < p.groups && first >= m0 && first < m0 + BM_ && first < p.M)
                p.out_meta[2 * (g0 + d) + 1] = d == 0 ? sout0 : (d == 1 ? sout1 : sout2);
        }
    }
    // FULL-LINE memory instructions.  A P32 line is [h0 h1 h2 h3 | l0 l1 l2 l3] (16 bytes = 8 channels each).  A thread
    // computes 8 channels of one row, i.e. owns h_j and l_j -- two 16-byte pieces 64 bytes apart, and four lanes together
    // would touch HALF a line per instruction (the L1 then sends 64-byte requests, and the number of requests a CU may have
    // in flight is what bounds these layers: ~45 per CU at ~900 cycles each).  So lanes work in groups of EIGHT on a PAIR of
    // rows (r, r + 1): lane 8 q + j (j < 4) computes chunk j of row r, lane 8 q + 4 + j chunk j of row r + 1, and memory
    // instruction X of a pair moves the whole line of row r + X, lane 8 q + t taking bytes 16 t .. 16 t + 15.  What a lane
    // moved for its partner (lane ^ 4) changes hands with one 16-byte lane exchange.
    const int cj = lane & 3, rs = (lane >> 2) & 1, lg = lane >> 3;
    const int line = lg % TN;                                // which of the wave's TN lines (32 channels) per row
    const int r_first = (lg / TN) * 2 + rs;                  // row of item 0 inside the wave's 32-row block
    const int co = n0 + ((wn * TN + line) * 4 + cj) * 8;
    float sc[8], bs[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        sc[q] = p.scale ? p.scale[co + q] : 1.0f;
        bs[q] = p.bias ? p.bias[co + q] : 0.0f;
    }
    const bool sigmoid_on = p.act == DEMIA_ACT_SIGMOID;
    const float act_lo = p.act == DEMIA_ACT_RELU ? 0.f : -INFINITY;
    const bool res_on = p.res_mode != DEMIA_RES_NONE;
    const bool single = P32_SINGLE != 0 && p.zero_low != 0;      // (the default object: constant false)
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(p.out, 0, (int)p.out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(res_on ? p.res : p.out), 0, res_on ? (int)p.res_bytes : 0, 0x00020000);
    const unsigned cbytes = (unsigned)p.Cout * 4u;                                     // bytes per P32 pixel
    const unsigned lofs = 128u + (unsigned)((co >> 5) * 128 + (lane & 7) * 16);   // header + this lane's 16 bytes of the pair's lines
    auto row_of = [&](int i, int k) { return m0 + wm * (TM * 32) + i * 32 + r_first + k * RPI; };
    auto res_off = [&](int m) -> unsigned {
        if (p.res_mode != DEMIA_RES_UP2 || (P32_ABLATE & 256)) return (unsigned)m * cbytes + lofs;   // (m < M + 256: no wrap below 4 GiB)
        const int n = m / p.HoWo;
        const int rem = m - n * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int Hr = (p.Ho + 1) >> 1, Wr = (p.Wo + 1) >> 1;
        return (unsigned)((n * Hr + (ho >> 1)) * Wr + (wo >> 1)) * cbytes + lofs;
    };
    // The exchange as two DPP moves per dword, no select: `row_shr:4` hands lanes 4..7 / 12..15 of a row (the lanes of
    // row r + 1, bank mask 0b1010) the value of the lane four below, `row_shl:4` hands lanes 0..3 / 8..11 (row r, bank mask
    // 0b0101) the value of the lane four above; the lanes a move does not write keep `own`.
    auto from_below = [&](u32x4 own, u32x4 theirs) -> u32x4 {   // lanes of row r + 1 take the partner's `theirs`
        u32x4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = (unsigned)__builtin_amdgcn_update_dpp((int)own[c], (int)theirs[c], 0x114, 0xf, 0xa, false);
        return r;
    };
    auto from_above = [&](u32x4 own, u32x4 theirs) -> u32x4 {   // lanes of row r take the partner's `theirs`
        u32x4 r;
#pragma unroll
        for (int c = 0; c < 4; ++c) r[c] = (unsigned)__builtin_amdgcn_update_dpp((int)own[c], (int)theirs[c], 0x104, 0xf, 0x5, false);
        return r;
    };
    u32x4 rh[ITEMS], rl[ITEMS];                              // as loaded: [0] = this lane's piece of row r, [1] = of row r + 1
    auto load_res = [&](int i) {
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int mp = row_of(i, k) - rs;                // row r of the pair
            if (P32_ABLATE & 32) { rh[k] = u32x4{0, 0, 0, 0}; rl[k] = rh[k]; continue; }
            rh[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, res_off(mp), 0, P32_RES_AUX);
            rl[k] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, res_off(mp + 1), 0, P32_RES_AUX);
        }
    };
    float vmax0 = 0.f, vmax1 = 0.f, vmax2 = 0.f;
    if (P32_ABLATE & 512) return;                            // timing-only: everything but the passes
    load_res(0);
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        // (compiler barriers only: the reads of pass i - 1 are issued before these writes, the writes before the reads below)
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        write_tile_row(i, reinterpret_cast<float*>(wsm));
        asm volatile("" ::: "memory");
        __builtin_amdgcn_wave_barrier();
        f16x8 ch[ITEMS], cl[ITEMS];
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            // a lane of row r loaded h_j(r) and h_j(r + 1), its partner l_j(r) and l_j(r + 1)
            ch[k] = __builtin_bit_cast(f16x8, from_below(rh[k], rl[k]));     // row r: own first load; row r + 1: the partner's second
            cl[k] = __builtin_bit_cast(f16x8, from_above(rl[k], rh[k]));     // row r: the partner's first; row r + 1: own second
        }
        if (i + 1 < TM) load_res(i + 1);
#pragma unroll
        for (int k = 0; k < ITEMS; ++k) {
            const int lr = r_first + k * RPI;
            const int m = row_of(i, k);
            const float4 x0 = *reinterpret_cast<const float4*>(wsm + lr * (LDW * 4) + (line * 4 + cj) * 32);
            const float4 x1 = *reinterpret_cast<const float4*>(wsm + lr * (LDW * 4) + (line * 4 + cj) * 32 + 16);
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            f32x2 v2[4] = {{x0.x, x0.y}, {x0.z, x0.w}, {x1.x, x1.y}, {x1.z, x1.w}};
            const int gd = (m >= b1) + (m >= b2);
            const float post = gd == 0 ? post0 : (gd == 1 ? post1 : post2);     // exact powers of two
            const float res_inv = gd == 0 ? resi0 : (gd == 1 ? resi1 : resi2);  // 0 without a residual
            const float s_out = gd == 0 ? sout0 : (gd == 1 ? sout1 : sout2);
            float v[8];
#pragma unroll
            for (int q = 0; q < 4; ++q) {                                       // packed f32: two channels per instruction
                const f32x2 scq = {sc[2 * q], sc[2 * q + 1]}, bsq = {bs[2 * q], bs[2 * q + 1]};
                const f32x2 t = (v2[q] * post) * scq + bsq;
                v[2 * q] = t.x;
                v[2 * q + 1] = t.y;
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // residual (h + l) r as two mixed-precision FMAs (h r and l r are exact: r is a power of two)
                v[q] = fmaf((float)cl[k][q], res_inv, fmaf((float)ch[k][q], res_inv, v[q]));
                v[q] = fmaxf(v[q], act_lo);
            }
            if (sigmoid_on && !(P32_ABLATE & 256)) {
#pragma unroll
                for (int q = 0; q < 8; ++q) v[q] = 1.0f / (1.0f + expf(-v[q]));
            }
            f16x8 h, l;
            float vm = 0.f;
#pragma unroll
            for (int q = 0; q < 8; q += 2) {
                vm = fmaxf(vm, fmaxf(fabsf(v[q]), fabsf(v[q + 1])));
                const f32x2 y = f32x2{v[q], v[q + 1]} * s_out;
                h[q] = (_Float16)y.x;
                h[q + 1] = (_Float16)y.y;
                l[q] = (_Float16)fmaf(-(float)h[q], 1.0f, y.x);
                l[q + 1] = (_Float16)fmaf(-(float)h[q + 1], 1.0f, y.y);
            }
            if (single) l = f16x8{0, 0, 0, 0, 0, 0, 0, 0};
            vm = m < p.M ? vm : 0.f;                                            // rows beyond M hold bias only: not part of the tensor
            vmax0 = fmaxf(vmax0, gd == 0 ? vm : 0.f);
            vmax1 = fmaxf(vmax1, gd == 1 ? vm : 0.f);
            vmax2 = fmaxf(vmax2, gd == 2 ? vm : 0.f);
            const u32x4 hu = __builtin_bit_cast(u32x4, h), lu = __builtin_bit_cast(u32x4, l);
            const u32x4 d0 = from_below(hu, lu);      // piece of row r's line: row r lanes their h_j, row r + 1 lanes the partner's l_j
            const u32x4 d1 = from_above(lu, hu);      // piece of row r + 1's line: row r lanes the partner's h_j, row r + 1 lanes their l_j
            const unsigned off = (unsigned)(m - rs) * cbytes + lofs;
            if (P32_ABLATE & 64) {
                asm volatile("" :: "v"(d0), "v"(d1), "v"(off));
            } else {
                __builtin_amdgcn_raw_buffer_store_b128(d0, rs_out, off, 0, P32_ST_AUX);
                __builtin_amdgcn_raw_buffer_store_b128(d1, rs_out, off + cbytes, 0, P32_ST_AUX);
            }
            hook();
        }
    }
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        if (d > 0 && (d == 1 ? b1 : b2) >= m0 + BM_) break;            // this tile has no rows of that group (block-uniform)
        if (g0 + d >= p.groups) break;
        float vmax = d == 0 ? vmax0 : (d == 1 ? vmax1 : vmax2);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) vmax = fmaxf(vmax, __shfl_xor(vmax, o));
        float* slot = p.out_meta + 2 * (g0 + d);
        if (lane == 0 && vmax > *reinterpret_cast<volatile const float*>(slot))
            atomicMax(reinterpret_cast<unsigned int*>(slot), __float_as_uint(vmax));
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// Epilogue of the fused-head layer (ConvTranspose2d(2, 2) as a GEMM + ReLU, the 1x1 mask predictor + sigmoid folded in;
// the P32 output is never written) STRAIGHT FROM THE ACCUMULATORS.  In the 16x16 MFMA result layout a lane holds, per
// tile, four consecutive rows of ONE column: it scales / shifts / activates its 2 TN columns of a row, multiplies them by
// the head weights of those columns, and the 16 lanes that share the rows add up with four DPP row rotations; the WN waves
// that share a row meet once in LDS ([WN][BM][HN] partial sums, one workgroup barrier), then one thread per (row, head row)
// adds the four partials in wave order, the bias, applies the head activation and stores.  The general epilogue took the
// tile through an LDS image in TM passes of two barriers each and reduced with a 5-step ds_bpermute butterfly per head row
// and item: 320 LDS crossbar operations per thread and tile for a layer that writes 8 bytes per row.
template <int WM, int WN, int TM, int TN, int HN>
__device__ __forceinline__ void p32_epilogue_head_direct(const ConvQ& p, const GroupScales& gs, char* smem, const f32x4 (&acc16)[2 * TM][2 * TN],
                                                         int wm, int wn, int m0, int n0) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    static_assert(BN == 256, "the fused head needs a 256-wide tile");
    const int tid = threadIdx.x, lane = tid & 63, c16 = lane & 15, rq = lane >> 4;
    const int b1 = __builtin_amdgcn_readfirstlane(gs.b1), b2 = __builtin_amdgcn_readfirstlane(gs.b2);
    const float post0 = uniform(gs.post0), post1 = uniform(gs.post1), post2 = uniform(gs.post2);
    float sc[2 * TN], bs[2 * TN], hw[HN][2 * TN];
#pragma unroll
    for (int jn = 0; jn < 2 * TN; ++jn) {
        const int cl = wn * TN * 32 + jn * 16 + c16;              // column inside the 256-wide tile
        sc[jn] = p.scale ? p.scale[n0 + cl] : 1.0f;
        bs[jn] = p.bias ? p.bias[n0 + cl] : 0.0f;
#pragma unroll
        for (int j = 0; j < HN; ++j) hw[j][jn] = j < p.head_n ? p.head_w[j * 256 + cl] : 0.f;
    }
    const float act_lo = p.act == DEMIA_ACT_RELU ? 0.f : -INFINITY;
    float* part = reinterpret_cast<float*>(smem);                 // [WN][BM][HN]; every wave has passed the K loop's last barrier
#pragma unroll
    for (int i = 0; i < 2 * TM; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = wm * TM * 32 + i * 16 + rq * 4 + r;
            const int m = m0 + row;
            const int gd = (m >= b1) + (m >= b2);
            const float post = gd == 0 ? post0 : (gd == 1 ? post1 : post2);
            float a[HN];
#pragma unroll
            for (int j = 0; j < HN; ++j) a[j] = 0.f;
#pragma unroll
            for (int jn = 0; jn < 2 * TN; ++jn) {
                const float v = fmaxf((acc16[i][jn][r] * post) * sc[jn] + bs[jn], act_lo);
#pragma unroll
                for (int j = 0; j < HN; ++j) a[j] = fmaf(v, hw[j][jn], a[j]);
            }
#pragma unroll
            for (int j = 0; j < HN; ++j) {
                // all-reduce over the 16 lanes of the DPP row (same rows, 16 different columns): rotations by 8, 4, 2, 1
                a[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[j]), 0x128, 0xf, 0xf, false));
                a[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[j]), 0x124, 0xf, 0xf, false));
                a[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[j]), 0x122, 0xf, 0xf, false));
                a[j] += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(a[j]), 0x121, 0xf, 0xf, false));
            }
            if (c16 == 0) {
#pragma unroll
                for (int j = 0; j < HN; ++j) part[((long)wn * BM + row) * HN + j] = a[j];
            }
        }
    }
    __syncthreads();
    const int tile_n_ = n0 / BN;
    for (int o = tid; o < BM * HN; o += WM * WN * 64) {
        const int row = o / HN, j = o - row * HN;
        const int m = m0 + row;
        if (m >= p.M || j >= p.head_n) continue;
        float z = 0.f;
#pragma unroll
        for (int w = 0; w < WN; ++w) z += part[((long)w * BM + row) * HN + j];
        z += p.head_b[j];
        p.head_out[((long)m * p.ntn + tile_n_) * p.head_ld + j] =
            p.head_act == DEMIA_ACT_SIGMOID ? 1.0f / (1.0f + expf(-z)) : (p.head_act == DEMIA_ACT_RELU ? fmaxf(z, 0.f) : z);
    }
}


constexpr int EPI_GENERIC = 0, EPI_PLANES = 1, EPI_HEAD = 2;

// HK (single-plane build only, Cin % 64 == 0): the K-step is 64 channels of ONE plane.  A single-plane product reads the high
// planes only, and with the plain K-step half of every DMA'd line (the low halves) is dead weight: at one MFMA per product the
// kernel is bound by the L2 -> LDS operand stream (10 TB/s over the chip at 666 TFLOP/s), not by the matrix pipe.  Here a
// lane fetches the HIGH half of channel group g (chunks 0..3 of the LDS line) or of group g + 1 (chunks 4..7) -- 64 bytes out
// of each of two adjacent 128-byte lines -- for A, and the high halves of weight K-steps (g, tap) and (g + 1, tap) for B: the
// same MFMAs and fragment reads per product, HALF the DMA bytes and half the barriers.  The fragment addresses are the ones
// the two-plane kernel uses for its planes (chunk = half * 4 + (lane >> 4)), so a stage is consumed as two K sub-steps.
template <int WM, int WN, int TM, int TN, bool M16 = true, int EPI = EPI_PLANES, int NST = 2, bool HK = false>
__global__ __launch_bounds__(WM * WN * 64, WM * WN == 8 ? 2 : 1) void conv_p32_kernel(const ConvQ p) {
    static_assert(NST == 2 || NST == 3, "two or three LDS stages");
    static_assert(!HK || (P32_SINGLE && M16 && NST == 2), "HK is a variant of the single-plane 16x16x32 kernel");
    // eight waves (two per SIMD, <= 256 registers each), or FOUR waves of a larger wave tile (one per SIMD, the whole
    // register file): a third less fragment traffic out of the LDS per output and half the barrier participants
    static_assert(WM * WN == 8 || (WM * WN == 4 && EPI == EPI_PLANES), "eight waves, or four (planes epilogue only)");
    constexpr int NW = WM * WN;
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    constexpr int STAGE = (BM + BN) * 128;
    constexpr int NIA = BM / 8, NIB = BN / 8;                     // 1-KiB DMA pieces (8 rows) per K-step

    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int swz = xcd_remap(blockIdx.x, p.nwg);
    const int tile_n = swz % p.ntn, tile_m = swz / p.ntn;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
#if P32_DEV_TILES
    if (p.stagger_ticks > 0) {
        // Short-K layers alternate a K loop that barely touches HBM with an epilogue that saturates it, and every CU starts
        // in the same phase: delaying half of the first resident set by about half a tile period puts one half of the
        // chip in its epilogue while the other half is in its K loop (the later workgroups inherit the offset).
        const int b = blockIdx.x;
        const bool late = p.resident > 256 ? (b >= p.resident / 2 && b < p.resident) : (b < p.resident && ((b >> 3) & 1));
        if (late) {
            const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
            while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)p.stagger_ticks) __builtin_amdgcn_s_sleep(32);
        }
    }
#endif

    const i32x4 rsrc_a = make_rsrc(p.in, p.in_bytes), rsrc_b = make_rsrc(p.w, p.w_bytes);
    const unsigned lds0 = (unsigned)(__SIZE_TYPE__)((lds_void*)smem);

    // ---- DMA bookkeeping: wave w moves pieces w, w + 8, ... (8 rows = 1 KiB each) of the A tile and of the B tile ----
    constexpr int QA = (NIA + NW - 1) / NW, QB = NIB / NW;
    constexpr bool A_EVEN = NIA % NW == 0;      // else the last round of A pieces is issued by the first NIA % NW waves only
    static_assert(NIB % NW == 0, "tile width is a multiple of 64");
    unsigned a_off[QA], a_msk[QA], b_off[QB];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int row = (wave + NW * q) * 8 + (lane >> 3);
        const int csw = (lane & 7) ^ ((row >> 1) & 7);
        const int m = m0 + row;
        const bool vm = m < p.M && row < BM;
        const int mm = vm ? m : 0;
        const int n = mm / p.HoWo;
        const int rem = mm - n * p.HoWo;
        const int ho = rem / p.Wo, wo = rem - ho * p.Wo;
        const int hi0 = ho * p.stride - p.pad, wi0 = wo * p.stride - p.pad;
        const long pix = ((long)n * p.H + hi0) * p.W + wi0;
        // HK: source chunk c = high half of channel group g (c < 4) or g + 1 (c >= 4): 16 (c & 3) bytes into line g + (c >> 2)
        a_off[q] = (unsigned)(128 + pix * (long)(p.Cin * 4) + (HK ? (csw >> 2) * 128 + (csw & 3) * 16 : csw * 16));
        unsigned mk = 0;
        for (int t = 0, th = 0, tw = 0; t < p.taps; ++t) {
            if (vm && (unsigned)(hi0 + th) < (unsigned)p.H && (unsigned)(wi0 + tw) < (unsigned)p.W) mk |= 1u << t;
            if (++tw == p.KW) { tw = 0; ++th; }
        }
        a_msk[q] = mk;
    }
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int row = (wave + NW * q) * 8 + (lane >> 3);
        const int csw = (lane & 7) ^ ((row >> 1) & 7);
        const int co = n0 + row;
        // HK: the high half of weight K-step (g, tap) (c < 4) or (g + 1, tap) (c >= 4), `taps` K-steps further on
        b_off[q] = (unsigned)(((co >> 6) * p.ksteps) * 8192 + (co & 63) * 128 + (HK ? (csw >> 2) * (p.taps * 8192) + (csw & 3) * 16 : csw * 16));
    }
    // K-step being REQUESTED: tap index, A byte offset of (tap, channel group), B byte offset -- all scalar
    int tap = 0, kw = 0, tstep = 0;
    unsigned sdelta = 0, srow = 0, sgrp = 0;
    const bool a_nt = P32_A_NT && p.taps == 1 && p.ntn == 1;      // (block-uniform)
    auto issue = [&](int st, int tp, unsigned sd, unsigned bd) {
        const unsigned sbase = __builtin_amdgcn_readfirstlane(lds0 + st * STAGE + wave * 1024);
        const bool first = bd < 2 * 8192u;
#pragma unroll
        for (int q = 0; q < QA; ++q) {
            if ((P32_ABLATE & 1) && !first) break;
            if (!A_EVEN && q == QA - 1 && wave + NW * q >= NIA) break;
            const bool ok = (a_msk[q] >> tp) & 1u;
            const unsigned vo = ok ? a_off[q] + sd : (a_off[q] & 0x70u);     // padding taps / rows beyond M: the zero header
            if (P32_A_NT && a_nt) dma16_nt(rsrc_a, sbase + q * (NW * 1024), vo, 0u);
            else dma16(rsrc_a, sbase + q * (NW * 1024), vo, 0u);
        }
#pragma unroll
        for (int q = 0; q < QB; ++q) {
            if ((P32_ABLATE & 2) && !first) break;
            dma16(rsrc_b, sbase + BM * 128 + q * (NW * 1024), b_off[q], bd);
        }
    };
    const unsigned pixb = (unsigned)(p.Cin * 4), rowb = (unsigned)(p.W * p.Cin * 4);
    // (tstep = the weight K-step being requested: HK walks channel groups in pairs, so it skips the second group's `taps` steps)
#define P32_ADVANCE()                                                                   \
    do {                                                                                \
        ++tstep;                                                                        \
        if (++tap == p.taps) { tap = 0; kw = 0; srow = 0; sgrp += HK ? 256u : 128u; sdelta = sgrp; if (HK) tstep += p.taps; } \
        else if (++kw == p.KW) { kw = 0; srow += rowb; sdelta = srow + sgrp; }          \
        else sdelta += pixb;                                                            \
    } while (0)

    // ---- fragment read addresses: row r of a tile at r * 128, 16-byte chunk c of it at (c ^ ((r >> 1) & 7)) * 16;
    //      chunk = plane * 4 + kk * 2 + (lane >> 5) ----
    const int fsw = (lane >> 1) & 7, hh = lane >> 5;
    int fa[4], fb[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const int ch = (((c >> 1) * 4 + (c & 1) * 2 + hh) ^ fsw) * 16;
        fa[c] = (wm * TM * 32 + (lane & 31)) * 128 + ch;
        fb[c] = BM * 128 + (wn * TN * 32 + (lane & 31)) * 128 + ch;
    }

    // M16 (default): v_mfma_f32_16x16x32_f16, one K sub-step per stage -- the chip holds a higher clock on that shape at
    // equal cycles per FLOP (MI355X_MICROARCH.md, DVFS give-back item 7): measured +6 % on the MFMA-bound layers (large 3x3
    // 420 -> 446, FC 427 -> 455 TFLOP/s).  M16 = false: 32x32x16, two K sub-steps per stage, kept for A/B (tile hints 31, 34).
    // The unused accumulator set is dead code.
    f32x16 acc[M16 ? 1 : TM][M16 ? 1 : TN];
    f32x4 acc16[M16 ? 2 * TM : 1][M16 ? 2 * TN : 1];
#pragma unroll
    for (int i = 0; i < (M16 ? 1 : TM); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 1 : TN); ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
    for (int i = 0; i < (M16 ? 2 * TM : 1); ++i)
#pragma unroll
        for (int j = 0; j < (M16 ? 2 * TN : 1); ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc16[i][j][r] = 0.f;
    // 16x16x32 operand: lane l holds row (l & 15), k = 8 (l >> 4) .. + 7 -> chunk plane * 4 + (l >> 4) of its row
    int fa16[2], fb16[2];
#pragma unroll
    for (int pl = 0; pl < 2; ++pl) {
        const int ch = ((pl * 4 + (lane >> 4)) ^ (((lane & 15) >> 1) & 7)) * 16;
        fa16[pl] = (wm * TM * 32 + (lane & 15)) * 128 + ch;
        fb16[pl] = BM * 128 + (wn * TN * 32 + (lane & 15)) * 128 + ch;
    }

    // (P32_FRAG_PIPE == 2, two stages: compute() also ends the K-step -- `s_waitcnt vmcnt(0); s_barrier` sits in FRONT of
    // the last tile-row's MFMAs, whose operands are in registers by then, so the wait for the slowest wave and for this
    // wave's DMA pieces runs under matrix work instead of after it)
    constexpr bool BARRIER_IN_COMPUTE = (P32_FRAG_PIPE == 2) && M16 && NST == 2;
    // single-plane operands (flagged --precision f16): the high-plane fragments only, ONE MFMA per product
    auto compute1 = [&](int st) {
        const char* sb = smem + st * STAGE;
        if constexpr (M16) {
            f16x8 bh[2 * TN];
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) bh[j] = *reinterpret_cast<const f16x8*>(sb + fb16[0] + j * 2048);
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + fa16[0] + i * 2048);
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) acc16[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], acc16[i][j], 0, 0, 0);
            }
        } else {
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                f16x8 bh[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) bh[j] = *reinterpret_cast<const f16x8*>(sb + fb[kk] + j * 4096);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + fa[kk] + i * 4096);
#pragma unroll
                    for (int j = 0; j < TN; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], acc[i][j], 0, 0, 0);
                }
            }
        }
        if (BARRIER_IN_COMPUTE) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    // HK: both halves of a stage line are high-plane K (k 0..31, 32..63): two MFMAs per tile, one per half
    auto compute_hk = [&](int st) {
        const char* sb = smem + st * STAGE;
        f16x8 b0[2 * TN], b1[2 * TN];
#pragma unroll
        for (int j = 0; j < 2 * TN; ++j) {
            b0[j] = *reinterpret_cast<const f16x8*>(sb + fb16[0] + j * 2048);
            b1[j] = *reinterpret_cast<const f16x8*>(sb + fb16[1] + j * 2048);
        }
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i) {
            const f16x8 a0 = *reinterpret_cast<const f16x8*>(sb + fa16[0] + i * 2048);
            const f16x8 a1 = *reinterpret_cast<const f16x8*>(sb + fa16[1] + i * 2048);
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) {
                f32x4 c = acc16[i][j];
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0, b0[j], c, 0, 0, 0);
                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[j], c, 0, 0, 0);
                acc16[i][j] = c;
            }
        }
    };
    auto compute = [&](int st) {
        if constexpr (HK) { compute_hk(st); return; }
        if constexpr (P32_SINGLE != 0) { compute1(st); return; }
        const char* sb = smem + st * STAGE;
        if constexpr (M16) {
            f16x8 bh[2 * TN], bl[2 * TN];
            f16x8 fake;                                  // timing-only builds (P32_ABLATE & 1024): no fragment reads
#pragma unroll
            for (int q = 0; q < 8; ++q) fake[q] = (_Float16)(float)(lane + q);
#if P32_FRAG_PIPE != 2
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) {
                bh[j] = (P32_ABLATE & 1024) ? fake : *reinterpret_cast<const f16x8*>(sb + fb16[0] + j * 2048);
                bl[j] = (P32_ABLATE & 1024) ? fake : *reinterpret_cast<const f16x8*>(sb + fb16[1] + j * 2048);
            }
#endif
#if P32_FRAG_PIPE == 2
            // Hand-placed fragment reads.  Left to itself hipcc keeps ONE register set for the A fragments, so every pair of
            // tile-rows opens with `ds_read x 4; s_waitcnt` in front of its MFMAs and the matrix pipe idles for an LDS round
            // trip four times per K-step (both waves of a SIMD at once: they run in lockstep between barriers).  Here the reads
            // are inline asm in program order: the K-step opens with A row 0 and all B fragments; then, per tile-row i, ONE asm
            // block requests the A fragments of row i + 1 into the OTHER register set and waits until only those two reads are
            // outstanding (LDS reads return in order: everything older -- row i, the B fragments -- has landed), and the row's
            // MFMAs follow behind a scheduling fence (hipcc moves register-only MFMAs across an inline `s_waitcnt` otherwise).
            // Request and wait sit in the same asm block, so nothing the compiler emits can come between them.
            const unsigned la0 = (unsigned)(__SIZE_TYPE__)((lds_void*)(smem)) + st * STAGE + fa16[0];
            const unsigned la1 = la0 - fa16[0] + fa16[1];
            const unsigned lb0 = la0 - fa16[0] + fb16[0], lb1 = la0 - fa16[0] + fb16[1];
            f16x8 ahp[2], alp[2];
            ds_read2_b128<0, -1>(ahp[0], alp[0], la0, la1);
            static_for<2 * TN>([&](auto jc) {
                constexpr int j = decltype(jc)::value;
                ds_read2_b128<j * 2048, -1>(bh[j], bl[j], lb0, lb1);
            });
            static_for<2 * TM>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                const f16x8 ah = ahp[i & 1], al = alp[i & 1];
                if constexpr (i == 0) {
                    // the K-step's first row starts as soon as ITS B fragments are there: in front of column block j only the
                    // 2 (2 TN - 1 - j) younger B reads and the two reads of A row 1 may still be outstanding
                    ds_read2_b128<2048, -1>(ahp[1], alp[1], la0, la1);
                    static_for<2 * TN>([&](auto jc) {
                        constexpr int j = decltype(jc)::value;
                        asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(2 * (2 * TN - 1 - j) + 2) : "memory");
                        __builtin_amdgcn_sched_barrier(0);
                        f32x4 c = acc16[0][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], c, 0, 0, 0);
                        acc16[0][j] = c;
                        __builtin_amdgcn_sched_barrier(0);
                    });
                } else {
                    if constexpr (i + 1 < 2 * TM) {
                        ds_read2_b128<(i + 1) * 2048, 2>(ahp[(i + 1) & 1], alp[(i + 1) & 1], la0, la1);
                    } else if constexpr (BARRIER_IN_COMPUTE) {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                    }
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < 2 * TN; ++j) {
                        f32x4 c = acc16[i][j];
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], c, 0, 0, 0);
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], c, 0, 0, 0);
                        acc16[i][j] = c;
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
            });
            return;
#elif P32_FRAG_PIPE
            // A fragments of tile-row i + 1 are requested BEFORE the MFMAs of tile-row i issue (two register sets): the
            // matrix pipe never waits for an LDS round trip inside a K-step except at its first tile-row.  The issue order
            // is pinned with sched_group_barrier (2 LDS reads, then the row's MFMAs): left to itself hipcc reuses ONE register
            // set, so every pair of tile-rows starts with `ds_read x 4; s_waitcnt` in front of its MFMAs.
            f16x8 ahp[2], alp[2];
            ahp[0] = *reinterpret_cast<const f16x8*>(sb + fa16[0]);
            alp[0] = *reinterpret_cast<const f16x8*>(sb + fa16[1]);
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                if (i + 1 < 2 * TM) {
                    ahp[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(sb + fa16[0] + (i + 1) * 2048);
                    alp[(i + 1) & 1] = *reinterpret_cast<const f16x8*>(sb + fa16[1] + (i + 1) * 2048);
                }
                const f16x8 ah = ahp[i & 1], al = alp[i & 1];
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    f32x4 c = acc16[i][j];
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], c, 0, 0, 0);
                    acc16[i][j] = c;
                }
                if (i + 1 < 2 * TM) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 6 * TN, 0);
            }
            return;
#else
#pragma unroll
            for (int i = 0; i < 2 * TM; ++i) {
                const f16x8 ah = (P32_ABLATE & 1024) ? fake : *reinterpret_cast<const f16x8*>(sb + fa16[0] + i * 2048);
                const f16x8 al = (P32_ABLATE & 1024) ? fake : *reinterpret_cast<const f16x8*>(sb + fa16[1] + i * 2048);
#pragma unroll
                for (int j = 0; j < 2 * TN; ++j) {
                    f32x4 c = acc16[i][j];
                    if (P32_ABLATE & 4) {
                        asm volatile("" :: "v"(ah), "v"(al), "v"(bh[j]), "v"(bl[j]));
                        continue;
                    }
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh[j], c, 0, 0, 0);
                    acc16[i][j] = c;
                }
            }
            return;
#endif
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            f16x8 bh[TN], bl[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(sb + fb[kk] + j * 4096);
                bl[j] = *reinterpret_cast<const f16x8*>(sb + fb[2 + kk] + j * 4096);
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const f16x8 ah = *reinterpret_cast<const f16x8*>(sb + fa[kk] + i * 4096);
                const f16x8 al = *reinterpret_cast<const f16x8*>(sb + fa[2 + kk] + i * 4096);
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    f32x16 c = acc[i][j];     // smallest terms first
                    if (P32_ABLATE & 4) {
                        asm volatile("" :: "v"(ah), "v"(al), "v"(bh[j]), "v"(bl[j]));
                        continue;
                    }
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh[j], c, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh[j], c, 0, 0, 0);
                    acc[i][j] = c;
                }
            }
        }
    };

    // ---- K loop: DMA of step t + 1 (NST = 3: also t + 2) in flight under the MFMAs of step t; one barrier per step ----
    // NST = 3 (tiles whose three stages fit the 160 KiB): the request for step t + 2 goes out at step t, so an operand has
    // TWO steps of MFMA time to arrive -- a 1x1 layer touches every line for the first time, and on a 160 x 256 tile one
    // step (~0.6 us) is shorter than an HBM round trip under load.  Loads complete in order, so `s_waitcnt vmcnt(n)` with
    // n = the DMA instructions this wave issued LAST leaves exactly the newest request outstanding.
    constexpr int N_LAST = QB + QA, N_LAST_SHORT = QB + QA - 1;          // per wave and issue (the last A round is partial)
    const bool short_wave = !A_EVEN && wave + NW * (QA - 1) >= NIA;
    auto wait_all_but_last_issue = [&]() {
        if (short_wave) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LAST_SHORT) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N_LAST) : "memory");
    };
    issue(0, tap, sdelta, 0u);
    P32_ADVANCE();
    bool two_ahead = false;
    if (NST == 3 && p.kloop > 1 && !(P32_ABLATE & 128)) {
        issue(1, tap, sdelta, (unsigned)tstep * 8192u);
        P32_ADVANCE();
        two_ahead = true;
    }
    constexpr bool HEAD = EPI == EPI_HEAD;
    const GroupScales gs = load_group_scales(p, m0, !HEAD && !p.out_f32);     // scalar loads, behind the first DMA
    if (two_ahead) wait_all_but_last_issue();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    int st = 0;
    for (int t = 0; t < ((P32_ABLATE & 128) ? 1 : p.kloop); ++t) {
        const int st_issue = NST == 2 ? (st ^ 1) : (st == 0 ? 2 : st - 1);      // (t + NST - 1) % NST
        const bool more = t + NST - 1 < p.kloop;
        if (more) {                               // that stage was last read before the previous barrier
            issue(st_issue, tap, sdelta, (unsigned)tstep * 8192u);
            P32_ADVANCE();
        }
        compute(st);
        // Before the barrier every wave has (a) seen its own DMA pieces of step t + 1 land and (b) got ALL its fragment
        // reads of this stage back: the barrier is what allows the other waves to start refilling the stage, and a read
        // still queued in the LDS when a fast DMA from L2 lands would return the step-after-next's operands.
        if (NST == 3 && more) {
            wait_all_but_last_issue();
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        } else if (!BARRIER_IN_COMPUTE) {
            asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        }
        st = NST == 2 ? (st ^ 1) : (st == 2 ? 0 : st + 1);
    }
#undef P32_ADVANCE

    if constexpr ((P32_ABLATE & 2048) != 0 && M16) {      // timing-only: prologue + K loop with its MFMAs, no epilogue at all
#pragma unroll
        for (int i = 0; i < 2 * TM; ++i)
#pragma unroll
            for (int j = 0; j < 2 * TN; ++j) asm volatile("" :: "v"(acc16[i][j]));
        return;
    }
    if constexpr (EPI == EPI_PLANES) {
        if constexpr (M16) {
            p32_epilogue_planes<WM, WN, TM, TN>(p, gs, smem, [&](int i, float* e) { write_acc16<TN, TN * 32 + 4>(acc16[2 * i], acc16[2 * i + 1], e, lane); }, wm, wn, m0, n0);
        } else {
            p32_epilogue_planes<WM, WN, TM, TN>(p, gs, smem, [&](int i, float* e) { write_acc32<TN, TN * 32 + 4>(acc[i], e, lane); }, wm, wn, m0, n0);
        }
    } else if constexpr (M16 && HEAD && P32_HEAD_DIRECT) {
        // (block-uniform) the layer's own activation is ReLU / none and the head has at most 2 rows (K = 2 classes): straight from
        // the accumulators; anything else through the general epilogue
        if (p.act != DEMIA_ACT_SIGMOID && p.head_n <= 2)
            p32_epilogue_head_direct<WM, WN, TM, TN, 2>(p, gs, smem, acc16, wm, wn, m0, n0);
        else
            p32_epilogue<WM, WN, TM, TN, HEAD>(p, gs, smem, [&](int i, float* e) { write_acc16<TN, BN + 4>(acc16[2 * i], acc16[2 * i + 1], e, lane); }, wm, wn, m0, n0);
    } else if constexpr (M16) {
        p32_epilogue<WM, WN, TM, TN, HEAD>(p, gs, smem, [&](int i, float* e) { write_acc16<TN, BN + 4>(acc16[2 * i], acc16[2 * i + 1], e, lane); }, wm, wn, m0, n0);
    } else {
        p32_epilogue<WM, WN, TM, TN, HEAD>(p, gs, smem, [&](int i, float* e) { write_acc32<TN, BN + 4>(acc[i], e, lane); }, wm, wn, m0, n0);
    }
}

#if P32_DEV_TILES   // experiments that did not pay (ping-pong schedules), kept for same-box A/B through tile hints: a file of their own
#include "conv_p32_dev.inc"
#endif  // P32_DEV_TILES

template <int WM, int WN, int TM, int TN, bool M16 = true, int EPI = EPI_PLANES, int NST = 2, bool HK = false>
int launch_q_(ConvQ p, hipStream_t st) {
    constexpr int BM = WM * TM * 32, BN = WN * TN * 32;
    // LDS: the K-loop stages, overlaid in the epilogue by the accumulator image (planes epilogue: one 32-row block per wave)
    constexpr int stages = NST * (BM + BN) * 128, image = EPI == EPI_PLANES ? WM * WN * 32 * (TN * 128 + 16) : WM * 32 * (BN * 4 + 16);
    static_assert(stages <= 160 * 1024, "LDS");
    constexpr int smem = stages > image ? stages : image;
    p.ntn = p.CoutPad / BN;
    p.nwg = p.ntn * cdiv(p.M, BM);
    p.resident = 256 * (160 * 1024 / smem >= 2 ? 2 : 1);
    p.kloop = HK ? p.ksteps / 2 : p.ksteps;
    auto k = conv_p32_kernel<WM, WN, TM, TN, M16, EPI, NST, HK>;
    static bool attr_done = false;
    if (!attr_done) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        attr_done = true;
    }
    hipLaunchKernelGGL(k, dim3(p.nwg), dim3(WM * WN * 64), smem, st, p);
    DEMIA_CHECK_LAUNCH("conv_p32_kernel");
    return DEMIA_OK;
}
// (single-plane build: layers whose channel count is a multiple of 64 take the h-only K-step of 64 -- half the operand stream)
template <int WM, int WN, int TM, int TN, bool M16 = true, int EPI = EPI_PLANES, int NST = 2>
int launch_q(ConvQ p, hipStream_t st) {
#if P32_SINGLE
    if constexpr (M16 && NST == 2) {
        if (p.Cin % 64 == 0 && !p.no_hk) return launch_q_<WM, WN, TM, TN, M16, EPI, NST, true>(p, st);
    }
#endif
    return launch_q_<WM, WN, TM, TN, M16, EPI, NST, false>(p, st);
}

// Tile choice: predicted launch time of every instantiated tile, from a four-parameter model fitted to a sweep of the
// R101 layer shapes on one MI355X (scripts/gpu_conv_p32_check.py, profiles/r02_conv_p32_tile_sweep.txt): a K-step costs
// 2.0 us x (tile area / 256^2) x (1 + 0.2 x (1 - area)) on a CU of its own, prologue + epilogue 6 + 6 x area us (x 1.5
// with a residual); a launch takes as many rounds as the busiest CU gets tiles; tiles whose two stages fit twice into
// the LDS run two workgroups per CU, which hides half of the prologue / epilogue.  The model picks within 1 % of the
// best measured configuration over the whole network (it ranks tiles; it is not a time estimate for the HBM-bound layers).
struct TileCfg { int id, bm, bn; };
constexpr TileCfg kTiles[] = {{1, 256, 256}, {2, 128, 256}, {4, 192, 256}, {12, 160, 256}, {13, 224, 256}, {6, 256, 128}, {7, 128, 128}, {9, 256, 64}, {11, 128, 64}};

inline double predict_us(const TileCfg& c, long M, int cout_pad, int ksteps, bool residual) {
    const long tiles = (long)cdiv(M, c.bm) * (cout_pad / c.bn);
    const int smem = 2 * (c.bm + c.bn) * 128;
    const int occ = 160 * 1024 / smem >= 2 ? 2 : 1;
    const double area = (double)c.bm * c.bn / 65536.0;
    const double step = 2.0 * area * (1.0 + 0.2 * (1.0 - area));
    const double edge = 6.0 + 6.0 * area * (residual ? 1.5 : 1.0);
    if (occ == 1) return (double)((tiles + 255) / 256) * (ksteps * step + edge);
    return (double)((tiles + 511) / 512) * (2.0 * ksteps * step + 2.0 * 0.5 * edge);
}

inline int choose_tile(long M, int cout_pad, int ksteps, bool residual, bool need256 = false) {
#if P32_SINGLE
    // one MFMA per product: the launch is bound by the operand stream, which the 256 x 256 tile uses best (mask-head 3x3 at 48
    // tiles: 1236 us against 1256-2022 us for the other shapes, gpu_conv_p32_check.py single) -- wherever it fills the chip twice
    if (cout_pad % 256 == 0 && (long)cdiv(M, 256) * (cout_pad / 256) >= 512) return 1;
#endif
    int best = 0;
    double best_t = 1e30;
    for (const TileCfg& c : kTiles) {
        if (cout_pad % c.bn || (need256 && c.bn != 256)) continue;
        const double t = predict_us(c, M, cout_pad, ksteps, residual);
        if (t < best_t) { best_t = t; best = c.id; }
    }
    return best;
}

}  // namespace

#if P32_SINGLE
extern "C" int demia_conv2d_p32_single(const demia_conv_p32_desc* d, void* stream) {
#else
extern "C" int demia_conv2d_p32_single(const demia_conv_p32_desc* d, void* stream);      // conv_p32_single.o: this file with -DP32_SINGLE=1
extern "C" int demia_conv2d_p32(const demia_conv_p32_desc* d, void* stream) {
    if (d && d->single) return demia_conv2d_p32_single(d, stream);                    // one MFMA per product, chosen per launch
#endif
    DEMIA_REQUIRE(d && d->in && d->in_meta && d->w && (d->out || d->head_n > 0), "null pointer");
    DEMIA_REQUIRE(d->out_f32 || d->out_meta || d->head_n > 0, "P32 output needs out_meta");
    DEMIA_REQUIRE(d->Cin > 0 && d->Cin % 32 == 0, "Cin must be a multiple of 32");
    DEMIA_REQUIRE(d->CoutPad >= d->Cout && d->CoutPad % 64 == 0, "CoutPad must be a multiple of 64");
    DEMIA_REQUIRE(d->out_f32 || d->Cout % 32 == 0, "P32 output needs Cout % 32 == 0");
    DEMIA_REQUIRE(d->KH > 0 && d->KW > 0 && d->KH * d->KW <= 32 && d->stride > 0 && d->pad >= 0, "kernel geometry");
    DEMIA_REQUIRE(d->Ho == (d->H + 2 * d->pad - d->KH) / d->stride + 1, "Ho");
    DEMIA_REQUIRE(d->Wo == (d->W + 2 * d->pad - d->KW) / d->stride + 1, "Wo");
    DEMIA_REQUIRE(d->res_mode == DEMIA_RES_NONE || (d->residual && d->res_meta), "residual pointer");
    DEMIA_REQUIRE((long)d->N * d->Ho * d->Wo < (1L << 31), "M overflow");
    const long in_bytes = 128 + (long)d->N * d->H * d->W * d->Cin * 4;
    DEMIA_REQUIRE(in_bytes < (1L << 32), "input planes must stay below 4 GiB");
    const long w_bytes = (long)d->CoutPad * d->KH * d->KW * d->Cin * 4;
    DEMIA_REQUIRE(w_bytes < (1L << 31), "weight planes must stay below 2 GiB");
    ConvQ p;
    p.in = d->in; p.in_meta = d->in_meta; p.w = d->w; p.scale = d->scale; p.bias = d->bias;
    p.res = d->residual; p.res_meta = d->res_meta; p.out = d->out; p.out_meta = d->out_meta;
    p.wbound = d->wbound; p.bbound = d->bbound;
    p.N = d->N; p.H = d->H; p.W = d->W; p.Cin = d->Cin; p.Ho = d->Ho; p.Wo = d->Wo; p.Cout = d->Cout; p.CoutPad = d->CoutPad;
    p.KH = d->KH; p.KW = d->KW; p.stride = d->stride; p.pad = d->pad;
    p.act = d->act; p.res_mode = d->res_mode; p.out_f32 = d->out_f32;
    p.out_ld = d->out_ld > 0 ? d->out_ld : d->Cout;
    DEMIA_REQUIRE(p.out_ld >= d->Cout, "out_ld");
    p.M = d->N * d->Ho * d->Wo;
    p.HoWo = d->Ho * d->Wo;
    p.taps = d->KH * d->KW;
    p.ksteps = p.taps * (d->Cin / 32);
    p.in_bytes = (unsigned)in_bytes; p.w_bytes = (unsigned)w_bytes;
    p.ntn = p.nwg = 0;
    p.zero_low = d->single == 2;
    {
        static const char* env = getenv("DEMIA_P32_NO_HK");
        p.no_hk = (env && env[0] == '1') ? 1 : 0;
    }
    p.kloop = 0;
    p.resident = 256;
    p.groups = d->groups > 1 ? d->groups : 1;
    p.group_rows = p.groups > 1 ? d->group_rows : (1 << 29);
    p.row0 = p.groups > 1 ? d->row0 : 0;
    DEMIA_REQUIRE(p.groups == 1 || (d->group_rows >= 128 && d->row0 >= 0 && ((long)p.M + d->row0 + d->group_rows - 1) / d->group_rows <= p.groups),
                  "scale groups: group_rows >= 128 and the rows of this call must fall inside `groups` groups");
    p.head_w = d->head_w; p.head_b = d->head_b; p.head_out = d->head_out; p.head_n = d->head_n; p.head_ld = d->head_ld; p.head_act = d->head_act;
    if (d->head_n > 0) {
        DEMIA_REQUIRE(d->head_n <= 4 && d->head_w && d->head_b && d->head_out && d->head_ld >= d->head_n, "fused head: at most 4 rows, pointers, head_ld");
        DEMIA_REQUIRE(d->CoutPad % 256 == 0 && d->Cout == d->CoutPad && !d->out_f32, "fused head needs Cout % 256 == 0 and a planes layer");
    }
    p.stagger_ticks = 0;
#if P32_DEV_TILES
    {
        static const char* env = getenv("DEMIA_P32_STAGGER_US");      // experiment switch (microseconds), dev build only
        p.stagger_ticks = env ? atoi(env) * 100 : 0;
    }
#endif
    if (p.M == 0) return DEMIA_OK;
    hipStream_t st = reinterpret_cast<hipStream_t>(stream);
    // tile_hint: 0 = auto (the model above), else one of the instantiated tiles (dev / tuning: scripts/gpu_conv_p32_check.py)
    int tile = d->tile_hint ? d->tile_hint : choose_tile(p.M, d->CoutPad, p.ksteps, d->res_mode != DEMIA_RES_NONE, d->head_n > 0);
    const bool n256 = d->CoutPad % 256 == 0, n128 = d->CoutPad % 128 == 0;
    if (d->head_n > 0) {
        // the fused head has its own instantiations: 256 x 256, 128 x 256 and 192 x 256
        switch (tile) {
            case 1: return launch_q<2, 4, 4, 2, true, EPI_HEAD>(p, st);
            case 2: return launch_q<2, 4, 2, 2, true, EPI_HEAD>(p, st);
            default: return launch_q<1, 8, 6, 1, true, EPI_HEAD>(p, st);
        }
    }
    // The straight-line planes epilogue needs whole tiles of channels (Cout % BN == 0) and buffers below 4 GiB; anything else
    // (f32 outputs, odd channel counts) runs the guarded epilogue, instantiated for the 128 x 128 and the 64-wide tiles only.
    const int tile_bn = (tile == 9 || tile == 10 || tile == 11 || tile == 49) ? 64 : ((tile >= 6 && tile <= 8) || tile == 26 ? 128 : 256);
    const long out_bytes = 128 + (long)p.M * d->Cout * 4;
    long res_bytes = 0;
    if (d->res_mode == DEMIA_RES_SAME) res_bytes = out_bytes;
    if (d->res_mode == DEMIA_RES_UP2) res_bytes = 128 + (long)d->N * ((d->Ho + 1) / 2) * ((d->Wo + 1) / 2) * d->Cout * 4;
    const bool planes = !d->out_f32 && d->Cout % tile_bn == 0 && out_bytes < (1L << 32) - 512 && res_bytes < (1L << 32) - 512 &&
                        (long)(p.M + 256) * d->Cout * 4 + 256 < (1L << 32);
    p.out_bytes = (unsigned)out_bytes; p.res_bytes = (unsigned)res_bytes;
#if !P32_SINGLE && P32_DEV_TILES
    if (tile == 60) {
        DEMIA_REQUIRE(!d->out_f32 && n128 && d->Cout % 128 == 0 && d->KH == 1 && d->KW == 1 && d->stride == 1 && d->pad == 0 &&
                      d->res_mode != DEMIA_RES_UP2 && out_bytes < (1L << 32) - 512 && res_bytes < (1L << 32) - 512 &&
                      (long)(p.M + 256) * d->Cout * 4 + 256 < (1L << 32),
                      "tile 60 (ping-pong) takes 1x1 stride-1 planes layers with Cout % 128 == 0, no nearest-2x residual, buffers below 4 GiB");
        return launch_pp1x1(p, st);
    }
#endif
    if (!planes) {
        if (tile != 7 && tile != 9 && tile != 10 && tile != 11 && tile != 21 && tile != 22 && tile != 26) tile = n128 ? 7 : 11;
        switch (tile) {
            case 7: DEMIA_REQUIRE(n128, "tile needs CoutPad % 128 == 0"); return launch_q<4, 2, 1, 2, true, EPI_GENERIC>(p, st);
            case 9: return launch_q<8, 1, 1, 2, true, EPI_GENERIC>(p, st);
            case 10: return launch_q<4, 2, 2, 1, true, EPI_GENERIC>(p, st);
            case 11: return launch_q<4, 2, 1, 1, true, EPI_GENERIC>(p, st);
            default: break;       // the ping-pong kernels keep the guarded epilogue
        }
    }
    switch (tile) {
        case 1: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<2, 4, 4, 2>(p, st);   // 256 x 256
        case 2: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<2, 4, 2, 2>(p, st);   // 128 x 256
        case 4: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 6, 1>(p, st);   // 192 x 256
        case 6: DEMIA_REQUIRE(n128, "tile needs CoutPad % 128 == 0"); return launch_q<4, 2, 2, 2>(p, st);   // 256 x 128
        case 7: DEMIA_REQUIRE(n128, "tile needs CoutPad % 128 == 0"); return launch_q<4, 2, 1, 2>(p, st);   // 128 x 128
        case 9: return launch_q<8, 1, 1, 2>(p, st);                                                           // 256 x 64
        case 11: return launch_q<4, 2, 1, 1>(p, st);                                                          // 128 x 64
        case 12: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 5, 1>(p, st);   // 160 x 256
        case 13: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 7, 1>(p, st);   // 224 x 256
        // (four waves of 128 x 128 / 64 x 128 -- `launch_q<2, 2, 4, 4>`, `<2, 2, 2, 4>`, the kernel supports WM * WN == 4 -- were
        //  measured 14-20 % SLOWER than the eight-wave tiles: with one wave per SIMD nothing hides the fragment-read latency
        //  unless the reads are interleaved with the MFMAs by hand; not instantiated)
        // three LDS stages (request two K-steps ahead): measured within +-3 % of the two-stage tiles on every R101 layer --
        // the K loop of the short-K layers is bound by LDS bandwidth (every wave of a 1 x 8 wave grid reads ALL A rows), not
        // by how far ahead the operands are requested; kept as tile hints for A/B only
#if P32_DEV_TILES
        case 3: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 8, 1>(p, st);   // 256 x 256, waves along N
        case 5: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 4, 1>(p, st);   // 128 x 256, waves along N
        case 8: DEMIA_REQUIRE(n128, "tile needs CoutPad % 128 == 0"); return launch_q<2, 4, 2, 1>(p, st);   // 128 x 128, waves along N
        case 10: return launch_q<4, 2, 2, 1>(p, st);                                                          // 256 x 64, two waves along N
        case 14: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<2, 4, 3, 2>(p, st);   // 192 x 256, 2 x 4 waves
        case 42: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<2, 4, 2, 2, true, EPI_PLANES, 3>(p, st);   // 128 x 256
        case 52: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 5, 1, true, EPI_PLANES, 3>(p, st);   // 160 x 256
        case 49: return launch_q<8, 1, 1, 2, true, EPI_PLANES, 3>(p, st);                                                           // 256 x 64
        case 31: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<2, 4, 4, 2, false>(p, st);  // 256 x 256, 32x32x16 MFMAs (A/B)
        case 34: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_q<1, 8, 6, 1, false>(p, st);  // 192 x 256, 32x32x16 MFMAs (A/B)
        case 21: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_pp<4, 2>(p, st);        // 256 x 256, ping-pong
        case 22: DEMIA_REQUIRE(n256, "tile needs CoutPad % 256 == 0"); return launch_pp<2, 2>(p, st);        // 128 x 256, ping-pong
        case 26: DEMIA_REQUIRE(n128, "tile needs CoutPad % 128 == 0"); return launch_pp<4, 1>(p, st);        // 256 x 128, ping-pong
#endif
        default: DEMIA_REQUIRE(false, "tile_hint");
    }
    return DEMIA_EINVAL;
}
