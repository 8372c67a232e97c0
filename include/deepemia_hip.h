/*
 * deepemia_hip.h -- C ABI of libdeepemia_hip.so (gfx950 / MI355X only).
 *
 * This is the drop-in boundary for ONE path of Deam0on/deepEMIA: the
 * Detectron2 `DefaultPredictor.__call__` that the reference invokes at
 *   src/functions/inference.py:1395, 1398, 1507, 1669   (predictor(image))
 * built by src/data/models.py:103-107 (load_model -> DefaultPredictor(cfg)),
 * plus the per-mask post-processing / dedup / measurement reductions that the
 * reference runs on the host afterwards.
 *
 * Conventions
 *   - every entry point is `extern "C"`, takes raw DEVICE pointers + sizes +
 *     a `hipStream_t` (passed as void*), and returns 0 on success or a negative
 *     DEMIA_E* code; no exceptions cross the ABI, nothing is allocated or freed
 *     behind the caller's back, no ownership is transferred;
 *   - all launches are asynchronous on the given stream and graph-capturable
 *     (no hipMalloc / hipFree / sync inside);
 *   - tensors are NHWC ("pixel-major, channel-minor"); `dtype` is 0 = f32,
 *     1 = bf16;
 *   - re-entrant per stream; one host thread per GPU.
 */
#ifndef DEEPEMIA_HIP_H
#define DEEPEMIA_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DEMIA_OK 0
#define DEMIA_EINVAL (-1)   /* bad shape / unsupported combination            */
#define DEMIA_ELAUNCH (-2)  /* hipLaunch / runtime error (see demia_last_error) */

#define DEMIA_F32 0
#define DEMIA_BF16 1
#define DEMIA_F32X3 2   /* conv only: f32 activations, weights pre-split into 3 bf16 planes, tiled (see demia_conv2d_nhwc) */
#define DEMIA_BF16X2 3  /* conv only: f32 activations, weights as 2 bf16 planes (same tiling): 16-bit operands */
#define DEMIA_F16X2 4   /* conv only: f32 activations, weights as 2 fp16 planes (same tiling), power-of-two operand scales */
#define DEMIA_P32 5     /* activations as two pre-scaled fp16 planes, [pixels][C / 32][2][32] behind a 128-byte zero header
                           (demia_conv2d_p32 below); taken by demia_roi_align and demia_maxpool3x3s2_p32 */

#define DEMIA_ACT_NONE 0
#define DEMIA_ACT_RELU 1
#define DEMIA_ACT_SIGMOID 2

#define DEMIA_RES_NONE 0
#define DEMIA_RES_SAME 1      /* residual has the output's shape                      */
#define DEMIA_RES_UP2 2       /* residual is (Ho/2 rounded up, Wo/2 rounded up): nearest x2  */

/* library / device info ------------------------------------------------------------*/
int demia_abi_version(void);
const char* demia_last_error(void);
const char* demia_build_arch(void);   /* "gfx950" */
/* "product": what the default path runs -- the P32 conv kernel (f16x2 / f16), the exact-f32 conv kernel (the parity suite's control
 * arithmetic), every non-conv kernel.  "dev" (`make DEV=1` -> libdeepemia_hip_dev.so, loaded with DEEPEMIA_DEV_LIB=1): also the
 * kernels of the non-default precisions behind demia_conv2d_nhwc (f32x3, bf16x2, f16x2r, bf16) and the experimental tiles /
 * schedules of the P32 kernel reachable through tile hints.  Same symbols in both. */
const char* demia_build_flavor(void);
/* A HIP stream restricted to the compute units whose bit is set in mask (bit i of word i / 32; words = 8 for the 256 CUs of an
 * MI355X) -- hipExtStreamCreateWithCUMask behind the C ABI, so that the host side (which holds streams as torch objects) can give
 * the network's stream a mask that leaves a few CUs per XCD to the latency-bound post-processing kernels of the images in flight.
 * The caller owns the stream: demia_stream_destroy.  Replaces nothing in the reference (its loop is sequential,
 * inference.py:713-942). */
int demia_stream_create_cu_mask(const uint32_t* mask, int words, void** stream);
int demia_stream_destroy(void* stream);
/* Single-plane arithmetic is chosen PER LAUNCH (`single` of demia_conv_p32_desc / demia_roialign_desc, the `single` argument of
 * the stem / pool entry points), never process-wide: engines of different precisions may share the library.
 * (demia_conv2d_p32_single is the conv entry point demia_conv2d_p32 forwards to when d->single != 0: the same source compiled
 * with one MFMA per product; same descriptor.) */
struct demia_conv_p32_desc;
int demia_conv2d_p32_single(const struct demia_conv_p32_desc* d, void* stream);

/* a3: convolution as implicit GEMM on MFMA -------------------------------------------
 * Replaces every Conv2d(+FrozenBatchNorm)(+ReLU)(+residual add) / Linear /
 * ConvTranspose2d(k2,s2) that Detectron2's GeneralizedRCNN runs inside
 * predictor(image) (inference.py:1395).
 *   in        [N, H, W, Cin]        dtype `dtype`
 *   w         [CoutPad, KH, KW, Cin] dtype `dtype`, CoutPad = multiple of 32 >= Cout
 *   scale/bias [Cout] f32 or NULL    y = acc * scale + bias   (FrozenBN folded / conv bias)
 *   residual  NULL or out-shaped (RES_SAME) / half-res (RES_UP2), dtype `out_dtype`
 *   out       [N, Ho, Wo, Cout]      dtype `out_dtype`
 * Cin must be a multiple of 64 (bf16) / 32 (f32, f32x3, f16x2, bf16x2).  dtype DEMIA_F32X3 computes the f32 product on the bf16
 * matrix pipe from three-way split operands (six bf16 MFMAs per f32 FMA tile, error ~ one f32 rounding): `in` is
 * f32, `w` holds the three bf16 planes of the f32 weights, CoutPad % 64 == 0, output f32.  DEMIA_BF16X2 is the same
 * with two planes and three MFMAs (16 significand bits per operand, error <= 3 * 2^-16 per product).
 * Layout of `w` for these two dtypes (NP = 3 / 2 planes, x = x1 + x2 + x3 with x1 = bf16(x), x2 = bf16(x - x1), ...):
 *   [CoutPad / 64][KH*KW*Cin / 32][NP][64][32] bf16,  K = (kh, kw, cin) walked in steps of 32
 * i.e. the 64-channel x 32-k piece of one plane that a K-step reads is 4 KiB contiguous (whole 128-byte lines per
 * wave load).  ABI version 3 (version 2 took row-major [NP][CoutPad][K] planes).
 * DEMIA_F16X2: two fp16 planes per operand (x1 = half(x), x2 = half(x - x1): 22 significand bits, error <= 3 * 2^-22
 * per product -- an f32-sized error from three MFMAs).  fp16 has five exponent bits, so the caller brings the weights
 * into range with an exact power of two per output channel BEFORE splitting them (and divides `scale` by it), and
 * the kernel scales the activations by 2^(13 - ilogb(*amax_in)); `amax_in` is a device scalar holding an upper bound
 * of |in| (NULL: no scaling, |in| must stay below 6e4).  `amax_out` (any dtype, may be NULL): device scalar into
 * which the kernel accumulates max |out| with an atomic max -- zero it before the launch; it is the next layer's
 * `amax_in`.  DEMIA_F16X2 takes its planes in the same tiling, [CoutPad / 64][K / 32][2][64][32] fp16.  */
typedef struct demia_conv_desc {
    const void* in;
    const void* w;
    const float* scale;
    const float* bias;
    const void* residual;
    void* out;
    int32_t N, H, W, Cin;
    int32_t Ho, Wo, Cout, CoutPad;
    int32_t KH, KW, stride, pad;
    int32_t dtype, out_dtype;
    int32_t act, res_mode;
    int32_t out_ld;          /* elements between consecutive output pixels (>= Cout); 0 -> Cout */
    int32_t tile_hint;       /* 0 = auto; else BN in {128, 64, 32} */
    const float* amax_in;    /* DEMIA_F16X2: device scalar, upper bound of |in| (see above); else ignored */
    float* amax_out;         /* NULL or device scalar: max |out| is accumulated (atomic max) */
} demia_conv_desc;
int demia_conv2d_nhwc(const demia_conv_desc* d, void* stream);
/* K-step of this build's DEMIA_F16X2 kernel = innermost extent of its weight-plane tiling (32 unless built otherwise) */
int demia_conv_f16x2_kstep(void);

/* a3: the same convolution with BOTH operands pre-split into fp16 planes ("P32" activations) -------------------
 * The default arithmetic of the product path (precision "f16x2"): f32-sized error from three fp16 MFMAs per
 * product, operands moved global -> LDS by LDS-DMA.  Replaces the same Detectron2 layers as demia_conv2d_nhwc.
 *   P32 activation buffer: 128 zero bytes, then [pixels][C / 32][2][32] fp16 -- per pixel and 32-channel group one
 *       128-byte line of 32 high halves + 32 low halves of x * s (s = meta[1], an exact power of two; x = (h + l) / s).
 *       The caller zeroes the 128-byte header once; padding taps read it.
 *   meta: 2 device floats per tensor: [0] = max |x| (atomic max, accumulated by the producing kernel; the caller zeroes
 *       it before every forward), [1] = s (written by the producing kernel).
 *   w: [CoutPad / 64][ksteps][64][2][32] fp16, ksteps = (Cin / 32) * KH * KW walked channel-group OUTER, tap inner;
 *       planes of w * 2^e(co) (max |.| in [2^14, 2^15) per output channel, the caller divides `scale` by 2^e(co)).
 *   wbound = max_co(|scale_co| * sum_k |w_co,k|) (true weights), bbound = max |bias|: the kernel scales its P32 output
 *       with s = 2^(14 - ilogb(in_meta[0] * wbound + bbound + res_meta[0])), so that |out * s| < 2^15.
 *   out: P32 (out_f32 = 0; Cout % 32 == 0; out_meta required) or plain f32 [M, out_ld] (out_f32 = 1).
 *   residual: P32 of the output's shape (RES_SAME) or half resolution (RES_UP2), with its meta.               */
typedef struct demia_conv_p32_desc {
    const void* in;
    const float* in_meta;
    const void* w;
    const float* scale;
    const float* bias;
    const void* residual;
    const float* res_meta;
    void* out;
    float* out_meta;
    float wbound, bbound;
    int32_t N, H, W, Cin;
    int32_t Ho, Wo, Cout, CoutPad;
    int32_t KH, KW, stride, pad;
    int32_t act, res_mode;
    int32_t out_f32;
    int32_t out_ld;          /* f32 output: elements between consecutive output pixels (>= Cout); 0 -> Cout */
    int32_t tile_hint;       /* 0 = auto (a cost model fitted on MI355X picks the tile); else the id of an instantiated tile (tuning) */
    /* Optional fused 1x1 head (head_n in 1..4, Cout % 256 == 0): instead of writing the P32 output, every 256-channel block
     * b of an activated output row m is multiplied by head_w [head_n][256] (+ head_b, then head_act) and stored as f32 at
     * head_out[(m * (Cout / 256) + b) * head_ld + j].  Used for the mask head: ConvTranspose2d(k2, s2) as a GEMM onto
     * (dy, dx, co) + ReLU, then the class predictor + sigmoid -- the 4 x 256-channel deconv output never reaches HBM.   */
    const float* head_w;
    const float* head_b;
    float* head_out;
    int32_t head_n, head_ld, head_act;
    /* Scale groups.  groups <= 1: one {max |x|, s} pair per tensor (in_meta, res_meta, out_meta point at 2 floats).
     * groups > 1: the metas are [groups][2] and output row m of this call belongs to group (m + row0) / group_rows -- one
     * group per IMAGE of the batch, so that the planes (and therefore the results) of an image do not depend on what
     * else is in the batch.  group_rows >= 128; row0 is the global index of this call's first output row when a tensor
     * goes through in several calls.                                                                                   */
    int32_t groups, group_rows, row0;
    /* 0: f16x2 -- three MFMAs per product (the default, f32-sized error).  1: ONE MFMA per product on the high planes of both
     * operands (fp16 operands, f32 accumulation: the arithmetic of the reference's autocast predictor, inference.py:1390-1395),
     * output still split into both planes -- the per-stage precision map of DESIGN.md section 7 runs single stages this way.
     * 2: as 1, and the output's low plane is written as zeros (`--precision f16`, flagged NON-parity, every layer).          */
    int32_t single;
} demia_conv_p32_desc;
int demia_conv2d_p32(const demia_conv_p32_desc* d, void* stream);

/* a3: Pillow-exact ResizeShortestEdge + (x - mean) + zero pad -------------------------
 * Replaces T.ResizeShortestEdge.get_transform(img).apply_image(img) +
 * normalisation + ImageList.from_tensors inside DefaultPredictor.__call__ /
 * GeneralizedRCNN.preprocess_image (Detectron2 0.6), called from inference.py:1395.
 * Two separable 8-bit fixed-point passes (22-bit coefficients) identical to
 * Pillow's ImagingResample.  Coefficient tables are computed by the host.
 *   src     [N, H, W, 3] u8 (BGR)         tmp [N, H, newW, 3] u8
 *   xmin/xsize [newW] i32, xk [newW, ksx] i32; same for y
 *   dst     [N, PH + 6, PW + 8, 4] `dtype`: interior at (+3, +3) holds the
 *           normalised image, channel 3 and every border / pad element = 0.   */
int demia_resize_h_u8(const uint8_t* src, uint8_t* tmp, int N, int H, int W, int newW,
                      const int32_t* xmin, const int32_t* xsize, const int32_t* xk, int ksx, void* stream);
int demia_resize_v_norm(const uint8_t* tmp, void* dst, int N, int H, int newW, int newH, int PH, int PW,
                        const int32_t* ymin, const int32_t* ysize, const int32_t* yk, int ksy,
                        const float* mean3, int dtype, void* stream);

/* a2: tile upscale, cv2.resize(tile, (w*f, h*f), INTER_LINEAR) on u8 BGR (inference.py:2379-2382) ------
 * OpenCV's fixed-point bilinear: xofs/yofs [out, 2] i32 source indices, ialpha/ibeta [out, 2] i16
 * 11-bit coefficients (host tables, deepemia_amd.engine.cv_linear_tables). src [N,H,W,3] -> dst [N,oh,ow,3] */
int demia_resize_linear_u8(const uint8_t* src, uint8_t* dst, int N, int H, int W, int out_h, int out_w,
                           const int32_t* xofs, const int16_t* ialpha, const int32_t* yofs, const int16_t* ibeta,
                           void* stream);

/* a3: ResNet stem: conv 7x7 s2 p3 (3->64) + FrozenBN + ReLU, then maxpool 3x3 s2 p1 ----
 *   in   [N, PH + 6, PW + 8, 4] f32 (see above)    w [7, 8, 4, 64] f32 (kh, kw, c, co; kw 7 and c 3 zero)
 *   mid  [N, PH/2, PW/2, 64]   out [N, PH/4, PW/4, 64]                              */
int demia_stem_conv(const void* in, const void* w, const float* scale, const float* bias, void* mid,
                    int N, int PH, int PW, int dtype, void* stream);
/* The same stem (7x7 s2 p3 + FrozenBN + ReLU, f32 NHWC out) on the matrix pipe in the f16x2 arithmetic: the input is split
 * into two fp16 planes of x * s_in (s_in an exact power of two with |x| s_in < 2^15) while it is staged, w_planes
 * [2][7][64][32] fp16 hold w * 2^e(co) with K ordered (kh, kw padded to 8, c padded to 4), and `scale` carries the FrozenBN
 * scale divided by s_in * 2^e(co).  Replaces the same Detectron2 BasicStem.conv1 as demia_stem_conv. */
int demia_stem_conv_mfma(const float* in, const void* w_planes, const float* scale, const float* bias, float* mid, int N,
                         int PH, int PW, float s_in, void* stream);
/* ... and stem + max pool (3x3 s2 p1) in ONE kernel, P32 planes out: the f32 stem output is never written.  out / out_meta /
 * s_out / groups as demia_maxpool3x3s2_p32 (s_out = the planes' power-of-two scale from the a-priori bound of the stem output). */
int demia_stem_pool_mfma(const float* in, const void* w_planes, const float* scale, const float* bias, void* out, float* out_meta,
                         int N, int PH, int PW, float s_in, float s_out, int groups, int single, void* stream);
int demia_maxpool3x3s2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);
/* the same pool from an f32 input into a P32 buffer scaled with the power of two `s` (the caller derives it from the
 * stem's a-priori bound); out_meta receives {max |out| (atomic max; zero it first), s} -- one pair (groups <= 1) or one
 * per image (groups == N, see demia_conv_p32_desc).  C % 32 == 0.  single != 0: the low plane is written as zeros
 * (`--precision f16`, as in demia_conv_p32_desc.single = 2). */
int demia_maxpool3x3s2_p32(const float* in, void* out, float* out_meta, float s, int N, int H, int W, int C, int groups, int single,
                           void* stream);
/* LastLevelMaxPool (kernel 1, stride 2): p6 = p5[:, ::2, ::2, :] */
int demia_subsample2(const void* in, void* out, int N, int H, int W, int C, int dtype, void* stream);

/* a3: RPN proposal selection --------------------------------------------------------
 * Replaces RPN.predict_proposals / find_top_rpn_proposals (Detectron2 0.6):
 * per image and level the top `pre_topk` objectness logits in STABLE descending
 * order (ties: lower flattened (h, w, a) index first), anchor generation + delta
 * decode (weights 1,1,1,1, clamp log(1000/16)), clip to the image, drop non-finite /
 * empty boxes, per-level hard NMS (IoU > thresh suppresses), then the `post_topk`
 * best survivors over all levels in stable descending score order.
 *   head[l]   [N, H_l, W_l, head_ld] f32: channels 0..2 = logits (a), 3..14 = deltas (a*4 + c)
 *   out_boxes [N, post_topk, 4] f32, out_scores [N, post_topk] f32, out_count [N] i32
 *   workspace: demia_rpn_workspace_bytes(N) bytes                                      */
typedef struct demia_rpn_desc {
    const float* head[5];
    int32_t H[5], W[5], stride[5];
    const float* cell_anchors;   /* HOST pointer, [5][3][4] f32: (x1,y1,x2,y2) of anchor a at the origin */
    int32_t head_ld;
    int32_t N;
    int32_t img_h, img_w;        /* unpadded network-input size boxes are clipped to */
    int32_t pre_topk, post_topk; /* <= 1024 each */
    float nms_thresh;
    float* out_boxes;
    float* out_scores;
    int32_t* out_count;
    void* workspace;
} demia_rpn_desc;
int64_t demia_rpn_workspace_bytes(int N);
int demia_rpn_proposals(const demia_rpn_desc* d, void* stream);

/* a3: ROIAlignV2 (aligned, sampling_ratio 0 = adaptive) over p2..p5 -------------------
 * Replaces ROIPooler.forward -> torchvision.ops.roi_align(aligned=True) incl. the
 * FPN level assignment floor(4 + log2(sqrt(area)/224 + 1e-8)) clamped to [2, 5].
 *   feat[l] [N, H_l, W_l, C] `dtype`; boxes [N, R, 4] f32 (network-input coords);
 *   count [N] i32 (rows >= count[n] are written as zeros); out [N, R, P, P, C] `dtype`
 * dtype DEMIA_P32: feat[l] / out are P32 buffers (pointers to their 128-byte headers); taps and averages are convex
 * combinations, so the output takes the coarsest of the levels' scales and the largest of their max |x|. */
typedef struct demia_roialign_desc {
    const void* feat[4];
    int32_t H[4], W[4];
    int32_t N, R, C, P, dtype;
    const float* boxes;
    const int32_t* count;
    void* out;
    const float* meta[4];    /* dtype DEMIA_P32: {max |x|, s} of each level (device floats); else ignored */
    float* out_meta;         /* dtype DEMIA_P32: receives {max over the levels' max |x|, min over the levels' s} */
    int32_t groups;          /* dtype DEMIA_P32: <= 1 = one meta pair per tensor; N = metas are [N][2], one scale group per image */
    int32_t single;          /* dtype DEMIA_P32: != 0 writes a zero low plane (`--precision f16`) */
    const int32_t* order;    /* NULL, or [N * R] from demia_roi_order: workgroup b pools ROI order[b] (the output row of a ROI stays
                              * where it is: same results, neighbouring workgroups share feature lines in one XCD's L2) */
} demia_roialign_desc;
int demia_roi_align(const demia_roialign_desc* d, void* stream);
/* The launch order for demia_roi_align: per image its R (<= 1024) ROIs sorted by (FPN level, row band, column) of the box centre,
 * dealt to the eight XCDs in runs of R / 8.  boxes [N, R, 4] f32 (network-input coords), count [N]; order [N * R] i32 out. */
int demia_roi_order(const float* boxes, const int32_t* count, int N, int R, int32_t* order, void* stream);

/* a3: Fast R-CNN inference ---------------------------------------------------------
 * Replaces FastRCNNOutputLayers.inference / fast_rcnn_inference_single_image:
 * softmax over K+1 logits, class-specific delta decode (weights 10,10,5,5), clip,
 * score > thresh (strict), candidates in row-major (proposal, class) order, stable
 * descending sort, per-class hard NMS (IoU > 0.5), first `topk` survivors.
 *   logits [N, R, ld] f32: channels 0..K = class logits, K+1 .. K+4K = deltas
 *   props  [N, R, 4] f32, prop_count [N] i32
 *   det_boxes [N, topk, 4] f32 (network-input coords), det_scores [N, topk] f32,
 *   det_classes [N, topk] i32, det_count [N] i32;  topk <= 128; any number of classes K as long as
 *   R * min(K, floor(1 / score_thresh)) <= 4096 (a softmax row has at most floor(1 / thresh) scores above thresh;
 *   R = 1000: every K for thresholds above 0.2, K <= 4 below)                                            */
typedef struct demia_dets_desc {
    const float* logits;
    int32_t ld;
    const float* props;
    const int32_t* prop_count;
    int32_t N, R, K;
    int32_t img_h, img_w;
    float score_thresh, nms_thresh;
    int32_t topk;
    float* det_boxes;
    float* det_scores;
    int32_t* det_classes;
    int32_t* det_count;
} demia_dets_desc;
int demia_box_detections(const demia_dets_desc* d, void* stream);

/* a3: mask paste (paste_masks_in_image / _do_paste_mask, Detectron2 0.6) --------------
 * Boxes are rescaled to the original image (scale_x = out_w / img_w ...), clipped,
 * empty boxes flagged; each 28x28 probability map is resampled with bilinear
 * grid_sample(align_corners=False, zero padding) and thresholded (>= 0.5).
 *   mask_logit_or_prob [N*D, 196, 4, ld] f32 -- deconv-blocked layout: pixel (2y+dy, 2x+dx)
 *        of instance i lives at [(i*196 + y*14 + x), dy*2+dx, class]; already sigmoid-ed
 *   det_boxes [N, D, 4] f32 network-input coords, det_classes, det_count as above
 *   out_boxes [N, D, 4] f32 output-image coords (clipped); valid [N, D] u8 (nonempty)
 *   packed    [N, D, out_h, ceil(out_w/32)] u32 : bit (x & 31) of word x>>5 = mask[y][x]; the padding bits of
 *             a partial last word are always 0.  Every packed-mask entry point takes the TRUE width W.   */
typedef struct demia_paste_desc {
    const float* mask_prob;
    int32_t ld;
    const float* det_boxes;
    const int32_t* det_classes;
    const int32_t* det_count;
    int32_t N, D;
    int32_t img_h, img_w;     /* network-input size */
    int32_t out_h, out_w;     /* original image size */
    float* out_boxes;
    uint8_t* valid;
    uint32_t* packed;
    int32_t* out_bbox;        /* optional [N, D, 4] i32: y0, x0, y1, x1 (inclusive) of the pixels the paste can set,
                                 -1 for invalid instances -- the bbox hint of the packed-mask entry points */
    const int32_t* prev_bbox; /* optional [N, D, 4] i32 (incremental paste): `packed` is known to be ZERO outside these boxes
                                 (the out_bbox of the previous paste into the same buffer; all -1 after a memset) -- only the
                                 union of each instance's old and new box is written instead of whole 512-KiB planes.
                                 Must not alias out_bbox; needs out_bbox. */
} demia_paste_desc;
int demia_paste_masks(const demia_paste_desc* d, void* stream);
/* packed bits -> Detectron2's (M, H, W) bool bytes (pred_masks drop-in layout) */
int demia_unpack_masks(const uint32_t* packed, uint8_t* out_bool, int64_t M, int H, int W, void* stream);
/* per-mask popcount ("np.sum(mask)", inference.py:1688, 2599) and tight bbox [M,4] = y0,x0,y1,x1 (inclusive; -1 if empty).
 * hint (optional, [M,4]): a box known to contain every set pixel of the mask; only that region is read. */
int demia_mask_area_bbox(const uint32_t* packed, const int32_t* hint, int32_t* area, int32_t* bbox, int64_t M, int H, int W,
                         void* stream);

/* a9-a15: packed-mask morphology (all masks [M, H, ceil(W/32)] u32, W = true pixel width) ---------------
 * Replace, on bit-packed device masks, what the reference does with scipy / scikit-image / numpy on
 * dense host arrays.  demia_mask_program runs a sequence of per-mask stages IN PLACE on the bbox region
 * of every mask (one workgroup per mask, region staged in LDS; the frame outside the box is not touched):
 *   DEMIA_MOP_FILL        scipy.ndimage.binary_fill_holes        (mask_utils.py:75; inference.py:193, 1780)
 *   DEMIA_MOP_DILATE /    skimage dilation / erosion, 3x3 cross, 'reflect' border
 *   DEMIA_MOP_ERODE                                              (mask_utils.py:76; inference.py:196-198, 1786-1796)
 *   DEMIA_MOP_DROP_MULTI  if skimage.measure.label(mask).max() > 1 (8-connected): mask[:] = 0
 *                                                                (mask_utils.py:79-81)
 *   DEMIA_MOP_FLAG_MULTI  the same test without changing the mask
 *   DEMIA_MOP_GATE        masks with active[m] == 0 (or active == NULL) stop here (inference.py:1443: the
 *                         fill -> erosion -> dilation of process_masks_parallel only runs for calls with > 2 masks)
 * program = up to 8 stage codes, 4 bits each, low nibble first, 0 ends it.
 *   bbox     [M, 4] i32 y0, x0, y1, x1 inclusive, -1 for empty masks; a SUPERSET of the tight box is fine
 *            (demia_mask_area_bbox, demia_paste_desc.out_bbox, or bbox_out of an earlier program)
 *   scratch  same shape as masks; only regions larger than 64 KiB of LDS use it
 *   area / bbox_out / flag (each optional): popcount and tight bbox of the result; flag = a *_MULTI stage fired.
 * Other entry points:
 *   overlap_prefix  overlap += mask; mask[overlap > 1] = 0 over the masks of one call, in order
 *                                                             (mask_utils.py:77-78); bbox optional (supersets ok)
 *   column_counts   np.sum(masks, axis=(0, 1)) (per-column pixel counts; `counts` pre-zeroed)
 *                                                             (mask_utils.py:62)
 *   pair_intersections  np.count_nonzero(a[pi[p]] & b[pj[p]]) (inference.py:431, 2710;
 *                                                              spatial_constraints.py:143, 186)
 *   place_tiles     cv2.resize(mask, (tile_w, tile_h), INTER_NEAREST) + paste into a zero (H, W)
 *                   frame at (x_off, y_off)                   (inference.py:2399-2420)
 * seg (overlap_prefix, column_counts): NULL, or a non-decreasing segment id per mask so that the masks of
 * many (tile, class) calls share ONE launch; column_counts then fills counts[S, W] (pre-zeroed).   */
enum { DEMIA_MOP_END = 0, DEMIA_MOP_FILL = 1, DEMIA_MOP_DILATE = 2, DEMIA_MOP_ERODE = 3, DEMIA_MOP_DROP_MULTI = 4,
       DEMIA_MOP_FLAG_MULTI = 5, DEMIA_MOP_GATE = 6 };
int demia_mask_program(uint32_t* masks, uint32_t* scratch, const int32_t* bbox, const uint8_t* active, uint32_t program,
                       int64_t M, int H, int W, int32_t* area, int32_t* bbox_out, int32_t* flag, void* stream);
/* The same with a WORKLIST [M + 2] i32 (device, contents ignored on entry): the launch over all masks handles the small
 * regions and lists the masks that need the 64-KiB-LDS variant, which then runs as a fixed grid over that list instead of
 * one workgroup per mask (three of four masks of a real batch are small).  worklist == NULL: demia_mask_program. */
int demia_mask_program_wl(uint32_t* masks, uint32_t* scratch, const int32_t* bbox, const uint8_t* active, uint32_t program,
                          int64_t M, int H, int W, int32_t* area, int32_t* bbox_out, int32_t* flag, int32_t* worklist, void* stream);
int demia_mask_overlap_prefix(uint32_t* masks, const int32_t* seg, const int32_t* bbox, int64_t M, int H, int W, void* stream);
int demia_mask_column_counts(const uint32_t* masks, const int32_t* seg, const int32_t* bbox, int64_t M, int H, int W,
                             int32_t* counts, void* stream);
int demia_mask_pair_intersections(const uint32_t* a, const uint32_t* b, const int32_t* pi, const int32_t* pj,
                                  const int32_t* bbox_a, const int32_t* bbox_b, int32_t* out, int64_t P,
                                  int H, int W, void* stream);
/* pair_matrix: the intersection counts of EVERY pair inside a segment (the masks of one tile or one class pass) in one
 * launch, with the pair list made on the device from the boxes -- what the greedy IoU loops (inference.py:1451-1459) and
 * step 2 of deduplicate_masks_smart (inference.py:2640-2671) ask for, without a host round trip to build the list.
 * first[i] / count[i]: first mask and length of mask i's segment; label (optional): only equal-label pairs are counted;
 * out [M, ld] pre-zeroed, out[i][j - first[i]] = np.count_nonzero(mask_i & mask_j) for j > i (upper triangle). */
int demia_mask_pair_matrix(const uint32_t* masks, const int32_t* bbox, const int32_t* first, const int32_t* count,
                           const int32_t* label, int32_t* out, int64_t M, int ld, int H, int W, void* stream);
/* Host-side decision loops (no GPU work): the reference's sequential greedy filters for a whole batch of tiles in one
 * native call over the integer tables the device reduced.  inter [n, ld] / row_first: the matrix of demia_mask_pair_matrix
 * (host copy).  greedy_keep: inference.py:1451-1459 per segment (keep[p] = 0 / 1).  dedup_smart: step 2 of
 * deduplicate_masks_smart, inference.py:2640-2671, bug for bug; items / scores / classes per tile entry,
 * bbox [n][4] = y0, x0, y1, x1 and area [n] per global mask; keep_out = kept LOCAL positions per tile in keeping order. */
int demia_host_greedy_keep(const int32_t* inter, int ld, const int32_t* row_first, const int64_t* area, const int32_t* seg_first,
                           const int32_t* seg_len, int S, double thr, uint8_t* keep);
int demia_host_dedup_smart(const int32_t* inter, int ld, const int32_t* row_first, const int64_t* area, const int64_t* bbox,
                           const int32_t* items, const double* scores, const int32_t* classes, const int32_t* tile_off, int T,
                           double thr, int32_t* keep_out, int32_t* keep_cnt);
/* The float columns of measurements_results.csv as text (inference.py:1209-1230 writes them with csv.writer, i.e. repr()):
 * vals [rows][cols] f64 -> per row the Python repr() of its floats joined by ',', rows separated by '\n'.  Returns the
 * bytes written, -1 if cap < rows * cols * 26 + rows. */
int64_t demia_host_repr_rows(const double* vals, int64_t rows, int cols, char* out, int64_t cap);
/* a16: the EncodedPixels column of R50_flip_results.csv (inference.py:917-925: `" ".join(map(str, rle_encoding(mask)))`,
 * mask_utils.py:17-35) for M masks in one host call, from the bbox-cropped packed words of demia_mask_crop_pack: 1-based
 * (start, length) pairs over the column-major flattening of an H-row frame, as decimal text separated by single blanks.
 * text_off [M + 1]: mask m's text is out[text_off[m] .. text_off[m + 1]).  Returns the bytes written, or -(bytes needed)
 * when cap is too small.  Host code: nothing here touches the GPU. */
int64_t demia_host_rle_text(const uint32_t* payload, const int32_t* bbox, const int64_t* offsets, int64_t M, int H,
                            char* out, int64_t cap, int64_t* text_off);
int demia_mask_place_tiles(const uint32_t* src, uint32_t* dst, const int32_t* x_off, const int32_t* y_off, int64_t T,
                           int src_h, int src_w, int tile_h, int tile_w, int H, int W, void* stream);
/* Instance tables (what the ranks exchange before the global duplicate / containment filters,
 * inference.py:2452-2460): the payload of mask m is its bbox rows x word columns, row-major, at payload[offsets[m]];
 * offsets [M] i64 = exclusive prefix sums of rows * word columns (0 for empty masks, bbox -1).  crop_unpack writes
 * the regions into `masks`, which the caller has zeroed.                                                           */
int demia_mask_crop_pack(const uint32_t* masks, const int32_t* bbox, const int64_t* offsets, int64_t M, int H, int W,
                         uint32_t* payload, void* stream);
/* dst[i] = src[index[i]] for masks that are ZERO OUTSIDE their bbox (every mask of this path is: paste hint, program
 * output): replaces the plane-to-plane gathers between the stages of the class loop (`masks[sel]` copies of the dense
 * reference, inference.py:1405-1440, 2452-2472).  Only the box is read; the destination plane is written once (zeros
 * outside the box).  index [M] i64 into src [*, H, W/32]; bbox [M, 4] (-1: empty -> a zero plane); dst [M, H, W/32]. */
int demia_mask_gather_regions(const uint32_t* src, const int64_t* index, const int32_t* bbox, int64_t M, int H, int W,
                              uint32_t* dst, void* stream);
/* The same gather into a POOL of planes that stay zero outside per-slot boxes: pool [>= M, H, W/32] zeroed once, prev [>= M, 4]
 * i32 initialised to -1.  Slot m receives src[index[m]] (zero outside bbox[m]); only the union of prev[m] and bbox[m] is
 * written, and prev[m] becomes bbox[m] grown by `grow` pixels -- the reach of the in-place stages that follow (a dilation:
 * 1).  For callers that consume a batch's planes before the pool comes round again (the tile-batch loop). */
int demia_mask_gather_regions_pooled(const uint32_t* src, const int64_t* index, const int32_t* bbox, int64_t M, int H, int W,
                                     uint32_t* pool, int32_t* prev, int grow, void* stream);
int demia_mask_crop_unpack(const uint32_t* payload, const int32_t* bbox, const int64_t* offsets, int64_t M, int H, int W,
                           uint32_t* masks, void* stream);

/* a18: contrast distribution (measurements.py:195-215, switched by `measure_contrast_distribution`, inference.py:58,1198):
 * per mask the 256-bin histogram of gray = cv2.cvtColor(image, COLOR_BGR2GRAY) over the mask's pixels -- what
 * np.histogram(gray[mask > 0], bins=256, range=(0, 255)) counts (integer data: bin i = pixels of value i).  The CDF and the
 * three np.interp calls on 256 numbers stay on the host.
 *   image [H, W, channels] u8 (channels 3 = BGR, 1 = already gray); bbox [M, 4] a superset of each mask's tight box (-1: empty);
 *   hist [M, 256] i32 (written, not accumulated).                                                                        */
int demia_mask_gray_histogram(const uint32_t* masks, const int32_t* bbox, const uint8_t* image, int channels,
                              int64_t M, int H, int W, int32_t* hist, void* stream);

/* a17/a18: contours and morphometrics ---------------------------------------------------------
 * demia_mask_contours = cv2.findContours(mask, RETR_EXTERNAL, CHAIN_APPROX_SIMPLE) per mask
 * (inference.py:1164, 2605) + cv2.contourArea + cv2.arcLength(closed) (inference.py:1175, 2607;
 * measurements.py:134-135).  Components nested in holes are skipped as RETR_EXTERNAL does (the
 * kernel floods the outside background of the bbox region itself).  bbox may be a superset of the tight box;
 * scratch (same shape as masks) is only used by regions larger than the LDS buffers.
 *   count [M]; info [M, C, 4] = start x, start y, n points, offset into `points`;
 *   red [M, C, 2] f64 = area, perimeter; points [max_points, 2] i32 (x, y);
 *   counters [4] i32: [0] points used, [1] error bits (1 candidates, 2 contours > C, 4 points),
 *            [2] / [3] statistics: masks whose parallel border walk fell back to the sequential one / used it.
 * Contours of one mask come out unordered: OpenCV's order is descending (start y, start x).
 * demia_contour_measure = calculate_measurements (measurements.py:114-233) per contour:
 *   out [M, out_c, 12] f64 = major_axis_length, minor_axis_length, eccentricity, Length, Width,
 *   CircularED, Aspect_Ratio, Circularity, Chords, Feret_diam, Roundness, Sphericity; out_c <= C is the number
 *   of contour slots per mask the caller wants back (contours c >= out_c are skipped): with one contour per mask
 *   the table that crosses PCIe is M x 12 doubles instead of M x C x 12.                         */
int64_t demia_contour_work_ints(int M, int C, int max_points);
int64_t demia_contour_work_floats(int M, int C, int max_points);
int64_t demia_contour_work_doubles(int M, int C, int max_points);
int demia_mask_contours(const uint32_t* masks, uint32_t* scratch, const int32_t* bbox, int M, int H, int W, int C,
                        int max_points, int32_t* count, int32_t* info, double* red, int32_t* points,
                        int32_t* counters, void* stream);
/* The same with a WORKLIST [M + 2] i32 (device, contents ignored on entry), as demia_mask_program_wl: the large-region variant
 * runs over the masks the small one listed instead of over all M.  worklist == NULL: demia_mask_contours. */
int demia_mask_contours_wl(const uint32_t* masks, uint32_t* scratch, const int32_t* bbox, int M, int H, int W, int C,
                           int max_points, int32_t* count, int32_t* info, double* red, int32_t* points,
                           int32_t* counters, int32_t* worklist, void* stream);
int demia_contour_measure(const int32_t* select /* [M] or NULL */, const int32_t* count, const int32_t* info,
                          const double* red, const int32_t* points, int M,
                          int C, int max_points, int32_t* work_i, float* work_f, double* work_d, double um_pix,
                          double* out, int out_c, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DEEPEMIA_HIP_H */
