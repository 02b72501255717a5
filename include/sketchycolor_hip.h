/*
 * sketchycolor_hip.h -- C ABI of libsketchycolor_hip.so (gfx950 / MI355X).
 *
 * The reference (SketchyScene/SketchySceneColorization) has no FFI: every op of
 * its hot path is a stock TensorFlow-1 op invoked from Python.  Each entry point
 * below therefore names the TF op / reference call site it replaces
 * (paths relative to Foreground_Instance_Colorization/obj_lib/).
 *
 * Conventions: every pointer is a DEVICE pointer unless it is a descriptor
 * struct (host memory, read during the call); `stream` is a hipStream_t passed
 * as void*; return value 0 = ok, non-zero = hipError_t or a negative argument
 * error.  No ownership is transferred, no global state is kept, nothing is
 * allocated: callers hand in workspaces.  Activations are NHWC fp32 internally
 * (channel counts padded to a multiple of 4 where noted); filters keep the TF
 * layouts ([kh,kw,Cin,Cout] for conv, [kh,kw,Cout,Cin] for conv-transpose).
 */
#ifndef SKETCHYCOLOR_HIP_H
#define SKETCHYCOLOR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* activation codes applied on load (after the per-channel affine) */
#define SSC_ACT_NONE 0
#define SSC_ACT_RELU 1   /* tf.nn.relu                 models_collection.py:518,531 */
#define SSC_ACT_LRELU 2  /* tf.maximum(0.2*x, x)       models_collection.py:51-53   */
#define SSC_ACT_TANH 3   /* only for ssc_affine_act / ssc_residual_merge outputs, never on load */
#define SSC_ACT_PRELU 5  /* tf.maximum(leak*x, x) with a trainable scalar leak (models_collection.py:56-60); ssc_concat_parts only,
                            the part's `ab` pointer then addresses the leak */
#define SSC_ACT_MIU 4    /* miu_relu (x + sqrt(0.09 + x^2))/2, models_collection.py:63-65; pointwise kernels only */

/*
 * A "gather view": one or two NHWC tensors seen as a single [N,H,W,C0+C1]
 * tensor (tf.concat on channels, models_collection.py:512-516), with an
 * optional per-channel affine a*x+b (the folded batch-statistics norm,
 * models_collection.py:36-46) followed by an activation, evaluated while the
 * tile is loaded.  Out-of-image taps read 0 (tf.pad CONSTANT, :388).
 */
typedef struct ssc_gview {
    const float* s0;   /* [N,H,W,C0] */
    const float* s1;   /* [N,H,W,C1] or NULL */
    const float* ab0;  /* [2][C0]: a then b for s0; NULL = identity */
    const float* ab1;  /* [2][C1]: a then b for s1; NULL = identity */
    int32_t C0, C1;    /* both multiples of 4 */
    int32_t H, W;
    int32_t act;       /* activation of s0 (and of s1 when act1 < 0) */
    int32_t act1;      /* activation of s1, or -1 = same as act (residual generators concatenate an already
                          activated block output with a raw+norm+lrelu encoder tensor, models_collection.py:660-664) */
} ssc_gview;

/*
 * Implicit-GEMM convolution, forward form:
 *   out[pix][n] = sum_{tap,k} X[pix@tap][k] * F(tap,k,n)
 * Replaces tf.nn.conv2d (models_collection.py:389, residual_util.py:24,33),
 * tf.nn.conv2d_transpose (models_collection.py:402; nphase=4 sub-pixel form)
 * and the data-gradient of both, plus tf.matmul (1x1 form; BasicLSTMCell and
 * fully_connected, models_collection.py:184-236, mru.py:52-92).
 */
typedef struct ssc_conv_desc {
    ssc_gview x;
    const float* w;       /* filter [KH][KW][wC0][wC1] */
    const float* bias;    /* [Nn] or NULL */
    float* out;           /* [NB,OH,OW,ldc] */
    int32_t NB, PH, PW;   /* output lattice: M = NB*PH*PW */
    int32_t TH, TW;       /* taps per lattice point */
    int32_t in_stride;    /* iy = py*in_stride + ioff_y + ty */
    int32_t ioff_y, ioff_x;
    int32_t nphase;       /* 1, or 4 = stride-2 transposed conv (k=4, pad 1) */
    int32_t ky0, kx0, kstep;  /* filter row = ky0 + ty*kstep */
    int32_t KH, KW, wC0, wC1;
    int32_t bmode;        /* 0: k->wC0, n->wC1 ("KN"); 1: n->wC0, k->wC1 ("NK") */
    int32_t k_real;       /* real K channels in the filter (<= x.C0+x.C1) */
    int32_t n_off, Nn;    /* first filter n index, number of real outputs */
    int32_t Nstore;       /* columns stored (>= Nn; extra ones are written 0) */
    int32_t OH, OW, ldc;
    int32_t out_stride, ooff_y, ooff_x; /* oy = py*out_stride + ooff_y */
    int32_t epi;          /* 0 none, 1 tanh (models_collection.py:533), 2 lrelu 0.2 (MRU gates, mru.py:407-413) */
    int32_t accumulate;   /* 1: out += result */
    float* stat_partial;  /* set by ssc_conv_forward_bn (callers leave it NULL): per row tile [2][Nstore] sums / sums of squares
                             of the output columns, written by the epilogue */
    uint32_t* sk_flags;   /* stream-K hand-off flags (device): SSC_SK_FLAG_WORDS words, zero before the first launch that
                             uses them and private to the launch stream (the kernel leaves them zero); NULL = the launch
                             may not split tiles across workgroups inside the kernel (split-K slabs + reduce kernel instead) */
    /* set by ssc_conv_forward_bnbwd (callers leave sb_x NULL): the output of this launch is the gradient g w.r.t. act(a*x+b)
       of a batch-statistics-normed tensor x laid out like the output; the rows of stat_partial then hold
       [sum dz | sum dz*xhat], dz = g * act'(a*x+b), xhat = (x - mean) * rstd -- the two sums of the norm's backward */
    const float* sb_x;
    const float* sb_ab;   /* [2][Nstore]: a, b */
    const float* sb_stats;/* [2][Nstore]: mean, 1/std */
    int32_t sb_ldx;       /* row stride of x */
    int32_t sb_act;       /* SSC_ACT_* of the consumer */
    int32_t sk_tag;       /* caller's id of this launch (31 bits): a hand-off that times out reports 0x80000000 | sk_tag */
    /* set by ssc_conv_forward_bnbwd2 (callers leave sb2_x NULL): the output columns [sb2_col0, Nstore) are the gradient w.r.t. a
       SECOND normed tensor (the data gradient of a layer that reads concat[x, x2] -- relu(concat[decoder_{k+1}, encoder_k]),
       models_collection.py:512-531 -- in one launch); the sb_* set then describes columns [0, sb2_col0) only (tables of
       sb2_col0 entries), this set the remaining Nstore - sb2_col0, its rows of sums going to stat_partial2 */
    int32_t sb2_col0;
    const float* sb2_x;
    const float* sb2_ab;
    const float* sb2_stats;
    float* stat_partial2;
    int32_t sb2_ldx;
    int32_t sb2_act;
    int32_t stat_mode;    /* set by ssc_conv_forward_minmax (callers leave 0): 1 = the rows of stat_partial hold the per-column
                             MINIMUM and MAXIMUM of the tile's (activated) outputs instead of sum and sum of squares */
    int32_t ws_kc;
    /* Filter pre-split into three bf16 planes (ssc_filter_split; orientation = bmode), or NULL.  When set and the launch
       qualifies (every 32-wide K-tile inside one tap and one source, n_off a multiple of 32, more than 32 stored columns) the
       contraction runs on the bf16 matrix pipe as six bf16 products per fp32 product with fp32 accumulation (igemm_bf16.hip:
       fp32-grade results at 6/16 of the fp32 MFMA's cycles); `w` must still point at the fp32 filter.  ws_kc / ws_nbp: the
       planes' chunk and block counts as ssc_filter_split_geom reports them. */
    const void* wsplit;
    int32_t ws_nbp;
    int32_t lds_hint;     /* 0: the form that is fastest alone.  1: the launch shares the chip with launches of other streams (a train
                             step's side-by-side chains): prefer the form with the smaller LDS footprint -- the bf16-split kernel on
                             ONE operand stage (42 KB instead of 75 KB: a filter-gradient workgroup fits beside two conv workgroups),
                             3-5 % slower alone, 2.5 % faster per Pix2Pix train iteration */
} ssc_conv_desc;
#define SSC_SK_FLAG_WORDS 8192   /* >= resident workgroups of the largest grid; the last word reports a hand-off timeout:
                                    0 = none, else 0x80000000 | sk_tag of the first launch whose owner workgroup gave up
                                    waiting for a K slice.  The output of that launch is WRONG (a partial sum): callers must
                                    read the word wherever they read results back (losses, snapshots) and fail; the flags
                                    are to be zeroed before the array is used again. */
/*
 * 3-way bf16 split of a filter W[taps][c0][c1] (fp32, c1 contiguous) for the bf16 form of ssc_conv_forward: x = h + m + l
 * exactly, each part a bf16, stored as the B operands of v_mfma_f32_32x32x16_bf16 (fragment-major, zero padded).
 * orient 0: GEMM k = c0, n = c1 (ssc_conv_desc.bmode 0);  orient 1: k = c1, n = c0 (bmode 1).
 * The planes must be refreshed whenever the fp32 filter changes (once per optimizer step).
 */
typedef struct ssc_split_job {
    const float* w;
    void* dst;              /* ssc_filter_split_bytes bytes, 16-byte aligned */
    int32_t taps, c0, c1, orient;
    int64_t first_thread;   /* batch form: running sum of the preceding jobs' `threads` (ssc_filter_split_geom) */
} ssc_split_job;
/* kc, nbp: the planes' chunk / block counts (ssc_conv_desc.ws_kc, ws_nbp); bytes: size of the planes buffer; threads: the
   job's share of a batch launch (whole blocks of 256) */
int ssc_filter_split_geom(int taps, int c0, int c1, int orient, int* kc, int* nbp, int64_t* bytes, int64_t* threads);
int ssc_filter_split(const float* w, int taps, int c0, int c1, int orient, void* dst, void* stream);
/* allocates the bf16 form's per-device constants (a hipMalloc + a copy: call once OUTSIDE any stream capture, before the first
   launch that carries `wsplit`; the launches call it too) */
int ssc_bf16_prepare(void);
/* jobs_dev: DEVICE array; total_threads = sum of the jobs' `threads` */
int ssc_filter_split_batch(const ssc_split_job* jobs_dev, int njobs, int64_t total_threads, void* stream);

/* hand-off wait bound in milliseconds (default 20000: only a deadlock guard -- the owner waits for workgroups that were
 * dispatched before it) and a test hook that makes the producers withhold their flags so that the bound is reached */
int ssc_sk_configure(int timeout_ms, int test_withhold);

/*
 * Implicit-GEMM filter gradient:
 *   dF[(tap,cg)][cd] = sum_pix G[pix@tap][cg] * D[pix][cd]
 * G is the gathered (higher-resolution) side, D the 1x1 side at the lattice.
 * Replaces the filter-gradient of tf.nn.conv2d / conv2d_transpose / matmul
 * produced by optim.compute_gradients (graph_single.py:24-30, 309-312).
 */
typedef struct ssc_wgrad_desc {
    ssc_gview g;
    ssc_gview d;
    float* out;           /* [(tap*Cg_real + cg)][ldc] */
    int32_t NB, PH, PW;
    int32_t TH, TW, in_stride, ioff_y, ioff_x;
    int32_t Cg_real;      /* real gathered channels (<= g.C0+g.C1) */
    int32_t Nn;           /* real dense channels (<= d.C0+d.C1) */
    int32_t ldc;
    int32_t accumulate;
    int32_t exact;        /* 1: the exact-fp32 MFMA kernels only; 0: the large layers may run as a 3-way bf16 split of both operands on
                             the bf16 matrix pipe (wgrad128_bf16.hip: six products per fp32 product, fp32 accumulate) */
    int32_t _pad;
} ssc_wgrad_desc;

/* library / device info */
int ssc_version(void);
/* the 16-hex-digit sha256 prefix of the sources (csrc + this header) the binary was built from: callers compare it with the hash
 * of the tree they run from (sketchyscenecolorization_amd/build.py tree_hash) and refuse a stale binary */
int ssc_build_hash(char* buf, int len);
int ssc_device_info(int* cu_count, int* wave_size, char* arch, int arch_len);

/* implicit GEMM (igemm.hip).  ws: split-K slab workspace (may be NULL). */
int ssc_conv_forward(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, void* stream);
int ssc_conv_wgrad(const ssc_wgrad_desc* d, float* ws, int64_t ws_bytes, void* stream);
/* filter gradients of the large layers on a 128 x 128 accumulator tile (wgrad128.hip); ssc_conv_wgrad dispatches to it when
 * _supported: the gathered channels a multiple of 128 inside one source (or exactly 64: two taps per tile), no channel
 * padding on the gathered side, >= 128 real dense channels, every tensor below 2 GiB */
int ssc_conv_wgrad128_supported(const ssc_wgrad_desc* d);
int ssc_conv_wgrad128(const ssc_wgrad_desc* d, float* ws, int64_t ws_bytes, void* stream);
/* filter gradients with 16 dense channels of the 3x3 conv over 16 and the 4x4 stride-1 conv over 64 gathered channels
 * (residual_util.py:92-96, 147-151) on the 16-column MFMA (wgn16.hip); ssc_conv_wgrad dispatches to it when _supported */
int ssc_conv_wgn16_supported(const ssc_wgrad_desc* d);
/* conv + the batch-statistics norm of its output [M*nphase rows, Nstore == ldc columns] folded to ab = [a; b] (y = a*x + b)
 * and stats = [mean; rstd] (models_collection.py:36-46 after :389 / :402): the column sums come out of the conv epilogue
 * when the launch allows it, else ssc_bn_stats reads the output back */
int ssc_conv_forward_bn(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, const float* scale, const float* offset,
                        float eps, float* ab, float* stats, void* stream);
/* the one-output patch head over a 512-channel tensor (discriminate_pix2pix layer_5, models_collection.py:833-835: 4x4,
 * stride 1, 512 -> 1) as streaming kernels (head1.hip): forward (ws: 64 B per input pixel), its data gradient (the
 * hip.conv_dgrad descriptor: one-channel dy, "NK" filter, flipped taps) and its filter gradient (ws: up to 256 x 32 KB slabs).
 * ssc_conv_forward / ssc_conv_wgrad dispatch to them when _supported; SSC_HEAD1=0 keeps the general kernels. */
int ssc_head1_forward_supported(const ssc_conv_desc* d);
int ssc_head1_forward(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, void* stream);
int ssc_head1_dgrad_supported(const ssc_conv_desc* d);
int ssc_head1_dgrad(const ssc_conv_desc* d, void* stream);
int ssc_head1_wgrad_supported(const ssc_wgrad_desc* d);
int ssc_head1_wgrad(const ssc_wgrad_desc* d, float* ws, int64_t ws_bytes, void* stream);
/* the head's data gradient fused with the backward of the batch norm + activation of its input tensor x [pixels][512]
 * (layer_4: models_collection.py:823-830): the gradient w.r.t. act(norm(x)) is recomputed in both passes of the norm backward
 * and never stored.  d = the hip.conv_dgrad descriptor (out ignored), ab = [a; b], stats = [mean; rstd], rowb [NB][512] or
 * NULL = a per-image term (the class head's gradient through its spatial mean) added with rowb_scale; dx [pixels][512];
 * dscale / doffset [512] or NULL; ws >= (2 * 512 + 2) * 512 floats.  -1: not this shape (callers then use ssc_conv_forward +
 * ssc_bn_act_backward_pre) */
int ssc_head1_dgrad_bn_backward(const ssc_conv_desc* d, const float* x, const float* ab, const float* stats, int act,
                                const float* rowb, float rowb_scale, float* dx, float* dscale, float* doffset, float* ws,
                                int64_t ws_bytes, void* stream);
/* rows of partial sums [nblk][2][C] of a norm backward -> coef [2][C] = [mean dz; mean dz*xhat], dscale / doffset (may be NULL) */
int ssc_bn_bwd_finalize(const float* partial, int nblk, int C, int64_t M, float* coef, float* dscale, float* doffset,
                        void* stream);
/* direct (vector-ALU, LDS patch) form for <= 4 output channels; ssc_conv_forward dispatches to it (narrow.hip) */
int ssc_conv_narrow_supported(const ssc_conv_desc* d);
int ssc_conv_narrow_forward(const ssc_conv_desc* d, void* stream);
/* 1 when ssc_conv_forward runs the launch on the few-input-channel kernel (fewchan.hip): 4x4 stride-2 pad-1 conv over a
 * single 4- or 8-channel raw source to 33..64 outputs, output lattice a multiple of 4 x 32 -- generator encoder_1
 * (models_collection.py:454-458), discriminator layer_1 (:798-801), the data gradient of decoder_1 (:529-534) */
/* the 1x1 expansion conv of the bottleneck blocks (K = 16 .. 128 input channels, N = 4 K outputs): streaming kernel (pw1x1.hip) */
int ssc_conv_pw1x1_supported(const ssc_conv_desc* d);
/* the 3x3 stride-1 conv of the bottleneck blocks at 16 / 32 channels and its data gradient (residual_util.py:92-96): streaming
   kernel (c3x3.hip); the batch statistics (ssc_conv_forward_bn) / norm-backward sums (ssc_conv_forward_bnbwd) ride as per-lane sums */
int ssc_conv_c3x3_supported(const ssc_conv_desc* d);
/* the 4x4 stride-2 conv 64 -> <= 16 channels (block_1 of the first encoder bottleneck, residual_util.py:87-91) on the 16-column
   MFMA with K split over the four wavefronts (s2n16.hip) */
int ssc_conv_s2n16_supported(const ssc_conv_desc* d);
/* the k = 4 stride-2 transposed conv 256 -> <= 16 channels (block_1 of the last decoder bottleneck) on the 16-column MFMA, one
   sub-pixel phase per workgroup (tr4n16.hip) */
int ssc_conv_tr4n16_supported(const ssc_conv_desc* d);
/* the Residual / Background generators' first conv, 7x7 stride 2 over the padded image channels -> 64 (fewchan7.hip) */
int ssc_conv_fewchan7_supported(const ssc_conv_desc* d);
/* the k = 4 stride-2 transposed convs of the Background generator's region branch (<= 4 channels in and out,
   bg_colorization_main.py:392-397): a thread per lattice pixel (tr4tiny.hip) */
int ssc_conv_tr4_tiny_supported(const ssc_conv_desc* d);
int ssc_conv_fewchan_supported(const ssc_conv_desc* d);
/* name of the tile configuration the launcher picks for a descriptor (host only; for profiling) */
int ssc_conv_forward_kernel_name(const ssc_conv_desc* d, char* buf, int len);
int ssc_conv_wgrad_kernel_name(const ssc_wgrad_desc* d, char* buf, int len);
/* the launch plan for a descriptor (host only; tuning and tests): out5 = {tile configuration, split-K slabs, whole
 * tiles, K slices per remaining tile (combined inside the launch), modelled kilo-cycles} */
int ssc_conv_forward_plan(const ssc_conv_desc* d, int64_t ws_bytes, int* out5);

/* --- layout (elementwise.hip) --- */
/* dst[n,hw,coff+c] = src[n,c,hw]; tf.transpose NCHW->NHWC (models_collection.py:381) */
int ssc_nchw_to_nhwc(const float* src, float* dst, int N, int C, int HW, int ldc, int coff, void* stream);
/* dst[n,c,hw] = src[n,hw,coff+c] */
int ssc_nhwc_to_nchw(const float* src, float* dst, int N, int C, int HW, int ldc, int coff, void* stream);
/* Device-side image pre / post-processing of the inference, test and validation procedures
 * (main_procedure.py:361-621).  src uint8 [N,H,W,3] -> dst float [N,H,W,4] = src/255*2-1 (channel 3 = 0), optionally
 * after thicken_drawings (input_pipeline.py:242-257: 2x2 grey dilation of the dark strokes of channel 0). */
int ssc_sketch_preprocess_u8(const uint8_t* src, int N, int H, int W, int thicken, float* dst, void* stream);
/* src float NHWC rows of ldc floats, image in channels [coff, coff+3) -> dst uint8 [M,3] = ((x+1)/2*255) truncated */
int ssc_image_postprocess_u8(const float* src, int ldc, int coff, int64_t M, uint8_t* dst, void* stream);
/* PIL.Image.resize of an 8-bit image on the device (resize_and_padding_mask_image, input_pipeline.py:199-239: ANTIALIAS =
 * LANCZOS; reverse_resize_image, Pipeline_utils/fg_color_utils.py:137-160: scipy.misc.imresize = PIL bilinear): Pillow's
 * two-pass 8-bit resampler, horizontal then vertical, bit for bit.  src uint8 [H,W,C]; chan >= 0: only that channel,
 * replicated over the OC channels of dst; chan < 0: all C channels (OC == C).  bnd_* [new][2] = {first tap, taps}, k_* [new][ks]
 * 22-bit fixed-point coefficients (host: obj_lib/input_pipeline.py::resample_coeffs); NULL tables = that axis keeps its size.
 * The result fills [top, top+new_h) x [left, left+new_w) of dst uint8 [OH,OW,OC], the rest of dst is `fill`.
 * tmp: H * new_w * (chan >= 0 ? 1 : C) bytes. */
int ssc_resample_u8(const uint8_t* src, int H, int W, int C, int chan, const int32_t* bnd_h, const int32_t* k_h, int ks_h,
                    int new_w, const int32_t* bnd_v, const int32_t* k_v, int ks_v, int new_h, uint8_t* tmp, uint8_t* dst,
                    int OH, int OW, int OC, int top, int left, int fill, void* stream);
/* Training-queue decode (get_paired_input, input_pipeline.py:77-131) of N raw records: img / sk uint8 [N,R,R,3] ->
 * img_out / sk_out float NCHW [N,3,size,size], R = f * size.  Image: pixel (f*y, f*x) (TF1 bilinear at an integer
 * factor), (v - min)/(max - min + 1) over the resized image, + noise [N,size,size,3] (uniform [0,1/256), may be NULL),
 * * 2 - 1.  Sketch: mean of the f x f block (TF1 area), / 255 * 2 - 1; read from sk_f32 [N,R,R,3] instead of sk when
 * that is not NULL (distance maps).  mnmx: [N,2] scratch (per-image min, max). */
int ssc_decode_paired_u8(const uint8_t* img, const uint8_t* sk, const float* sk_f32, int N, int R, int size,
                         const float* noise, float* img_out, float* sk_out, float* mnmx, void* stream);
/* --distance_map 1 (input_pipeline.py:86-96): sk uint8 [N,R,R,3] -> out float [N,R,R,3] = exact Euclidean distance of
 * every voxel to the nearest stroke voxel (sk < 250; scipy.ndimage.distance_transform_edt over [R,R,3]) / max * 255.
 * Pass it to ssc_decode_paired_u8 as sk_f32 (then sk may be NULL).  ws: (2*N*R*R*3 + N) int32 of scratch. */
int ssc_distance_map_u8(const uint8_t* sk, int N, int R, float* out, int32_t* ws, int64_t ws_bytes, void* stream);
/* host-side CRC-32C (Castagnoli) of n bytes: TFRecord record framing (tf.TFRecordReader, input_pipeline.py:57-59) */
uint32_t ssc_crc32c(const uint8_t* data, int64_t n);
int ssc_fill(float* dst, float value, int64_t n, void* stream);
/* profiling aid: *dst = the device's 100 MHz wall clock at the time the launch runs (a one-lane kernel; capturable) */
int ssc_timestamp(uint64_t* dst, void* stream);
/* ---- MRU blocks (mru.py:353-461 mru_conv_block_v3, :527-591 mru_deconv_block_v2), NHWC fp32 -------------------- */
/* mean_pool (mru.py:15-19): out[n, y, x, c] = mean of the 2x2 block */
int ssc_mean_pool2(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, void* stream);
/* conditional batch norm (models_collection.py:22-35): stats = [mean(C); rstd(C)] (ssc_bn_stats), scale_m/offset_m
 * [n_labels, C]; abn[n] = [a(C); b(C)] with a = scale[label_n]*rstd, b = offset[label_n] - mean*a */
int ssc_cbn_fold(const float* stats, const float* scale_m, const float* offset_m, const int32_t* labels, int N, int C,
                 float* abn, void* stream);
/* the fold of per-split rows [N][nsplit][2][C] (min row, max row) to mnmx [N][2][C] */
int ssc_minmax_finalize(const float* part, int nsplit, int N, int C, float* mnmx, void* stream);
/* ssc_conv_forward + ssc_minmax_hw of its output (the MRU gates: conv + bias + lrelu, then reduce_min / reduce_max over the
 * positions of every sample and channel, mru.py:407-415): when the launch qualifies (uniform-tap kernel, whole tiles finished
 * in one workgroup, a sample's positions a multiple of the tile's rows) the per-tile minima / maxima come out of the conv
 * epilogue and only the fold remains; otherwise the output is read back once. */
int ssc_conv_forward_minmax(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, float* mnmx, void* stream);
/* mnmx[n] = [min(C); max(C)] over the P = H*W rows of sample n (tf.reduce_min/max(axis=[2,3]), mru.py:414-415) */
int ssc_minmax_hw(const float* x, int ld, int N, int P, int C, float* mnmx, float* workspace, int64_t workspace_bytes,
                  void* stream);
/* channel-concat writer: out[row, :] = [part0 | part1 | part2], each part = act(a*x+b) (per-sample a,b when
 * ab_sample_stride != 0), optionally read through a nearest 2x upsample (mru.py:22-28) and multiplied by a min-max
 * normalised gate (rg * ht, mru.py:571) */
typedef struct {
    const float* x;
    const float* ab;      /* NULL or [a(C); b(C)] (+ n*ab_sample_stride) */
    const float* gate;    /* NULL or raw gate [N,H,W,C] */
    const float* mnmx;    /* [N,2,C] when gate != NULL */
    int32_t ld, C, ab_sample_stride, act, upsample, _pad;
} ssc_cat_part;
typedef struct {
    ssc_cat_part p[3];
    float* out;
    int32_t nparts, ldo, N, H, W, _pad;
} ssc_cat_desc;
int ssc_concat_parts(const ssc_cat_desc* d, void* stream);
/* out = ht + (rg - min)/(max - min) * img      (mru.py:422) */
int ssc_mru_gate_merge(const float* ht, const float* rg, const float* mnmx, const float* img, float* out, int N,
                       int64_t P, int C, void* stream);
/* out = hp*(1 - z) + miu(a2*h2+b2)*z, z = (zg - min)/(max - min), hp = ht_ab ? miu(a*ht+b) : ht, ht read at
 * (y/2, x/2) when ht_lowres      (mru.py:583-589) */
/* ---- their backward passes ---- */
/* out[c] (+)= sum_r x[r, c]: bias gradients */
int ssc_colsum(const float* x, int ld, int64_t M, int C, float* out, int accumulate, float* workspace,
               int64_t workspace_bytes, void* stream);
/* dst[r, 0:C] (+)= src[r, 0:C] with independent row strides (splitting a concat gradient) */
int ssc_strided_copy(const float* src, int lds, float* dst, int ldd, int64_t M, int C, int accumulate, void* stream);
/* out (+)= scale * (sum of each 2x2 block): gradient of the nearest upsample (scale 1) / forward mean_pool (0.25) */
int ssc_pool2(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int C, float scale, int accumulate,
              void* stream);
/* backward of y = act(cond_batchnorm(x)) (act = SSC_ACT_MIU or NONE): dx (+)= ..., table gradients
 * dscale_m/doffset_m [n_labels, C] (may be NULL).  gy rows have stride ldg (a channel slice of a concat gradient). */
int ssc_cbn_act_backward(const float* x, const float* abn, const float* stats, const float* scale_m,
                         const int32_t* labels, int n_labels, const float* gy, int ldg, int act, int N, int P, int C,
                         float* dx, int lddx, int accumulate_dx, float* dscale_m, float* doffset_m,
                         int accumulate_params, float* workspace, int64_t workspace_bytes, void* stream);
/* backward of prelu: dx (+)= gy * (leak*x >= x ? leak : 1) (dx may be NULL), dleak (+)= sum gy*x over that set */
int ssc_prelu_backward(const float* x, int ldx, const float* leak, const float* gy, int ldg, int64_t M, int C, float* dx,
                       int lddx, int accumulate_dx, float* dleak, int accumulate_leak, float* workspace,
                       int64_t workspace_bytes, void* stream);
/* backward of r = (g - min)/(max - min) with g = lrelu(pre) stored: dpre from gr = d loss / d r (TF tie rule) */
int ssc_minmax_gate_backward(const float* g, const float* mnmx, const float* gr, int N, int P, int C, float* dpre,
                             float* workspace, int64_t workspace_bytes, void* stream);
/* ht_plus = ht + r*img: gr = g*img, gimg = g*r */
int ssc_mru_gate_merge_backward(const float* ghtp, const float* rg, const float* mnmx, const float* img, float* gr,
                                float* gimg, int N, int64_t P, int C, void* stream);
/* out = hp*(1-z) + h*z: ghp = g*(1-z), gh = g*z, gz = g*(h - hp) (arguments as ssc_mru_blend) */
int ssc_mru_blend_backward(const float* gout, const float* ht, const float* ht_ab, int ht_lowres, const float* h2,
                           const float* h2_ab, const float* zg, const float* mnmx, float* ghp, float* gh, float* gz,
                           int N, int H, int W, int C, void* stream);
/* G[:, 0:C] = d loss / d (r * up(ht)): gr = G*up(ht); G[:, 0:C] <- G*r in place */
int ssc_mru_in2_gate_backward(float* G, int ldG, const float* rg, const float* mnmx, const float* ht_low, float* gr,
                              int N, int H, int W, int C, void* stream);
int ssc_mru_blend(const float* ht, const float* ht_ab, int ht_lowres, const float* h2, const float* h2_ab,
                  const float* zg, const float* mnmx, float* out, int N, int H, int W, int C, void* stream);

/* out[r, c] = act(ab[c]*x[r, c] + ab[ldab + c])  (ab may be NULL): materialises a normalised tensor (final
 * tanh(batchnorm(.)), models_collection.py:664-667) or re-strides a channel-padded one (ldx != ldo) */
int ssc_affine_act(const float* x, int ldx, const float* ab, int ldab, int act, float* out, int ldo, int64_t M, int C,
                   void* stream);
/* bottleneck_residual_{en,de,pu} 'block_add' (residual_util.py:103-109, 138-146, 165-167):
 * out = act(a1*x1+b1 + (ab2 ? a2*x2+b2 : x2)) */
int ssc_residual_merge(const float* x1, const float* ab1, const float* x2, const float* ab2, int act, float* out,
                       int64_t M, int C, void* stream);

/* --- batch-statistics norm (tf.nn.moments + tf.nn.batch_normalization,
 *     models_collection.py:36-46) --- */
/* per-channel mean/biased var over M rows of x[M,ldx] (C channels) folded with
 * scale/offset into ab=[a;b] (y = a*x+b); stats = [mean; rstd].  ws >= nblk*2*C floats. */
int ssc_bn_stats(const float* x, int64_t M, int C, int ldx, const float* scale, const float* offset,
                 float eps, float* ab, float* stats, float* ws, int64_t ws_bytes, void* stream);
/* the second half of ssc_bn_stats: partial [nblk][2][C] column sums / sums of squares over M rows -> ab, stats */
int ssc_bn_finalize(const float* partial, int nblk, int C, int64_t M, const float* scale, const float* offset, float eps,
                    float* ab, float* stats, void* stream);
/*
 * Backward of y = act(a*x+b) for up to two consumers (g1 through act1, g2
 * through act2):  dz = g1*act1'(z) + g2*act2'(z).
 * has_bn=1: dx = a*(dz - mean(dz) - xhat*mean(dz*xhat)), dscale = sum(dz*xhat),
 * doffset = sum(dz).  has_bn=0: dx = dz (ab/stats ignored, z = x).
 */
int ssc_bn_act_backward(const float* x, int64_t M, int C, int ldx, const float* ab, const float* stats,
                        const float* scale, const float* g1, int ldg1, int act1, const float* g2, int ldg2,
                        int act2, int has_bn, float* dx, int lddx, float* dscale, float* doffset, float* ws,
                        int64_t ws_bytes, void* stream);
/*
 * The same with the two per-channel sums taken elsewhere: `pre` = nrows rows [2][C] of partial sums
 * [sum dz | sum dz*xhat] written by the epilogues of the launches that produced g1 (and g2) -- see
 * ssc_conv_forward_bnbwd; pre = NULL: as ssc_bn_act_backward.  Saves the pass over x, g1, g2 of
 * the gradient chain generate_pix2pix -> batchnorm -> lrelu/relu (models_collection.py:36-46, 434-439).
 * rowb != NULL (only with pre == NULL): g1[r][c] += rowb[r / rowb_P][c] * rowb_scale on the fly -- the gradient of the
 * discriminator's class head through its spatial mean (models_collection.py:836-840), one row per image.
 */
int ssc_bn_act_backward_pre(const float* x, int64_t M, int C, int ldx, const float* ab, const float* stats,
                            const float* g1, int ldg1, int act1, const float* g2, int ldg2, int act2, int has_bn,
                            float* dx, int lddx, float* dscale, float* doffset, const float* pre, int nrows,
                            const float* rowb, float rowb_scale, int rowb_P, float* ws, int64_t ws_bytes, void* stream);
/*
 * The norm backward in two steps (sums, streaming apply pass).  `job` describes the site exactly
 * as the arguments of ssc_bn_act_backward_pre do (host memory, read during the call).
 *   ssc_bn_bwd_sums     partial sums (from `pre` / nrows when given, else a pass over x, g1, g2) -> coef [2][C] = mean dz,
 *                       mean dz*xhat, and the scale / offset gradients.  coef is the caller's buffer: it must stay untouched
 *                       until the apply pass has run (the workspace is not a place for it).
 *   ssc_bn_bwd_apply    dx = a*(dz - coef0 - xhat*coef1) (has_bn) or dx = dz, as a launch of its own.
 * Same bits as ssc_bn_act_backward_pre.  (Round 4's ssc_conv_wgrad_hosting, the apply pass inside the filter-gradient launch of
 * the layer above, measured slower in the step and was removed in round 5.)
 */
typedef struct ssc_bn_apply_job {
    const float* x;       /* the normed tensor (raw), [M][ldx] */
    int64_t M;
    int32_t C, ldx;
    const float* ab;      /* [2][C] folded norm (has_bn) */
    const float* stats;   /* [2][C] mean, 1/std (has_bn) */
    const float* g1;      /* gradient w.r.t. act1(z) */
    const float* g2;      /* second consumer's gradient (through act2) or NULL */
    int32_t ldg1, act1, ldg2, act2;
    int32_t has_bn;       /* 0: dx = dz (activation only) */
    int32_t rowb_P;       /* with rowb: g1[r] += rowb[r / rowb_P] * rowb_scale */
    const float* rowb;
    float rowb_scale;
    int32_t lddx;
    const float* coef;    /* [2][C], written by ssc_bn_bwd_sums (has_bn) */
    float* dx;            /* [M][lddx] */
} ssc_bn_apply_job;
int ssc_bn_bwd_sums(const ssc_bn_apply_job* job, const float* pre, int nrows, float* coef, float* dscale, float* doffset,
                    float* ws, int64_t ws_bytes, void* stream);
int ssc_bn_bwd_apply(const ssc_bn_apply_job* job, void* stream);
/*
 * Backward through the output of a bottleneck block, out = act(norm_A(xa) + shortcut) (residual_util.py:103-109, 138-146,
 * 165-167): dz = g * act'(out) and, with that one dz, the backward of block_3's norm (site A: dxa, scale / offset gradients)
 * and -- en / de blocks, xb != NULL -- of the projection shortcut's norm (site B).  Three launches for what is an activation
 * pass + two ssc_bn_act_backward otherwise.  dz_out != NULL: dz is also written (the identity shortcut of a pu block needs
 * it).  All tensors dense [M][C]; coef: [3][C] scratch of the caller.
 */
int ssc_block_out_backward(const float* out, const float* g, int64_t M, int C, int act, const float* xa, const float* aba,
                           const float* sta, const float* xb, const float* abb, const float* stb, float* dz_out, float* dxa,
                           float* dxb, float* dscale_a, float* doffset_a, float* dscale_b, float* doffset_b, float* coef,
                           float* ws, int64_t ws_bytes, void* stream);
/*
 * ssc_conv_forward for a launch whose output g is the gradient w.r.t. act(a*x+b) of a batch-statistics-normed
 * tensor x ([rows][ldx], addressed like the output): when the launch qualifies (uniform-tap kernel, no split-K
 * slabs, whole 16-byte aligned rows) its epilogue also writes rows of `partial` ([2][Nstore] each: sum dz,
 * sum dz*xhat with dz = g*act'(a*x+b)) and *nrows is their count; else *nrows = 0 (take the sums with
 * ssc_bn_act_backward).  partial_bytes >= nphase * ceil(M / 64) * 2 * Nstore * 4 always suffices.
 */
int ssc_conv_forward_bnbwd(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, const float* x, int ldx,
                           const float* ab, const float* stats, int act, float* partial, int64_t partial_bytes,
                           int* nrows, void* stream);

/*
 * The same for a launch whose output columns are the gradients w.r.t. TWO normed tensors side by side: columns [0, C0) belong to
 * x0 ([rows][ldx0], C0 channels), columns [C0, Nstore) to x1.  C0 must be a multiple of 128 (a column tile never straddles the
 * two).  Each tensor gets its own rows of partial sums ([2][C0] / [2][Nstore - C0] each); *nrows is the row count of either
 * (0: the launch did not qualify, take the sums with ssc_bn_act_backward).
 */
typedef struct ssc_bnbwd_site {
    const float* x;       /* the normed tensor (raw values), rows addressed like the launch's output */
    const float* ab;      /* [2][C]: a, b */
    const float* stats;   /* [2][C]: mean, 1/std */
    float* partial;       /* rows of [2][C] sums */
    int64_t partial_bytes;
    int32_t ldx;
    int32_t act;          /* SSC_ACT_* of the consumer whose gradient the columns are */
} ssc_bnbwd_site;
int ssc_conv_forward_bnbwd2(const ssc_conv_desc* d, float* ws, int64_t ws_bytes, const ssc_bnbwd_site* site0,
                            const ssc_bnbwd_site* site1, int C0, int* nrows, void* stream);

/* --- caption branch (text_lstm.hip): encode_feat_with_text, models_collection.py:150-248 --- */
/* tf.nn.embedding_lookup (:182) and its (dense) gradient; tok rows are time-major [T*N] */
int ssc_embedding_gather(const float* table, const int* tok, int rows, int C, float* out, void* stream);
int ssc_embedding_scatter_add(float* dtable, int vocab, const int* tok, int rows, int C, const float* g, void* stream);
/* tf.nn.l2_normalize over channels (:202,216); z = a*x+b when ab != NULL; ss[row] = sum z^2 */
int ssc_row_l2norm_fwd(const float* x, int ldx, const float* ab, int64_t M, int C, float* y, float* ss, void* stream);
int ssc_row_l2norm_bwd(const float* y, const float* ss, const float* dy, int64_t M, int C, float* dz, int accumulate,
                       void* stream);
/* BasicLSTMCell gate math (state_is_tuple=False, forget_bias=1) with the tf.cond pad-token skip (:235).
 * gates[row] = g0[row] + g1[row] + g2[row/div2]; acts = activated i,j,f,o kept for the backward pass. */
int ssc_lstm_pointwise_fwd(const float* g0, const float* g1, const float* g2, int div2, const int* mask, int mdiv,
                           const float* c_in, const float* h_in, int64_t rows, int C, float* c_out, float* h_out,
                           float* acts, void* stream);
/* One recurrent step in one launch: gates = h_in @ Kh (Kh: [C rows][4C], row stride ldk; skipped when with_gemm == 0,
 * i.e. h_in = 0) + g1[row] + g2[row/div2], then the cell above (tf.matmul + BasicLSTMCell of models_collection.py:230-236). */
int ssc_lstm_step_fwd(const float* h_in, const float* Kh, int ldk, const float* g1, const float* g2, int div2,
                      const int* mask, int mdiv, const float* c_in, int64_t rows, int C, int with_gemm, float* c_out,
                      float* h_out, float* acts, void* stream);
/* The same step on the bf16 matrix pipe, six bf16 products per fp32 product (fp32-grade results); C % 128 == 0.
 * Kp = the planes of Kh [C, 4C] from ssc_filter_split(Kh, 1, C, 4 * C, 0, ...), nbp = their blocks per plane
 * (ssc_filter_split_geom).  hp_in = the bf16 planes of h_in in the A-operand fragment layout (ssc_lstm_hsplit, or the previous
 * step's hp_out: ceil(rows / 64) * 2 * (C / 16) * 3 KiB -- whole 64-row tiles); NULL = h_in is zero, no product (with_gemm = 0 above).  hp_out (may be
 * NULL) receives the planes of h_out; acts may be NULL (inference: nobody reads the activated gates). */
int ssc_lstm_step_fwd_bf(const float* h_in, const void* hp_in, const void* Kp, int nbp, const float* g1, const float* g2,
                         int div2, const int* mask, int mdiv, const float* c_in, int64_t rows, int C, float* c_out,
                         float* h_out, void* hp_out, float* acts, void* stream);
int ssc_lstm_hsplit(const float* h, int64_t rows, int C, void* hp, void* stream);
int ssc_lstm_pointwise_bwd(const float* dh, const float* dc, const float* acts, const float* c_in, const float* c_out,
                           const int* mask, int mdiv, int64_t rows, int C, float* dg, float* dc_in, float* dh_pass,
                           float* gacc, void* stream);
/* relu(0.5*(log(1+1e-3+h) - log(1+1e-3-h)))  (:238-242) */
int ssc_squash_fwd(const float* h, int64_t n, float* o, void* stream);
int ssc_squash_bwd(const float* h, const float* o, const float* go, int64_t n, float* dh, void* stream);
/* out[g][c] (+)= sum of G consecutive rows (bias gradients, 6x6 tile reduction) */
int ssc_group_rowsum(const float* x, int ldx, int64_t groups, int G, int C, float* out, int accumulate, void* stream);
/* tf.reduce_mean over H,W of act(a*x+b) (:838) and the broadcast of its gradient */
int ssc_act_mean_hw(const float* x, const float* ab, int act, int N, int P, int C, float* out, void* stream);
int ssc_add_row_bcast(float* g, const float* v, float scale, int N, int P, int C, void* stream);
/* miu_relu (:63-65) fused with the [N,C*P] -> [N,P,C] reshape of the noise head (:493-499) */
int ssc_miu_permute_fwd(const float* pre, int N, int Cc, int P, float* out, void* stream);
int ssc_miu_permute_bwd(const float* pre, const float* g, int N, int Cc, int P, float* dpre, void* stream);

/* --- losses / optimizer (losses_optim.hip): graph_single.py:317-593 ---
 * loss_acc points to a DOUBLE device scalar that the kernels add into (zero it first). */
/* loss_acc += scale*sum softplus(sign*x[r*ld]); grad[r*ld] = gscale*sign*sigmoid(sign*x)   (:401-402) */
int ssc_softplus_loss(const float* x, int ld, int64_t rows, float sign, float scale, double* loss_acc, float* grad,
                      float gscale, void* stream);
/* sparse softmax CE (focal=0) or (1-p_true)^2 * CE (focal=1), mean over N, times coef   (:340-353) */
int ssc_acgan_loss(const float* logits, const int* labels, int N, int K, int focal, float coef, double* loss_acc,
                   float* dlogits, void* stream);
/* smooth-L1(img - gen) mean * coef (:551-555) + incoming discriminator gradient, through tanh' */
int ssc_gen_output_grad(const float* gen, int ldg, const float* img, int ldi, const float* gd, int ldd, int64_t npix,
                        float coef, double* loss_acc, float* dpre, void* stream);
/* ly.l2_regularizer: loss_acc += rate*sum(w^2)/2; grad += rate*w   (mru.py:55,60) */
int ssc_l2_reg(const float* w, int64_t n, float rate, double* loss_acc, float* grad, void* stream);
/* ---- Background_Colorization losses (bg_colorization_main.py:596-627), vanilla GAN on sigmoid(z) ------------------ */
/* mode 0: loss += scale*sum -log(sigmoid(z)+1e-12); mode 1: -log(1-sigmoid(z)+1e-12); dz = gscale * d/dz (may be NULL) */
int ssc_bg_gan_loss(const float* z, int64_t n, int mode, float scale, double* loss_acc, float* dz, float gscale,
                    void* stream);
/* count[0] = #(labels != 0): the pixels the masked L1 averages over (:612-616) */
int ssc_count_nonzero_i32(const int32_t* labels, int64_t n, float* count, float* workspace, int64_t workspace_bytes,
                          void* stream);
/* img = tanh(pre) [M,3]: loss += l1w * mean_{label != 0} |tgt - img|; dpre [M,4] = (dL1 + dgan [M,4]) * (1 - img^2) */
int ssc_bg_output_grad(const float* img, const float* tgt, const int32_t* labels, const float* count, float l1w,
                       const float* dgan, double* loss_acc, float* dpre, int64_t M, void* stream);
/* region-mask loss (:589-591): loss += w * mean softmax-CE(logits [M,K<=4], labels); dlogits [M, ldg] (pad columns 0) */
int ssc_seg_ce_loss(const float* logits, int K, const int32_t* labels, int64_t M, float w, double* loss_acc,
                    float* dlogits, int ldg, void* stream);
/* tf.train.AdamOptimizer dense apply (graph_single.py:588); lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the host, or read
 * from the device scalar lr_dev when it is not NULL (so a captured hipGraph can be replayed with a new step size) */
int ssc_adam_tf(float* var, const float* grad, float* m, float* v, int64_t n, float lr_t, const float* lr_dev,
                float beta1, float beta2, float eps, float gscale, void* stream);
/* the other optimizers get_optimizer offers (graph_single.py:584-593), TF dense-apply formulas: kind 1 RMSProp
 * (h0 = decay, h1 = momentum, h2 = epsilon; s1 = ms, s2 = mom), 2 Adagrad (s1 = accumulator), 3 Adadelta (h0 = rho,
 * h2 = epsilon; s1 = accum, s2 = accum_update).  lr is read from device memory; grad is scaled by gscale first. */
int ssc_optimizer_step(int kind, float* var, const float* grad, float* s1, float* s2, int64_t n, const float* lr_dev,
                       float h0, float h1, float h2, float gscale, void* stream);
/* spectral_normed_weight, one power iteration (sn.py:12-52) and its full gradient */
int ssc_sn_forward(const float* W, const float* u, int m, int n, float* v, float* u_new, float* wbar, float* aux,
                   void* stream);
int ssc_sn_backward(const float* W, const float* u, const float* v, const float* u_new, const float* aux,
                    const float* G, int m, int n, float* dW, int accumulate, float* scratch, void* stream);
/* the same for weights of any size (every conv / FC weight of the MRU discriminator): multi-workgroup, two-stage
 * fixed-order reductions.  forward workspace >= 64*n floats; backward workspace >= 1024+n floats, scratch m floats */
int ssc_sn_forward_any(const float* W, const float* u, int m, int n, float* v, float* u_new, float* wbar, float* aux,
                       float* workspace, int64_t workspace_bytes, void* stream);
int ssc_sn_backward_any(const float* W, const float* u, const float* v, const float* u_new, const float* aux,
                        const float* G, int m, int n, float* dW, int accumulate, float* scratch, float* workspace,
                        int64_t workspace_bytes, void* stream);
int ssc_axpy(float* y, const float* x, float a, int64_t n, void* stream);
/* fully_connected with <= 64 outputs (class logits, models_collection.py:839): y = x W + b, and its gradients */
int ssc_fc_small_fwd(const float* x, const float* W, const float* b, int N, int K, int J, float* y, void* stream);
int ssc_fc_small_bwd(const float* x, const float* W, const float* dy, int N, int K, int J, float* dx, float* dW,
                     float* db, int accumulate, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SKETCHYCOLOR_HIP_H */
