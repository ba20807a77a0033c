/*
 * include/ossid_hip.h -- C ABI of libossid_hip.so, the MI355X (gfx950) drop-in for OSSID's hot path.
 *
 * The reference (r-pad/OSSID_code) is pure Python and has no FFI of its own: each
 * entry point below names the Python interface it stands behind (paths relative to
 * /root/reference/python/ossid). INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the name ends in _host; the caller owns all memory,
 *     including workspaces (sizes from the *_workspace_bytes queries); nothing is allocated inside;
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls only enqueue work;
 *   - return 0 on success, a negative errno-style code otherwise (OSSID_EINVAL bad argument,
 *     OSSID_ELAUNCH a HIP launch error); no exceptions cross the boundary;
 *   - thread-safe for distinct streams; no global mutable state;
 *   - Zephyr entry points (ossid_zephyr_*, ossid_pn2_*): all arithmetic is IEEE binary32 with the operation order
 *     fixed by SPEC.md, so results are bit-identical to oracle/zephyr_oracle.c. DTOID entry points: binary32 tensors
 *     and accumulation; the convolution kernels' products as documented at ossid_conv_desc (`exact`).
 */
#ifndef OSSID_HIP_H
#define OSSID_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OSSID_OK 0
#define OSSID_EINVAL (-22)
#define OSSID_ELAUNCH (-5)

/* The version of the struct layouts and signatures below; bumped whenever one changes (3: ossid_conv_desc gained
 * scratch / scratch_bytes / exact; 5: ossid_wgrad_desc gained dy_add / dy_add_scale / dy_add_shift, ossid_dense_dgrad1_acc its dz_add arguments; 6: ossid_seq_replay / ossid_seq_op, ossid_bn_fold_bwd's zero_row). A binding compares it with ossid_abi_version() when it loads the library. */
#define OSSID_ABI_VERSION 6

/* library / device probe: returns OSSID_ABI_VERSION of the build; arch_out_host (may be NULL, >=32 bytes)
 * receives the gcnArchName of the current device, e.g. "gfx950:sramecc+:xnack-". */
int ossid_abi_version(char* arch_out_host, int len);

/* ---------------------------------------------------------------------------------------------
 * Z0  networkInference preprocessing          utils/zephyr_utils.py:13-14
 * cv2.GaussianBlur(img,(5,5),0) on u8 RGB [H,W,3] (blur=1) then /255 -> float, interleaved with
 * depth [H,W] (metres, 0 = invalid) into the staged frame rgbd[H,W,4] = (r,g,b,depth).
 * --------------------------------------------------------------------------------------------- */
int ossid_zephyr_prep_frame_u8(const uint8_t* img_rgb, const float* depth, int H, int W, int blur,
                               float* rgbd, void* stream);
/* same, for callers that already hold the float image (the tensor getPointNetData receives,
 * utils/zephyr_utils.py:14): img_rgb is float [H,W,3] in [0,1]. */
int ossid_zephyr_prep_frame_f32(const float* img_rgb, const float* depth, int H, int W, float* rgbd,
                                void* stream);

/* model table: tab[M][12] = (point xyz, normal xyz, HSV of the model colour, 3 pad)
 * from model_points / model_normals / model_colors [M,3] (utils/zephyr_utils.py:18-20). */
int ossid_zephyr_prep_model(const float* points, const float* normals, const float* colors_rgb, int M,
                            float* tab, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Z1  zephyr.utils.projectPointsUv(pose_hypos, model_points, meta_data)
 *     call site utils/zephyr_utils.py:58; K2meta utils/__init__.py:148-156
 * transforms [N,4,4] row-major float, points [M,3] -> uv [N,M,2] int32, uv[...,0]=x(col),
 * uv[...,1]=y(row); (-1,-1) marks a point at or behind the camera plane.
 * --------------------------------------------------------------------------------------------- */
int ossid_zephyr_project_uv(const float* transforms, const float* points, int N, int M, float fx, float fy,
                            float cx, float cy, int32_t* uv, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Z2  ScoreDataset.getPointNetData(data, return_uv_original)   call site utils/zephyr_utils.py:31
 * (a) free-space-violation count per hypothesis, the input of the inconst_ratio_th filter
 *     (scripts/online_learning.py:174,184,196; utils/zephyr_utils.py:42-43);
 * (b) features of the hypotheses sel[0..Nsel) (sel NULL = identity):
 *     point_x[Nsel][M][8] = (x, y, 0, dH, dS, dV, dD, cosN), uv_original[Nsel][M][2] (may be NULL).
 * interp: 0 nearest pixel (default), 1 bilinear colour/depth.
 * --------------------------------------------------------------------------------------------- */
int ossid_zephyr_inconst_count(const float* rgbd, int H, int W, const float* transforms, int N,
                               const float* tab, int M, float fx, float fy, float cx, float cy,
                               float margin, int32_t* count, void* stream);
int ossid_zephyr_featurize(const float* rgbd, int H, int W, const float* transforms, const int32_t* sel,
                           int Nsel, const float* tab, int M, float fx, float fy, float cx, float cy,
                           int interp, float* point_x, int32_t* uv_original, void* stream);

/* SURVEY 8f-2 (the step right before Z0): per-hypothesis ADD / ADI pose error
 *   pp_err = [err_func(R, t, R_gt, t_gt, model_points) for mat in poses_all]       scripts/online_learning.py:452
 * err_func = zephyr.utils.metrics.add (symmetric=0) or adi (symmetric=1); float64 like the numpy reference.
 * transforms [N,4,4], transform_gt [4,4], points [M,3] (M*24 B <= 150 KB for ADI) -> err [N]. */
int ossid_pose_errors(const double* transforms, const double* transform_gt, const double* points, int N, int M,
                      int symmetric, double* err, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Z3  zephyr.models.pointnet2.PointNet2SSG.forward({"point_x": ...})
 *     ctor scripts/online_learning.py:212-227; call utils/zephyr_utils.py:34
 * Stage entry points (pointnet2_ops: furthest_point_sample, ball_query) and the whole scorer.
 * --------------------------------------------------------------------------------------------- */
/* xyz [B][n][stride] (first 3 floats of each row) -> idx [B][npoint], new_xyz [B][npoint][3] */
int ossid_pn2_fps(const float* xyz, int stride, int B, int n, int npoint, int32_t* idx, float* new_xyz,
                  void* stream);
/* idx [B][npoint][nsample], nsample must be 64 */
int ossid_pn2_ball_query(const float* xyz, int stride, int B, int n, const float* new_xyz, int npoint,
                         float radius, int nsample, int32_t* idx, void* stream);

/* Packed weight blob (built on the host by ossid_code_amd.zephyr.pack_pn2_weights, uploaded once):
 * float offsets into `blob` of the 12 layers' packed weights and biases, SPEC.md 4.3/4.4. */
typedef struct ossid_pn2_weights {
    const float* blob;
    int64_t w_off[12];
    int64_t b_off[12];
    int64_t wxyz2_off; /* SA2 L1 xyz columns, [3][128] */
    int32_t npoint1, npoint2; /* 512, 128 (npoint2 % 32 == 0, npoint1 % 32 == 0) */
    float radius1, radius2;   /* 0.2, 0.4 */
} ossid_pn2_weights;

size_t ossid_pn2_workspace_bytes(int B, int M, int npoint1, int npoint2);

/* point_x [B][M][8] -> scores [B]. workspace >= ossid_pn2_workspace_bytes(...), 256-byte aligned.
 * dbg_* (all may be NULL) receive copies of stage results for parity tests:
 * fps1 [B][np1] i32, ball1 [B][np1][64] i32, feat1 [B][np1][128], fps2 [B][np2] i32,
 * ball2 [B][np2][64] i32, feat2 [B][np2][256], feat3 [B][1024]. */
int ossid_pn2_score(const float* point_x, int B, int M, const ossid_pn2_weights* w, void* workspace,
                    size_t workspace_bytes, float* scores, int32_t* dbg_fps1, int32_t* dbg_ball1,
                    float* dbg_feat1, int32_t* dbg_fps2, int32_t* dbg_ball2, float* dbg_feat2,
                    float* dbg_feat3, void* const* stage_events_host, void* stream);

/* Kernel order of ossid_pn2_score; with stage_events_host != NULL (an array of OSSID_PN2_NSTAGES+1 hipEvent_t made
 * by ossid_event_create) event[i] is recorded on `stream` before stage i and event[NSTAGES] after the last, so a
 * caller can time each kernel of the launch it is actually measuring (bench.py's roofline leg). */
#define OSSID_PN2_NSTAGES 9
/* "fps1,ball1,sa1,p2,fps2,ball2,sa2,sa3,fc" */
const char* ossid_pn2_stage_names(void);

int ossid_event_create(void** event_out_host);
int ossid_event_destroy(void* event);
int ossid_event_record(void* event, void* stream);
/* waits for `stop`, then *ms_out_host = elapsed(start, stop) */
int ossid_event_elapsed_ms(void* start, void* stop, float* ms_out_host);

/* Names of the kernels the scorer launches, for profile post-processing (static string). */
const char* ossid_pn2_kernel_names(void);

/* =============================================================================================
 * DTOID ops (paths under /root/reference/python/ossid/models/dtoid)
 * ============================================================================================= */

/* D5  conv2d_dw_group(x, kernel, padding=1)      network.py:186-192 (backbone) and :365-371 (head)
 * F.conv2d(x.view(1,B*C,H,W), k.view(B*C,1,3,3), groups=B*C, padding=1): per-plane 3x3 cross-correlation with a
 * data-dependent kernel. planes = B*C; x, out, dout, dx [planes][H][W]; k, dk [planes][3][3].
 * _bwd_x: gradient w.r.t. x; _bwd_k: gradient w.r.t. the kernel (gradients flow to both operands). */
int ossid_dw_xcorr_fwd(const float* x, int x_planes, const float* k, int planes, int H, int W, float* out,
                       void* stream);   /* x_planes = planes, or C when ONE image [C][H][W] is shared by all B kernels */
/* the same correlation, channels-last, ONE image x [H][W][channels] against `batch` kernel sets k [batch][channels][3][3]
 * (test time: the image broadcast over the templates) -> out [batch][H][W][channels]; channels % 4 == 0 */
int ossid_dw_xcorr_nhwc_bcast(const float* x, const float* k, int batch, int channels, int H, int W, float* out,
                              void* stream);
int ossid_dw_xcorr_bwd_x(const float* dout, const float* k, int planes, int H, int W, float* dx, void* stream);
int ossid_dw_xcorr_bwd_k(const float* x, const float* dout, int planes, int H, int W, float* dk, void* stream);

/* D4, D6-D8  the dense convolutions at test time: the head's 3x3 layers (network.py:102-110, :135-143, :288-326)
 * and the DenseNet-121 blocks of the image backbone (network.py:164-184), channels-last, f32 in and out, on the matrix cores
 * (arithmetic of the reduction: `exact`, below).
 *   out[b][y][x][out_channel_offset + co] = post( act( bias[co] + sum_{ci,tap} w[co][ci][tap] * pre(x)[b][y+dy][x+dx][ci] ) )
 * taps 9: 3x3 / stride 1 / padding 1 (zero halo);  taps 1: 1x1.
 * pre  = x * pre_scale[ci] + pre_shift[ci] (then ReLU if pre_relu) on real pixels only -- eval-mode BatchNorm(+ReLU)
 *        in FRONT of the conv (DenseNet's BN-ReLU-Conv); NULL = identity.
 * act  = 0 none, 1 ELU(alpha 1), 2 ReLU (SqueezeNet's Fire modules);  post = * post_scale[co] + post_shift[co] -- eval-mode BatchNorm BEHIND the ELU
 *        (`norm(F.elu(conv(x)))`); NULL = identity.
 * src_height/width > 0 (3x3 only): x is [B][src_h][src_w][..] and is nearest-neighbour up-sampled to [H][W] on the fly
 *        (F.interpolate(mode="nearest") in front of the conv, network.py:354-357).
 * in_channel_stride / out_channel_stride (0 = cin / cout) and out_channel_offset let a layer read the first cin channels of
 *        a wider resident buffer and append its output to it in place (DenseNet concatenation without torch.cat).
 * taps 4: the four PHASES of a 3x3 / pad 1 convolution applied to a 2x nearest-neighbour up-sampled input (the decoder,
 *        network.py:354-356): output pixel (2i+a, 2j+b) only sees source pixels (i+a-1, i+a) x (j+b-1, j+b), so each phase
 *        is a 2x2 convolution of the SOURCE with row/column-merged weights -- 4/9 of the multiply-adds. height/width are
 *        the SOURCE size, out is [B][2*height][2*width][..]; wpk = four ossid_conv_pack_weights(.., taps 4) sets, phase
 *        2a+b, of the merged [cout][cin][2][2] kernels (rows: a=0 -> (w0, w1+w2), a=1 -> (w0+w1, w2); columns likewise).
 * cin % 16 == 0; cout, strides and offset % 4 == 0. wpk = ossid_conv_pack_weights(w [cout][cin][kh][kw]). */
typedef struct ossid_conv_desc {
    const float* x;
    const float* wpk;
    const float* bias;
    const float* pre_scale;
    const float* pre_shift;
    const float* post_scale;
    const float* post_shift;
    float* out;
    int32_t batch, height, width, cin, cout, taps, act, pre_relu;
    int32_t src_height, src_width, in_channel_stride, out_channel_stride, out_channel_offset;
    int32_t pre_batch_stride;   /* floats between the pre_scale / pre_shift rows of consecutive images; 0 = one row */
    int64_t in_batch_stride;    /* floats between consecutive images of x; < 0 = dense (src_h * src_w * channel stride);
                                   0 = ONE image shared by the whole batch (a per-image pre-affine then makes each
                                   batch entry a different affine view of it: image_feat * avg_t, image_feat - avg_t
                                   of network.py:344-347 without materialising them) */
    /* Scratch a launch may use (optional): the Winograd entry's tail split (ossid_conv3x3_wino_workspace_bytes) keeps the
     * raw sums of its slices here; -DOSSID_TIMING diagnostic builds write per-wave time stamps. scratch_bytes = its size. */
    void* scratch;
    int64_t scratch_bytes;
    /* Arithmetic of the reduction in ossid_conv_nhwc_fwd (csrc/conv.hip). 0 (default): every f32 product w*x is formed as
     * three bf16 matrix-core products (w = hi + lo, x = hi + lo as bf16 pairs; w_lo*x_hi + w_hi*x_lo + w_hi*x_hi accumulated
     * in f32; the dropped lo*lo term is ~2^-16 of a product): ~5e-6 of the output scale against float64, tests hold 2e-5.
     * 1: v_mfma_f32_32x32x2_f32, exact f32 products (~1e-6), 5x the matrix-pipe time. 2: the three-way split -- operands as
     * three bf16 pieces (24 significant bits: the f32 value itself), six products, f32-level accuracy (~1e-6) at 2x the
     * pipe time of form 0 -- for layers whose output feeds a hard decision that a later pass repeats (the training forward
     * of the ReLU / max-pool networks: a pre-activation that lands on the other side of zero changes which units the
     * gradient flows through). wpk must be packed for the same form (ossid_conv_pack_weights_form; form 2 needs
     * ossid_conv_packed_floats_form floats, 1.5x the others). A library built with -DOSSID_CONV_F32 runs every launch exact;
     * ossid_conv_split_bf16() says whether the split form exists. The Winograd entry ignores the field (its own build
     * switch, below). */
    int32_t exact;
} ossid_conv_desc;
int ossid_conv_split_bf16(void);
size_t ossid_conv_packed_floats(int Cout, int Cin, int taps);
size_t ossid_conv_packed_floats_form(int Cout, int Cin, int taps, int exact);
/* w [Cout][Cin][taps] -> the operand layout of ossid_conv_nhwc_fwd (common.h, ossid_conv_pack_quad). dgrad != 0: the layer
 * of the DATA gradient (Cin output channels, Cout reduction channels, taps reversed; needs ossid_conv_packed_floats(Cin,
 * Cout, taps) floats). exact: the form of the launches that will read it. ossid_conv_pack_weights = (dgrad 0, exact 0). */
int ossid_conv_pack_weights_form(const float* w, int Cout, int Cin, int taps, int dgrad, int exact, float* wpk, void* stream);
int ossid_conv_pack_weights(const float* w, int Cout, int Cin, int taps, float* wpk, void* stream);
int ossid_conv_nhwc_fwd(const ossid_conv_desc* desc_host, void* stream);

/* D6-D8 / D16  the same 3x3 convolution (stride 1, pad 1; ossid_conv_desc with taps = 9, no fused up-sampling, no
 * training extras) as Winograd F(2x2, 3x3): 16 multiplies per 2x2 output tile and channel pair instead of 36, f32
 * throughout (csrc/wino.hip; results differ from ossid_conv_nhwc_fwd by rounding only). Stands behind the same nn.Conv2d
 * layers (network.py:102-110, :135-143, :288-326) and their data gradients in the finetune step. desc->wpk must be the
 * layout of ossid_conv_pack_weights_wino: U = G g G^T of w [Cout][Cin][3][3]; dgrad != 0 packs the data gradient's layer
 * (Cin output channels, Cout reduction channels, filter rotated by 180 degrees; the desc then carries cin = Cout,
 * cout = Cin). The reduction channel count must be a multiple of 16.
 * Arithmetic of the channel reduction: by default every f32 product U*V is formed as three bf16 matrix-core products
 * (U = hi + lo, V = vh + vl as bf16 pairs; lo*vh + hi*vl + hi*vh accumulated in f32; the dropped lo*vl term is ~2^-16 of
 * a product): measured 7e-6 of the output scale against float64, tests hold <= 2e-5. A library built with
 * -DOSSID_WINO_F32 runs the same layout on v_mfma_f32_32x32x2_f32 (1e-6). ossid_conv_wino_split_bf16 says which. */
int ossid_conv_wino_split_bf16(void);
size_t ossid_conv_wino_packed_floats(int Cout, int Cin);
int ossid_conv_pack_weights_wino(const float* w, int Cout, int Cin, int dgrad, float* wpk, void* stream);
int ossid_conv3x3_wino_fwd(const ossid_conv_desc* desc_host, void* stream);
/* Scratch the Winograd launch wants for cutting the TAIL of its grid along the reduction (csrc/wino.hip: the last, partial
 * round of resident workgroups runs as ks slices per workgroup + a finishing launch; 1 576 workgroups on 512 slots cost
 * ~3.3 rounds instead of 4). 0 = no split planned for this shape. Pass the buffer in desc->scratch and its size in
 * desc->scratch_bytes (for the pair entry: in the first descriptor); without it the launch runs whole
 * workgroups. Results are bit-reproducible either way (fixed summation order), but differ in rounding between the two forms. */
size_t ossid_conv3x3_wino_workspace_bytes(const ossid_conv_desc* desc);
size_t ossid_conv3x3_wino_pair_workspace_bytes(const ossid_conv_desc* d0, const ossid_conv_desc* d1);
/* two independent layers (the i-th convolutions of the classification and the regression trunk, network.py:113-121 /
 * :146-154) in ONE grid: their workgroups fill the chip's slots together instead of each launch ending in a ragged round */
int ossid_conv3x3_wino_fwd_pair(const ossid_conv_desc* desc0_host, const ossid_conv_desc* desc1_host, void* stream);

/* D6 (tail)  the last two layers of the segmentation decoder in one launch (network.py:357-362):
 *   out[b][y][x] = b2 + conv3x3_{16->1}( post( ELU( b1 + conv3x3_{32->16}( nearest_upsample(x -> [H][W]) ) ) ) )
 * x [B][src_height][src_width][in_channel_stride] channels-last (the first 32 channels are read); w1p =
 * ossid_seg_tail_pack_weights(w1 [16][32][3][3]); post = * post_scale[16] + post_shift[16] (eval-mode BatchNorm);
 * w2 [16][3][3] (the [1][16][3][3] weight); b2 [1] (device, may be NULL); out [B][H][W]. Both convolutions zero-pad at the [H][W] border. The
 * 16-channel full-resolution tensor is never materialised. Returns OSSID_EINVAL when the up-sampling ratio is
 * below ~1.5 (the source footprint of a tile would not fit the staged patch): run the two layers separately then.
 * Arithmetic: the 32 -> 16 convolution on split-bf16 matrix-core products (as ossid_conv_desc.exact = 0: 8e-6 of the output
 * scale measured against float64; -DOSSID_SEGTAIL_F32 builds: exact f32, 1e-6), the 16 -> 1 convolution as f32 fmaf chains. */
int ossid_seg_tail_split_bf16(void);
size_t ossid_seg_tail_packed_floats(void);
int ossid_seg_tail_pack_weights(const float* w1, float* w1p, void* stream);
int ossid_seg_tail_fwd(const float* x, int batch, int src_height, int src_width, int in_channel_stride, int height,
                       int width, const float* w1p, const float* b1, const float* post_scale, const float* post_shift,
                       const float* w2, const float* b2, float* out, void* stream);

/* D16  weight gradient of the 3x3 / stride 1 / pad 1 convolution (loss.backward() of the finetune step,
 * scripts/online_learning.py:670-672): dw[co][ci][ky][kx] (+)= sum_{b,y,x} dy[b][y][x][co] * x[b][y+ky-1][x+kx-1][ci].
 * x [B][H][W][in_channel_stride], dy [B][H][W][dy_channel_stride] channels-last (0 = Cin / Cout); dw in torch layout
 * [Cout][Cin][3][3]; accumulate != 0 adds to dw. Split-K partial slabs in `workspace`, summed in a fixed order
 * (bit-reproducible). The data gradient needs no entry point of its own: it is ossid_conv_nhwc_fwd on dy with the
 * 180-degree-rotated, transposed weights. */
int ossid_conv3x3_wgrad_splits(int B, int H, int W, int Cin, int Cout);
size_t ossid_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ossid_conv3x3_wgrad(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int in_channel_stride,
                        int dy_channel_stride, void* workspace, size_t workspace_bytes, float* dw, int accumulate,
                        void* stream);

/* D16  weight gradient of a 3x3 / pad 1 or 1x1 convolution whose INPUT carried a fused prologue in the forward pass
 * (training-mode BatchNorm folded to a per-channel affine, + ReLU: DenseNet's BN-ReLU-Conv, models/dtoid/network.py:164-184;
 * the head's `norm(F.elu(conv(x)))` feeding the next conv, :330-357):
 *   dw[co][ci][tap] (+)= sum_{b,y,x} dy[b][y][x][co] * P(x)[b][y+dy][x+dx][ci],  P(v) = relu?(v * pre_scale[ci] + pre_shift[ci])
 * on real pixels, zero outside the image (pre_scale NULL = identity). x [B][H][W][in_channel_stride], dy
 * [B][H][W][dy_channel_stride] channels-last (0 = cin / cout); dw in torch layout [cout][cin][kh][kw]. Operands are staged
 * through LDS with the pixels on K: as bf16 hi/lo images read back K-major (ds_read_b64_tr_b16) for three
 * v_mfma_f32_32x32x16_bf16 per f32 product (split-bf16, ~5e-6 of the result's scale; ossid_conv_wgrad_split_bf16() != 0),
 * or as f32 for v_mfma_f32_32x32x2_f32 (every layer of a -DOSSID_WGRAD_F32 build); split-K slabs in `workspace`
 * (ossid_conv_wgrad_workspace_bytes), summed in a fixed order -- bit-reproducible, no float atomics. cin, cout % 4 == 0. */
typedef struct ossid_wgrad_desc {
    const float* x;
    const float* dy;
    const float* pre_scale;
    const float* pre_shift;
    float* dw;
    void* workspace;
    size_t workspace_bytes;
    int32_t batch, height, width, cin, cout, taps, pre_relu, accumulate;
    int32_t in_channel_stride, dy_channel_stride;
    int32_t src_height, src_width;   /* > 0 (3x3 only): x is [B][src_h][src_w][..], nearest-neighbour up-sampled to
                                        [height][width] on the fly, as in the forward (ossid_conv_desc) */
    /* ABI 5. Optional: the output gradient is dy + dy_add_scale[co] * dy_add + dy_add_shift[co], formed while dy is staged
     * (dy_add [B][H][W][dy_channel_stride] like dy): the statistics term of a training BatchNorm's backward, dz = g + c_x y + c_1,
     * without a pass that materialises dz. Only the 1x1 problems csrc/wgrad_t9.hip takes (a dense layer's c -> 128) support it;
     * any other problem with dy_add != NULL is rejected. */
    const float* dy_add;
    const float* dy_add_scale;
    const float* dy_add_shift;
} ossid_wgrad_desc;
int ossid_conv_wgrad_split_bf16(void);
size_t ossid_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps);
int ossid_conv_wgrad(const ossid_wgrad_desc* desc_host, void* stream);
/* Up to 48 INDEPENDENT weight gradients (e.g. the 2 x L of one DenseNet block, each too small to fill the chip) as one
 * launch per tiling variant plus one grouped slab reduction; each descriptor's `workspace` field is ignored, the slabs
 * of all problems live in the one `workspace` of ossid_conv_wgrad_group_workspace_bytes(descs, n) bytes. */
size_t ossid_conv_wgrad_group_workspace_bytes(const ossid_wgrad_desc* descs_host, int n);
int ossid_conv_wgrad_group(const ossid_wgrad_desc* descs_host, int n, void* workspace, size_t workspace_bytes, void* stream);

/* D16  packed weights of the DATA gradient: dx = conv(dy, W') with W'[ci][co][tap] = w[co][ci][taps-1-tap] (transposed,
 * rotated by 180 degrees), in the layout ossid_conv_nhwc_fwd reads for a [cout' = Cin][cin' = Cout] layer -- the data
 * gradient itself is that forward kernel on dy. Cout % 16 == 0; wpk has ossid_conv_packed_floats(Cin, Cout, taps) floats. */
int ossid_conv_pack_weights_dgrad(const float* w, int Cout, int Cin, int taps, float* wpk, void* stream);

/* D16  one generic channels-last pass of the training step (rows = B*H*W pixels, `channels` % 4 == 0):
 *     m   = 1 (mask_mode 0) | [mask_scale[c] * x + mask_shift[c] > 0] (1: ReLU behind a folded BatchNorm)
 *           | (x > 0 ? 1 : x + 1) (2: ELU'(v) written in terms of x = ELU(v)) | [x > 0] (3: ReLU'(v) in terms of x = ReLU(v))
 *     r   = (alpha[c] * g + beta[c] * x + kappa[c]) * m              (NULL alpha/beta/kappa = 1 / 0 / 0)
 *     out = r, or out + r when accumulate != 0                        (out NULL: sums only)
 *     sums[0][c], sums[1][c] = sum over rows of (g*m, g*m*x) (sum_mode 1), (r, r*x) (sum_mode 2) or, for BatchNorm batch
 *           statistics, (g - pivot[c], (g - pivot[c])^2) (sum_mode 3: sums about a per-channel pivot -- the tensor's first
 *           row -- so that a channel whose spread is small against its mean loses nothing to E[x^2] - E[x]^2; the
 *           pivot is copied to a third row sums[2][c]); 0 = none
 * Instances: BatchNorm batch statistics (g = x, sum_mode 1: sum x, sum x^2); backward of ELU + BatchNorm statistics
 * (alpha 1, beta/kappa from ossid_bn_fold_bwd, mask 2, sum_mode 2 -> the bias gradient); backward of a folded
 * BatchNorm+ReLU prologue (g = d out of the data gradient, mask 1, alpha = scale, sum_mode 1 -> d shift, d scale);
 * DenseNet's concatenation gradient (accumulate into the block's gradient buffer). Column sums go through per-block
 * partials [ossid_chan_op_partials(rows, channels)][2][channels] floats and are combined in a fixed order in double. */
typedef struct ossid_chan_op_desc {
    const float* g;
    const float* x;
    float* out;
    const float* alpha;
    const float* beta;
    const float* kappa;
    const float* mask_scale;
    const float* mask_shift;
    float* partials;
    float* sums;
    const float* pivot;              /* sum_mode 3: [channels] floats the statistics are taken about (row 0 of the tensor) */
    int64_t n_rows;
    int32_t channels, g_stride, x_stride, out_stride, mask_mode, accumulate, sum_mode;
    int32_t sums_row_stride;         /* floats between sums[0][.] and sums[1][.] (0 = channels): lets a layer write the
                                        statistics of its channel slice into a block-wide [2][C_total] table */
    int32_t defer_finalize;          /* != 0: leave the column sums as the ossid_chan_op_partials(rows, channels)
                                        per-block partials in `partials` (sums may be NULL); the consumer --
                                        ossid_bn_fold_fwd / _bwd with n_partials > 0 -- combines them itself */
} ossid_chan_op_desc;
int ossid_chan_op_partials(long long n_rows, int channels);
int ossid_chan_op(const ossid_chan_op_desc* desc_host, void* stream);

/* Zero `bytes` bytes at `ptr` on `stream` (an accumulator a recorded launch sequence must clear on every replay: the
 * coefficient table of a dense block's backward pass; replaces torch.zeros inside such a sequence). */
int ossid_fill_zero(void* ptr, size_t bytes, void* stream);

/* Replay of a recorded launch sequence. The finetune step (scripts/online_learning.py:650-679) is ~1 800 small launches
 * of fixed shape from one host thread; a binding records the launches one fixed-shape piece makes (a DenseNet block's
 * forward, a template encoder's backward: models/dtoid/network.py:160-279) ONCE, as calls of this library's own
 * stream-taking entry points with their argument values, and hands the list back here every step: the loop below re-issues
 * them with the current stream handles, so a recorded launch costs the host its hipLaunchKernel and nothing else.
 *   fn        one of this library's entry points of the form int f(..., void* stream) (every argument an integer, a
 *             pointer, a size_t, a float or a double; the stream LAST), or NULL for a stream-order op: streams[slot]
 *             waits for everything streams[wait_for] holds when the op is reached (an event owned by the op, created at
 *             its first replay; ossid_seq_release destroys it);
 *   iarg      the integer-class arguments in order, WITHOUT the stream (pointers and sign-extended integers);
 *   fparg     the floating-point arguments in order: a float's bits in the low 32, a double's in all 64;
 *   slot      index into the replay's stream table of the stream the call gets.
 * Descriptor structs a call points to (ossid_conv_desc ...) are read at every replay: they, and all device memory they
 * name, must stay alive and at the same addresses for as long as the sequence is replayed. Returns the first non-zero
 * status (failed_at_host, may be NULL, then holds the op's index; -1 after a clean replay). x86-64 System V hosts only. */
#define OSSID_SEQ_MAX_INT 24
#define OSSID_SEQ_MAX_FP 8
typedef struct ossid_seq_op {
    void* fn;
    void* event;
    int32_t slot, wait_for, n_int, n_fp;
    uint64_t iarg[OSSID_SEQ_MAX_INT];
    uint64_t fparg[OSSID_SEQ_MAX_FP];
} ossid_seq_op;
int ossid_seq_replay(ossid_seq_op* ops_host, int n, void* const* streams_host, int n_streams, int* failed_at_host);
int ossid_seq_release(ossid_seq_op* ops_host, int n);
/* Calling-convention probe for the tests of ossid_seq_replay (no device work): writes its 14 arguments and the stream
 * handle, each converted to double, to out_host[0..14]. */
int ossid_seq_probe(int32_t i0, float f0, const void* p1, double d1, int64_t l2, int32_t i3, float f2, size_t s4, int32_t i5,
                    int32_t i6, int32_t i7, double d3, int32_t i8, int64_t l9, double* out_host, void* stream);

/* Segmentation term of DtoidNet.forward's loss and its metric (models/dtoid/__init__.py:210-232) in one pass:
 *   prob = sigmoid(logit) [B][hw];  out[0] = BCELoss(prob, mask) (mean over B*hw, torch's clamps: log terms >= -100);
 *   out[1 + b] = IoU of (prob > 0.5) vs (mask > 0) of image b (0 for an empty union: pl.metrics iou(ignore_index=0));
 *   dlogit_sum [B][hw] = d(SUM of the BCE terms)/d(logit) -- the backward pass scales it by upstream / (B*hw).
 * workspace >= ossid_seg_bce_iou_workspace_bytes(B), 8-byte aligned. Deterministic (fixed-order double sums). */
size_t ossid_seg_bce_iou_workspace_bytes(int B);
int ossid_seg_bce_iou_fwd(const float* logit, const float* mask, int B, long long hw, float* prob, float* dlogit_sum,
                          float* out, void* workspace, size_t workspace_bytes, void* stream);

/* 3x3 / pad 1 convolution with ONE output channel in training -- the decoder's `seg_final` (nn.Conv2d(16, 1, 3, padding=1),
 * models/dtoid/network.py:326, :362): x [B][H][W][C] channels-last (C % 4 == 0), w [C][3][3] = the parameter's layout,
 * out / g [B][H][W]. Forward (+ bias), data gradient, weight + bias gradient (fixed-order sums through `workspace` >=
 * ossid_conv3x3_c1_wgrad_workspace_bytes()). Vector-ALU kernels: a one-row output is no matrix-core shape. */
int ossid_conv3x3_c1_fwd(const float* x, int B, int H, int W, int C, const float* w, const float* bias, float* out, void* stream);
int ossid_conv3x3_c1_dgrad(const float* g, int B, int H, int W, int C, const float* w, float* dx, void* stream);
size_t ossid_conv3x3_c1_wgrad_workspace_bytes(void);
int ossid_conv3x3_c1_wgrad(const float* x, const float* g, int B, int H, int W, int C, void* workspace, size_t workspace_bytes,
                           float* dw, float* db, void* stream);

/* Convolution weights [cout][cin][k][k] -> the [cout][kpad] matrix in ossid_im2col_stem's column order
 * ((ky * k + kx) * cin + c, zero-padded), or back (inverse = 1: a weight GRADIENT computed on the im2col columns returns to
 * the parameter's layout). The strided stems (7x7 / 2 of the image backbone, network.py:164-170; 3x3 / 2 of the template
 * encoders, :203-208) run as im2col + a 1x1 MFMA convolution. */
int ossid_stem_weight_relayout(const float* src, float* dst, int cout, int cin, int k, int kpad, int inverse, void* stream);

/* D16  training-mode BatchNorm2d (nn.BatchNorm2d in train(), online_learning.py:656) folded into the per-channel affine
 * the NEXT convolution applies while staging its input: from sums = (sum x, sum x^2) over n rows,
 *   (sums[c], sums[sums_row_stride + c]; 0 = C; taken about pivot[c] when pivot != NULL: mean = pivot + S1/n,
 *   var = S2/n - (S1/n)^2) mean, biased var -> scale = gamma * rstd, shift = beta - mean * scale; running statistics updated in place with
 *   `momentum` (unbiased variance), as torch does. Backward: (d scale, d shift) -> d gamma, d beta and the coefficients of
 *   the statistics' own gradient  dx += coef_x[c] * x + coef_1[c]  (= d mean / n + 2 (x - mean) d var / n), which the
 *   producer's ossid_chan_op pass applies (accumulate != 0: += onto coef_x / coef_1, several consumers of one tensor).
 *   n_partials > 0: the two sums (forward: sum x, sum x^2; backward: d shift, d scale) are read as the per-block partials a
 *   deferred ossid_chan_op left in `partials` and combined here, in the same fixed order. */
int ossid_bn_fold_fwd(const float* sums, int sums_row_stride, const float* partials, int n_partials, const float* pivot, int C,
                      double n, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, void* stream);
/* ossid_bn_fold_fwd for a dense block's norm1: `table` [3][row_stride] (sums, sums of squares, pivots: what ossid_chan_op sum_mode 3
 * finalizes) is complete for the first tail_c0 channels; the last C - tail_c0 channels still are the partial rows of a DEFERRED
 * ossid_chan_op over the slab the previous layer appended (tail_partials [n_partials][2][C - tail_c0], tail_pivot = that launch's
 * pivot row): they are finalized into the table here and all C channels folded -- one launch instead of finalize + fold.
 * tail_c0 % 32 == 0. */
int ossid_bn_fold_fwd_tail(float* table, int row_stride, int tail_c0, const float* tail_partials, int n_partials,
                           const float* tail_pivot, int C, double n, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out,
                           void* stream);
/* zero_row (may be NULL): C floats set to 0 by the same launch -- the unused third row of the [3][C] gradient a caller hands
 * back for the statistics (constant term, coefficient of x, pivot: the pivot has no gradient). */
int ossid_bn_fold_bwd(const float* dscale, const float* dshift, const float* partials, int n_partials, const float* gamma,
                      const float* mean, const float* rstd, int C, double n, float* dgamma, float* dbeta, float* coef_x,
                      float* coef_1, int accumulate, float* zero_row, void* stream);

/* sums[c], sums[sums_row_stride + c] = the column sums left as n_partials partial rows [n][2][C] by ossid_chan_op
 * (defer_finalize), combined in a fixed order in double. */
int ossid_colsum_finalize(const float* partials, int n_partials, int C, float* sums, int sums_row_stride, void* stream);

/* D16  all convolution weights of a training step re-packed in ONE launch (they change every optimizer step): a device
 * table with one row per (layer, layout): kind 0 = the forward layout of ossid_conv_pack_weights, 1 = the data-gradient
 * layout of ossid_conv_pack_weights_dgrad, 2 / 3 = ossid_conv_pack_weights_wino with dgrad = 0 / 1 (taps = 9), 4 / 5 = kinds
 * 0 / 1 for exact launches (ossid_conv_desc::exact = 1), 6 / 7 = for three-way-split launches (exact = 2); first_block = prefix sum of ceil(packed float4 / 256) over the rows before. */
typedef struct ossid_pack_row {
    const float* w;
    float* wpk;
    int64_t first_block;
    int32_t cout, cin, taps, kind;
} ossid_pack_row;
int ossid_conv_pack_weights_table(const ossid_pack_row* rows_device, int n_rows, long long total_blocks, void* stream);

/* D16  backward of a dense layer's 1x1 convolution (torchvision _DenseLayer.conv1 inside ImageFeatExtract, network.py:164-184)
 * with the block's gradient accumulation fused in (csrc/dense_bwd.hip): for the first c channels of the gradient buffer G
 * [n_rows][channel_stride],  G[r][ch] += alpha[ch] * m * (dz[r][:] . W1[:, ch]),  m = (mask_scale[ch] x[r][ch] + mask_shift[ch] > 0)
 * -- the data gradient of conv1 (c -> 128; wpk_dgrad = ossid_conv_pack_weights_dgrad(w [128][c][1])) through relu(norm1(.)) --
 * and norm1's column sums as partial rows for ossid_bn_fold_bwd: partials [P][2][c], row 0 = sum of g m, row 1 = sum of g m x
 * (g = the raw data gradient), P = ossid_dense_dgrad1_acc_partials(n_rows). What ossid_conv_nhwc_fwd (data-gradient weights)
 * followed by ossid_chan_op (mask_mode 1, accumulate, sum_mode 1) computes, without the [n_rows][c] tensor in between; same
 * three-product bf16 arithmetic, f32 sums in another order. c % 32 == 0, c <= 1024, n_rows * channel_stride < 2^32; returns
 * OSSID_EINVAL in a -DOSSID_CONV_F32 build. dz_add (NULL = none): dz is dz + dz_add_scale[k] * dz_add + dz_add_shift[k], formed
 * while dz is staged (dz_add [n_rows][128]: see ossid_wgrad_desc.dy_add). */
/* ... and its FORWARD with norm2's batch statistics in the epilogue: y1 [n_rows][128] = conv1(relu(pre_scale x + pre_shift)) on
 * the first c channels of x [n_rows][channel_stride], wpk_x6 = ossid_conv_pack_weights_form(w [128][c][1], exact = 2) (the
 * three-way split: f32-level accuracy), and P = ossid_dense_fwd1_stats_partials(n_rows) partial rows [P][3][128] = per-channel
 * (sum of (y1 - p_w), sum of (y1 - p_w)^2, p_w) about each workgroup's OWN pivot p_w (its first output of the channel) with
 * counts [P] = pixels per row: ossid_conv_nhwc_fwd (exact = 2) followed by ossid_chan_op (sum_mode 3) without the second read of
 * y1. ossid_bn_fold_fwd_rows is ossid_bn_fold_fwd for such rows (moved to one common pivot in double). c % 32 == 0. */
int ossid_dense_fwd1_stats_partials(long long n_rows);
int ossid_dense_fwd1_stats(const float* x, int channel_stride, int c, const float* pre_scale, const float* pre_shift,
                           const float* wpk_x6, long long n_rows, float* y1, float* partials, float* counts, void* stream);
int ossid_bn_fold_fwd_rows(const float* partials, const float* counts, int n_partials, int C, double n, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                           float* shift, float* mean_out, float* rstd_out, void* stream);
/* ... and the layer's 3x3 data gradient (32 -> 128) with norm2 / ReLU's backward in its epilogue: db [B*H*W][128] = alpha m (g * W2'),
 * m = (mask_scale y1 + mask_shift > 0), g = the gradient of the layer's 32 channels ([B][H][W][g_channel_stride]: a channel slice),
 * wpk_dgrad = ossid_conv_pack_weights_dgrad(w2 [32][128][9]), and partial rows [P][2][128] (sum g' m, sum g' m y1, g' the raw data
 * gradient) for ossid_bn_fold_bwd, P = ossid_dense_dgrad3_mask_partials(B, H, W): ossid_conv_nhwc_fwd / ossid_conv3x3_wino_fwd on
 * the data-gradient weights followed by ossid_chan_op (mask_mode 1, sum_mode 1), without the second pass. */
int ossid_dense_dgrad3_mask_partials(int B, int H, int W);
int ossid_dense_dgrad3_mask(const float* g, int g_channel_stride, const float* wpk_dgrad, const float* y1, float* db, int B, int H,
                            int W, const float* alpha, const float* mask_scale, const float* mask_shift, float* partials,
                            void* stream);
int ossid_dense_dgrad1_acc_partials(long long n_rows);
int ossid_dense_dgrad1_acc(const float* dz, const float* wpk_dgrad, const float* x, float* G, long long n_rows, int c,
                           int channel_stride, const float* alpha, const float* mask_scale, const float* mask_shift, float* partials,
                           const float* dz_add, const float* dz_add_scale, const float* dz_add_shift, void* stream);

/* D4  nn.AvgPool2d(2, stride) of the DenseNet transitions (stride 2; the third one stride 1, network.py:165),
 * channels-last. backward != 0: x is d out [B][Ho][Wo][C] and out receives d in [B][H][W][C]. */
int ossid_avgpool2_nhwc(const float* x, int B, int H, int W, int C, int stride, float* out, int backward, void* stream);

/* D6  gradient of F.interpolate(mode="nearest") [Hs][Ws] -> [H][W] (network.py:354-357), channels-last: every source
 * pixel sums the destination pixels that read it, rows row_start[sy] .. row_start[sy+1]-1 (int32 [Hs+1], device),
 * columns likewise -- tables built by the caller with the forward's index formula. */
int ossid_upsample_nearest_bwd_nhwc(const float* dup, int B, int Hs, int Ws, int H, int W, int C, const int32_t* row_start,
                                    const int32_t* col_start, float* dsrc, void* stream);

/* D15  DetectionLoss.forward (models/dtoid/loss.py:46-175) and its gradient in three launches: focal classification
 * loss (alpha, gamma) with IoU anchor assignment (>= 0.5 positive, < 0.4 negative, else ignored; probabilities clamped to
 * [1e-4, 1 - 1e-4]) + smooth-L1 (beta 1/9) box regression on the positives against the encoded assigned box
 * ((dx, dy) / 0.1, (log dw, log dh) / 0.2), per image normalised by #positives (x 4), averaged over the batch.
 * cls [B][A][C] probabilities, reg [B][A][4], anchors [A][4], annotations [B][G][5] (x1,y1,x2,y2,label; label -1 = padding;
 * G <= 16). fwd writes losses2 = (cls_loss, reg_loss), the un-normalised gradients dcls_raw / dreg_raw (same shapes as
 * cls / reg) and scales2B [2][B]; bwd: dcls = grad_losses2[0] * scales[0][b] * dcls_raw, dreg likewise (grad_losses2 =
 * the upstream gradient of the two losses, device). workspace: ossid_focal_smoothl1_loss_workspace_floats(B, A) floats.
 * Sums are formed in a fixed order (per-block partials, one wave per image): bit-reproducible. */
size_t ossid_focal_smoothl1_loss_workspace_floats(int B, int A);
int ossid_focal_smoothl1_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* annotations, int B,
                                  int A, int C, int G, float alpha, float gamma, float* dcls_raw, float* dreg_raw,
                                  float* workspace, float* losses2, float* scales2B, void* stream);
int ossid_focal_smoothl1_loss_bwd(const float* dcls_raw, const float* dreg_raw, const float* scales2B, const float* grad_losses2,
                                  int B, int A, int C, float* dcls, float* dreg, void* stream);

/* D1-D4  the strided stems of the two backbones (too few input channels for the MFMA convolution's channel tiling) as
 * im2col + a 1x1 convolution on ossid_conv_nhwc_fwd: out [B][Ho][Wo][Kpad] channels-last, column (ky*k + kx)*Cin + ci =
 * normalised img[b][ci][yo*stride - pad + ky][xo*stride - pad + kx] (0 outside the image and in the columns past
 * k*k*Cin); img is NCHW as the caller holds it; mean / inv_std [Cin] (device, NULL = none) = normalizeImageRange
 * (utils/__init__.py:33-39) applied to real pixels on the way. DenseNet conv0 (network.py:164): k 7, stride 2, pad 3,
 * Kpad 160; SqueezeNet stem (:203-208): k 3, stride 2, pad 0, Kpad 48. The host re-lays the conv weight [Cout][Cin][k][k] to
 * [Cout][Kpad] in the same column order. */
int ossid_im2col_stem(const float* img_nchw, int B, int Cin, int H, int W, int k, int stride, int pad, int Kpad,
                      const float* mean, const float* inv_std, float* out, void* stream);
/* D6  the small remainders of the correlation head, deterministic (fixed-order sums):
 * ossid_conv1x1_c1_fwd: nn.Conv2d(C, 1, 1) on channels-last x [rows][C] (`corr_conv_heatmap`, network.py:334, :349):
 *   out[r] = sum_c w[c] x[r][c] + bias[0], through a sigmoid when sigmoid != 0 (heat_map = torch.sigmoid(...)); C % 4 == 0.
 * ossid_conv1x1_c1_bwd: dx[r][c] = g[r] w[c] (dx may be NULL), dw_db [C + 1] = (sum_r g[r] x[r][c], sum_r g[r]); C / 4 a power
 *   of two <= 256; workspace >= ossid_conv1x1_c1_bwd_workspace_floats(rows, C) floats.
 * ossid_spatial_mean: F.avg_pool2d(x, full window) (`avg_pool2d(template_feat, 7)`, network.py:343): x [B][C][HW] (NCHW) or
 *   [B][HW][C] (channels_last != 0) -> out [B][C]; backward != 0: x = g [B][C] -> out = g / HW broadcast in x's layout.
 * ossid_small_matmul: out [M][N] = a [M][K] b [K][N] (row-major, M <= 65535: a handful of rows against a wide matrix). */
int ossid_conv1x1_c1_fwd(const float* x, long long rows, int C, const float* w, const float* bias, int sigmoid, float* out,
                         void* stream);
size_t ossid_conv1x1_c1_bwd_workspace_floats(long long rows, int C);
int ossid_conv1x1_c1_bwd(const float* x, const float* g, long long rows, int C, const float* w, float* workspace, float* dx,
                         float* dw_db, void* stream);
int ossid_spatial_mean(const float* x, int B, int HW, int C, int channels_last, int backward, float* out, void* stream);
int ossid_small_matmul(const float* a, const float* b, int M, int K, int N, float* out, void* stream);

/* D4  a DenseNet block of ImageFeatExtract at TEST time with one launch per layer (csrc/dense.hip; network.py:164-184 builds
 * torchvision's densenet121: each _DenseLayer is norm1 -> ReLU -> conv1 1x1 (c -> 128) -> norm2 -> ReLU -> conv2 3x3
 * (128 -> 32) on the concatenation of everything before it; eval mode, BatchNorms as per-channel scale / shift).
 * The 1x1 is linear in its input channels, so the bottleneck sums y [L][pixels][128] of ALL layers are kept up to date
 * incrementally: ossid_dense_entry writes every layer's share of the block's c0 input channels, and ossid_dense_layer(l)
 * runs layer l's 3x3 on relu(norm2(y[l])), appends its 32 channels to buf [B][H][W][ctot] at channel c0 + 32 l, and adds
 * their share to y[m] for every later layer m. Same values as ossid_conv_nhwc_fwd per layer up to the order of the f32
 * summation (same three-product bf16 arithmetic, same packed weights).
 * table: device array of nlayers records {const float* w1pk; const float* s1; const float* t1; int64 units} (32 bytes each,
 * ossid_dense_table_bytes): conv1 packed by ossid_conv_pack_weights(w [128][c_m][1]), norm1 as scale / shift [c_m],
 * units = c_m / 16 with c_m = c0 + 32 m. w2pk = ossid_conv_pack_weights(conv2.weight [32][128][9]); s2 / t2 [128] = norm2.
 * c0 in {64, 128, 256, 512}; growth 32 and bottleneck 128 are fixed. ossid_dense_fused_available() = 0 in a
 * -DOSSID_CONV_F32 build (no split form): both entries then return OSSID_EINVAL and the caller keeps the per-layer path. */
int ossid_dense_fused_available(void);
size_t ossid_dense_table_bytes(int nlayers);
int ossid_dense_entry(const float* buf, int ctot, int c0, long long pixels, int nlayers, const void* table, float* y, void* stream);
int ossid_dense_layer(float* y, float* buf, int B, int H, int W, int ctot, int c0, int layer, int nlayers, const float* w2pk,
                      const float* s2, const float* t2, const void* table, void* stream);

/* D4  DenseNet-121 conv0 = nn.Conv2d(3, 64, 7, stride 2, padding 3) of ImageFeatExtract (network.py:164-170, :175-177) as an
 * IMPLICIT-im2col convolution on the f32 matrix cores (csrc/stem.hip; exact f32: an fmaf chain), and its weight gradient
 * (the image is an input: there is no data gradient). img NCHW [B][3][H][W] as the caller holds it; mean / inv_std [3]
 * (device, NULL = none) = normalizeImageRange (utils/__init__.py:33-39) applied to real pixels while staging, zero padding
 * in normalised space as the reference has it; weight / dweight [64][3][7][7] = the parameter's own layout (no packing);
 * out / dy [B][Ho][Wo][64] channels-last, Ho = (H - 1) / 2 + 1; bias [64] or NULL. Only this geometry is accepted
 * (Cin 3, Cout 64, k 7, stride 2, pad 3): anything else returns OSSID_EINVAL. The weight gradient sums per-workgroup
 * partial slabs in a fixed order (bit-reproducible); workspace >= ossid_stem_conv_wgrad_workspace_bytes(B, H, W), 16-byte
 * aligned; accumulate != 0 adds to dweight. */
int ossid_stem_conv_fwd(const float* img_nchw, int B, int Cin, int H, int W, const float* weight, int Cout, int k, int stride,
                        int pad, const float* bias, const float* mean, const float* inv_std, float* out, void* stream);
size_t ossid_stem_conv_wgrad_workspace_bytes(int B, int H, int W);
int ossid_stem_conv_wgrad(const float* img_nchw, const float* dy, int B, int Cin, int H, int W, int Cout, int k, int stride,
                          int pad, const float* mean, const float* inv_std, void* workspace, size_t workspace_bytes,
                          float* dweight, int accumulate, void* stream);
/* D4  relu(scale[c] * (x0 + conv2d_dw_group(x0, kernels)) + shift[c]) (network.py:178-181: the global-template modulation
 * of the stem output, norm0 in eval mode, relu0), channels-last; kernels [B or 1][C][3][3], kernels_batch_stride = C*9 or
 * 0 (one kernel set for the whole batch). */
int ossid_stem_tail_nhwc(const float* x0, const float* kernels, int kernels_batch_stride, const float* scale, const float* shift,
                         int B, int H, int W, int C, float* out, void* stream);
/* D4 at test time, fused further: ossid_stem_tail_pool_nhwc = ossid_stem_tail_nhwc followed by pool0 = nn.MaxPool2d(3, 2, 1)
 * (network.py:170-179) in one pass, written into the first C channels of a buffer with out_channel_stride floats per pixel (the
 * dense block's resident buffer): out [B][Ho][Wo][out_channel_stride], Ho = (H - 1) / 2 + 1.
 * ossid_bn_relu_avgpool2_nhwc: relu(scale * x + shift) averaged over 2 x 2 windows (stride 1 or 2) -- the front of a DenseNet
 * transition (norm -> relu -> conv 1x1 -> AvgPool2d) with the pool moved in front of the bias-free 1x1 convolution, with which
 * it commutes: x [B][H][W][in_channel_stride] (first C channels) -> out [B][Ho][Wo][C]. */
int ossid_stem_tail_pool_nhwc(const float* x0, const float* kernels, int kernels_batch_stride, const float* scale, const float* shift,
                              int B, int H, int W, int C, float* out, int out_channel_stride, void* stream);
int ossid_bn_relu_avgpool2_nhwc(const float* x, int B, int H, int W, int C, int in_channel_stride, const float* scale,
                                const float* shift, int stride, float* out, void* stream);
/* D2-D4  nn.MaxPool2d(k, stride, padding, ceil_mode) channels-last (DenseNet pool0: 3, 2, 1; SqueezeNet: 3, 2, 0, ceil). */
int ossid_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, int ceil_mode, float* out,
                       void* stream);

/* Separable linear resampling of a channels-last image by tap tables (training path of the template encoders,
 * models/dtoid/network.py:223-239, :265-279):
 *   out[b][oy][ox][out_channel_offset + c] = sum_{i,j < T} wy[oy][i] wx[ox][j] x[b][iy[oy][i]][ix[ox][j]][c]
 * taps_*_idx [n_out][T] int32 (-1 = unused tap), taps_*_w [n_out][T] float32, T <= 8; channel strides in floats (0 = C).
 * Covers F.interpolate(mode="bilinear", align_corners=False) (T = 2), its adjoint (transposed tables), the crop that
 * turns a padded 3x3 convolution into the reference's valid one (T = 1) and the crop's adjoint (zero padding). */
int ossid_resample_taps_nhwc(const float* x, int B, int Hin, int Win, int C, int x_channel_stride, int Hout, int Wout,
                             const int32_t* taps_y_idx, const float* taps_y_w, const int32_t* taps_x_idx, const float* taps_x_w,
                             int T, float* out, int out_channel_stride, int out_channel_offset, void* stream);

/* D16  training-side companions of the stem kernels (finetune step, channels-last):
 * ossid_dw_add_nhwc: out = x + conv2d_dw_group(x, kernels) (network.py:178-179), flip != 0: the taps rotated by 180 degrees
 *   (= the gradient with respect to x of the same expression applied to the upstream gradient);
 * ossid_dw_bwd_k_nhwc: dk[b][c][ky][kx] = sum_px g[b][px][c] * x[b][px + tap][c] (gradient with respect to the per-sample
 *   kernels), per-row-chunk partials in `workspace` (ossid_dw_bwd_k_workspace_floats), summed in a fixed order;
 * ossid_maxpool_idx_nhwc / ossid_maxpool_bwd_nhwc: nn.MaxPool2d forward that also stores the window position of the
 *   maximum (uint8, first maximum in (ky, kx) order, as torch) and the backward as a gather over the windows containing
 *   each input pixel -- no atomics. */
int ossid_dw_add_nhwc(const float* x, const float* kernels, int kernels_batch_stride, int B, int H, int W, int C, int flip,
                      float* out, void* stream);
/* ossid_dw_add_nhwc that also leaves the column sums of its OUTPUT y about pivot[c] = y[0][0][0][c] -- (sum (y - pivot),
 * sum (y - pivot)^2), the batch statistics of the BatchNorm that follows (norm0) -- as ossid_dw_add_stats_partials(B, H, W, C)
 * partial rows [P][2][C] for ossid_bn_fold_fwd (n_partials = P, pivot = pivot_out [C]): the statistics cost no pass of
 * their own. C / 4 must be a power of two <= 64 (these four entry points). partials / pivot_out both NULL: no statistics. */
int ossid_dw_add_stats_partials(int B, int H, int W, int C);
int ossid_dw_add_stats_nhwc(const float* x, const float* kernels, int kernels_batch_stride, int B, int H, int W, int C, int flip,
                            float* out, float* partials, float* pivot_out, void* stream);
/* D4  norm0 + relu0 + pool0 of the training stem without ever writing the normalised tensor (network.py:180-181 with
 * torchvision's MaxPool2d(3, 2, 1)):  out [B][Ho][Wo][C] = maxpool(relu(scale[c] * m + shift[c])), argmax = the window
 * position (ky * 3 + kx, uint8) of the first maximum, Ho = (H - 1) / 2 + 1.
 * Backward in two passes over m, the un-pooled gradient gm = [scale m + shift > 0] * (sum of dpooled whose maximum sat at
 * this pixel) formed on the fly both times: dm == NULL: (sum gm, sum gm * m) -> ossid_stem_pool_bwd_partials(B, H, W, C)
 * partial rows for ossid_bn_fold_bwd; partials == NULL: dm = scale gm + coef_x m + coef_1 (coefficients from that fold). */
int ossid_stem_pool_fwd(const float* m, const float* scale, const float* shift, int B, int H, int W, int C, float* out,
                        uint8_t* argmax, void* stream);
int ossid_stem_pool_bwd_partials(int B, int H, int W, int C);
int ossid_stem_pool_bwd(const float* m, const uint8_t* argmax, const float* dpooled, const float* scale, const float* shift,
                        const float* coef_x, const float* coef_1, int B, int H, int W, int C, float* partials, float* dm,
                        void* stream);
size_t ossid_dw_bwd_k_workspace_floats(int B, int H, int W, int C);
int ossid_dw_bwd_k_nhwc(const float* x, const float* g, int B, int H, int W, int C, float* workspace, float* dk, void* stream);
int ossid_maxpool_idx_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, int ceil_mode, float* out,
                           uint8_t* argmax, void* stream);
int ossid_maxpool_bwd_nhwc(const float* dout, const uint8_t* argmax, int B, int H, int W, int C, int k, int stride, int pad, int Ho,
                           int Wo, float* dx, void* stream);

/* D12  torch.topk(scores, k) (network.py:555: the 1000 best of the ~570 k (template, anchor) object scores of a frame):
 * values [k] in decreasing order and their int64 indices; equal scores are ordered (and, at the cut, chosen) by increasing
 * index. Radix select (three histogram passes) + ordered gather + one-workgroup sort of the k survivors; k <= 2048.
 * NaN scores order above +inf. workspace: ossid_topk_workspace_bytes(n, k) bytes. */
size_t ossid_topk_workspace_bytes(int n, int k);
int ossid_topk(const float* scores, int n, int k, void* workspace, size_t workspace_bytes, float* values, long long* indices,
               void* stream);

/* D12  torchvision.ops.nms(boxes, scores, iou_threshold)      network.py:563, models/dtoid/utils.py:33
 * boxes [n][4] (x1,y1,x2,y2) ALREADY sorted by descending score (network.py:555 feeds it the top-k order);
 * keep [n] receives the indices of the survivors in that order, *num_keep their count. n <= 16384. */
size_t ossid_nms_workspace_bytes(int n);
int ossid_nms(const float* boxes, int n, float iou_threshold, void* workspace, size_t workspace_bytes,
              int32_t* keep, int32_t* num_keep, void* stream);

/* D12  the post-processing of a frame as launches only (network.py:543-566: decode + clip, top-k of the object scores over all
 * templates, NMS), so that it can live in the frame's captured graph: scores = the object column of the class probabilities
 * (element i at scores[i * score_stride], n = templates * A of them, i = template * A + anchor), anchors [A][4], deltas [n][4].
 * Out: the k best scores in decreasing order (ties by increasing index, as ossid_topk), their indices, their decoded and
 * clipped boxes (the arithmetic of ossid_decode_clip_boxes, applied to the k survivors only), and ossid_nms's keep list over
 * them with its count. k <= 2048, n % A == 0. Nothing is read back by the host.
 * ossid_detect_emit then writes the detection list for the first `count` kept candidates (network.py:566-581; the host knows
 * count = min(*num_keep, topk) by then): row j = candidate keep[j]: score, box, index of the template that fired (as float,
 * network.py:561), that template's row of seg [templates][seg_row_floats] -- through a sigmoid when seg_sigmoid != 0
 * (models/dtoid/__init__.py:147) -- and of heat [templates][heat_row_floats]. */
size_t ossid_detect_post_workspace_bytes(int n, int k);
int ossid_detect_post(const float* scores, int n, int score_stride, int k, const float* anchors, const float* deltas, int A,
                      float img_w, float img_h, float iou_threshold, void* workspace, size_t workspace_bytes, float* out_scores,
                      long long* out_indices, float* out_boxes, int32_t* keep, int32_t* num_keep, void* stream);
int ossid_detect_emit(const float* scores, const long long* indices, const float* boxes, const int32_t* keep, int count, int A,
                      const float* seg, long long seg_row_floats, const float* heat, long long heat_row_floats, int seg_sigmoid,
                      float* out_scores, float* out_boxes, float* out_obj, float* out_seg, float* out_heat, void* stream);

/* D6  `norm(F.elu(conv(image_feat - avg_t)))` (network.py:346) for ONE image against all templates without a convolution
 * per template: conv is linear, so conv(x - a_t) = conv(x) - conv(a_t). S [H][W][channels] = conv(x) + bias, computed
 * once per frame; csub [templates][9][channels] = the response to the per-channel constant image a_t for each of the 9
 * border patterns p = 3*rowclass + colclass (class 0: first row/column, 1: interior, 2: last), one small GEMM per object.
 * out[t][y][x][out_channel_offset + o] = post(ELU(S[y][x][o] - csub[t][p(y,x)][o])). H, W >= 2; channels % 4 == 0. */
int ossid_bcast_sub_epilogue(const float* S, const float* csub, int templates, int H, int W, int channels,
                             const float* post_scale, const float* post_shift, float* out, int out_channel_stride,
                             int out_channel_offset, void* stream);

/* D6  `conv(image_feat * avg_t)` (network.py:345) with the channel contraction last, for MANY templates against one image:
 * ossid_dot_expand builds G[c][y][x][o] = sum_taps w[o][c][tap] * x[y+dy][x+dx][c] (zero padding) from the channels-last
 * image x [H][W][channels] and the weights re-laid as w_cto [channels][9][cout]; the caller then contracts
 * z[t][(y,x,o)] = sum_c avg_t[c] * G[c][(y,x,o)] with a library GEMM and finishes with ossid_bias_elu_affine_slice:
 * out[r][out_channel_offset + o] = post(ELU(z[r][o] + bias[o])), rows r = (template, pixel). cout % 4 == 0,
 * 256 % (cout/4) == 0. G is channels * H * W * cout floats (0.74 GB for 640 x 29 x 39 x 256): caller-owned. */
int ossid_dot_expand(const float* x, const float* w_cto, int channels, int cout, int H, int W, float* G, void* stream);
int ossid_bias_elu_affine_slice(const float* z, long long rows, int channels, const float* bias, const float* post_scale,
                                const float* post_shift, float* out, int out_channel_stride, int out_channel_offset,
                                void* stream);

/* D12/D13  out[j][:] = src[idx[j]][:] for j < k, rows of row_floats (% 4 == 0) floats, with sigmoid applied on the way
 * when apply_sigmoid != 0: the per-detection segmentation maps gathered from the per-template ones
 * (network.py:575-579) + `torch.sigmoid(seg)` (models/dtoid/__init__.py:147) in one pass. idx: int64, each in
 * [0, n_src_rows) (caller-checked). */
int ossid_gather_rows(const float* src, int n_src_rows, long long row_floats, const long long* idx, int k,
                      int apply_sigmoid, float* out, void* stream);

/* D10  BBoxTransform.forward + ClipBoxes.forward      network.py:42-70, :78-88
 * anchors [A][4] (shared by all rows), deltas [rows][A][4] -> boxes [rows][A][4]; std (.1,.1,.2,.2), mean 0. */
int ossid_decode_clip_boxes(const float* anchors, const float* deltas, int rows, int A, float img_w, float img_h,
                            float* boxes, void* stream);

/* D16  torch.optim.Adam(lr, weight_decay, amsgrad=True).step()      scripts/online_learning.py:258-263, :672
 * one launch over flat float buffers of n elements (n % 4 == 0); `step` is the 1-based update count. */
int ossid_amsgrad_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                       size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       void* stream);

/* =============================================================================================
 * The steps either side of the hot path, one frame at a time, all on the device
 * ============================================================================================= */

/* 8f-1  DTOID batch producer: processData (utils/data.py:7-83: depth2xyz at the original resolution, bilinear resize of
 * image / mask / xyz to [H][W], image /255, CHW), datasets/dtoid_bop_dataset.py:256-290.
 * img u8 [Ho][Wo][3], depth f32 [Ho][Wo] (m), mask f32 [Ho][Wo] in [0,1] -> img_out [3][H][W], xyz_out [3][H][W], mask_out [H][W].
 * Equal sizes (480x640 LM-O / YCB-V frames) are an exact copy. */
int ossid_dtoid_prep_sample(const uint8_t* img, const float* depth, const float* mask, int Ho, int Wo, float fx, float fy,
                            float cx, float cy, int H, int W, float* img_out, float* xyz_out, float* mask_out, void* stream);
/* mask -> bbox_gt (x1,y1,x2,y2,label) = min/max of the non-zero pixels (dtoid_bop_dataset.py:274-279; label -1 for an
 * empty mask) and the Gaussian heat map around its centre, float64 (utils/__init__.py:354-367; :283-287). */
int ossid_mask_bbox_heatmap(const float* mask, int H, int W, int heat_h, int heat_w, double heat_scale, double sigma,
                            int32_t* bbox5, double* heatmap, void* stream);

/* 8f-3  post-score step (scripts/online_learning.py:485-500, :557-558): a depth-only point-splat renderer in place of
 * pyrender (z-buffer of the posed model cloud, (2*radius+1)^2 pixel splats; zbuf_workspace = H*W*4 bytes), the bop19
 * visibility mask visib = (d_pred - d_obs <= delta or d_obs == 0) and d_pred > 0, and the set sizes of both IoUs:
 * counts4 = |pred&gt|, |pred|gt|, |visib&gt_visib|, |visib|gt_visib| (gt masks u8, may be NULL). */
int ossid_render_depth_points(const float* transform, const float* points, int M, float fx, float fy, float cx, float cy,
                              int H, int W, int radius, void* zbuf_workspace, float* depth_out, void* stream);
int ossid_visib_mask_iou(const float* depth_obs, const float* depth_pred, const uint8_t* gt_mask, const uint8_t* gt_mask_visib,
                         int H, int W, float delta, uint8_t* pred_mask, uint8_t* pred_mask_visib, int32_t* counts4,
                         void* stream);

#ifdef __cplusplus
}
#endif
#endif /* OSSID_HIP_H */
