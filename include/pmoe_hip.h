/* libpmoe_hip.so -- C ABI of the MI355X (gfx950) kernels behind PMoE's stage-2 policy networks.
 *
 * The reference (mhnazeri/PMoE) has no native layer at all: its hot path reaches cuDNN/cuBLAS/ATen
 * through torch.nn.  Each entry point below therefore names the torch.nn / torch.nn.functional call
 * site in the reference whose arithmetic it replaces (paths relative to /root/reference/PMoE).
 * `pmoe_amd/hip.py` is the ctypes binding the Python host uses; INTEGRATION.md shows how a
 * maintainer of the reference would bind it.
 *
 * Conventions
 *   - plain C: raw DEVICE pointers, ints, floats; no torch / STL types; nothing is allocated or
 *     freed here (the caller owns every buffer, scratch included); launches go to `stream`
 *     (a hipStream_t passed as void*), never the null stream implicitly, and never synchronise.
 *   - return value: 0 = launched; >0 = hipError_t; PMOE_ERR_* (<0) = rejected arguments.
 *   - activations are NHWC ("channels last"), element type `dtype` (PMOE_DT_BF16 | PMOE_DT_F32),
 *     with the E experts folded into the image index: image n belongs to expert n / ipe.
 *     Every channel count is padded by the caller to a multiple of 16 (zero filled).
 *   - re-entrant, no global mutable state besides one-time kernel attribute setup.
 */
#ifndef PMOE_HIP_H
#define PMOE_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define PMOE_DT_BF16 0
#define PMOE_DT_F32 1

#define PMOE_ERR_ARG (-1)
#define PMOE_ERR_UNSUPPORTED (-2)

#define PMOE_ACT_NONE 0
#define PMOE_ACT_RELU 1
#define PMOE_ACT_ELU 2
#define PMOE_ACT_TANH 3    /* make_mlp `act` choices of model/blocks/basics.py:23-28 */
#define PMOE_ACT_SIGMOID 4

#define PMOE_RES_NONE 0
#define PMOE_RES_ADD 1     /* out = acc + res                                   (residual / grad sum) */
#define PMOE_RES_DRELU 2   /* out = acc * relu'(res), res = saved layer output  (MLP backward)        */
#define PMOE_RES_DELU 3    /* out = acc * elu'(res),  res = saved layer output                        */
#define PMOE_RES_DTANH 4   /* out = acc * (1 - res^2)                                                  */
#define PMOE_RES_DSIGMOID 5 /* out = acc * res (1 - res)                                               */
/* round 3: data gradient of a conv whose input was a = relu(BatchNorm(z)) (train mode, no residual) -- `res` = z, the
 * BatchNorm's pre-activation input; `bn_coef` = its [4][n/bn_ipe][cout] f32 coefficients (mean, invstd, gamma*invstd,
 * beta).  out = g = acc * [ (z - mean) * gamma*invstd + beta > 0 ], and `stats` receives the two reductions of that
 * BatchNorm's backward -- sum g and sum g * (z - mean) * invstd per channel, in the row layout of the forward statistics --
 * so the separate reduce pass over (dy, z) disappears (autograd of nn.BatchNorm2d + nn.ReLU in torchvision's BasicBlock
 * and model/blocks/basics.py:93-100).  bf16, 3x3 stride-1 layers on the LDS-DMA kernels only (pmoe_conv2d_plan says). */
#define PMOE_RES_DBN 6
/* round 4 (BASELINE config 4, the frozen U-Nets' conv -> BatchNorm -> ReLU -> conv pairs of model/blocks/unet.py:14-24 in train
 * mode): no side input in the epilogue (`res` must be NULL) -- the convolution's INPUT is the pre-activation z of a BatchNorm +
 * ReLU, `bn_coef` = that BatchNorm's [4][n/bn_ipe][cin] f32 coefficients, and the value the matrix cores see is
 * bf16(max((in - mean) * gamma*invstd + beta, 0)): exactly what pmoe_bn_apply(relu = 1) would have written, evaluated in LDS on the
 * halo patch, so that pass and the activation tensor between the two convolutions do not exist.  Forward launches without
 * activation / dropout; bf16; the 64 -> 64-channel 3x3 stride-1 resident-filter kernel (no bias; pmoe_conv2d_plan returns 1267) and
 * the 1x1 direct kernel (bias allowed, with or without shuffle_c: 1412 | 1414 | 1462 | 1464 -- the U-Nets' ConvTranspose2d layers and
 * final classifier behind a frozen block); anything else PMOE_ERR_UNSUPPORTED: the caller then runs pmoe_bn_apply + a plain launch. */
#define PMOE_RES_INBN 7

/* ABI revision of this header: bumped whenever a descriptor struct, an argument list or a buffer contract changes
 * (100: round 1; 200: round 2 -- pmoe_conv_desc 160 -> 176 bytes, pmoe_wgrad_desc.part_ws, pmoe_bn_bwd_reduce's gmask_out,
 * dw_ws overwritten instead of accumulated; 300: round 3; 400: round 4 -- pmoe_wgrad_desc.bn_*; 401: PMOE_RES_INBN).  A binding compares pmoe_version() with the value it was
 * written against before its first launch (pmoe_amd/hip.py:load does). */
#define PMOE_ABI_VERSION 401
int pmoe_version(void);
const char* pmoe_error_string(int code);
/* sizeof() of the descriptor structs as compiled (which: 0 = pmoe_conv_desc, 1 = pmoe_wgrad_desc);
 * lets a foreign-language binding verify its struct layout without launching anything */
int pmoe_abi_sizeof(int which); /* 0: pmoe_conv_desc, 1: pmoe_wgrad_desc, 2: pmoe_opt_tensor */

/* ---- convolution / grouped GEMM -------------------------------------------------------------
 * Replaces nn.Conv2d in model/blocks/basics.py:93-100,113-120 (stem), the torchvision ResNet body
 * built at model/blocks/backbone.py:57-70, and nn.Linear in model/blocks/basics.py:31 /
 * model/moe.py:71-72 (a Linear over the per-expert batch is a 1x1 conv on 1x1 images).
 * Also their data gradients: stride-1 dgrad = the same conv with flipped/transposed weights,
 * stride-2 dgrad = `dilate` (transposed conv: the source is read as if zero-upsampled by 2). */
typedef struct pmoe_conv_desc {
    const void* in;       /* [Nin][H][W][in_ld]; reduces over channels [in_coff, in_coff+cin)      */
    const void* w;        /* packed [E][coutp][ks*ks][cin] (pmoe_pack_conv_weights)                */
    void* out;            /* [N][Ho][Wo][out_ld]; writes channels [out_coff, out_coff+cout)        */
    const void* res;      /* optional, geometry of out with res_ld/res_coff (see PMOE_RES_*)       */
    const float* bias;    /* optional [E][coutp] f32                                               */
    float* stats;         /* optional [pmoe_conv2d_stat_rows()][2][coutp] f32 BN partial sums      */
    int32_t n, h, w_, cin;
    int32_t ho, wo, cout, coutp;
    int32_t in_ld, in_coff, out_ld, out_coff, res_ld, res_coff;
    int32_t ipe;          /* images per expert                                                     */
    int32_t in_shared;    /* 1: `in` holds ipe images read by every expert                         */
    int32_t ks, stride, pad, dilate;
    int32_t act, res_mode;
    float drop_p;         /* >0: inverted dropout on the output (nn.Dropout, basics.py:39)         */
    uint64_t seed;
    int32_t dtype;
    /* BASELINE config 5 (e4m3 weights on the fp8 matrix cores; forward convolutions, dtype BF16, cin % 64 == 0):
     * w_fp8 = 1: `w` holds OCP e4m3 bytes [E][coutp][ks*ks][cin] from pmoe_pack_conv_weights_fp8; the bf16 activations
     * are converted to e4m3(x * in_scale) on their way into LDS, the accumulators are multiplied by out_scale[e][cout]
     * (= weight scale / in_scale) before bias / residual / activation. */
    int32_t w_fp8;
    float in_scale;
    const float* out_scale;
    int32_t in_fp8;       /* round 3 (with w_fp8): `in` holds e4m3 BYTES too -- e4m3(x * in_scale), written by pmoe_bn_apply's fp8
                           * side output; in_ld / in_coff count bytes.  Dense 3x3 stride 1, cin % 128 == 0: the block-scaled
                           * matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 (plan code 8507), twice the bf16 rate */
    const float* bn_coef; /* PMOE_RES_DBN: [4][n / bn_ipe][cout] f32 = mean, invstd, gamma*invstd, beta of the BatchNorm  */
    int32_t bn_ipe;       /* PMOE_RES_DBN: images per BatchNorm parameter set (= ipe unless the conv runs per image)     */
    int32_t shuffle_c;    /* round 4, > 0: nn.ConvTranspose2d(k=2, s=2) of model/blocks/unet.py:28-45 in ONE launch -- the layer is a 1x1
                           * convolution into 4*shuffle_c channels (row (2 dy + dx)*shuffle_c + c of the packed weights) whose
                           * output pixel (oy, ox) is scattered to (2 oy + dy, 2 ox + dx): `out` is then the DESTINATION
                           * [n][2 ho][2 wo][out_ld] and receives channels [out_coff, out_coff + shuffle_c) (the "up" half of the
                           * skip-concatenation buffer); cout = 4*shuffle_c, ho / wo powers of two.  conv1x1_direct_kernel only
                           * (pmoe_conv2d_plan returns 1452 | 1454, anything else PMOE_ERR_UNSUPPORTED); replaces the
                           * pmoe_pixel_shuffle2 launch and the [n][ho][wo][4c] intermediate. */
} pmoe_conv_desc;

int pmoe_conv2d_igemm(const pmoe_conv_desc* d, void* stream);
/* number of [2][coutp] partial-sum rows the launch writes to d->stats (rows of expert e are
 * contiguous: [e*rows/E, (e+1)*rows/E) ); <0 = error */
int pmoe_conv2d_stat_rows(const pmoe_conv_desc* d);
/* which kernel instantiation pmoe_conv2d_igemm would run for this descriptor (nothing is launched; used by bench.py
 * to attribute measured launch times to kernel symbols that a rocprofv3 kernel trace shows):
 *   3000                     gemm_skinny_kernel<4|8>               (expert MLP layers / <= 2048 output pixels per expert, bf16)
 *   1000 + LOG_RB            conv3x3_res_kernel<LOG_RB>            (resident-filter kernel, conv_res.hip)
 *   1400 + MT                conv1x1_direct_kernel<MT>             (1x1, stride 1 | 2, >= 8192 pixels per expert, 64..512 input channels, conv_c1x1.hip)
 *   1410 + MT | 1460 + MT    conv1x1_direct_kernel<MT, true>       (the same with PMOE_RES_INBN: BatchNorm + ReLU of the input applied to the operand registers; 1450 + MT / 1460 + MT: with shuffle_c)
 *   1316                     conv3x3_c16_kernel                    (16 input channels: direct MFMA form, no LDS staging, conv_c16.hip)
 *   1207 + 10 b + 20 m       conv3x3_respipe_kernel<b, m>          (resident filter bank, halo patches by LDS-DMA, read-out of tile t in registers under the MFMAs
 *                                                                   of tile t+1, conv_res.hip; b: per-expert bias row, m: 0 plain | 1 PMOE_RES_ADD | 2 PMOE_RES_DBN | 3 PMOE_RES_INBN)
 *   1107 | 1117              conv3x3_resdma_kernel<false | true>   (its LDS-staged predecessor, PMOE_RES_PIPE=0)
 *   2000 + LOG_RB            conv_igemm_lite_kernel<T, LOG_RB>     (8-wave 256 x 128 tile, two workgroups per CU)
 *   5007 | 5017              conv3x3_dma_kernel<false | true>      (LDS-DMA staged 3x3 stride-1 kernel, >= 128 channels, conv_dma.hip; <true>: 16x16x32 MFMA shape, >= 256 input channels)
 *   5027 | 5037              conv3x3_dma_kernel<false | true, true> (round 4: the same with a ninth, request-only producer wave; launches without a side input / bias / activation)
 *   5047 | 5057              conv3x3_dma_stream_kernel<false | true> (round 4: those launches on persistent workgroups whose request stream runs across tile boundaries)
 *   5067                     conv3x3_dma_stream_kernel<false, true>  (round 4: its 256-pixel x 64-output-channel tiles for 64 output channels over >= 128 input channels)
 *   5207                     conv3x3s2_dma_kernel                  (its stride-2 forward sibling: parity planes gathered by the DMA, conv_dma.hip)
 *   8000 + one of the above  the same tile with e4m3 operands (w_fp8)
 *   8507                     conv3x3_dma_f8_kernel                 (w_fp8 + in_fp8: LDS-DMA kernel on the block-scaled fp8 MFMA, conv_dma.hip)
 *   LOG_RB*100 + WM*10 + WN  conv_igemm_kernel<T, LOG_RB, WM, WN>  (halo-patch implicit GEMM, conv_igemm.hip)
 *   4000 + the latter        the four parity-class launches of a stride-2 3x3 data gradient */
int pmoe_conv2d_plan(const pmoe_conv_desc* d);

/* Weight gradient of the same layers (autograd of nn.Conv2d / nn.Linear at the call sites above).
 * WRITES dw_ws [E][ks*ks][coutp][cinp] f32 (every element stored once; coutp / cinp = cout / cin rounded up to the
 * kernel's channel tile: 64 for bf16, 32 for f32).  Deterministic: the pixel (GEMM-K) split over workgroups goes through
 * per-workgroup slabs in `part_ws` that a second launch folds in fixed order -- no float atomics, bit-reproducible. */
typedef struct pmoe_wgrad_desc {
    const void* x;        /* layer input  [Nin][H][W][x_ld]  */
    const void* dy;       /* output grad  [N][Ho][Wo][dy_ld] */
    float* dw_ws;
    int32_t n, h, w_, cin, cinp;
    int32_t ho, wo, cout, coutp;
    int32_t x_ld, x_coff, dy_ld, dy_coff;
    int32_t ipe, x_shared;
    int32_t ks, stride, pad;
    int32_t dtype;
    int32_t per_image;    /* 1: dw_ws is [N][ks*ks][coutp][cinp], one slab per image (used to derive the ECA
                           * gate gradient of the stem from per-image filter gradients); needs ho*wo >= 256 */
    float* part_ws;       /* K-split scratch, pmoe_conv2d_wgrad_ws_floats() floats (may be null when that is 0) */
    int64_t part_ws_floats;
    float* grads;         /* optional (round 3): the parameters' own gradient [E][cout_real][cin_real][ks][ks] f32 (what
                           * autograd leaves in nn.Conv2d.weight.grad / nn.Linear.weight.grad of the E experts, contiguous);
                           * when set, the K-split fold writes it directly and dw_ws is only scratch: no
                           * pmoe_unpack_conv_wgrad launch afterwards.  Not with per_image. */
    int32_t cout_real, cin_real;
    int32_t defer_fold;   /* 1: pmoe_conv2d_wgrad runs the MFMA launch only; the caller runs pmoe_conv2d_wgrad_fold(d) afterwards
                           * (so that a profiler / per-launch events see the two kernels apart) */
    /* round 4: BatchNorm backward applied ON LOAD (the stem's conv1 -> BatchNorm2d -> ReLU, model/blocks/basics.py:113-120,
     * whose output gradient has this filter gradient as its only consumer).  bn_fused = 1: `dy` holds g, the ReLU-masked
     * gradient w.r.t. the BatchNorm OUTPUT (what PMOE_RES_DBN leaves), `bn_z` the BatchNorm's input (the conv's own output,
     * geometry of dy with row length bn_z_ld), `bn_coef` [4][n/ipe][cout] f32 = mean, invstd, gamma*invstd, beta and
     * `bn_c1` / `bn_c2` [n/ipe][cout] the two means pmoe_bn_bwd_finalize leaves; the kernel evaluates
     * dz = g*A + ((z - mean)*Bx + K) exactly like pmoe_bn_bwd_apply (same arithmetic, same bf16 rounding) between its
     * loads and its LDS tile, so the gradient tensor dz is never written or read.  bf16, per_image, 3x3 stride 1 pad 1,
     * cin <= 16, cout <= 64 (plan code 7209); anything else: PMOE_ERR_UNSUPPORTED. */
    int32_t bn_fused;
    const void* bn_z;
    const float* bn_coef;
    const float* bn_c1;
    const float* bn_c2;
    int32_t bn_z_ld;
} pmoe_wgrad_desc;
int pmoe_conv2d_wgrad(const pmoe_wgrad_desc* d, void* stream);
/* the tail of pmoe_conv2d_wgrad for the same descriptor: K-split slabs -> dw_ws, or -> `grads` when set */
int pmoe_conv2d_wgrad_fold(const pmoe_wgrad_desc* d, void* stream);
/* floats of part_ws the launch needs for this descriptor (0: single K slice or per_image); <0 = error.  Pointers in the
 * descriptor are not read. */
int64_t pmoe_conv2d_wgrad_ws_floats(const pmoe_wgrad_desc* d);
/* which kernel serves the descriptor (nothing is launched; bench.py attributes launch times to rocprof symbols with it):
 *   7009 = conv_wgrad_dma_kernel (LDS-DMA staged, dense 3x3 stride 1, bf16; 7109 = its wave layout for <= 32 input channels);  6000 + taps * 100 + MAXV = conv_wgrad_kernel<T, taps, MAXV>;
 *   7209 = conv_wgrad_bnbwd_kernel (bn_fused);  7309 = conv_wgrad_dma2_kernel (round 4: the 7009 launches with one accumulating wave per SIMD and a request-only producer wave) */
int pmoe_conv2d_wgrad_plan(const pmoe_wgrad_desc* d);

/* round 4: weight and bias gradient of a Linear layer of the expert MLPs (autograd of nn.Linear in make_mlp,
 * model/blocks/basics.py:31, and of the heads model/moe.py:70-72) in ONE launch, written in the parameters' own layout:
 *   grads[e][o][i] = sum over the expert's ipe batch rows m of dy[e*ipe + m][dy_coff + o] * x[(x_shared ? m : e*ipe + m)][x_coff + i]
 *   bias_grads[e][o] = sum_m dy[e*ipe + m][dy_coff + o]          (optional)
 * x [Nx][x_ld], dy [n][dy_ld]: bf16 rows (1x1 "images" of the grouped engine); cin / cout: staged channel counts (multiples
 * of 8, zero-padded columns), cin_real / cout_real: the Linear's in_features / out_features.  Deterministic (fixed order). */
int pmoe_mlp_wgrad(const void* x, const void* dy, float* grads, float* bias_grads, int32_t n, int32_t ipe, int32_t x_shared,
                   int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real, int32_t x_ld, int32_t x_coff,
                   int32_t dy_ld, int32_t dy_coff, int32_t dtype, void* stream);

/* Master weights live in the reference's own layout (one f32 OIHW / [out][in] tensor per expert,
 * state_dict keys of SURVEY.md section 8b); these repack all E experts of a layer in one launch.
 * src_ptrs: device array of E pointers to f32 [cout][cin][ks][ks].
 * fwd : [E][coutp][ks*ks][cinp]                       (conv / linear forward, wgrad layout)
 * dgrd: [E][cinp2][ks*ks][coutp2] with taps reversed   (operand of the data-gradient launch)
 * either destination may be null. */
int pmoe_pack_conv_weights(const void* const* src_ptrs, void* fwd, void* dgrd, int32_t E, int32_t cout,
                           int32_t cin, int32_t ks, int32_t coutp, int32_t cinp, int32_t cinp2, int32_t coutp2,
                           int32_t dtype, void* stream);
/* BASELINE config 5: the same master weights quantised to OCP e4m3 with ONE POWER-OF-TWO scale per output channel,
 * s = 2^ceil(log2(max|W[co]| / 448)), q = e4m3(W / s) round-to-nearest-even (the CPU statement of this policy is
 * oracle/fp8_policy.py).  fwd_e4m3 [E][coutp][ks*ks][cinp] bytes; dgrd_bf16 [E][cinp2][ks*ks flipped][coutp2] = q * s
 * (exact in bf16: the data gradient sees exactly the weights the forward used -- straight-through estimator);
 * wscale / oscale [E][coutp] f32 = s and s / in_scale (pmoe_conv_desc.out_scale).  Replaces the weight operand of the
 * nn.Conv2d call sites model/blocks/basics.py:93-100,113-120 and model/blocks/backbone.py:57-70. */
int pmoe_pack_conv_weights_fp8(const void* const* src_ptrs, void* fwd_e4m3, void* dgrd_bf16, float* wscale, float* oscale,
                               float in_scale, int32_t E, int32_t cout, int32_t cin, int32_t ks, int32_t coutp, int32_t cinp,
                               int32_t cinp2, int32_t coutp2, void* stream);
/* Inference (callers autoagents/image_agent.py:127-177, the validation loop train_2.py:245-275):
 * eval-mode BatchNorm folded into the preceding convolution, bn(conv(x, W)) = conv(x, W * s) + (beta - mean * s) with
 * s = gamma / sqrt(running_var + eps).  scale / shift / mean: [E][cout] f32 as produced by pmoe_bn_finalize(training=0);
 * fwd [E][coutp][ks*ks][cinp], bias [E][coutp] f32 (the conv then runs with bias + ReLU / residual in its epilogue). */
int pmoe_pack_conv_weights_scaled(const void* const* src_ptrs, const float* scale, const float* shift, const float* mean,
                                  void* fwd, float* bias, int32_t E, int32_t cout, int32_t cin, int32_t ks, int32_t coutp,
                                  int32_t cinp, int32_t dtype, void* stream);
/* Per-IMAGE weight packs with the ECA gate folded in (basics.py:69-76 in front of basics.py:113):
 * conv(x * g[n,c], W) == conv(x, W * g[n,c]); the convolution then runs with ipe = 1 (one "expert" per image) on
 * fwd [N][coutp][ks*ks][cinp] / dgrd [N][cinp2][ks*ks][coutp2] and the gated activation is never written.
 * gate [N][gate_ld] f32; src_ptrs: the E = N / ipe per-expert OIHW f32 parameters. */
int pmoe_pack_conv_weights_gated(const void* const* src_ptrs, const float* gate, int32_t gate_ld, void* fwd, void* dgrd,
                                 int32_t N, int32_t ipe, int32_t cout, int32_t cin, int32_t ks, int32_t coutp,
                                 int32_t cinp, int32_t cinp2, int32_t coutp2, int32_t dtype, void* stream);
/* dw_ws [E][ks*ks][coutp][cinp] f32 -> grads [E][cout][cin][ks][ks] f32 (contiguous arena slice) */
int pmoe_unpack_conv_wgrad(const float* dw_ws, float* grads, int32_t E, int32_t cout, int32_t cin, int32_t ks,
                           int32_t coutp, int32_t cinp, void* stream);
/* bias pack: E pointers to f32 [cout] -> f32 [E][coutp] */
int pmoe_pack_bias(const void* const* src_ptrs, float* dst, int32_t E, int32_t cout, int32_t coutp, void* stream);

/* Stand-alone activation + inverted dropout over n contiguous elements (n % (16 / sizeof(element)) == 0):
 * make_mlp with bn=True puts BatchNorm1d between the Linear and its activation (model/blocks/basics.py:30-39), so the
 * activation cannot ride in the GEMM epilogue there.  act = PMOE_ACT_*; backward from the saved output y. */
int pmoe_act_fwd(const void* x, void* y, int64_t n, int32_t act, float drop_p, uint64_t seed, int32_t dtype, void* stream);
int pmoe_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int32_t act, float drop_p, int32_t dtype, void* stream);

/* ---- BatchNorm2d, training mode (nn.BatchNorm2d at basics.py:101,121; torchvision bn1/bn2/downsample.1)
 * rows = N*H*W activations of C channels; expert e owns rows [e*rows_per_expert, ...).            */
/* per-(expert,channel) partial sums: part [E][nparts][2][C] f32 of (x - c) and (x - c)^2, where c = the channel's
 * value in the expert's first row, written to shiftc [E][C] (null: c = 0, plain sums).  Summing deviations from a
 * sample keeps the variance exact-to-rounding even when |mean| >> std. */
int pmoe_colstats(const void* x, int64_t rows_per_expert, int32_t E, int32_t C, int32_t ld, int32_t coff,
                  float* part, int32_t nparts, float* shiftc, int32_t dtype, void* stream);
/* deterministic tree step: part_in [E][nin][W] -> part_out [E][nout][W] (W = 2*C floats) */
int pmoe_reduce_partials(const float* part_in, float* part_out, int32_t E, int32_t nin, int32_t nout, int32_t width,
                         void* stream);
/* finalize: mean/var from partials; writes scale = gamma*invstd, shift = beta, mean, invstd; every consumer
 * evaluates the centred form (x - mean)*scale + shift (no cancellation when |mean| >> std)
 * ([E][C] f32 each); updates running_mean / running_var in place (momentum, unbiased var) through
 * per-expert pointer tables when they are non-null.  training=0: stats come from the running buffers. */
int pmoe_bn_finalize(const float* part, int32_t nparts, int64_t count, const void* const* gamma_ptrs,
                     const void* const* beta_ptrs, void* const* rmean_ptrs, void* const* rvar_ptrs, float momentum,
                     float eps, int32_t training, float* scale, float* shift, float* mean, float* invstd, int32_t E,
                     int32_t C, const float* shiftc, void* stream);
/* y = [relu]( (x - mean)*scale + shift [+ res] ); y is dense (y_ld = 0 or C) or the channel window
 * [y_coff, y_coff + C) of rows y_ld wide -- the skip half of a U-Net concatenation buffer (unet.py:72) */
/* y_fp8 (optional, bf16 + dense y only; round 3, BASELINE config 5): the same activation once more as e4m3(bf16(y) * in_scale)
 * bytes [rows][C] -- the input of the block-scaled fp8 convolution that consumes it (pmoe_conv_desc.in_fp8) */
int pmoe_bn_apply(const void* x, const void* res, void* y, const float* scale, const float* shift, const float* mean,
                  int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, int32_t y_ld, int32_t y_coff,
                  int32_t dtype, void* y_fp8, float in_scale, void* stream);
/* pmoe_bn_apply (no residual, dense y) that ALSO returns pmoe_gap_partial(y)'s partial sums [N][nparts][C], bit-identical to a
 * separate pmoe_gap_partial pass over the stored y: BatchNorm -> ReLU -> EfficientBlock of the stem (basics.py:113-123) without
 * re-reading the activation for the block's global average pool.  N images, ipe images per expert, HW pixels per image. */
int pmoe_bn_apply_gap(const void* x, void* y, const float* scale, const float* shift, const float* mean, float* gap_part,
                      int32_t nparts, int32_t N, int32_t ipe, int64_t HW, int32_t C, int32_t relu, int32_t dtype, void* stream);
/* backward reduce: g = dy * (relu ? y>0 : 1); partial sums of g and g*xhat -> part [E][nparts][2][C].
 * y may be NULL when relu is set and the forward had no residual: the mask is then recomputed as
 * x*scale+shift > 0 and the saved output is not read at all (one tensor pass less).
 * gmask_out (may be NULL): g itself is stored there -- pmoe_bn_bwd_apply then takes it as dy with relu = 0 and y = NULL
 * (it no longer reads the saved output) and the residual branch takes it as its gradient. */
int pmoe_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                       const float* scale, const float* shift, int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, float* part, int32_t nparts,
                       void* gmask_out, int32_t dtype, void* stream);
/* dgamma/dbeta [E][C] (written to the grad arena) + the two per-channel means used by bwd_apply */
int pmoe_bn_bwd_finalize(const float* part, int32_t nparts, int64_t count, float* dgamma, float* dbeta, float* c1,
                         float* c2, int32_t E, int32_t C, void* stream);
/* dx = gamma*invstd * (g - c1 - xhat*c2); optionally also stores g (the masked grad) to gmask_out */
/* round 4: y = [relu](BatchNorm(x)) into the channel window [y_coff, y_coff + C) of rows y_ld wide AND pooled = MaxPool2d(2, 2)(y)
 * (dense [E*ipe][H/2][W/2][C]) in one pass -- nn.Sequential(conv3) followed by nn.MaxPool2d(2) of the U-Net's down path
 * (model/blocks/unet.py:54-68) without re-reading y.  x dense [E*ipe][H][W][C]; H, W even; bit-identical to pmoe_bn_apply +
 * pmoe_maxpool2s2_fwd. */
int pmoe_bn_apply_pool2(const void* x, void* y, void* pooled, const float* scale, const float* shift, const float* mean,
                        int32_t ipe, int32_t H, int32_t W, int32_t E, int32_t C, int32_t relu, int32_t y_ld, int32_t y_coff,
                        int32_t dtype, void* stream);
int pmoe_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                      const float* scale, const float* shift, const float* c1, const float* c2, void* dx, void* gmask_out,
                      int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, int32_t dtype, void* stream);

/* ---- pooling (nn.MaxPool2d(3,2,1) and AdaptiveAvgPool2d(1) of the torchvision ResNet; the GAP of
 * EfficientBlock, basics.py:70) */
int pmoe_maxpool3s2_fwd(const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream);
int pmoe_maxpool3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream);
/* partial per-(image,channel) sums over HW: part [N][nparts][C] f32; optional second operand b:
 * sums of a*b (used for the ECA scale gradient) */
int pmoe_gap_partial(const void* a, const void* b, float* part, int32_t N, int64_t HW, int32_t C, int32_t nparts,
                     int32_t b_shared_ipe, int32_t dtype, void* stream);
/* mean over HW from partials, written as T into out[n*out_ld + out_coff + c] (feature concat slot) */
int pmoe_gap_finish(const float* part, void* out, int32_t N, int32_t C, int32_t nparts, int64_t HW, int32_t out_ld,
                    int32_t out_coff, int32_t dtype, void* stream);
/* dx[n][hw][c] = g[n*g_ld + g_coff + c] / HW */
int pmoe_gap_bwd(const void* g, void* dx, int32_t N, int64_t HW, int32_t C, int32_t g_ld, int32_t g_coff,
                 int32_t dtype, void* stream);

/* ---- fused stem tail: z2 -> BN+ReLU (conv2.1/.2, basics.py:121-123) -> BN+ReLU (ResNet bn1/relu) -> MaxPool(3,2,1).
 * The two intermediate activations and their gradients are re-derived from z2 in registers, never stored.
 * All per-channel arrays are [E][C] f32.  part: [E][nparts][2][C] partial sums (finish with pmoe_bn_finalize /
 * pmoe_bn_bwd_finalize). */
int pmoe_stem_tail_stats(const void* z2, const float* sc2, const float* sh2, const float* mu2, float* part,
                         int32_t nparts, float* shiftc, float* part_x, int32_t E, int32_t ipe, int32_t H, int32_t W,
                         int32_t C, int32_t dtype, void* stream);
/* part_x (optional) [E][nparts][3][C]: channel moments sum m, sum m*u, sum a2*u (m = [a2 > 0], u = z2 - mu2) that the
 * train-mode backward combines with the pooled pass below.  The argmax byte of pmoe_stem_tail_pool carries the winning
 * tap in bits 0..3 and [winner's a2 > 0] in bit 7. */
int pmoe_stem_tail_pool(const void* z2, void* y, uint8_t* argmax, const float* sc2, const float* sh2, const float* sc1,
                        const float* sh1, const float* mu2, const float* mu1, int32_t N, int32_t ipe, int32_t H, int32_t W, int32_t C, int32_t dtype,
                        void* stream);
/* phase 1: sums for the bn1 backward; phase 2: sums for the conv2-BN backward; phase 3: writes dz2.
 * consts: HOST array of 12 device pointers: sc2 sh2 sc1 sh1 mu1 is1 mu2 is2 c11 c21 c12 c22 (later ones may be
 * null in earlier phases). */
int pmoe_stem_tail_bwd(int32_t phase, const void* z2, const void* dpool, const uint8_t* argmax, void* dz2,
                       const float* const* consts, float* part, int32_t nparts, int32_t E, int32_t ipe, int32_t H,
                       int32_t W, int32_t C, int32_t dtype, void* stream);
/* Train mode: phases 1 and 2 as ONE pass over the pooled tensors (y = pooled output, dpool, argmax; 1/4 of the
 * elements) + a closed-form combine.  The pooled gradient only reaches the winning pixels, whose a3 is y itself, so
 * xhat1 / xhat2 there are recovered from y; the BatchNorm-backward terms that touch every pixel reduce to the channel
 * moments part_x of pmoe_stem_tail_stats.  part4 [E][nparts][4][C]; out1 / out2 [E][2][C] are the (sum g, sum g*xhat)
 * rows of bn1 / of the conv2 BatchNorm for pmoe_bn_bwd_finalize; count = pixels per expert (B*H*W).
 * (A channel whose BatchNorm scale is exactly 0 contributes xhat = 0: its value cannot be recovered from y.) */
int pmoe_stem_tail_pooled(const void* y, const void* dpool, const uint8_t* argmax, const float* const* consts,
                          float* part4, int32_t nparts, int32_t E, int64_t rows_per_expert, int32_t C, int32_t dtype,
                          void* stream);
int pmoe_stem_tail_combine(const float* part4, int32_t np4, const float* part_x, int32_t npx,
                           const float* const* consts, int64_t count, float* out1, float* out2, int32_t E, int32_t C,
                           void* stream);

/* ---- ECA channel attention (EfficientBlock.forward, basics.py:69-76) -------------------------
 * gate[n][c] = sigmoid(sum_j w[e][j] * mean_hw(x)[n][c + j - k/2]) from GAP partials (creal = number
 * of real channels the conv1d sees, rest is padding) */
int pmoe_eca_gate(const float* gap_part, int32_t nparts, int64_t HW, const void* const* w_ptrs, int32_t k,
                  float* gate, float* gapmean, int32_t N, int32_t ipe, int32_t in_ipe, int32_t C, int32_t creal,
                  void* stream);
/* y[n] = x[n or n%ipe] * gate[n]   (x_shared_ipe>0: x holds ipe images shared by all experts) */
int pmoe_eca_scale(const void* x, const float* gate, void* y, int32_t N, int64_t HW, int32_t C, int32_t x_shared_ipe,
                   int32_t dtype, void* stream);
/* from dgate-sums (sum_hw dy*x, partials) -> dpre, dgap [N][C] (times dgap_scale: 1, or 1/HW when the result is used
 * directly as the per-image bias of the gate-folded data-gradient convolution) and dw [E][k] (written to the arena);
 * dw_scratch: [N][k] f32 caller-owned scratch (per-image partials, summed per expert in fixed order) */
int pmoe_eca_bwd_small(const float* dot_part, int32_t nparts, const float* gate, const float* gapmean,
                       const void* const* w_ptrs, int32_t k, float* dgap, float* dw, float* dw_scratch, int32_t N,
                       int32_t ipe, int32_t C, int32_t creal, float dgap_scale, void* stream);
/* Stem input stage (conv1 reads frames * ECA gate): from per-image filter gradients G [N][ks*ks][coutp][cinp]
 * (pmoe_conv2d_wgrad with per_image=1 on the unscaled, shared frames) produce
 *   dw [E][cout][cin][ks][ks] = sum_n gate[n][c] * G[n]      and      ds [N][cinp] = sum_{k,t} W[e] * G[n],
 * the gradient of conv1.weight and of the ECA gate; no data-gradient convolution of conv1 is needed.
 * gate, ds: [N][gate_ld] f32; w_ptrs: E pointers to f32 [cout][cin][ks][ks]. */
int pmoe_eca_stem_fold(const float* G, const float* gate, const void* const* w_ptrs, float* dw, float* ds, int32_t N,
                       int32_t ipe, int32_t cout, int32_t cin, int32_t ks, int32_t coutp, int32_t cinp, int32_t gate_ld,
                       void* stream);
/* dx = dy*gate + dgap/HW */
int pmoe_eca_bwd_apply(const void* dy, const float* gate, const float* dgap, void* dx, int32_t N, int64_t HW,
                       int32_t C, int32_t dtype, void* stream);

/* ---- layout: images f32 [B][Cin][H][W] (images.view(B,-1,H,W), moe.py:90-92) -> T [B][H][W][Cp] */
int pmoe_nchw_to_nhwc(const float* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp,
                      int32_t dtype, void* stream);
/* small f32 [B][K] host-side inputs (speed, command) -> T [B][Kp] zero padded */
int pmoe_pad_rows(const float* src, void* dst, int32_t B, int32_t K, int32_t Kp, int32_t dtype, void* stream);

/* ---- gate softmax + Gaussian-mixture head (moe.py:98-100,150-153) and moe_loss (trainer/loss.py:121-132)
 * head [E*B][head_ld] T: cols 0..1 mean, 2..3 raw std, 4 raw alpha; spd [E*B][spd_ld] T col 0.
 * probs [B][E], mean/std [B][E][2], speeds [B][E][1] f32.  alpha_relu: bit 0 = ReLU on alpha (BaseExpert, moe.py:100) vs none
 * (BaseExpertAlt, moe.py:127); bit 1 = no softmax, `probs` receives alpha itself (a lone BaseExpert.forward, moe.py:74-101).
 * One 64-lane wave handles 64/G samples, G = pow2 >= E lanes per sample, xor-shuffle reductions over E.
 * shared = 1 is MixtureOfExpertsShared (moe.py:180-233): head [B][head_ld] with cols 4e..4e+3 = mean/raw std of
 * expert e and col 4E+e = its alpha (softmax without ReLU); spd [B][spd_ld] col 0; speeds is then [B][1]. */
int pmoe_gate_mixture_fwd(const void* head, int32_t head_ld, const void* spd, int32_t spd_ld, float* probs,
                          float* mean, float* std_, float* speeds, int32_t B, int32_t E, int32_t alpha_relu,
                          int32_t shared, int32_t dtype, void* stream);
int pmoe_gate_mixture_bwd(const void* head, int32_t head_ld, const float* probs, const float* dprobs,
                          const float* dmean, const float* dstd, const float* dspeeds, void* dhead, void* dspd,
                          int32_t spd_ld, int32_t B, int32_t E, int32_t alpha_relu, int32_t shared, int32_t dtype,
                          void* stream);
/* loss = c0 * mean_b(-logsumexp_e(log p + sum_d logN)) + c1 * mean((speeds-target)^2)/E; also the
 * gradients of loss wrt probs/mean/std/speeds for the backward pass.  shared_speed = 1: speeds is [B][1]
 * (MixtureOfExpertsShared) and the speed term is plain mse(speeds, target) (loss.py:129-130). */
int pmoe_moe_loss(const float* probs, const float* mean, const float* std_, const float* speeds,
                  const float* actions, const float* target_speed, float c0, float c1, float* loss, float* loglik,
                  float* dprobs, float* dmean, float* dstd, float* dspeeds, int32_t B, int32_t E, int32_t shared_speed,
                  void* stream);

/* ---- PU-Net / PMoE model types.  The U-Nets are frozen on this path
 * (moe.py:280): forward only.
 * MaxPool2d(2,2) (blocks/unet.py:29): x is a channel window [x_coff, x_coff+C) of rows x_ld wide, y dense [N][H/2][W/2][C]. */
int pmoe_maxpool2s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t x_ld,
                        int32_t x_coff, int32_t dtype, void* stream);
/* ConvTranspose2d(k=2, s=2) (unet.py:34-44) = one 1x1 GEMM with 4*C output rows (row (dy*2+dx)*C + c, packed by the
 * host from the [Cin][Cout][2][2] parameter) followed by this interleave into dst[n][2y+dy][2x+dx][dst_coff + c]
 * (a channel window of the skip-concatenation buffer, unet.py:71-84). */
int pmoe_pixel_shuffle2(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t src_ld,
                        int32_t dst_ld, int32_t dst_coff, int32_t dtype, void* stream);
/* torch.cat along channels (punet.py:104,113; unet.py:72) / view(B,-1,H,W) (moe.py:311-313): rows x C elements from one
 * channel window to another; 16-byte accesses when every offset allows it, element-wise otherwise (23-class masks). */
int pmoe_copy_window(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                     int64_t rows, int32_t C, int32_t dtype, void* stream);
/* PUNetExpert tail (moe.py:317): actions [B][2] = tanh(head[:, 0:2]), speeds [B] = spd[:, 0]; and its backward
 * (all head_ld / spd_ld columns of dhead / dspd are written, padding with zeros). */
int pmoe_action_head_fwd(const void* head, int32_t head_ld, const void* spd, int32_t spd_ld, float* actions,
                         float* speeds, int32_t B, int32_t dtype, void* stream);
int pmoe_action_head_bwd(const float* actions, const float* dactions, const float* dspeeds, void* dhead,
                         int32_t head_ld, void* dspd, int32_t spd_ld, int32_t B, int32_t dtype, void* stream);
/* punet_loss (loss.py:135-142): c0 * L1(actions, gt) + c1 * MSE(speeds, gt), with gradients; speeds = NULL gives
 * pmoe_loss (loss.py:145-151) = c0 * L1 (pass c0 = 1). */
int pmoe_action_loss(const float* actions, const float* speeds, const float* actions_gt, const float* speed_gt,
                     float c0, float c1, float* loss, float* dactions, float* dspeeds, int32_t B, void* stream);
/* PMoE blend (moe.py:353-356): out[b][j] = tanh(w_j . [moe[b][j], punet[b][j]] + bias_j), j = 0 lat_weights,
 * j = 1 long_weights (each nn.Linear(2,1)); backward gives the six parameter gradients and d punet (NULL to skip). */
int pmoe_blend_fwd(const float* moe_actions, const float* punet_actions, const float* lat_w, const float* lat_b,
                   const float* long_w, const float* long_b, float* out, int32_t B, void* stream);
int pmoe_blend_bwd(const float* moe_actions, const float* punet_actions, const float* lat_w, const float* long_w,
                   const float* out, const float* dout, float* dlat_w, float* dlat_b, float* dlong_w, float* dlong_b,
                   float* dpunet, int32_t B, void* stream);

/* ---- stage-1 PU-Net training (trainer/train_1.py:129-141) ------------------------------------------
 * Backward of nn.MaxPool2d(2,2) (blocks/unet.py:52-62): dx = scatter of dy to the FIRST maximum of each 2x2 window of x
 * (torch's tie rule) + dskip, the gradient that reaches the same activation through the skip concatenation
 * (unet.py:71-84; NULL = none).  x and dskip may be channel windows (ld, coff) of wider buffers; dx, dy are dense. */
int pmoe_maxpool2s2_bwd(const void* x, int32_t x_ld, int32_t x_coff, const void* dy, const void* dskip, int32_t dskip_ld,
                        int32_t dskip_coff, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                        void* stream);
/* backward of pmoe_pixel_shuffle2 (nn.ConvTranspose2d(k=2,s=2) scatter, unet.py:34-44):
 * dst[n,y,x,(dy*2+dx)*C + c] = src[n,2y+dy,2x+dx,src_coff + c];  src [N][2H][2W][src_ld], dst [N][H][W][dst_ld >= 4C] */
int pmoe_pixel_unshuffle2(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t N, int32_t H,
                          int32_t W, int32_t C, int32_t dtype, void* stream);
/* gradient of torch.cat along channels (punet.py:104,113): dst[r, dst_coff + c] += src[r, src_coff + c] */
int pmoe_add_window(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                    int64_t rows, int32_t C, int32_t dtype, void* stream);
/* torch.cat of K <= 8 equally wide channel windows in one launch (punet.py:104,113 four 23-class masks; moe.py:311-313
 * view(B,-1,H,W) of the F predicted masks): dst[r, j] = srcs[j / c][r, src_coff + j % c] for j < K*c and 0 for
 * K*c <= j < dst_c (dst_c a multiple of the 16-byte vector, <= dst_ld).  srcs is a HOST array of K device pointers,
 * every source [rows][src_ld]. */
int pmoe_cat_windows(const void* const* srcs, int32_t K, int32_t c, int32_t src_ld, int32_t src_coff, void* dst,
                     int32_t dst_ld, int32_t dst_c, int64_t rows, int32_t dtype, void* stream);
/* module boundary of PredictiveUnet.forward (punet.py:117-120 returns NCHW f32): src T [N][HW][src_ld] channel window
 * -> dst f32 [N][C][HW] */
int pmoe_nhwc_to_nchw(const void* src, int32_t src_ld, int32_t src_coff, float* dst, int32_t N, int64_t HW, int32_t C,
                      int32_t dtype, void* stream);
/* AutoregressiveCriterion (trainer/loss.py:86-118).  logits f32 [B][F][C][H][W], target int64 [B][F][H][W], C <= 32.
 * mode 0 'tversky': per frame ce_weight * cross_entropy(weight = class_dice, loss.py:6-17,48-57) + tversky_weight *
 * tversky_loss (loss.py:34-45 -- its TP/FP/FN ratio is formed per (class, image column), as the reference does);
 * mode 1 'l1' / mode 2 'l2': mean |x - onehot| / (x - onehot)^2 per frame.  loss[0] = sum over frames, loss[1+f] per frame.
 * Scratch (f32): partT [F][pmoe_seg_loss_rows()][3][pmoe_seg_loss_cp(C)][W] (upper bound), partG [F][rows][4][32],
 * coefG [F][32], coefT [F][2][cp][W]; coefG / coefT feed pmoe_seg_loss_bwd, which writes d loss / d logits scaled by
 * dloss[0] (device scalar; NULL = 1). */
int pmoe_seg_loss_rows(int32_t B, int32_t H, int32_t W);
int pmoe_seg_loss_cp(int32_t C);
int pmoe_seg_loss_fwd(const float* logits, const int64_t* target, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W,
                      int32_t mode, float ce_weight, float tversky_weight, float alpha, float beta, float* partT, float* partG,
                      float* coefG, float* coefT, float* loss, void* stream);
int pmoe_seg_loss_bwd(const float* logits, const int64_t* target, const float* coefG, const float* coefT, const float* dloss,
                      float* dlogits, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W, int32_t mode, void* stream);

/* ---- fused optimizer tail of the stage-2 step (reference caller trainer/train_2.py:157-165,184 + conf
 * stage_2_pmoe.yaml:11,137-144): torch.nn.utils.clip_grad_norm_, torch.optim.Adam(amsgrad=True).step() and
 * torch.optim.swa_utils.AveragedModel.update_parameters, each as ONE launch over a chunk table instead of a few
 * launches (and a host sync) per parameter tensor.  `table` is a device array of pmoe_opt_tensor (all f32);
 * workgroup i handles elements [chunk_index[i]*PMOE_OPT_CHUNK, +PMOE_OPT_CHUNK) of tensor chunk_tensor[i]. */
#define PMOE_OPT_CHUNK 16384
typedef struct pmoe_opt_tensor {
    float* param;
    const float* grad;
    float* exp_avg;
    float* exp_avg_sq;
    float* max_exp_avg_sq; /* amsgrad only */
    float* swa;            /* averaged copy (pmoe_mt_swa_update only) */
    int64_t numel;
    float bc1;             /* 1 - beta1^step of THIS tensor */
    float bc2_sqrt;        /* sqrt(1 - beta2^step) */
} pmoe_opt_tensor;
/* norm[0] = global L2 norm of all gradients, norm[1] = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1);
 * scale_grads != 0 also multiplies the gradients by norm[1] in place (= clip_grad_norm_).  partial: n_chunks floats. */
int pmoe_mt_grad_norm(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                      int32_t n_chunks, float max_norm, float* partial, float* norm, int32_t scale_grads, void* stream);
/* Adam / AMSGrad update (torch/optim/adam.py single-tensor formulas); norm != NULL applies the clip coefficient
 * norm[1] to the gradient on the fly (clip + step fused: the gradients themselves stay unscaled).
 * bc1_all / bc2_sqrt_all > 0: bias corrections shared by all tensors (every tensor at the same step, the usual case;
 * the cached table is then reused unchanged step after step); <= 0: the per-tensor bc1 / bc2_sqrt of the table. */
int pmoe_mt_adam(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index, int32_t n_chunks,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int32_t amsgrad, float bc1_all,
                 float bc2_sqrt_all, const float* norm, void* stream);
/* swa = param (n_averaged == 0) or swa + (param - swa) / (n_averaged + 1) */
int pmoe_mt_swa_update(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                       int32_t n_chunks, int64_t n_averaged, void* stream);

/* ---- input pipeline (callers: model/data_loader.py:255-275, autoagents/image_agent.py:71-78,132-136):
 * Crop([top, bottom]) -> torchvision Resize((h, w)) on a PIL image (= Pillow ImagingResample, BILINEAR with the support
 * stretched by the down-scaling factor) -> ToTensor (uint8 HWC -> f32 CHW / 255), bit-exact.  Two passes like Pillow:
 * horizontal over the cropped rows into an 8-bit intermediate, then vertical.  bounds [out][2] = (first tap, tap count),
 * coeffs [out][ksize] = 22-bit fixed-point weights, both from Resample.c:precompute_coeffs / normalize_coeffs_8bpc
 * (pmoe_amd/preprocess.py computes them in double precision). */
int pmoe_resample_u8_horizontal(const uint8_t* src, uint8_t* dst, int32_t n_img, int32_t H0, int32_t W0, int32_t row0,
                                int32_t rows, int32_t C, int32_t Wout, const int32_t* bounds, const int32_t* coeffs,
                                int32_t ksize, void* stream);
int pmoe_resample_u8_vertical_to_f32(const uint8_t* src, float* dst_nchw, int32_t n_img, int32_t Hin, int32_t W,
                                     int32_t C, int32_t Hout, const int32_t* bounds, const int32_t* coeffs,
                                     int32_t ksize, void* stream);
/* stage-1 label pipeline (data_loader.py:282-286,305-309: Crop -> Resize -> MaskPILToTensor on a single-channel class-id
 * image): the same vertical pass writing the 8-bit result as int64 [n][C][Hout][W].  (The reference resizes its label
 * images with BILINEAR like the frames -- class ids are blended at region borders; reproduced as is.) */
int pmoe_resample_u8_vertical_to_i64(const uint8_t* src, int64_t* dst_nchw, int32_t n_img, int32_t Hin, int32_t W,
                                     int32_t C, int32_t Hout, const int32_t* bounds, const int32_t* coeffs,
                                     int32_t ksize, void* stream);

#ifdef __cplusplus
}
#endif
#endif
