// Internal launcher interfaces shared by the .hip translation units and the C-ABI layer (api.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pmoe_hip.h"

struct ConvArgs {
    const void* in;        // [Nin][H][W][in_ld] T ; channels [in_coff, in_coff+Cin) are reduced over
    const void* w;         // [E][CoutP][ks*ks][Cin] T
    void* out;             // [N][Ho][Wo][out_ld] T ; channels [out_coff, out_coff+Cout) are written
    const void* res;       // residual / saved activation, geometry of `out` with its own ld/coff
    const float* bias;     // [E][CoutP] f32 or null
    float* stats;          // [mblocks][2][CoutP] f32 partial (sum, sum of squares) or null
    const float* oscale;   // w_fp8: [E][CoutP] f32, accumulator scale = weight scale / in_scale
    float in_scale;        // w_fp8: activations are stored in LDS as e4m3(x * in_scale)
    int w_fp8;             // 1: `w` holds e4m3 bytes [E][CoutP][ks*ks][Cin]
    int in_fp8;            // 1: `in` holds e4m3 bytes too (e4m3(x * in_scale), written by pmoe_bn_apply's fp8 side output)
    int N, H, W, Cin;
    int Ho, Wo, Cout, CoutP;
    int in_ld, in_coff, out_ld, out_coff, res_ld, res_coff;
    int ipe;               // images per expert (N = E * ipe)
    int in_shared;         // 1: the input holds ipe images shared by all experts
    int ks, stride, pad, dilate;
    int act, res_mode;
    float drop_p;
    unsigned long long seed;
    // filled by the launcher
    int lTW, lTH, TN, n_groups, tiles_y, tiles_x;
    // tap window + output lattice (defaults: kh = kw = ks, identity tap order, dense output).  A stride-2 data gradient
    // is issued as 4 launches, one per output parity class (py,px): a kh x kw = (1+py) x (1+px) stride-1 correlation over
    // dy whose taps are `tapmap` entries of the 3x3 filter, written to dx[2a+py][2b+px] (out_step 2).
    int kh, kw, use_tapmap, tapmap[4];
    int out_step, out_offy, out_offx, OH, OW;
    const float* bn;       // PMOE_RES_DBN: [4][N / bn_ipe][Cout] f32 (mean, invstd, scale, beta)
    int bn_ipe;
    int stagger;           // launcher: waves 4-7 of the 8-wave tile run one k-substep behind their SIMD partners
    int prefetch;          // launcher: 2 patch buffers, the next channel chunk's halo patch is fetched under the MFMAs
    int shuf_c;            // > 0 (1x1 direct kernel, round 4): ConvTranspose2d(k2,s2) scatter fused -- out is [N][2Ho][2Wo][out_ld], output
                           // channel q*shuf_c + c (q = 2 dy + dx) goes to pixel (2 oy + dy, 2 ox + dx), channel out_coff + c
};

struct WgradArgs {
    const void* x;         // conv input  [Nin][H][W][x_ld] T
    const void* dy;        // output grad [N][Ho][Wo][dy_ld] T
    float* dw;             // [E][taps][CoutP][CinP] f32, overwritten (every element is stored once; no atomics)
    float* part;           // K-split partial slabs [nsplit][E][taps][CoutP][CinP] f32 (conv_wgrad_ws_floats; may be null if 0)
    long long part_floats;
    int N, H, W, Cin, CinP;      // Cin: multiple of the channel chunk; CinP: row length of dw
    int Ho, Wo, Cout, CoutP;     // Cout: multiple of the channel chunk actually reduced
    int x_ld, x_coff, dy_ld, dy_coff;
    int ipe, x_shared;
    int ks, stride, pad;
    int per_image;         // 1: dw is [N][taps][CoutP][CinP] -- one slab per IMAGE (no sum over the expert's images)
    float* grads;          // optional: the parameter's own gradient [E][cout_real][cin_real][ks][ks] f32, written INSTEAD of dw
    int cout_real, cin_real;
    int defer_fold;        // 1: conv_wgrad_launch stops after the MFMA kernel (conv_wgrad_fold runs the tail)
    int lTW, lTH, TN, n_groups, tiles_y, tiles_x, mb_per_wg;
    int slice_fastest;     // launcher: grid order of round 1 (A/B switch)
};

// grouped skinny GEMM for the expert MLP layers (gemm_skinny.hip)
bool gemm_skinny_ok(const ConvArgs& a, int dtype);
int gemm_skinny_launch(const ConvArgs& a, hipStream_t st);

// LDS-DMA 3x3 kernel for the >= 128-channel stride-1 layers (conv_dma.hip)
bool conv_dma_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf);
int conv_dma_launch(ConvArgs a, hipStream_t st);
bool conv_dma_uses_mf16(const ConvArgs& a);
bool conv_dma_uses_producer(const ConvArgs& a);
bool conv_dma_uses_stream(const ConvArgs& a);
bool conv_dma_is_narrow(const ConvArgs& a);
int conv_dma_plan_code(const ConvArgs& a);
// ... on the block-scaled fp8 matrix instruction (e4m3 weights and activations)
bool conv_dma_f8_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf);
int conv_dma_f8_launch(ConvArgs a, hipStream_t st);      // which instantiation: <true> = v_mfma_f32_16x16x32_bf16
// ... and its stride-2 forward sibling (parity planes gathered by the DMA's per-lane source addresses)
bool conv_dma_s2_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf);
int conv_dma_s2_launch(ConvArgs a, hipStream_t st);
// ... and one parity class of a stride-2 3x3 data gradient on the same kernel (ConvArgs with the class fields set)
bool conv_dma_s2cls_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf);
int conv_dma_s2cls_launch(ConvArgs a, hipStream_t st);

struct ResPlan;
int conv_igemm_launch(const ConvArgs& a, int dtype, hipStream_t st);
int conv_igemm_mblocks(const ConvArgs& a, int dtype);
int conv_igemm_plan(const ConvArgs& a, int dtype);
int conv_wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st);
int conv_wgrad_fold(const WgradArgs& a, int dtype, hipStream_t st);
long long conv_wgrad_ws_floats(const WgradArgs& a, int dtype);
int conv_wgrad_plan(const WgradArgs& a, int dtype);
// per-image filter gradient with the BatchNorm backward applied on load (conv_wgrad.hip, round 4)
int conv_wgrad_bnbwd_launch(WgradArgs a, const void* z, int z_ld, const float* coef, const float* c1, const float* c2,
                            int dtype, hipStream_t st, bool plan);
