// Grouped "skinny" GEMM for the expert MLPs (make_mlp, blocks/basics.py:10-44; heads moe.py:70-72): per expert
//   Y[m][n] = epilogue( sum_k X[m][k] * W[n][k] ),   m < images-per-expert (the batch, 64), n = 16..1536, k = 16..1536
// i.e. the 1x1 "convolutions" over 1x1 images of the grouped engine -- and, with a loop over the filter taps (row m = output
// pixel, its operand row = the tap-shifted input pixel, zero outside the image), the 3x3 / 1x1 convolutions of feature maps
// with at most PMOE_SKINNY_MAXROWS (default 3200) pixels per expert (closed-loop inference at B = 1: layer2 .. layer4; the 14x14
// bottleneck of the stage-1 U-Net at B = 10; beyond that the LDS-staged kernels win: measured).  On the generic implicit-GEMM kernel these are a
// serial chain of 8..24 channel chunks on 16 workgroups (~90 us per launch: 2 TFLOP/s).  Here one workgroup owns a
// 64-row x 64-column tile of ONE expert and its 4 waves split K between them: every wave streams its k-slices of both
// operands straight from global memory in MFMA-fragment shape (16 bytes per lane, K-contiguous rows of the packed
// weights / of the activation rows), with no LDS staging and no barrier in the loop -- all loads of a wave are
// independent -- and the four partial tiles meet once in LDS for the epilogue (bias / act' / activation / dropout /
// residual, the same arithmetic as conv_igemm.hip's epilogue).  bf16 operands, f32 accumulation.
#include <stdlib.h>
#include "common.h"
#include "kernels.h"

static constexpr int BM = 64, BN = 64, VE = 8;
static constexpr int NREG = 4;          // partial-tile regions in LDS (4 x 16 KiB); 8-wave workgroups fold waves 4-7 onto 0-3 first

template <int NW>
__global__ void __launch_bounds__(NW * 64) gemm_skinny_kernel(const ConvArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* stg = reinterpret_cast<float*>(smem);            // [NW][BM][BN] f32, 16-byte units XOR-swizzled by row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int e = blockIdx.z, m0 = blockIdx.y * BM, cout0 = blockIdx.x * BN;
    const int TAPS = a.ks * a.ks;
    const int HWo = a.Ho * a.Wo, rows_pe = a.ipe * HWo;          // output rows (pixels) of one expert: contiguous in NHWC
    const bf16* W = (const bf16*)a.w + ((size_t)e * a.CoutP + cout0) * TAPS * a.Cin;
    const bf16* X = (const bf16*)a.in;

    const bf16* pa[2];
    int nimg[2], iy0[2], ix0[2];
    bool bok[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        pa[t] = W + (size_t)(t * 32 + l31) * TAPS * a.Cin + h * 8;
        const int m = m0 + t * 32 + l31;
        bok[t] = m < rows_pe;
        const int mm = bok[t] ? m : 0;
        const int nl = mm / HWo, rem = mm - nl * HWo, oy = rem / a.Wo, ox = rem - oy * a.Wo;
        nimg[t] = (a.in_shared ? 0 : e * a.ipe) + nl;
        iy0[t] = oy * a.stride - a.pad;
        ix0[t] = ox * a.stride - a.pad;
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const v4i zero = {0, 0, 0, 0};
    for (int tap = 0; tap < TAPS; ++tap) {
        const int r = tap / a.ks, q = tap - r * a.ks;
        const bf16* pb[2];
        bool ok[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int iy = iy0[t] + r, ix = ix0[t] + q;
            ok[t] = bok[t] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            pb[t] = X + (((size_t)nimg[t] * a.H + (ok[t] ? iy : 0)) * a.W + (ok[t] ? ix : 0)) * a.in_ld + a.in_coff + h * 8;
        }
        const int woff = tap * a.Cin;
        // wave w takes the 16-wide k-slices w, w+4, w+8, ... of every tap
#pragma unroll 4
        for (int k = wave * 16; k < a.Cin; k += NW * 16) {
            v4i af[2], bfr[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                af[t] = ldg16(pa[t] + woff + k);
                bfr[t] = ok[t] ? ldg16(pb[t] + k) : zero;
            }
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                          __builtin_bit_cast(bf16x8, bfr[mt]), acc[nt][mt], 0, 0, 0);
        }
    }
    // ---- partial tiles -> LDS: D[cout][m], row m of wave w at stg[w][m][.]
    constexpr int UPR = BN / 4;
    float* mine = stg + (wave % NREG) * (BM * BN);
#pragma unroll
    for (int round = 0; round < NW / NREG; ++round) {
        // round 0: the upper waves (or all, with 4 waves) store; round 1: waves 0-3 add their own tile on top
        const bool my_turn = (NW == NREG) || (round == 0 ? wave >= NREG : wave < NREG);
        if (my_turn) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const int p = mt * 32 + l31;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        const int u = nt * 8 + 2 * g + h;
                        f32x4* dst = reinterpret_cast<f32x4*>(mine + p * BN + ((u ^ (p & (UPR - 1))) << 2));
                        f32x4 v = (NW > NREG && round == 1) ? *dst : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                        for (int i = 0; i < 4; ++i) v[i] += acc[nt][mt][4 * g + i];
                        *dst = v;
                    }
                }
        }
        __syncthreads();
    }
    // ---- epilogue (conv_igemm.hip's, on 1x1 images: output pixel index = image index)
    constexpr int CPO = BN / VE, PROWS = NW * 64 / CPO;
    const int cc = tid % CPO, pr = tid / CPO;
    const int cout = cout0 + cc * VE;
    const bool cvalid = cout < a.Cout;
    float bias[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) bias[i] = (a.bias && cvalid) ? a.bias[(size_t)e * a.CoutP + cout + i] : 0.f;
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    bf16* out = (bf16*)a.out;
    const bf16* res = (const bf16*)a.res;
    for (int p = pr; p < BM; p += PROWS) {
        const int m = m0 + p;
        if (!cvalid || m >= rows_pe) continue;
        float v[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) v[i] = bias[i];
#pragma unroll
        for (int w = 0; w < NREG; ++w)
#pragma unroll
            for (int k = 0; k < VE / 4; ++k) {
                const int u = cc * (VE / 4) + k;
                const f32x4 tt = *reinterpret_cast<const f32x4*>(stg + w * (BM * BN) + p * BN + ((u ^ (p & (UPR - 1))) << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * k + i] += tt[i];
            }
        const size_t opix = (size_t)e * rows_pe + m;            // dense NHWC output: pixel index = expert base + row
        if (a.res_mode) {
            float rv[VE];
            unpack16<bf16>(ldg16(res + opix * a.res_ld + a.res_coff + cout), rv);
            if (a.res_mode == PMOE_RES_ADD) {
#pragma unroll
                for (int i = 0; i < VE; ++i) v[i] += rv[i];
            } else if (a.res_mode == PMOE_RES_DRELU) {
#pragma unroll
                for (int i = 0; i < VE; ++i) v[i] = rv[i] > 0.f ? v[i] * keep_scale : 0.f;
            } else if (a.res_mode >= PMOE_RES_DELU) {               // saved output y = act(z) * mask * keep_scale
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    const float y = rv[i] * (1.f / keep_scale);
                    const float d = act_deriv_from_output(a.res_mode, y);
                    v[i] = (a.drop_p > 0.f && rv[i] == 0.f) ? 0.f : v[i] * d * keep_scale;
                }
            }
        }
        if (a.act == PMOE_ACT_RELU) {
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] = fmaxf(v[i], 0.f);
        } else if (a.act != PMOE_ACT_NONE) {                     // elu / tanh / sigmoid
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] = act_apply(a.act, v[i]);
        }
        if (a.drop_p > 0.f && a.res_mode < PMOE_RES_DRELU) {
            const unsigned long long base = (unsigned long long)opix * (unsigned)a.Cout + cout;
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] = hash_uniform(a.seed, base + i) >= a.drop_p ? v[i] * keep_scale : 0.f;
        }
        stg16(out + opix * a.out_ld + a.out_coff + cout, pack16<bf16>(v));
    }
}

// bf16, no fused statistics, dense output lattice, and FEW output rows per expert: the expert MLP layers (1x1 "images") and
// their data gradients, and the 3x3 / 1x1 convolutions of tiny feature maps at tiny batches (closed-loop inference, B = 1:
// layer3 / layer4 are 14x14 / 7x7 -- on the generic kernel a serial chain of 36..72 tap-chunks on a dozen workgroups)
bool gemm_skinny_ok(const ConvArgs& a, int dtype) {
    static int on = -1, maxrows = 0;   // PMOE_GEMM_SKINNY=0: back to the generic implicit-GEMM kernel (A/B measurements)
    if (on < 0) {
        const char* ev = getenv("PMOE_GEMM_SKINNY");
        on = ev ? atoi(ev) : 1;
        const char* mr = getenv("PMOE_SKINNY_MAXROWS");
        maxrows = mr ? atoi(mr) : 3200;
    }
    if (!on || dtype != PMOE_DT_BF16 || a.stats || a.dilate || a.use_tapmap || a.out_step != 1) return false;
    if ((a.ks != 1 && a.ks != 3) || a.kh != a.ks || a.kw != a.ks || (a.stride != 1 && a.stride != 2)) return false;
    const bool mlp = a.H == 1 && a.W == 1 && a.Ho == 1 && a.Wo == 1 && a.ks == 1;
    if (!mlp && (long long)a.ipe * a.Ho * a.Wo > maxrows) return false;
    if (a.Ho != (a.H + 2 * a.pad - a.ks) / a.stride + 1 || a.Wo != (a.W + 2 * a.pad - a.ks) / a.stride + 1) return false;
    return a.Cin > 0 && a.Cin % 16 == 0 && a.CoutP % BN == 0 && a.Cout % VE == 0 && a.ipe > 0 && a.N % a.ipe == 0 &&
           a.in_ld % VE == 0 && a.in_coff % VE == 0 && a.out_ld % VE == 0 && a.out_coff % VE == 0 &&
           (!a.res_mode || (a.res && a.res_ld % VE == 0 && a.res_coff % VE == 0));
}

template <int NW> static int skinny_launch_nw(const ConvArgs& a, const dim3& grid, hipStream_t st) {
    const int smem = NREG * BM * BN * (int)sizeof(float);
    HIP_RET((ensure_dyn_lds<gemm_skinny_kernel<NW>>(smem)));
    hipLaunchKernelGGL(gemm_skinny_kernel<NW>, grid, dim3(NW * 64), smem, st, a);
    return (int)hipGetLastError();
}

int gemm_skinny_launch(const ConvArgs& a, hipStream_t st) {
    const dim3 grid(a.CoutP / BN, (a.ipe * a.Ho * a.Wo + BM - 1) / BM, a.N / a.ipe);
    static int nw8 = -1;            // PMOE_SKINNY_NW8: reduction length (taps * Cin) from which 8 waves split K (default 0 = never: measured no faster)
    if (nw8 < 0) { const char* ev = getenv("PMOE_SKINNY_NW8"); nw8 = ev ? atoi(ev) : 0; }
    if (nw8 > 0 && a.ks * a.ks * a.Cin >= nw8) return skinny_launch_nw<8>(a, grid, st);
    return skinny_launch_nw<4>(a, grid, st);
}


// ------------------------------------------------------------------------------------------------
// Round 4: weight (and bias) gradients of the expert MLP layers (autograd of nn.Linear in make_mlp, blocks/basics.py:31, and of
// the heads moe.py:70-72):   dW[e][n][k] = sum over the expert's batch rows m of dY[m][n] * X[m][k],   db[e][n] = sum_m dY[m][n].
// GEMM-K is the batch (64..256 rows): 10 launches per step that the generic conv_wgrad_kernel served at ~45 us each -- its
// 256-pixel m-block is three quarters zero fill at 64 rows, its workgroups are one per CU (96 KB of LDS tiles), its K-split
// machinery adds a fold launch -- plus a column-sum launch chain per bias.  Here: one 4-wave workgroup per 64 x 64 tile of
// ONE expert (24 KB of LDS: six workgroups per CU, the whole grid resident at once), both operands staged as [row][channel]
// rows of 192 bytes and read TRANSPOSED (ds_read_b64_tr_b16, as conv_wgrad.hip), the tile written straight into the
// parameter's own [E][out][in] gradient, the bias gradient folded from the staged dY tile by the workgroups of input-tile 0.
// Fixed summation order, no atomics: bit-reproducible.
struct MlpWgradArgs {
    const void* x;         // [Nx][x_ld] bf16 rows; channels [x_coff, x_coff + Cin)
    const void* dy;        // [N][dy_ld] bf16 rows; channels [dy_coff, dy_coff + Cout)
    float* grads;          // [E][cout_real][cin_real] f32
    float* bias_grads;     // [E][cout_real] f32 or null
    int N, ipe, x_shared;
    int Cin, Cout, cin_real, cout_real;      // Cin / Cout: staged channel counts (multiples of 8)
    int x_ld, x_coff, dy_ld, dy_coff;
};

typedef __attribute__((address_space(3))) s16x4 lds_s16x4_m;
__device__ __forceinline__ s16x4 mlp_tr_read(const char* p) { return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4_m*)(p)); }

__global__ void __launch_bounds__(256) mlp_wgrad_kernel(const MlpWgradArgs a) {
    constexpr int RS = 192;
    __shared__ __attribute__((aligned(16))) char dyt[64 * RS];
    __shared__ __attribute__((aligned(16))) char xt[64 * RS];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int co_sub = wave & 1, ci_sub = wave >> 1;
    const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64, e = blockIdx.z;
    const bf16* dy = (const bf16*)a.dy + (size_t)e * a.ipe * a.dy_ld + a.dy_coff + co0;
    const bf16* x = (const bf16*)a.x + (a.x_shared ? (size_t)0 : (size_t)e * a.ipe * a.x_ld) + a.x_coff + ci0;
    f32x16 acc;
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    const int sj = tid & 7, sr = tid >> 3;                           // staging: chunk sj of rows sr, sr + 32
    const bool dy_cok = co0 + sj * 8 < a.Cout, x_cok = ci0 + sj * 8 < a.Cin;
    const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const int ca = (co_sub * 32 + 16 * (g & 1) + 4 * pc) * 2, cb = (ci_sub * 32 + 16 * (g & 1) + 4 * pc) * 2;
    float bsum = 0.f;                                                // threads 0..63 of the input-tile-0 workgroups: column tid of dY
    const bool want_bias = a.bias_grads && blockIdx.x == 0;
    for (int r0 = 0; r0 < a.ipe; r0 += 64) {
        v4i dv[2], xv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int r = r0 + sr + 32 * u;
            dv[u] = xv[u] = v4i{0, 0, 0, 0};
            if (r < a.ipe) {
                if (dy_cok) dv[u] = ldg16(dy + (size_t)r * a.dy_ld + sj * 8);
                if (x_cok) xv[u] = ldg16(x + (size_t)r * a.x_ld + sj * 8);
            }
        }
        __syncthreads();                                             // the previous chunk's fragment reads are done
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            *reinterpret_cast<v4i*>(dyt + (sr + 32 * u) * RS + sj * 16) = dv[u];
            *reinterpret_cast<v4i*>(xt + (sr + 32 * u) * RS + sj * 16) = xv[u];
        }
        __syncthreads();
#pragma unroll
        for (int kb = 0; kb < 4; ++kb) {
            bf16x8 fa, fb;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int p = kb * 16 + 8 * (g >> 1) + 4 * tt + q;
                const bf16x4 ra = __builtin_bit_cast(bf16x4, mlp_tr_read(dyt + p * RS + ca));
                const bf16x4 rb = __builtin_bit_cast(bf16x4, mlp_tr_read(xt + p * RS + cb));
#pragma unroll
                for (int i = 0; i < 4; ++i) { fa[4 * tt + i] = ra[i]; fb[4 * tt + i] = rb[i]; }
            }
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc, 0, 0, 0);
        }
        if (want_bias && tid < 64) {
            const bf16* col = reinterpret_cast<const bf16*>(dyt) + tid;
            for (int r = 0; r < 64; ++r) bsum += (float)col[r * (RS / 2)];       // rows beyond the batch were staged as zeros
        }
    }
    const int cin = ci0 + ci_sub * 32 + l31;
    if (cin < a.cin_real) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cout = co0 + co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (cout < a.cout_real) a.grads[((size_t)e * a.cout_real + cout) * a.cin_real + cin] = acc[r];
        }
    }
    if (want_bias && tid < 64 && co0 + tid < a.cout_real) a.bias_grads[(size_t)e * a.cout_real + co0 + tid] = bsum;
}

int mlp_wgrad_launch(const MlpWgradArgs& a, hipStream_t st) {
    if (!a.x || !a.dy || !a.grads || a.ipe <= 0 || a.N % a.ipe) return PMOE_ERR_ARG;
    if (a.Cin % 8 || a.Cout % 8 || a.x_ld % 8 || a.x_coff % 8 || a.dy_ld % 8 || a.dy_coff % 8) return PMOE_ERR_ARG;
    if (a.x_coff + a.Cin > a.x_ld || a.dy_coff + a.Cout > a.dy_ld) return PMOE_ERR_ARG;
    if (a.cin_real <= 0 || a.cin_real > a.Cin || a.cout_real <= 0 || a.cout_real > a.Cout) return PMOE_ERR_ARG;
    const dim3 grid((a.Cin + 63) / 64, (a.Cout + 63) / 64, a.N / a.ipe);
    hipLaunchKernelGGL(mlp_wgrad_kernel, grid, dim3(256), 0, st, a);
    return (int)hipGetLastError();
}

extern "C" int pmoe_mlp_wgrad(const void* x, const void* dy, float* grads, float* bias_grads, int32_t n, int32_t ipe,
                              int32_t x_shared, int32_t cin, int32_t cout, int32_t cin_real, int32_t cout_real, int32_t x_ld,
                              int32_t x_coff, int32_t dy_ld, int32_t dy_coff, int32_t dtype, void* stream) {
    if (dtype != PMOE_DT_BF16) return PMOE_ERR_UNSUPPORTED;
    MlpWgradArgs a{x, dy, grads, bias_grads, n, ipe, x_shared, cin, cout, cin_real, cout_real, x_ld, x_coff, dy_ld, dy_coff};
    return mlp_wgrad_launch(a, (hipStream_t)stream);
}
