// Grouped (per-expert) NHWC convolution as an implicit GEMM on MFMA, gfx950.
//
// One workgroup owns an output patch of TN x TH x TW pixels (BM = 64*WM of them) x BN = 64*WN
// output channels.  Per input-channel chunk the (TH-1)*S+KS by (TW-1)*S+KS input halo patch is
// staged ONCE in LDS (XOR-swizzled 16-byte chunks, zero filled outside the image) and every
// filter tap reads its MFMA operand from it at a shifted pixel address, so an input byte crosses
// L2->LDS once per chunk instead of once per tap.  Weight tiles [BN][chunk] stream per (chunk,tap)
// through a 2-deep LDS ring, prefetched into registers under the MFMAs of the previous tap.
// MFMA roles: A = weights (rows = cout), B = pixels (cols = pixel); D[cout][pixel] is staged
// through LDS in f32 and written back as whole 16-byte channel vectors per pixel (coalesced),
// where bias / activation / residual / dropout / per-channel BatchNorm partial sums are fused.
//
// The same kernel serves: forward conv 3x3/1x1 stride 1/2, stride-1 data-gradient (flipped,
// transposed weights), stride-2 data-gradient (`dilate`: the source is read as if zero-upsampled
// by 2, i.e. a transposed conv), and the expert MLP heads (1x1 conv on 1x1 images = grouped GEMM
// over experts with TN samples per tile).
#include "conv_common.h"
#include <stdlib.h>
#include "kernels.h"

template <int LOG_RB> __device__ __forceinline__ int swz(int x) { return swz_chunk<LOG_RB, 0>(x); }

template <typename T> struct Mma;
template <> struct Mma<bf16> {
    static __device__ __forceinline__ void run(const v4i& a, const v4i& b, f32x16& c) {
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
    }
};
template <> struct Mma<float> {
    // 16 bytes = 4 f32 per lane-half: MFMA j consumes element j of both operands (k = 4h + j).
    static __device__ __forceinline__ void run(const v4i& a, const v4i& b, f32x16& c) {
        const f32x4 fa = __builtin_bit_cast(f32x4, a), fb = __builtin_bit_cast(f32x4, b);
#pragma unroll
        for (int j = 0; j < 4; ++j) c = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[j], fb[j], c, 0, 0, 0);
    }
};

template <> struct Mma<fp8> {
    // 16 bytes = 16 e4m3 per lane-half: two K=16 MFMAs (bf16 rate), each takes 8 bytes of both operands.  A and B use the
    // same byte -> k assignment, so the order of k inside the 32-channel step is immaterial.
    static __device__ __forceinline__ void run(const v4i& a, const v4i& b, f32x16& c) {
        typedef long l2 __attribute__((ext_vector_type(2)));
        const l2 la = __builtin_bit_cast(l2, a), lb = __builtin_bit_cast(l2, b);
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la[0], lb[0], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(la[1], lb[1], c, 0, 0, 0);
    }
};

// -DPMOE_STAMP (tools/stamp_conv.py only, never the product build): per-wave cycle accounting of the main loop with
// s_memtime (scalar registers: no VGPR cost).  The five sums land in a.stats instead of the BatchNorm partial sums.
#ifdef PMOE_STAMP
#define STAMP_INIT unsigned long long st_prev = __builtin_amdgcn_s_memtime(); unsigned st_acc[5] = {0, 0, 0, 0, 0};
#define LAP(i) { const unsigned long long st_t = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(st_t - st_prev); st_prev = st_t; }
#else
#define STAMP_INIT
#define LAP(i)
#endif

// LITE: the 8-wave tile without chunk prefetch and stagger and with the epilogue staged in two halves -- few enough
// registers (launch bound: 2 workgroups per CU) and LDS for TWO resident workgroups, whose prologues / epilogues / barriers
// then overlap each other's MFMAs.
// TL: element type of the MFMA operands in LDS (and of the packed weights in HBM): T, or fp8 -- e4m3 weights with a
// per-output-channel scale (pmoe_pack_conv_weights_fp8), bf16 activations converted x * in_scale -> e4m3 by the patch
// loader, v_mfma_f32_32x32x16_fp8_fp8, accumulators multiplied by a.oscale[e][cout] in the epilogue (BASELINE config 5).
template <typename T, int LOG_RB, int WM, int WN, bool LITE, typename TL = T>
__device__ __forceinline__ void conv_igemm_body(const ConvArgs& a) {
    constexpr int RB = 1 << LOG_RB, CPR = RB / 16, LOG_CPR = LOG_RB - 4;
    constexpr int NTHR = WM * WN * 64;
    constexpr int BM = WM * 64, BN = WN * 64;
    constexpr int VE = 16 / (int)sizeof(T);        // output elements per 16 bytes
    constexpr int VEL = 16 / (int)sizeof(TL);      // operand elements per 16-byte LDS chunk
    constexpr int CK = RB / (int)sizeof(TL);
    constexpr bool F8 = sizeof(TL) != sizeof(T);
    constexpr int KSUB = RB / 32;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave / WN, wn = wave % WN;

    // LITE launches a flat grid: output-channel block fastest, so that the workgroups that read the SAME input patch (one per
    // cout block) are neighbours in dispatch order and -- through the XCD remap -- on the same XCD's L2
    const int nblk = LITE ? a.CoutP / BN : 1;
    const unsigned flat = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned mb = LITE ? flat / nblk : flat;
    const int nb = LITE ? (int)(flat % nblk) : (int)blockIdx.y;
    int t = (int)mb;
    const int px = t % a.tiles_x; t /= a.tiles_x;
    const int py = t % a.tiles_y; t /= a.tiles_y;
    const int ng = t % a.n_groups;
    const int e = t / a.n_groups;
    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    // a strided 1x1 conv stages only the pixels it uses: LDS pixel stride 1, source step = stride
    const int KS = a.ks, TAPS = KS * KS;           // TAPS = tap stride of the packed weights
    const int KH = a.kh, KW = a.kw, NTAP = KH * KW; // taps actually visited (a sub-window for the parity classes)
    const int S = KS == 1 ? 1 : a.stride, step = KS == 1 ? a.stride : 1;
    const int PW = (TW - 1) * S + KW, PH = (TH - 1) * S + KH;
    const int NPIX = a.TN * PH * PW;
    const int n0 = e * a.ipe + ng * a.TN, n_end = (e + 1) * a.ipe;
    const int oy0 = py * TH, ox0 = px * TW;
    const int Y0 = oy0 * a.stride - a.pad, X0 = ox0 * a.stride - a.pad;
    const int cout0 = nb * BN;

    // one halo-patch buffer, or two when the launcher asked for chunk prefetch (a.prefetch)
    const int PB = (NPIX * RB + 255) & ~255;
    char* patch0 = smem;
    char* wbuf = smem + PB * ((!LITE && a.prefetch) ? 2 : 1);

    int ppb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        ppb[mt] = (pn * PH + my * S) * PW + mx * S;
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const TL* wbase = (const TL*)a.w + ((size_t)e * a.CoutP + cout0) * TAPS * a.Cin;
    constexpr int WV = (BN * CPR + NTHR - 1) / NTHR;
    v4i wreg[WV];
    auto w_issue = [&](int c0, int tap) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < BN * CPR) {
                const int n = v >> LOG_CPR, j = v & (CPR - 1);
                wreg[i] = ldg16(wbase + ((size_t)n * TAPS + tap) * a.Cin + c0 + j * VEL);
            }
        }
    };
    auto w_commit = [&](int buf) {
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int v = tid + i * NTHR;
            if (v < BN * CPR) {
                const int n = v >> LOG_CPR, j = v & (CPR - 1);
                *reinterpret_cast<v4i*>(wbuf + buf * (BN * RB) + n * RB + ((j ^ swz<LOG_RB>(n)) << 4)) = wreg[i];
            }
        }
    };

    PatchGeom geo;
    geo.n0 = n0; geo.n_end = n_end; geo.e_first_img = e * a.ipe;
    geo.Y0 = Y0; geo.X0 = X0; geo.PH = PH; geo.PW = PW; geo.NPIX = NPIX;
    geo.H = a.H; geo.W = a.W; geo.ld = a.in_ld; geo.coff = a.in_coff; geo.cmax = a.Cin;
    geo.dilate = a.dilate; geo.shared = a.in_shared; geo.step = step;
    const T* in = (const T*)a.in;
    // vectors per thread held in registers while a patch is in flight: the 8-wave tile needs half as many per thread
    constexpr int MV1 = NTHR == 512 ? 6 : 12, MV2 = NTHR == 512 ? 10 : 20;
    const bool one_batch = NPIX * CPR <= MV1 * NTHR;    // every load of the patch in flight at once
    const bool big_batch = NPIX * CPR <= MV2 * NTHR;    // stride-2 halos (17x33 / 33x33 pixels)
    auto load_patch = [&](char* patch, int c0) {
        if constexpr (F8) {                          // convert on the way in (two loads per LDS chunk)
            load_halo_patch<T, LOG_RB, NTHR, 0, TL>(patch, in, geo, c0, tid, a.in_scale);
        } else if (LITE) {                           // batches of 4 vectors: few registers, the other resident workgroup hides it
            load_halo_patch<T, LOG_RB, NTHR, 0>(patch, in, geo, c0, tid);
        } else if (one_batch) {
            PatchStage<T, LOG_RB, NTHR, MV1> ps;
            ps.issue(in, geo, c0, tid);
            ps.template commit<0>(patch, NPIX, tid);
        } else if (big_batch) {
            PatchStage<T, LOG_RB, NTHR, MV2> ps;
            ps.issue(in, geo, c0, tid);
            ps.template commit<0>(patch, NPIX, tid);
        } else {
            load_halo_patch<T, LOG_RB, NTHR, 0>(patch, in, geo, c0, tid);
        }
    };

    const int nchunks = a.Cin / CK;
    STAMP_INIT
    int cur = 0;
    auto wtap = [&](int i) { return a.use_tapmap ? a.tapmap[i] : i; };
    w_issue(0, wtap(0));
    w_commit(0);
    PatchStage<T, LOG_RB, NTHR, 6> nxt;             // prefetch registers (live across the tap loop only when a.prefetch)
    // Stagger (MI355X_MICROARCH.md "two waves that run the same program"): waves 4-7 -- the SIMD partners of waves 0-3 in the
    // 8-wave tile -- hold back the MFMAs of each tap's LAST k-substep and issue them after the barrier, i.e. under the
    // partners' first LDS reads of the next tap, so the two waves of a SIMD stop reaching MFMA bursts, LDS bursts and
    // barriers in lockstep.  Fragments wait in registers; results are bit-identical (same MFMAs, same order per accumulator).
    const bool stag = !LITE && (WM * WN == 8) && wave >= 4 && a.stagger;
    constexpr int NDEF = KSUB >= 2 ? KSUB / 2 : 1;      // k-substeps held back: half a tap
    v4i paf[NDEF][2], pbf[NDEF][2];
    bool pend = false;
    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * CK;
        char* patch = (!LITE && a.prefetch) ? patch0 + (ch & 1) * PB : patch0;
        if (LITE || !a.prefetch || ch == 0) {
            load_patch(patch, c0);
            __syncthreads();
            LAP(0)                                       // halo patch of this chunk staged (incl. its barrier)
        }
        const bool more = !LITE && a.prefetch && ch + 1 < nchunks;
        if (more) nxt.issue(in, geo, c0 + CK, tid);      // in flight under this chunk's taps of MFMAs
        for (int r = 0; r < KH; ++r) {
            for (int q = 0; q < KW; ++q) {
                const int tap = r * KW + q;
                const bool last = (ch == nchunks - 1) && (tap == NTAP - 1);
                int ntap = tap + 1, nc0 = c0;
                if (ntap == NTAP) { ntap = 0; nc0 += CK; }
                if (!last) w_issue(nc0, wtap(ntap));
                const char* wb = wbuf + cur * (BN * RB);
                const int tapoff = r * PW + q;
                int pp[2], fp[2];
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    pp[mt] = ppb[mt] + tapoff;
                    fp[mt] = swz<LOG_RB>(pp[mt]);
                }
                if (pend) {
#pragma unroll
                    for (int d = 0; d < NDEF; ++d)
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) Mma<TL>::run(paf[d][nt], pbf[d][mt], acc[nt][mt]);
                    pend = false;
                }
#pragma unroll
                for (int ks = 0; ks < KSUB; ++ks) {
                    v4i af[2], bfr[2];
#pragma unroll
                    for (int nt = 0; nt < 2; ++nt) {
                        const int row = wn * 64 + nt * 32 + l31;
                        af[nt] = *reinterpret_cast<const v4i*>(wb + row * RB + (((ks * 2 + h) ^ swz<LOG_RB>(row)) << 4));
                    }
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        bfr[mt] = *reinterpret_cast<const v4i*>(patch + pp[mt] * RB + (((ks * 2 + h) ^ fp[mt]) << 4));
                    if (stag && ks >= KSUB - NDEF) {
                        const int d = ks - (KSUB - NDEF);
                        paf[d][0] = af[0]; paf[d][1] = af[1]; pbf[d][0] = bfr[0]; pbf[d][1] = bfr[1];
                        pend = true;
                    } else {
#pragma unroll
                        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt) Mma<TL>::run(af[nt], bfr[mt], acc[nt][mt]);
                    }
                }
                LAP(1)                                   // fragment reads + MFMA issue of this tap
                if (!last) w_commit(cur ^ 1);
                if (more && tap == NTAP - 1) nxt.template commit<0>(patch0 + ((ch + 1) & 1) * PB, NPIX, tid);
                LAP(2)                                   // wait for the next tap's weights + their LDS writes
                __syncthreads();
                LAP(3)                                   // barrier
                cur ^= 1;
            }
        }
    }

    if (pend) {
#pragma unroll
        for (int d = 0; d < NDEF; ++d)
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) Mma<TL>::run(paf[d][nt], pbf[d][mt], acc[nt][mt]);
    }
    // ---- epilogue: D[cout][pixel] -> LDS f32 [BM / EP][BN] (16B units XOR-swizzled by pixel) ----
    constexpr int UPR = BN / 4;
    constexpr int EP = LITE ? 2 : 1, BMH = BM / EP;      // LITE: two halves of 128 pixel rows through a 64 KiB staging area
    float* stg = reinterpret_cast<float*>(smem);

    constexpr int CPO = BN / VE;
    constexpr int PROWS = NTHR / CPO;
    const int cc = tid % CPO, pr = tid / CPO;
    const int cout = cout0 + cc * VE;
    const bool cvalid = cout < a.Cout;
    float bias[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) bias[i] = (a.bias && cvalid) ? a.bias[(size_t)e * a.CoutP + cout + i] : 0.f;
    float osc[F8 ? VE : 1];                          // fp8: weight scale / activation scale per output channel
    if constexpr (F8) {
#pragma unroll
        for (int i = 0; i < VE; ++i) osc[i] = cvalid ? a.oscale[(size_t)e * a.CoutP + cout + i] : 0.f;
    }
    float s1[VE], s2[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) s1[i] = s2[i] = 0.f;
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    T* out = (T*)a.out;
    const T* res = (const T*)a.res;
  for (int half = 0; half < EP; ++half) {
    if (half) __syncthreads();                      // the previous half has been read out
    if (EP == 1 || wm / (WM / EP) == half) {
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int p = wm * 64 + mt * 32 + l31 - half * BMH;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int u = wn * 16 + nt * 8 + 2 * g + h;
                    f32x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = acc[nt][mt][4 * g + i];
                    *reinterpret_cast<f32x4*>(stg + p * BN + ((u ^ (p & (UPR - 1))) << 2)) = v;
                }
            }
    }
    __syncthreads();
    for (int ph = pr; ph < BMH; ph += PROWS) {
        const int p = ph + half * BMH;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        const int n = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
        const bool ok = cvalid && n < n_end && oy < a.Ho && ox < a.Wo;
        float v[VE];
#pragma unroll
        for (int k = 0; k < VE / 4; ++k) {
            const int u = cc * (VE / 4) + k;
            const f32x4 tt = *reinterpret_cast<const f32x4*>(stg + ph * BN + ((u ^ (ph & (UPR - 1))) << 2));
#pragma unroll
            for (int i = 0; i < 4; ++i) v[4 * k + i] = tt[i];
        }
        if (ok) {
            const size_t opix = ((size_t)n * a.OH + oy * a.out_step + a.out_offy) * a.OW + ox * a.out_step + a.out_offx;
            if constexpr (F8) {
#pragma unroll
                for (int i = 0; i < VE; ++i) v[i] *= osc[i];
            }
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] += bias[i];
            if (a.res_mode) {
                float rv[VE];
                unpack16<T>(ldg16(res + opix * a.res_ld + a.res_coff + cout), rv);
                if (a.res_mode == PMOE_RES_ADD) {
#pragma unroll
                    for (int i = 0; i < VE; ++i) v[i] += rv[i];
                } else if (a.res_mode == PMOE_RES_DRELU) {          // saved output y: relu'(y) (with dropout scale)
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
                for (int i = 0; i < VE; ++i)
                    v[i] = hash_uniform(a.seed, base + i) >= a.drop_p ? v[i] * keep_scale : 0.f;
            }
            const v4i pk = pack16<T>(v);
#ifndef PMOE_STAMP
            if (a.stats) {
                float rr[VE];
                unpack16<T>(pk, rr);
#pragma unroll
                for (int i = 0; i < VE; ++i) { s1[i] += rr[i]; s2[i] += rr[i] * rr[i]; }
            }
#endif
            stg16(out + opix * a.out_ld + a.out_coff + cout, pk);
        }
    }
  }
#ifdef PMOE_STAMP
    LAP(4)                                               // epilogue
    if (a.stats && lane == 0) {
        float* o = a.stats + ((size_t)blockIdx.x * (WM * WN) + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = (float)st_acc[i];
    }
    return;
#endif
    if (a.stats) {
        // BatchNorm partial sums of this tile: lanes sharing a channel vector combine by xor-shuffle,
        // waves through LDS; one [2][CoutP] row per m-block, reduced later in fixed order.
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);              // [waves][2][BN]
#pragma unroll
        for (int i = 0; i < VE; ++i) {
#pragma unroll
            for (int off = CPO; off < 64; off <<= 1) {
                s1[i] += __shfl_xor(s1[i], off);
                s2[i] += __shfl_xor(s2[i], off);
            }
        }
        if (CPO >= 64 || lane < CPO) {
            // CPO > 64 cannot occur (BN<=128, VE>=4 -> CPO<=32)
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                red[(wave * 2 + 0) * BN + cc * VE + i] = s1[i];
                red[(wave * 2 + 1) * BN + cc * VE + i] = s2[i];
            }
        }
        __syncthreads();
        if (tid < 2 * BN) {
            const int which = tid / BN, c = tid % BN;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < WM * WN; ++w) s += red[(w * 2 + which) * BN + c];
            if (cout0 + c < a.CoutP) a.stats[((size_t)mb * 2 + which) * a.CoutP + cout0 + c] = s;
        }
    }
}

template <typename T, int LOG_RB, int WM, int WN, typename TL = T>
__global__ void __launch_bounds__(WM * WN * 64) conv_igemm_kernel(const ConvArgs a) {
    conv_igemm_body<T, LOG_RB, WM, WN, false, TL>(a);
}

template <typename T, int LOG_RB, typename TL = T>
__global__ void __launch_bounds__(512, 4) conv_igemm_lite_kernel(const ConvArgs a) {
    conv_igemm_body<T, LOG_RB, 4, 2, true, TL>(a);
}

template <typename T, int LOG_RB, typename TL = T>
static int launch_lite(const ConvArgs& a, int mblocks, size_t smem, hipStream_t st) {
    auto k = conv_igemm_lite_kernel<T, LOG_RB, TL>;
    HIP_RET((ensure_dyn_lds<conv_igemm_lite_kernel<T, LOG_RB, TL>>(160 * 1024)));
    dim3 grid(mblocks * (a.CoutP / 128), 1, 1), block(512, 1, 1);
    hipLaunchKernelGGL(k, grid, block, smem, st, a);
    return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
template <typename T, int LOG_RB, int WM, int WN, typename TL = T>
static int launch_cfg(const ConvArgs& a, int mblocks, size_t smem, hipStream_t st) {
    auto k = conv_igemm_kernel<T, LOG_RB, WM, WN, TL>;
    HIP_RET((ensure_dyn_lds<conv_igemm_kernel<T, LOG_RB, WM, WN, TL>>(160 * 1024)));
    dim3 grid(mblocks, a.CoutP / (WN * 64), 1), block(WM * WN * 64, 1, 1);
    hipLaunchKernelGGL(k, grid, block, smem, st, a);
    return (int)hipGetLastError();
}

template <typename T, typename TL = T> static int launch_dtype(ConvArgs a, hipStream_t st, int* out_mblocks, int* out_cfg = nullptr) {
    const int esz = (int)sizeof(TL);                // operand element size in LDS
    constexpr bool F8 = sizeof(TL) != sizeof(T);
    if (a.Cin <= 0 || a.CoutP % 64 || a.Cout % (16 / (int)sizeof(T)) || a.Cin % (32 / esz)) return PMOE_ERR_ARG;
    if (F8 && (a.Cin % 64 || (!out_mblocks && (!a.oscale || !(a.in_scale > 0.f))))) return PMOE_ERR_ARG;
    if ((a.ks != 1 && a.ks != 3) || (a.stride != 1 && a.stride != 2) || (a.dilate && a.stride != 1)) return PMOE_ERR_ARG;
    if (a.N % a.ipe) return PMOE_ERR_ARG;
    const int E = a.N / a.ipe;
    const bool wide = (a.CoutP % 128 == 0);
    // bf16, >= 128 output channels: 8 waves on a 256-pixel x 128-channel tile (each weight tile and each barrier serves
    // twice the pixels of the 4-wave 128 x 128 tile: +5-8 % on the 3x3 layers, +30 % on the stride-2 forward convs)
    static int cfg42 = -1;      // PMOE_CONV_CFG42=0 switches back to the 4-wave tile (A/B measurements)
    if (cfg42 < 0) { const char* ev = getenv("PMOE_CONV_CFG42"); cfg42 = ev ? atoi(ev) : 1; }
    const bool big = wide && cfg42 && sizeof(T) == 2 && (long long)a.ipe * a.Ho * a.Wo >= 4096;   // not the MLP GEMMs
    const int log_rb_min = F8 ? 6 : 5;              // fp8: 64 or 128 channels per chunk
    // (measured on the stage-1 U-Net at B = 10, where the 28x28 / 14x14 layers give < 256 workgroups: falling back to the
    // 128-pixel 4-wave tile to double the workgroup count is SLOWER, 41.9 vs 32.3 ms of conv time per step)
    const int BM = wide ? (big ? 256 : 128) : 256, BN = wide ? 128 : 64;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    // candidate chunk widths (bytes per pixel row in LDS), widest first
    for (int log_rb = 7; log_rb >= log_rb_min; --log_rb) {
        const int rb = 1 << log_rb, ck = rb / esz;
        if (a.Cin % ck) continue;
        int lTW = p2(a.Wo); if (lTW > 5) lTW = 5;
        int lBM = p2(BM);
        int lTH = p2(a.Ho); if (lTH > lBM - lTW) lTH = lBM - lTW;
        const int TN = BM >> (lTW + lTH);
        const int TW = 1 << lTW, TH = 1 << lTH;
        const int lstride = a.ks == 1 ? 1 : a.stride;
        const int PW = (TW - 1) * lstride + a.kw, PH = (TH - 1) * lstride + a.kh;
        const size_t pbytes = ((size_t)TN * PH * PW * rb + 255) & ~(size_t)255;
        // chunk prefetch: 8-wave tile, >= 2 channel chunks, the whole patch in <= 6 vectors per thread (24 VGPRs)
        const int nthr = big ? 512 : 256;
        // 64-channel chunks: the LITE instantiation (<= 128 VGPRs, 73 KiB LDS) puts TWO 8-wave workgroups on a CU, which
        // beats one workgroup with chunk prefetch + stagger by 12-30 % (l2 +12 %, l3 +30 %, l4 +25 %: 980-1000 TFLOP/s).
        // PMOE_CONV_LITE=0 switches back for A/B runs.
        static int lite_on = -1;
        if (lite_on < 0) { const char* ev = getenv("PMOE_CONV_LITE"); lite_on = ev ? atoi(ev) : 1; }
        const bool lite = lite_on && big && log_rb == 7 && pbytes + 2 * (size_t)BN * rb <= 80 * 1024;   // two must fit a CU
        static int stg_on = -1;     // PMOE_CONV_STAGGER=0: A/B switch
        if (stg_on < 0) { const char* ev = getenv("PMOE_CONV_STAGGER"); stg_on = ev ? atoi(ev) : 1; }
        a.stagger = stg_on && big && !lite;
        // (LITE with every load of a chunk's halo patch in flight at once -- one L2 round trip instead of two -- measured
        // within noise, +-2 %, and tools/stamp_conv.py still showed 28 % of the wave cycles in patch staging: the cost is
        // not the number of round trips)
        static int pf_on = -1;      // PMOE_CONV_PREFETCH=0: A/B switch
        if (pf_on < 0) { const char* ev = getenv("PMOE_CONV_PREFETCH"); pf_on = ev ? atoi(ev) : 1; }
        // (also the expert MLP GEMMs: 1x1 "images", K = 512..1536 in 64-channel chunks -- a latency chain of 8..24 chunks)
        a.prefetch = !F8 && !lite && pf_on && (big || (a.H == 1 && a.W == 1 && a.ks == 1)) && a.Cin / ck >= 2 && (size_t)TN * PH * PW * (rb / 16) <= (size_t)6 * nthr &&
                     2 * pbytes + 2 * (size_t)BN * rb <= 150 * 1024;
        size_t smem = pbytes * (a.prefetch ? 2 : 1) + 2 * (size_t)BN * rb;
        const size_t stg = (size_t)BM * BN * 4 / (lite ? 2 : 1);
        if (smem < stg) smem = stg;
        if (smem > 150 * 1024) continue;
        a.lTW = lTW; a.lTH = lTH; a.TN = TN;
        a.n_groups = (a.ipe + TN - 1) / TN;
        a.tiles_y = (a.Ho + TH - 1) / TH;
        a.tiles_x = (a.Wo + TW - 1) / TW;
        const int mblocks = E * a.n_groups * a.tiles_y * a.tiles_x;
        if (out_cfg) *out_cfg = (F8 ? 8000 : 0) + (lite ? 2000 + log_rb : log_rb * 100 + (wide ? (big ? 42 : 22) : 41));   // see conv_igemm_plan
        if (out_mblocks) { *out_mblocks = mblocks; return 0; }
        if constexpr (F8) {
            if (lite) return launch_lite<T, 7, TL>(a, mblocks, smem, st);
            if (log_rb == 7 && big) return launch_cfg<T, 7, 4, 2, TL>(a, mblocks, smem, st);
            if (log_rb == 6 && big) return launch_cfg<T, 6, 4, 2, TL>(a, mblocks, smem, st);
            if (log_rb == 7) return wide ? launch_cfg<T, 7, 2, 2, TL>(a, mblocks, smem, st) : launch_cfg<T, 7, 4, 1, TL>(a, mblocks, smem, st);
            return wide ? launch_cfg<T, 6, 2, 2, TL>(a, mblocks, smem, st) : launch_cfg<T, 6, 4, 1, TL>(a, mblocks, smem, st);
        }
        if (lite) return launch_lite<T, 7>(a, mblocks, smem, st);
        if (log_rb == 7 && big) return launch_cfg<T, 7, 4, 2>(a, mblocks, smem, st);
        if (log_rb == 6 && big) return launch_cfg<T, 6, 4, 2>(a, mblocks, smem, st);
        if (log_rb == 5 && big) return launch_cfg<T, 5, 4, 2>(a, mblocks, smem, st);
        if (log_rb == 7) return wide ? launch_cfg<T, 7, 2, 2>(a, mblocks, smem, st) : launch_cfg<T, 7, 4, 1>(a, mblocks, smem, st);
        if (log_rb == 6) return wide ? launch_cfg<T, 6, 2, 2>(a, mblocks, smem, st) : launch_cfg<T, 6, 4, 1>(a, mblocks, smem, st);
        return wide ? launch_cfg<T, 5, 2, 2>(a, mblocks, smem, st) : launch_cfg<T, 5, 4, 1>(a, mblocks, smem, st);
    }
    return PMOE_ERR_UNSUPPORTED;
}

// resident-weight ping-pong kernel for the <=64-channel 3x3 stride-1 layers (conv_res.hip)
struct ResPlan {
    int lTW, lTH, TN, n_groups, tiles_y, tiles_x, tiles_per_expert, wgs_per_expert, log_rb;
    size_t smem;
};
bool conv_res_plan(const ConvArgs& a, int dtype, ResPlan* plan);
bool conv_res_pipe_ok(const ConvArgs& a);
bool conv_res_dma_ok(const ConvArgs& a, const ResPlan& p, int* pbuf, int* magic_pw, int* magic_ph, size_t* smem);
bool conv_c16_plan(const ConvArgs& a, int dtype, int* wgs_per_expert, int* tiles_x, int* tiles_per_expert);
int conv_c16_launch(const ConvArgs& a, hipStream_t st);
bool conv_c1x1_plan(const ConvArgs& a, int dtype, int* wgs_per_expert, int* tiles_per_expert, int* n_slabs, int* mt, size_t* smem);
int conv_c1x1_launch(const ConvArgs& a, hipStream_t st);
int conv_res_launch(ConvArgs a, const ResPlan& p, hipStream_t st);

// stride-2 3x3 data gradient (forward pad 1) by output parity class instead of a zero-dilated source: dx[2a+py][2b+px]
// only receives the taps ky = 1 (py = 0) or ky in {2, 0} at dy rows {a, a+1} (py = 1), same along x -> 1+2+2+4 = 9 tap
// visits per 2x2 output pixels instead of 36.  Weight taps are those of the flipped dgrad packing (fy = 2 - ky).
// (One launch with the class in the block index was measured slower: LDS sized for the widest class, uneven work.)
static int launch_stride2_dgrad(const ConvArgs& a, int dtype, hipStream_t st) {
    for (int py = 0; py < 2; ++py)
        for (int px = 0; px < 2; ++px) {
            ConvArgs c = a;
            c.dilate = 0; c.stride = 1; c.pad = 0;
            c.kh = 1 + py; c.kw = 1 + px;
            c.Ho = (a.Ho - py + 1) / 2; c.Wo = (a.Wo - px + 1) / 2;
            if (c.Ho <= 0 || c.Wo <= 0) continue;
            c.OH = a.Ho; c.OW = a.Wo; c.out_step = 2; c.out_offy = py; c.out_offx = px;
            c.use_tapmap = 1;
            for (int r = 0; r < c.kh; ++r)
                for (int q = 0; q < c.kw; ++q) {
                    const int fy = py ? (r == 0 ? 0 : 2) : 1, fx = px ? (q == 0 ? 0 : 2) : 1;
                    c.tapmap[r * c.kw + q] = fy * 3 + fx;
                }
            {
                ConvArgs d = c;
                int mb, pb;
                size_t sm;
                if (conv_dma_s2cls_plan(d, dtype, &mb, &sm, &pb)) {          // LDS-DMA kernel (conv_dma.hip)
                    const int rcd = conv_dma_s2cls_launch(c, st);
                    if (rcd) return rcd;
                    continue;
                }
            }
            const int rc = dtype == PMOE_DT_BF16 ? launch_dtype<bf16>(c, st, nullptr)
                         : dtype == PMOE_DT_F32 ? launch_dtype<float>(c, st, nullptr) : PMOE_ERR_ARG;
            if (rc) return rc;
        }
    return 0;
}

// stride-2 1x1 data gradient (the ResNet downsample conv) ACCUMULATED IN PLACE into an existing gradient: only the
// even-even pixels receive anything, so one class-(0,0) launch adds there and the other 3/4 of the tensor is not touched
// (the zero-dilated form reads and rewrites all of it).
static int launch_stride2_1x1_inplace(const ConvArgs& a, int dtype, hipStream_t st) {
    ConvArgs c = a;
    c.dilate = 0; c.stride = 1; c.pad = 0; c.kh = c.kw = 1;
    c.Ho = (a.Ho + 1) / 2; c.Wo = (a.Wo + 1) / 2;
    c.OH = a.Ho; c.OW = a.Wo; c.out_step = 2; c.out_offy = c.out_offx = 0;
    return dtype == PMOE_DT_BF16 ? launch_dtype<bf16>(c, st, nullptr)
         : dtype == PMOE_DT_F32 ? launch_dtype<float>(c, st, nullptr) : PMOE_ERR_ARG;
}

int conv_igemm_launch(const ConvArgs& a, int dtype, hipStream_t st) {
    if (a.w_fp8) {                                   // e4m3 weights: forward convs
        if (dtype != PMOE_DT_BF16 || a.dilate) return PMOE_ERR_ARG;
        if (a.in_fp8) {                              // e4m3 activations too: the block-scaled MFMA kernel, or nothing
            ConvArgs c = a;
            int mb, pb;
            size_t sm;
            return conv_dma_f8_plan(c, dtype, &mb, &sm, &pb) ? conv_dma_f8_launch(a, st) : PMOE_ERR_UNSUPPORTED;
        }
        return launch_dtype<bf16, fp8>(a, st, nullptr);
    }
    if (a.shuf_c) {                                  // ConvTranspose2d scatter fused into the store: the 1x1 direct kernel only
        int wpe, tpe, slabs, mt;
        size_t sm;
        return conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &sm) ? conv_c1x1_launch(a, st) : PMOE_ERR_UNSUPPORTED;
    }
    if (a.res_mode == PMOE_RES_INBN) {               // BatchNorm + ReLU of the INPUT on load: conv3x3_respipe_kernel<false, 3> only
        ResPlan plan;
        int pb, mpw, mph;
        size_t sm;
        if (!a.bias && conv_res_plan(a, dtype, &plan) && conv_res_dma_ok(a, plan, &pb, &mpw, &mph, &sm) && conv_res_pipe_ok(a))
            return conv_res_launch(a, plan, st);
        {
            int wpe, tpe, slabs, mt;                    // 1x1 layers: conv1x1_direct_kernel<MT, true>
            size_t smx;
            if (conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &smx)) return conv_c1x1_launch(a, st);
        }
        return PMOE_ERR_UNSUPPORTED;
    }
    if (a.res_mode == PMOE_RES_DBN) {                // BatchNorm-backward reductions in the epilogue: the two LDS-DMA kernels only
        ResPlan plan;
        int pb, mpw, mph, mb;
        size_t sm;
        if (conv_res_plan(a, dtype, &plan) && conv_res_dma_ok(a, plan, &pb, &mpw, &mph, &sm)) return conv_res_launch(a, plan, st);
        ConvArgs c = a;
        if (!gemm_skinny_ok(a, dtype) && conv_dma_plan(c, dtype, &mb, &sm, &pb)) return conv_dma_launch(a, st);
        return PMOE_ERR_UNSUPPORTED;
    }
    if (a.dilate && a.ks == 3 && a.pad == 1 && a.kh == 3) return launch_stride2_dgrad(a, dtype, st);
    if (a.dilate && a.ks == 1 && a.pad == 0 && a.res_mode == PMOE_RES_ADD && a.res == a.out && a.res_ld == a.out_ld &&
        a.res_coff == a.out_coff && !a.bias && a.act == PMOE_ACT_NONE && a.drop_p == 0.f)
        return launch_stride2_1x1_inplace(a, dtype, st);
    if (gemm_skinny_ok(a, dtype)) return gemm_skinny_launch(a, st);
    {
        int wpe, tx, tpe, slabs, mt;
        size_t sm;
        if (conv_c16_plan(a, dtype, &wpe, &tx, &tpe)) return conv_c16_launch(a, st);
        if (conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &sm)) return conv_c1x1_launch(a, st);
    }
    ResPlan plan;
    if (conv_res_plan(a, dtype, &plan)) return conv_res_launch(a, plan, st);
    {
        ConvArgs c = a;
        int mb, pb;
        size_t sm;
        if (conv_dma_plan(c, dtype, &mb, &sm, &pb)) return conv_dma_launch(a, st);
        c = a;
        if (conv_dma_s2_plan(c, dtype, &mb, &sm, &pb)) return conv_dma_s2_launch(a, st);
    }
    if (dtype == PMOE_DT_BF16) return launch_dtype<bf16>(a, st, nullptr);
    if (dtype == PMOE_DT_F32) return launch_dtype<float>(a, st, nullptr);
    return PMOE_ERR_ARG;
}

// which kernel a descriptor runs on (no launch): 3000 = gemm_skinny_kernel; 1000 + LOG_RB = conv3x3_res_kernel<LOG_RB>; 1207 + 10 bias + 20 mode =
// conv3x3_respipe_kernel<bias, mode> (1107 | 1117 = conv3x3_resdma_kernel with PMOE_RES_PIPE=0); 1316 = conv3x3_c16_kernel; 1400 + MT = conv1x1_direct_kernel<MT>; 5007 = conv3x3_dma_kernel; 2000 + LOG_RB =
// conv_igemm_lite_kernel<T, LOG_RB>; LOG_RB*100 + WM*10 + WN = conv_igemm_kernel<T, LOG_RB, WM, WN>; + 4000 = the four
// parity-class launches of a stride-2 data gradient
int conv_igemm_plan(const ConvArgs& a, int dtype) {
    ConvArgs c = a;
    int extra = 0;
    if (a.w_fp8 && a.in_fp8) {
        ConvArgs d = a;
        int mb, pb;
        size_t sm;
        return conv_dma_f8_plan(d, dtype, &mb, &sm, &pb) ? 8507 : PMOE_ERR_UNSUPPORTED;      // conv3x3_dma_f8_kernel
    }
    if (a.w_fp8) {                                   // 8000 + the bf16 code of the same tile
        int mb = 0, cfg = 0;
        const int rc = dtype == PMOE_DT_BF16 && !a.dilate ? launch_dtype<bf16, fp8>(c, nullptr, &mb, &cfg) : PMOE_ERR_ARG;
        return rc ? rc : cfg;
    }
    if (a.shuf_c) {
        int wpe, tpe, slabs, mt;
        size_t smx;
        return conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &smx) ? 1450 + mt + (a.res_mode == PMOE_RES_INBN ? 10 : 0)
                                                                       : PMOE_ERR_UNSUPPORTED;
    }
    if (a.res_mode == PMOE_RES_INBN) {
        ResPlan plan;
        int pb, mpw, mph;
        size_t sm;
        if (!a.bias && conv_res_plan(a, dtype, &plan) && conv_res_dma_ok(a, plan, &pb, &mpw, &mph, &sm) && conv_res_pipe_ok(a))
            return 1267;                                 // conv3x3_respipe_kernel<false, 3>
        int wpe, tpe, slabs, mt;
        size_t smx;
        return conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &smx) ? 1410 + mt : PMOE_ERR_UNSUPPORTED;      // conv1x1_direct_kernel<MT, true>
    }
    if (a.res_mode == PMOE_RES_DBN) {
        ResPlan plan;
        int pb, mpw, mph, mb;
        size_t sm;
        if (conv_res_plan(a, dtype, &plan) && conv_res_dma_ok(a, plan, &pb, &mpw, &mph, &sm))
            return 1000 + plan.log_rb + (conv_res_pipe_ok(a) ? 240 : 100) + (a.bias ? 10 : 0);
        if (!gemm_skinny_ok(a, dtype) && conv_dma_plan(c, dtype, &mb, &sm, &pb)) return conv_dma_plan_code(a);
        return PMOE_ERR_UNSUPPORTED;
    }
    if (gemm_skinny_ok(a, dtype)) return 3000;           // gemm_skinny_kernel
    if (a.dilate && a.ks == 3 && a.pad == 1 && a.kh == 3) {
        c.dilate = 0; c.stride = 1; c.pad = 0; c.kh = c.kw = 2;
        c.Ho = a.Ho / 2; c.Wo = a.Wo / 2; c.OH = a.Ho; c.OW = a.Wo; c.out_step = 2;
        if (c.Ho <= 0 || c.Wo <= 0) return PMOE_ERR_ARG;
        extra = 4000;
        {
            ConvArgs d = c;
            d.use_tapmap = 1; d.out_offy = d.out_offx = 1;
            int mb, pb;
            size_t sm;
            if (conv_dma_s2cls_plan(d, dtype, &mb, &sm, &pb)) return 4000 + 5207;      // 4 launches of conv3x3s2_dma_kernel<true>
        }
    } else {
        int wpe, tx, tpe, slabs, mt;
        size_t smx;
        if (conv_c16_plan(a, dtype, &wpe, &tx, &tpe)) return 1316;             // conv3x3_c16_kernel
        if (conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &smx)) return 1400 + mt;   // conv1x1_direct_kernel<MT>
        ResPlan plan;
        if (conv_res_plan(a, dtype, &plan)) {
            int pb, mpw, mph;
            size_t sm;
            const int mode = a.res_mode == PMOE_RES_ADD ? 1 : 0;      // (PMOE_RES_DBN returned above)
            return 1000 + plan.log_rb + (conv_res_dma_ok(a, plan, &pb, &mpw, &mph, &sm) ? (conv_res_pipe_ok(a) ? 200 + 20 * mode : 100) + (a.bias ? 10 : 0) : 0);
        }
        ConvArgs d = a;
        int mbd, pb;
        size_t sm;
        if (conv_dma_plan(d, dtype, &mbd, &sm, &pb)) return conv_dma_plan_code(a);       // conv3x3_dma_kernel<MF16, PROD> | conv3x3_dma_stream_kernel<MF16, NT>
        d = a;
        if (conv_dma_s2_plan(d, dtype, &mbd, &sm, &pb)) return 5207;       // conv3x3s2_dma_kernel
    }
    int mb = 0, cfg = 0;
    const int rc = dtype == PMOE_DT_BF16 ? launch_dtype<bf16>(c, nullptr, &mb, &cfg)
                 : dtype == PMOE_DT_F32 ? launch_dtype<float>(c, nullptr, &mb, &cfg) : PMOE_ERR_ARG;
    return rc ? rc : cfg + extra;
}

int conv_igemm_mblocks(const ConvArgs& a, int dtype) {
    if (a.w_fp8 && a.in_fp8) {
        ConvArgs d = a;
        int mb, pb;
        size_t sm;
        return conv_dma_f8_plan(d, dtype, &mb, &sm, &pb) ? mb : PMOE_ERR_UNSUPPORTED;
    }
    if (a.w_fp8) {
        int mb = 0;
        const int rc = dtype == PMOE_DT_BF16 ? launch_dtype<bf16, fp8>(a, nullptr, &mb) : PMOE_ERR_ARG;
        return rc ? rc : mb;
    }
    {
        int wpe, tx, tpe;
        // (a launch that asks for statistics never takes the skinny kernel: no gemm_skinny_ok test here, as for the resident kernel)
        if (conv_c16_plan(a, dtype, &wpe, &tx, &tpe)) return (a.N / a.ipe) * wpe;
        int slabs, mt;
        size_t smx;
        if (conv_c1x1_plan(a, dtype, &wpe, &tpe, &slabs, &mt, &smx)) return (a.N / a.ipe) * wpe;
    }
    ResPlan plan;
    if (conv_res_plan(a, dtype, &plan)) return (a.N / a.ipe) * plan.wgs_per_expert;
    {
        ConvArgs d = a;
        int mbd, pb;
        size_t sm;
        // (no gemm_skinny_ok test: this function sizes the STATISTICS rows, and a launch that asks for statistics never takes
        //  the skinny kernel -- plan and launch must see the same predicate)
        if (conv_dma_plan(d, dtype, &mbd, &sm, &pb)) return mbd;
        d = a;
        if (conv_dma_s2_plan(d, dtype, &mbd, &sm, &pb)) return mbd;
    }
    int mb = 0;
    int rc = (dtype == PMOE_DT_BF16) ? launch_dtype<bf16>(a, nullptr, &mb) : launch_dtype<float>(a, nullptr, &mb);
    return rc ? rc : mb;
}
