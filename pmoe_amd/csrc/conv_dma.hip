// Dense 3x3 stride-1 convolution (forward and data gradient) for the >= 128-channel layers, bf16, gfx950:
// ResNet layer2-4 (model/blocks/backbone.py:57-70 via torchvision's BasicBlock) -- the launches that dominate the step.
//
// Same implicit GEMM as conv_igemm.hip (workgroup = 256 output pixels x 128 output channels, 8 waves as 4 x 2, the input
// halo patch of a 64-channel chunk staged ONCE in LDS and read by all 9 taps at shifted addresses), but every byte reaches
// LDS by LDS-DMA (`buffer_load_dwordx4 ... lds`): no staging registers, no VALU, and the requests stay in flight across the
// per-tap barriers behind COUNTED `s_waitcnt vmcnt(N)`:
//   * weights  : ring of 4 tap tiles [128 couts][128 B]; the tile of tap t+2 is requested at tap t into the slot last read
//                at tap t-2 (two barriers earlier).  (A hand-pipelined variant -- ring lookahead 3, the next tap's first
//                fragments read under the current tap's last MFMAs, order pinned with sched_barrier -- measured 2-4 %
//                SLOWER in an interleaved A/B, tools/ab_conv.py: hipcc's own just-in-time read placement wins here);
//   * patches  : 2 buffers; the NEXT chunk's halo patch is requested piece by piece (1 KiB per wave-instruction) at taps
//                1..6 of the current chunk; pixels outside the image / beyond the expert's last image are zero-filled by the
//                buffer's range check (their offset is parked beyond num_records) -- the halo costs no instructions;
//   * swizzle  : LDS-DMA writes lane-linearly, so the XOR swizzle of the 16-byte chunks sits in the per-lane SOURCE
//                address and in the fragment reads (both sides or neither).  The swizzle key is the patch COLUMN
//                ((px >> 1) & 7), not the linear pixel index: a 16-lane group of a ds_read_b128 that spans two patch rows
//                (16-pixel-wide tiles, 16x16 maps) then still touches 16 different bank slots for every tap shift
//                (the linear key cost 15 % bank-conflict cycles there, profiles/r01_conv_lds_pmc.json).
// One barrier per tap, 16 MFMAs per wave between barriers, all 9 taps unrolled (static tap offsets), one workgroup per CU
// (150 KiB of LDS, 2 waves per SIMD).  Epilogue = that of conv_igemm.hip (bias / residual modes / activation / dropout /
// fused BatchNorm partial sums): staged through LDS as bf16 rows in ONE phase when nothing is added to the accumulator
// before it is rounded (forward + statistics, plain and ReLU'-masked data gradients: 8.0 k -> 5.9 k cycles per tile), as
// f32 in two halves otherwise.
// Measured and NOT adopted (interleaved A/B in one process, tools/ab_conv.py): persistent workgroups with the DMA stream
// running across tile boundaries (the next tile's operands land under the epilogue) -- 4-8 % slower: the epilogue's
// stores share the in-order vmcnt counter with the prefetch, the cold next-tile patch stalls the counted waits of the
// last chunk, and the kernel needs all 256 VGPRs; DMA requests issued between the MFMA groups, staggered between the two
// waves of a SIMD -- within 1 %.
#include "conv_common.h"
#include <stdlib.h>
#include "kernels.h"

namespace {

constexpr int RB = 128, LOG_RB = 7, CK = 64, BM = 256, BN = 128, NTHR = 512, VE = 8;
constexpr int WSLOT = BN * RB;            // 16 KiB per tap tile
constexpr int RING = 4;
typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int cswz(int x) { return (x >> 1) & 7; }

#define VMCNT(n) asm volatile("s_waitcnt vmcnt(" #n ")" ::: "memory")

// -DPMOE_STAMP (tools/stamp_conv.py only, never the product build): s_memtime laps of the four phases of a workgroup's life
// + its wall time (s_memrealtime, 100 MHz) land in a.stats instead of the BatchNorm partial sums.
#ifdef PMOE_STAMP
#define STAMP_INIT unsigned long long st_prev = __builtin_amdgcn_s_memtime(); const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime(); unsigned st_acc[4] = {0, 0, 0, 0};
#define LAP(i) { const unsigned long long st_t = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(st_t - st_prev); st_prev = st_t; }
#else
#define STAMP_INIT
#define LAP(i)
#endif

__global__ void __launch_bounds__(NTHR, 2) conv3x3_dma_kernel(const ConvArgs a, const int pbuf_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    STAMP_INIT
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    // flat grid, output-channel block fastest: the workgroups that read the same input patch are dispatch neighbours and,
    // through the XCD remap, share one XCD's L2
    const int nblk = a.CoutP / BN;
    const unsigned flat = xcd_remap(blockIdx.x, gridDim.x);
    const unsigned mb = flat / nblk;
    const int nb = (int)(flat % nblk);
    int t = (int)mb;
    const int px_t = t % a.tiles_x; t /= a.tiles_x;
    const int py_t = t % a.tiles_y; t /= a.tiles_y;
    const int ng = t % a.n_groups;
    const int e = t / a.n_groups;
    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int n0 = e * a.ipe + ng * a.TN, n_end = (e + 1) * a.ipe;
    const int oy0 = py_t * TH, ox0 = px_t * TW;
    const int cout0 = nb * BN;

    char* wring = smem + 2 * pbuf_bytes;

    // ---- buffer resources (wave-uniform: built from kernel arguments and block-derived scalars only)
    const bf16* inb = (const bf16*)a.in + (size_t)n0 * a.H * a.W * a.in_ld + a.in_coff;
    long long in_bytes = ((long long)(n_end - n0) * a.H * a.W * a.in_ld - a.in_coff) * 2;
    if (in_bytes > 0x7ff00000ll) in_bytes = 0x7ff00000ll;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, (int)in_bytes, 0x00020000);
    const bf16* wb = (const bf16*)a.w + ((size_t)e * a.CoutP + cout0) * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wb, (short)0, BN * 9 * a.Cin * 2, 0x00020000);

    // ---- per-lane source offsets of this wave's DMA pieces (loop invariant; the channel chunk and the tap travel in soffset)
    constexpr int OOB = 0x7ff80000;                    // beyond every num_records: the lane's 16 bytes arrive as zeros
    int pvoff[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pp = ((wave + 8 * i) << 3) + (lane >> 3);
        const int jj = lane & 7;
        int px = pp % PW;
        const int rowq = pp / PW;
        const int prow = rowq % PH, pn = rowq / PH;
        const int Y = oy0 - 1 + prow, X = ox0 - 1 + px;
        const bool ok = pp < NPIX && n0 + pn < n_end && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
        pvoff[i] = ok ? ((((pn * a.H + Y) * a.W + X) * a.in_ld) << 1) + ((jj ^ cswz(px)) << 4) : OOB;
    }
    int wvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = ((wave + 8 * i) << 3) + (lane >> 3);
        wvoff[i] = ((row * 9 * a.Cin) << 1) + (((lane & 7) ^ cswz(row)) << 4);
    }
    const int my_pieces = (NPIECE - wave + 7) >> 3;    // pieces wave, wave + 8, ... < NPIECE

    auto dma_patch = [&](int i, int buf, int c0) {     // piece `i` of this wave into patch buffer `buf`
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(smem + buf * pbuf_bytes + ((wave + 8 * i) << 10)), 16,
                                                 pvoff[i], c0 << 1, 0, 0);
    };
    auto dma_w = [&](int slot, int tap, int c0) {      // this wave's 2 pieces of the tap tile
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(wring + slot * WSLOT + ((wave + 8 * i) << 10)), 16,
                                                     wvoff[i], (tap * a.Cin + c0) << 1, 0, 0);
    };

    // ---- fragment addressing: B = pixels (columns of D), A = weights (rows of D)
    int pbase[2], pcol[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        pbase[mt] = ((pn * PH + my) * PW + mx) << LOG_RB;
        pcol[mt] = mx;
    }
    int aoff[2][4];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = wn * 64 + nt * 32 + l31;
            aoff[nt][ks] = row * RB + (((ks * 2 + h) ^ cswz(row)) << 4);
        }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int nchunks = a.Cin / CK;
    const int T = nchunks * 9;

    // ---- prologue: patch of chunk 0, weight tiles of taps 0 and 1
#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i < my_pieces) dma_patch(i, 0, 0);
    dma_w(0, 0, 0);
    dma_w(1, 1, 0);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * CK;
        const char* patch = smem + (ch & 1) * pbuf_bytes;
        const bool more = ch + 1 < nchunks;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int tt = ch * 9 + tap;
            // W(tt) has landed for this wave once at most {W(tt+1), the patch piece issued in the previous iteration}
            // are outstanding (vmcnt counts in issue order)
            const bool prev_piece = tap >= 2 && tap <= 7 && more && (tap - 2) < my_pieces;
            if (tt + 1 >= T) VMCNT(0);
            else if (prev_piece) VMCNT(3);
            else VMCNT(2);
            __builtin_amdgcn_s_barrier();                // every wave's share of tap tt (and of this chunk's patch) is in LDS
#ifdef PMOE_STAMP
            if (tt == 0) LAP(0)                          // prologue: descriptors, offsets, first patch + two weight tiles landed
#endif
            if (tap >= 1 && tap <= 6 && more && (tap - 1) < my_pieces) dma_patch(tap - 1, (ch + 1) & 1, c0 + CK);
            if (tt + 2 < T) {
                int ntap = tap + 2, nc0 = c0;
                if (ntap >= 9) { ntap -= 9; nc0 += CK; }
                dma_w((tt + 2) & (RING - 1), ntap, nc0);
            }
            const char* wt = wring + (tt & (RING - 1)) * WSLOT;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) << LOG_RB;
            int bsw[2];
            const char* bp[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bp[mt] = patch + pbase[mt] + tapoff;
                bsw[mt] = cswz(pcol[mt] + (tap % 3));
            }
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                v4i af[2], bfr[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) af[nt] = *reinterpret_cast<const v4i*>(wt + aoff[nt][ks]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    bfr[mt] = *reinterpret_cast<const v4i*>(bp[mt] + (((ks * 2 + h) ^ bsw[mt]) << 4));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                              __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                              acc[nt][mt], 0, 0, 0);
            }
        }
    }

    LAP(1)                                               // main loop
    // ---- epilogue: D[cout][pixel] -> LDS f32 [128 pixel rows][BN] per half (16-byte units XOR-swizzled by pixel) -> whole
    // 16-byte channel vectors per pixel, with bias / residual / activation / dropout / BatchNorm partial sums fused
    constexpr int UPR = BN / 4, BMH = BM / 2;
    float* stg = reinterpret_cast<float*>(smem);
    constexpr int CPO = BN / VE, PROWS = NTHR / CPO;
    const int cc = tid % CPO, pr = tid / CPO;
    const int cout = cout0 + cc * VE;
    const bool cvalid = cout < a.Cout;
    float bias[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) bias[i] = (a.bias && cvalid) ? a.bias[(size_t)e * a.CoutP + cout + i] : 0.f;
    float s1[VE], s2[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) s1[i] = s2[i] = 0.f;
    const float keep_scale = a.drop_p > 0.f ? 1.f / (1.f - a.drop_p) : 1.f;
    bf16* out = (bf16*)a.out;
    const bf16* res = (const bf16*)a.res;
    // PMOE_RES_DBN: this launch is the data gradient into a = relu(BatchNorm(z)); res = z.  The epilogue masks the gradient with
    // the recomputed ReLU decision and leaves the BatchNorm backward's two channel reductions in `stats` (s2 = sum g * xhat)
    const bool dbn = a.res_mode == PMOE_RES_DBN;
    float bmu[VE], bis[VE], bsc[VE], bsh[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) bmu[i] = bis[i] = bsc[i] = bsh[i] = 0.f;
    if (dbn && cvalid) {
        const int nset = a.N / a.bn_ipe;
        const float* b = a.bn + (size_t)((e * a.ipe) / a.bn_ipe) * a.Cout + cout;
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            bmu[i] = b[i]; bis[i] = b[(size_t)nset * a.Cout + i];
            bsc[i] = b[(size_t)2 * nset * a.Cout + i]; bsh[i] = b[(size_t)3 * nset * a.Cout + i];
        }
    }
    // fused tail of one 8-channel vector v of output pixel p of the tile
    auto finish = [&](int p, float* v) {
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        const int n = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
        if (!(cvalid && n < n_end && oy < a.Ho && ox < a.Wo)) return;
            const size_t opix = ((size_t)n * a.Ho + oy) * a.Wo + ox;
#pragma unroll
            for (int i = 0; i < VE; ++i) v[i] += bias[i];
            float xh[VE];
            if (a.res_mode) {
                float rv[VE];
                unpack16<bf16>(ldg16(res + opix * a.res_ld + a.res_coff + cout), rv);
                if (dbn) {
#pragma unroll
                    for (int i = 0; i < VE; ++i) {
                        const float d = rv[i] - bmu[i];
                        v[i] = (d * bsc[i] + bsh[i]) > 0.f ? v[i] : 0.f;
                        xh[i] = d * bis[i];
                    }
                } else
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
            const v4i pk = pack16<bf16>(v);
#ifndef PMOE_STAMP
            if (a.stats) {
                float rr[VE];
                unpack16<bf16>(pk, rr);
                if (dbn) {
#pragma unroll
                    for (int i = 0; i < VE; ++i) { s1[i] += rr[i]; s2[i] += rr[i] * xh[i]; }
                } else {
#pragma unroll
                    for (int i = 0; i < VE; ++i) { s1[i] += rr[i]; s2[i] += rr[i] * rr[i]; }
                }
            }
#endif
            stg16(out + opix * a.out_ld + a.out_coff + cout, pk);
    };
    // bf16 staging is exact when nothing is added to the accumulator before it is rounded (forward + statistics, plain and
    // ReLU'-masked data gradients): ONE phase, all 8 waves write their 64 x 64 sub-tiles as bf16 rows [256 px][256 B]
    // (16-byte chunks XOR-swizzled by pixel), one barrier, 8 vectors per thread -- half the LDS bytes and half the
    // barriers of the f32 path below
    const bool fast_epi = !a.bias && a.act == PMOE_ACT_NONE && a.drop_p == 0.f &&
                          (a.res_mode == PMOE_RES_NONE || a.res_mode == PMOE_RES_DRELU || a.res_mode == PMOE_RES_DBN);
    if (fast_epi) {
        __syncthreads();
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int p = wm * 64 + mt * 32 + l31;
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    bf16x4 v;
#pragma unroll
                    for (int i = 0; i < 4; ++i) v[i] = (bf16)acc[nt][mt][4 * g + i];
                    const int c16 = wn * 8 + nt * 4 + g;
                    *reinterpret_cast<bf16x4*>(smem + p * 256 + ((c16 ^ (p & 15)) << 4) + 8 * h) = v;
                }
            }
        __syncthreads();
#pragma unroll 2
        for (int u = 0; u < BM / PROWS; ++u) {
            const int ph = pr + u * PROWS;
            float v[VE];
            unpack16<bf16>(*reinterpret_cast<const v4i*>(smem + ph * 256 + ((cc ^ (ph & 15)) << 4)), v);
            finish(ph, v);
        }
    } else
    for (int half = 0; half < 2; ++half) {
        __syncthreads();                                 // main loop reads / the previous half's read-out are done
        if (wm / 2 == half) {
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
            float v[VE];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const int u = cc * 2 + k;
                const f32x4 tt = *reinterpret_cast<const f32x4*>(stg + ph * BN + ((u ^ (ph & (UPR - 1))) << 2));
#pragma unroll
                for (int i = 0; i < 4; ++i) v[4 * k + i] = tt[i];
            }
            finish(p, v);
        }
    }
#ifdef PMOE_STAMP
    LAP(2)                                               // epilogue (staging + fused read-out + stores)
    if (a.stats && lane == 0) {
        float* o = a.stats + ((size_t)blockIdx.x * 8 + wave) * 8;
        for (int i = 0; i < 3; ++i) o[i] = (float)st_acc[i];
        o[3] = (float)(unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
    }
    return;
#endif
    if (a.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);     // [8 waves][2][BN]
#pragma unroll
        for (int i = 0; i < VE; ++i) {
#pragma unroll
            for (int off = CPO; off < 64; off <<= 1) {
                s1[i] += __shfl_xor(s1[i], off);
                s2[i] += __shfl_xor(s2[i], off);
            }
        }
        if (lane < CPO) {
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
            for (int w = 0; w < 8; ++w) s += red[(w * 2 + which) * BN + c];
            if (cout0 + c < a.CoutP) a.stats[((size_t)mb * 2 + which) * a.CoutP + cout0 + c] = s;
        }
    }
}

}  // namespace

// Which launches take this kernel: bf16, dense 3x3 stride 1 pad 1 (forward, or the flipped-filter data gradient), whole
// 64-channel chunks, >= 128 output-channel rows, per-expert maps of >= 4096 pixels that tile into 16- or 32-pixel-wide
// strips.  PMOE_CONV_DMA=0 sends them back to conv_igemm_lite_kernel (A/B runs; read per launch, so that tools/ab_conv.py can
// interleave the two in one process).
bool conv_dma_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf) {
    const char* ev = getenv("PMOE_CONV_DMA");
    if ((ev && !atoi(ev)) || dtype != PMOE_DT_BF16 || a.w_fp8) return false;
    if (a.ks != 3 || a.kh != 3 || a.kw != 3 || a.use_tapmap || a.stride != 1 || a.pad != 1 || a.dilate || a.in_shared) return false;
    if (a.out_step != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.Cin % CK || a.CoutP % BN || a.Cout % 8 || a.N % a.ipe) return false;
    if ((long long)a.ipe * a.Ho * a.Wo < 4096) return false;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.Wo); if (lTW > 5) lTW = 5;
    if (lTW < 4) return false;
    int lTH = p2(a.Ho); if (lTH > 8 - lTW) lTH = 8 - lTW;
    const int TN = BM >> (lTW + lTH);
    const int NPIX = TN * ((1 << lTH) + 2) * ((1 << lTW) + 2);
    const int npiece = (NPIX + 7) / 8;
    if (npiece > 48) return false;                      // 6 pieces per wave
    const int pb = npiece * 1024;
    const size_t need = (size_t)2 * pb + RING * WSLOT;
    if (need > 160 * 1024) return false;
    // 32-bit source offsets inside one expert's images
    if ((long long)a.ipe * a.H * a.W * a.in_ld * 2 >= 0x7ff00000ll) return false;
    a.lTW = lTW; a.lTH = lTH; a.TN = TN;
    a.n_groups = (a.ipe + TN - 1) / TN;
    a.tiles_y = (a.Ho + (1 << lTH) - 1) >> lTH;
    a.tiles_x = (a.Wo + (1 << lTW) - 1) >> lTW;
    *mblocks = (a.N / a.ipe) * a.n_groups * a.tiles_y * a.tiles_x;
    *smem = need < (size_t)BM / 2 * BN * 4 ? (size_t)BM / 2 * BN * 4 : need;
    *pbuf = pb;
    return true;
}

int conv_dma_launch(ConvArgs a, hipStream_t st) {
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return PMOE_ERR_UNSUPPORTED;
    HIP_RET((ensure_dyn_lds<conv3x3_dma_kernel>(160 * 1024)));
    hipLaunchKernelGGL(conv3x3_dma_kernel, dim3(mblocks * (a.CoutP / BN)), dim3(NTHR), smem, st, a, pbuf);
    return (int)hipGetLastError();
}
