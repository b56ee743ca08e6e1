// Resident-weight 3x3 convolution (stride 1, pad 1, <= 64 input and <= 64 output channels, bf16):
// the stem convs, the four layer1 convs and all their data gradients -- 45 % of the network's conv
// FLOPs and its most HBM-heavy layers.
//
// Persistent workgroups (one per CU, 8 waves, all 160 KiB of LDS): the whole filter bank of the expert
// [9 taps][64 couts][Cin] stays in LDS for the life of the workgroup; the two 4-wave halves of the
// workgroup PING-PONG over consecutive 256-pixel tiles.  While half A runs the 144 MFMAs per wave of
// tile t out of its own halo patch, half B -- on the same SIMDs, in the MFMA shadow -- stages tile
// t-1's accumulators through its LDS region, writes them out as whole 16-byte channel vectors
// (+ residual add, + BatchNorm partial sums kept in registers for the whole workgroup lifetime),
// and commits the halo patch of tile t+1 whose global loads it issued one full tile earlier
// (issue early / write late).  Then the roles swap.  Three workgroup barriers per tile, no weight
// traffic, no prologue / epilogue bubble: the MFMA pipe of each SIMD always has one wave feeding it.
// (Since round 2 the 64 -> 64 layers run on conv3x3_resdma_kernel further down -- all 8 waves compute, halo patches by LDS-DMA,
// +16-25 % -- and the 16-channel stem convolution on conv_c16.hip; this kernel serves what those two decline and their A/B switches.)
#include "conv_common.h"
#include <stdlib.h>
#include "kernels.h"

template <int LOG_RB> __device__ __forceinline__ int rswz(int x) { return swz_chunk<LOG_RB, 0>(x); }

struct ResPlan {
    int lTW, lTH, TN, n_groups, tiles_y, tiles_x, tiles_per_expert, wgs_per_expert, log_rb;
    size_t smem;
};

// BIAS: per-expert (= per-image for the gate-folded layers) channel bias in the store path.  A separate instantiation,
// so the register allocation of the common (bias-free) kernel is untouched.
template <int LOG_RB, bool BIAS = false>
__global__ void __launch_bounds__(512) conv3x3_res_kernel(const ConvArgs a, const int tiles_per_expert,
                                                         const int wgs_per_expert, const int region_bytes) {
    constexpr int RB = 1 << LOG_RB, CPR = RB / 16, LOG_CPR = LOG_RB - 4;
    constexpr int KSUB = RB / 32;
    constexpr int VE = 8;                                // bf16 per 16 bytes
    constexpr int MAXV = 11;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = wave >> 2, gw = wave & 3, gtid = tid & 255;
    const int l31 = lane & 31, hh = lane >> 5;
    const int e = blockIdx.y, wg = blockIdx.x;
    const int t0 = (int)((long long)wg * tiles_per_expert / wgs_per_expert);
    const int t1 = (int)((long long)(wg + 1) * tiles_per_expert / wgs_per_expert);
    const int n = t1 - t0;

    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;

    char* Wl = smem;                                     // [9][64][RB]
    char* region = smem + 9 * 64 * RB + grp * region_bytes;   // this half's patch / staging area

    // per-expert channel bias: 64 floats in LDS behind the two regions (read in the store path; no registers held)
    float* lbias = reinterpret_cast<float*>(smem + 9 * 64 * RB + 2 * region_bytes);
    if (BIAS && tid < 64) lbias[tid] = a.bias[(size_t)e * a.CoutP + tid];
    // ---- resident filter bank of expert e
    {
        const bf16* wsrc = (const bf16*)a.w + (size_t)e * a.CoutP * 9 * a.Cin;
        for (int v = tid; v < 9 * 64 * CPR; v += 512) {
            const int j = v & (CPR - 1);
            const int row = (v >> LOG_CPR) & 63;
            const int tap = v >> (LOG_CPR + 6);
            const v4i val = ldg16(wsrc + ((size_t)row * 9 + tap) * a.Cin + j * VE);
            *reinterpret_cast<v4i*>(Wl + (tap * 64 + row) * RB + ((j ^ rswz<LOG_RB>(row)) << 4)) = val;
        }
    }

    auto geom = [&](int t, PatchGeom& g, int& n0, int& oy0, int& ox0) {
        int q = t0 + t;
        const int px = q % a.tiles_x; q /= a.tiles_x;
        const int py = q % a.tiles_y; q /= a.tiles_y;
        n0 = e * a.ipe + q * a.TN;
        oy0 = py * TH; ox0 = px * TW;
        g.n0 = n0; g.n_end = (e + 1) * a.ipe; g.e_first_img = e * a.ipe;
        g.Y0 = oy0 - 1; g.X0 = ox0 - 1; g.PH = PH; g.PW = PW; g.NPIX = NPIX;
        g.H = a.H; g.W = a.W; g.ld = a.in_ld; g.coff = a.in_coff; g.cmax = a.Cin;
        g.dilate = 0; g.shared = a.in_shared; g.step = 1;
    };

    const bf16* in = (const bf16*)a.in;
    PatchStage<bf16, LOG_RB, 256, MAXV> ps;
    auto issue_tile = [&](int t) {
        PatchGeom g; int n0, oy0, ox0;
        geom(t, g, n0, oy0, ox0);
        ps.issue(in, g, 0, gtid);
    };

    // per-lane pixel bases of the B operand (pixels gw*64 + mt*32 + l31 of the tile)
    int ppb[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = gw * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        ppb[mt] = (pn * PH + my) * PW + mx;
    }
    int arow_off[2][KSUB];                                // loop-invariant A (weight) fragment offsets
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int ks = 0; ks < KSUB; ++ks) {
            const int row = nt * 32 + l31;
            arow_off[nt][ks] = row * RB + (((ks * 2 + hh) ^ rswz<LOG_RB>(row)) << 4);
        }

    f32x16 acc[2][2];
    float s1[VE], s2[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) s1[i] = s2[i] = 0.f;
    const int cc = gtid & 7, pr = gtid >> 3;              // store phase: channel chunk / pixel lane
    const bool cvalid = cc * VE < a.Cout;
    bf16* out = (bf16*)a.out;
    const bf16* res = (const bf16*)a.res;

    // ---- prologue
    if (grp == 0 && n > 0) {
        issue_tile(0);
        ps.template commit<0>(region, NPIX, gtid);
    }
    __syncthreads();

    auto taps = [&](int tap_lo) {
        // opaque copies: keeps LICM from hoisting all 72 per-(tap,k-step) operand addresses out of the tile
        // loop (they would be spilled to scratch and reloaded between the MFMAs)
        int pb[2] = {ppb[0], ppb[1]};
        asm volatile("" : "+v"(pb[0]), "+v"(pb[1]));
#pragma unroll
        for (int tp = 0; tp < 3; ++tp) {
            const int tap = tap_lo + tp;
            const int tapoff = (tap / 3) * PW + (tap % 3);
            const char* wt = Wl + tap * 64 * RB;
            int pp[2], fp[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                pp[mt] = pb[mt] + tapoff;
                fp[mt] = rswz<LOG_RB>(pp[mt]);
            }
#pragma unroll
            for (int ks = 0; ks < KSUB; ++ks) {
                v4i af[2], bfr[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) af[nt] = *reinterpret_cast<const v4i*>(wt + arow_off[nt][ks]);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
                    bfr[mt] = *reinterpret_cast<const v4i*>(region + (pp[mt] << LOG_RB) + (((ks * 2 + hh) ^ fp[mt]) << 4));
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                              __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                              acc[nt][mt], 0, 0, 0);
            }
        }
    };

    for (int h = 0; h <= n; ++h) {
        if (grp == (h & 1)) {
            // ================= MFMA role: tile h =================
            const bool live = h < n;
            if (live) {
                if (BIAS) {
                    // accumulators start from the channel bias (acc[nt][*][4g+i] is cout (nt*4+g)*8 + hh*4 + i):
                    // the store path stays that of the bias-free kernel
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + (i * 4 + g) * 8 + hh * 4);
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int k = 0; k < 4; ++k) acc[i][j][4 * g + k] = b[k];
                        }
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i)
#pragma unroll
                        for (int j = 0; j < 2; ++j)
#pragma unroll
                            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
                }
                taps(0);
            }
            __syncthreads();
            if (live) taps(3);
            __syncthreads();
            if (live) taps(6);
            __syncthreads();
        } else {
            // ================= I/O role: write out tile h-1, bring in tile h+1 =================
            // Order matters for the vmcnt queue (loads and stores share one in-order counter on CDNA): the
            // patch loads of tile h+1 are issued first and this tile's global stores LAST, after the patch
            // commit, so the commit's wait never covers stores issued moments ago.
            const bool have = h >= 1, next = h + 1 < n;
            PatchGeom gg; int n0 = 0, oy0 = 0, ox0 = 0;
            v4i rb[8];
            if (next) issue_tile(h + 1);                   // ~1.5 us of staging / barriers before its commit
            if (have) {
                geom(h - 1, gg, n0, oy0, ox0);
                // D[cout][pixel] -> bf16 [256 px][64 cout] staging, 16-byte chunks XOR-swizzled by pixel
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        const int p = gw * 64 + mt * 32 + l31;
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            bf16x4 v;
#pragma unroll
                            for (int i = 0; i < 4; ++i) v[i] = (bf16)acc[nt][mt][4 * g + i];
                            *reinterpret_cast<bf16x4*>(region + p * 128 + (((nt * 4 + g) ^ (p & 7)) << 4) + 8 * hh) = v;
                        }
                    }
            }
            __syncthreads();
            if (have) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = pr + 32 * u;
                    rb[u] = *reinterpret_cast<const v4i*>(region + p * 128 + ((cc ^ (p & 7)) << 4));
                }
            }
            __syncthreads();
            if (next) ps.template commit<0>(region, NPIX, gtid);
            if (have) {
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = pr + 32 * u;
                    const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                    const int nn = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
                    if (cvalid && nn < gg.n_end && oy < a.Ho && ox < a.Wo) {
                        const size_t opix = ((size_t)nn * a.Ho + oy) * a.Wo + ox;
                        v4i pk = rb[u];
                        if (a.res_mode == PMOE_RES_ADD) {
                            float v[VE], rv[VE];
                            unpack16<bf16>(rb[u], v);
                            unpack16<bf16>(ldg16(res + opix * a.res_ld + a.res_coff + cc * VE), rv);
#pragma unroll
                            for (int i = 0; i < VE; ++i) v[i] += rv[i];
                            pk = pack16<bf16>(v);
                        }
                        if (a.stats) {
                            float rr[VE];
                            unpack16<bf16>(pk, rr);
#pragma unroll
                            for (int i = 0; i < VE; ++i) { s1[i] += rr[i]; s2[i] += rr[i] * rr[i]; }
                        }
                        stg16(out + opix * a.out_ld + a.out_coff + cc * VE, pk);
                    }
                }
            }
            __syncthreads();
        }
    }

    if (a.stats) {
        // one [2][CoutP] partial row per workgroup: lanes sharing a channel chunk combine by xor-shuffle,
        // the 8 waves through LDS (the filter bank is dead by now)
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [8 waves][2][64]
#pragma unroll
        for (int i = 0; i < VE; ++i) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                s1[i] += __shfl_xor(s1[i], off);
                s2[i] += __shfl_xor(s2[i], off);
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                red[(wave * 2 + 0) * 64 + cc * VE + i] = s1[i];
                red[(wave * 2 + 1) * 64 + cc * VE + i] = s2[i];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[(w * 2 + which) * 64 + c];
            a.stats[(((size_t)e * wgs_per_expert + wg) * 2 + which) * a.CoutP + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round-2 variant for the 64 -> 64 layers (stem conv2, layer1; forward and data gradient): the filter bank stays RESIDENT
// in LDS as above, but ALL 8 waves compute every tile (two waves per SIMD cover each other's LDS latency -- the ping-pong
// kernel above has ONE MFMA-role wave per SIMD and hipcc sinks every fragment read to its use: ~9 k cycles for 4.6 k
// cycles of MFMA per tile), and the halo patches arrive by LDS-DMA (`buffer_load_dwordx4 ... lds`, zero fill through the
// buffer range check, XOR swizzle on the source address keyed by the patch column) into two buffers: the patch of tile
// i+1 is requested at the start of tile i and has the whole tile to land.  No barrier inside a tile's 9 taps; the
// epilogue stages the tile as bf16 rows in the patch buffer it has just finished with (nothing is added before the
// rounding that the ping-pong kernel did not also add after it: residual add on the rounded value, bias in the
// accumulator), waits for the prefetched patch BEFORE issuing its stores (one in-order vmcnt counter), and carries the
// BatchNorm partial sums in registers across tiles.
// Wave tile = 64 pixels x 32 output channels (4 x 2 waves): 3 ds_read_b128 per 2 MFMAs, ~75 % LDS-array load.
// -DPMOE_STAMP (tools/stamp_conv.py only, never the product build): s_memtime laps of the five phases of a tile + the workgroup's
// wall time (s_memrealtime, 100 MHz) land in a.stats instead of the BatchNorm partial sums.
#ifdef PMOE_STAMP
#define RSTAMP_INIT unsigned long long st_prev = __builtin_amdgcn_s_memtime(); const unsigned long long st_r0 = __builtin_amdgcn_s_memrealtime(); unsigned st_acc[5] = {0, 0, 0, 0, 0};
#define RLAP(i) { const unsigned long long st_t = __builtin_amdgcn_s_memtime(); st_acc[i] += (unsigned)(st_t - st_prev); st_prev = st_t; }
#else
#define RSTAMP_INIT
#define RLAP(i)
#endif

template <bool BIAS>
__global__ void __launch_bounds__(512, 2) conv3x3_resdma_kernel(const ConvArgs a, const int tiles_per_expert,
                                                                const int wgs_per_expert, const int pbuf_bytes,
                                                                const int magic_pw, const int magic_ph) {
    constexpr int RB = 128, LOG_RBK = 7, VE = 8;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int e = blockIdx.y, wg = blockIdx.x;
    const int t0 = (int)((long long)wg * tiles_per_expert / wgs_per_expert);
    const int t1 = (int)((long long)(wg + 1) * tiles_per_expert / wgs_per_expert);
    const int ntile = t1 - t0;

    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int my_pieces = (NPIECE - wave + 7) >> 3;

    char* Wl = smem;                                     // [9][64][128 B], chunks swizzled by row
    char* pbuf = smem + 9 * 64 * RB;                     // 2 patch buffers
    float* lbias = reinterpret_cast<float*>(pbuf + 2 * pbuf_bytes);
    if (BIAS && tid < 64) lbias[tid] = a.bias[(size_t)e * a.CoutP + tid];
    // PMOE_RES_DBN (this launch is the data gradient into a = relu(BatchNorm(z)), res = z): the BatchNorm's coefficients
    // [mean | invstd | gamma*invstd | beta][64] stay in LDS; the read-out masks the gradient with the recomputed ReLU decision
    // and accumulates the BatchNorm backward's two channel reductions in s1 / s2 (s2 = sum g * xhat)
    const bool dbn = a.res_mode == PMOE_RES_DBN;
    float* lbn = lbias + 64;
    if (dbn && tid < 256) {
        const int nset = a.N / a.bn_ipe;
        lbn[tid] = (tid & 63) < a.Cout ? a.bn[((size_t)(tid >> 6) * nset + (e * a.ipe) / a.bn_ipe) * a.Cout + (tid & 63)] : 0.f;
    }

    const bf16* inb = (const bf16*)a.in + (size_t)e * a.ipe * a.H * a.W * a.in_ld + a.in_coff;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)inb, (short)0, (int)(((long long)a.ipe * a.H * a.W * a.in_ld - a.in_coff) * 2), 0x00020000);
    const bf16* wsrc = (const bf16*)a.w + (size_t)e * a.CoutP * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, (short)0, 64 * 9 * 64 * 2, 0x00020000);
    constexpr int OOB = 0x7ff80000;

    // resident filter bank: 72 pieces of 8 rows (piece = tap * 8 + row block), 9 per wave
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int piece = wave + 8 * i;
        const int tap = piece >> 3, row = ((piece & 7) << 3) + (lane >> 3);
        const int voff = ((row * 9 + tap) * 64 * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(Wl + (piece << 10)), 16, voff, 0, 0, 0);
    }

    auto tile_geo = [&](int t, int& n0, int& oy0, int& ox0) {
        int qq = t0 + t;
        const int px = qq % a.tiles_x; qq /= a.tiles_x;
        const int py = qq % a.tiles_y; qq /= a.tiles_y;
        n0 = qq * a.TN;                                  // image index inside the expert
        oy0 = py * TH; ox0 = px * TW;
    };
    auto issue_patch = [&](int t, int buf) {
        int n0, oy0, ox0;
        tile_geo(t, n0, oy0, ox0);
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < my_pieces) {
                const int pp = ((wave + 8 * i) << 3) + (lane >> 3);
                const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
                const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
                const int n = n0 + pn, Y = oy0 - 1 + prow, X = ox0 - 1 + px;
                const bool ok = pp < NPIX && n < a.ipe && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
                const int voff = ok ? ((((n * a.H + Y) * a.W + X) * a.in_ld) << 1) + (((lane & 7) ^ ((px >> 1) & 7)) << 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(pbuf + buf * pbuf_bytes + ((wave + 8 * i) << 10)), 16,
                                                         voff, 0, 0, 0);
            }
        }
    };

    // fragment addressing: B = pixels wm*64 + mt*32 + l31 (columns of D), A = couts wn*32 + l31 (rows of D)
    int pbase[2], pcol[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        pbase[mt] = ((pn * PH + my) * PW + mx) << LOG_RBK;
        pcol[mt] = mx;
    }
    int aoff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int row = wn * 32 + l31;
        aoff[ks] = row * RB + (((ks * 2 + hh) ^ ((row >> 1) & 7)) << 4);
    }

    float s1[VE], s2[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) s1[i] = s2[i] = 0.f;
    const int cc = tid & 7, pr = tid >> 3;               // read-out: 8 channel chunks x 64 pixel rows of threads
    const bool cvalid = cc * VE < a.Cout;
    bf16* out = (bf16*)a.out;
    const bf16* res = (const bf16*)a.res;
    const bool res_pref = (a.res_mode == PMOE_RES_ADD || dbn) && a.prefetch;      // (a.prefetch: launcher, PMOE_RES_PREFETCH=0 = A/B)

    if (ntile > 0) issue_patch(0, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // filter bank + first patch (DMA), the bias row (ds_write)
    __builtin_amdgcn_s_barrier();
    // PMOE_RES_DBN: this thread's 8 channels of the BatchNorm coefficients, in registers for all of its tiles
    float kmu[VE], kis[VE], ksc[VE], ksh[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) kmu[i] = kis[i] = ksc[i] = ksh[i] = 0.f;
    if (dbn) {
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            kmu[i] = lbn[cc * VE + i]; kis[i] = lbn[64 + cc * VE + i]; ksc[i] = lbn[128 + cc * VE + i]; ksh[i] = lbn[192 + cc * VE + i];
        }
    }

    RSTAMP_INIT
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        if (t + 1 < ntile) issue_patch(t + 1, buf ^ 1);  // its buffer was released by the barrier that ended tile t-1
        // residual add (a data gradient accumulating into the gradient another consumer left): the four 16-byte pieces this thread
        // will add in the read-out are requested NOW and arrive under the MFMAs -- loaded in the epilogue they cost +50 % on the
        // layer1 launches (0.31 -> 0.47 ms), every wave waiting on its own global loads between two barriers
        v4i rpre[4];
        if (res_pref) {
            int n0, oy0, ox0;
            tile_geo(t, n0, oy0, ox0);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int p = pr + 64 * u;
                const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                const int nn = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
                rpre[u] = v4i{0, 0, 0, 0};
                if (cvalid && nn < a.ipe && oy < a.Ho && ox < a.Wo)
                    rpre[u] = ldg16(res + ((((size_t)e * a.ipe + nn) * a.Ho + oy) * a.Wo + ox) * a.res_ld + a.res_coff + cc * VE);
            }
        }
        const char* patch = pbuf + buf * pbuf_bytes;
        f32x16 acc[2];
        if (BIAS) {
            // accumulators start from the channel bias (acc[mt][4g+i] is cout wn*32 + 8g + 4hh + i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + wn * 32 + g * 8 + hh * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[mt][4 * g + k] = b[k];
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[mt][k] = 0.f;
        }
        RLAP(0)                                          // patch requests, residual prefetch, accumulator init
        auto read_step = [&](int st, v4i& af, v4i* bfr) {     // fragments of step st = tap * 4 + ks
            const int tap = st >> 2, ks = st & 3;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) << LOG_RBK;
            af = *reinterpret_cast<const v4i*>(Wl + tap * 64 * RB + aoff[ks]);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                bfr[mt] = *reinterpret_cast<const v4i*>(patch + pbase[mt] + tapoff +
                                                        (((ks * 2 + hh) ^ (((pcol[mt] + (tap % 3)) >> 1) & 7)) << 4));
        };
        // (measured and dropped, interleaved A/B: a one-step read-ahead pinned with sched_barrier -- within 1 %; keeping 12 / 20
        //  / all 36 weight fragments of the wave in registers instead of re-reading them from LDS -- 0..-3 % / x0.5 (spills): the
        //  loop is not LDS-bandwidth bound)
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            v4i af, bfr[2];
            read_step(st, af, bfr);
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                  acc[mt], 0, 0, 0);
        }
        // ---- epilogue: bf16 rows [256 px][128 B] (16-byte chunks XOR-swizzled by pixel) in this tile's patch buffer
        char* stage = pbuf + buf * pbuf_bytes;
        RLAP(1)                                          // 36 steps of fragment reads + MFMAs
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // every wave is done reading the patch
        RLAP(2)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            const int p = wm * 64 + mt * 32 + l31;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bf16x4 v;
#pragma unroll
                for (int i = 0; i < 4; ++i) v[i] = (bf16)acc[mt][4 * g + i];
                *reinterpret_cast<bf16x4*>(stage + p * 128 + (((wn * 4 + g) ^ (p & 7)) << 4) + 8 * hh) = v;
            }
        }
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // + the next tile's patch, AHEAD of this tile's stores
        __builtin_amdgcn_s_barrier();
        RLAP(3)                                          // staging writes, wait for the prefetched patch / earlier stores, barrier
        int n0, oy0, ox0;
        tile_geo(t, n0, oy0, ox0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = pr + 64 * u;
            const v4i raw = *reinterpret_cast<const v4i*>(stage + p * 128 + ((cc ^ (p & 7)) << 4));
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            const int nn = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
            if (cvalid && nn < a.ipe && oy < a.Ho && ox < a.Wo) {
                const size_t opix = (((size_t)e * a.ipe + nn) * a.Ho + oy) * a.Wo + ox;
                v4i pk = raw;
                if (dbn) {
                    float v[VE], rv[VE];
                    unpack16<bf16>(raw, v);
                    unpack16<bf16>(res_pref ? rpre[u] : ldg16(res + opix * a.res_ld + a.res_coff + cc * VE), rv);
#pragma unroll
                    for (int i = 0; i < VE; ++i) {
                        const float d = rv[i] - kmu[i];
                        // the value the apply pass will read is the ROUNDED masked gradient: the sums use it too
                        const float gq = (d * ksc[i] + ksh[i]) > 0.f ? v[i] : 0.f;
                        v[i] = gq;
                        s1[i] += gq;
                        s2[i] += gq * (d * kis[i]);
                    }
                    pk = pack16<bf16>(v);
                } else {
                if (a.res_mode == PMOE_RES_ADD) {
                    float v[VE], rv[VE];
                    unpack16<bf16>(raw, v);
                    unpack16<bf16>(res_pref ? rpre[u] : ldg16(res + opix * a.res_ld + a.res_coff + cc * VE), rv);
#pragma unroll
                    for (int i = 0; i < VE; ++i) v[i] += rv[i];
                    pk = pack16<bf16>(v);
                }
                if (a.stats) {
                    float rr[VE];
                    unpack16<bf16>(pk, rr);
#pragma unroll
                    for (int i = 0; i < VE; ++i) { s1[i] += rr[i]; s2[i] += rr[i] * rr[i]; }
                }
                }
                stg16(out + opix * a.out_ld + a.out_coff + cc * VE, pk);
            }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                    // staging read out: the buffer may receive the patch of tile t+2
        RLAP(4)                                          // read-out (+ residual / BatchNorm reductions), stores, barrier
    }
#ifdef PMOE_STAMP
    if (a.stats && lane == 0) {
        float* o = a.stats + (((size_t)e * wgs_per_expert + wg) * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = (float)st_acc[i];
        o[5] = (float)(unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
        o[6] = (float)ntile;
    }
    return;
#endif

    if (a.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [8 waves][2][64]  (the filter bank is dead by now)
#pragma unroll
        for (int i = 0; i < VE; ++i) {
#pragma unroll
            for (int off = 8; off < 64; off <<= 1) {
                s1[i] += __shfl_xor(s1[i], off);
                s2[i] += __shfl_xor(s2[i], off);
            }
        }
        if (lane < 8) {
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                red[(wave * 2 + 0) * 64 + cc * VE + i] = s1[i];
                red[(wave * 2 + 1) * 64 + cc * VE + i] = s2[i];
            }
        }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 8; ++w) s += red[(w * 2 + which) * 64 + c];
            a.stats[(((size_t)e * wgs_per_expert + wg) * 2 + which) * a.CoutP + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round-3 variant of conv3x3_resdma_kernel: the same resident filter bank, LDS-DMA halo patches and 36-step MFMA loop, but the
// epilogue of tile t no longer stops the matrix pipe.  Cycle stamps of the kernel above (tools/stamp_conv.py, profiles/
// r03_resdma_stamps.log): 9450 cycles per tile of which 4460 in the MFMA loop -- 1690 in the next patch's address arithmetic,
// 860 + 770 + 1670 in the three barriers, the LDS staging and the read-out, all of it VALU / LDS work during which no wave of
// the workgroup issues an MFMA.  Here
//   * the accumulators of tile t are rounded to bf16 IN REGISTERS, two v_permlane32_swap per 8-byte piece turn the MFMA layout
//     (4 consecutive channels per lane half) into whole 16-byte channel vectors, and the stores + BatchNorm sums / residual add /
//     masked BatchNorm-backward reductions of tile t are issued BETWEEN THE MFMAs OF TILE t+1 (straight-line code: invalid
//     pixels park their buffer offset out of range instead of branching, so the scheduler may interleave freely);
//   * no LDS staging, ONE barrier per tile (patch t is free for the request of tile t+2);
//   * the per-lane geometry of the 6 patch pieces and 2 output pixels is tile-invariant and computed once; the tile walk is
//     incremental (no divisions in the loop).
// MODE: 0 = plain (+ BatchNorm partial sums of the stored values when a.stats), 1 = PMOE_RES_ADD, 2 = PMOE_RES_DBN,
// 3 = PMOE_RES_INBN (round 4, BASELINE config 4): mode 0 whose INPUT is the pre-activation z of a BatchNorm + ReLU -- the frozen
// U-Nets' `conv -> BatchNorm -> ReLU -> conv` pairs (model/blocks/unet.py:14-24) in train mode, where nothing is saved for a
// backward pass and z has this one consumer: the pmoe_bn_apply pass between the two convolutions (read z, write a: 1.07 GB at
// 256 x 256 x 64 channels x 64 images, 0.2 ms, 20 of them per step) disappears.  Each wave turns the patch pieces IT requested
// into bf16(max((z - mean) * scale + shift, 0)) in place -- the arithmetic and rounding of bn_apply_kernel, so the MFMA operand
// is bit-identical to the tensor that pass would have written -- between the MFMAs of the tile BEFORE the one that reads them
// (steps 20..31: the requests were issued at the start of the tile, ahead of the read-out's stores in the wave's in-order
// vmcnt stream, so `vmcnt(4)` = they have landed); a lane owns channel group lane & 7 of its pixel row, i.e. the physical 16-byte
// slot (lane & 7) ^ swizzle(column), so its 24 coefficients are the same for every piece.  Pixels outside the image are not
// touched: the DMA's zero fill stays zero (relu(bn(0)) is not 0).
// Measured (tools/ab_inbn.py, 64 images of 256 x 256): plain launch 0.325 ms, this mode 0.396 ms, the pass it replaces 0.206 ms: a
// pair costs 0.40 instead of 0.53 ms.  The +0.07 ms is not the vmcnt(4) wait (a build with the wait and no arithmetic: 0.328 ms),
// not the coefficient reads (kept in registers for the kernel's lifetime: 0.390), not the VALU count (-25 %: 0.399) and not the
// load -> use distance (three steps instead of one: 0.393); grouping the six stores in two bursts made it 0.57 ms -- the MFMA block
// of this kernel has no idle issue slots to give (it runs at 93 % of the matrix pipe's time), so whatever is added to it is paid.
template <bool BIAS, int MODE, bool RZ_LATE = true>
__global__ void __launch_bounds__(512, 2) conv3x3_respipe_kernel(const ConvArgs a, const int tiles_per_expert,
                                                                 const int wgs_per_expert, const int pbuf_bytes,
                                                                 const int magic_pw, const int magic_ph) {
    constexpr int RB = 128, LOG_RBK = 7;
    typedef __attribute__((address_space(3))) void lds_void;
    typedef unsigned v4u __attribute__((ext_vector_type(4)));
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, hh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int e = blockIdx.y, wg = blockIdx.x;
    const int t0 = (int)((long long)wg * tiles_per_expert / wgs_per_expert);
    const int t1 = (int)((long long)(wg + 1) * tiles_per_expert / wgs_per_expert);
    const int ntile = t1 - t0;

    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int my_pieces = (NPIECE - wave + 7) >> 3;

    char* Wl = smem;                                     // [9][64][128 B], chunks swizzled by row
    char* pbuf = smem + 9 * 64 * RB;                     // 2 patch buffers
    float* lbias = reinterpret_cast<float*>(pbuf + 2 * pbuf_bytes);
    if (BIAS && tid < 64) lbias[tid] = a.bias[(size_t)e * a.CoutP + tid];
    float* lbn = lbias + 64;                             // MODE 2: [mean | invstd | gamma*invstd | beta][64]
    if (MODE == 2 && tid < 256) {
        const int nset = a.N / a.bn_ipe;
        lbn[tid] = (tid & 63) < a.Cout ? a.bn[((size_t)(tid >> 6) * nset + (e * a.ipe) / a.bn_ipe) * a.Cout + (tid & 63)] : 0.f;
    }
    constexpr bool HASRES = MODE == 1 || MODE == 2, INBN = MODE == 3;
    if (INBN && tid < 256) {                             // the INPUT's BatchNorm: same four rows over the 64 input channels
        const int nset = a.N / a.bn_ipe;
        lbn[tid] = a.bn[((size_t)(tid >> 6) * nset + (e * a.ipe) / a.bn_ipe) * a.Cin + (tid & 63)];
    }

    constexpr int OOB = 0x7ff80000;
    const bf16* inb = (const bf16*)a.in + (size_t)e * a.ipe * a.H * a.W * a.in_ld + a.in_coff;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)inb, (short)0, (int)(((long long)a.ipe * a.H * a.W * a.in_ld - a.in_coff) * 2), 0x00020000);
    const bf16* wsrc = (const bf16*)a.w + (size_t)e * a.CoutP * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wsrc, (short)0, 64 * 9 * 64 * 2, 0x00020000);
    bf16* outb = (bf16*)a.out + (size_t)e * a.ipe * a.Ho * a.Wo * a.out_ld + a.out_coff;
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)outb, (short)0, (int)(((long long)a.ipe * a.Ho * a.Wo * a.out_ld - a.out_coff) * 2), 0x00020000);
    const bf16* resb = HASRES ? (const bf16*)a.res + (size_t)e * a.ipe * a.Ho * a.Wo * a.res_ld + a.res_coff : (const bf16*)a.in;
    const __amdgpu_buffer_rsrc_t rs_res = __builtin_amdgcn_make_buffer_rsrc(
        (void*)resb, (short)0, HASRES ? (int)(((long long)a.ipe * a.Ho * a.Wo * a.res_ld - a.res_coff) * 2) : 0, 0x00020000);

    // resident filter bank: 72 pieces of 8 rows (piece = tap * 8 + row block), 9 per wave
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        const int piece = wave + 8 * i;
        const int tap = piece >> 3, row = ((piece & 7) << 3) + (lane >> 3);
        const int voff = ((row * 9 + tap) * 64 * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(Wl + (piece << 10)), 16, voff, 0, 0, 0);
    }

    // ---- tile-invariant geometry of this lane's patch pieces: byte offset relative to the patch origin (+ source swizzle) and
    // the packed (image, row, column) inside the patch
    int prel[6], pgeo[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pp = ((wave + 8 * i) << 3) + (lane >> 3);
        const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
        const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
        prel[i] = ((((pn * a.H + prow) * a.W + px) * a.in_ld) << 1) + (((lane & 7) ^ ((px >> 1) & 7)) << 4);
        pgeo[i] = pp < NPIX ? (pn << 20) | (prow << 10) | px : (0x7ff << 20);        // beyond the patch: never in range
    }
    // the tile walk: (column tile, row tile, image group) of the current tile, advanced by one per iteration
    struct Walk { int px, py, q; };
    Walk cur;
    {
        int qq = t0;
        cur.px = qq % a.tiles_x; qq /= a.tiles_x;
        cur.py = qq % a.tiles_y; cur.q = qq / a.tiles_y;
    }
    auto advance = [&](Walk w) {
        if (++w.px == a.tiles_x) { w.px = 0; if (++w.py == a.tiles_y) { w.py = 0; ++w.q; } }
        return w;
    };
    auto issue_patch = [&](const Walk& w, int buf) {
        const int n0 = w.q * a.TN, Y0 = w.py * TH - 1, X0 = w.px * TW - 1;
        const int tbase = (((n0 * a.H + Y0) * a.W + X0) * a.in_ld) << 1;
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < my_pieces) {
                const int n = n0 + (pgeo[i] >> 20), Y = Y0 + ((pgeo[i] >> 10) & 0x3ff), X = X0 + (pgeo[i] & 0x3ff);
                const bool ok = n < a.ipe && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(pbuf + buf * pbuf_bytes + ((wave + 8 * i) << 10)), 16,
                                                         ok ? tbase + prel[i] : OOB, 0, 0, 0);
            }
        }
    };

    // fragment addressing: B = pixels wm*64 + mt*32 + l31 (columns of D), A = couts wn*32 + l31 (rows of D)
    int pbase[2], pcol[2];
    // read-out: this lane owns, for its two pixels (mt), the channel vectors [c0 + 16 k, + 8), k = 0, 1
    unsigned ovo[2], rvo[2];
    int ogeo[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        pbase[mt] = ((pn * PH + my) * PW + mx) << LOG_RBK;
        pcol[mt] = mx;
        const int orel = (pn * a.Ho + my) * a.Wo + mx;
        ogeo[mt] = (pn << 20) | (my << 10) | mx;
        ovo[mt] = (unsigned)(orel * a.out_ld + wn * 32 + 8 * hh) * 2u;
        rvo[mt] = (unsigned)(orel * a.res_ld + wn * 32 + 8 * hh) * 2u;
    }
    const int c0 = wn * 32 + 8 * hh;
    const bool cval[2] = {c0 < a.Cout, c0 + 16 < a.Cout};
    int aoff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        const int row = wn * 32 + l31;
        aoff[ks] = row * RB + (((ks * 2 + hh) ^ ((row >> 1) & 7)) << 4);
    }

    f32x2 s1[2][4], s2[2][4];                            // sums of this lane's 2 x 8 channels, all of its pixels and tiles
#pragma unroll
    for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int q = 0; q < 4; ++q) s1[k][q] = s2[k][q] = f32x2{0.f, 0.f};

    // PMOE_RES_INBN: piece i of this wave's share of the patch of tile `w` in buffer `buf`, BatchNorm + ReLU in place
    auto unpack2 = [](unsigned w) { return f32x2{__builtin_bit_cast(float, w << 16), __builtin_bit_cast(float, w & 0xffff0000u)}; };
    auto pack2 = [](f32x2 v) {
        typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
        return __builtin_bit_cast(unsigned, bf16x2{(bf16)v[0], (bf16)v[1]});
    };
    // (straight-line code, in two halves so that the scheduler can put MFMAs between the LDS read and its use: a branch here would
    //  park the wave -- and its SIMD partner, which is in the same phase -- for the read's round trip)
    f32x2 cmu[4], csc[4], csh[4];                        // this lane's channel group: 8 x (mean, gamma * invstd, beta)
    auto xform_coef = [&]() {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            cmu[q] = *reinterpret_cast<const f32x2*>(lbn + (lane & 7) * 8 + 2 * q);
            csc[q] = *reinterpret_cast<const f32x2*>(lbn + 128 + (lane & 7) * 8 + 2 * q);
            csh[q] = *reinterpret_cast<const f32x2*>(lbn + 192 + (lane & 7) * 8 + 2 * q);
        }
    };
    // (every wave has at least five pieces -- a 256-pixel tile's patch has >= 324 pixels = 41 pieces -- and a wave without a sixth
    //  rewrites its piece 0 unchanged: no wave-level branch in the MFMA block; the lanes of pixels outside the image skip the store)
    auto xform_addr = [&](int buf, const int i) {
        const int ii = i < 5 || i < my_pieces ? i : 0;
        const int px = (i < 5 || i < my_pieces ? pgeo[i] : pgeo[0]) & 0x3ff;
        return pbuf + buf * pbuf_bytes + ((wave + 8 * ii) << 10) + ((lane >> 3) << 7) + (((lane & 7) ^ ((px >> 1) & 7)) << 4);
    };
    auto xform_load = [&](int buf, const int i) { return *reinterpret_cast<const v4u*>(xform_addr(buf, i)); };
    auto xform_store = [&](const Walk& w, int buf, const int i, const v4u z) {
        const int n0 = w.q * a.TN, Y0 = w.py * TH - 1, X0 = w.px * TW - 1;
        const int n = n0 + (pgeo[i] >> 20), Y = Y0 + ((pgeo[i] >> 10) & 0x3ff), X = X0 + (pgeo[i] & 0x3ff);
        const bool ok = n < a.ipe && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
        typedef short s16x2 __attribute__((ext_vector_type(2)));
        v4u o;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const f32x2 v = __builtin_elementwise_fma(unpack2(z[q]) - cmu[q], csc[q], csh[q]);      // (bn_apply_kernel: centred, one fma)
            // ReLU on the rounded pair: a negative bf16 is a negative int16 (fmaxf on the f32 values gives the same bits except
            // that -0 becomes +0 here)
            const unsigned pk = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pack2(v)), s16x2{0, 0}));
            o[q] = i < 5 || i < my_pieces ? pk : z[q];
        }
        if (ok || !(i < 5 || i < my_pieces)) *reinterpret_cast<v4u*>(xform_addr(buf, i)) = o;      // (the DMA's zero fill stays zero)
    };

    if (ntile > 0) issue_patch(cur, 0);
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");     // filter bank + first patch (DMA), bias / coefficient rows
    if (INBN && ntile > 0) {
        __builtin_amdgcn_s_barrier();                    // (the coefficient rows of every wave are in LDS)
        xform_coef();
#pragma unroll
        for (int i = 0; i < 6; ++i) xform_store(cur, 0, i, xform_load(0, i));
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    __builtin_amdgcn_s_barrier();

    // the tile whose read-out is pending: packed results, its residual / z vectors, validity and store offset
    v4u prev[2][2], rprev[2][2];
    bool pval[2] = {false, false};
    unsigned psoff = 0;
#pragma unroll
    for (int mt = 0; mt < 2; ++mt)
#pragma unroll
        for (int k = 0; k < 2; ++k) prev[mt][k] = rprev[mt][k] = v4u{0u, 0u, 0u, 0u};

    // read-out of channel vector k of the pending tile (both pixels): straight-line, stores of invalid pixels go out of range
    auto readout = [&](const int k) {
        v4u pk[2];
        const bool ok[2] = {pval[0] && cval[k], pval[1] && cval[k]};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            f32x2 mu, is, sc, sh;                        // MODE 2: the coefficients of this channel pair (LDS broadcast reads)
            if (MODE == 2) {
                mu = *reinterpret_cast<const f32x2*>(lbn + c0 + 16 * k + 2 * q);
                is = *reinterpret_cast<const f32x2*>(lbn + 64 + c0 + 16 * k + 2 * q);
                sc = *reinterpret_cast<const f32x2*>(lbn + 128 + c0 + 16 * k + 2 * q);
                sh = *reinterpret_cast<const f32x2*>(lbn + 192 + c0 + 16 * k + 2 * q);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned w = ok[mt] ? prev[mt][k][q] : 0u;   // pixels outside the image contribute nothing to the sums
                f32x2 v = unpack2(w);
                if (MODE == 2) {
                    const f32x2 d = unpack2(rprev[mt][k][q]) - mu;
                    const f32x2 y = __builtin_elementwise_fma(d, sc, sh);
                    // the value the apply pass will read is the ROUNDED masked gradient: the sums use it too
                    v = f32x2{y[0] > 0.f ? v[0] : 0.f, y[1] > 0.f ? v[1] : 0.f};
                    s1[k][q] += v;
                    s2[k][q] = __builtin_elementwise_fma(v, d * is, s2[k][q]);
                    pk[mt][q] = pack2(v);
                } else if (MODE == 1) {
                    v += unpack2(rprev[mt][k][q]);         // (no statistics in this mode: conv_res_pipe_ok)
                    pk[mt][q] = pack2(v);
                } else {
                    pk[mt][q] = w;
                    s1[k][q] += v;
                    s2[k][q] = __builtin_elementwise_fma(v, v, s2[k][q]);
                }
            }
        }
        // (tile base in the VECTOR offset, not in soffset: the registers of `pk` are rewritten by the next arithmetic a few
        //  instructions later, and hipcc pads the ISA's ">64-bit store data -> VALU write" hazard only for stores WITHOUT an SGPR
        //  offset -- with one, lanes 12-15 / 44-47 of the first data dword were stored after they had been overwritten)
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
            __builtin_amdgcn_raw_buffer_store_b128(pk[mt], rs_out, (int)(ok[mt] ? ovo[mt] + psoff + 32u * k : (unsigned)OOB), 0, 0);
    };

    RSTAMP_INIT
    for (int t = 0; t < ntile; ++t) {
        const int buf = t & 1;
        const Walk nxt = advance(cur);
        if (t + 1 < ntile) issue_patch(nxt, buf ^ 1);    // its buffer was released by the barrier that ended tile t-1
        // this tile's output pixels
        const int n0 = cur.q * a.TN, oy0 = cur.py * TH, ox0 = cur.px * TW;
        const unsigned osoff = (unsigned)(((n0 * a.Ho + oy0) * a.Wo + ox0) * a.out_ld) * 2u;
        bool val[2];
        v4u rz[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            val[mt] = n0 + (ogeo[mt] >> 20) < a.ipe && oy0 + ((ogeo[mt] >> 10) & 0x3ff) < a.Ho && ox0 + (ogeo[mt] & 0x3ff) < a.Wo;
        }
        // residual (a data gradient accumulating into the gradient another consumer left) / z of the BatchNorm whose backward
        // reductions this launch carries: consumed one tile later.  RZ_LATE (round 4, cycle stamps profiles/r04_respipe_dbn_stamps.log):
        // requested at the start of the tile, the four loads made the MFMA-idle phase before the first MFMA 3240-3470 cycles long
        // instead of 1525 (80 instead of 48 vector-memory instructions of the workgroup queue at the texture path's port while nothing
        // else runs); requested between the MFMAs of steps 22 / 26, behind the read-out's stores, they queue there under the other
        // waves' MFMAs, and the tile-end wait leaves them in flight (the youngest four of the wave's in-order stream).
        const unsigned rsoff = HASRES ? (unsigned)(((n0 * a.Ho + oy0) * a.Wo + ox0) * a.res_ld) * 2u : 0u;
        auto load_rz = [&](const int mt) {
#pragma unroll
            for (int k = 0; k < 2; ++k)
                rz[mt][k] = __builtin_amdgcn_raw_buffer_load_b128(rs_res, (int)(val[mt] && cval[k] ? rvo[mt] : (unsigned)OOB),
                                                                  (int)(rsoff + 32u * k), 0);
        };
        if (HASRES && !RZ_LATE) { load_rz(0); load_rz(1); }
        const char* patch = pbuf + buf * pbuf_bytes;
        f32x16 acc[2];
        if (BIAS) {
            // accumulators start from the channel bias (acc[mt][4g+i] is cout wn*32 + 8g + 4hh + i)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 b = *reinterpret_cast<const f32x4*>(lbias + wn * 32 + g * 8 + hh * 4);
#pragma unroll
                for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                    for (int k = 0; k < 4; ++k) acc[mt][4 * g + k] = b[k];
            }
        } else {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[mt][k] = 0.f;
        }
        RLAP(0)                                          // patch requests, residual prefetch, accumulator init
        v4u xz = v4u{0u, 0u, 0u, 0u};
#pragma unroll
        for (int st = 0; st < 36; ++st) {
            const int tap = st >> 2, ks = st & 3;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) << LOG_RBK;
            const v4i af = *reinterpret_cast<const v4i*>(Wl + tap * 64 * RB + aoff[ks]);
            v4i bfr[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                bfr[mt] = *reinterpret_cast<const v4i*>(patch + pbase[mt] + tapoff +
                                                        (((ks * 2 + hh) ^ (((pcol[mt] + (tap % 3)) >> 1) & 7)) << 4));
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af), __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                  acc[mt], 0, 0, 0);
            if (st == 3) readout(0);                     // the previous tile leaves under this tile's MFMAs ...
            if (st == 11) readout(1);
            // ... in their first half (nothing is scheduled across this fence): the stores have the second half to be
            // acknowledged before the vmcnt(0) below has to wait for them
            if (st == 19) __builtin_amdgcn_sched_barrier(0);
            if (HASRES && RZ_LATE && st == 22) load_rz(0);
            if (HASRES && RZ_LATE && st == 26) load_rz(1);
            if (INBN && t + 1 < ntile) {
                // the next tile's patch: this wave's requests (issued before the read-out's four stores) have landed.  One piece per
                // two steps: its load at an even step, arithmetic and store at the next one
                if (st == 19) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); xform_coef(); }
                if (st >= 20 && st < 32) {
                    if (!(st & 1)) xz = xform_load(buf ^ 1, (st - 20) >> 1);
                    else xform_store(nxt, buf ^ 1, (st - 20) >> 1, xz);
                }
            }
        }
        RLAP(1)                                          // 36 steps of fragment reads + MFMAs (+ the previous tile's read-out)
        // ---- this tile becomes the pending one: round to bf16, 8-byte pieces -> 16-byte channel vectors across the lane halves
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                unsigned x[2], y[2];
#pragma unroll
                for (int d = 0; d < 2; ++d) {
                    x[d] = pack2(f32x2{acc[mt][8 * k + 2 * d], acc[mt][8 * k + 2 * d + 1]});          // piece g = 2k   (couts 16k + 4hh + ..)
                    y[d] = pack2(f32x2{acc[mt][8 * k + 4 + 2 * d], acc[mt][8 * k + 4 + 2 * d + 1]});  // piece g = 2k+1 (couts 16k + 8 + 4hh + ..)
                    const auto r = __builtin_amdgcn_permlane32_swap(x[d], y[d], false, false);        // x[32..63] <-> y[0..31]
                    x[d] = r[0]; y[d] = r[1];
                }
                prev[mt][k] = v4u{x[0], x[1], y[0], y[1]};       // channels wn*32 + 16k + 8hh + 0..7 of pixel (mt, l31)
                if (HASRES) rprev[mt][k] = rz[mt][k];
            }
            pval[mt] = val[mt];
        }
        psoff = osoff;
        cur = nxt;
        if (HASRES && RZ_LATE) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");      // (all but this tile's four side-input loads)
        else
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // the next tile's patch has landed (+ residual vectors, earlier stores)
        __builtin_amdgcn_s_barrier();                    // ... for every wave, and every wave is done reading this tile's patch
        RLAP(2)
    }
    readout(0);
    readout(1);
#ifdef PMOE_STAMP
    if (a.stats && lane == 0) {
        float* o = a.stats + (((size_t)e * wgs_per_expert + wg) * 8 + wave) * 8;
        for (int i = 0; i < 5; ++i) o[i] = (float)st_acc[i];
        o[5] = (float)(unsigned)(__builtin_amdgcn_s_memrealtime() - st_r0);
        o[6] = (float)ntile;
    }
    return;
#endif

    if (a.stats) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);      // [8 waves][2][64]  (the filter bank is dead by now)
#pragma unroll
        for (int k = 0; k < 2; ++k)
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    float u = s1[k][q][c], v = s2[k][q][c];
#pragma unroll
                    for (int off = 1; off < 32; off <<= 1) {
                        u += __shfl_xor(u, off);
                        v += __shfl_xor(v, off);
                    }
                    if (l31 == 0) {
                        red[(wave * 2 + 0) * 64 + c0 + 16 * k + 2 * q + c] = u;
                        red[(wave * 2 + 1) * 64 + c0 + 16 * k + 2 * q + c] = v;
                    }
                }
        __syncthreads();
        if (tid < 128) {
            const int which = tid >> 6, c = tid & 63;
            float s = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) s += red[((w * 2 + (c >> 5)) * 2 + which) * 64 + c];
            a.stats[(((size_t)e * wgs_per_expert + wg) * 2 + which) * a.CoutP + c] = s;
        }
    }
}

// ------------------------------------------------------------------------------------------------
bool conv_res_plan(const ConvArgs& a, int dtype, ResPlan* plan) {
    static int res_on = -1;       // PMOE_CONV_RES=0: route the <=64-channel layers to the generic kernel (A/B runs)
    if (res_on < 0) { const char* ev = getenv("PMOE_CONV_RES"); res_on = ev ? atoi(ev) : 1; }
    if (!res_on) return false;
    if (dtype != PMOE_DT_BF16 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.dilate) return false;
    if (a.CoutP != 64 || a.Cout % 8 || (a.Cin != 64 && a.Cin != 16)) return false;
    if (a.act != PMOE_ACT_NONE || a.drop_p > 0.f) return false;
    if (a.bias && ((a.res_mode != PMOE_RES_NONE && a.res_mode != PMOE_RES_DBN) || a.Cin != 64)) return false;
    if (a.res_mode != PMOE_RES_NONE && a.res_mode != PMOE_RES_ADD && a.res_mode != PMOE_RES_DBN && a.res_mode != PMOE_RES_INBN) return false;
    if (a.res_mode == PMOE_RES_DBN && a.Cin != 64) return false;    // (conv3x3_resdma_kernel only: conv_igemm_launch)
    if (a.res_mode == PMOE_RES_INBN && (a.Cin != 64 || a.in_shared || !a.bn)) return false;      // (conv3x3_respipe_kernel<false, 3> only)
    if (a.N % a.ipe || a.Ho != a.H || a.Wo != a.W) return false;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.Wo); if (lTW > 5) lTW = 5;
    int lTH = p2(a.Ho); if (lTH > 8 - lTW) lTH = 8 - lTW;
    const int TN = 256 >> (lTW + lTH);
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int NPIX = TN * (TH + 2) * (TW + 2);
    const int log_rb = a.Cin == 64 ? 7 : 5;
    const int rb = 1 << log_rb, cpr = rb / 16;
    if (NPIX * cpr > 11 * 256) return false;
    size_t region = (size_t)NPIX * rb;
    if (region < 256 * 128) region = 256 * 128;
    region = (region + 255) & ~(size_t)255;
    const size_t smem = (size_t)9 * 64 * rb + 2 * region;
    if (smem + (a.bias ? 256 : 0) > 163840) return false;
    const int E = a.N / a.ipe;
    plan->lTW = lTW; plan->lTH = lTH; plan->TN = TN;
    plan->n_groups = (a.ipe + TN - 1) / TN;
    plan->tiles_y = (a.Ho + TH - 1) / TH;
    plan->tiles_x = (a.Wo + TW - 1) / TW;
    plan->tiles_per_expert = plan->n_groups * plan->tiles_y * plan->tiles_x;
    int wpe = 256 / E; if (wpe < 1) wpe = 1;
    if (wpe > plan->tiles_per_expert) wpe = plan->tiles_per_expert;
    plan->wgs_per_expert = wpe;
    plan->log_rb = log_rb;
    plan->smem = smem;
    return true;
}

// does the all-waves-compute / LDS-DMA variant (conv3x3_resdma_kernel) take this plan?  PMOE_RES_DMA=0: A/B switch back to
// the ping-pong kernel (read per launch)
bool conv_res_dma_ok(const ConvArgs& a, const ResPlan& p, int* pbuf, int* magic_pw, int* magic_ph, size_t* smem) {
    if (p.log_rb != 7 || a.in_shared) return false;
    const char* ev = getenv("PMOE_RES_DMA");
    if (ev && !atoi(ev)) return false;
    const int PW = (1 << p.lTW) + 2, PH = (1 << p.lTH) + 2;
    const int npiece = (p.TN * PH * PW + 7) / 8;
    int pb = npiece * 1024;
    if (pb < 256 * 128) pb = 256 * 128;
    const int mpw = 65536 / PW + 1, mph = 65536 / PH + 1;
    for (int pp = 0; pp < npiece * 8; ++pp)
        if (((pp * mpw) >> 16) != pp / PW || ((((pp / PW) * mph) >> 16) != (pp / PW) / PH)) return false;
    const size_t sm = (size_t)9 * 64 * 128 + 2 * (size_t)pb + 256 + 1024;     // + bias row + BatchNorm coefficient rows
    if (npiece > 48 || sm > 163840 || p.lTW < 4 || (long long)a.ipe * a.H * a.W * a.in_ld * 2 >= 0x7ff00000ll) return false;
    *pbuf = pb; *magic_pw = mpw; *magic_ph = mph; *smem = sm;
    return true;
}

// does the software-pipelined variant (conv3x3_respipe_kernel: register read-out under the next tile's MFMAs) take what
// conv_res_dma_ok accepted?  PMOE_RES_PIPE=0: A/B switch back to conv3x3_resdma_kernel (read per launch)
bool conv_res_pipe_ok(const ConvArgs& a) {
    const char* ev = getenv("PMOE_RES_PIPE");
    if (ev && !atoi(ev)) return false;
    if (a.ipe > 2047 || (a.res_mode == PMOE_RES_ADD && a.stats)) return false;
    // (measured and not kept, same box, profiles/r03_kernel_ab.log: the requests for the next patch as straight-line code inside
    //  the MFMA block -- range checks as one guarded subtraction per bound, a wave without a 6th piece requesting its 5th again --
    //  and the residual / z vectors loaded into the registers the read-out has just freed: within 1 % on the stem, 3-5 % slower
    //  on layer1)
    if ((long long)a.ipe * a.Ho * a.Wo * a.out_ld * 2 >= 0x7ff00000ll) return false;
    if (a.res_mode != PMOE_RES_NONE && (long long)a.ipe * a.Ho * a.Wo * a.res_ld * 2 >= 0x7ff00000ll) return false;
    return true;
}

template <bool BIAS, int MODE>
static int launch_respipe(const ConvArgs& a, const ResPlan& p, dim3 grid, size_t sm, int pb, int mpw, int mph, hipStream_t st) {
    if constexpr (MODE == 1 || MODE == 2) {              // PMOE_RES_RZ_LATE=0: the side-input loads back at the start of the tile (A/B)
        const char* ev = getenv("PMOE_RES_RZ_LATE");
        if (ev && !atoi(ev)) {
            HIP_RET((ensure_dyn_lds<conv3x3_respipe_kernel<BIAS, MODE, false>>(163840)));
            hipLaunchKernelGGL((conv3x3_respipe_kernel<BIAS, MODE, false>), grid, dim3(512), sm, st, a, p.tiles_per_expert, p.wgs_per_expert, pb,
                               mpw, mph);
            return (int)hipGetLastError();
        }
    }
    HIP_RET((ensure_dyn_lds<conv3x3_respipe_kernel<BIAS, MODE>>(163840)));
    hipLaunchKernelGGL((conv3x3_respipe_kernel<BIAS, MODE>), grid, dim3(512), sm, st, a, p.tiles_per_expert, p.wgs_per_expert, pb, mpw, mph);
    return (int)hipGetLastError();
}

int conv_res_launch(ConvArgs a, const ResPlan& p, hipStream_t st) {
    a.lTW = p.lTW; a.lTH = p.lTH; a.TN = p.TN; a.n_groups = p.n_groups; a.tiles_y = p.tiles_y; a.tiles_x = p.tiles_x;
    const int E = a.N / a.ipe;
    const int region = (int)((p.smem - (size_t)9 * 64 * (1 << p.log_rb)) / 2);
    dim3 grid(p.wgs_per_expert, E), block(512);
    int pb = 0, mpw = 0, mph = 0;
    size_t sm = 0;
    if (conv_res_dma_ok(a, p, &pb, &mpw, &mph, &sm)) {
        if (conv_res_pipe_ok(a)) {
            const int mode = a.res_mode == PMOE_RES_DBN ? 2 : a.res_mode == PMOE_RES_ADD ? 1 : 0;
            if (a.res_mode == PMOE_RES_INBN) return launch_respipe<false, 3>(a, p, grid, sm, pb, mpw, mph, st);
            if (a.bias) return mode == 2 ? launch_respipe<true, 2>(a, p, grid, sm, pb, mpw, mph, st)
                             : mode == 1 ? launch_respipe<true, 1>(a, p, grid, sm, pb, mpw, mph, st)
                                         : launch_respipe<true, 0>(a, p, grid, sm, pb, mpw, mph, st);
            return mode == 2 ? launch_respipe<false, 2>(a, p, grid, sm, pb, mpw, mph, st)
                 : mode == 1 ? launch_respipe<false, 1>(a, p, grid, sm, pb, mpw, mph, st)
                             : launch_respipe<false, 0>(a, p, grid, sm, pb, mpw, mph, st);
        }
        if (a.res_mode == PMOE_RES_INBN) return PMOE_ERR_UNSUPPORTED;
        const char* evp = getenv("PMOE_RES_PREFETCH");
        a.prefetch = !(evp && !atoi(evp));
        if (a.bias) {
            HIP_RET((ensure_dyn_lds<conv3x3_resdma_kernel<true>>(163840)));
            hipLaunchKernelGGL(conv3x3_resdma_kernel<true>, grid, block, sm, st, a, p.tiles_per_expert, p.wgs_per_expert, pb, mpw, mph);
        } else {
            HIP_RET((ensure_dyn_lds<conv3x3_resdma_kernel<false>>(163840)));
            hipLaunchKernelGGL(conv3x3_resdma_kernel<false>, grid, block, sm, st, a, p.tiles_per_expert, p.wgs_per_expert, pb, mpw, mph);
        }
        return (int)hipGetLastError();
    }
    if (a.res_mode == PMOE_RES_INBN) return PMOE_ERR_UNSUPPORTED;
    if (p.log_rb == 7 && a.bias) {
        HIP_RET((ensure_dyn_lds<conv3x3_res_kernel<7, true>>(163840)));
        hipLaunchKernelGGL((conv3x3_res_kernel<7, true>), grid, block, p.smem + 256, st, a, p.tiles_per_expert,
                           p.wgs_per_expert, region);
    } else if (p.log_rb == 7) {
        HIP_RET((ensure_dyn_lds<conv3x3_res_kernel<7, false>>(163840)));
        hipLaunchKernelGGL((conv3x3_res_kernel<7, false>), grid, block, p.smem, st, a, p.tiles_per_expert, p.wgs_per_expert, region);
    } else {
        HIP_RET((ensure_dyn_lds<conv3x3_res_kernel<5, false>>(163840)));
        hipLaunchKernelGGL((conv3x3_res_kernel<5, false>), grid, block, p.smem, st, a, p.tiles_per_expert, p.wgs_per_expert, region);
    }
    return (int)hipGetLastError();
}
