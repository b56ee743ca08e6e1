// Grouped (per-expert) convolution weight gradient on MFMA, gfx950.
//
//   dW[e][tap][cout][cin] += sum over the expert's output pixels of dY[pixel][cout] * X[pixel shifted by tap][cin]
//
// The reduction (GEMM K) dimension is the pixel index, which is the SLOW axis of both NHWC
// operands.  A workgroup stages a dY tile [pixels][CK couts] and the matching X halo patch
// [pixels+halo][CK cins] in LDS exactly as the forward kernel does and reads both MFMA operands
// TRANSPOSED: bf16 with ds_read_b64_tr_b16 (4 pixels x 16 channels per 16-lane group), f32 with
// scalar ds_read_b32 (the f32 MFMA takes one element per lane).  One dY fragment is reused for all
// ks*ks taps, so each tap costs one transposed X read per MFMA.  Every wave keeps its
// [32 cout][32 cin] x taps accumulators in registers across ALL the m-blocks the workgroup walks
// (K-split over the flat grid) and flushes ONCE, deterministically: the waves that split the pixels of one output tile
// are summed through LDS in fixed order, the workgroup stores its tile with plain stores (lanes along cin: 2 x 128-byte
// segments per wave-instruction) into its own slab of a partial workspace, and `wgrad_reduce_kernel` folds the K-split
// slabs in fixed order.  No float atomics anywhere: weight gradients are bit-reproducible run to run.
//
// Pipeline (one workgroup per CU, one wave per SIMD, the whole 512-register file): the global loads
// of m-block i+1 (dY tile + X halo patch, up to 28 x 16 B per lane) are issued into registers
// right after the barrier that publishes m-block i and land under its ~4600 cycles of MFMA; they
// are written to LDS after the next barrier (issue early / write late).
#include "conv_common.h"
#include <stdlib.h>
#include "kernels.h"

template <typename T> struct WgradCfg;
// bf16: 8 waves = 2(cout) x 2(cin) x 2(pixel halves): two waves per SIMD share an output tile and split the
// pixels, so one wave's LDS reads / address math run under the other's MFMAs.  f32: 4 waves split pixels 4 ways.
template <> struct WgradCfg<bf16> { static constexpr int WCO = 2, WCI = 2, WK = 2, NW = 8; };   // 64 x 64 channels / WG
template <> struct WgradCfg<float> { static constexpr int WCO = 1, WCI = 1, WK = 4, NW = 4; };  // 32 x 32 channels / WG

typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ s16x4 tr_read(const char* p) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
}

template <typename T, int TAPS, int MAXV>
__global__ void __launch_bounds__(WgradCfg<T>::NW * 64) conv_wgrad_kernel(const WgradArgs a) {
    constexpr int NTHR = WgradCfg<T>::NW * 64;
    constexpr int LOG_RB = 7, RB = 128;
    constexpr int VE = 16 / (int)sizeof(T);
    constexpr int CKW = RB / (int)sizeof(T);            // channels per operand tile: 64 bf16 / 32 f32
    constexpr int WCI = WgradCfg<T>::WCI, WK = WgradCfg<T>::WK;
    constexpr int KS = TAPS == 9 ? 3 : 1;
    constexpr int MAXDY = 256 * 8 / NTHR;               // dY tile: up to 256 pixels x 8 chunks
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, h = lane >> 5;
    constexpr int WCO = WgradCfg<T>::WCO;
    const int ci_sub = wave % WCI;
    const int co_sub = (wave / WCI) % WCO;
    const int k_sub = wave / (WCI * WCO);

    const int e = blockIdx.z;
    const int n_ci_blk = (a.Cin + CKW - 1) / CKW;
    // flat grid, channel-tile pair FASTEST: the workgroups that walk the SAME pixels (every (cout, cin) tile pair of one K
    // slice reads the same dY tiles / X patches) are dispatch neighbours and, through the XCD remap, share one XCD's L2 --
    // with the slice index fastest they were a whole grid row apart and each operand byte came from HBM once per pair
    // (1.10 GB fetched per launch against 0.54 GB compulsory, profiles/traffic.json of round 1)
    const int npairs = n_ci_blk * ((a.Cout + CKW - 1) / CKW);
    const unsigned flat = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = (int)(gridDim.x / npairs);
    // (a.slice_fastest: the round-1 order, K slice fastest -- A/B switch PMOE_WGRAD_SLICE_FASTEST=1)
    const int pair = a.slice_fastest ? (int)(blockIdx.x / nsplit) : (int)(flat % npairs);
    const int split = a.slice_fastest ? (int)(blockIdx.x % nsplit) : (int)(flat / npairs);
    const int cob = pair / n_ci_blk, cib = pair % n_ci_blk;
    const int co0 = cob * CKW, ci0 = cib * CKW;

    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int S = KS == 1 ? 1 : a.stride, step = KS == 1 ? a.stride : 1;   // strided 1x1: stage only used pixels
    const int PW = (TW - 1) * S + KS, PH = (TH - 1) * S + KS;
    const int NPIX = a.TN * PH * PW;
    const int BMP = a.TN << (lTW + lTH);
    const int mbpe = a.n_groups * a.tiles_y * a.tiles_x;

    char* dyt = smem;                                   // [BMP][192 B]
    constexpr int RS = 192;                             // LDS row stride (128 B of channels + 64 B pad)
    char* patch = smem + BMP * RS;                      // [NPIX][192 B]

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    const T* x = (const T*)a.x;
    const T* dy = (const T*)a.dy;
    const int mb_begin = split * a.mb_per_wg;
    int mb_end = mb_begin + a.mb_per_wg;
    if (mb_end > mbpe) mb_end = mbpe;

    PatchStage<T, LOG_RB, NTHR, MAXV> xs;
    v4i dyv[MAXDY];
    const int dyj = tid & 7;
    const bool dy_cok = co0 + dyj * VE < a.Cout;

    auto issue = [&](int mbi) {
        int t = mbi;
        const int px = t % a.tiles_x; t /= a.tiles_x;
        const int py = t % a.tiles_y; t /= a.tiles_y;
        const int ng = t;
        const int n0 = e * a.ipe + ng * a.TN, n_end = (e + 1) * a.ipe;
        const int oy0 = py * TH, ox0 = px * TW;
#pragma unroll
        for (int u = 0; u < MAXDY; ++u) {
            const int p = (tid >> 3) + u * (NTHR / 8);
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            const int n = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
            dyv[u] = v4i{0, 0, 0, 0};
            if (p < BMP && dy_cok && n < n_end && oy < a.Ho && ox < a.Wo)
                dyv[u] = ldg16(dy + (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.dy_ld + a.dy_coff + co0 + dyj * VE);
        }
        PatchGeom geo;
        geo.n0 = n0; geo.n_end = n_end; geo.e_first_img = e * a.ipe;
        geo.Y0 = oy0 * a.stride - a.pad; geo.X0 = ox0 * a.stride - a.pad; geo.PH = PH; geo.PW = PW; geo.NPIX = NPIX;
        geo.H = a.H; geo.W = a.W; geo.ld = a.x_ld; geo.coff = a.x_coff; geo.cmax = a.Cin;
        geo.dilate = 0; geo.shared = a.x_shared; geo.step = step;
        xs.issue(x, geo, ci0, tid);
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < MAXDY; ++u) {
            const int p = (tid >> 3) + u * (NTHR / 8);
            if (p < BMP) *reinterpret_cast<v4i*>(dyt + lds_chunk_off<LOG_RB, 2>(p, dyj)) = dyv[u];
        }
        xs.template commit<2>(patch, NPIX, tid);
    };

    if (mb_begin < mb_end) issue(mb_begin);
    for (int mbi = mb_begin; mbi < mb_end; ++mbi) {
        __syncthreads();                                // previous tile's operand reads are done
        commit();
        __syncthreads();
        if (mbi + 1 < mb_end) issue(mbi + 1);           // in flight under this tile's MFMAs

#pragma unroll 2
        for (int kb = k_sub; kb < (BMP >> 4); kb += WK) {
            const int p0 = kb << 4;
            if constexpr (sizeof(T) == 2) {
                // transposed fragments: lane (g = lane>>4: channel block g&1, k half g>>1;
                // q = (lane>>2)&3: pixel row it addresses; pc = lane&3: 4-channel column group)
                const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
                const int cblk = 16 * (g & 1) + 4 * pc;              // channel inside the wave's 32
                int pA[2], ppB[2];
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const int p = p0 + 8 * (g >> 1) + 4 * tt + q;
                    pA[tt] = p;
                    const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                    ppB[tt] = (pn * PH + my * S) * PW + mx * S;
                }
                const int ca = (co_sub * 32 + cblk) * 2, cb = (ci_sub * 32 + cblk) * 2;   // byte offsets in a row
                bf16x8 fa;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const s16x4 r = tr_read(dyt + pA[tt] * RS + ca);
                    const bf16x4 rb = __builtin_bit_cast(bf16x4, r);
#pragma unroll
                    for (int i = 0; i < 4; ++i) fa[4 * tt + i] = rb[i];
                }
                const char* bB[2] = {patch + ppB[0] * RS + cb, patch + ppB[1] * RS + cb};
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) {
                    const int tapoff = ((tap / KS) * PW + (tap % KS)) * RS;      // wave-uniform
                    bf16x8 fb;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const s16x4 r = tr_read(bB[tt] + tapoff);
                        const bf16x4 rb = __builtin_bit_cast(bf16x4, r);
#pragma unroll
                        for (int i = 0; i < 4; ++i) fb[4 * tt + i] = rb[i];
                    }
                    acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
                }
            } else {
                // f32: MFMA 32x32x2, lane holds A[i = l31][k = h]; 8 MFMAs cover 16 pixels
#pragma unroll
                for (int m = 0; m < 8; ++m) {
                    const int p = p0 + 2 * m + h;
                    const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                    const int ppb = (pn * PH + my * S) * PW + mx * S;
                    const float av = *reinterpret_cast<const float*>(dyt + p * RS + (l31 << 2));
#pragma unroll
                    for (int tap = 0; tap < TAPS; ++tap) {
                        const int pp = ppb + (tap / KS) * PW + (tap % KS);
                        const float bv = *reinterpret_cast<const float*>(patch + pp * RS + (l31 << 2));
                        acc[tap] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc[tap], 0, 0, 0);
                    }
                }
            }
        }
    }

    // flush: D[row = cout][col = cin]; lanes run along cin -> contiguous 128-byte segments.  The WK waves that share
    // an output tile meet in LDS (TPR taps per round: the operand tiles are dead by now), wave k_sub 0 adds the others'
    // tiles in fixed order and stores.  Destination: the workgroup's own K-split slab (or dw itself when there is no split).
    constexpr int TPR = TAPS == 9 ? 3 : 1;
    constexpr int TILE_WAVES = WCO * WCI;
    float* red = reinterpret_cast<float*>(smem);
    const int tw = wave % TILE_WAVES;
    const int cin = ci0 + ci_sub * 32 + l31;
    const size_t slab = a.per_image ? (size_t)e * a.ipe + split               // per image
                                    : (size_t)split * gridDim.z + e;            // per (K-split, expert); split 0 = dw itself
    float* dst = (a.per_image || nsplit == 1) ? a.dw : a.part;
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TPR) {
        __syncthreads();
        if (k_sub > 0) {
#pragma unroll
            for (int tp = 0; tp < TPR; ++tp)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[((((k_sub - 1) * TPR + tp) * TILE_WAVES + tw) * 16 + r) * 64 + lane] = acc[t0 + tp][r];
        }
        __syncthreads();
        if (k_sub == 0) {
#pragma unroll
            for (int tp = 0; tp < TPR; ++tp) {
                float* base = dst + ((slab * TAPS + t0 + tp) * a.CoutP) * a.CinP;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t0 + tp][r];
#pragma unroll
                    for (int k = 1; k < WK; ++k) v += red[((((k - 1) * TPR + tp) * TILE_WAVES + tw) * 16 + r) * 64 + lane];
                    const int cout = co0 + co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    base[(size_t)cout * a.CinP + cin] = v;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// LDS-DMA variant for the dense 3x3 stride-1 layers in bf16 (the 15 big launches of a step): the dY tile and the X halo
// patch of the NEXT m-block travel global -> LDS by `buffer_load_dwordx4 ... lds` while the current m-block is computed
// out of the other buffer pair: no staging registers (40 VGPRs of the kernel above), no commit pass, ONE barrier per
// m-block instead of two.  LDS-DMA writes lane-linearly (8 pixel rows of 128 B per wave-instruction), so the 192-byte row
// padding of the kernel above is not available; the transposed reads stay conflict-free through an XOR on the 64-byte half
// of a row keyed by bit 1 of the pixel's COLUMN (the 4 rows of a ds_read_b64_tr_b16 block then hit 4 different 64-byte
// bank ranges for every tap shift), applied to the per-lane SOURCE address and to the reads.  The key of a read is
// lane-constant per tap column (3 precomputed offsets), so a tap still costs one immediate row offset.  Out-of-image
// pixels, images beyond the expert's last and channels beyond Cin / Cout arrive as zeros from the buffer range check.
// WCI = 1 (layers with <= 32 input channels: the per-image gradient of the 16-channel stem convolution): the 8 waves are
// 2 (cout) x 4 (pixel quarters) instead of 2 x 2 x 2 -- with WCI = 2 half of them multiplied the zero-filled channels 32..63.
// PAIRS (WCI = 1, <= 16 input channels: the stem's first convolution, 12 -> 16): the 32 columns of an MFMA hold TWO taps x
// 16 channels instead of one tap x 16 channels + 16 columns of zero fill -- lanes 16..31 of a fragment read the next tap's
// pixel (a per-lane tap offset), 5 MFMAs per k-block instead of 9 (round 3: 0.83 -> ~0.5 ms on the per-image stem gradient).
// REQ (round 3; cycle stamps of the forward kernels showed the request phase -- two runtime divisions and ~30 VALU
// instructions per piece, ten pieces per wave, every wave of the workgroup at once right behind the barrier -- costing a third of
// an m-block's MFMA time): 1 = the per-lane geometry of the pieces is tile-invariant and computed once, the tile walk is
// incremental, range checks are one guarded subtraction per bound (~7 VALU per piece); 2 = ... and the requests are issued
// between the k-blocks of the first half of the m-block instead of all at its top.  0 = the round-2 code (A/B).
// Measured (profiles/r03_kernel_ab.log, one box, interleaved): 1 = +1.5..4 % on every layer (shipped); 2 = 4-8 % SLOWER than 0 --
// the requests were never what held this kernel back; its k-loop is (the MFMA of tap t waits for the transposed X fragment
// requested one MFMA earlier), and the fully unrolled loop that mode 2 needs schedules worse.
template <int PIN, int WCI = 2, bool PAIRS = false, int REQ = 0>
__global__ void __launch_bounds__(512) conv_wgrad_dma_kernel(const WgradArgs a, const int magic_pw, const int magic_ph) {
    constexpr int TAPS = 9, RS = 128, CKW = 64;
    static_assert(!PAIRS || WCI == 1, "tap pairs: the narrow wave layout only");
    static_assert(REQ == 0 || (WCI == 2 && !PAIRS && PIN == 1), "precomputed requests: the 2 x 2 x 2 wave layout only");
    constexpr int WCO = 2, WK = 8 / (WCI * WCO);
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int ci_sub = wave % WCI, co_sub = (wave / WCI) % WCO, k_sub = wave / (WCI * WCO);

    const int e = blockIdx.z;
    const int n_ci_blk = (a.Cin + CKW - 1) / CKW;
    const int npairs = n_ci_blk * ((a.Cout + CKW - 1) / CKW);
    const unsigned flat = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = (int)(gridDim.x / npairs);
    const int pair = (int)(flat % npairs), split = (int)(flat / npairs);
    const int cob = pair / n_ci_blk, cib = pair % n_ci_blk;
    const int co0 = cob * CKW, ci0 = cib * CKW;

    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int BMP = a.TN << (lTW + lTH);                 // 256
    const int XB = NPIECE << 10;                         // bytes of one X patch buffer
    const int pair_bytes = (BMP << 7) + XB;              // dY tile + X patch of one m-block
    const int my_pieces = (NPIECE - wave + 7) >> 3;

    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    const int mbpe = a.n_groups * a.tiles_y * a.tiles_x;
    const int mb_begin = split * a.mb_per_wg;
    int mb_end = mb_begin + a.mb_per_wg;
    if (mb_end > mbpe) mb_end = mbpe;

    // buffer resources: one expert's images (offsets < 2^31 bytes; the launcher checks)
    const bf16* xb = (const bf16*)a.x + (a.x_shared ? (size_t)0 : (size_t)e * a.ipe * a.H * a.W * a.x_ld) + a.x_coff + ci0;
    const bf16* dyb = (const bf16*)a.dy + (size_t)e * a.ipe * a.Ho * a.Wo * a.dy_ld + a.dy_coff + co0;
    const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
        (void*)xb, (short)0, (int)(((long long)a.ipe * a.H * a.W * a.x_ld - a.x_coff - ci0) * 2), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
        (void*)dyb, (short)0, (int)(((long long)a.ipe * a.Ho * a.Wo * a.dy_ld - a.dy_coff - co0) * 2), 0x00020000);
    constexpr int OOB = 0x7ff80000;

    auto issue = [&](int mbi, int buf) {
        int t = mbi;
        const int px_t = t % a.tiles_x; t /= a.tiles_x;
        const int py_t = t % a.tiles_y; t /= a.tiles_y;
        const int n0 = t * a.TN;                         // image index inside the expert
        const int oy0 = py_t * TH, ox0 = px_t * TW;
        char* base = smem + buf * pair_bytes;
        // dY tile: 32 pieces of 8 pixels, 4 per wave
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = ((wave + 8 * i) << 3) + (lane >> 3);
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            const int n = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
            const int j = (lane & 7) ^ (((p >> 1) & 1) << 2);
            const bool ok = n < a.ipe && oy < a.Ho && ox < a.Wo && co0 + j * 8 < a.Cout;
            const int voff = ok ? ((((n * a.Ho + oy) * a.Wo + ox) * a.dy_ld) << 1) + (j << 4) : OOB;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_void*)(base + ((wave + 8 * i) << 10)), 16, voff, 0, 0, 0);
        }
        // X halo patch: NPIECE pieces, up to 6 per wave
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (i < my_pieces) {
                const int pp = ((wave + 8 * i) << 3) + (lane >> 3);
                const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
                const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
                const int n = n0 + pn, Y = oy0 - a.pad + prow, X = ox0 - a.pad + px;
                const int j = (lane & 7) ^ (((px >> 1) & 1) << 2);
                const bool ok = pp < NPIX && n < a.ipe && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W &&
                                ci0 + j * 8 < a.Cin;
                const int voff = ok ? ((((n * a.H + Y) * a.W + X) * a.x_ld) << 1) + (j << 4) : OOB;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(base + (BMP << 7) + ((wave + 8 * i) << 10)), 16,
                                                         voff, 0, 0, 0);
            }
        }
    };

    // REQ >= 1: (image, row, column) of a piece's pixel inside the tile / patch as three 9-bit fields with a guard bit each: the
    // guard of a field survives `x | G - lo` iff x >= lo, so one subtraction per bound checks all three.  A wave without a 6th X
    // piece requests its 5th again (same bytes to the same place): every wave issues the same instruction stream.
    constexpr int GUARD = (1 << 9) | (1 << 19) | (1 << 29);
    int drel[4], dgeo[4], xrel[6], xgeo[6], xslot[6];
    struct Walk { int px, py, ng; };
    Walk nxtw{0, 0, 0};
    if constexpr (REQ >= 1) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = ((wave + 8 * i) << 3) + (lane >> 3);
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            const int j = (lane & 7) ^ (((p >> 1) & 1) << 2);
            drel[i] = ((((pn * a.Ho + my) * a.Wo + mx) * a.dy_ld) << 1) + (j << 4);
            dgeo[i] = (co0 + j * 8 < a.Cout ? (pn << 20) | (my << 10) | mx : (511 << 20)) | GUARD;
        }
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            xslot[i] = wave + 8 * (i < my_pieces ? i : my_pieces - 1);
            const int pp = (xslot[i] << 3) + (lane >> 3);
            const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
            const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
            const int j = (lane & 7) ^ (((px >> 1) & 1) << 2);
            xrel[i] = ((((pn * a.H + prow) * a.W + px) * a.x_ld) << 1) + (j << 4);
            xgeo[i] = (pp < NPIX && ci0 + j * 8 < a.Cin ? (pn << 20) | (prow << 10) | px : (511 << 20)) | GUARD;
        }
        int t = mb_begin;
        nxtw.px = t % a.tiles_x; t /= a.tiles_x;
        nxtw.py = t % a.tiles_y; nxtw.ng = t / a.tiles_y;
    }
    auto imin = [](int x, int y) { return x < y ? x : y; };
    // scalars of the m-block `nxtw` points at (uniform), then one request per call; `live` = false parks the offsets out of range
    int rq_dbase = 0, rq_dkk = 0, rq_xbase = 0, rq_xlo = 0, rq_xkk = 0, rq_want = 0;
    auto req_begin = [&](bool live) {
        const int n0 = nxtw.ng * a.TN, oy0 = nxtw.py * TH, ox0 = nxtw.px * TW, Y0 = oy0 - a.pad, X0 = ox0 - a.pad;
        const int nmax = imin(a.ipe - n0, 511) - 1;
        rq_dbase = (((n0 * a.Ho + oy0) * a.Wo + ox0) * a.dy_ld) << 1;
        rq_dkk = (((nmax << 20) | ((imin(TH, a.Ho - oy0) - 1) << 10) | (imin(TW, a.Wo - ox0) - 1)) | GUARD) + GUARD;
        rq_xbase = (((n0 * a.H + Y0) * a.W + X0) * a.x_ld) << 1;
        rq_xlo = ((Y0 < 0 ? -Y0 : 0) << 10) | (X0 < 0 ? -X0 : 0);
        rq_xkk = (((nmax << 20) | ((imin(PH, a.H - Y0) - 1) << 10) | (imin(PW, a.W - X0) - 1)) | GUARD) + GUARD;
        rq_want = live ? GUARD : -1;                     // (no branch on `live`: a comparison that cannot hold)
        if (++nxtw.px == a.tiles_x) { nxtw.px = 0; if (++nxtw.py == a.tiles_y) { nxtw.py = 0; ++nxtw.ng; } }
    };
    auto req_dy = [&](int i, int buf) {
        const bool ok = ((rq_dkk - dgeo[i]) & GUARD) == rq_want;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_void*)(smem + buf * pair_bytes + ((wave + 8 * i) << 10)), 16,
                                                 ok ? rq_dbase + drel[i] : OOB, 0, 0, 0);
    };
    auto req_x = [&](int i, int buf) {
        const bool ok = ((xgeo[i] - rq_xlo) & (rq_xkk - xgeo[i]) & GUARD) == rq_want;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(smem + buf * pair_bytes + (BMP << 7) + (xslot[i] << 10)), 16,
                                                 ok ? rq_xbase + xrel[i] : OOB, 0, 0, 0);
    };
    auto req_all = [&](int buf, bool live) {
        req_begin(live);
#pragma unroll
        for (int i = 0; i < 4; ++i) req_dy(i, buf);
#pragma unroll
        for (int i = 0; i < 6; ++i) req_x(i, buf);
    };

    // lane-constant parts of the transposed-fragment addresses (g = lane >> 4: channel block g & 1, k half g >> 1;
    // q = (lane >> 2) & 3: pixel of the 4-row block this lane addresses; pc = lane & 3: 4-channel column group)
    const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const int ca_sw = (((co_sub * 4 + (g & 1) * 2 + (pc >> 1)) ^ (((q >> 1) & 1) << 2)) << 4) + 8 * (pc & 1);
    int cb_sw[3];
#pragma unroll
    for (int kx = 0; kx < 3; ++kx)
        cb_sw[kx] = (((ci_sub * 4 + (g & 1) * 2 + (pc >> 1)) ^ ((((q + kx) >> 1) & 1) << 2)) << 4) + 8 * (pc & 1);

    // PAIRS: lane-constant byte offset of pair i inside a patch row group: tap 2i for the column block (g & 1) = 0, tap 2i+1 for
    // block 1 (pair 4's second half repeats tap 8 and is dropped by the flush); channels 0..15 for both
    int pairoff[5];
    if constexpr (PAIRS) {
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int tl = 2 * i + (g & 1) > 8 ? 8 : 2 * i + (g & 1);
            const int kx = tl % 3;
            pairoff[i] = ((tl / 3) * PW + kx) * RS + ((((pc >> 1)) ^ ((((q + kx) >> 1) & 1) << 2)) << 4) + 8 * (pc & 1);
        }
    }

    if constexpr (REQ >= 1) req_all(0, mb_begin < mb_end);
    else if (mb_begin < mb_end) issue(mb_begin, 0);
    for (int mbi = mb_begin; mbi < mb_end; ++mbi) {
        const int buf = (mbi - mb_begin) & 1;
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // this wave's pieces of m-block mbi; its reads of mbi-1
        __builtin_amdgcn_s_barrier();                   // every wave's pieces are in LDS; the other buffer pair is free
        if constexpr (REQ == 1) req_all(buf ^ 1, mbi + 1 < mb_end);
        else if constexpr (REQ == 2) req_begin(mbi + 1 < mb_end);
        else if (mbi + 1 < mb_end) issue(mbi + 1, buf ^ 1);  // lands under this m-block's MFMAs
        const char* dyt = smem + buf * pair_bytes;
        const char* patch = dyt + (BMP << 7);
        // k-blocks of 16 pixels; the X fragment of tap t+1 is read BEFORE the MFMA of tap t, order pinned: with one dY fragment
        // per 9 MFMAs every MFMA otherwise waits for the two transposed reads hipcc sinks directly in front of it
        // (interleaved A/B, tools/ab_conv.py: +8-10 % on layer1-4; a depth of 2, or reading ahead across k-block boundaries,
        // adds nothing; an explicit cross-k-block prefetch of the dY and first X fragment measured 2-3 % SLOWER)
        auto frag_addr = [&](int kb, int* pA, int* ppB) {
            const int p0 = kb << 4;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int p = p0 + 8 * (g >> 1) + 4 * tt + q;
                pA[tt] = p;
                const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                ppB[tt] = (pn * PH + my) * PW + mx;
            }
        };
        auto read_a = [&](const int* pA) {
            bf16x8 fa;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const s16x4 r = tr_read(dyt + pA[tt] * RS + ca_sw);
                const bf16x4 rb = __builtin_bit_cast(bf16x4, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[4 * tt + i] = rb[i];
            }
            return fa;
        };
        auto read_b = [&](const int* ppB, int tap) {
            const int tapoff = ((tap / 3) * PW + (tap % 3)) * RS;                 // wave-uniform
            bf16x8 fb;
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const s16x4 r = tr_read(patch + ppB[tt] * RS + tapoff + cb_sw[tap % 3]);
                const bf16x4 rb = __builtin_bit_cast(bf16x4, r);
#pragma unroll
                for (int i = 0; i < 4; ++i) fb[4 * tt + i] = rb[i];
            }
            return fb;
        };
#pragma unroll(REQ == 2 ? 8 : 2)
        for (int kb = k_sub; kb < (BMP >> 4); kb += WK) {
            if constexpr (REQ == 2) {
                // k-block it of 8: the ten requests of the next m-block leave in the first four (two to three each), so that the
                // last of them still has half an m-block to land
                const int it = (kb - k_sub) / WK;
                if (it < 4) req_dy(it, buf ^ 1);
                if (it < 2) { req_x(2 * it, buf ^ 1); req_x(2 * it + 1, buf ^ 1); }
                else if (it < 4) req_x(it + 2, buf ^ 1);
            }
            int pA[2], ppB[2];
            frag_addr(kb, pA, ppB);
            if (PIN >= 2) {   // tools build (-DPMOE_STAMP) only, timing, WRONG results: 2 = one X fragment per k-block, 3 = no LDS reads at all
                              // (layer4: 0.306 ms as shipped, 0.274 / 0.209 ms = 1481 TFLOP/s: what a stream of 32x32x16 bf16 MFMAs
                              // sustains on this part under its power limit)
                bf16x8 fa = PIN == 2 ? read_a(pA) : __builtin_bit_cast(bf16x8, v4i{kb, lane, kb, lane});
                bf16x8 fb = PIN == 2 ? read_b(ppB, 0) : __builtin_bit_cast(bf16x8, v4i{lane, kb, lane, kb});
#pragma unroll
                for (int tap = 0; tap < TAPS; ++tap) acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[tap], 0, 0, 0);
                continue;
            }
            if constexpr (PAIRS) {
                const bf16x8 fa = read_a(pA);
                auto read_pair = [&](int i) {
                    bf16x8 fb;
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) {
                        const s16x4 r = tr_read(patch + ppB[tt] * RS + pairoff[i]);
                        const bf16x4 rb = __builtin_bit_cast(bf16x4, r);
#pragma unroll
                        for (int j = 0; j < 4; ++j) fb[4 * tt + j] = rb[j];
                    }
                    return fb;
                };
                bf16x8 fbp[2];
                fbp[0] = read_pair(0);
#pragma unroll
                for (int i = 0; i < 5; ++i) {
                    if (i + 1 < 5) fbp[(i + 1) & 1] = read_pair(i + 1);
                    __builtin_amdgcn_sched_barrier(0);
                    acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbp[i & 1], acc[i], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                }
                continue;
            }
            const bf16x8 fa = read_a(pA);
            bf16x8 fbq[2];                                // static double buffer: tap t lives in fbq[t & 1]
            fbq[0] = read_b(ppB, 0);
#pragma unroll
            for (int tap = 0; tap < TAPS; ++tap) {
                if (tap + 1 < TAPS) fbq[(tap + 1) & 1] = read_b(ppB, tap + 1);
                if (PIN == 1) __builtin_amdgcn_sched_barrier(0);       // (PIN = 0: A/B switch, hipcc places the reads)
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fbq[tap & 1], acc[tap], 0, 0, 0);
                if (PIN == 1) __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush (as in conv_wgrad_kernel): fixed-order fold of the two pixel halves through LDS, plain stores
    constexpr int TPR = 3, TILE_WAVES = WCO * WCI;
    float* red = reinterpret_cast<float*>(smem);
    const int tw = wave % TILE_WAVES;
    const int cin = ci0 + ci_sub * 32 + l31;
    const size_t slab = a.per_image ? (size_t)e * a.ipe + split : (size_t)split * gridDim.z + e;
    float* dst = (a.per_image || nsplit == 1) ? a.dw : a.part;
    if constexpr (PAIRS) {
        // accumulator i, column c: tap 2i + (c >> 4), input channel c & 15.  One pair per round through LDS (fixed-order fold of the
        // 4 pixel quarters); the 48 columns past the 16 channels are written as zeros
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            __syncthreads();
            if (k_sub > 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) red[(((k_sub - 1) * TILE_WAVES + tw) * 16 + r) * 64 + lane] = acc[i][r];
            }
            __syncthreads();
            const int tap = 2 * i + (l31 >> 4);
            if (k_sub == 0 && tap < TAPS) {
                float* base = dst + ((slab * TAPS + tap) * a.CoutP) * a.CinP;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[i][r];
#pragma unroll
                    for (int k = 1; k < WK; ++k) v += red[(((k - 1) * TILE_WAVES + tw) * 16 + r) * 64 + lane];
                    const int cout = co0 + co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    float* row = base + (size_t)cout * a.CinP + ci0 + (l31 & 15);
                    row[0] = v;
                    for (int z = 16; ci0 + (l31 & 15) + z < a.CinP; z += 16) row[z] = 0.f;
                }
            }
        }
        return;
    }
#pragma unroll
    for (int t0 = 0; t0 < TAPS; t0 += TPR) {
        __syncthreads();
        if (k_sub > 0) {
#pragma unroll
            for (int tp = 0; tp < TPR; ++tp)
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    red[((((k_sub - 1) * TPR + tp) * TILE_WAVES + tw) * 16 + r) * 64 + lane] = acc[t0 + tp][r];
        }
        __syncthreads();
        if (k_sub == 0) {
#pragma unroll
            for (int tp = 0; tp < TPR; ++tp) {
                float* base = dst + ((slab * TAPS + t0 + tp) * a.CoutP) * a.CinP;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float v = acc[t0 + tp][r];
#pragma unroll
                    for (int k = 1; k < WK; ++k) v += red[((((k - 1) * TPR + tp) * TILE_WAVES + tw) * 16 + r) * 64 + lane];
                    const int cout = co0 + co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    base[(size_t)cout * a.CinP + cin] = v;
                    if (WCI == 1 && cin + 32 < a.CinP) base[(size_t)cout * a.CinP + cin + 32] = 0.f;   // columns nobody computes
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: conv_wgrad_dma_kernel with the roles SEPARATED -- four accumulating waves, ONE per SIMD, and a
// fifth wave that issues every LDS-DMA request of the workgroup and nothing else.
// Why: the 8-wave kernel above sits at 0.46 MFMA utilisation although neither the LDS array (55 % busy) nor HBM limits it.  Its
// k-loop reads the transposed X fragment of tap t+1 one MFMA (32 cycles, 64 with the SIMD partner's turn) ahead of its use, and
// a ds_read_b64_tr_b16 under this load returns after ~130-200 cycles: every MFMA waits for its operand, and the partner wave --
// in the same phase of the same barrier-synchronous loop -- waits for ITS operand at the same time.  Deeper read-ahead did not
// fit: two waves per SIMD leave 256 registers, 144 of them accumulators.  And every wave spends ~1300 of an m-block's ~10 000
// cycles at the texture path's request port, where it cannot issue MFMAs.
// Here a wave owns its SIMD: same 32 x 32 x 9-tap accumulators (144 registers), the fragments of a WHOLE k-block (2 dY + 18 X
// transposed reads = 40 registers) double-buffered, the 20 reads of k-block kb+1 issued between the 9 MFMAs of k-block kb --
// a read-ahead of 288 MFMA cycles instead of 32 -- and all 16 k-blocks of an m-block on one wave (no pixel split: the flush
// needs no LDS fold).  The accumulating waves have no vector-memory instruction at all; the producer wave requests m-block
// i+1 right behind the barrier that publishes m-block i and waits for its own vmcnt(0) before the next one.  320 threads:
// five waves, <= 2 per SIMD, 256 registers each.  Same LDS image (swizzles, tile geometry, zero fill by the buffer range
// check), same accumulation order per output element as conv_wgrad_dma_kernel<1, 2> with its pixel halves folded -- but the
// halves are now one chain, so results agree to f32 summation order, not bit for bit.  PMOE_WGRAD_V2=0: A/B switch.
// Measured, first form (ONE request-only wave, 320 threads): 12-19 % SLOWER than the 8-wave kernel on every layer
// (profiles/r04_kernel_ab.log) -- a wave issues an LDS-DMA request every 60-150 cycles (MI355X_MICROARCH.md, cycle constants), so
// the 80 requests of an m-block took one wave 6-12 k cycles against the 4.6 k of its MFMAs: the request STREAM of a workgroup needs
// several issuing waves even though the texture path accepts a request per ~16 cycles.  Second form (this one): one accumulating
// wave AND one request-only wave per SIMD (512 threads, pieces striped over the four producers).
template <int LTW, int AHEAD = 5>
__global__ void __launch_bounds__(512) conv_wgrad_dma2_kernel(const WgradArgs a, const int magic_pw, const int magic_ph) {
    constexpr int TAPS = 9, RS = 128, CKW = 64, TW = 1 << LTW, PW = TW + 2, BMP = 256;
    typedef __attribute__((address_space(3))) void lds_void;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;

    const int e = blockIdx.z;
    const int n_ci_blk = (a.Cin + CKW - 1) / CKW;
    const int npairs = n_ci_blk * ((a.Cout + CKW - 1) / CKW);
    const unsigned flat = xcd_remap(blockIdx.x, gridDim.x);
    const int nsplit = (int)(gridDim.x / npairs);
    const int pair = (int)(flat % npairs), split = (int)(flat / npairs);
    const int cob = pair / n_ci_blk, cib = pair % n_ci_blk;
    const int co0 = cob * CKW, ci0 = cib * CKW;

    const int lTH = a.lTH, TH = 1 << lTH, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int XB = NPIECE << 10;
    const int pair_bytes = (BMP << 7) + XB;
    const int mbpe = a.n_groups * a.tiles_y * a.tiles_x;
    const int mb_begin = split * a.mb_per_wg;
    int mb_end = mb_begin + a.mb_per_wg;
    if (mb_end > mbpe) mb_end = mbpe;

    if (wave >= 4) {
        // ---------------- producer waves: the request stream of conv_wgrad_dma_kernel (REQ = 1 geometry); producer pw takes the
        // pieces pw, pw + 4, ... (8 of the dY tile's 32, up to 12 of the X patch's 48)
        const int pw = wave - 4;
        const bf16* xb = (const bf16*)a.x + (a.x_shared ? (size_t)0 : (size_t)e * a.ipe * a.H * a.W * a.x_ld) + a.x_coff + ci0;
        const bf16* dyb = (const bf16*)a.dy + (size_t)e * a.ipe * a.Ho * a.Wo * a.dy_ld + a.dy_coff + co0;
        const __amdgpu_buffer_rsrc_t rs_x = __builtin_amdgcn_make_buffer_rsrc(
            (void*)xb, (short)0, (int)(((long long)a.ipe * a.H * a.W * a.x_ld - a.x_coff - ci0) * 2), 0x00020000);
        const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(
            (void*)dyb, (short)0, (int)(((long long)a.ipe * a.Ho * a.Wo * a.dy_ld - a.dy_coff - co0) * 2), 0x00020000);
        constexpr int OOB = 0x7ff80000;
        constexpr int GUARD = (1 << 9) | (1 << 19) | (1 << 29);
        int drel[8], dgeo[8], xrel[12], xgeo[12];
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int p = ((pw + 4 * i) << 3) + (lane >> 3);
            const int mx = p & (TW - 1), my = (p >> LTW) & (TH - 1), pn = p >> (LTW + lTH);
            const int j = (lane & 7) ^ (((p >> 1) & 1) << 2);
            drel[i] = ((((pn * a.Ho + my) * a.Wo + mx) * a.dy_ld) << 1) + (j << 4);
            dgeo[i] = (co0 + j * 8 < a.Cout ? (pn << 20) | (my << 10) | mx : (511 << 20)) | GUARD;
        }
        // (a piece index beyond the patch's last requests the LAST piece again -- same bytes to the same place -- so that the
        //  request stream is branch-free straight-line code)
        int xslot[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) {
            xslot[i] = pw + 4 * i < NPIECE ? pw + 4 * i : NPIECE - 1;
            const int pp = (xslot[i] << 3) + (lane >> 3);
            const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
            const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
            const int j = (lane & 7) ^ (((px >> 1) & 1) << 2);
            xrel[i] = ((((pn * a.H + prow) * a.W + px) * a.x_ld) << 1) + (j << 4);
            xgeo[i] = (pp < NPIX && ci0 + j * 8 < a.Cin ? (pn << 20) | (prow << 10) | px : (511 << 20)) | GUARD;
        }
        int wpx, wpy, wng;
        {
            int t = mb_begin;
            wpx = t % a.tiles_x; t /= a.tiles_x;
            wpy = t % a.tiles_y; wng = t / a.tiles_y;
        }
        auto imin = [](int x, int y) { return x < y ? x : y; };
        auto req_all = [&](int buf) {                    // every piece of the m-block the walk points at, then advance the walk
            const int n0 = wng * a.TN, oy0 = wpy * TH, ox0 = wpx * TW, Y0 = oy0 - a.pad, X0 = ox0 - a.pad;
            const int nmax = imin(a.ipe - n0, 511) - 1;
            const int dbase = (((n0 * a.Ho + oy0) * a.Wo + ox0) * a.dy_ld) << 1;
            const int dkk = (((nmax << 20) | ((imin(TH, a.Ho - oy0) - 1) << 10) | (imin(TW, a.Wo - ox0) - 1)) | GUARD) + GUARD;
            const int xbase = (((n0 * a.H + Y0) * a.W + X0) * a.x_ld) << 1;
            const int xlo = ((Y0 < 0 ? -Y0 : 0) << 10) | (X0 < 0 ? -X0 : 0);
            const int xkk = (((nmax << 20) | ((imin(PH, a.H - Y0) - 1) << 10) | (imin(PW, a.W - X0) - 1)) | GUARD) + GUARD;
            char* base = smem + buf * pair_bytes;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const bool ok = ((dkk - dgeo[i]) & GUARD) == GUARD;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_dy, (lds_void*)(base + ((pw + 4 * i) << 10)), 16, ok ? dbase + drel[i] : OOB,
                                                         0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < 12; ++i) {
                const bool ok = ((xgeo[i] - xlo) & (xkk - xgeo[i]) & GUARD) == GUARD;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lds_void*)(base + (BMP << 7) + (xslot[i] << 10)), 16,
                                                         ok ? xbase + xrel[i] : OOB, 0, 0, 0);
            }
            if (++wpx == a.tiles_x) { wpx = 0; if (++wpy == a.tiles_y) { wpy = 0; ++wng; } }
        };
        // iteration it: publish m-block it - 1 (wait for its requests, barrier), then request m-block it into the other buffer pair
        const int nmb = mb_end - mb_begin;
#pragma unroll 1
        for (int it = 0; it <= nmb; ++it) {
            if (it > 0) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();            // m-block it - 1 is in LDS; the other buffer pair is free
            }
            if (it < nmb) req_all(it & 1);
        }
        return;
    }

    // ---------------- accumulating waves: 2 (cout halves) x 2 (cin halves), all 16 k-blocks of every m-block
    const int co_sub = wave >> 1, ci_sub = wave & 1;
    f32x16 acc[TAPS];
#pragma unroll
    for (int t = 0; t < TAPS; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;
    // lane-constant parts of the transposed-fragment addresses (as conv_wgrad_dma_kernel): g = lane >> 4 (channel block g & 1,
    // k half g >> 1), q = (lane >> 2) & 3 (pixel of the 4-row block), pc = lane & 3 (4-channel column group)
    const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    int la[2], lb[2][3];
#pragma unroll
    for (int tt = 0; tt < 2; ++tt) {
        const int pl = 8 * (g >> 1) + 4 * tt + q;                     // pixel inside the 16-pixel k-block = its column offset
        la[tt] = pl * RS + (((co_sub * 4 + (g & 1) * 2 + (pc >> 1)) ^ (((q >> 1) & 1) << 2)) << 4) + 8 * (pc & 1);
#pragma unroll
        for (int kx = 0; kx < 3; ++kx)
            lb[tt][kx] = (pl + kx) * RS + (((ci_sub * 4 + (g & 1) * 2 + (pc >> 1)) ^ ((((q + kx) >> 1) & 1) << 2)) << 4) + 8 * (pc & 1);
    }
    // fragment registers: the dY fragment (A) of the current and the next k-block, and a ring of SIX X fragments (B) that runs
    // AHEAD taps ahead of the MFMA stream (u = 9 kb + tap indexes the stream; slot u % 6; the 18-tap loop body keeps every
    // index static).  A whole k-block of read-ahead (2 x 40 registers) spilled: 144 accumulators leave ~100 registers.
    constexpr int RINGB = AHEAD < 6 ? 6 : 9;             // (divides the 18-tap body, > AHEAD)
    s16x4 fa_[2][2], fb_[RINGB][2];
    auto kb_brow = [&](const char* patch, int kb) {      // address of (patch row, column base) of k-block kb
        const int r = LTW == 5 ? kb >> 1 : kb;
        const int colb = LTW == 5 ? (kb & 1) * 16 : 0;
        const int my = r & (TH - 1), pn = r >> lTH;
        return patch + ((pn * PH + my) * PW + colb) * RS;
    };
    auto frag8 = [](const s16x4& x, const s16x4& y) {
        const bf16x4 u = __builtin_bit_cast(bf16x4, x), v = __builtin_bit_cast(bf16x4, y);
        return bf16x8{u[0], u[1], u[2], u[3], v[0], v[1], v[2], v[3]};
    };

    for (int mbi = mb_begin; mbi < mb_end; ++mbi) {
        const int buf = (mbi - mb_begin) & 1;
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (all fragment reads of the previous m-block were consumed)
        __builtin_amdgcn_s_barrier();
        const char* dyt = smem + buf * pair_bytes;
        const char* patch = dyt + (BMP << 7);
        // prologue: A(0) and the first AHEAD taps of k-block 0
        {
            const char* br = kb_brow(patch, 0);
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) fa_[0][tt] = tr_read(dyt + la[tt]);
#pragma unroll
            for (int u = 0; u < AHEAD; ++u)
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) fb_[u % RINGB][tt] = tr_read(br + ((u / 3) * PW) * RS + lb[tt][u % 3]);
        }
#pragma unroll 1
        for (int kb0 = 0; kb0 < 16; kb0 += 2) {
            // rows of the three k-blocks this body touches (the read-ahead reaches at most into k-block kb0 + 2)
            const char* brow[3] = {kb_brow(patch, kb0), kb_brow(patch, kb0 + 1), kb_brow(patch, kb0 + 2 < 16 ? kb0 + 2 : 0)};
#pragma unroll
            for (int v = 0; v < 18; ++v) {               // v = 9 (kb - kb0) + tap: the ring slot (9 kb0 + v) % 6 = v % 6 (kb0 even)
                const int kbl = v / 9, tap = v % 9;
                const bf16x8 fa = frag8(fa_[kbl][0], fa_[kbl][1]);
                acc[tap] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, frag8(fb_[v % RINGB][0], fb_[v % RINGB][1]), acc[tap], 0, 0, 0);
                // reads issued behind this MFMA: the X fragment AHEAD taps on; at tap 3 the next k-block's dY fragment
                const int w = v + AHEAD, wk = w / 9, wt = w % 9;
                if (kb0 + wk < 16) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt)
                        fb_[w % RINGB][tt] = tr_read(brow[wk] + ((wt / 3) * PW) * RS + lb[tt][wt % 3]);
                }
                if (tap == 3 && kb0 + kbl + 1 < 16) {
#pragma unroll
                    for (int tt = 0; tt < 2; ++tt) fa_[(kbl + 1) & 1][tt] = tr_read(dyt + (kb0 + kbl + 1) * (16 * RS) + la[tt]);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // flush: D[row = cout][col = cin], lanes along cin; one wave owns a whole (cout half, cin half) tile: plain stores, no fold
    const int cin = ci0 + ci_sub * 32 + l31;
    const size_t slab = a.per_image ? (size_t)e * a.ipe + split : (size_t)split * gridDim.z + e;
    float* dst = (a.per_image || nsplit == 1) ? a.dw : a.part;
#pragma unroll
    for (int tap = 0; tap < TAPS; ++tap) {
        float* base = dst + ((slab * TAPS + tap) * a.CoutP) * a.CinP;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int cout = co0 + co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            base[(size_t)cout * a.CinP + cin] = acc[tap][r];
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: the stem's first BatchNorm backward WITHOUT its apply pass.  conv1 (12 -> 16 input channels, 64 outputs, full
// resolution) is followed by BatchNorm + ReLU (blocks/basics.py:113-120); the gradient dz1 of its output feeds exactly one
// consumer -- this per-image filter gradient (the ECA-folded stem needs no data gradient into the frames) -- so the 2.15 GB
// tensor never has to exist: the kernel takes g (the ReLU-masked gradient w.r.t. the BatchNorm output, left by conv2's
// PMOE_RES_DBN data gradient) and z (conv1's output), and evaluates
//     dz = g * A + ((z - mean) * Bx + K),   A = gamma*invstd,  Bx = -A * invstd * c2,  K = -A * c1     (bn_bwd_apply_kernel)
// in registers between the global loads and the LDS store of the dY tile -- same arithmetic, same bf16 rounding as the
// stand-alone pass, so the operand the MFMAs see is bit-identical to the tensor that pass would have written.  HBM: reads
// g + z + the shared frames (4.4 GB at the headline shape) instead of read g, z + write dz (6.4 GB) and read dz again (2.2 GB).
// Structure: the register-staged kernel above (issue early / write late, two barriers per m-block) in the narrow PAIRS
// layout of conv_wgrad_dma_kernel<1, 1, true>: 8 waves = 2 (cout halves) x 4 (pixel quarters), two taps x 16 channels per
// MFMA column block (5 MFMAs per k-block), X patch rows of 32 bytes (16 channels: nothing else is staged).
struct BnBwdFuse {
    const void* z;         // [N][H][W][z_ld] bf16: the BatchNorm's input (conv1's output), geometry of dy
    const float* coef;     // [4][E][C]: mean, invstd, gamma*invstd, beta (engine._bn_coeffs)
    const float* c1;       // [E][C]: mean of g over the expert's rows       (pmoe_bn_bwd_finalize)
    const float* c2;       // [E][C]: mean of g * xhat
    int z_ld, C;
};

__global__ void __launch_bounds__(512) conv_wgrad_bnbwd_kernel(const WgradArgs a, const BnBwdFuse f) {
    constexpr int TAPS = 9, RS = 192, XRS = 32, WK = 4, TILE_WAVES = 2;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int co_sub = wave & 1, k_sub = wave >> 1;
    const int e = blockIdx.z, img = blockIdx.x;                       // one workgroup walks exactly one image
    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2, NPIX = PH * PW;
    constexpr int BMP = 256;
    char* dyt = smem;                                                 // [256][192 B]
    char* patch = smem + BMP * RS;                                    // [NPIX][32 B]

    f32x16 acc[5];
#pragma unroll
    for (int t = 0; t < 5; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) acc[t][k] = 0.f;

    const size_t n_out = (size_t)e * a.ipe + img;
    const bf16* dy = (const bf16*)a.dy + n_out * a.Ho * a.Wo * a.dy_ld + a.dy_coff;
    const bf16* zz = (const bf16*)f.z + n_out * a.Ho * a.Wo * f.z_ld;
    const bf16* x = (const bf16*)a.x + (a.x_shared ? (size_t)img : n_out) * a.H * a.W * a.x_ld + a.x_coff;

    // per-thread constants: its 8 channels of the dY tile (chunk dyj of every pixel row it stages)
    const int dyj = tid & 7;
    float A[8], Bx[8], K[8], mu[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int c = dyj * 8 + k;
        const bool ok = c < f.C;
        const size_t ec = (size_t)e * f.C + (ok ? c : 0);
        const size_t EC = (size_t)gridDim.z * f.C;
        const float sc = ok ? f.coef[2 * EC + ec] : 0.f, is = ok ? f.coef[EC + ec] : 0.f;
        mu[k] = ok ? f.coef[ec] : 0.f;
        A[k] = sc;
        Bx[k] = ok ? -sc * is * f.c2[ec] : 0.f;
        K[k] = ok ? -sc * f.c1[ec] : 0.f;
    }
    const bool dy_cok = dyj * 8 < a.Cout;
    // X patch items of this thread: (pixel, 16-byte half of its 32-byte row); the decode is tile-invariant
    int xiy[2], xix[2], xoff[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int it = tid + u * 512, pp = it >> 1;
        xiy[u] = pp < NPIX ? pp / PW : -100000;
        xix[u] = pp - (pp / PW) * PW;
        xoff[u] = pp * XRS + (it & 1) * 16;
    }
    const bool x_cok[2] = {0 < a.Cin, 8 < a.Cin};

    v4i dyv[4], zv[4], xv[2];
    unsigned live = 0;                                  // bit u: pixel u of this thread's staging rows lies inside the image
    auto issue = [&](int mbi) {
        const int px = mbi % a.tiles_x, py = mbi / a.tiles_x;
        const int oy0 = py * TH, ox0 = px * TW;
        live = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = (tid >> 3) + u * 64;
            const int mx = p & (TW - 1), my = p >> lTW;
            const int oy = oy0 + my, ox = ox0 + mx;
            dyv[u] = zv[u] = v4i{0, 0, 0, 0};
            if (dy_cok && oy < a.Ho && ox < a.Wo) {
                live |= 1u << u;
                const size_t pix = (size_t)oy * a.Wo + ox;
                dyv[u] = ldg16(dy + pix * a.dy_ld + dyj * 8);
                zv[u] = ldg16(zz + pix * f.z_ld + dyj * 8);
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int Y = oy0 - a.pad + xiy[u], X = ox0 - a.pad + xix[u];
            xv[u] = v4i{0, 0, 0, 0};
            if (x_cok[(tid + u * 512) & 1] && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W)
                xv[u] = ldg16(x + ((size_t)Y * a.W + X) * a.x_ld + ((tid + u * 512) & 1) * 8);
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = (tid >> 3) + u * 64;
            float gv[8], xz[8], o[8];
            unpack16<bf16>(dyv[u], gv);
            unpack16<bf16>(zv[u], xz);
#pragma unroll
            for (int k = 0; k < 8; ++k) o[k] = gv[k] * A[k] + ((xz[k] - mu[k]) * Bx[k] + K[k]);
            // pixels of a ragged tile beyond the image carry NO gradient (the affine map does not send g = z = 0 to 0, and
            // their left / upper neighbours' patch pixels are real)
            *reinterpret_cast<v4i*>(dyt + p * RS + (dyj << 4)) = (live >> u) & 1u ? pack16<bf16>(o) : v4i{0, 0, 0, 0};
        }
#pragma unroll
        for (int u = 0; u < 2; ++u)
            if (xiy[u] >= 0) *reinterpret_cast<v4i*>(patch + xoff[u]) = xv[u];
    };

    // lane-constant parts of the transposed-fragment addresses (conv_wgrad_kernel / the PAIRS layout of conv_wgrad_dma_kernel)
    const int g = lane >> 4, q = (lane >> 2) & 3, pc = lane & 3;
    const int ca = (co_sub * 32 + 16 * (g & 1) + 4 * pc) * 2;
    int pairoff[5];
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        const int tl = 2 * i + (g & 1) > 8 ? 8 : 2 * i + (g & 1);
        pairoff[i] = ((tl / 3) * PW + (tl % 3)) * XRS + 8 * pc;
    }

    const int nmb = a.tiles_y * a.tiles_x;
    issue(0);
    for (int mbi = 0; mbi < nmb; ++mbi) {
        __syncthreads();                                // the previous m-block's fragment reads are done
        commit();
        __syncthreads();
        if (mbi + 1 < nmb) issue(mbi + 1);              // in flight under this m-block's MFMAs
#pragma unroll
        for (int kb = k_sub; kb < (BMP >> 4); kb += WK) {
            const int p0 = kb << 4;
            bf16x8 fa;
            const char* bB[2];
#pragma unroll
            for (int tt = 0; tt < 2; ++tt) {
                const int p = p0 + 8 * (g >> 1) + 4 * tt + q;
                const int mx = p & (TW - 1), my = p >> lTW;
                bB[tt] = patch + (my * PW + mx) * XRS;
                const bf16x4 rb = __builtin_bit_cast(bf16x4, tr_read(dyt + p * RS + ca));
#pragma unroll
                for (int i = 0; i < 4; ++i) fa[4 * tt + i] = rb[i];
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                bf16x8 fb;
#pragma unroll
                for (int tt = 0; tt < 2; ++tt) {
                    const bf16x4 rb = __builtin_bit_cast(bf16x4, tr_read(bB[tt] + pairoff[i]));
#pragma unroll
                    for (int j = 0; j < 4; ++j) fb[4 * tt + j] = rb[j];
                }
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa, fb, acc[i], 0, 0, 0);
            }
        }
    }

    // flush: accumulator i, column c = tap 2i + (c >> 4), input channel c & 15; fixed-order fold of the 4 pixel quarters through
    // LDS, one pair per round; the 48 columns past the 16 channels are written as zeros (dw is [N][9][CoutP][CinP], overwritten)
    float* red = reinterpret_cast<float*>(smem);
    const size_t slab = n_out;
#pragma unroll
    for (int i = 0; i < 5; ++i) {
        __syncthreads();
        if (k_sub > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((k_sub - 1) * TILE_WAVES + co_sub) * 16 + r) * 64 + lane] = acc[i][r];
        }
        __syncthreads();
        const int tap = 2 * i + (l31 >> 4);
        if (k_sub == 0 && tap < TAPS) {
            float* base = a.dw + ((slab * TAPS + tap) * a.CoutP) * a.CinP;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][r];
#pragma unroll
                for (int k = 1; k < WK; ++k) v += red[(((k - 1) * TILE_WAVES + co_sub) * 16 + r) * 64 + lane];
                const int cout = co_sub * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float* row = base + (size_t)cout * a.CinP + (l31 & 15);
                row[0] = v;
                for (int z = 16; (l31 & 15) + z < a.CinP; z += 16) row[z] = 0.f;
            }
        }
    }
}

// per-image filter gradient of a <= 16-input-channel 3x3 stride-1 convolution whose output gradient is given as (g, z) of the
// BatchNorm + ReLU behind it (see the kernel).  plan == true: 0 if the descriptor fits the kernel, else an error code.
int conv_wgrad_bnbwd_launch(WgradArgs a, const void* z, int z_ld, const float* coef, const float* c1, const float* c2,
                            int dtype, hipStream_t st, bool plan) {
    if (dtype != PMOE_DT_BF16 || !a.per_image || a.ks != 3 || a.stride != 1 || a.pad != 1) return PMOE_ERR_UNSUPPORTED;
    if (a.Cin > 16 || a.Cin % 8 || a.Cout > 64 || a.Cout % 8 || a.CoutP != 64 || a.CinP != 64) return PMOE_ERR_UNSUPPORTED;
    if (a.Ho != a.H || a.Wo != a.W || a.N % a.ipe || (long long)a.H * a.W < 256) return PMOE_ERR_UNSUPPORTED;
    if (z_ld % 8 || z_ld < a.Cout || a.x_ld % 8 || a.dy_ld % 8) return PMOE_ERR_ARG;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.W); if (lTW > 5) lTW = 5;
    const int lTH = 8 - lTW;
    a.lTW = lTW; a.lTH = lTH; a.TN = 1; a.n_groups = a.ipe;
    a.tiles_y = (a.H + (1 << lTH) - 1) >> lTH;
    a.tiles_x = (a.W + (1 << lTW) - 1) >> lTW;
    const int NPIX = ((1 << lTW) + 2) * ((1 << lTH) + 2);
    if (NPIX * 2 > 1024) return PMOE_ERR_UNSUPPORTED;                 // two 16-byte patch items per thread
    if (plan) return 0;
    if (!z || !coef || !c1 || !c2 || !a.x || !a.dy || !a.dw) return PMOE_ERR_ARG;
    size_t smem = (size_t)256 * 192 + (size_t)NPIX * 32;
    if (smem < (size_t)3 * 2 * 4096) smem = (size_t)3 * 2 * 4096;     // fold room: (WK - 1) x 2 tile waves x 4 KiB
    HIP_RET((ensure_dyn_lds<conv_wgrad_bnbwd_kernel>(160 * 1024)));
    BnBwdFuse f{z, coef, c1, c2, z_ld, a.Cout};
    hipLaunchKernelGGL(conv_wgrad_bnbwd_kernel, dim3(a.ipe, 1, a.N / a.ipe), dim3(512), smem, st, a, f);
    return (int)hipGetLastError();
}

// dw[i] = sum over the K-split slabs, fixed order (slab s at part + s * total)
__global__ void __launch_bounds__(256) wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw,
                                                           const int nsplit, const long long total4) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total4) return;
    const f32x4* p = reinterpret_cast<const f32x4*>(part);
    f32x4 s = p[i];
    for (int k = 1; k < nsplit; ++k) {
        const f32x4 v = p[(long long)k * total4 + i];
        s[0] += v[0]; s[1] += v[1]; s[2] += v[2]; s[3] += v[3];
    }
    reinterpret_cast<f32x4*>(dw)[i] = s;
}

// round 3: the fold of the K-split slabs AND the layout change into the parameter's own [E][cout][cin][kh][kw] gradient in one
// launch (WgradArgs.grads): thread = (expert, cout, cin) reads its taps x nsplit values (lanes along cin: coalesced) and writes
// `taps` consecutive floats (consecutive threads: one contiguous block).  Replaces wgrad_reduce_kernel + the strided
// unpack_wgrad_kernel launch (221 MB read + written once less per step, 29 launches fewer; 150 in the stage-1 step).
template <int TAPS>
__global__ void __launch_bounds__(256) wgrad_fold_unpack_kernel(const float* __restrict__ slabs, float* __restrict__ g,
                                                                const int nsplit, const long long slab_stride, const int cout,
                                                                const int cin, const int CoutP, const int CinP) {
    // block = 64 consecutive (cout, cin) pairs x 4 tap groups (a wave = one tap group: its 64 lanes read 64 consecutive input
    // channels of one slab row, 256 contiguous bytes); the 64 x TAPS results leave through LDS as one contiguous run
    const int e = blockIdx.y;
    const long long n = (long long)cout * cin;
    const int pr = threadIdx.x & 63, tg = threadIdx.x >> 6;
    __shared__ float sm[64 * TAPS];
    for (long long i0 = (long long)blockIdx.x * 64; i0 < n; i0 += (long long)gridDim.x * 64) {
        const long long i = i0 + pr;
        if (i < n) {
            const int ci = (int)(i % cin), co = (int)(i / cin);
            for (int t = tg; t < TAPS; t += 4) {
                const float* p = slabs + (((size_t)e * TAPS + t) * CoutP + co) * CinP + ci;
                float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;     // (fixed association: bit-reproducible)
                int k = 0;
                for (; k + 4 <= nsplit; k += 4) {
                    s0 += p[(size_t)k * slab_stride]; s1 += p[(size_t)(k + 1) * slab_stride];
                    s2 += p[(size_t)(k + 2) * slab_stride]; s3 += p[(size_t)(k + 3) * slab_stride];
                }
                for (; k < nsplit; ++k) s0 += p[(size_t)k * slab_stride];
                sm[pr * TAPS + t] = (s0 + s1) + (s2 + s3);
            }
        }
        __syncthreads();
        const long long cnt = (n - i0 < 64 ? n - i0 : 64) * TAPS;
        float* o = g + ((size_t)e * n + i0) * TAPS;
        for (int j = threadIdx.x; j < cnt; j += 256) o[j] = sm[j];
        __syncthreads();
    }
}

// the tail of a weight-gradient launch: slabs -> a.dw (wgrad_reduce_kernel) or, with a.grads, straight into the parameter layout
static int wgrad_finish(const WgradArgs& a, int E, int taps, int nsplit, hipStream_t st, bool force = false) {
    if (a.per_image || (a.defer_fold && !force)) return 0;
    const long long total = (long long)E * taps * a.CoutP * a.CinP;
    if (a.grads) {
        const float* src = nsplit > 1 ? a.part : a.dw;
        long long blocks = ((long long)a.cout_real * a.cin_real + 63) / 64;
        if (blocks > 4096) blocks = 4096;
        dim3 grid((unsigned)blocks, E);
        if (taps == 9)
            hipLaunchKernelGGL(wgrad_fold_unpack_kernel<9>, grid, dim3(256), 0, st, src, a.grads, nsplit, total, a.cout_real,
                               a.cin_real, a.CoutP, a.CinP);
        else
            hipLaunchKernelGGL(wgrad_fold_unpack_kernel<1>, grid, dim3(256), 0, st, src, a.grads, nsplit, total, a.cout_real,
                               a.cin_real, a.CoutP, a.CinP);
        return (int)hipGetLastError();
    }
    if (nsplit > 1) {
        hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total / 4 + 255) / 256)), dim3(256), 0, st, a.part, a.dw, nsplit,
                           total / 4);
        return (int)hipGetLastError();
    }
    return 0;
}

template <typename T, int TAPS, int MAXV> static int launch_wg(const WgradArgs& a, int E, size_t smem, hipStream_t st) {
    HIP_RET((ensure_dyn_lds<conv_wgrad_kernel<T, TAPS, MAXV>>(160 * 1024)));
    constexpr int CKW = 128 / (int)sizeof(T);
    const int mbpe = a.n_groups * a.tiles_y * a.tiles_x;
    const int nsplit = (mbpe + a.mb_per_wg - 1) / a.mb_per_wg;
    dim3 grid(nsplit * ((a.Cout + CKW - 1) / CKW) * ((a.Cin + CKW - 1) / CKW), 1, E), block(WgradCfg<T>::NW * 64, 1, 1);
    // room for the in-workgroup fold of the pixel-split waves: (WK-1) x TPR taps x tile waves x 4 KiB
    constexpr size_t fold = (size_t)(WgradCfg<T>::WK - 1) * (TAPS == 9 ? 3 : 1) * WgradCfg<T>::WCO * WgradCfg<T>::WCI * 4096;
    if (smem < fold) smem = fold;
    hipLaunchKernelGGL((conv_wgrad_kernel<T, TAPS, MAXV>), grid, block, smem, st, a);
    HIP_RET(hipGetLastError());
    return wgrad_finish(a, E, TAPS, nsplit, st);
}

// the separated-roles kernel serves the wide layout (>= 64 input channels) on 16- or 32-pixel-wide tiles
static bool wgrad_v2_ok(const WgradArgs& a, int lTW) {
    const char* ev = getenv("PMOE_WGRAD_V2");
    if (ev && !atoi(ev)) return false;
    const char* evx = getenv("PMOE_WGRAD_PIPE");
    if (evx) return false;                              // (the PIN / timing modes belong to the 8-wave kernel)
    return (lTW == 4 || lTW == 5) && a.ipe <= 511;
}

// plan == true: nothing is launched, *ws_floats receives the size of the K-split workspace the launch needs (0: none)
template <typename T> static int wgrad_dtype(WgradArgs a, hipStream_t st, bool plan, long long* ws_floats, int* code = nullptr) {
    constexpr int CKW = 128 / (int)sizeof(T);
    constexpr int VEh = 16 / (int)sizeof(T);
    if (a.Cin % VEh || a.Cout % VEh || a.CinP < a.Cin || a.CoutP < a.Cout) return PMOE_ERR_ARG;
    // the workgroup tiles must cover dw exactly (every element is WRITTEN, nothing is accumulated onto old contents)
    if (a.CinP != (a.Cin + CKW - 1) / CKW * CKW || a.CoutP != (a.Cout + CKW - 1) / CKW * CKW) return PMOE_ERR_ARG;
    if ((a.ks != 1 && a.ks != 3) || (a.stride != 1 && a.stride != 2) || a.N % a.ipe) return PMOE_ERR_ARG;
    const int E = a.N / a.ipe;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    for (int lBM = 8; lBM >= 6; --lBM) {
        int lTW = p2(a.Wo); if (lTW > 5) lTW = 5;
        int lTH = p2(a.Ho); if (lTH > lBM - lTW) lTH = lBM - lTW;
        if (lTW > lBM) { lTW = lBM; lTH = 0; }
        const int BMP = 1 << lBM;
        const int TN = BMP >> (lTW + lTH);
        const int TW = 1 << lTW, TH = 1 << lTH;
        const int lstride = a.ks == 1 ? 1 : a.stride;
        const int PW = (TW - 1) * lstride + a.ks, PH = (TH - 1) * lstride + a.ks;
        const int NPIX = TN * PH * PW;
        constexpr int NTHR = WgradCfg<T>::NW * 64;
        const int need = (NPIX * 8 + NTHR - 1) / NTHR;  // 16-byte patch loads per thread
        constexpr int M1 = 48 / WgradCfg<T>::NW, M2 = 80 / WgradCfg<T>::NW;   // patch loads per thread: 6|10 (8 waves), 12|20 (4 waves)
        if (need > M2) continue;
        const size_t smem = (size_t)BMP * 192 + (size_t)NPIX * 192;
        if (smem > 150 * 1024) continue;
        a.lTW = lTW; a.lTH = lTH; a.TN = TN;
        a.n_groups = (a.ipe + TN - 1) / TN;
        a.tiles_y = (a.Ho + TH - 1) / TH;
        a.tiles_x = (a.Wo + TW - 1) / TW;
        const int mbpe = a.n_groups * a.tiles_y * a.tiles_x;
        // K-split: one workgroup per CU is resident; ONE round of 256 workgroups measured best (each walks more m-blocks,
        // flushes its 9 x 64 x 64 accumulators once, no second-round tail): 2 rounds -10 %, 1.25-1.5 rounds -25..30 %.
        // PMOE_WGRAD_WGS overrides the target for A/B runs.
        const int pairs = ((a.Cout + CKW - 1) / CKW) * ((a.Cin + CKW - 1) / CKW) * E;
        static int target = 0;
        if (!target) { const char* ev = getenv("PMOE_WGRAD_WGS"); target = ev ? atoi(ev) : 256; }
        int want = (target + pairs - 1) / pairs;
        if (want < 1) want = 1;
        if (want > mbpe) want = mbpe;
        a.mb_per_wg = (mbpe + want - 1) / want;
        if (a.per_image) {                              // one workgroup walks exactly one image
            if (TN != 1) return PMOE_ERR_UNSUPPORTED;
            a.mb_per_wg = a.tiles_y * a.tiles_x;
        }
        { const char* ev = getenv("PMOE_WGRAD_SLICE_FASTEST"); a.slice_fastest = ev ? atoi(ev) : 0; }
        const int nsplit = (mbpe + a.mb_per_wg - 1) / a.mb_per_wg;
        const long long ws = (a.per_image || nsplit == 1) ? 0 : (long long)nsplit * E * a.ks * a.ks * a.CoutP * a.CinP;
        if (plan && !code) { *ws_floats = ws; return 0; }
        if (!plan && ws > 0 && (!a.part || a.part_floats < ws)) return PMOE_ERR_ARG;
        if constexpr (sizeof(T) == 2) {
            // dense 3x3 stride 1 in bf16: the LDS-DMA variant (PMOE_WGRAD_DMA=0: A/B switch back to register staging)
            const char* evd = getenv("PMOE_WGRAD_DMA");      // (read per launch: tools/ab_conv.py flips it inside one process)
            const int dma_on = evd ? atoi(evd) : 1;
            const long long xbytes = (long long)a.ipe * a.H * a.W * a.x_ld * 2, dybytes = (long long)a.ipe * a.Ho * a.Wo * a.dy_ld * 2;
            const int npiece = (NPIX + 7) / 8;
            const int mpw = 65536 / PW + 1, mph = 65536 / PH + 1;
            bool exact = true;
            for (int pp = 0; pp < npiece * 8 && exact; ++pp)
                exact = ((pp * mpw) >> 16) == pp / PW && ((((pp / PW) * mph) >> 16) == (pp / PW) / PH);
            if (dma_on && a.ks == 3 && a.stride == 1 && BMP == 256 && lTW >= 2 && npiece <= 48 && exact &&
                xbytes < 0x7ff00000ll && dybytes < 0x7ff00000ll && 2 * ((size_t)BMP * 128 + (size_t)npiece * 1024) <= 160 * 1024) {
                const char* evn = getenv("PMOE_WGRAD_NARROW");           // A/B: 0 = the 2 x 2 x 2 wave layout for every layer
                const bool narrow = a.Cin <= 32 && !(evn && !atoi(evn));
                if (plan) { *code = narrow ? 7109 : wgrad_v2_ok(a, lTW) ? 7309 : 7009; return 0; }   // conv_wgrad_dma_kernel<1, 1> / dma2 / <1, 2>
                size_t sm = 2 * ((size_t)BMP * 128 + (size_t)npiece * 1024);
                if (sm < 49152) sm = 49152;               // room for the flush's fold
                const int nsp = (mbpe + a.mb_per_wg - 1) / a.mb_per_wg;
                dim3 grid(nsp * pairs / E, 1, E), block(512, 1, 1);
                const char* evx = getenv("PMOE_WGRAD_PIPE");
                const char* evp = getenv("PMOE_WGRAD_PAIRS");            // A/B: 0 = one tap per MFMA column block
#ifdef PMOE_STAMP          // tools build only (tools/stamp_conv.py --build): timing modes with WRONG results, see the kernel's main loop
                if (evx && atoi(evx) == 2) {
                    HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<2>>(160 * 1024)));
                    hipLaunchKernelGGL(conv_wgrad_dma_kernel<2>, grid, block, sm, st, a, mpw, mph);
                } else if (evx && atoi(evx) == 3) {
                    HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<3>>(160 * 1024)));
                    hipLaunchKernelGGL(conv_wgrad_dma_kernel<3>, grid, block, sm, st, a, mpw, mph);
                } else
#endif
                if (narrow && a.Cin <= 16 && !(evp && !atoi(evp))) {
                    if (sm < (size_t)3 * 3 * 2 * 4096) sm = (size_t)3 * 3 * 2 * 4096;
                    HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<1, 1, true>>(160 * 1024)));
                    hipLaunchKernelGGL((conv_wgrad_dma_kernel<1, 1, true>), grid, block, sm, st, a, mpw, mph);
                } else if (narrow) {
                    if (sm < (size_t)3 * 3 * 2 * 4096) sm = (size_t)3 * 3 * 2 * 4096;      // fold room: (WK - 1) x 3 taps x 2 tile waves
                    HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<1, 1>>(160 * 1024)));
                    hipLaunchKernelGGL((conv_wgrad_dma_kernel<1, 1>), grid, block, sm, st, a, mpw, mph);
                } else if (evx && !atoi(evx)) {
                    HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<0>>(160 * 1024)));
                    hipLaunchKernelGGL(conv_wgrad_dma_kernel<0>, grid, block, sm, st, a, mpw, mph);
                } else if (wgrad_v2_ok(a, lTW)) {
                    // round 4: four accumulating waves (one per SIMD, k-block-deep fragment read-ahead) + one request-only wave
                    const char* eva = getenv("PMOE_WGRAD_AHEAD");            // A/B: fragment read-ahead in taps (5 | 6)
                    const int ahead = eva ? atoi(eva) : 5;
                    if (lTW == 5 && ahead == 6) {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma2_kernel<5, 6>>(160 * 1024)));
                        hipLaunchKernelGGL((conv_wgrad_dma2_kernel<5, 6>), grid, dim3(512, 1, 1), sm, st, a, mpw, mph);
                    } else if (lTW == 5) {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma2_kernel<5>>(160 * 1024)));
                        hipLaunchKernelGGL(conv_wgrad_dma2_kernel<5>, grid, dim3(512, 1, 1), sm, st, a, mpw, mph);
                    } else if (ahead == 6) {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma2_kernel<4, 6>>(160 * 1024)));
                        hipLaunchKernelGGL((conv_wgrad_dma2_kernel<4, 6>), grid, dim3(512, 1, 1), sm, st, a, mpw, mph);
                    } else {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma2_kernel<4>>(160 * 1024)));
                        hipLaunchKernelGGL(conv_wgrad_dma2_kernel<4>, grid, dim3(512, 1, 1), sm, st, a, mpw, mph);
                    }
                } else {
                    const char* evr = getenv("PMOE_WGRAD_REQ");             // A/B: 0 = round-2 request code, 1 = precomputed, at the top
                    const int req = a.ipe > 511 ? 0 : evr ? atoi(evr) : 1;
                    if (req == 2) {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<1, 2, false, 2>>(160 * 1024)));
                        hipLaunchKernelGGL((conv_wgrad_dma_kernel<1, 2, false, 2>), grid, block, sm, st, a, mpw, mph);
                    } else if (req == 1) {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<1, 2, false, 1>>(160 * 1024)));
                        hipLaunchKernelGGL((conv_wgrad_dma_kernel<1, 2, false, 1>), grid, block, sm, st, a, mpw, mph);
                    } else {
                        HIP_RET((ensure_dyn_lds<conv_wgrad_dma_kernel<1>>(160 * 1024)));
                        hipLaunchKernelGGL(conv_wgrad_dma_kernel<1>, grid, block, sm, st, a, mpw, mph);
                    }
                }
                HIP_RET(hipGetLastError());
                return wgrad_finish(a, E, 9, nsp, st);
            }
        }
        if (plan) { *code = 6000 + a.ks * a.ks * 100 + (need <= M1 ? M1 : M2); return 0; }      // conv_wgrad_kernel<T, taps, MAXV>
        if (need <= M1) return a.ks == 3 ? launch_wg<T, 9, M1>(a, E, smem, st) : launch_wg<T, 1, M1>(a, E, smem, st);
        return a.ks == 3 ? launch_wg<T, 9, M2>(a, E, smem, st) : launch_wg<T, 1, M2>(a, E, smem, st);
    }
    return PMOE_ERR_UNSUPPORTED;
}

// the deferred tail (WgradArgs.defer_fold): the K-split count is the one the launch used (same planning code)
int conv_wgrad_fold(const WgradArgs& a, int dtype, hipStream_t st) {
    if (a.per_image) return 0;
    long long ws = 0;
    const int rc = dtype == PMOE_DT_BF16 ? wgrad_dtype<bf16>(a, nullptr, true, &ws)
                 : dtype == PMOE_DT_F32 ? wgrad_dtype<float>(a, nullptr, true, &ws) : PMOE_ERR_ARG;
    if (rc) return rc;
    const int E = a.N / a.ipe, taps = a.ks * a.ks;
    const long long total = (long long)E * taps * a.CoutP * a.CinP;
    const int nsplit = ws > 0 ? (int)(ws / total) : 1;
    if (nsplit > 1 && (!a.part || a.part_floats < ws)) return PMOE_ERR_ARG;
    return wgrad_finish(a, E, taps, nsplit, st, true);
}

int conv_wgrad_launch(const WgradArgs& a, int dtype, hipStream_t st) {
    long long ws = 0;
    if (dtype == PMOE_DT_BF16) return wgrad_dtype<bf16>(a, st, false, &ws);
    if (dtype == PMOE_DT_F32) return wgrad_dtype<float>(a, st, false, &ws);
    return PMOE_ERR_ARG;
}

// which kernel a descriptor runs on: 7009 = conv_wgrad_dma_kernel (7109: its <= 32-input-channel wave layout); 6000 + taps * 100 + MAXV = conv_wgrad_kernel<T, taps, MAXV>
int conv_wgrad_plan(const WgradArgs& a, int dtype) {
    long long ws = 0;
    int code = 0;
    const int rc = dtype == PMOE_DT_BF16 ? wgrad_dtype<bf16>(a, nullptr, true, &ws, &code)
                 : dtype == PMOE_DT_F32 ? wgrad_dtype<float>(a, nullptr, true, &ws, &code) : PMOE_ERR_ARG;
    return rc ? rc : code;
}

long long conv_wgrad_ws_floats(const WgradArgs& a, int dtype) {
    long long ws = 0;
    const int rc = dtype == PMOE_DT_BF16 ? wgrad_dtype<bf16>(a, nullptr, true, &ws)
                 : dtype == PMOE_DT_F32 ? wgrad_dtype<float>(a, nullptr, true, &ws) : PMOE_ERR_ARG;
    return rc ? rc : ws;
}
