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
// waves of a SIMD -- within 1 %.  Round 3: TWO workgroups per CU (one patch buffer + a 2-slot weight ring = 75 KiB, <= 128 VGPRs,
// weight-tile lookahead of one tap, the next chunk's patch requested after the current chunk's last tap) so that one
// workgroup's prologue / epilogue / barrier waits are covered by the other's MFMAs: layer2 -10 %, layer3 -2 % / -9 %, layer4
// +3 % / 0 % (forward / data gradient) -- the exposed DMA round trip per chunk and the thin weight ring cost what the
// co-residence buys; dropped.
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

// MF16 (default for >= 256 input channels, conv_dma_launch): the same 64 x 64 wave tile on v_mfma_f32_16x16x32_bf16 -- 16 MFMAs of half the
// cycles per 32 channels instead of 4 per 16, identical LDS traffic (8 ds_read_b128 per 32 channels either way).  On this part
// the clock an MFMA-dense loop holds depends on the MFMA shape (MI355X_MICROARCH.md, DVFS give-back item 7).
//
// PROD (round 4): a NINTH wave that issues every LDS-DMA request of the workgroup and nothing else.  In the
// 8-wave kernel each wave issues up to three requests per tap right behind the barrier (60-185 cycles each at the texture
// path's port, MI355X_MICROARCH.md cycle constants) and cannot issue MFMAs meanwhile.  Here the eight accumulating waves have
// NO vector-memory instruction in the main loop at all -- barrier, fragment reads, 16 MFMAs -- and the producer wave, which
// owns the in-order vmcnt bookkeeping of the whole stream (all pieces are its own), publishes a tap with the same barrier:
// before barrier(tt) it waits until only the requests issued after W(tt) are outstanding (= the group it issued during tap
// tt-1: the patch pieces of the next chunk first, the weight tile of tap tt+1 last).  576 threads: three waves on one SIMD, so
// the kernel must fit 168 VGPRs (it needs ~150 without the z / residual prefetch) -- forward launches only (no res, no bias, no
// activation); the data-gradient modes keep the 8-wave kernel.  The producer ends after the last tap; s_barrier then counts
// the surviving eight waves (the epilogue's barriers).  PMOE_DMA_PRODUCER=0: A/B switch.
#define VMCASE(n) case n: VMCNT(n); break;
__device__ __forceinline__ void vm_wait_prod(int n) {     // s_waitcnt vmcnt(n), n <= 24 (a wave-uniform scalar)
    switch (n) {
        VMCASE(0) VMCASE(1) VMCASE(2) VMCASE(3) VMCASE(4) VMCASE(5) VMCASE(6) VMCASE(7) VMCASE(8) VMCASE(9) VMCASE(10) VMCASE(11)
        VMCASE(12) VMCASE(13) VMCASE(14) VMCASE(15) VMCASE(16) VMCASE(17) VMCASE(18) VMCASE(19) VMCASE(20) VMCASE(21) VMCASE(22)
        VMCASE(23)
        default: VMCNT(24); break;
    }
}

template <bool MF16, bool PROD = false>
__global__ void __launch_bounds__(PROD ? NTHR + 64 : NTHR, PROD ? 1 : 2) conv3x3_dma_kernel(const ConvArgs a_in, const int pbuf_bytes) {
    ConvArgs a = a_in;
    if constexpr (PROD) {                                // forward launches only (conv_dma_uses_producer): the epilogue's side-input /
        a.res_mode = PMOE_RES_NONE; a.res = nullptr;     // bias / activation / dropout branches fold away at compile time (168 VGPRs)
        a.bias = nullptr; a.act = PMOE_ACT_NONE; a.drop_p = 0.f; a.bn = nullptr;
    }
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
    if constexpr (PROD) {
        if (wave == 8) {
            // the producer wave: per-lane offsets of ALL pieces (48 patch + 16 weight-tile), then the request stream
            int pv[48], wv[16];
#pragma unroll
            for (int i = 0; i < 48; ++i) {
                const int pp = (i << 3) + (lane >> 3);
                const int jj = lane & 7;
                const int px = pp % PW;
                const int rowq = pp / PW;
                const int prow = rowq % PH, pn = rowq / PH;
                const int Y = oy0 - 1 + prow, X = ox0 - 1 + px;
                const bool ok = pp < NPIX && n0 + pn < n_end && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
                pv[i] = ok ? ((((pn * a.H + Y) * a.W + X) * a.in_ld) << 1) + ((jj ^ cswz(px)) << 4) : OOB;
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int row = (i << 3) + (lane >> 3);
                wv[i] = ((row * 9 * a.Cin) << 1) + (((lane & 7) ^ cswz(row)) << 4);
            }
            auto req_w = [&](int slot, int tap, int c0) {
#pragma unroll
                for (int i = 0; i < 16; ++i)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(wring + slot * WSLOT + (i << 10)), 16, wv[i],
                                                             (tap * a.Cin + c0) << 1, 0, 0);
            };
            const int nch = a.Cin / CK, TT = nch * 9;
            // (the prologue -- first patch, weight tiles of taps 0 and 1 -- is requested by the EIGHT accumulating waves, as in the
            //  8-wave kernel: one wave issues a request every 60-150 cycles, and 75 requests in a row from this wave alone cost the
            //  two-chunk layers (layer2) 9 % -- profiles/r04_kernel_ab.log; from the first barrier on the stream is this wave's)
            int after = 0;                               // own requests issued after W(tt): the group of the previous tap
            for (int ch = 0; ch < nch; ++ch) {
                const int c0 = ch * CK;
                const bool more = ch + 1 < nch;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int tt = ch * 9 + tap;
                    vm_wait_prod(after);
                    __builtin_amdgcn_s_barrier();
                    int n = 0;
                    if (tap >= 1 && tap <= 6 && more) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int idx = (tap - 1) * 8 + j;
                            if (idx < NPIECE) {
                                __builtin_amdgcn_raw_ptr_buffer_load_lds(
                                    rs_in, (lds_void*)(smem + ((ch + 1) & 1) * pbuf_bytes + (idx << 10)), 16, pv[idx], (c0 + CK) << 1, 0, 0);
                                ++n;
                            }
                        }
                    }
                    if (tt + 2 < TT) {
                        int ntap = tap + 2, nc0 = c0;
                        if (ntap >= 9) { ntap -= 9; nc0 += CK; }
                        req_w((tt + 2) & (RING - 1), ntap, nc0);
                        n += 16;
                    }
                    after = n;
                }
            }
            return;
        }
    }
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
    // MF16: 4 x 4 tiles of 16 x 16; A rows = couts wn*64 + nt*16 + (lane & 15), B columns = pixels wm*64 + mt*16 + (lane & 15),
    // k = 8 (lane >> 4) + j inside a 32-channel step -> 16-byte chunk ks*4 + (lane >> 4) of the 128-byte row
    f32x4 acc16[4][4];
    int pbase16[4], pcol16[4], aoff16[4][2];
    if constexpr (MF16) {
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int p = wm * 64 + mt * 16 + (lane & 15);
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            pbase16[mt] = ((pn * PH + my) * PW + mx) << LOG_RB;
            pcol16[mt] = mx;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = wn * 64 + nt * 16 + (lane & 15);
                aoff16[nt][ks] = row * RB + (((ks * 4 + (lane >> 4)) ^ cswz(row)) << 4);
            }
    }

    const int nchunks = a.Cin / CK;
    const int T = nchunks * 9;

    // PMOE_RES_DBN (data gradient into relu(BatchNorm(z)) with the BatchNorm-backward reductions in the epilogue): the 8 vectors
    // of z this thread will need in the read-out are requested at the first tap of the LAST chunk and arrive under its MFMAs
    // (read in the epilogue they cost +1.3 ms per step on the 9 launches: every wave waiting on its own loads between two
    // barriers).  Plain global loads share the in-order vmcnt counter with the DMA stream: the wait of the next tap lets them
    // stay in flight (+8); the one after retires them (hipcc may order them before OR after that tap's weight-tile request --
    // different address spaces -- so only the first wait can count on their position).
    // The same prefetch serves PMOE_RES_ADD (a data gradient accumulating into the gradient the identity branch left: layer2-4
    // `.1.conv1`), which then also takes the one-phase bf16 read-out (the sum is formed on the rounded accumulator, as the
    // resident-filter kernel always did: 0.46 -> 0.37 ms on layer2.1.conv1's data gradient).
    const bool zpre_on = !PROD && (a.res_mode == PMOE_RES_DBN || a.res_mode == PMOE_RES_ADD) && !a.bias && a.act == PMOE_ACT_NONE &&
                         a.drop_p == 0.f;
    v4i zpre[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) zpre[u] = v4i{0, 0, 0, 0};

    // ---- prologue: patch of chunk 0, weight tiles of taps 0 and 1 (PROD: too -- every wave issues its share, see the producer)
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
            if constexpr (PROD) {
                if (tt == 0) VMCNT(0);                   // this wave's share of the prologue (W(1) too: the producer does not track it)
            } else {
                if (tt + 1 >= T) VMCNT(0);
                else if (zpre_on && !more && tap == 1) VMCNT(10);      // W(tt) is older than both W(tt+1) and the 8 loads of tap 0
                else if (prev_piece) VMCNT(3);
                else VMCNT(2);
            }
            __builtin_amdgcn_s_barrier();                // every wave's share of tap tt (and of this chunk's patch) is in LDS
#ifdef PMOE_STAMP
            if (tt == 0) LAP(0)                          // prologue: descriptors, offsets, first patch + two weight tiles landed
#endif
            if constexpr (!PROD) {
            if (tap >= 1 && tap <= 6 && more && (tap - 1) < my_pieces) dma_patch(tap - 1, (ch + 1) & 1, c0 + CK);
            if (tt + 2 < T) {
                int ntap = tap + 2, nc0 = c0;
                if (ntap >= 9) { ntap -= 9; nc0 += CK; }
                dma_w((tt + 2) & (RING - 1), ntap, nc0);
            }
            }
            if (tap == 0 && zpre_on && !more) {
                const int zc = tid & 15, zr = tid >> 4;      // the read-out's (cc, pr): 16 channel vectors x 32 pixel rows of threads
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int p = zr + u * 32;
                    const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
                    const int n = n0 + pn, oy = oy0 + my, ox = ox0 + mx;
                    // EVERY wave issues all 8 loads (the counted waits below assume exactly 8): lanes without a pixel read the
                    // tensor's first vector, which the read-out never uses
                    const bool ok = cout0 + zc * VE < a.Cout && n < n_end && oy < a.Ho && ox < a.Wo;
                    const size_t off = ok ? (((size_t)n * a.Ho + oy) * a.Wo + ox) * a.res_ld + a.res_coff + cout0 + zc * VE : (size_t)0;
                    zpre[u] = ldg16((const bf16*)a.res + off);
                }
            }
            const char* wt = wring + (tt & (RING - 1)) * WSLOT;
            const int tapoff = ((tap / 3) * PW + (tap % 3)) << LOG_RB;
            if constexpr (MF16) {
                int bsw16[4];
                const char* bp16[4];
#pragma unroll
                for (int mt = 0; mt < 4; ++mt) {
                    bp16[mt] = patch + pbase16[mt] + tapoff;
                    bsw16[mt] = cswz(pcol16[mt] + (tap % 3));
                }
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    v4i af[4], bfr[4];
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt) af[nt] = *reinterpret_cast<const v4i*>(wt + aoff16[nt][ks]);
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt)
                        bfr[mt] = *reinterpret_cast<const v4i*>(bp16[mt] + (((ks * 4 + (lane >> 4)) ^ bsw16[mt]) << 4));
#pragma unroll
                    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            acc16[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                                    __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                                    acc16[nt][mt], 0, 0, 0);
                }
                continue;
            }
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

#define DMA_ZPRE zpre
#define DMA_ZPRE_ON zpre_on
#define DMA_HAS_MF16
#include "conv_dma_epilogue.inc"
#undef DMA_ZPRE
#undef DMA_ZPRE_ON
#undef DMA_HAS_MF16
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4: PERSISTENT workgroups whose request stream runs ACROSS tile boundaries (forward launches: the modes of the producer-wave
// instantiation above).  A tile of conv3x3_dma_kernel spends 7.5 k cycles in its prologue (first patch + two weight tiles: a cold
// round trip) and 6 k in its epilogue before / after a main loop of 25 k (two chunks: layer2, the 128-channel U-Net levels) to 100 k
// (layer4) -- and one workgroup per CU (150 KiB of LDS) means nothing overlaps them.  Round 2 tried persistence and lost 4-8 %: the
// epilogue's stores shared the in-order vmcnt counter with the prefetch.  With a request-only ninth wave that problem is gone -- the
// stream and its counter belong to the producer alone -- so here the producer treats the tiles of its workgroup as ONE stream of
// chunks and taps: during the last chunk of tile i it requests the first patch of tile i+1 (into the patch buffer that chunk does not
// use) and at its last two taps the weight tiles of taps 0 and 1 of tile i+1 (the ring simply continues, slot = stream tap & 3).
// The eight accumulating waves go main loop -> epilogue -> main loop; their epilogue stages the tile in what the stream is NOT
// filling at that moment: the patch buffer of the chunk just finished (rows 0..159) and the two ring slots that hold neither W(0)
// nor W(1) of the next tile (rows 160..223, 224..255).  The producer joins the epilogue's barriers (s_barrier counts every live wave)
// and issues nothing between the last tap of a tile and the first barrier of the next one.
// PMOE_DMA_STREAM=0: A/B switch back to the one-tile-per-workgroup kernels.
// NARROW (NT = 1; round 4, BASELINE config 4): tiles of 256 pixels x 64 output channels for the 64-output-channel layers with >= 128 input
// channels -- the frozen U-Nets' 128 -> 64 decoder convolutions at full resolution (model/blocks/unet.py:57-63, 10 launches per
// step that the generic register-staged kernel served at 610-630 TFLOP/s).  Same 4 x 2 wave grid, a wave's tile is 64 pixels x 32
// couts (one A fragment, two B fragments, two MFMAs per 16 channels), weight tiles of 8 KiB, staging rows of 128 bytes.
// Measured and REMOVED (round 4, late; profiles/r04_dma_stamps.log): ONE workgroup barrier per TWO taps.  The one-tile kernel's stamps show
// 1510-1690 cycles per tap in the main loop against 1024 of MFMA time per SIMD, so the barrier of every odd stream tap was dropped (tiles
// with whole pairs of channel chunks: the producers publish W(t), W(t+1) at barrier t and request W(t+2), W(t+3) into the two slots the
// previous step read; a chunk's patch lands one tap earlier where its first tap is odd; both copies of the tap loop unrolled so that the
// scheduler may carry fragment reads across the unsynchronised boundary).  Bit-identical, and worth +1.8 % on layer2, +3.5 % on layer3
// (in-process A/B) -- but the paired 16x16x32 instantiation needs 178 registers of a 12-wave workgroup's 168, the 32x32x16 one it falls back
// to is slower on the 256-channel levels, and config 4 as a whole went 109.1 -> 110.4 ms.  The per-tap overhead is not the barrier: at
// 0.6-0.65 MFMA utilisation these kernels run at 1.9-2.0 GHz, a bare stream of the same MFMAs at ~1.45 (1481 TFLOP/s): the part is
// power-limited, and cycles saved come back as clock lost.
// Measured and REMOVED (round 4, tools/ab_ops.py): the data-gradient modes with a side input (PMOE_RES_DBN / PMOE_RES_ADD) on this kernel.
// The 8-wave kernel prefetches the read-out's z / residual vectors under the last chunk in 32 registers; a 12-wave workgroup has 168 per
// wave and the loop needs 156-162, so they were requested after the accumulators had been staged, flying across an LDS-only staging
// barrier.  layer2.1.conv1's data gradient (PMOE_RES_ADD): 0.377 ms on the 8-wave kernel, 0.491 ms here (the plain data gradient of the
// same shape: 0.287) -- the instantiation spills ~50 registers around the epilogue and the exposed round trip is paid per tile.
// Measured and REMOVED (round 4, tools/ab_inbn.py): PMOE_RES_INBN on this kernel -- the producer waves turning every halo patch into
// relu(BatchNorm(z)) in LDS between its landing (requests moved to taps 0..2, counted wait at barrier 4) and its first use, bit-identical
// to pmoe_bn_apply + the plain launch.  With a quarter of the pieces on each of the four producers (three per tap at taps 4..7) a
// 128-channel launch went 0.325 -> 0.404 ms against the 0.101 ms pass it replaces (pair 0.426 -> 0.404) and a 256-channel one 0.267 ->
// 0.331 against 0.042 (a loss); with the two patch producers alone (six pieces per tap, reads grouped) 0.431 / 0.373 ms.  A piece costs
// a producer ~300 cycles (two magic-number divisions, 28 VALU, an LDS round trip) on a SIMD it shares with two accumulating waves, and
// every producer that arrives late holds the per-tap barrier for all twelve waves.  The 64-channel kernel keeps its variant (conv_res.hip).
template <bool MF16, bool NARROW = false>
__global__ void __launch_bounds__(NTHR + 4 * 64, 1) conv3x3_dma_stream_kernel(const ConvArgs a_in, const int pbuf_bytes, const int ntiles,
                                                                          const int magic_pw, const int magic_ph) {
    constexpr int NT = NARROW ? 1 : 2, WPC = 4 * NT;     // WPC: pieces of a weight tile per producer wave
    static_assert(!(NARROW && MF16), "the 64-cout tile exists on the 32x32x16 shape only");
    constexpr int BN = 64 * NT, WSLOT = BN * RB;        // (shadow the 128-cout constants of this file)
    STAMP_INIT                                           // (the -DPMOE_STAMP tools build never launches this kernel: conv_dma_uses_stream)
    ConvArgs a = a_in;
    a.res_mode = PMOE_RES_NONE; a.res = nullptr; a.bias = nullptr; a.act = PMOE_ACT_NONE; a.drop_p = 0.f; a.bn = nullptr;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = (wave >> 1) & 3, wn = wave & 1;
    const int nblk = a.CoutP / BN;
    const int lTW = a.lTW, lTH = a.lTH;
    const int TW = 1 << lTW, TH = 1 << lTH;
    const int PW = TW + 2, PH = TH + 2;
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    char* wring = smem + 2 * pbuf_bytes;
    const int nchunks = a.Cin / CK;
    const int T = nchunks * 9;
    const int G = (int)gridDim.x;
    // step s of workgroup b: logical tile s*G + xcd_remap(b, G) -- the workgroups of one XCD walk neighbouring tiles at every step
    const int b0 = (int)xcd_remap(blockIdx.x, gridDim.x);
    const int nbar_epi = a.stats ? 4 : 2;               // barriers of the epilogue (conv_dma_epilogue.inc, one-phase read-out)
    struct Tile { int e, n0, n_end, oy0, ox0, cout0; unsigned mb; };
    auto decode = [&](int L) {
        Tile q;
        q.mb = (unsigned)(L / nblk);
        const int nb = L % nblk;
        int t = (int)q.mb;
        const int px_t = t % a.tiles_x; t /= a.tiles_x;
        const int py_t = t % a.tiles_y; t /= a.tiles_y;
        const int ng = t % a.n_groups;
        q.e = t / a.n_groups;
        q.n0 = q.e * a.ipe + ng * a.TN; q.n_end = (q.e + 1) * a.ipe;
        q.oy0 = py_t * TH; q.ox0 = px_t * TW; q.cout0 = nb * BN;
        return q;
    };
    constexpr int OOB = 0x7ff80000;

    if (wave >= 8) {
        // ---------------- producers: waves 8 and 9 share the weight tiles, waves 10 and 11 the patches (pieces of their parity).  ONE
        // producer issuing the 24 requests of a tap needs about the tap's own 1400 cycles for them (60-100 per request): with the
        // whole stream on one wave layer3 / layer4 ran 25 % / 40 % slower than the one-tile kernels (profiles/r04_kernel_ab.log, block 5).
        // Every producer owns the in-order vmcnt bookkeeping of ITS requests: before barrier(tt) it waits until only the group it
        // issued during the previous tap is outstanding.
        const int pw = wave - 8;
        // one input descriptor per EXPERT (conv_dma_plan: an expert's images span < 2^31 bytes), the tile's image group in the offset
        auto rs_in_of = [&](const Tile& q) {
            const bf16* inb = (const bf16*)a.in + (size_t)q.e * a.ipe * a.H * a.W * a.in_ld + a.in_coff;
            return __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, (int)(((long long)a.ipe * a.H * a.W * a.in_ld - a.in_coff) * 2),
                                                     0x00020000);
        };
        auto rs_w_of = [&](const Tile& q) {
            const bf16* wb = (const bf16*)a.w + ((size_t)q.e * a.CoutP + q.cout0) * 9 * a.Cin;
            return __builtin_amdgcn_make_buffer_rsrc((void*)wb, (short)0, BN * 9 * a.Cin * 2, 0x00020000);
        };
        int L = b0;
        if (L >= ntiles) return;
        int gtt = 0, gch = 0;                            // stream position of the current tile's first tap / chunk
        if (pw >= 2) {
            // ---- patch producers (waves 10, 11): pieces of parity pp2 = pw - 2.  No per-lane offset tables: 96 registers of them plus
            // the per-tile copies hipcc hoists out of the chunk loop spill at the 168-register budget of a 12-wave workgroup, and scratch
            // traffic counts in the wave's vmcnt stream.  A piece's source offset is computed when it is requested (two magic-number
            // divisions + ~10 VALU); two waves halve that cost per tap.  `opaque` (0 at run time, unknown at compile time) keeps the
            // offsets from being hoisted into tables again.
            const int pp2 = pw - 2;
            const int lrow = lane >> 3, ljj = lane & 7;
            auto req_patch = [&](const Tile& q, const __amdgpu_buffer_rsrc_t& rs, int i, int buf, int c0, int opaque) {
                const int pp = (i << 3) + lrow + opaque;
                const int rowq = (pp * magic_pw) >> 16, px = pp - rowq * PW;
                const int pn = (rowq * magic_ph) >> 16, prow = rowq - pn * PH;
                const int n = q.n0 - q.e * a.ipe + pn, Y = q.oy0 - 1 + prow, X = q.ox0 - 1 + px;
                const bool ok = pp < NPIX && n < a.ipe && (unsigned)Y < (unsigned)a.H && (unsigned)X < (unsigned)a.W;
                const int off = ((((n * a.H + Y) * a.W + X) * a.in_ld) << 1) + ((ljj ^ cswz(px)) << 4);
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(smem + buf * pbuf_bytes + (i << 10)), 16, ok ? off : OOB, c0 << 1, 0, 0);
            };
            Tile cur = decode(L);
            __amdgpu_buffer_rsrc_t rin = rs_in_of(cur);
            for (int i = pp2; i < NPIECE; i += 2) req_patch(cur, rin, i, 0, 0, 0);
            int after = 0;
            for (;;) {
                const bool has_next = L + G < ntiles;
                const Tile nxt = decode(has_next ? L + G : L);
                const __amdgpu_buffer_rsrc_t rin_n = rs_in_of(nxt);
                for (int ch = 0; ch < nchunks; ++ch) {
                    const int c0 = ch * CK;
                    const bool more = ch + 1 < nchunks;
                    int opaque = 0;
                    asm volatile("" : "+s"(opaque));
#pragma unroll
                    for (int tap = 0; tap < 9; ++tap) {
                        vm_wait_prod(after);
                        __builtin_amdgcn_s_barrier();
                        int n = 0;
                        if (tap >= 1 && tap <= 6 && (more || has_next)) {
                            // the next chunk's patch: of this tile, or chunk 0 of the NEXT tile during this tile's last chunk
#pragma unroll
                            for (int j = 0; j < 4; ++j) {
                                const int idx = (tap - 1) * 8 + 2 * j + pp2;
                                if (idx < NPIECE) {
                                    if (more) req_patch(cur, rin, idx, (gch + ch + 1) & 1, c0 + CK, opaque);
                                    else req_patch(nxt, rin_n, idx, (gch + ch + 1) & 1, 0, opaque);
                                    ++n;
                                }
                            }
                        }
                        after = n;
                    }
                }
                for (int i = 0; i < nbar_epi; ++i) __builtin_amdgcn_s_barrier();      // the accumulating waves' epilogue
                if (!has_next) return;
                gch += nchunks;
                L += G; cur = nxt; rin = rin_n;
            }
        }
        // ---- weight producers: pieces of parity pw of every tap tile
        int wv[8];                                       // (WPC of them are used: a dependent array bound captured by the lambda below trips hipcc's host pass)
#pragma unroll
        for (int i = 0; i < WPC; ++i) {
            const int row = ((2 * i + pw) << 3) + (lane >> 3);
            wv[i] = ((row * 9 * a.Cin) << 1) + (((lane & 7) ^ cswz(row)) << 4);
        }
        auto req_w = [&](const __amdgpu_buffer_rsrc_t& rs, int slot, int tap, int c0) {
#pragma unroll
            for (int i = 0; i < WPC; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_void*)(wring + slot * WSLOT + ((2 * i + pw) << 10)), 16, wv[i],
                                                         (tap * a.Cin + c0) << 1, 0, 0);
        };
        Tile cur = decode(L);
        __amdgpu_buffer_rsrc_t rw = rs_w_of(cur);
        req_w(rw, 0, 0, 0);
        req_w(rw, 1, 1, 0);
        int after = WPC;                                 // own requests issued after this wave's share of W(stream tap)
        for (;;) {
            const bool has_next = L + G < ntiles;
            const Tile nxt = decode(has_next ? L + G : L);
            const __amdgpu_buffer_rsrc_t rw_n = rs_w_of(nxt);
            for (int ch = 0; ch < nchunks; ++ch) {
                const int c0 = ch * CK;
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int tt = ch * 9 + tap;
                    vm_wait_prod(after);
                    __builtin_amdgcn_s_barrier();
                    int n = 0;
                    if (tt + 2 < T) {
                        int ntap = tap + 2, nc0 = c0;
                        if (ntap >= 9) { ntap -= 9; nc0 += CK; }
                        req_w(rw, (gtt + tt + 2) & (RING - 1), ntap, nc0);
                        n = WPC;
                    } else if (has_next) {               // taps 0 / 1 of the next tile's first chunk
                        req_w(rw_n, (gtt + tt + 2) & (RING - 1), tt + 2 - T, 0);
                        n = WPC;
                    }
                    after = n;
                }
            }
            for (int i = 0; i < nbar_epi; ++i) __builtin_amdgcn_s_barrier();          // the accumulating waves' epilogue
            if (!has_next) return;
            gtt += T;
            L += G; cur = nxt; rw = rw_n;
        }
    }

    // ---------------- accumulating waves
    int pbase[2], pcol[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        pbase[mt] = ((pn * PH + my) * PW + mx) << LOG_RB;
        pcol[mt] = mx;
    }
    int aoff[NT][4];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int row = wn * (32 * NT) + nt * 32 + l31;
            aoff[nt][ks] = row * RB + (((ks * 2 + h) ^ cswz(row)) << 4);
        }
    int pbase16[4], pcol16[4], aoff16[4][2];
    if constexpr (MF16) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const int p = wm * 64 + mt * 16 + (lane & 15);
            const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
            pbase16[mt] = ((pn * PH + my) * PW + mx) << LOG_RB;
            pcol16[mt] = mx;
        }
#pragma unroll
        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const int row = wn * 64 + nt * 16 + (lane & 15);
                aoff16[nt][ks] = row * RB + (((ks * 4 + (lane >> 4)) ^ cswz(row)) << 4);
            }
    }
    int gtt = 0, gch = 0;
    for (int L = b0; L < ntiles; L += G) {
        const Tile q = decode(L);
        const int e = q.e, n0 = q.n0, n_end = q.n_end, oy0 = q.oy0, ox0 = q.ox0, cout0 = q.cout0;
        const unsigned mb = q.mb;
        f32x16 acc[NT][2];
        f32x4 acc16[4][4];
        if constexpr (MF16) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) acc16[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        } else {
#pragma unroll
            for (int i = 0; i < NT; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;
        }
        for (int ch = 0; ch < nchunks; ++ch) {
            const char* patch = smem + ((gch + ch) & 1) * pbuf_bytes;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int tt = ch * 9 + tap;
                __builtin_amdgcn_s_barrier();            // the producer has this tap's weight tile (and this chunk's patch) in LDS
                const char* wt = wring + ((gtt + tt) & (RING - 1)) * WSLOT;
                const int tapoff = ((tap / 3) * PW + (tap % 3)) << LOG_RB;
                if constexpr (MF16) {
                    int bsw16[4];
                    const char* bp16[4];
#pragma unroll
                    for (int mt = 0; mt < 4; ++mt) {
                        bp16[mt] = patch + pbase16[mt] + tapoff;
                        bsw16[mt] = cswz(pcol16[mt] + (tap % 3));
                    }
#pragma unroll
                    for (int ks = 0; ks < 2; ++ks) {
                        v4i af[4], bfr[4];
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt) af[nt] = *reinterpret_cast<const v4i*>(wt + aoff16[nt][ks]);
#pragma unroll
                        for (int mt = 0; mt < 4; ++mt)
                            bfr[mt] = *reinterpret_cast<const v4i*>(bp16[mt] + (((ks * 4 + (lane >> 4)) ^ bsw16[mt]) << 4));
#pragma unroll
                        for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 4; ++mt)
                                acc16[nt][mt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                                        __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                                        acc16[nt][mt], 0, 0, 0);
                    }
                } else {
                    int bsw[2];
                    const char* bp[2];
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt) {
                        bp[mt] = patch + pbase[mt] + tapoff;
                        bsw[mt] = cswz(pcol[mt] + (tap % 3));
                    }
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {
                        v4i af[NT], bfr[2];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) af[nt] = *reinterpret_cast<const v4i*>(wt + aoff[nt][ks]);
#pragma unroll
                        for (int mt = 0; mt < 2; ++mt)
                            bfr[mt] = *reinterpret_cast<const v4i*>(bp[mt] + (((ks * 2 + h) ^ bsw[mt]) << 4));
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                            for (int mt = 0; mt < 2; ++mt)
                                acc[nt][mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[nt]),
                                                                                      __builtin_bit_cast(bf16x8, bfr[mt]),
                                                                                      acc[nt][mt], 0, 0, 0);
                    }
                }
            }
        }
        gtt += T; gch += nchunks;
        // staging map of this tile's epilogue: the patch buffer of the chunk just finished + the two ring slots that hold neither of the
        // next tile's first two weight tiles (slots gtt & 3 and (gtt + 1) & 3)
        char* stg_lo = smem + ((gch - 1) & 1) * pbuf_bytes;
        char* stg_s0 = wring + ((gtt + 2) & (RING - 1)) * WSLOT;
        char* stg_s1 = wring + ((gtt + 3) & (RING - 1)) * WSLOT;
        // (NT = 1: 256 rows of 128 bytes, all of them in the patch buffer)
#define DMA_STG_ROW(p) (NT == 1 ? stg_lo + (p) * 128 : (p) < 160 ? stg_lo + (p) * 256 : (p) < 224 ? stg_s0 + ((p) - 160) * 256 : stg_s1 + ((p) - 224) * 256)
#define DMA_RED_BASE stg_lo
#define DMA_HAS_MF16
#define DMA_NT NT
#include "conv_dma_epilogue.inc"
#undef DMA_STG_ROW
#undef DMA_RED_BASE
#undef DMA_HAS_MF16
#undef DMA_NT
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 4, measured and REMOVED: this kernel with the roles separated like conv_wgrad_dma2_kernel (conv_wgrad.hip) -- four
// accumulating waves of 64 pixels x 128 output channels (one per SIMD, fragments double-buffered a 16-channel step ahead, the
// per-tap barrier executed before the tap's last eight MFMAs) + four request-only waves.  Correct (the conv suite passed on it) and
// 4 / 8 / 14 % SLOWER on layer2 / 3 / 4 (profiles/r04_kernel_ab.log, block 4): per tap the workgroup issues 24 LDS-DMA requests for
// 1024 cycles of MFMAs per SIMD; a wave issues one request per 60-150 cycles, so four issuers need 600-900 cycles per tap and
// arrive at every barrier last, while the weight-gradient kernel's 80 requests per 4600-cycle m-block leave its four issuers idle
// half of the time.  This kernel's request rate needs all eight waves as issuers, which is what it has.
// ------------------------------------------------------------------------------------------------------------------------
// Round 3: the three stride-2 3x3 forward convolutions (ResNet layer2-4 `.0.conv1`, torchvision BasicBlock with stride 2:
// model/blocks/backbone.py:57-70) on the same LDS-DMA structure.  They ran on the generic register-staged kernel at
// 0.14 of the matrix peak with 1.85x the compulsory HBM traffic.
//
// A stride-2 3x3 convolution is four stride-1 convolutions over the input's PARITY PLANES: input row 2b+p is row b of plane
// p; output row oy reads (plane 1, block oy-1) for ky = 0, (plane 0, block oy) for ky = 1, (plane 1, block oy) for ky = 2,
// the same along x -- plane (py,px) carries (1+py)(1+px) of the 9 taps.  No plane-split copy of the activation is needed:
// an LDS-DMA lane reads its 16 bytes from ANY global address, so the patch of a plane is gathered straight out of the
// NHWC tensor with a pixel step of 2 (whole 128-byte channel rows per pixel; the neighbouring rows belong to the other
// planes of the same workgroup: L2 hits).  A step = (plane, 64-channel chunk): its halo patch [(TH+1) x (TW+1) blocks]
// by DMA into one of THREE buffers, then its 4 / 2 / 2 / 1 taps.  Steps are as short as one tap, so the patch of step s+2
// is requested at the first tap of step s (its buffer was step s-1's), and the weight ring has 3 slots (tile t+2 goes into
// the slot read at tap t-1, released by the barrier of tap t).  Tiles are 16 x 16 output pixels: 17 x 17 blocks = 37 KiB per
// patch, 3 patches + 3 weight tiles = 159 KiB.  The in-order `vmcnt` bookkeeping of this irregular schedule is done with
// scalar sequence numbers: every DMA instruction the wave issues is counted, the count at the last instruction of each weight
// tile / patch is remembered, and a tap waits until at most (issued - needed) instructions are outstanding.
__device__ __forceinline__ void vm_wait_upto(int n) {     // s_waitcnt vmcnt(min(n, 16)): the immediate must be a constant
    switch (n) {
        case 0: VMCNT(0); break; case 1: VMCNT(1); break; case 2: VMCNT(2); break; case 3: VMCNT(3); break;
        case 4: VMCNT(4); break; case 5: VMCNT(5); break; case 6: VMCNT(6); break; case 7: VMCNT(7); break;
        case 8: VMCNT(8); break; case 9: VMCNT(9); break; case 10: VMCNT(10); break; case 11: VMCNT(11); break;
        case 12: VMCNT(12); break; case 13: VMCNT(13); break; case 14: VMCNT(14); break; case 15: VMCNT(15); break;
        default: VMCNT(16); break;
    }
}

constexpr int RING3 = 3;

// CLS = true: ONE PARITY CLASS of a stride-2 3x3 DATA GRADIENT on the same schedule (layer2-4 `.0.conv1`, the four launches the
// generic kernel served at 0.09-0.2 of the matrix peak): dx[2a+py][2b+px] is a stride-1 correlation of dy with the (1+py)(1+px)
// taps of the flipped filter that reach that parity (a.kh x a.kw taps at block offsets (r, q) in {0,1}^2, filter taps a.tapmap),
// written to the class's lattice of dx (a.out_step / out_offy / out_offx over [a.OH][a.OW]).  One "plane" (pixel step 1, halo on
// the bottom / right), a step = a 64-channel chunk of dy with all of the class's taps.
template <bool CLS>
__global__ void __launch_bounds__(NTHR, 2) conv3x3s2_dma_kernel(const ConvArgs a, const int pbuf_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    STAMP_INIT
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

    const int nblk = (a.CoutP + BN - 1) / BN;            // (CLS: a 64-row gradient runs half a tile of zero weights)
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
    const int PW = TW + 1, PH = TH + 1;                  // blocks: one halo row / column on the top / left only
    const int NPIX = a.TN * PH * PW;
    const int NPIECE = (NPIX + 7) >> 3;
    const int n0 = e * a.ipe + ng * a.TN, n_end = (e + 1) * a.ipe;
    const int oy0 = py_t * TH, ox0 = px_t * TW;
    const int cout0 = nb * BN;

    char* wring = smem + 3 * pbuf_bytes;

    const bf16* inb = (const bf16*)a.in + (size_t)n0 * a.H * a.W * a.in_ld + a.in_coff;
    long long in_bytes = ((long long)(n_end - n0) * a.H * a.W * a.in_ld - a.in_coff) * 2;
    if (in_bytes > 0x7ff00000ll) in_bytes = 0x7ff00000ll;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, (int)in_bytes, 0x00020000);
    const bf16* wb = (const bf16*)a.w + ((size_t)e * a.CoutP + cout0) * 9 * a.Cin;
    const int wrows = a.CoutP - cout0 < BN ? a.CoutP - cout0 : BN;      // rows beyond the bank arrive as zeros
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wb, (short)0, wrows * 9 * a.Cin * 2, 0x00020000);

    // ---- per-lane source offsets of this wave's patch pieces: pixel (2 By, 2 Bx) of the block; the plane's (py, px) and the
    // channel chunk travel in the scalar offset.  4 validity bits per piece: rows 2By / 2By+1 and columns 2Bx / 2Bx+1 inside
    // the image (odd image sides: the last block has only its plane-0 row / column)
    constexpr int OOB = 0x7ff80000;
    int pvoff[6];
    unsigned vbits = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i) {
        const int pp = ((wave + 8 * i) << 3) + (lane >> 3);
        const int jj = lane & 7;
        const int bx = pp % PW;
        const int rowq = pp / PW;
        const int by = rowq % PH, pn = rowq / PH;
        if constexpr (CLS) {                              // dy pixel (oy0 + by, ox0 + bx): rows / columns beyond dy read as zero
            const int By = oy0 + by, Bx = ox0 + bx;
            const bool ok = pp < NPIX && n0 + pn < n_end && By < a.H && Bx < a.W;
            pvoff[i] = ok ? ((((pn * a.H + By) * a.W + Bx) * a.in_ld) << 1) + ((jj ^ cswz(bx)) << 4) : OOB;
        } else {
        const int By = oy0 - 1 + by, Bx = ox0 - 1 + bx;
        const bool okb = pp < NPIX && n0 + pn < n_end && By >= 0 && Bx >= 0;
        pvoff[i] = okb ? ((((pn * a.H + 2 * By) * a.W + 2 * Bx) * a.in_ld) << 1) + ((jj ^ cswz(bx)) << 4) : OOB;
        const unsigned vb = (okb && 2 * By < a.H ? 1u : 0u) | (okb && 2 * By + 1 < a.H ? 2u : 0u) |
                            (2 * Bx < a.W ? 4u : 0u) | (2 * Bx + 1 < a.W ? 8u : 0u);
        vbits |= vb << (4 * i);
        }
    }
    int wvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = ((wave + 8 * i) << 3) + (lane >> 3);
        wvoff[i] = ((row * 9 * a.Cin) << 1) + (((lane & 7) ^ cswz(row)) << 4);
    }
    const int my_pieces = (NPIECE - wave + 7) >> 3;

    // plane index pl: 0 = (py 1, px 1) 4 taps, 1 = (1, 0) 2 taps, 2 = (0, 1) 2 taps, 3 = (0, 0) 1 tap
    auto dma_patch = [&](int buf, int pl, int c0) {      // all pieces of this wave for step (pl, c0)
        const int py = pl < 2, px = (pl & 1) == 0;
        const int soff = (c0 << 1) + (CLS ? 0 : (((py * a.W + px) * a.in_ld) << 1));
#pragma unroll
        for (int i = 0; i < 6; ++i)
            if (i < my_pieces) {
                const bool ok = CLS || (((vbits >> (4 * i + py)) & 1u) && ((vbits >> (4 * i + 2 + px)) & 1u));
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(smem + buf * pbuf_bytes + ((wave + 8 * i) << 10)), 16,
                                                         ok ? pvoff[i] : OOB, soff, 0, 0);
            }
    };
    auto dma_w = [&](int slot, int tap, int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(wring + slot * WSLOT + ((wave + 8 * i) << 10)), 16,
                                                     wvoff[i], (tap * a.Cin + c0) << 1, 0, 0);
    };
    // tap t of plane pl -> filter tap ky * 3 + kx and the block offset (oy, ox) inside the patch
    auto tap_of = [&](int pl, int tp, int& ktap, int& oy, int& ox) {
        if constexpr (CLS) {
            oy = a.kw == 2 ? (tp >> 1) : tp; ox = a.kw == 2 ? (tp & 1) : 0;
            ktap = tp == 0 ? a.tapmap[0] : tp == 1 ? a.tapmap[1] : tp == 2 ? a.tapmap[2] : a.tapmap[3];
            return;
        }
        const int py = pl < 2, px = (pl & 1) == 0;
        const int iy = px ? (tp >> 1) : tp, ix = px ? (tp & 1) : 0;
        const int ky = py ? 2 * iy : 1, kx = px ? 2 * ix : 1;
        ktap = ky * 3 + kx;
        oy = ky != 0; ox = kx != 0;
    };
    auto ntaps = [&](int pl) { return CLS ? a.kh * a.kw : pl == 0 ? 4 : pl == 3 ? 1 : 2; };

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

    const int nch = a.Cin / CK;
    const int T = CLS ? nch * a.kh * a.kw : nch * 9, NSTEP = CLS ? nch : nch * 4;

    // ---- scalar bookkeeping of the DMA stream (wave-uniform).  Issue order inside iteration j: weight tile of tap j+2 (2
    // instructions), then -- at the first tap of a step -- the my_pieces patch pieces of step s+2.  Tap tt needs W(tt), the first
    // thing iteration tt-2 issued: everything behind it may stay in flight = pieces(tt-2) + all of iteration tt-1.  A step's
    // patch was issued two steps earlier and is older than W(tt) unless both steps in between were single taps (then it is
    // the LAST thing iteration tt-2 issued and only iteration tt-1 may stay in flight).  No arrays: scratch accesses would
    // themselves count in vmcnt.
    int c1 = 0, p1 = 0, p2 = 0;                          // instructions of iteration tt-1; patch pieces of tt-1 / tt-2
    int len1 = 0, len2 = 0;                              // taps of steps s-1 / s-2
    // step s -> (plane, chunk): plane-major, all chunks of a plane in a row
    auto step_plane = [&](int s) { return CLS ? 0 : s / nch; };
    auto step_c0 = [&](int s) { return (s % nch) * CK; };
    // look-ahead iterator two taps ahead of the current one
    int pl2 = 0, ch2 = 0, tp2 = 0;
    auto advance2 = [&]() {
        if (++tp2 == ntaps(pl2)) { tp2 = 0; if (++ch2 == nch) { ch2 = 0; ++pl2; } }
    };

    // prologue: patches of steps 0 and 1, weight tiles of taps 0 and 1
    dma_patch(0, step_plane(0), step_c0(0));
    if (NSTEP > 1) dma_patch(1, step_plane(1), step_c0(1));
    {
        int kt, oy, ox;
        tap_of(pl2, tp2, kt, oy, ox); dma_w(0, kt, ch2 * CK); advance2();
        if (T > 1) { tap_of(pl2, tp2, kt, oy, ox); dma_w(1, kt, ch2 * CK); advance2(); c1 = 2; }      // W(1) sits behind W(0)
    }

    int tt = 0, wslot = 0;                               // wslot = tt % 3
    for (int s = 0, sbuf = 0; s < NSTEP; ++s, sbuf = sbuf == 2 ? 0 : sbuf + 1) {      // sbuf = s % 3
        const int pl = step_plane(s);
        const char* patch = smem + sbuf * pbuf_bytes;
        const int nt_s = ntaps(pl);
        for (int tp = 0; tp < nt_s; ++tp, ++tt, wslot = wslot == 2 ? 0 : wslot + 1) {
            const bool patch_last = tp == 0 && s >= 2 && len1 == 1 && len2 == 1;
            vm_wait_upto(c1 + (patch_last ? 0 : p2));
            __builtin_amdgcn_s_barrier();                // every wave's share of tap tt (and of this step's patch) is in LDS
            int cnow = 0, pnow = 0;
            if (tt + 2 < T) {
                int kt, oy2, ox2;
                tap_of(pl2, tp2, kt, oy2, ox2);
                dma_w(wslot == 0 ? 2 : wslot - 1, kt, ch2 * CK);            // slot (tt + 2) % 3
                cnow = 2;
                advance2();
            }
            if (tp == 0 && s + 2 < NSTEP) {              // the buffer of step s-1 is free since this barrier
                dma_patch(sbuf == 0 ? 2 : sbuf - 1, step_plane(s + 2), step_c0(s + 2));      // buffer (s + 2) % 3
                pnow = my_pieces;
            }
            c1 = cnow + pnow; p2 = p1; p1 = pnow;
            int ktap, oy, ox;
            tap_of(pl, tp, ktap, oy, ox);
            const char* wt = wring + wslot * WSLOT;
            const int tapoff = (oy * PW + ox) << LOG_RB;
            int bsw[2];
            const char* bp[2];
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                bp[mt] = patch + pbase[mt] + tapoff;
                bsw[mt] = cswz(pcol[mt] + ox);
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
        len2 = len1; len1 = nt_s;
    }

    if constexpr (CLS) {
#define DMA_LATTICE
#include "conv_dma_epilogue.inc"
#undef DMA_LATTICE
    } else {
#include "conv_dma_epilogue.inc"
    }
}

// ------------------------------------------------------------------------------------------------------------------------
// Round 3, BASELINE config 5: the dense 3x3 stride-1 forward convolutions with >= 128 input channels on the BLOCK-SCALED fp8
// matrix instruction v_mfma_scale_f32_32x32x64_f8f6f4 -- twice the bf16 rate per clock (MI355X_MICROARCH.md, matrix cores), the
// one way fp8 operands pay on gfx950 (the non-scaled 32x32x16_fp8_fp8 form of round 2 runs at the bf16 rate).  Both operands are
// e4m3 BYTES in HBM: the weights from pmoe_pack_conv_weights_fp8 (one power-of-two scale per output channel), the activations
// written as e4m3(x * in_scale) by the producing BatchNorm pass (pmoe_bn_apply's fp8 side output) -- no conversion in this
// kernel, and half the activation bytes through L2 and LDS.  Same schedule as conv3x3_dma_kernel with the SAME byte geometry: a
// 128-byte LDS row now holds 128 channels, so a channel chunk is 128 channels and a tap is two 64-deep MFMA steps (4 + 4
// instructions of 64 cycles per wave instead of 16 of 32 for 64 bf16 channels).  Lane half h feeds bytes [32 h, 32 h + 32) of a
// 64-channel step to both operands (any split works as long as A and B agree: tools/probe_mfma_f8.hip); unit block scales
// (E8M0 0x7f), the real scales are applied per output channel in the epilogue (DMA_OSCALE).
typedef int v8i __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(NTHR, 2) conv3x3_dma_f8_kernel(const ConvArgs a, const int pbuf_bytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    STAMP_INIT
    constexpr int CK8 = 128;                             // channels (= bytes) per chunk
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, h = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;

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

    const unsigned char* inb = (const unsigned char*)a.in + (size_t)n0 * a.H * a.W * a.in_ld + a.in_coff;
    long long in_bytes = (long long)(n_end - n0) * a.H * a.W * a.in_ld - a.in_coff;
    if (in_bytes > 0x7ff00000ll) in_bytes = 0x7ff00000ll;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc((void*)inb, (short)0, (int)in_bytes, 0x00020000);
    const unsigned char* wb = (const unsigned char*)a.w + ((size_t)e * a.CoutP + cout0) * 9 * a.Cin;
    const __amdgpu_buffer_rsrc_t rs_w = __builtin_amdgcn_make_buffer_rsrc((void*)wb, (short)0, BN * 9 * a.Cin, 0x00020000);

    constexpr int OOB = 0x7ff80000;
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
        pvoff[i] = ok ? (((pn * a.H + Y) * a.W + X) * a.in_ld) + ((jj ^ cswz(px)) << 4) : OOB;
    }
    int wvoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = ((wave + 8 * i) << 3) + (lane >> 3);
        wvoff[i] = (row * 9 * a.Cin) + (((lane & 7) ^ cswz(row)) << 4);
    }
    const int my_pieces = (NPIECE - wave + 7) >> 3;

    auto dma_patch = [&](int i, int buf, int c0) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_in, (lds_void*)(smem + buf * pbuf_bytes + ((wave + 8 * i) << 10)), 16,
                                                 pvoff[i], c0, 0, 0);
    };
    auto dma_w = [&](int slot, int tap, int c0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lds_void*)(wring + slot * WSLOT + ((wave + 8 * i) << 10)), 16,
                                                     wvoff[i], tap * a.Cin + c0, 0, 0);
    };

    int pbase[2], pcol[2];
#pragma unroll
    for (int mt = 0; mt < 2; ++mt) {
        const int p = wm * 64 + mt * 32 + l31;
        const int mx = p & (TW - 1), my = (p >> lTW) & (TH - 1), pn = p >> (lTW + lTH);
        pbase[mt] = ((pn * PH + my) * PW + mx) << LOG_RB;
        pcol[mt] = mx;
    }
    int arow[2], asw[2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int row = wn * 64 + nt * 32 + l31;
        arow[nt] = row * RB;
        asw[nt] = cswz(row);
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc[i][j][k] = 0.f;

    const int nchunks = a.Cin / CK8;
    const int T = nchunks * 9;

#pragma unroll
    for (int i = 0; i < 6; ++i)
        if (i < my_pieces) dma_patch(i, 0, 0);
    dma_w(0, 0, 0);
    dma_w(1, 1, 0);

    for (int ch = 0; ch < nchunks; ++ch) {
        const int c0 = ch * CK8;
        const char* patch = smem + (ch & 1) * pbuf_bytes;
        const bool more = ch + 1 < nchunks;
        // (NOT unrolled over the taps: with 8-register operand tuples hipcc hoists the fragment reads of later taps and spills
        //  them -- scratch traffic that would also break the vmcnt accounting)
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            const int tt = ch * 9 + tap;
            const bool prev_piece = tap >= 2 && tap <= 7 && more && (tap - 2) < my_pieces;
            if (tt + 1 >= T) VMCNT(0);
            else if (prev_piece) VMCNT(3);
            else VMCNT(2);
            __builtin_amdgcn_s_barrier();
            if (tap >= 1 && tap <= 6 && more && (tap - 1) < my_pieces) dma_patch(tap - 1, (ch + 1) & 1, c0 + CK8);
            if (tt + 2 < T) {
                int ntap = tap + 2, nc0 = c0;
                if (ntap >= 9) { ntap -= 9; nc0 += CK8; }
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
            for (int ks = 0; ks < 2; ++ks) {
                v8i af[2], bfr[2];
#pragma unroll
                for (int nt = 0; nt < 2; ++nt) {
                    const v4i lo = *reinterpret_cast<const v4i*>(wt + arow[nt] + (((ks * 4 + 2 * h) ^ asw[nt]) << 4));
                    const v4i hi = *reinterpret_cast<const v4i*>(wt + arow[nt] + (((ks * 4 + 2 * h + 1) ^ asw[nt]) << 4));
                    af[nt] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int mt = 0; mt < 2; ++mt) {
                    const v4i lo = *reinterpret_cast<const v4i*>(bp[mt] + (((ks * 4 + 2 * h) ^ bsw[mt]) << 4));
                    const v4i hi = *reinterpret_cast<const v4i*>(bp[mt] + (((ks * 4 + 2 * h + 1) ^ bsw[mt]) << 4));
                    bfr[mt] = v8i{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                }
#pragma unroll
                for (int nt = 0; nt < 2; ++nt)
#pragma unroll
                    for (int mt = 0; mt < 2; ++mt)
                        acc[nt][mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(af[nt], bfr[mt], acc[nt][mt], 0, 0, 0, 0x7f, 0, 0x7f);
            }
        }
    }

#define DMA_OSCALE
#include "conv_dma_epilogue.inc"
#undef DMA_OSCALE
}

}  // namespace

// the persistent kernel's piece decode: the magic-number divisions reproduce / and %, the packed fields fit, and rows 0..159 (NT = 2)
// or all 256 half-rows (NT = 1) of the epilogue's staging fit the patch buffer.  Expects the tile geometry of conv_dma_plan in `a`.
static bool stream_geometry_ok(const ConvArgs& a, int pbuf) {
    if (pbuf < 160 * 256) return false;
    const int PW = (1 << a.lTW) + 2, PH = (1 << a.lTH) + 2;
    const int npiece = (a.TN * PH * PW + 7) / 8;
    const int mpw = 65536 / PW + 1, mph = 65536 / PH + 1;
    for (int pp = 0; pp < npiece * 8; ++pp)
        if (((pp * mpw) >> 16) != pp / PW || ((((pp / PW) * mph) >> 16) != (pp / PW) / PH)) return false;
    return a.TN * PH < 1024 && PW < 1024 && a.ipe < 2047;      // the packed (image, row, column) fields of a piece
}

// 64 output-channel rows over >= 128 input channels, nothing but the convolution (+ statistics): conv3x3_dma_stream_kernel<false, true>.
// PMOE_DMA_NARROW=0: back to the generic kernel (A/B runs)
bool conv_dma_is_narrow(const ConvArgs& a) {
#ifdef PMOE_STAMP
    return false;                                        // (the stamped epilogue ends the workgroup after its first tile)
#endif
    const char* ev = getenv("PMOE_DMA_NARROW");
    if (ev && !atoi(ev)) return false;
    return a.CoutP == 64 && a.Cin >= 2 * CK && a.res_mode == PMOE_RES_NONE && !a.bias && a.act == PMOE_ACT_NONE && a.drop_p == 0.f;
}

// Which launches take this kernel: bf16, dense 3x3 stride 1 pad 1 (forward, or the flipped-filter data gradient), whole
// 64-channel chunks, >= 128 output-channel rows, per-expert maps of >= 4096 pixels that tile into 16- or 32-pixel-wide
// strips.  PMOE_CONV_DMA=0 sends them back to conv_igemm_lite_kernel (A/B runs; read per launch, so that tools/ab_conv.py can
// interleave the two in one process).
bool conv_dma_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf) {
    const char* ev = getenv("PMOE_CONV_DMA");
    if ((ev && !atoi(ev)) || dtype != PMOE_DT_BF16 || a.w_fp8) return false;
    if (a.ks != 3 || a.kh != 3 || a.kw != 3 || a.use_tapmap || a.stride != 1 || a.pad != 1 || a.dilate || a.in_shared) return false;
    if (a.out_step != 1 || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.Cin % CK || a.Cout % 8 || a.N % a.ipe) return false;
    // 64 output-channel rows: the NT = 1 instantiation of the persistent kernel only (forward launches, >= 2 channel chunks; the
    // 64 -> 64 layers belong to conv_res.hip's resident-filter kernels, which conv_igemm_launch asks first)
    const bool narrow = conv_dma_is_narrow(a);
    if (a.CoutP % BN && !narrow) return false;
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
    if (narrow && !stream_geometry_ok(a, pb)) return false;
    *mblocks = (a.N / a.ipe) * a.n_groups * a.tiles_y * a.tiles_x;
    *smem = need < (size_t)BM / 2 * BN * 4 ? (size_t)BM / 2 * BN * 4 : need;
    *pbuf = pb;
    return true;
}

bool conv_dma_uses_mf16(const ConvArgs& a) {
    const char* ev = getenv("PMOE_DMA_MF16");
    return ev ? atoi(ev) != 0 : a.Cin >= 256;
}

// the persistent, streaming instantiation (round 4): forward launches whose piece decode the magic-number division reproduces
bool conv_dma_uses_stream(const ConvArgs& a0) {
#ifdef PMOE_STAMP
    return false;                                        // (the stamped epilogue ends the workgroup after its first tile)
#endif
    if (conv_dma_is_narrow(a0)) return true;             // (conv_dma_plan has checked the geometry)
    const char* ev = getenv("PMOE_DMA_STREAM");
    if (ev && !atoi(ev)) return false;
    if (!(a0.res_mode == PMOE_RES_NONE && !a0.bias && a0.act == PMOE_ACT_NONE && a0.drop_p == 0.f)) return false;
    // two and four channel chunks (layer2: +24 %, layer3: +6 %); with eight (layer4: the prologue is 5 % of the tile) the 12-wave
    // workgroup is 4 % slower than the 9-wave producer instantiation -- profiles/r04_kernel_ab.log, block 5.  PMOE_DMA_STREAM=2: always
    if (a0.Cin > 256 && !(ev && atoi(ev) == 2)) return false;
    ConvArgs a = a0;
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return false;
    return stream_geometry_ok(a, pbuf);
}

// the producer-wave instantiation (round 4): forward launches -- nothing added to or derived from a side input in the epilogue
bool conv_dma_uses_producer(const ConvArgs& a) {
    const char* ev = getenv("PMOE_DMA_PRODUCER");
    if (ev && !atoi(ev)) return false;
    // >= 4 channel chunks: layer3 / layer4 forward +8 % / +12 %; with two chunks (layer2) the tile is prologue + epilogue for a
    // third of its time and the ninth wave buys nothing (-2 %): profiles/r04_kernel_ab.log.  PMOE_DMA_PRODUCER=2: every layer
    if (a.Cin < 256 && !(ev && atoi(ev) == 2)) return false;
    return a.res_mode == PMOE_RES_NONE && !a.bias && a.act == PMOE_ACT_NONE && a.drop_p == 0.f;
}

// plan code of a launch conv_dma_plan accepts: 5007 / 5017 = conv3x3_dma_kernel<MF16>, + 20 with the producer wave, + 40 =
// conv3x3_dma_stream_kernel<MF16>, 5067 = conv3x3_dma_stream_kernel<false, true> (64-cout tiles)
int conv_dma_plan_code(const ConvArgs& a) {
    if (conv_dma_is_narrow(a)) return 5067;
    return (conv_dma_uses_mf16(a) ? 5017 : 5007) + (conv_dma_uses_stream(a) ? 40 : conv_dma_uses_producer(a) ? 20 : 0);
}

int conv_dma_launch(ConvArgs a, hipStream_t st) {
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return PMOE_ERR_UNSUPPORTED;
    // v_mfma_f32_16x16x32_bf16 for the layers with >= 4 channel chunks (interleaved A/B, profiles/r03_kernel_ab.log: layer3 forward
    // +0.6 %, data gradient +3.5 %; layer4 +3.5 % / +5 %; layer2 -3 % / +0.7 %: the shorter the main loop, the less the shape's
    // higher sustained clock buys).  PMOE_DMA_MF16=0 | 1 forces one shape (read per launch).
    if (conv_dma_uses_stream(a)) {
        // persistent workgroups, one per CU; the request stream of each runs across its tiles
        const bool narrow = conv_dma_is_narrow(a);
        const int ntiles = narrow ? mblocks : mblocks * (a.CoutP / BN);
        const int PWp = (1 << a.lTW) + 2, PHp = (1 << a.lTH) + 2;
        const int mpw = 65536 / PWp + 1, mph = 65536 / PHp + 1;
        int grid = ntiles < 256 ? ntiles : 256;
        if (narrow) {
            HIP_RET((ensure_dyn_lds<conv3x3_dma_stream_kernel<false, true>>(160 * 1024)));
            hipLaunchKernelGGL((conv3x3_dma_stream_kernel<false, true>), dim3(grid), dim3(NTHR + 4 * 64), smem, st, a, pbuf, ntiles, mpw, mph);
        } else if (conv_dma_uses_mf16(a)) {
            HIP_RET((ensure_dyn_lds<conv3x3_dma_stream_kernel<true>>(160 * 1024)));
            hipLaunchKernelGGL((conv3x3_dma_stream_kernel<true>), dim3(grid), dim3(NTHR + 4 * 64), smem, st, a, pbuf, ntiles, mpw, mph);
        } else {
            HIP_RET((ensure_dyn_lds<conv3x3_dma_stream_kernel<false>>(160 * 1024)));
            hipLaunchKernelGGL((conv3x3_dma_stream_kernel<false>), dim3(grid), dim3(NTHR + 4 * 64), smem, st, a, pbuf, ntiles, mpw, mph);
        }
    } else if (conv_dma_uses_producer(a)) {
        if (conv_dma_uses_mf16(a)) {
            HIP_RET((ensure_dyn_lds<conv3x3_dma_kernel<true, true>>(160 * 1024)));
            hipLaunchKernelGGL((conv3x3_dma_kernel<true, true>), dim3(mblocks * (a.CoutP / BN)), dim3(NTHR + 64), smem, st, a, pbuf);
        } else {
            HIP_RET((ensure_dyn_lds<conv3x3_dma_kernel<false, true>>(160 * 1024)));
            hipLaunchKernelGGL((conv3x3_dma_kernel<false, true>), dim3(mblocks * (a.CoutP / BN)), dim3(NTHR + 64), smem, st, a, pbuf);
        }
    } else if (conv_dma_uses_mf16(a)) {
        HIP_RET((ensure_dyn_lds<conv3x3_dma_kernel<true>>(160 * 1024)));
        hipLaunchKernelGGL(conv3x3_dma_kernel<true>, dim3(mblocks * (a.CoutP / BN)), dim3(NTHR), smem, st, a, pbuf);
    } else {
        HIP_RET((ensure_dyn_lds<conv3x3_dma_kernel<false>>(160 * 1024)));
        hipLaunchKernelGGL(conv3x3_dma_kernel<false>, dim3(mblocks * (a.CoutP / BN)), dim3(NTHR), smem, st, a, pbuf);
    }
    return (int)hipGetLastError();
}


// Stride-2 3x3 forward convolutions (pad 1) of >= 128 output-channel rows over whole 64-channel chunks, bf16, output maps of
// >= 4096 pixels per expert that tile into 16 x 16 squares: conv3x3s2_dma_kernel.  PMOE_CONV_S2DMA=0: back to the generic
// kernel (A/B runs; read per launch).
bool conv_dma_s2_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf) {
    const char* ev = getenv("PMOE_CONV_S2DMA");
    if ((ev && !atoi(ev)) || dtype != PMOE_DT_BF16 || a.w_fp8) return false;
    if (a.ks != 3 || a.kh != 3 || a.kw != 3 || a.use_tapmap || a.stride != 2 || a.pad != 1 || a.dilate || a.in_shared) return false;
    if (a.out_step != 1 || a.Ho != (a.H - 1) / 2 + 1 || a.Wo != (a.W - 1) / 2 + 1) return false;
    if (a.Cin % CK || a.CoutP % BN || a.Cout % 8 || a.N % a.ipe) return false;
    if (a.res_mode == PMOE_RES_DBN) return false;
    if ((long long)a.ipe * a.Ho * a.Wo < 4096) return false;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.Wo); if (lTW > 4) lTW = 4;
    if (lTW < 4) return false;
    int lTH = p2(a.Ho); if (lTH > 8 - lTW) lTH = 8 - lTW;
    const int TN = BM >> (lTW + lTH);
    const int NPIX = TN * ((1 << lTH) + 1) * ((1 << lTW) + 1);
    const int npiece = (NPIX + 7) / 8;
    if (npiece > 48) return false;
    const int pb = npiece * 1024;
    const size_t need = (size_t)3 * pb + RING3 * WSLOT;
    if (need > 160 * 1024) return false;
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

int conv_dma_s2_launch(ConvArgs a, hipStream_t st) {
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_s2_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return PMOE_ERR_UNSUPPORTED;
    HIP_RET((ensure_dyn_lds<conv3x3s2_dma_kernel<false>>(160 * 1024)));
    hipLaunchKernelGGL(conv3x3s2_dma_kernel<false>, dim3(mblocks * (a.CoutP / BN)), dim3(NTHR), smem, st, a, pbuf);
    return (int)hipGetLastError();
}

// One parity class of a stride-2 3x3 data gradient (the class fields of ConvArgs set by launch_stride2_dgrad, conv_igemm.hip):
// bf16, whole 64-channel chunks of dy, gradient rows in multiples of 64, class lattices of >= 4096 pixels per expert that tile
// into 16 x 16 squares.  PMOE_CONV_S2DMA=0: back to the generic kernel.
bool conv_dma_s2cls_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf) {
    const char* ev = getenv("PMOE_CONV_S2DMA");
    if ((ev && !atoi(ev)) || dtype != PMOE_DT_BF16 || a.w_fp8) return false;
    if (a.ks != 3 || !a.use_tapmap || a.out_step != 2 || a.stride != 1 || a.pad != 0 || a.dilate || a.in_shared) return false;
    if (a.kh < 1 || a.kh > 2 || a.kw < 1 || a.kw > 2) return false;
    // (64 gradient rows -- layer2.0.conv1 -- would run half a tile of zero weights: measured 0.77 vs 0.58 ms on the generic kernel;
    //  PMOE_S2CLS_64=1 admits them for that A/B)
    const char* e64 = getenv("PMOE_S2CLS_64");
    if (a.Cin % CK || a.CoutP % ((e64 && atoi(e64)) ? 64 : BN) || a.Cout % 8 || a.N % a.ipe || a.stats || a.res_mode > PMOE_RES_ADD) return false;
    if ((long long)a.ipe * a.Ho * a.Wo < 4096) return false;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.Wo); if (lTW > 4) lTW = 4;
    if (lTW < 4) return false;
    int lTH = p2(a.Ho); if (lTH > 8 - lTW) lTH = 8 - lTW;
    const int TN = BM >> (lTW + lTH);
    const int NPIX = TN * ((1 << lTH) + 1) * ((1 << lTW) + 1);
    const int npiece = (NPIX + 7) / 8;
    if (npiece > 48) return false;
    const int pb = npiece * 1024;
    const size_t need = (size_t)3 * pb + RING3 * WSLOT;
    if (need > 160 * 1024) return false;
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

int conv_dma_s2cls_launch(ConvArgs a, hipStream_t st) {
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_s2cls_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return PMOE_ERR_UNSUPPORTED;
    HIP_RET((ensure_dyn_lds<conv3x3s2_dma_kernel<true>>(160 * 1024)));
    hipLaunchKernelGGL(conv3x3s2_dma_kernel<true>, dim3(mblocks * ((a.CoutP + BN - 1) / BN)), dim3(NTHR), smem, st, a, pbuf);
    return (int)hipGetLastError();
}


// BASELINE config 5 on the block-scaled fp8 matrix instruction: e4m3 weights AND e4m3 activations in HBM (ConvArgs.in_fp8), dense
// 3x3 stride 1, whole 128-channel chunks, >= 128 output-channel rows, maps of >= 4096 pixels per expert.  PMOE_CONV_F8DMA=0: the
// activations go through the bf16 -> e4m3 converting loaders of round 2 instead.
bool conv_dma_f8_plan(ConvArgs& a, int dtype, int* mblocks, size_t* smem, int* pbuf) {
    const char* ev = getenv("PMOE_CONV_F8DMA");
    if ((ev && !atoi(ev)) || dtype != PMOE_DT_BF16 || !a.w_fp8 || !a.in_fp8 || !a.oscale) return false;
    if (a.ks != 3 || a.kh != 3 || a.kw != 3 || a.use_tapmap || a.stride != 1 || a.pad != 1 || a.dilate || a.in_shared) return false;
    if (a.out_step != 1 || a.Ho != a.H || a.Wo != a.W || a.res_mode != PMOE_RES_NONE) return false;
    if (a.Cin % 128 || a.CoutP % BN || a.Cout % 8 || a.N % a.ipe || a.in_ld % 16 || a.in_coff % 16) return false;
    if ((long long)a.ipe * a.Ho * a.Wo < 4096) return false;
    auto p2 = [](int v) { int l = 0; while ((1 << l) < v) ++l; return l; };
    int lTW = p2(a.Wo); if (lTW > 5) lTW = 5;
    if (lTW < 4) return false;
    int lTH = p2(a.Ho); if (lTH > 8 - lTW) lTH = 8 - lTW;
    const int TN = BM >> (lTW + lTH);
    const int NPIX = TN * ((1 << lTH) + 2) * ((1 << lTW) + 2);
    const int npiece = (NPIX + 7) / 8;
    if (npiece > 48) return false;
    const int pb = npiece * 1024;
    const size_t need = (size_t)2 * pb + RING * WSLOT;
    if (need > 160 * 1024) return false;
    if ((long long)a.ipe * a.H * a.W * a.in_ld >= 0x7ff00000ll) return false;
    a.lTW = lTW; a.lTH = lTH; a.TN = TN;
    a.n_groups = (a.ipe + TN - 1) / TN;
    a.tiles_y = (a.Ho + (1 << lTH) - 1) >> lTH;
    a.tiles_x = (a.Wo + (1 << lTW) - 1) >> lTW;
    *mblocks = (a.N / a.ipe) * a.n_groups * a.tiles_y * a.tiles_x;
    *smem = need < (size_t)BM / 2 * BN * 4 ? (size_t)BM / 2 * BN * 4 : need;
    *pbuf = pb;
    return true;
}

int conv_dma_f8_launch(ConvArgs a, hipStream_t st) {
    int mblocks = 0, pbuf = 0;
    size_t smem = 0;
    if (!conv_dma_f8_plan(a, PMOE_DT_BF16, &mblocks, &smem, &pbuf)) return PMOE_ERR_UNSUPPORTED;
    HIP_RET((ensure_dyn_lds<conv3x3_dma_f8_kernel>(160 * 1024)));
    hipLaunchKernelGGL(conv3x3_dma_f8_kernel, dim3(mblocks * (a.CoutP / BN)), dim3(NTHR), smem, st, a, pbuf);
    return (int)hipGetLastError();
}
