// Halo-patch staging shared by the implicit-GEMM conv kernel and the weight-gradient kernel.
#pragma once
#include "common.h"

// LDS chunk swizzles (16-byte chunk index j of a pixel row is stored at j ^ swz(pixel)):
//  mode 0 (ds_read_b128 operand reads, 32 different pixels per half-wave): conflict free when a
//         16-lane group's pixels are distinct mod 16.
//  mode 1 (ds_read_b64_tr_b16 reads of 4 consecutive pixels x 64 B): toggles the 64-byte half of
//         the row with bit 1 of the pixel so the 4 rows of a transposed block hit 4 bank ranges.
template <int LOG_RB, int MODE> __device__ __forceinline__ int swz_chunk(int x) {
    if (MODE == 0) return (x >> (8 - LOG_RB)) & ((1 << (LOG_RB - 4)) - 1);
    return ((x >> 1) & 1) << (LOG_RB - 5);
}

// byte offset of 16-byte chunk j of patch pixel pp.  MODE 2 (weight-gradient kernel): no swizzle, pixel rows
// padded from 128 to 192 bytes -- 4 consecutive pixels x 64 B (one ds_read_b64_tr_b16 block) then fall in 4
// different 64-byte bank ranges (0,192,128,64 mod 256), and a tap shift is ONE wave-uniform add.
template <int LOG_RB, int MODE> __device__ __forceinline__ int lds_chunk_off(int pp, int j) {
    if (MODE == 2) return pp * 192 + (j << 4);
    return (pp << LOG_RB) + ((j ^ swz_chunk<LOG_RB, MODE>(pp)) << 4);
}

struct PatchGeom {
    int n0, n_end, e_first_img;   // first image of the tile, one-past-last image of the expert, e*ipe
    int Y0, X0;                   // source coordinate of patch pixel (0,0)
    int PH, PW, NPIX;
    int H, W, ld, coff, cmax;      // cmax: channels [0,cmax) past coff exist, the rest reads as 0
    int dilate, shared;
    int step;                      // source pixels per patch pixel (2 for a strided 1x1 conv: only the used pixels are staged)
};

// Stage channels [c0, c0 + RB/sizeof(TL)) of the halo patch into LDS.  thread -> (16-byte chunk,
// pixel slot); the pixel -> (image,row,col) decode advances incrementally (no divisions in the loop).
// TL = element type of the LDS image: T itself, or fp8 (bf16 activations converted x * in_scale -> e4m3 between the global
// load and the LDS store: a 16-byte LDS chunk then holds 16 channels = two 16-byte loads).
template <typename T, int LOG_RB, int NTHR, int MODE, typename TL = T>
__device__ __forceinline__ void load_halo_patch(char* patch, const T* in, const PatchGeom& g, int c0, int tid,
                                                float in_scale = 1.f) {
    constexpr int RB = 1 << LOG_RB, CPR = RB / 16, LOG_CPR = LOG_RB - 4;
    constexpr int PSTEP = NTHR / CPR;
    constexpr int VE = 16 / (int)sizeof(TL);
    constexpr bool CVT = sizeof(TL) != sizeof(T);
    constexpr int UB = CVT ? 2 : 4;                   // chunks in flight per batch (a converting chunk is two loads)
    static_assert(!CVT || (sizeof(T) == 2 && sizeof(TL) == 1), "conversion on load: bf16 -> fp8 only");
    const int pj = tid & (CPR - 1);
    int pp = tid >> LOG_CPR;
    int ix = pp % g.PW;
    const int row = pp / g.PW;
    int iy = row % g.PH, pn = row / g.PH;
    while (pp < g.NPIX) {
        v4i v[UB], v2[CVT ? UB : 1];
        int dst[UB];
#pragma unroll
        for (int u = 0; u < UB; ++u) {
            dst[u] = -1;
            if (pp < g.NPIX) {
                const int n = g.n0 + pn;
                int Y = g.Y0 + iy * g.step, X = g.X0 + ix * g.step;
                bool ok = n < g.n_end && (c0 + pj * VE) < g.cmax;
                if (g.dilate) {
                    ok = ok && Y >= 0 && X >= 0 && !((Y | X) & 1);
                    Y >>= 1; X >>= 1;
                    ok = ok && Y < g.H && X < g.W;
                } else {
                    ok = ok && (unsigned)Y < (unsigned)g.H && (unsigned)X < (unsigned)g.W;
                }
                v[u] = v4i{0, 0, 0, 0};
                if (CVT) v2[u] = v4i{0, 0, 0, 0};
                if (ok) {
                    const int nin = g.shared ? (n - g.e_first_img) : n;
                    const T* src = in + (((size_t)nin * g.H + Y) * g.W + X) * g.ld + g.coff + c0 + pj * VE;
                    v[u] = ldg16(src);
                    if (CVT) v2[u] = ldg16(src + 8);
                }
                dst[u] = pp * RB + ((pj ^ swz_chunk<LOG_RB, MODE>(pp)) << 4);
                pp += PSTEP;
                ix += PSTEP;
                while (ix >= g.PW) {
                    ix -= g.PW;
                    if (++iy == g.PH) { iy = 0; ++pn; }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < UB; ++u)
            if (dst[u] >= 0) {
                if (CVT) *reinterpret_cast<v4i*>(patch + dst[u]) = cvt16_bf16_to_fp8(v[u], v2[u], in_scale);
                else *reinterpret_cast<v4i*>(patch + dst[u]) = v[u];
            }
    }
}

// Split-phase variant (T14 "issue early / write late"): issue() puts every 16-byte load of one
// patch in flight into registers; commit() writes them to LDS later (after the MFMA phase that hid
// their latency and the barrier that retired the previous patch's readers).  MAXV bounds the loads
// per thread (NPIX * CPR <= MAXV * NTHR, checked by the launcher).
template <typename T, int LOG_RB, int NTHR, int MAXV>
struct PatchStage {
    static constexpr int RB = 1 << LOG_RB, CPR = RB / 16, LOG_CPR = LOG_RB - 4;
    static constexpr int PSTEP = NTHR / CPR;
    static constexpr int VE = 16 / (int)sizeof(T);
    v4i v[MAXV];

    __device__ __forceinline__ void issue(const T* in, const PatchGeom& g, int c0, int tid) {
        const int pj = tid & (CPR - 1);
        int pp = tid >> LOG_CPR;
        int ix = pp % g.PW;
        const int row = pp / g.PW;
        int iy = row % g.PH, pn = row / g.PH;
        const bool cok = (c0 + pj * VE) < g.cmax;
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            v[u] = v4i{0, 0, 0, 0};
            if (pp < g.NPIX) {
                const int n = g.n0 + pn;
                int Y = g.Y0 + iy * g.step, X = g.X0 + ix * g.step;
                bool ok = cok && n < g.n_end;
                if (g.dilate) {
                    ok = ok && Y >= 0 && X >= 0 && !((Y | X) & 1);
                    Y >>= 1; X >>= 1;
                    ok = ok && Y < g.H && X < g.W;
                } else {
                    ok = ok && (unsigned)Y < (unsigned)g.H && (unsigned)X < (unsigned)g.W;
                }
                if (ok) {
                    const int nin = g.shared ? (n - g.e_first_img) : n;
                    v[u] = ldg16(in + (((size_t)nin * g.H + Y) * g.W + X) * g.ld + g.coff + c0 + pj * VE);
                }
            }
            pp += PSTEP;
            ix += PSTEP;
            while (ix >= g.PW) {
                ix -= g.PW;
                if (++iy == g.PH) { iy = 0; ++pn; }
            }
        }
    }

    template <int MODE> __device__ __forceinline__ void commit(char* patch, int npix, int tid) const {
        const int pj = tid & (CPR - 1);
#pragma unroll
        for (int u = 0; u < MAXV; ++u) {
            const int pp = (tid >> LOG_CPR) + u * PSTEP;
            if (pp < npix) *reinterpret_cast<v4i*>(patch + lds_chunk_off<LOG_RB, MODE>(pp, pj)) = v[u];
        }
    }
};
