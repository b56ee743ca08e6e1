#include <stdlib.h>
// Kernels that only STAGE-1 PU-Net training needs (reference trainer/train_1.py:129-141):
//   * backward of MaxPool2d(2,2) fused with the skip-connection gradient            (blocks/unet.py:52-62,71-84)
//   * backward of the ConvTranspose2d(k=2,s=2) scatter (pixel un-shuffle)           (unet.py:34-44)
//   * accumulate a channel window (gradient of torch.cat along C, punet.py:104,113)
//   * NHWC T -> NCHW f32 (the module boundary returns torch.stack(outs,1), punet.py:117-120)
//   * AutoregressiveCriterion (trainer/loss.py:86-118): per frame 0.5*CE(weight = 1 - class dice) + 0.5*Tversky, or the
//     L1 / L2 variants on one-hot targets -- forward statistics, finalize, gradient.
// All HBM-bound streaming kernels; every reduction is fixed-order (partial rows + one finalize), no float atomics.
#include "common.h"

#define DISPATCH_DT(dtype, CALL)                      \
    do {                                              \
        if ((dtype) == PMOE_DT_BF16) { using T = bf16; CALL; } \
        else if ((dtype) == PMOE_DT_F32) { using T = float; CALL; } \
        else return PMOE_ERR_ARG;                     \
    } while (0)

static inline int grid_for(long long n, int cap = 8192) {
    long long g = (n + 255) / 256;
    if (g < 1) g = 1;
    return (int)(g > cap ? cap : g);
}

// ------------------------------------------------------------------------------------------------
// dx[n,2oy+i,2ox+j,c] = (first maximum of the 2x2 window in row-major order ? dy[n,oy,ox,c] : 0) + dskip[n,2oy+i,2ox+j,c]
// (torch max_pool2d keeps the FIRST maximal element; ties are common after ReLU).  x and dskip may be channel windows.
template <typename T>
__global__ void __launch_bounds__(256) maxpool2_bwd_kernel(const T* __restrict__ x, int x_ld, int x_coff,
                                                          const T* __restrict__ dy, const T* __restrict__ dskip,
                                                          int ds_ld, int ds_coff, T* __restrict__ dx, int N, int H, int W,
                                                          int C) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, Ho = H / 2, Wo = W / 2;
    const long long total = (long long)N * Ho * Wo * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const size_t pix = ((size_t)n * H + 2 * oy) * W + 2 * ox;
        float v[4][VE], g[VE], o[4][VE];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t pq = pix + (size_t)(q >> 1) * W + (q & 1);
            unpack16<T>(ldg16(x + pq * x_ld + x_coff + cv * VE), v[q]);
            if (dskip) unpack16<T>(ldg16(dskip + pq * ds_ld + ds_coff + cv * VE), o[q]);
            else
#pragma unroll
                for (int k = 0; k < VE; ++k) o[q][k] = 0.f;
        }
        unpack16<T>(ldg16(dy + (size_t)i * VE), g);
#pragma unroll
        for (int k = 0; k < VE; ++k) {
            int best = 0;
            float m = v[0][k];
#pragma unroll
            for (int q = 1; q < 4; ++q)
                if (v[q][k] > m || v[q][k] != v[q][k]) { m = v[q][k]; best = q; }
#pragma unroll
            for (int q = 0; q < 4; ++q) o[q][k] += (q == best) ? g[k] : 0.f;
        }
#pragma unroll
        for (int q = 0; q < 4; ++q)
            stg16(dx + (pix + (size_t)(q >> 1) * W + (q & 1)) * C + cv * VE, pack16<T>(o[q]));
    }
}

// inverse of pixel_shuffle2 (punet.hip): dst[n,y,x,(dy*2+dx)*C + c] = src[n,2y+dy,2x+dx, src_coff + c]
template <typename T>
__global__ void __launch_bounds__(256) pixel_unshuffle2_kernel(const T* __restrict__ src, int src_ld, int src_coff,
                                                              T* __restrict__ dst, int dst_ld, int N, int H, int W, int C) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const long long total = (long long)N * H * W * 4 * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int q = (int)(t & 3); t >>= 2;
        const int x = (int)(t % W); t /= W;
        const int y = (int)(t % H);
        const int n = (int)(t / H);
        const v4i v = ldg16(src + (((size_t)n * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * src_ld + src_coff + cv * VE);
        stg16(dst + (((size_t)n * H + y) * W + x) * dst_ld + q * C + cv * VE, v);
    }
}

// dst[r, dst_coff + c] += src[r, src_coff + c]
template <typename T, int VEC>
__global__ void __launch_bounds__(256) add_window_kernel(const T* __restrict__ src, int src_ld, int src_coff,
                                                        T* __restrict__ dst, int dst_ld, int dst_coff, long long rows, int C) {
    const int CV = C / VEC;
    const long long total = rows * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long long r = i / CV;
        const T* s = src + (size_t)r * src_ld + src_coff + cv * VEC;
        T* d = dst + (size_t)r * dst_ld + dst_coff + cv * VEC;
        if (VEC == 1) {
            *d = from_f32<T>(to_f32(*d) + to_f32(*s));
        } else {
            float a[VEC], b[VEC];
            unpack16<T>(ldg16(s), a);
            unpack16<T>(ldg16(d), b);
#pragma unroll
            for (int k = 0; k < VEC; ++k) b[k] += a[k];
            stg16(d, pack16<T>(b));
        }
    }
}

// torch.cat of K equally wide channel windows (punet.py:104,113: four 23-class masks -> 92 channels; moe.py:311-313: F masks
// -> F*23 channels) in ONE launch: thread = one 16-byte vector of the destination row, gathered element-wise from the K
// sources (c is not a multiple of the vector width, so the per-source copies are 2-byte strided accesses), zero padding
// included.  dst[r, j] = src[j / c][r, src_coff + j % c] for j < K*c, 0 for K*c <= j < dst_c.
struct CatSrcs { const void* p[8]; };
template <typename T>
__global__ void __launch_bounds__(256) cat_windows_kernel(CatSrcs srcs, int K, int c, int src_ld, int src_coff,
                                                         T* __restrict__ dst, int dst_ld, int dst_c, long long rows) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = dst_c / VE;
    const long long total = rows * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long long r = i / CV;
        float v[VE];
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int j = cv * VE + e, k = j / c;
            v[e] = k < K ? to_f32(reinterpret_cast<const T*>(srcs.p[k])[(size_t)r * src_ld + src_coff + (j - k * c)]) : 0.f;
        }
        stg16(dst + (size_t)r * dst_ld + cv * VE, pack16<T>(v));
    }
}

// round 3: the same gather with whole 16-byte vectors on both sides of HBM (source row pitches that are a multiple of the
// vector width: the 32-channel mask rows of the PU-Net): a workgroup assembles CAT_ROWS destination rows in LDS -- the vectors
// that cover a source window are loaded aligned and scattered element-wise into the LDS rows -- and writes them out as whole
// vectors.  (0.95 -> ~0.5 ms per concatenation of the PU-Net expert step: the element-wise kernel above did 8 two-byte loads
// and 8 runtime divisions per output vector.)
constexpr int CAT_ROWS = 32;
template <typename T>
__global__ void __launch_bounds__(256) cat_windows_lds_kernel(CatSrcs srcs, int K, int c, int src_ld, int src_coff,
                                                             T* __restrict__ dst, int dst_ld, int dst_c, long long rows) {
    constexpr int VE = 16 / (int)sizeof(T);
    extern __shared__ __attribute__((aligned(16))) char cat_sm[];
    T* tile = reinterpret_cast<T*>(cat_sm);                        // [CAT_ROWS][dst_c]
    const long long r0 = (long long)blockIdx.x * CAT_ROWS;
    const int nr = (int)((rows - r0) < CAT_ROWS ? (rows - r0) : CAT_ROWS);
    const int CV = dst_c / VE, used = K * c, padc = dst_c - used;
    for (int i = threadIdx.x; i < nr * padc; i += 256) {
        const int r = i / padc;
        tile[r * dst_c + used + (i - r * padc)] = from_f32<T>(0.f);
    }
    const int v0 = src_coff / VE, nv = (src_coff + c - 1) / VE - v0 + 1, per_row = K * nv;
    for (int i = threadIdx.x; i < nr * per_row; i += 256) {
        const int r = i / per_row, rem = i - r * per_row, k = rem / nv, v = rem - k * nv;
        const T* sp = reinterpret_cast<const T*>(srcs.p[k]) + (size_t)(r0 + r) * src_ld + (size_t)(v0 + v) * VE;
        const v4i raw = ldg16(sp);
        const T* vals = reinterpret_cast<const T*>(&raw);
        T* drow = tile + r * dst_c + k * c;
#pragma unroll
        for (int e = 0; e < VE; ++e) {
            const int j = (v0 + v) * VE + e - src_coff;
            if (j >= 0 && j < c) drow[j] = vals[e];
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nr * CV; i += 256) {
        const int r = i / CV, cv = i - r * CV;
        stg16(dst + (size_t)(r0 + r) * dst_ld + cv * VE, *reinterpret_cast<const v4i*>(tile + r * dst_c + cv * VE));
    }
}

// src T [N][HW][ld] channel window [coff, coff+C)  ->  dst f32 [N][C][HW]: 64 pixels x C channels per workgroup through LDS
template <typename T>
__global__ void __launch_bounds__(256) nhwc_to_nchw_kernel(const T* __restrict__ src, int ld, int coff,
                                                          float* __restrict__ dst, int N, long long HW, int C) {
    extern __shared__ float tile[];          // [64][C + 1]
    const long long nblk = (HW + 63) / 64;
    const int n = (int)(blockIdx.x / nblk);
    const long long p0 = (blockIdx.x % nblk) * 64;
    const int np = (int)((HW - p0) < 64 ? (HW - p0) : 64);
    const int CS = C + 1;
    for (int i = threadIdx.x; i < np * C; i += 256) {
        const int p = i / C, c = i - p * C;
        tile[p * CS + c] = to_f32(src[((size_t)n * HW + p0 + p) * ld + coff + c]);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * C; i += 256) {
        const int c = i >> 6, p = i & 63;
        if (p < np) dst[((size_t)n * C + c) * HW + p0 + p] = tile[p * CS + c];
    }
}

// ------------------------------------------------------------------------------------------------
// AutoregressiveCriterion.  logits f32 [B][F][C][H][W], target int64 [B][F][H][W].
// The reference's tversky_loss reduces over batch and image ROWS only (loss.py:40 takes the axes from the target's
// rank), so its TP / FP / FN ratio is formed per (class, image COLUMN): one thread owns one column w and walks rows,
// which makes those sums thread-private (no reduction); only the class-dice / cross-entropy sums are block-reduced.
// partT [F][nsplit][3][CP][W]  (q: 0 target count, 1 sum p*[t=c], 2 sum p)   row ranges; columns are thread-private
// NS = nsplit * xblocks rows of block-reduced sums:
// partG [F][NS][4][32]     (q: 0 arg-max count, 1 arg-max & target count, 2 sum_{t=c} -log p_t, 3 target count)
//                          L1/L2 modes: partG[f][s][0][0] = sum |x - onehot| or (x - onehot)^2
template <int CP>
__global__ void __launch_bounds__(256) seg_loss_stats_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ target, int B, int F, int C,
                                                            int H, int W, int mode, float* __restrict__ partT,
                                                            float* __restrict__ partG, int nsplit) {
    const int f = blockIdx.z, split = blockIdx.y, w = blockIdx.x * 256 + threadIdx.x;
    const int NS = nsplit * gridDim.x, srow = split * gridDim.x + blockIdx.x;
    const long long rows = (long long)B * H;
    const long long r0 = rows * split / nsplit, r1 = rows * (split + 1) / nsplit;
    const size_t HW = (size_t)H * W;
    float tc[CP], ti[CP], tp[CP], gp[CP], gi[CP], ga[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) tc[c] = ti[c] = tp[c] = gp[c] = gi[c] = ga[c] = 0.f;
    float lsum = 0.f;
    if (w < W) {
        for (long long r = r0; r < r1; ++r) {
            const int b = (int)(r / H), h = (int)(r % H);
            const float* px = logits + ((size_t)(b * F + f) * C) * HW + (size_t)h * W + w;
            const int t = (int)target[((size_t)(b * F + f) * H + h) * W + w];
            float x[CP];
#pragma unroll
            for (int c = 0; c < CP; ++c) x[c] = c < C ? px[(size_t)c * HW] : -INFINITY;
            if (mode != 0) {
#pragma unroll
                for (int c = 0; c < CP; ++c)
                    if (c < C) {
                        const float d = x[c] - (c == t ? 1.f : 0.f);
                        lsum += mode == 1 ? fabsf(d) : d * d;
                    }
                continue;
            }
            float m = x[0];
            int am = 0;
#pragma unroll
            for (int c = 1; c < CP; ++c)
                if (x[c] > m) { m = x[c]; am = c; }
            float s = 0.f, xt = 0.f;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                x[c] = c < C ? expf(x[c] - m) : 0.f;
                s += x[c];
            }
            const float inv = 1.f / s;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                const float p = x[c] * inv;
                const bool it = c == t;
                tp[c] += p;
                ti[c] += it ? p : 0.f;
                tc[c] += it ? 1.f : 0.f;
                gp[c] += c == am ? 1.f : 0.f;
                gi[c] += (it && c == am) ? 1.f : 0.f;
                xt = it ? p : xt;
            }
            const float nl = -logf(xt);
#pragma unroll
            for (int c = 0; c < CP; ++c) ga[c] += c == t ? nl : 0.f;
        }
        if (mode == 0) {
            float* pt = partT + ((size_t)(f * nsplit + split) * 3) * CP * W + w;
#pragma unroll
            for (int c = 0; c < CP; ++c) {
                pt[(size_t)(0 * CP + c) * W] = tc[c];
                pt[(size_t)(1 * CP + c) * W] = ti[c];
                pt[(size_t)(2 * CP + c) * W] = tp[c];
            }
        }
    }
    // block reduction of the global class sums: xor-shuffle inside the wave, 4 wave rows through LDS
    __shared__ float red[4][4][32];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (mode != 0) {
        for (int off = 32; off > 0; off >>= 1) lsum += __shfl_xor(lsum, off);
        if (lane == 0) red[wave][0][0] = lsum;
    } else {
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            float a0 = gp[c], a1 = gi[c], a2 = ga[c], a3 = tc[c];
            for (int off = 32; off > 0; off >>= 1) {
                a0 += __shfl_xor(a0, off);
                a1 += __shfl_xor(a1, off);
                a2 += __shfl_xor(a2, off);
                a3 += __shfl_xor(a3, off);
            }
            if (lane == 0) {
                red[wave][0][c] = a0;
                red[wave][1][c] = a1;
                red[wave][2][c] = a2;
                red[wave][3][c] = a3;
            }
        }
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int q = threadIdx.x >> 5, c = threadIdx.x & 31;
        float v = 0.f;
        if (mode != 0) v = (q == 0 && c == 0) ? red[0][0][0] + red[1][0][0] + red[2][0][0] + red[3][0][0] : 0.f;
        else if (c < CP) v = red[0][q][c] + red[1][q][c] + red[2][q][c] + red[3][q][c];
        partG[((size_t)(f * NS + srow) * 4 + q) * 32 + c] = v;
    }
}

// one workgroup per frame.  coefG [F][32]: CE coefficient ce_w * w_c / sum_c(w_c * n_c);
// coefT [F][2][CP][W]: Tversky d loss / d p = coefT0[c][w] * [t = c] + coefT1[c][w];  loss[0] total, loss[1+f] per frame.
__global__ void __launch_bounds__(256) seg_loss_finalize_kernel(const float* __restrict__ partT,
                                                               const float* __restrict__ partG, int B, int F, int C, int CP,
                                                               int H, int W, int mode, int NS, int nsplit, float ce_w, float tv_w,
                                                               float alpha, float beta, float* __restrict__ coefG,
                                                               float* __restrict__ coefT, float* __restrict__ loss) {
    __shared__ float g[4][32];
    __shared__ float red[256];
    {
        const int f = blockIdx.x;
        if (threadIdx.x < 128) {
            const int q = threadIdx.x >> 5, c = threadIdx.x & 31;
            float s = 0.f;
            for (int r = 0; r < NS; ++r) s += partG[((size_t)(f * NS + r) * 4 + q) * 32 + c];
            g[q][c] = s;
        }
        __syncthreads();
        if (mode != 0) {
            if (threadIdx.x == 0) loss[1 + f] = g[0][0] / ((float)B * (float)C * (float)H * (float)W);
            return;
        }
        float rsum = 0.f;
        for (int i = threadIdx.x; i < C * W; i += 256) {
            const int c = i / W, w = i - c * W;
            float n = 0.f, I = 0.f, P = 0.f;
            for (int r = 0; r < nsplit; ++r) {
                const float* pt = partT + ((size_t)(f * nsplit + r) * 3) * CP * W + w;
                n += pt[(size_t)(0 * CP + c) * W];
                I += pt[(size_t)(1 * CP + c) * W];
                P += pt[(size_t)(2 * CP + c) * W];
            }
            const float D = I + alpha * (P - I) + beta * (n - I);
            rsum += I / D;
            const float k = tv_w / ((float)C * (float)W);
            coefT[((size_t)(f * 2 + 0) * CP + c) * W + w] = -k * (alpha * P + beta * n) / (D * D);
            coefT[((size_t)(f * 2 + 1) * CP + c) * W + w] = k * alpha * I / (D * D);
        }
        red[threadIdx.x] = rsum;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) {
            float num = 0.f, den = 0.f, wc[32];
            for (int c = 0; c < C; ++c) {      // class_dice (loss.py:6-17): 1 - 2 (inter + eps) / (pred + target + eps)
                wc[c] = 1.f - 2.f * (g[1][c] + 1e-6f) / (g[0][c] + g[3][c] + 1e-6f);
                num += wc[c] * g[2][c];
                den += wc[c] * g[3][c];
            }
            for (int c = 0; c < 32; ++c) coefG[f * 32 + c] = c < C ? ce_w * wc[c] / den : 0.f;
            loss[1 + f] = ce_w * (num / den) + tv_w * (1.f - red[0] / ((float)C * (float)W));
        }
    }
}

__global__ void seg_loss_total_kernel(float* __restrict__ loss, int F) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) s += loss[1 + f];      // AutoregressiveCriterion.forward: final_loss += loss(frame t)
    loss[0] = s;
}

template <int CP>
__global__ void __launch_bounds__(256) seg_loss_bwd_kernel(const float* __restrict__ logits,
                                                          const long long* __restrict__ target,
                                                          const float* __restrict__ coefG, const float* __restrict__ coefT,
                                                          const float* __restrict__ dloss, float* __restrict__ dlogits,
                                                          int B, int F, int C, int H, int W, int mode) {
    const size_t HW = (size_t)H * W;
    const long long total = (long long)B * F * (long long)HW;
    const float up = dloss ? dloss[0] : 1.f;
    const float invn = 1.f / ((float)B * (float)C * (float)H * (float)W);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % W);
        const long long bf = i / (long long)HW;
        const int f = (int)(bf % F);
        const size_t pix = (size_t)(i % (long long)HW);
        const float* px = logits + (size_t)bf * C * HW + pix;
        float* pg = dlogits + (size_t)bf * C * HW + pix;
        const int t = (int)target[i];
        float x[CP];
#pragma unroll
        for (int c = 0; c < CP; ++c) x[c] = c < C ? px[(size_t)c * HW] : -INFINITY;
        if (mode != 0) {
#pragma unroll
            for (int c = 0; c < CP; ++c)
                if (c < C) {
                    const float d = x[c] - (c == t ? 1.f : 0.f);
                    pg[(size_t)c * HW] = up * invn * (mode == 1 ? (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) : 2.f * d);
                }
            continue;
        }
        float m = x[0];
#pragma unroll
        for (int c = 1; c < CP; ++c) m = fmaxf(m, x[c]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            x[c] = c < C ? expf(x[c] - m) : 0.f;
            s += x[c];
        }
        const float inv = 1.f / s;
        const float* c0 = coefT + ((size_t)(f * 2 + 0) * CP) * W + w;
        const float* c1 = coefT + ((size_t)(f * 2 + 1) * CP) * W + w;
        float dp[CP], dot = 0.f, kce = 0.f;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
            x[c] *= inv;
            dp[c] = 0.f;
            if (c < C) {
                dp[c] = c1[(size_t)c * W] + (c == t ? c0[(size_t)c * W] : 0.f);
                kce = c == t ? coefG[f * 32 + c] : kce;
            }
            dot += x[c] * dp[c];
        }
#pragma unroll
        for (int c = 0; c < CP; ++c)
            if (c < C) pg[(size_t)c * HW] = up * (kce * (x[c] - (c == t ? 1.f : 0.f)) + x[c] * (dp[c] - dot));
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int pmoe_maxpool2s2_bwd(const void* x, int32_t x_ld, int32_t x_coff, const void* dy, const void* dskip, int32_t dskip_ld,
                        int32_t dskip_coff, void* dx, int32_t N, int32_t H, int32_t W, int32_t C, int32_t dtype,
                        void* stream) {
    if (N < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || x_ld % VE || x_coff % VE || x_coff + C > x_ld) return PMOE_ERR_ARG;
        if (dskip && (dskip_ld % VE || dskip_coff % VE || dskip_coff + C > dskip_ld)) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((maxpool2_bwd_kernel<T>), dim3(grid_for((long long)N * (H / 2) * (W / 2) * (C / VE))), dim3(256),
                           0, (hipStream_t)stream, (const T*)x, x_ld, x_coff, (const T*)dy, (const T*)dskip, dskip_ld,
                           dskip_coff, (T*)dx, N, H, W, C);
        return (int)hipGetLastError();
    });
}

int pmoe_pixel_unshuffle2(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t N, int32_t H,
                          int32_t W, int32_t C, int32_t dtype, void* stream) {
    if (N < 1 || H < 1 || W < 1) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || src_ld % VE || src_coff % VE || src_coff + C > src_ld || dst_ld % VE || dst_ld < 4 * C) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((pixel_unshuffle2_kernel<T>), dim3(grid_for((long long)N * H * W * 4 * (C / VE))), dim3(256), 0,
                           (hipStream_t)stream, (const T*)src, src_ld, src_coff, (T*)dst, dst_ld, N, H, W, C);
        return (int)hipGetLastError();
    });
}

int pmoe_add_window(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                    int64_t rows, int32_t C, int32_t dtype, void* stream) {
    if (rows < 0 || C < 1 || src_coff + C > src_ld || dst_coff + C > dst_ld) return PMOE_ERR_ARG;
    if (rows == 0) return 0;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        const bool vec = !(C % VE) && !(src_ld % VE) && !(src_coff % VE) && !(dst_ld % VE) && !(dst_coff % VE);
        if (vec)
            hipLaunchKernelGGL((add_window_kernel<T, VE>), dim3(grid_for(rows * (C / VE))), dim3(256), 0, (hipStream_t)stream,
                               (const T*)src, src_ld, src_coff, (T*)dst, dst_ld, dst_coff, (long long)rows, C);
        else
            hipLaunchKernelGGL((add_window_kernel<T, 1>), dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)src, src_ld, src_coff, (T*)dst, dst_ld, dst_coff, (long long)rows, C);
        return (int)hipGetLastError();
    });
}

int pmoe_cat_windows(const void* const* srcs, int32_t K, int32_t c, int32_t src_ld, int32_t src_coff, void* dst,
                     int32_t dst_ld, int32_t dst_c, int64_t rows, int32_t dtype, void* stream) {
    if (!srcs || K < 1 || K > 8 || c < 1 || src_coff < 0 || src_coff + c > src_ld || rows < 0 || dst_c > dst_ld ||
        (long long)K * c > dst_c)
        return PMOE_ERR_ARG;
    if (rows == 0) return 0;
    CatSrcs cs;
    for (int k = 0; k < 8; ++k) cs.p[k] = k < K ? srcs[k] : nullptr;       /* srcs is a HOST array of device pointers */
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (dst_c % VE || dst_ld % VE) return PMOE_ERR_ARG;
        bool aligned = src_ld % VE == 0 && (size_t)CAT_ROWS * dst_c * sizeof(T) <= 48 * 1024;
        for (int k = 0; k < K; ++k) aligned = aligned && ((uintptr_t)srcs[k] & 15) == 0;
        static int lds_on = -1;             // PMOE_CAT_LDS=0: the element-wise kernel for every call (A/B)
        if (lds_on < 0) { const char* ev = getenv("PMOE_CAT_LDS"); lds_on = ev ? atoi(ev) : 1; }
        if (aligned && lds_on) {
            hipLaunchKernelGGL((cat_windows_lds_kernel<T>), dim3((unsigned)((rows + CAT_ROWS - 1) / CAT_ROWS)), dim3(256),
                               (size_t)CAT_ROWS * dst_c * sizeof(T), (hipStream_t)stream, cs, K, c, src_ld, src_coff, (T*)dst,
                               dst_ld, dst_c, (long long)rows);
            return (int)hipGetLastError();
        }
        hipLaunchKernelGGL((cat_windows_kernel<T>), dim3(grid_for(rows * (dst_c / VE))), dim3(256), 0, (hipStream_t)stream, cs,
                           K, c, src_ld, src_coff, (T*)dst, dst_ld, dst_c, (long long)rows);
        return (int)hipGetLastError();
    });
}

int pmoe_nhwc_to_nchw(const void* src, int32_t src_ld, int32_t src_coff, float* dst, int32_t N, int64_t HW, int32_t C,
                      int32_t dtype, void* stream) {
    if (N < 1 || HW < 1 || C < 1 || src_coff + C > src_ld || C > 1024) return PMOE_ERR_ARG;
    const long long nblk = (HW + 63) / 64 * N;
    if (nblk > 0x7fffffffLL) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((nhwc_to_nchw_kernel<T>), dim3((unsigned)nblk), dim3(256), 64 * (C + 1) * sizeof(float),
                           (hipStream_t)stream, (const T*)src, src_ld, src_coff, dst, N, (long long)HW, C);
        return (int)hipGetLastError();
    });
}

static int seg_cp(int C) { return C <= 8 ? 8 : C <= 16 ? 16 : C <= 24 ? 24 : 32; }

int pmoe_seg_loss_rows(int32_t B, int32_t H, int32_t W) {
    /* partial rows per frame = nsplit * xblocks */
    if (B < 1 || H < 1 || W < 1) return PMOE_ERR_ARG;
    const long long rows = (long long)B * H;
    int nsplit = (int)(rows < 64 ? rows : 64);
    return nsplit * ((W + 255) / 256);
}

int pmoe_seg_loss_fwd(const float* logits, const int64_t* target, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W,
                      int32_t mode, float ce_weight, float tversky_weight, float alpha, float beta, float* partT, float* partG,
                      float* coefG, float* coefT, float* loss, void* stream) {
    if (B < 1 || F < 1 || C < 2 || C > 32 || H < 1 || W < 1 || mode < 0 || mode > 2 || !partG || !loss) return PMOE_ERR_ARG;
    if (mode == 0 && (!partT || !coefG || !coefT)) return PMOE_ERR_ARG;
    const int xb = (W + 255) / 256;
    const int NS = pmoe_seg_loss_rows(B, H, W);
    const int nsplit = NS / xb, CP = seg_cp(C);
    const dim3 grid(xb, nsplit, F);
#define SEG_STATS(CPV)                                                                                                   \
    hipLaunchKernelGGL((seg_loss_stats_kernel<CPV>), grid, dim3(256), 0, (hipStream_t)stream, logits,                    \
                       (const long long*)target, B, F, C, H, W, mode, partT, partG, nsplit)
    if (CP == 8) SEG_STATS(8);
    else if (CP == 16) SEG_STATS(16);
    else if (CP == 24) SEG_STATS(24);
    else SEG_STATS(32);
#undef SEG_STATS
    hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(F), dim3(256), 0, (hipStream_t)stream, partT, partG, B, F, C, CP, H, W,
                       mode, NS, nsplit, ce_weight, tversky_weight, alpha, beta, coefG, coefT, loss);
    hipLaunchKernelGGL(seg_loss_total_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, loss, F);
    return (int)hipGetLastError();
}

int pmoe_seg_loss_bwd(const float* logits, const int64_t* target, const float* coefG, const float* coefT, const float* dloss,
                      float* dlogits, int32_t B, int32_t F, int32_t C, int32_t H, int32_t W, int32_t mode, void* stream) {
    if (B < 1 || F < 1 || C < 2 || C > 32 || H < 1 || W < 1 || mode < 0 || mode > 2 || !dlogits) return PMOE_ERR_ARG;
    if (mode == 0 && (!coefG || !coefT)) return PMOE_ERR_ARG;
    const int CP = seg_cp(C);
    const int grid = grid_for((long long)B * F * H * W, 16384);
#define SEG_BWD(CPV)                                                                                                     \
    hipLaunchKernelGGL((seg_loss_bwd_kernel<CPV>), dim3(grid), dim3(256), 0, (hipStream_t)stream, logits,                \
                       (const long long*)target, coefG, coefT, dloss, dlogits, B, F, C, H, W, mode)
    if (CP == 8) SEG_BWD(8);
    else if (CP == 16) SEG_BWD(16);
    else if (CP == 24) SEG_BWD(24);
    else SEG_BWD(32);
#undef SEG_BWD
    return (int)hipGetLastError();
}

int pmoe_seg_loss_cp(int32_t C) { return (C < 2 || C > 32) ? PMOE_ERR_ARG : seg_cp(C); }

}  // extern "C"
