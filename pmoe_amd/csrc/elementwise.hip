// HBM-bound passes of the PMoE backbone on gfx950: BatchNorm statistics / apply / backward,
// max-pool, global-average-pool, ECA channel attention, layout conversion.  All are streaming
// kernels: 16-byte vector accesses per lane, channel vectors innermost (NHWC), reductions done
// per workgroup in LDS and combined from fixed-order partial rows (bitwise reproducible, no float
// atomics).  Each C-ABI wrapper cites the reference call site in include/pmoe_hip.h.
#include "common.h"

#define DISPATCH_DT(dtype, CALL)                      \
    do {                                              \
        if ((dtype) == PMOE_DT_BF16) { using T = bf16; CALL; } \
        else if ((dtype) == PMOE_DT_F32) { using T = float; CALL; } \
        else return PMOE_ERR_ARG;                     \
    } while (0)

static inline bool pow2(int v) { return v > 0 && !(v & (v - 1)); }

// ------------------------------------------------------------------------------------------------
// Column statistics.  MODE 0: sum x, sum x^2.  MODE 1 (BN backward): g = dy*(relu? y>0), sums of g
// and g*xhat.  block = 256 threads = (256/CV row lanes) x (CV channel vectors).
template <typename T, int MODE>
__global__ void __launch_bounds__(256) colstats_kernel(const T* __restrict__ x, const T* __restrict__ dy,
                                                      const T* __restrict__ y, const float* __restrict__ mean,
                                                      const float* __restrict__ invstd, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, long long rpe, int C, int ld,
                                                      int coff, int relu, float* __restrict__ part, int nparts,
                                                      float* __restrict__ shiftc, T* __restrict__ gmask) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, RL = 256 / CV;
    const int tid = threadIdx.x, cv = tid % CV, rl = tid / CV;
    const int e = blockIdx.y, pi = blockIdx.x;
    const long long rpp = (rpe + nparts - 1) / nparts;
    long long r0 = (long long)pi * rpp, r1 = r0 + rpp;
    if (r1 > rpe) r1 = rpe;
    float s1[VE], s2[VE], mu[VE], is[VE], sc[VE], sh[VE];
    const bool remask = MODE && relu && !y;       // ReLU mask recomputed from x (no residual): y is not read
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        s1[i] = s2[i] = 0.f;
        mu[i] = MODE ? mean[e * C + cv * VE + i] : 0.f;
        is[i] = MODE ? invstd[e * C + cv * VE + i] : 0.f;
        sc[i] = remask ? scale[e * C + cv * VE + i] : 0.f;
        sh[i] = remask ? shift[e * C + cv * VE + i] : 0.f;
    }
    const size_t ebase = (size_t)e * rpe;
    // MODE 0 accumulates DEVIATIONS from one sample of the channel (row 0 of the expert): sum(x-c), sum(x-c)^2.
    // var = E[(x-c)^2] - E[x-c]^2 then has no catastrophic cancellation even for a near-constant channel
    // (|mean| >> std), where the textbook E[x^2]-mean^2 in f32 loses all digits of the variance.
    float c0v[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) c0v[i] = 0.f;
    if (MODE == 0 && shiftc) {
        unpack16<T>(ldg16(x + ebase * ld + coff + cv * VE), c0v);
        if (pi == 0 && rl == 0) {
#pragma unroll
            for (int i = 0; i < VE; ++i) shiftc[e * C + cv * VE + i] = c0v[i];
        }
    }
    for (long long r = r0 + rl; r < r1; r += RL) {
        const size_t off = (ebase + r) * ld + coff + cv * VE;
        float xv[VE];
        unpack16<T>(ldg16(x + off), xv);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { const float d = xv[i] - c0v[i]; s1[i] += d; s2[i] += d * d; }
        } else {
            float gv[VE], yv[VE];
            unpack16<T>(ldg16(dy + off), gv);
            if (remask) {
#pragma unroll
                for (int i = 0; i < VE; ++i) gv[i] = ((xv[i] - mu[i]) * sc[i] + sh[i]) > 0.f ? gv[i] : 0.f;
            } else if (relu) {
                unpack16<T>(ldg16(y + off), yv);
#pragma unroll
                for (int i = 0; i < VE; ++i) gv[i] = yv[i] > 0.f ? gv[i] : 0.f;
            }
            if (gmask) stg16(gmask + off, pack16<T>(gv));      // the masked gradient: what the apply pass and the residual path consume
#pragma unroll
            for (int i = 0; i < VE; ++i) { s1[i] += gv[i]; s2[i] += gv[i] * (xv[i] - mu[i]) * is[i]; }
        }
    }
    __shared__ float red[2][256 * 8 / 1];   // [which][rl][cv*VE] flattened: 256*VE floats max (VE<=8)
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        red[0][rl * C + cv * VE + i] = s1[i];
        red[1][rl * C + cv * VE + i] = s2[i];
    }
    __syncthreads();
    for (int c = tid; c < 2 * C; c += 256) {
        const int which = c / C, cc = c % C;
        float s = 0.f;
        for (int k = 0; k < RL; ++k) s += red[which][k * C + cc];
        part[(((size_t)e * nparts + pi) * 2 + which) * C + cc] = s;
    }
}

template <typename T>
static int colstats_launch(const void* x, const void* dy, const void* y, const float* mean, const float* invstd,
                           const float* scale, const float* shift, long long rpe, int E, int C, int ld, int coff, int relu, float* part, int nparts, int mode,
                           float* shiftc, hipStream_t st, void* gmask = nullptr) {
    constexpr int VE = 16 / (int)sizeof(T);
    if (C % VE || !pow2(C / VE) || C / VE > 256 || nparts < 1) return PMOE_ERR_ARG;
    dim3 grid(nparts, E), block(256);
    if (mode == 0)
        hipLaunchKernelGGL((colstats_kernel<T, 0>), grid, block, 0, st, (const T*)x, nullptr, nullptr, nullptr, nullptr,
                           nullptr, nullptr, rpe, C, ld, coff, 0, part, nparts, shiftc, nullptr);
    else
        hipLaunchKernelGGL((colstats_kernel<T, 1>), grid, block, 0, st, (const T*)x, (const T*)dy, (const T*)y, mean,
                           invstd, scale, shift, rpe, C, ld, coff, relu, part, nparts, nullptr, (T*)gmask);
    return (int)hipGetLastError();
}

__global__ void __launch_bounds__(256) reduce_partials_kernel(const float* __restrict__ in, float* __restrict__ out,
                                                             int nin, int nout, int width) {
    const int e = blockIdx.y, po = blockIdx.x;
    const int per = (nin + nout - 1) / nout;
    int i0 = po * per, i1 = i0 + per;
    if (i1 > nin) i1 = nin;
    for (int c = threadIdx.x; c < width; c += 256) {
        float s = 0.f;
        for (int i = i0; i < i1; ++i) s += in[((size_t)e * nin + i) * width + c];
        out[((size_t)e * nout + po) * width + c] = s;
    }
}

// finalize: block = 32 channels x 32 partial lanes (round 3: up to 2048 partial rows are folded HERE in fixed order instead of by
// a reduce_partials launch in front of every finalize: ~250 launches fewer in the stage-1 step, ~200 in the PU-Net expert's).
// Round 4: these 44 launches per headline step ran 13-15 us each for ~1 MB of partial rows -- E x C/64 workgroups whose threads
// walked up to 128 rows one dependent load at a time.  Now 32 lanes per channel, four independent accumulation chains per lane
// (fixed association: ((a0 + a1) + (a2 + a3)), bit-reproducible), i.e. <= 16 dependent steps with 4 loads in flight each.
constexpr int FIN_LANES = 32, FIN_CH = 32;
__device__ __forceinline__ void fin_fold(const float* __restrict__ part, int nparts, int C, int e, int c, int pl, bool on,
                                         double (*red)[FIN_LANES][FIN_CH], double& t1, double& t2) {
    double a1[4] = {0.0, 0.0, 0.0, 0.0}, a2[4] = {0.0, 0.0, 0.0, 0.0};
    if (on) {
        const float* p = part + ((size_t)e * nparts * 2) * C + c;
        int i = pl;
        for (; i + 3 * FIN_LANES < nparts; i += 4 * FIN_LANES) {
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                a1[k] += p[((size_t)(i + k * FIN_LANES) * 2 + 0) * C];
                a2[k] += p[((size_t)(i + k * FIN_LANES) * 2 + 1) * C];
            }
        }
        for (; i < nparts; i += FIN_LANES) {                  // (< 4 rows left for this lane)
            a1[0] += p[((size_t)i * 2 + 0) * C];
            a2[0] += p[((size_t)i * 2 + 1) * C];
        }
    }
    const int cl = threadIdx.x & (FIN_CH - 1);
    red[0][pl][cl] = (a1[0] + a1[1]) + (a1[2] + a1[3]);
    red[1][pl][cl] = (a2[0] + a2[1]) + (a2[2] + a2[3]);
    __syncthreads();
    t1 = t2 = 0.0;
    if (pl == 0) {
#pragma unroll
        for (int k = 0; k < FIN_LANES; ++k) { t1 += red[0][k][cl]; t2 += red[1][k][cl]; }
    }
}

__global__ void __launch_bounds__(FIN_CH * FIN_LANES) bn_finalize_kernel(const float* __restrict__ part, int nparts, long long count,
                                                         const float* const* gamma, const float* const* beta,
                                                         float* const* rmean, float* const* rvar, float momentum,
                                                         float eps, int training, float* scale, float* shift,
                                                         float* mean_o, float* invstd_o, int C,
                                                         const float* __restrict__ shiftc) {
    const int e = blockIdx.y, c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), pl = threadIdx.x / FIN_CH;
    __shared__ double red[2][FIN_LANES][FIN_CH];          // (partial rows folded in double: up to 2048 f32 rows per channel)
    double t1, t2;
    fin_fold(part, nparts, C, e, c, pl, c < C && training, red, t1, t2);
    if (pl == 0 && c < C) {
        float mean, var;
        if (training) {
            const double md = t1 / (double)count;                  // mean of the deviations x - c
            double v = t2 / (double)count - md * md;
            if (v < 0.0) v = 0.0;
            const double m = md + (shiftc ? (double)shiftc[e * C + c] : 0.0);
            mean = (float)m;
            var = (float)v;
            if (rmean && rmean[e]) {
                const float unb = count > 1 ? (float)(v * (double)count / (double)(count - 1)) : var;
                rmean[e][c] = (1.f - momentum) * rmean[e][c] + momentum * mean;
                rvar[e][c] = (1.f - momentum) * rvar[e][c] + momentum * unb;
            }
        } else {
            mean = rmean[e][c];
            var = rvar[e][c];
        }
        const float is = rsqrtf(var + eps);
        const float g = gamma ? gamma[e][c] : 1.f, b = beta ? beta[e][c] : 0.f;
        scale[e * C + c] = g * is;
        shift[e * C + c] = b;                    // beta: consumers evaluate (x - mean)*scale + beta (no cancellation)
        mean_o[e * C + c] = mean;
        invstd_o[e * C + c] = is;
    }
}

__global__ void __launch_bounds__(FIN_CH * FIN_LANES) bn_bwd_finalize_kernel(const float* __restrict__ part, int nparts,
                                                             long long count, float* dgamma, float* dbeta, float* c1,
                                                             float* c2, int C) {
    const int e = blockIdx.y, c = blockIdx.x * FIN_CH + (threadIdx.x & (FIN_CH - 1)), pl = threadIdx.x / FIN_CH;
    __shared__ double red[2][FIN_LANES][FIN_CH];
    double t1, t2;
    fin_fold(part, nparts, C, e, c, pl, c < C, red, t1, t2);
    if (pl == 0 && c < C) {
        if (dbeta) dbeta[e * C + c] = (float)t1;
        if (dgamma) dgamma[e * C + c] = (float)t2;
        c1[e * C + c] = (float)(t1 / (double)count);
        c2[e * C + c] = (float)(t2 / (double)count);
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res,
                                                      T* __restrict__ y, const float* __restrict__ scale,
                                                      const float* __restrict__ shift, const float* __restrict__ mean,
                                                      long long rpe, int C, int relu, int y_ld, int y_coff, int log_cv,
                                                      unsigned char* __restrict__ y8, float in_scale) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;                       // power of two <= 256*gridDim.x: a thread keeps ONE channel vector
    const int e = blockIdx.y;
    const long long nvec = rpe * CV;
    const size_t ebase = (size_t)e * rpe * C;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cv = (int)(i0 % CV);
    float sc[VE], sh[VE], mu[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) {
        sc[k] = scale[e * C + cv * VE + k]; sh[k] = shift[e * C + cv * VE + k]; mu[k] = mean[e * C + cv * VE + k];
    }
    for (long long i = i0; i < nvec; i += (long long)gridDim.x * 256) {
        float xv[VE];
        const size_t off = ebase + (size_t)i * VE;
        unpack16<T>(ldg16(x + off), xv);
#pragma unroll
        for (int k = 0; k < VE; ++k) xv[k] = (xv[k] - mu[k]) * sc[k] + sh[k];      // centred: no cancellation
        if (res) {
            float rv[VE];
            unpack16<T>(ldg16(res + off), rv);
#pragma unroll
            for (int k = 0; k < VE; ++k) xv[k] += rv[k];
        }
        if (relu) {
#pragma unroll
            for (int k = 0; k < VE; ++k) xv[k] = fmaxf(xv[k], 0.f);
        }
        // dense output, or a channel window [y_coff, y_coff + C) of rows y_ld wide (skip-concatenation buffers)
        const size_t yoff = y_ld == C ? off : ((size_t)e * rpe + (size_t)(i >> log_cv)) * y_ld + y_coff + (size_t)cv * VE;
        const v4i pk = pack16<T>(xv);
        stg16(y + yoff, pk);
        if constexpr (sizeof(T) == 2) {
            // BASELINE config 5: the same activation as e4m3(bf16(y) * in_scale) bytes, dense [rows][C], for the block-scaled fp8
            // convolution that consumes it (the quantiser of conv_common.h: clamp to +-448, round to nearest even)
            if (y8) {
                float q[VE];
                unpack16<T>(pk, q);                      // the STORED (bf16-rounded) value is what the policy quantises
                int w0 = 0, w1 = 0;
#pragma unroll
                for (int k = 0; k < VE; ++k) q[k] = fminf(fmaxf(q[k] * in_scale, -PMOE_FP8_MAX), PMOE_FP8_MAX);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[0], q[1], w0, false);
                w0 = __builtin_amdgcn_cvt_pk_fp8_f32(q[2], q[3], w0, true);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[4], q[5], w1, false);
                w1 = __builtin_amdgcn_cvt_pk_fp8_f32(q[6], q[7], w1, true);
                *reinterpret_cast<int2*>(y8 + ebase + (size_t)i * VE) = int2{w0, w1};
            }
        }
    }
}

template <typename T>
__global__ void __launch_bounds__(256) bn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ y,
                                                          const T* __restrict__ x, const float* __restrict__ mean,
                                                          const float* __restrict__ invstd,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ c1,
                                                          const float* __restrict__ c2, T* __restrict__ dx,
                                                          T* __restrict__ gm, long long rpe, int C, int relu) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const int e = blockIdx.y;
    const long long nvec = rpe * CV;
    const size_t ebase = (size_t)e * rpe * C;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cv = (int)(threadIdx.x & (CV - 1)); // CV is a power of two <= 256: constant per thread over the grid stride
    // dx = scale*(g - c1 - xhat*c2) = g*A + (x - mean)*Bx + K   with per-channel A, Bx, K (centred x: stable)
    float A[VE], Bx[VE], K[VE], sc[VE], sh[VE], mu[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) {
        const int c = e * C + cv * VE + k;
        sc[k] = scale[c];
        sh[k] = shift ? shift[c] : 0.f;
        mu[k] = mean[c];
        A[k] = sc[k];
        Bx[k] = -sc[k] * invstd[c] * c2[c];
        K[k] = -sc[k] * c1[c];
    }
    const bool remask = relu && !y;
    for (long long i = i0; i < nvec; i += (long long)gridDim.x * 256) {
        const size_t off = ebase + (size_t)i * VE;
        float gv[VE], xv[VE], o[VE];
        unpack16<T>(ldg16(dy + off), gv);
        unpack16<T>(ldg16(x + off), xv);
        if (remask) {
#pragma unroll
            for (int k = 0; k < VE; ++k) gv[k] = ((xv[k] - mu[k]) * sc[k] + sh[k]) > 0.f ? gv[k] : 0.f;
        } else if (relu) {
            float yv[VE];
            unpack16<T>(ldg16(y + off), yv);
#pragma unroll
            for (int k = 0; k < VE; ++k) gv[k] = yv[k] > 0.f ? gv[k] : 0.f;
        }
#pragma unroll
        for (int k = 0; k < VE; ++k) o[k] = gv[k] * A[k] + ((xv[k] - mu[k]) * Bx[k] + K[k]);
        stg16(dx + off, pack16<T>(o));
        if (gm) stg16(gm + off, pack16<T>(gv));
    }
}

// ------------------------------------------------------------------------------------------------
// Round 4: y = relu(BatchNorm(z)) AND MaxPool2d(2, 2)(y) in one pass (the U-Net's down blocks, blocks/unet.py:54-68: the last
// BatchNorm + ReLU of a block writes the skip half of the concatenation buffer, the pooled tensor feeds the next block): a thread
// owns one 2x2 pixel quad x one 16-byte channel vector -- 4 loads, 4 stores + the pooled store; the separate pooling launch read
// y again (2.5 ms per PU-Net expert step).  max of the ROUNDED values = rounded max (rounding is monotone): bit-identical to the pair.
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_pool2_kernel(const T* __restrict__ x, T* __restrict__ y, T* __restrict__ pooled,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            const float* __restrict__ mean, int ipe, int H, int W, int C, int relu,
                                                            int y_ld, int y_coff) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const int e = blockIdx.y, Ho = H / 2, Wo = W / 2;
    const long long nq = (long long)ipe * Ho * Wo * CV;
    const long long i0 = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cv = (int)(i0 % CV);                   // CV is a power of two dividing 256 * gridDim.x: constant over the grid stride
    float sc[VE], sh[VE], mu[VE];
#pragma unroll
    for (int k = 0; k < VE; ++k) {
        sc[k] = scale[e * C + cv * VE + k]; sh[k] = shift[e * C + cv * VE + k]; mu[k] = mean[e * C + cv * VE + k];
    }
    const size_t rpe = (size_t)ipe * H * W;
    for (long long i = i0; i < nq; i += (long long)gridDim.x * 256) {
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const size_t r00 = (size_t)e * rpe + ((size_t)n * H + 2 * oy) * W + 2 * ox;
        float m[VE];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t r = r00 + (q >> 1) * (size_t)W + (q & 1);
            float xv[VE];
            unpack16<T>(ldg16(x + r * C + (size_t)cv * VE), xv);
#pragma unroll
            for (int k = 0; k < VE; ++k) {
                xv[k] = (xv[k] - mu[k]) * sc[k] + sh[k];
                if (relu) xv[k] = fmaxf(xv[k], 0.f);
            }
            const v4i pk = pack16<T>(xv);
            stg16(y + r * y_ld + y_coff + (size_t)cv * VE, pk);
            float rr[VE];
            unpack16<T>(pk, rr);                     // the pooled value is the max of the STORED values
#pragma unroll
            for (int k = 0; k < VE; ++k) m[k] = q ? fmaxf(m[k], rr[k]) : rr[k];
        }
        stg16(pooled + ((size_t)e * (rpe / 4) + ((size_t)n * Ho + oy) * Wo + ox) * C + (size_t)cv * VE, pack16<T>(m));
    }
}

// ------------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, pad 1).  First maximum in (row, col) scan order wins, as in
// ATen's CPU kernel; the winning tap (0..8) is kept per element for the backward gather.
template <typename T>
__global__ void __launch_bounds__(256) maxpool_fwd_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                         uint8_t* __restrict__ am, int N, int H, int W, int C, int Ho,
                                                         int Wo) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const long long total = (long long)N * Ho * Wo * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        float best[VE];
        int bi[VE];
#pragma unroll
        for (int k = 0; k < VE; ++k) { best[k] = -INFINITY; bi[k] = 0; }
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int yy = 2 * oy - 1 + r, xx = 2 * ox - 1 + q;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    float v[VE];
                    unpack16<T>(ldg16(x + (((size_t)n * H + yy) * W + xx) * C + cv * VE), v);
#pragma unroll
                    for (int k = 0; k < VE; ++k)
                        if (first || v[k] > best[k]) { best[k] = v[k]; bi[k] = r * 3 + q; }
                    first = false;
                }
            }
        const size_t off = (size_t)i * VE;
        stg16(y + off, pack16<T>(best));
#pragma unroll
        for (int k = 0; k < VE; ++k) am[off + k] = (uint8_t)bi[k];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) maxpool_bwd_kernel(const T* __restrict__ dy, const uint8_t* __restrict__ am,
                                                         T* __restrict__ dx, int N, int H, int W, int C, int Ho,
                                                         int Wo) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const long long total = (long long)N * H * W * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int xx = (int)(t % W); t /= W;
        const int yy = (int)(t % H);
        const int n = (int)(t / H);
        float g[VE];
#pragma unroll
        for (int k = 0; k < VE; ++k) g[k] = 0.f;
        const int oy_lo = yy >> 1, oy_hi = (yy + 1) >> 1, ox_lo = xx >> 1, ox_hi = (xx + 1) >> 1;
        for (int oy = oy_lo; oy <= oy_hi; ++oy)
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (oy >= Ho || ox >= Wo) continue;
                const int r = yy - (2 * oy - 1), q = xx - (2 * ox - 1);
                const size_t off = ((((size_t)n * Ho + oy) * Wo + ox) * CV + cv) * VE;
                float d[VE];
                unpack16<T>(ldg16(dy + off), d);
                const int tap = r * 3 + q;
#pragma unroll
                for (int k = 0; k < VE; ++k)
                    if (am[off + k] == tap) g[k] += d[k];
            }
        stg16(dx + (size_t)i * VE, pack16<T>(g));
    }
}

// ------------------------------------------------------------------------------------------------
// per-(image, channel) partial sums over HW of a (or a*b).  grid (nparts, N)
template <typename T>
__global__ void __launch_bounds__(256) gap_partial_kernel(const T* __restrict__ a, const T* __restrict__ b,
                                                         float* __restrict__ part, long long HW, int C, int nparts,
                                                         int b_ipe) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, RL = 256 / CV;
    const int tid = threadIdx.x, cv = tid % CV, rl = tid / CV;
    const int n = blockIdx.y, pi = blockIdx.x;
    const long long rpp = (HW + nparts - 1) / nparts;
    long long r0 = (long long)pi * rpp, r1 = r0 + rpp;
    if (r1 > HW) r1 = HW;
    const int nb = b_ipe > 0 ? n % b_ipe : n;
    float s[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) s[i] = 0.f;
    if (rl >= RL) r0 = r1;                 // channel-vector counts that do not divide 256 leave a few idle lanes
    for (long long r = r0 + rl; r < r1; r += RL) {
        float av[VE];
        unpack16<T>(ldg16(a + ((size_t)n * HW + r) * C + cv * VE), av);
        if (b) {
            float bv[VE];
            unpack16<T>(ldg16(b + ((size_t)nb * HW + r) * C + cv * VE), bv);
#pragma unroll
            for (int i = 0; i < VE; ++i) s[i] += av[i] * bv[i];
        } else {
#pragma unroll
            for (int i = 0; i < VE; ++i) s[i] += av[i];
        }
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int i = 0; i < VE; ++i)
        if (rl < RL) red[rl * C + cv * VE + i] = s[i];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < RL; ++k) t += red[k * C + c];
        part[((size_t)n * nparts + pi) * C + c] = t;
    }
}

// y = [relu]((x - mean) * scale + shift) AND the per-(image, channel) partial sums over HW of the stored y, in gap_partial_kernel's
// own decomposition and summation order (bit-identical `part`): the ECA block that follows a BatchNorm (stem: bn -> relu ->
// EfficientBlock, basics.py:113-123) gets its global-average-pool input from the pass that writes the activation instead of
// re-reading 2.1 GB.  grid (nparts, N)
template <typename T>
__global__ void __launch_bounds__(256) bn_apply_gap_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                          const float* __restrict__ scale, const float* __restrict__ shift,
                                                          const float* __restrict__ mean, float* __restrict__ part,
                                                          long long HW, int C, int nparts, int ipe, int relu) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, RL = 256 / CV;
    const int tid = threadIdx.x, cv = tid % CV, rl = tid / CV;
    const int n = blockIdx.y, pi = blockIdx.x, e = n / ipe;
    const long long rpp = (HW + nparts - 1) / nparts;
    long long r0 = (long long)pi * rpp, r1 = r0 + rpp;
    if (r1 > HW) r1 = HW;
    float s[VE], sc[VE], sh[VE], mu[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        s[i] = 0.f;
        const int c = e * C + (rl < RL ? cv : 0) * VE + i;
        sc[i] = scale[c]; sh[i] = shift[c]; mu[i] = mean[c];
    }
    if (rl >= RL) r0 = r1;
    for (long long r = r0 + rl; r < r1; r += RL) {
        const size_t off = ((size_t)n * HW + r) * C + cv * VE;
        float xv[VE];
        unpack16<T>(ldg16(x + off), xv);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            xv[i] = (xv[i] - mu[i]) * sc[i] + sh[i];
            if (relu) xv[i] = fmaxf(xv[i], 0.f);
        }
        const v4i pk = pack16<T>(xv);
        stg16(y + off, pk);
        float yv[VE];
        unpack16<T>(pk, yv);                                   // the STORED value, as gap_partial_kernel would read it back
#pragma unroll
        for (int i = 0; i < VE; ++i) s[i] += yv[i];
    }
    __shared__ float red[256 * 8];
#pragma unroll
    for (int i = 0; i < VE; ++i)
        if (rl < RL) red[rl * C + cv * VE + i] = s[i];
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float t = 0.f;
        for (int k = 0; k < RL; ++k) t += red[k * C + c];
        part[((size_t)n * nparts + pi) * C + c] = t;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gap_finish_kernel(const float* __restrict__ part, T* __restrict__ out, int N,
                                                        int C, int nparts, float inv_hw, int out_ld, int out_coff) {
    const long long total = (long long)N * C;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int c = (int)(i % C), n = (int)(i / C);
        float s = 0.f;
        for (int k = 0; k < nparts; ++k) s += part[((size_t)n * nparts + k) * C + c];
        out[(size_t)n * out_ld + out_coff + c] = from_f32<T>(s * inv_hw);
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gap_bwd_kernel(const T* __restrict__ g, T* __restrict__ dx, long long HW, int C,
                                                     int g_ld, int g_coff, float inv_hw) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const int n = blockIdx.y;
    const long long nvec = HW * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        float v[VE];
        unpack16<T>(ldg16(g + (size_t)n * g_ld + g_coff + cv * VE), v);
#pragma unroll
        for (int k = 0; k < VE; ++k) v[k] *= inv_hw;
        stg16(dx + ((size_t)n * HW) * C + (size_t)i * VE, pack16<T>(v));
    }
}

// ------------------------------------------------------------------------------------------------
// ECA gate: one block per output image n (expert e = n / ipe); gap partials indexed by
// n (in_ipe == 0) or n % ipe (input shared by the experts).
__global__ void __launch_bounds__(256) eca_gate_kernel(const float* __restrict__ part, int nparts, float inv_hw,
                                                      const float* const* w, int k, float* __restrict__ gate,
                                                      float* __restrict__ gapmean, int ipe, int in_ipe, int C,
                                                      int creal) {
    const int n = blockIdx.x, e = n / ipe;
    const int ni = in_ipe > 0 ? n % ipe : n;
    __shared__ float gm[1024];
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int i = 0; i < nparts; ++i) s += part[((size_t)ni * nparts + i) * C + c];
        gm[c] = c < creal ? s * inv_hw : 0.f;
    }
    __syncthreads();
    const float* we = w[e];
    const int pad = k / 2;
    for (int c = threadIdx.x; c < C; c += 256) {
        float z = 0.f;
        for (int j = 0; j < k; ++j) {
            const int cc = c + j - pad;
            if (cc >= 0 && cc < creal) z += we[j] * gm[cc];
        }
        gate[(size_t)n * C + c] = c < creal ? 1.f / (1.f + expf(-z)) : 0.f;
        gapmean[(size_t)n * C + c] = gm[c];
    }
}

template <typename T>
__global__ void __launch_bounds__(256) eca_scale_kernel(const T* __restrict__ x, const float* __restrict__ gate,
                                                       T* __restrict__ y, long long HW, int C, int x_ipe) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const int n = blockIdx.y;
    const int nx = x_ipe > 0 ? n % x_ipe : n;
    const long long nvec = HW * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        float v[VE];
        unpack16<T>(ldg16(x + (size_t)nx * HW * C + (size_t)i * VE), v);
#pragma unroll
        for (int k = 0; k < VE; ++k) v[k] *= gate[(size_t)n * C + cv * VE + k];
        stg16(y + (size_t)n * HW * C + (size_t)i * VE, pack16<T>(v));
    }
}

// grid (N): one workgroup per image: dpre = ds*g*(1-g); dgap = conv1d-transpose(dpre); per-image dw[j] = sum_c dpre*gm shifted
__global__ void __launch_bounds__(256) eca_bwd_small_kernel(const float* __restrict__ dot_part, int nparts,
                                                           const float* __restrict__ gate,
                                                           const float* __restrict__ gapmean, const float* const* w,
                                                           int k, float* __restrict__ dgap, float* __restrict__ dw_img,
                                                           int ipe, int C, int creal, float dgap_scale) {
    const int n = blockIdx.x, e = n / ipe;
    __shared__ float dpre[1024];
    __shared__ float wacc[256][9];
    const float* we = w[e];
    const int pad = k / 2;
    float dwl[9];
    for (int j = 0; j < 9; ++j) dwl[j] = 0.f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float s = 0.f;
        for (int i = 0; i < nparts; ++i) s += dot_part[((size_t)n * nparts + i) * C + c];
        const float g = gate[(size_t)n * C + c];
        dpre[c] = c < creal ? s * g * (1.f - g) : 0.f;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        float d = 0.f;
        for (int j = 0; j < k; ++j) {
            const int cc = c - j + pad;
            if (cc >= 0 && cc < creal) d += we[j] * dpre[cc];
        }
        if (dgap) dgap[(size_t)n * C + c] = c < creal ? d * dgap_scale : 0.f;
        if (c < creal)
            for (int j = 0; j < k; ++j) {
                const int cc = c + j - pad;
                if (cc >= 0 && cc < creal) dwl[j] += dpre[c] * gapmean[(size_t)n * C + cc];
            }
    }
    for (int j = 0; j < 9; ++j) wacc[threadIdx.x][j] = dwl[j];
    __syncthreads();
    if (threadIdx.x < k) {
        float s = 0.f;
        for (int t = 0; t < 256; ++t) s += wacc[t][threadIdx.x];
        dw_img[(size_t)n * k + threadIdx.x] = s;
    }
}

// grid (E): dw[e][j] = sum over the expert's images, fixed order
__global__ void __launch_bounds__(64) eca_bwd_dw_kernel(const float* __restrict__ dw_img, float* __restrict__ dw, int ipe,
                                                       int k) {
    const int e = blockIdx.x, j = threadIdx.x;
    if (j < k) {
        float s = 0.f;
        for (int n = e * ipe; n < (e + 1) * ipe; ++n) s += dw_img[(size_t)n * k + j];
        dw[e * k + j] = s;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) eca_bwd_apply_kernel(const T* __restrict__ dy, const float* __restrict__ gate,
                                                           const float* __restrict__ dgap, T* __restrict__ dx,
                                                           long long HW, int C, float inv_hw) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const int n = blockIdx.y;
    const long long nvec = HW * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        float v[VE];
        const size_t off = (size_t)n * HW * C + (size_t)i * VE;
        unpack16<T>(ldg16(dy + off), v);
#pragma unroll
        for (int k = 0; k < VE; ++k)
            v[k] = v[k] * gate[(size_t)n * C + cv * VE + k] + dgap[(size_t)n * C + cv * VE + k] * inv_hw;
        stg16(dx + off, pack16<T>(v));
    }
}

// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void __launch_bounds__(256) nchw_to_nhwc_kernel(const float* __restrict__ src, T* __restrict__ dst, int B,
                                                          int C, int H, int W, int Cp) {
    const long long total = (long long)B * H * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long hw = i % ((long long)H * W);
        const int b = (int)(i / ((long long)H * W));
        T* d = dst + (size_t)i * Cp;
        for (int c = 0; c < Cp; ++c) {
            const float v = c < C ? src[((size_t)b * C + c) * H * W + hw] : 0.f;
            d[c] = from_f32<T>(v);
        }
    }
}

// same, 64 pixels x Cp channels per workgroup through LDS: reads coalesced along the pixel axis of every channel plane,
// writes one contiguous 64*Cp-element run of the NHWC tensor (the per-pixel scalar stores above are strided by Cp)
template <typename T, int PIX>
__global__ void __launch_bounds__(256) nchw_to_nhwc_tiled_kernel(const float* __restrict__ src, T* __restrict__ dst, int B,
                                                                int C, long long HW, int Cp) {
    extern __shared__ float tile[];          // [PIX][Cp + 1]
    const long long nblk = (HW + PIX - 1) / PIX;
    const int b = (int)(blockIdx.x / nblk);
    const long long p0 = (blockIdx.x % nblk) * PIX;
    const int np = (int)((HW - p0) < PIX ? (HW - p0) : PIX);
    const int CS = Cp + 1;
    for (int i = threadIdx.x; i < PIX * Cp; i += 256) {
        const int c = i / PIX, p = i - c * PIX;          // consecutive threads: consecutive pixels of one channel plane
        if (p < np) tile[p * CS + c] = c < C ? src[((size_t)b * C + c) * HW + p0 + p] : 0.f;
    }
    __syncthreads();
    T* d = dst + ((size_t)b * HW + p0) * Cp;
    for (int i = threadIdx.x; i < np * Cp; i += 256) d[i] = from_f32<T>(tile[(i / Cp) * CS + (i % Cp)]);
}

template <typename T>
__global__ void __launch_bounds__(256) pad_rows_kernel(const float* __restrict__ src, T* __restrict__ dst, int B, int K,
                                                      int Kp) {
    const int total = B * Kp;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int k = i % Kp, b = i / Kp;
        dst[i] = from_f32<T>(k < K ? src[b * K + k] : 0.f);
    }
}

// ================================================================================================
// ------------------------------------------------------------------------------------------------
// Stand-alone activation (+ inverted dropout) of make_mlp's BatchNorm1d layout (basics.py:30-39: Linear -> BN1d -> act ->
// Dropout); without BatchNorm the activation lives in the GEMM epilogue.  Backward from the SAVED output.
template <typename T>
__global__ void __launch_bounds__(256) act_fwd_kernel(const T* __restrict__ x, T* __restrict__ y, long long nvec, int act,
                                                     float drop_p, unsigned long long seed) {
    constexpr int VE = 16 / (int)sizeof(T);
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float v[VE];
        unpack16<T>(ldg16(x + i * VE), v);
#pragma unroll
        for (int k = 0; k < VE; ++k) {
            v[k] = act_apply(act, v[k]);
            if (drop_p > 0.f) v[k] = hash_uniform(seed, (unsigned long long)i * VE + k) >= drop_p ? v[k] * keep_scale : 0.f;
        }
        stg16(y + i * VE, pack16<T>(v));
    }
}

template <typename T>
__global__ void __launch_bounds__(256) act_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ y, T* __restrict__ dx,
                                                     long long nvec, int act, float drop_p) {
    constexpr int VE = 16 / (int)sizeof(T);
    const float keep_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;
    const int mode = act == PMOE_ACT_ELU ? PMOE_RES_DELU : act == PMOE_ACT_TANH ? PMOE_RES_DTANH : PMOE_RES_DSIGMOID;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (long long)gridDim.x * 256) {
        float g[VE], yv[VE];
        unpack16<T>(ldg16(dy + i * VE), g);
        unpack16<T>(ldg16(y + i * VE), yv);
#pragma unroll
        for (int k = 0; k < VE; ++k) {
            if (act == PMOE_ACT_NONE) g[k] = (drop_p > 0.f && yv[k] == 0.f) ? 0.f : g[k] * keep_scale;
            else if (act == PMOE_ACT_RELU) g[k] = yv[k] > 0.f ? g[k] * keep_scale : 0.f;
            else g[k] = (drop_p > 0.f && yv[k] == 0.f) ? 0.f : g[k] * act_deriv_from_output(mode, yv[k] * (1.f / keep_scale)) * keep_scale;
        }
        stg16(dx + i * VE, pack16<T>(g));
    }
}

static inline int grid_for(long long nvec, int cap = 4096) {
    long long g = (nvec + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

extern "C" {

int pmoe_colstats(const void* x, int64_t rows_per_expert, int32_t E, int32_t C, int32_t ld, int32_t coff, float* part,
                  int32_t nparts, float* shiftc, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, return colstats_launch<T>(x, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, rows_per_expert, E, C, ld, coff,
                                                 0, part, nparts, 0, shiftc, (hipStream_t)stream));
}

int pmoe_bn_bwd_reduce(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                       const float* scale, const float* shift, int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, float* part, int32_t nparts,
                       void* gmask_out, int32_t dtype, void* stream) {
    if (relu && !y && (!scale || !shift)) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, return colstats_launch<T>(x, dy, y, mean, invstd, scale, shift, rows_per_expert, E, C, C, 0, relu, part,
                                                 nparts, 1, nullptr, (hipStream_t)stream, gmask_out));
}

int pmoe_reduce_partials(const float* part_in, float* part_out, int32_t E, int32_t nin, int32_t nout, int32_t width,
                         void* stream) {
    if (nin < 1 || nout < 1 || width < 1) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(reduce_partials_kernel, dim3(nout, E), dim3(256), 0, (hipStream_t)stream, part_in, part_out, nin,
                       nout, width);
    return (int)hipGetLastError();
}

int pmoe_bn_finalize(const float* part, int32_t nparts, int64_t count, const void* const* gamma_ptrs,
                     const void* const* beta_ptrs, void* const* rmean_ptrs, void* const* rvar_ptrs, float momentum,
                     float eps, int32_t training, float* scale, float* shift, float* mean, float* invstd, int32_t E,
                     int32_t C, const float* shiftc, void* stream) {
    if (!training && (!rmean_ptrs || !rvar_ptrs)) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH, E), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, part, nparts,
                       (long long)count, (const float* const*)gamma_ptrs, (const float* const*)beta_ptrs,
                       (float* const*)rmean_ptrs, (float* const*)rvar_ptrs, momentum, eps, training, scale, shift, mean,
                       invstd, C, shiftc);
    return (int)hipGetLastError();
}

int pmoe_bn_bwd_finalize(const float* part, int32_t nparts, int64_t count, float* dgamma, float* dbeta, float* c1,
                         float* c2, int32_t E, int32_t C, void* stream) {
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + FIN_CH - 1) / FIN_CH, E), dim3(FIN_CH * FIN_LANES), 0, (hipStream_t)stream, part, nparts,
                       (long long)count, dgamma, dbeta, c1, c2, C);
    return (int)hipGetLastError();
}

int pmoe_bn_apply(const void* x, const void* res, void* y, const float* scale, const float* shift, const float* mean,
                  int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, int32_t y_ld, int32_t y_coff, int32_t dtype,
                  void* y_fp8, float in_scale, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || !pow2(C / VE) || C / VE > 256) return PMOE_ERR_ARG;   // kernels keep one channel vector per thread
        if (y_ld <= 0) { y_ld = C; y_coff = 0; }
        if (y_fp8 && (dtype != PMOE_DT_BF16 || y_ld != C)) return PMOE_ERR_ARG;
        if (y_ld % VE || y_coff % VE || y_coff + C > y_ld) return PMOE_ERR_ARG;
        int log_cv = 0;
        while ((1 << log_cv) < C / VE) ++log_cv;
        const long long nvec = rows_per_expert * (C / VE);
        hipLaunchKernelGGL((bn_apply_kernel<T>), dim3(grid_for(nvec, 2048), E), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (const T*)res, (T*)y, scale, shift, mean, (long long)rows_per_expert, C, relu,
                           y_ld, y_coff, log_cv, (unsigned char*)y_fp8, in_scale);
        return (int)hipGetLastError();
    });
}

int pmoe_bn_apply_pool2(const void* x, void* y, void* pooled, const float* scale, const float* shift, const float* mean,
                        int32_t ipe, int32_t H, int32_t W, int32_t E, int32_t C, int32_t relu, int32_t y_ld, int32_t y_coff,
                        int32_t dtype, void* stream) {
    if (!x || !y || !pooled || ipe < 1 || E < 1 || H < 2 || W < 2 || (H & 1) || (W & 1)) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || !pow2(C / VE) || C / VE > 256) return PMOE_ERR_ARG;
        if (y_ld <= 0) { y_ld = C; y_coff = 0; }
        if (y_ld % VE || y_coff % VE || y_coff + C > y_ld) return PMOE_ERR_ARG;
        const long long nq = (long long)ipe * (H / 2) * (W / 2) * (C / VE);
        hipLaunchKernelGGL((bn_apply_pool2_kernel<T>), dim3(grid_for(nq, 2048), E), dim3(256), 0, (hipStream_t)stream, (const T*)x,
                           (T*)y, (T*)pooled, scale, shift, mean, ipe, H, W, C, relu, y_ld, y_coff);
        return (int)hipGetLastError();
    });
}

int pmoe_bn_bwd_apply(const void* dy, const void* y, const void* x, const float* mean, const float* invstd,
                      const float* scale, const float* shift, const float* c1, const float* c2, void* dx, void* gmask_out,
                      int64_t rows_per_expert, int32_t E, int32_t C, int32_t relu, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || !pow2(C / VE) || C / VE > 256) return PMOE_ERR_ARG;   // kernels keep one channel vector per thread
        const long long nvec = rows_per_expert * (C / VE);
        // every thread first loads 6 per-channel constant vectors: keep ~1024 workgroups in total (4 per CU) so that this
        // prologue is amortised over many rows (8192 workgroups cost a flat 170 us on the small layer3/4 tensors)
        int cap = 1024 / (E > 0 ? E : 1);
        cap = cap < 64 ? 64 : cap;
        hipLaunchKernelGGL((bn_bwd_apply_kernel<T>), dim3(grid_for(nvec, cap), E), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dy, (const T*)y, (const T*)x, mean, invstd, scale, shift, c1, c2, (T*)dx, (T*)gmask_out,
                           (long long)rows_per_expert, C, relu);
        return (int)hipGetLastError();
    });
}

int pmoe_act_fwd(const void* x, void* y, int64_t n, int32_t act, float drop_p, uint64_t seed, int32_t dtype, void* stream) {
    if (!x || !y || n <= 0 || act < PMOE_ACT_NONE || act > PMOE_ACT_SIGMOID || drop_p < 0.f || drop_p >= 1.f) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (n % VE) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((act_fwd_kernel<T>), dim3(grid_for(n / VE)), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y,
                           (long long)(n / VE), act, drop_p, (unsigned long long)seed);
        return (int)hipGetLastError();
    });
}

int pmoe_act_bwd(const void* dy, const void* y, void* dx, int64_t n, int32_t act, float drop_p, int32_t dtype, void* stream) {
    if (!dy || !y || !dx || n <= 0 || act < PMOE_ACT_NONE || act > PMOE_ACT_SIGMOID || drop_p < 0.f || drop_p >= 1.f) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (n % VE) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((act_bwd_kernel<T>), dim3(grid_for(n / VE)), dim3(256), 0, (hipStream_t)stream, (const T*)dy,
                           (const T*)y, (T*)dx, (long long)(n / VE), act, drop_p);
        return (int)hipGetLastError();
    });
}

int pmoe_maxpool3s2_fwd(const void* x, void* y, uint8_t* argmax, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE) return PMOE_ERR_ARG;
        const long long nvec = (long long)N * Ho * Wo * (C / VE);
        hipLaunchKernelGGL((maxpool_fwd_kernel<T>), dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)x, (T*)y, argmax, N, H, W, C, Ho, Wo);
        return (int)hipGetLastError();
    });
}

int pmoe_maxpool3s2_bwd(const void* dy, const uint8_t* argmax, void* dx, int32_t N, int32_t H, int32_t W, int32_t C,
                        int32_t dtype, void* stream) {
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE) return PMOE_ERR_ARG;
        const long long nvec = (long long)N * H * W * (C / VE);
        hipLaunchKernelGGL((maxpool_bwd_kernel<T>), dim3(grid_for(nvec, 8192)), dim3(256), 0, (hipStream_t)stream,
                           (const T*)dy, argmax, (T*)dx, N, H, W, C, Ho, Wo);
        return (int)hipGetLastError();
    });
}

int pmoe_gap_partial(const void* a, const void* b, float* part, int32_t N, int64_t HW, int32_t C, int32_t nparts,
                     int32_t b_shared_ipe, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || C / VE > 256 || nparts < 1) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((gap_partial_kernel<T>), dim3(nparts, N), dim3(256), 0, (hipStream_t)stream, (const T*)a,
                           (const T*)b, part, (long long)HW, C, nparts, b_shared_ipe);
        return (int)hipGetLastError();
    });
}

int pmoe_bn_apply_gap(const void* x, void* y, const float* scale, const float* shift, const float* mean, float* gap_part,
                      int32_t nparts, int32_t N, int32_t ipe, int64_t HW, int32_t C, int32_t relu, int32_t dtype, void* stream) {
    if (!x || !y || !scale || !shift || !mean || !gap_part || nparts < 1 || N < 1 || ipe < 1 || N % ipe || HW < 1) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || C / VE > 256) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((bn_apply_gap_kernel<T>), dim3(nparts, N), dim3(256), 0, (hipStream_t)stream, (const T*)x, (T*)y, scale,
                           shift, mean, gap_part, (long long)HW, C, nparts, ipe, relu);
        return (int)hipGetLastError();
    });
}

int pmoe_gap_finish(const float* part, void* out, int32_t N, int32_t C, int32_t nparts, int64_t HW, int32_t out_ld,
                    int32_t out_coff, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((gap_finish_kernel<T>), dim3(grid_for((long long)N * C, 1024)), dim3(256), 0,
                           (hipStream_t)stream, part, (T*)out, N, C, nparts, 1.f / (float)HW, out_ld, out_coff);
        return (int)hipGetLastError();
    });
}

int pmoe_gap_bwd(const void* g, void* dx, int32_t N, int64_t HW, int32_t C, int32_t g_ld, int32_t g_coff, int32_t dtype,
                 void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || g_ld % VE || g_coff % VE) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((gap_bwd_kernel<T>), dim3(grid_for(HW * (C / VE), 256), N), dim3(256), 0,
                           (hipStream_t)stream, (const T*)g, (T*)dx, (long long)HW, C, g_ld, g_coff, 1.f / (float)HW);
        return (int)hipGetLastError();
    });
}

int pmoe_eca_gate(const float* gap_part, int32_t nparts, int64_t HW, const void* const* w_ptrs, int32_t k, float* gate,
                  float* gapmean, int32_t N, int32_t ipe, int32_t in_ipe, int32_t C, int32_t creal, void* stream) {
    if (C > 1024 || k > 9 || k < 1 || !(k & 1)) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(eca_gate_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, gap_part, nparts, 1.f / (float)HW,
                       (const float* const*)w_ptrs, k, gate, gapmean, ipe, in_ipe, C, creal);
    return (int)hipGetLastError();
}

int pmoe_eca_scale(const void* x, const float* gate, void* y, int32_t N, int64_t HW, int32_t C, int32_t x_shared_ipe,
                   int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((eca_scale_kernel<T>), dim3(grid_for(HW * (C / VE), 512), N), dim3(256), 0,
                           (hipStream_t)stream, (const T*)x, gate, (T*)y, (long long)HW, C, x_shared_ipe);
        return (int)hipGetLastError();
    });
}

int pmoe_eca_bwd_small(const float* dot_part, int32_t nparts, const float* gate, const float* gapmean,
                       const void* const* w_ptrs, int32_t k, float* dgap, float* dw, float* dw_scratch, int32_t N,
                       int32_t ipe, int32_t C, int32_t creal, float dgap_scale, void* stream) {
    if (C > 1024 || k > 9 || k < 1 || N % ipe || !dw_scratch) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(eca_bwd_small_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, dot_part, nparts, gate,
                       gapmean, (const float* const*)w_ptrs, k, dgap, dw_scratch, ipe, C, creal, dgap_scale);
    hipLaunchKernelGGL(eca_bwd_dw_kernel, dim3(N / ipe), dim3(64), 0, (hipStream_t)stream, dw_scratch, dw, ipe, k);
    return (int)hipGetLastError();
}

int pmoe_eca_bwd_apply(const void* dy, const float* gate, const float* dgap, void* dx, int32_t N, int64_t HW, int32_t C,
                       int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((eca_bwd_apply_kernel<T>), dim3(grid_for(HW * (C / VE), 512), N), dim3(256), 0,
                           (hipStream_t)stream, (const T*)dy, gate, dgap, (T*)dx, (long long)HW, C, 1.f / (float)HW);
        return (int)hipGetLastError();
    });
}

int pmoe_nchw_to_nhwc(const float* src, void* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cp,
                      int32_t dtype, void* stream) {
    if (B < 1 || C < 1 || H < 1 || W < 1 || Cp < C) return PMOE_ERR_ARG;
    const long long HW = (long long)H * W, nblk = (HW + 63) / 64 * B, nblk4 = (HW + 255) / 256 * B;
    DISPATCH_DT(dtype, {
        if (Cp <= 32 && nblk4 <= 0x7fffffffLL)           // few channels (camera frames): 256 pixels per workgroup
            hipLaunchKernelGGL((nchw_to_nhwc_tiled_kernel<T, 256>), dim3((unsigned)nblk4), dim3(256),
                               256 * (Cp + 1) * sizeof(float), (hipStream_t)stream, src, (T*)dst, B, C, HW, Cp);
        else if (Cp <= 240 && nblk <= 0x7fffffffLL)
            hipLaunchKernelGGL((nchw_to_nhwc_tiled_kernel<T, 64>), dim3((unsigned)nblk), dim3(256),
                               64 * (Cp + 1) * sizeof(float), (hipStream_t)stream, src, (T*)dst, B, C, HW, Cp);
        else
            hipLaunchKernelGGL((nchw_to_nhwc_kernel<T>), dim3(grid_for((long long)B * H * W, 8192)), dim3(256), 0,
                               (hipStream_t)stream, src, (T*)dst, B, C, H, W, Cp);
        return (int)hipGetLastError();
    });
}

int pmoe_pad_rows(const float* src, void* dst, int32_t B, int32_t K, int32_t Kp, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((pad_rows_kernel<T>), dim3(grid_for((long long)B * Kp, 64)), dim3(256), 0,
                           (hipStream_t)stream, src, (T*)dst, B, K, Kp);
        return (int)hipGetLastError();
    });
}

}  // extern "C"
