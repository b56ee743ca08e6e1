// Weight (re)packing between the reference's per-expert f32 parameter tensors and the grouped
// kernel layouts, the gate-softmax / Gaussian-mixture head and the fused moe_loss.
#include "common.h"

// ------------------------------------------------------------------------------------------------
// src[e]: f32 [cout][cin][taps].  fwd: T [E][coutp][taps][cinp].  dgrd: T [E][cinp2][taps][coutp2]
// with dgrd[e][ci][taps-1-t][co] = src[e][co][ci][t]  (the flipped, transposed operand of dgrad).
template <typename T>
__global__ void __launch_bounds__(256) pack_w_kernel(const float* const* __restrict__ src, T* __restrict__ fwd,
                                                    T* __restrict__ dgrd, int cout, int cin, int taps, int coutp,
                                                    int cinp, int cinp2, int coutp2) {
    // (32-bit index arithmetic: one expert's pack is < 2^31 elements, the launcher checks -- the 64-bit divisions of the first
    //  version were most of the kernel's 0.5 ms per optimizer step)
    const int e = blockIdx.y;
    const float* s = src[e];
    const unsigned nf = fwd ? (unsigned)(coutp * taps * cinp) : 0u;
    const unsigned nd = dgrd ? (unsigned)(cinp2 * taps * coutp2) : 0u;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < nf + nd; i += gridDim.x * 256u) {
        if (i < nf) {
            const unsigned ci = i % (unsigned)cinp;
            const unsigned t = i / (unsigned)cinp;
            const unsigned tp = t % (unsigned)taps;
            const unsigned co = t / (unsigned)taps;
            const float v = ((int)co < cout && (int)ci < cin) ? s[((size_t)co * cin + ci) * taps + tp] : 0.f;
            fwd[(size_t)e * nf + i] = from_f32<T>(v);
        } else {
            const unsigned k = i - nf;
            const unsigned co = k % (unsigned)coutp2;
            const unsigned t = k / (unsigned)coutp2;
            const unsigned tp = t % (unsigned)taps;
            const unsigned ci = t / (unsigned)taps;
            const float v = ((int)co < cout && (int)ci < cin) ? s[((size_t)co * cin + ci) * taps + (taps - 1 - tp)] : 0.f;
            dgrd[(size_t)e * nd + k] = from_f32<T>(v);
        }
    }
}

// Per-IMAGE weights with the ECA gate folded in (basics.py:69-76 feeding basics.py:113): conv(x * g[n,c], W) ==
// conv(x, W * g[n,c]), so the gated activation is never materialised -- the conv runs with "one expert per image"
// (ipe = 1) on these packs.  fwd[n][co][tap][ci] = W[e][co][ci][tap] * g[n][ci]; dgrd[n][ci][tap'][co] likewise (flipped).
template <typename T>
__global__ void __launch_bounds__(256) pack_w_gated_kernel(const float* const* __restrict__ src,
                                                          const float* __restrict__ gate, int gate_ld, T* __restrict__ fwd,
                                                          T* __restrict__ dgrd, int ipe, int cout, int cin, int taps,
                                                          int coutp, int cinp, int cinp2, int coutp2) {
    const int n = blockIdx.y;
    const float* s = src[n / ipe];
    const float* g = gate + (size_t)n * gate_ld;
    const long long nf = fwd ? (long long)coutp * taps * cinp : 0;
    const long long nd = dgrd ? (long long)cinp2 * taps * coutp2 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long long)gridDim.x * 256) {
        if (i < nf) {
            const int ci = (int)(i % cinp);
            long long t = i / cinp;
            const int tp = (int)(t % taps);
            const int co = (int)(t / taps);
            const float v = (co < cout && ci < cin) ? s[((size_t)co * cin + ci) * taps + tp] * g[ci] : 0.f;
            fwd[(size_t)n * nf + i] = from_f32<T>(v);
        } else {
            const long long k = i - nf;
            const int co = (int)(k % coutp2);
            long long t = k / coutp2;
            const int tp = (int)(t % taps);
            const int ci = (int)(t / taps);
            const float v = (co < cout && ci < cin) ? s[((size_t)co * cin + ci) * taps + (taps - 1 - tp)] * g[ci] : 0.f;
            dgrd[(size_t)n * nd + k] = from_f32<T>(v);
        }
    }
}

// Inference: eval-mode BatchNorm folded into the preceding conv:
//   bn(conv(x, W)) = conv(x, W * s[co]) + (beta - mean * s),   s = gamma / sqrt(running_var + eps)
// fwd[e][co][tap][ci] = W[e][co][ci][tap] * scale[e][co];  bias[e][co] = shift[e][co] - mean[e][co] * scale[e][co]
template <typename T>
__global__ void __launch_bounds__(256) pack_w_scaled_kernel(const float* const* __restrict__ src,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           const float* __restrict__ mean, T* __restrict__ fwd,
                                                           float* __restrict__ bias, int cout, int cin, int taps, int coutp,
                                                           int cinp) {
    const int e = blockIdx.y;
    const float* s = src[e];
    const long long nf = (long long)coutp * taps * cinp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nf; i += (long long)gridDim.x * 256) {
        const int ci = (int)(i % cinp);
        long long t = i / cinp;
        const int tp = (int)(t % taps);
        const int co = (int)(t / taps);
        const float v = (co < cout && ci < cin) ? s[((size_t)co * cin + ci) * taps + tp] * scale[e * cout + co] : 0.f;
        fwd[(size_t)e * nf + i] = from_f32<T>(v);
    }
    for (int co = blockIdx.x * 256 + threadIdx.x; co < coutp; co += gridDim.x * 256)
        bias[e * coutp + co] = co < cout ? shift[e * cout + co] - mean[e * cout + co] * scale[e * cout + co] : 0.f;
}

// ---- BASELINE config 5: e4m3 weights with one power-of-two scale per output channel --------------------------------
// s[e][co] = 2^ceil(log2(amax_co / 448)) (1 if the row is all zero), q = e4m3(W / s) (round to nearest even, clamped to the
// finite range): power-of-two scales make W / s and q * s exact, so the data-gradient operand (bf16) holds exactly the
// dequantised weights the forward used.  wscale[e][co] = s; oscale[e][co] = s / in_scale (the conv epilogue's factor).
__global__ void __launch_bounds__(256) fp8_row_scale_kernel(const float* const* __restrict__ src, float* __restrict__ wscale,
                                                           float* __restrict__ oscale, float in_scale, int cout, int coutp,
                                                           int row) {
    const int co = blockIdx.x, e = blockIdx.y;
    __shared__ float red[256];
    float m = 0.f;
    if (co < cout) {
        const float* s = src[e] + (size_t)co * row;
        for (int i = threadIdx.x; i < row; i += 256) m = fmaxf(m, fabsf(s[i]));
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] = fmaxf(red[threadIdx.x], red[threadIdx.x + o]);
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float sc = 1.f;
        if (red[0] > 0.f) {
            int ex;
            const float mant = frexpf(red[0] / PMOE_FP8_MAX, &ex);      // amax / 448 = mant * 2^ex, mant in [0.5, 1)
            if (mant == 0.5f) ex -= 1;                                   // exact power of two: ceil(log2) = ex - 1
            sc = ldexpf(1.f, ex);
        }
        wscale[e * coutp + co] = sc;
        oscale[e * coutp + co] = sc / in_scale;
    }
}

__device__ __forceinline__ unsigned char to_e4m3(float v) {
    v = fminf(fmaxf(v, -PMOE_FP8_MAX), PMOE_FP8_MAX);
    return (unsigned char)(__builtin_amdgcn_cvt_pk_fp8_f32(v, 0.f, 0, false) & 0xff);
}

// decode one OCP e4m3 byte (bias 7, no infinities; S.1111.111 = NaN does not occur: inputs are clamped)
__device__ __forceinline__ float from_e4m3(unsigned char b) {
    const int ex = (b >> 3) & 15, mt = b & 7;
    const float mag = ex ? ldexpf((float)(8 + mt), ex - 10) : ldexpf((float)mt, -9);
    return (b & 0x80) ? -mag : mag;
}

__global__ void __launch_bounds__(256) pack_w_fp8_kernel(const float* const* __restrict__ src, const float* __restrict__ wscale,
                                                        unsigned char* __restrict__ fwd, bf16* __restrict__ dgrd, int cout, int cin,
                                                        int taps, int coutp, int cinp, int cinp2, int coutp2) {
    const int e = blockIdx.y;
    const float* s = src[e];
    const long long nf = fwd ? (long long)coutp * taps * cinp : 0;
    const long long nd = dgrd ? (long long)cinp2 * taps * coutp2 : 0;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < nf + nd; i += (long long)gridDim.x * 256) {
        if (i < nf) {
            const int ci = (int)(i % cinp);
            long long t = i / cinp;
            const int tp = (int)(t % taps);
            const int co = (int)(t / taps);
            const float v = (co < cout && ci < cin) ? s[((size_t)co * cin + ci) * taps + tp] / wscale[e * coutp + co] : 0.f;
            fwd[(size_t)e * nf + i] = to_e4m3(v);
        } else {
            const long long k = i - nf;
            const int co = (int)(k % coutp2);
            long long t = k / coutp2;
            const int tp = (int)(t % taps);
            const int ci = (int)(t / taps);
            float v = 0.f;
            if (co < cout && ci < cin) {
                const float sc = wscale[e * coutp + co];
                v = from_e4m3(to_e4m3(s[((size_t)co * cin + ci) * taps + (taps - 1 - tp)] / sc)) * sc;
            }
            dgrd[(size_t)e * nd + k] = (bf16)v;
        }
    }
}

__global__ void __launch_bounds__(256) unpack_wgrad_kernel(const float* __restrict__ ws, float* __restrict__ g, int cout,
                                                          int cin, int taps, int coutp, int cinp) {
    const int e = blockIdx.y;
    const long long n = (long long)cout * cin * taps;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const int tp = (int)(i % taps);
        long long t = i / taps;
        const int ci = (int)(t % cin);
        const int co = (int)(t / cin);
        g[(size_t)e * n + i] = ws[(((size_t)e * taps + tp) * coutp + co) * cinp + ci];
    }
}

__global__ void __launch_bounds__(256) pack_bias_kernel(const float* const* __restrict__ src, float* __restrict__ dst,
                                                       int cout, int coutp) {
    const int e = blockIdx.y;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < coutp; i += gridDim.x * 256)
        dst[e * coutp + i] = i < cout ? src[e][i] : 0.f;
}

// ------------------------------------------------------------------------------------------------
// Gate softmax + mixture parameters.  G = pow2 >= E lanes per sample; a wave covers 64/G samples and
// reduces over the expert axis with xor-shuffles (no LDS, no atomics).
// Two head layouts:
//   shared = 0 (MixtureOfExperts, moe.py:131-158): head [E*B][ld], row (e*B + b) = {mean0, mean1, rawstd0, rawstd1, alpha};
//                                                   spd [E*B][ld] col 0 -> speeds [B][E][1]
//   shared = 1 (MixtureOfExpertsShared, moe.py:180-233): head [B][ld], cols 4e..4e+3 = expert e's mean/rawstd,
//                                                   col 4E+e = its alpha;  spd [B][ld] col 0 -> speeds [B][1]
struct GateIdx {
    int shared, B, E, ld;
    __device__ __forceinline__ size_t comp(int b, int e, int i) const {          // mean / rawstd element i of (b, e)
        return shared ? (size_t)b * ld + 4 * e + i : ((size_t)e * B + b) * ld + i;
    }
    __device__ __forceinline__ size_t alpha(int b, int e) const {
        return shared ? (size_t)b * ld + 4 * E + e : ((size_t)e * B + b) * ld + 4;
    }
};

template <typename T>
__global__ void __launch_bounds__(256) gate_fwd_kernel(const T* __restrict__ head, int head_ld, const T* __restrict__ spd,
                                                      int spd_ld, float* __restrict__ probs, float* __restrict__ mean,
                                                      float* __restrict__ sd, float* __restrict__ speeds, int B, int E,
                                                      int G, int alpha_relu, int shared) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int b = gid / G, e = gid % G;
    const bool live = b < B && e < E;
    const GateIdx ix{shared, B, E, head_ld};
    float hv[5] = {0, 0, 0, 0, 0};
    float sp = 0.f;
    if (live) {
#pragma unroll
        for (int i = 0; i < 4; ++i) hv[i] = to_f32(head[ix.comp(b, e, i)]);
        hv[4] = to_f32(head[ix.alpha(b, e)]);
        sp = to_f32(spd[(shared ? (size_t)b : (size_t)e * B + b) * spd_ld]);
    }
    const bool raw = alpha_relu & 2;        // lone expert (moe.py:74-101): alpha itself, no softmax over the group
    float a = (alpha_relu & 1) ? fmaxf(hv[4], 0.f) : hv[4];
    if (!live) a = -INFINITY;
    float m = a;
    for (int off = 1; off < G; off <<= 1) m = fmaxf(m, __shfl_xor(m, off));
    const float ex = live ? expf(a - m) : 0.f;
    float s = ex;
    for (int off = 1; off < G; off <<= 1) s += __shfl_xor(s, off);
    if (live) {
        const size_t o = (size_t)b * E + e;
        probs[o] = raw ? a : ex / s;
        mean[o * 2 + 0] = hv[0];
        mean[o * 2 + 1] = hv[1];
        sd[o * 2 + 0] = (hv[2] > 0.f ? hv[2] : expm1f(hv[2])) + 1.f;
        sd[o * 2 + 1] = (hv[3] > 0.f ? hv[3] : expm1f(hv[3])) + 1.f;
        if (!shared) speeds[o] = sp;
        else if (e == 0) speeds[b] = sp;
    }
}

template <typename T>
__global__ void __launch_bounds__(256) gate_bwd_kernel(const T* __restrict__ head, int head_ld,
                                                      const float* __restrict__ probs, const float* __restrict__ dprobs,
                                                      const float* __restrict__ dmean, const float* __restrict__ dstd,
                                                      const float* __restrict__ dspeeds, T* __restrict__ dhead,
                                                      T* __restrict__ dspd, int spd_ld, int B, int E, int G,
                                                      int alpha_relu, int shared) {
    const int gid = blockIdx.x * 256 + threadIdx.x;
    const int b = gid / G, e = gid % G;
    const bool live = b < B && e < E;
    const GateIdx ix{shared, B, E, head_ld};
    float p = 0.f, dp = 0.f;
    if (live) {
        p = probs[(size_t)b * E + e];
        dp = dprobs ? dprobs[(size_t)b * E + e] : 0.f;
    }
    float dot = p * dp;
    for (int off = 1; off < G; off <<= 1) dot += __shfl_xor(dot, off);
    if (live) {
        const size_t o = (size_t)b * E + e;
        const float r2 = to_f32(head[ix.comp(b, e, 2)]), r3 = to_f32(head[ix.comp(b, e, 3)]);
        const float r4 = to_f32(head[ix.alpha(b, e)]);
        float da = (alpha_relu & 2) ? dp : p * (dp - dot);
        if ((alpha_relu & 1) && !(r4 > 0.f)) da = 0.f;
        dhead[ix.comp(b, e, 0)] = from_f32<T>(dmean ? dmean[o * 2 + 0] : 0.f);
        dhead[ix.comp(b, e, 1)] = from_f32<T>(dmean ? dmean[o * 2 + 1] : 0.f);
        dhead[ix.comp(b, e, 2)] = from_f32<T>(dstd ? dstd[o * 2 + 0] * (r2 > 0.f ? 1.f : expf(r2)) : 0.f);
        dhead[ix.comp(b, e, 3)] = from_f32<T>(dstd ? dstd[o * 2 + 1] * (r3 > 0.f ? 1.f : expf(r3)) : 0.f);
        dhead[ix.alpha(b, e)] = from_f32<T>(da);
        if (!shared) {
            T* drow = dhead + ((size_t)e * B + b) * head_ld;
            for (int i = 5; i < head_ld; ++i) drow[i] = from_f32<T>(0.f);
            T* srow = dspd + ((size_t)e * B + b) * spd_ld;
            srow[0] = from_f32<T>(dspeeds ? dspeeds[o] : 0.f);
            for (int i = 1; i < spd_ld; ++i) srow[i] = from_f32<T>(0.f);
        } else if (e == 0) {
            T* drow = dhead + (size_t)b * head_ld;
            for (int i = 5 * E; i < head_ld; ++i) drow[i] = from_f32<T>(0.f);
            T* srow = dspd + (size_t)b * spd_ld;
            srow[0] = from_f32<T>(dspeeds ? dspeeds[b] : 0.f);
            for (int i = 1; i < spd_ld; ++i) srow[i] = from_f32<T>(0.f);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// moe_loss (trainer/loss.py:121-132) with MixtureSameFamily.log_prob written out:
//   ll_b = logsumexp_e( log_softmax(log clamp(p/sum p))_e + sum_d logN(a_d; mu_ed, sigma_ed) )
//   loss = c0 * (-mean_b ll_b) + c1 * mean_{b,e}((speed_be - target_b)^2) / E
// One workgroup; lanes over (sample, expert) as in the gate kernel; gradients are produced in the
// same pass (they are tiny) so backward is a scale by the incoming grad.
__global__ void __launch_bounds__(256) moe_loss_kernel(const float* __restrict__ probs, const float* __restrict__ mean,
                                                      const float* __restrict__ sd, const float* __restrict__ speeds,
                                                      const float* __restrict__ act, const float* __restrict__ tgt,
                                                      float c0, float c1, float* __restrict__ loss,
                                                      float* __restrict__ loglik, float* __restrict__ dprobs,
                                                      float* __restrict__ dmean, float* __restrict__ dstd,
                                                      float* __restrict__ dspeeds, int B, int E, int G,
                                                      int shared_speed) {
    const float EPS = 1.1920929e-07f, HALF_LOG_2PI = 0.9189385332046727f;
    const int e = threadIdx.x % G;
    float nll_acc = 0.f, mse_acc = 0.f;
    const int spb = 256 / G;
    for (int b0 = 0; b0 < B; b0 += spb) {
        const int b = b0 + threadIdx.x / G;
        const bool live = b < B && e < E;
        float p = 0.f, comp = 0.f, mu[2] = {0, 0}, sg[2] = {1, 1}, a[2] = {0, 0}, spv = 0.f, tg = 0.f;
        if (live) {
            const size_t o = (size_t)b * E + e;
            p = probs[o];
            for (int d = 0; d < 2; ++d) {
                mu[d] = mean[o * 2 + d];
                sg[d] = sd[o * 2 + d];
                a[d] = act[b * 2 + d];
                const float z = (a[d] - mu[d]) / sg[d];
                comp += -0.5f * z * z - logf(sg[d]) - HALF_LOG_2PI;
            }
            spv = shared_speed ? speeds[b] : speeds[o];
            tg = tgt[b];
        }
        float ps = p;
        for (int off = 1; off < G; off <<= 1) ps += __shfl_xor(ps, off);
        const float pn = live ? p / ps : 0.f;
        const bool clamped = pn < EPS || pn > 1.f - EPS;
        const float l = live ? logf(fminf(fmaxf(pn, EPS), 1.f - EPS)) : -INFINITY;
        float lm = l;
        for (int off = 1; off < G; off <<= 1) lm = fmaxf(lm, __shfl_xor(lm, off));
        float ls = live ? expf(l - lm) : 0.f;
        for (int off = 1; off < G; off <<= 1) ls += __shfl_xor(ls, off);
        const float lsm = l - (lm + logf(ls));                 // log_softmax of the logits
        const float t = live ? comp + lsm : -INFINITY;
        float tm = t;
        for (int off = 1; off < G; off <<= 1) tm = fmaxf(tm, __shfl_xor(tm, off));
        float ts = live ? expf(t - tm) : 0.f;
        for (int off = 1; off < G; off <<= 1) ts += __shfl_xor(ts, off);
        const float ll = tm + logf(ts);
        if (live) {
            const size_t o = (size_t)b * E + e;
            const float r = expf(t - ll);                       // responsibility of expert e
            const float q = expf(lsm);
            const float gs = -c0 / (float)B;                      // d loss / d ll_b
            if (e == 0) { nll_acc += -ll; if (loglik) loglik[b] = ll; }
            // d ll / d p_j = (r_j - q_j)/p_j  (normalisation terms cancel because sum r = sum q = 1)
            dprobs[o] = clamped ? 0.f : gs * (r - q) / (pn * ps);
            for (int d = 0; d < 2; ++d) {
                const float diff = a[d] - mu[d], iv = 1.f / (sg[d] * sg[d]);
                dmean[o * 2 + d] = gs * r * diff * iv;
                dstd[o * 2 + d] = gs * r * (diff * diff * iv / sg[d] - 1.f / sg[d]);
            }
            const float ds = spv - tg;
            if (!shared_speed) {                      // mse over [B,E,1] then / E  (loss.py:126-128)
                mse_acc += ds * ds;
                dspeeds[o] = c1 * 2.f * ds / ((float)B * (float)E * (float)E);
            } else if (e == 0) {                      // mse over [B,1]             (loss.py:129-130)
                mse_acc += ds * ds;
                dspeeds[b] = c1 * 2.f * ds / (float)B;
            }
        }
    }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = nll_acc;
    r2[threadIdx.x] = mse_acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) { r1[threadIdx.x] += r1[threadIdx.x + s]; r2[threadIdx.x] += r2[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        loss[0] = c0 * r1[0] / (float)B + c1 * r2[0] / (shared_speed ? (float)B : (float)B * (float)E * (float)E);
}

// ------------------------------------------------------------------------------------------------
// Stem input stage: conv1 reads x0 * gate[n][c] (ECA, basics.py:69-76).  From the per-IMAGE filter gradients
// G[n][t][k][c] = sum_p dz1[n][p][k] * x0[n % B][p + t][c] (conv_wgrad with per_image=1 on the UNscaled frames):
//   dW[e][k][c][t] = sum_{n in e} gate[n][c] * G[n][t][k][c]          (gradient of conv1.weight)
//   ds [n][c]      = sum_{k,t} W[e][k][c][t] * G[n][t][k][c]          (gradient wrt the ECA gate)
// which replaces the whole data-gradient convolution of conv1 (2.2 GB read + 0.5 GB written at B=64).
__global__ void __launch_bounds__(256) eca_stem_fold_dw_kernel(const float* __restrict__ G, const float* __restrict__ gate,
                                                              float* __restrict__ dw, int ipe, int cout, int cin,
                                                              int taps, int coutp, int cinp, int gate_ld) {
    const int e = blockIdx.y;
    const int total = cout * cin * taps;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < total; i += gridDim.x * 256) {
        const int t = i % taps;
        const int c = (i / taps) % cin;
        const int k = i / (taps * cin);
        float s = 0.f;
        for (int n = e * ipe; n < (e + 1) * ipe; ++n)
            s += gate[(size_t)n * gate_ld + c] * G[(((size_t)n * taps + t) * coutp + k) * cinp + c];
        dw[(size_t)e * total + i] = s;
    }
}

__global__ void __launch_bounds__(256) eca_stem_fold_ds_kernel(const float* __restrict__ G, const float* const* __restrict__ w,
                                                              float* __restrict__ ds, int ipe, int cout, int cin, int taps,
                                                              int coutp, int cinp, int gld) {
    const int n = blockIdx.x, e = n / ipe;
    const float* we = w[e];
    __shared__ float red[256];
    // thread -> (channel c, slice of the cout*taps terms); gld = row length of gate / ds (16 for the 12-ch stem)
    const int lanes_per_c = 256 / gld;
    const int c = threadIdx.x % gld, sl = threadIdx.x / gld;
    float s = 0.f;
    if (c < cin && sl < lanes_per_c)       // row lengths that do not divide 256 (144 for the 138-ch stem) leave idle lanes
        for (int kt = sl; kt < cout * taps; kt += lanes_per_c) {
            const int k = kt / taps, t = kt % taps;
            s += we[((size_t)k * cin + c) * taps + t] * G[(((size_t)n * taps + t) * coutp + k) * cinp + c];
        }
    red[threadIdx.x] = s;
    __syncthreads();
    if (threadIdx.x < gld) {
        float tsum = 0.f;
        for (int q = 0; q < lanes_per_c; ++q) tsum += red[q * gld + threadIdx.x];
        ds[(size_t)n * gld + threadIdx.x] = threadIdx.x < cin ? tsum : 0.f;
    }
}

static inline int pow2ceil(int v) { int g = 1; while (g < v) g <<= 1; return g; }

extern "C" {

int pmoe_pack_conv_weights_fp8(const void* const* src_ptrs, void* fwd_e4m3, void* dgrd_bf16, float* wscale, float* oscale,
                               float in_scale, int32_t E, int32_t cout, int32_t cin, int32_t ks, int32_t coutp, int32_t cinp,
                               int32_t cinp2, int32_t coutp2, void* stream) {
    if (!src_ptrs || !wscale || !oscale || !(in_scale > 0.f)) return PMOE_ERR_ARG;
    if (coutp < cout || cinp < cin || (dgrd_bf16 && (cinp2 < cin || coutp2 < cout))) return PMOE_ERR_ARG;
    const int taps = ks * ks;
    hipLaunchKernelGGL(fp8_row_scale_kernel, dim3(coutp, E), dim3(256), 0, (hipStream_t)stream,
                       (const float* const*)src_ptrs, wscale, oscale, in_scale, cout, coutp, cin * taps);
    const long long n = (fwd_e4m3 ? (long long)coutp * taps * cinp : 0) + (dgrd_bf16 ? (long long)cinp2 * taps * coutp2 : 0);
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    if (n > 0)
        hipLaunchKernelGGL(pack_w_fp8_kernel, dim3((int)g, E), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, wscale, (unsigned char*)fwd_e4m3, (bf16*)dgrd_bf16, cout, cin, taps,
                           coutp, cinp, cinp2, coutp2);
    return (int)hipGetLastError();
}

int pmoe_pack_conv_weights(const void* const* src_ptrs, void* fwd, void* dgrd, int32_t E, int32_t cout, int32_t cin,
                           int32_t ks, int32_t coutp, int32_t cinp, int32_t cinp2, int32_t coutp2, int32_t dtype,
                           void* stream) {
    if (coutp < cout || cinp < cin || (dgrd && (cinp2 < cin || coutp2 < cout))) return PMOE_ERR_ARG;
    const int taps = ks * ks;
    const long long n = (fwd ? (long long)coutp * taps * cinp : 0) + (dgrd ? (long long)cinp2 * taps * coutp2 : 0);
    if (n >= 0x7fffffffll) return PMOE_ERR_ARG;
    long long g = (n + 255) / 256;
    if (g > 1024) g = 1024;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((pack_w_kernel<bf16>), dim3((int)g, E), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, (bf16*)fwd, (bf16*)dgrd, cout, cin, taps, coutp, cinp, cinp2,
                           coutp2);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((pack_w_kernel<float>), dim3((int)g, E), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, (float*)fwd, (float*)dgrd, cout, cin, taps, coutp, cinp, cinp2,
                           coutp2);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_pack_conv_weights_gated(const void* const* src_ptrs, const float* gate, int32_t gate_ld, void* fwd, void* dgrd,
                                 int32_t N, int32_t ipe, int32_t cout, int32_t cin, int32_t ks, int32_t coutp, int32_t cinp,
                                 int32_t cinp2, int32_t coutp2, int32_t dtype, void* stream) {
    if (N < 1 || ipe < 1 || N % ipe || coutp < cout || cinp < cin || gate_ld < cin || !gate ||
        (dgrd && (cinp2 < cin || coutp2 < cout)))
        return PMOE_ERR_ARG;
    const int taps = ks * ks;
    const long long n = (fwd ? (long long)coutp * taps * cinp : 0) + (dgrd ? (long long)cinp2 * taps * coutp2 : 0);
    long long g = (n + 255) / 256;
    if (g > 64) g = 64;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((pack_w_gated_kernel<bf16>), dim3((int)g, N), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, gate, gate_ld, (bf16*)fwd, (bf16*)dgrd, ipe, cout, cin, taps, coutp,
                           cinp, cinp2, coutp2);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((pack_w_gated_kernel<float>), dim3((int)g, N), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, gate, gate_ld, (float*)fwd, (float*)dgrd, ipe, cout, cin, taps, coutp,
                           cinp, cinp2, coutp2);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_pack_conv_weights_scaled(const void* const* src_ptrs, const float* scale, const float* shift, const float* mean,
                                  void* fwd, float* bias, int32_t E, int32_t cout, int32_t cin, int32_t ks, int32_t coutp,
                                  int32_t cinp, int32_t dtype, void* stream) {
    if (E < 1 || coutp < cout || cinp < cin || !scale || !shift || !mean || !fwd || !bias) return PMOE_ERR_ARG;
    const int taps = ks * ks;
    long long g = ((long long)coutp * taps * cinp + 255) / 256;
    if (g > 1024) g = 1024;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((pack_w_scaled_kernel<bf16>), dim3((int)g, E), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, scale, shift, mean, (bf16*)fwd, bias, cout, cin, taps, coutp, cinp);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((pack_w_scaled_kernel<float>), dim3((int)g, E), dim3(256), 0, (hipStream_t)stream,
                           (const float* const*)src_ptrs, scale, shift, mean, (float*)fwd, bias, cout, cin, taps, coutp, cinp);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_unpack_conv_wgrad(const float* dw_ws, float* grads, int32_t E, int32_t cout, int32_t cin, int32_t ks,
                           int32_t coutp, int32_t cinp, void* stream) {
    const int taps = ks * ks;
    long long g = ((long long)cout * cin * taps + 255) / 256;
    if (g > 1024) g = 1024;
    hipLaunchKernelGGL(unpack_wgrad_kernel, dim3((int)g, E), dim3(256), 0, (hipStream_t)stream, dw_ws, grads, cout, cin,
                       taps, coutp, cinp);
    return (int)hipGetLastError();
}

int pmoe_pack_bias(const void* const* src_ptrs, float* dst, int32_t E, int32_t cout, int32_t coutp, void* stream) {
    hipLaunchKernelGGL(pack_bias_kernel, dim3((coutp + 255) / 256, E), dim3(256), 0, (hipStream_t)stream,
                       (const float* const*)src_ptrs, dst, cout, coutp);
    return (int)hipGetLastError();
}

int pmoe_eca_stem_fold(const float* G, const float* gate, const void* const* w_ptrs, float* dw, float* ds, int32_t N,
                       int32_t ipe, int32_t cout, int32_t cin, int32_t ks, int32_t coutp, int32_t cinp, int32_t gate_ld,
                       void* stream) {
    if (N % ipe || gate_ld > 256 || cin > cinp || cin > gate_ld || cout > coutp) return PMOE_ERR_ARG;
    const int taps = ks * ks, E = N / ipe;
    int g = (cout * cin * taps + 255) / 256;
    hipLaunchKernelGGL(eca_stem_fold_dw_kernel, dim3(g, E), dim3(256), 0, (hipStream_t)stream, G, gate, dw, ipe, cout, cin,
                       taps, coutp, cinp, gate_ld);
    hipLaunchKernelGGL(eca_stem_fold_ds_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, G,
                       (const float* const*)w_ptrs, ds, ipe, cout, cin, taps, coutp, cinp, gate_ld);
    return (int)hipGetLastError();
}

int pmoe_gate_mixture_fwd(const void* head, int32_t head_ld, const void* spd, int32_t spd_ld, float* probs, float* mean,
                          float* std_, float* speeds, int32_t B, int32_t E, int32_t alpha_relu, int32_t shared,
                          int32_t dtype, void* stream) {
    if (E < 1 || E > 64 || head_ld < (shared ? 5 * E : 5)) return PMOE_ERR_ARG;
    const int G = pow2ceil(E);
    const int blocks = (B * G + 255) / 256;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((gate_fwd_kernel<bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)head,
                           head_ld, (const bf16*)spd, spd_ld, probs, mean, std_, speeds, B, E, G, alpha_relu, shared);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((gate_fwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)head,
                           head_ld, (const float*)spd, spd_ld, probs, mean, std_, speeds, B, E, G, alpha_relu, shared);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_gate_mixture_bwd(const void* head, int32_t head_ld, const float* probs, const float* dprobs, const float* dmean,
                          const float* dstd, const float* dspeeds, void* dhead, void* dspd, int32_t spd_ld, int32_t B,
                          int32_t E, int32_t alpha_relu, int32_t shared, int32_t dtype, void* stream) {
    if (E < 1 || E > 64 || head_ld < (shared ? 5 * E : 5)) return PMOE_ERR_ARG;
    const int G = pow2ceil(E);
    const int blocks = (B * G + 255) / 256;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((gate_bwd_kernel<bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const bf16*)head,
                           head_ld, probs, dprobs, dmean, dstd, dspeeds, (bf16*)dhead, (bf16*)dspd, spd_ld, B, E, G,
                           alpha_relu, shared);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((gate_bwd_kernel<float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const float*)head,
                           head_ld, probs, dprobs, dmean, dstd, dspeeds, (float*)dhead, (float*)dspd, spd_ld, B, E, G,
                           alpha_relu, shared);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_moe_loss(const float* probs, const float* mean, const float* std_, const float* speeds, const float* actions,
                  const float* target_speed, float c0, float c1, float* loss, float* loglik, float* dprobs, float* dmean,
                  float* dstd, float* dspeeds, int32_t B, int32_t E, int32_t shared_speed, void* stream) {
    if (E < 1 || E > 64) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(moe_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, probs, mean, std_, speeds, actions,
                       target_speed, c0, c1, loss, loglik, dprobs, dmean, dstd, dspeeds, B, E, pow2ceil(E), shared_speed);
    return (int)hipGetLastError();
}

}  // extern "C"
