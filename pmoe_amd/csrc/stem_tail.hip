// Fused "stem tail" of the ECA/ResNet stem:   z2 -> BN(c2)+ReLU -> BN(bn1)+ReLU -> MaxPool(3,2,1)
// (blocks/basics.py:121-123 conv2.{1,2}; torchvision ResNet.bn1 / relu / maxpool kept by backbone.py:63-65).
//
// At 256x256 these are the three largest activations of the network (2.1 GB each for E*B = 256
// images in bf16).  Unfused, forward + backward move ~37 GB through HBM for them; here the two
// intermediate activations (a2, a3) and their gradients are never materialised: every pass
// re-derives them from z2 in registers (two fused multiply-adds and two max per element).
//   forward : stats pass (sum a2, sum a2^2 for bn1)            1 read of z2
//             pool pass  (a3 -> 3x3/s2 max + winning tap)      1 read of z2, 1/4-size write
//   backward: phase 1 (bn1:  sum g3, sum g3*xhat1)             1 read of z2 + pooled grad
//             phase 2 (c2 :  sum g2, sum g2*xhat2)             1 read of z2 + pooled grad
//             phase 3 (dz2 = BN backward)                      1 read of z2 + pooled grad, 1 write
// Train mode replaces phases 1 and 2 by ONE pass over the POOLED tensors (1/4 of the elements) + closed forms:
// the pooled gradient only reaches the winning pixels, whose a3 IS the stored pooled output y, so
// (a2 - mu1) = (y - b1)/sc1 and (z - mu2) = (a2 - b2)/sc2 are recovered from y; the terms of the BatchNorm backward
// that touch every pixel (xhat1*P + Q) only need channel moments of z2 (sum m, sum m*u, sum a2*u with m = [a2>0],
// u = z - mu2), which the forward statistics pass accumulates alongside.  See stem_tail_combine_kernel.
// Reductions: per-workgroup LDS, fixed-order partial rows (bit-reproducible).
#include "common.h"

struct TailConsts {            // per (expert, channel) f32 arrays
    const float *sc2, *sh2;    // BN(c2) scale/shift
    const float *sc1, *sh1;    // bn1 scale/shift
    const float *mu1, *is1;    // bn1 batch mean / invstd (of a2)
    const float *mu2, *is2;    // BN(c2) batch mean / invstd (of z2)
    const float *c11, *c21;    // bn1 backward means  (sum g3 / M, sum g3*xhat1 / M)
    const float *c12, *c22;    // BN(c2) backward means
};

// MODE 0: forward statistics of a2.  MODE 1/2/3: backward phases.
template <typename T, int MODE>
__global__ void __launch_bounds__(256) stem_tail_kernel(const T* __restrict__ z2, const T* __restrict__ dpool,
                                                       const uint8_t* __restrict__ amax, T* __restrict__ dz2,
                                                       TailConsts k, float* __restrict__ part, int nparts, int ipe,
                                                       int H, int W, int C, float* __restrict__ shiftc,
                                                       float* __restrict__ part_x) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, RL = 256 / CV;
    const int tid = threadIdx.x, cv = tid % CV, rl = tid / CV;
    const int e = blockIdx.y, pi = blockIdx.x;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    // a workgroup walks whole image rows: the (image,row) decode is wave-uniform, lanes sweep the row
    const int rows_total = ipe * H;
    const int rpp = (rows_total + nparts - 1) / nparts;
    int r0 = pi * rpp, r1 = r0 + rpp;
    if (r1 > rows_total) r1 = rows_total;

    // per-channel constants.  Everything is evaluated on CENTRED values ((z - mu2), (a2 - mu1)): no cancellation.
    //   a2 = relu((z - mu2)*sc2 + b2)   a3 = (a2 - mu1)*sc1 + b1
    //   xhat1 = (a2 - mu1)*is1          da2 = g3*sc1 + xhat1*P + Q     P = -sc1*c21, Q = -sc1*c11   (bn1 backward)
    //   xhat2 = (z - mu2)*is2           dz2 = g2*sc2 + xhat2*R + S     R = -sc2*c22, S = -sc2*c12   (conv2-BN backward)
    float sc2[VE], sh2[VE], mu2[VE], sc1[VE], sh1[VE], mu1[VE], is1[VE], P[VE], Q[VE], is2[VE], R[VE], S[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        const int c = e * C + cv * VE + i;
        sc2[i] = k.sc2[c]; sh2[i] = k.sh2[c]; mu2[i] = k.mu2[c];
        sc1[i] = MODE ? k.sc1[c] : 0.f; sh1[i] = MODE ? k.sh1[c] : 0.f; mu1[i] = MODE ? k.mu1[c] : 0.f;
        is1[i] = MODE ? k.is1[c] : 0.f;
        P[i] = Q[i] = is2[i] = R[i] = S[i] = 0.f;
        if (MODE >= 2) {
            P[i] = -sc1[i] * k.c21[c];
            Q[i] = -sc1[i] * k.c11[c];
            is2[i] = k.is2[c];
        }
        if (MODE == 3) {
            R[i] = -sc2[i] * k.c22[c];
            S[i] = -sc2[i] * k.c12[c];
        }
    }
    float s1[VE], s2[VE], x0[VE], x1[VE], x2[VE];       // x*: MODE 0 extra moments (sum m, sum m*u, sum a2*u)
#pragma unroll
    for (int i = 0; i < VE; ++i) s1[i] = s2[i] = x0[i] = x1[i] = x2[i] = 0.f;
    // MODE 0: deviations from the channel's value at the expert's first pixel (see colstats_kernel)
    float c0v[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) c0v[i] = 0.f;
    if (MODE == 0 && shiftc) {
        float z0[VE];
        unpack16<T>(ldg16(z2 + (size_t)e * ipe * H * W * C + cv * VE), z0);
#pragma unroll
        for (int i = 0; i < VE; ++i) c0v[i] = fmaxf((z0[i] - mu2[i]) * sc2[i] + sh2[i], 0.f);
        if (pi == 0 && rl == 0) {
#pragma unroll
            for (int i = 0; i < VE; ++i) shiftc[e * C + cv * VE + i] = c0v[i];
        }
    }

    for (int row = r0; row < r1; ++row) {
      // wave-uniform row geometry: scalar base pointers, lanes add a small 32-bit offset
      const int yy = row % H;
      const int n = e * ipe + row / H;
      const T* zrow = z2 + ((size_t)n * H + yy) * W * C;
      T* dzrow = MODE == 3 ? dz2 + ((size_t)n * H + yy) * W * C : nullptr;
      const int oyA = yy >> 1, oyB = (yy + 1) >> 1;                  // the one or two pooled rows whose windows hold yy
      const bool twoRows = oyB != oyA && oyB < Ho;
      const size_t prA = ((size_t)n * Ho + oyA) * Wo * C, prB = ((size_t)n * Ho + oyB) * Wo * C;
      const int tapA = (yy - (2 * oyA - 1)) * 3, tapB = (yy - (2 * oyB - 1)) * 3;
      for (int xx = rl; xx < W; xx += RL) {
        const int voff = xx * C + cv * VE;
        float zv[VE], a2[VE];
        unpack16<T>(ldg16(zrow + voff), zv);
#pragma unroll
        for (int i = 0; i < VE; ++i) a2[i] = fmaxf((zv[i] - mu2[i]) * sc2[i] + sh2[i], 0.f);
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { const float d = a2[i] - c0v[i]; s1[i] += d; s2[i] += d * d; }
            if (part_x) {
#pragma unroll
                for (int i = 0; i < VE; ++i) {
                    const float u = zv[i] - mu2[i];
                    x0[i] += a2[i] > 0.f ? 1.f : 0.f;
                    x1[i] += a2[i] > 0.f ? u : 0.f;
                    x2[i] += a2[i] * u;
                }
            }
            continue;
        }
        // gradient arriving at a3 through the max-pool: gather from the <= 4 windows that contain (yy,xx)
        float g3[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) g3[i] = 0.f;
        const int oxA = xx >> 1, oxB = (xx + 1) >> 1;
        auto gather = [&](size_t prow, int ox, int tap) {
            const size_t po = prow + (size_t)(ox * C + cv * VE);
            float d[VE];
            unpack16<T>(ldg16(dpool + po), d);
            unsigned long long amw;                                  // the VE winning taps as one packed word
            if (VE == 8) amw = *reinterpret_cast<const unsigned long long*>(amax + po);
            else amw = *reinterpret_cast<const unsigned*>(amax + po);
#pragma unroll
            for (int i = 0; i < VE; ++i)
                if (((unsigned)(amw >> (8 * i)) & 0x7fu) == (unsigned)tap) g3[i] += d[i];   // bit 7 = winner's a2 > 0
        };
        const int tqA = xx - (2 * oxA - 1), tqB = xx - (2 * oxB - 1);
        const bool twoCols = oxB != oxA && oxB < Wo;
        gather(prA, oxA, tapA + tqA);
        if (twoCols) gather(prA, oxB, tapA + tqB);
        if (twoRows) {
            gather(prB, oxA, tapB + tqA);
            if (twoCols) gather(prB, oxB, tapB + tqB);
        }
#pragma unroll
        for (int i = 0; i < VE; ++i) g3[i] = ((a2[i] - mu1[i]) * sc1[i] + sh1[i]) > 0.f ? g3[i] : 0.f;     // ReLU after bn1
        if (MODE == 1) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { s1[i] += g3[i]; s2[i] += g3[i] * ((a2[i] - mu1[i]) * is1[i]); }
            continue;
        }
        float g2[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            const float da2 = g3[i] * sc1[i] + (((a2[i] - mu1[i]) * is1[i]) * P[i] + Q[i]);    // bn1 backward
            g2[i] = a2[i] > 0.f ? da2 : 0.f;                                                   // ReLU after BN(c2)
        }
        if (MODE == 2) {
#pragma unroll
            for (int i = 0; i < VE; ++i) { s1[i] += g2[i]; s2[i] += g2[i] * ((zv[i] - mu2[i]) * is2[i]); }
            continue;
        }
        float o[VE];
#pragma unroll
        for (int i = 0; i < VE; ++i) o[i] = g2[i] * sc2[i] + (((zv[i] - mu2[i]) * is2[i]) * R[i] + S[i]);
        stg16(dzrow + voff, pack16<T>(o));
      }
    }
    if (MODE == 3) return;
    __shared__ float red[2][256 * 8];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        red[0][rl * C + cv * VE + i] = s1[i];
        red[1][rl * C + cv * VE + i] = s2[i];
    }
    __syncthreads();
    for (int c = tid; c < 2 * C; c += 256) {
        const int which = c / C, cc = c % C;
        float s = 0.f;
        for (int q = 0; q < RL; ++q) s += red[which][q * C + cc];
        part[(((size_t)e * nparts + pi) * 2 + which) * C + cc] = s;
    }
    if (MODE == 0 && part_x) {
        for (int w = 0; w < 3; ++w) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < VE; ++i) red[0][rl * C + cv * VE + i] = w == 0 ? x0[i] : (w == 1 ? x1[i] : x2[i]);
            __syncthreads();
            for (int c = tid; c < C; c += 256) {
                float s = 0.f;
                for (int q = 0; q < RL; ++q) s += red[0][q * C + c];
                part_x[(((size_t)e * nparts + pi) * 3 + w) * C + c] = s;
            }
        }
    }
}

// Train-mode backward, pooled pass: over the POOLED tensors only.  g = dpool * [y > 0]; recovered (a2 - mu1) = (y - b1)/sc1,
// xhat1 = that * is1, m = bit 7 of the argmax byte, xhat2 = ((a2 - b2)/sc2) * is2.
// part4 [E][nparts][4][C] = (sum g, sum g*xhat1, sum g*m, sum g*m*xhat2).
template <typename T>
__global__ void __launch_bounds__(256) stem_tail_pooled_kernel(const T* __restrict__ y, const T* __restrict__ dpool,
                                                              const uint8_t* __restrict__ amax, TailConsts k,
                                                              float* __restrict__ part4, int nparts, long long rpe, int C) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE, RL = 256 / CV;
    const int tid = threadIdx.x, cv = tid % CV, rl = tid / CV;
    const int e = blockIdx.y, pi = blockIdx.x;
    const long long rpp = (rpe + nparts - 1) / nparts;
    long long r0 = (long long)pi * rpp, r1 = r0 + rpp;
    if (r1 > rpe) r1 = rpe;
    float sh1[VE], mu1[VE], is1[VE], isc1[VE], sh2[VE], isc2[VE], is2[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        const int c = e * C + cv * VE + i;
        const float a = k.sc1[c], b = k.sc2[c];
        sh1[i] = k.sh1[c]; mu1[i] = k.mu1[c]; is1[i] = k.is1[c]; sh2[i] = k.sh2[c]; is2[i] = k.is2[c];
        isc1[i] = a != 0.f ? 1.f / a : 0.f;          // gamma == 0: the channel's xhat cannot be recovered (contributes 0)
        isc2[i] = b != 0.f ? 1.f / b : 0.f;
    }
    float s[4][VE];
#pragma unroll
    for (int w = 0; w < 4; ++w)
#pragma unroll
        for (int i = 0; i < VE; ++i) s[w][i] = 0.f;
    for (long long r = r0 + rl; r < r1; r += RL) {
        const size_t off = ((size_t)e * rpe + r) * C + cv * VE;
        float yv[VE], dv[VE];
        unpack16<T>(ldg16(y + off), yv);
        unpack16<T>(ldg16(dpool + off), dv);
        unsigned long long amw;
        if (VE == 8) amw = *reinterpret_cast<const unsigned long long*>(amax + off);
        else amw = *reinterpret_cast<const unsigned*>(amax + off);
#pragma unroll
        for (int i = 0; i < VE; ++i) {
            const float g = yv[i] > 0.f ? dv[i] : 0.f;
            const float da = (yv[i] - sh1[i]) * isc1[i];                 // a2 - mu1 at the winning pixel
            const float gm = ((unsigned)(amw >> (8 * i)) & 0x80u) ? g : 0.f;
            const float u = (da + mu1[i] - sh2[i]) * isc2[i];            // z - mu2 there (only used where a2 > 0)
            s[0][i] += g;
            s[1][i] += g * (da * is1[i]);
            s[2][i] += gm;
            s[3][i] += gm * (u * is2[i]);
        }
    }
    __shared__ float red[256 * 8];
    for (int w = 0; w < 4; ++w) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < VE; ++i) red[rl * C + cv * VE + i] = s[w][i];
        __syncthreads();
        for (int c = tid; c < C; c += 256) {
            float t = 0.f;
            for (int q = 0; q < RL; ++q) t += red[q * C + c];
            part4[(((size_t)e * nparts + pi) * 4 + w) * C + c] = t;
        }
    }
}

// grid (E), threads over channels: pooled sums + forward moments -> the two (sum g, sum g*xhat) rows that
// bn_bwd_finalize expects for bn1 (out1) and for the conv2 BatchNorm (out2).  count = pixels per expert (B*H*W).
__global__ void __launch_bounds__(256) stem_tail_combine_kernel(const float* __restrict__ part4, int np4,
                                                               const float* __restrict__ part_x, int npx, TailConsts k,
                                                               float count, float* __restrict__ out1,
                                                               float* __restrict__ out2, int C) {
    const int e = blockIdx.x;
    for (int c = threadIdx.x; c < C; c += 256) {
        float p[4] = {0, 0, 0, 0}, x[3] = {0, 0, 0};
        for (int i = 0; i < np4; ++i)
#pragma unroll
            for (int w = 0; w < 4; ++w) p[w] += part4[(((size_t)e * np4 + i) * 4 + w) * C + c];
        for (int i = 0; i < npx; ++i)
#pragma unroll
            for (int w = 0; w < 3; ++w) x[w] += part_x[(((size_t)e * npx + i) * 3 + w) * C + c];
        const int ec = e * C + c;
        const float sc1 = k.sc1[ec], mu1 = k.mu1[ec], is1 = k.is1[ec], is2 = k.is2[ec];
        const float c11 = p[0] / count, c21 = p[1] / count;              // bn1 backward means
        const float P = -sc1 * c21, Q = -sc1 * c11;
        const float M0 = x[0], Mu = x[1], Au = x[2];
        const float Mx1 = is1 * mu1 * (count - M0);                      // sum m*xhat1 = is1*(sum a2 - mu1*sum m), sum a2 = count*mu1
        const float Mx2 = is2 * Mu;                                      // sum m*xhat2
        const float Mx12 = is1 * is2 * (Au - mu1 * Mu);                  // sum m*xhat1*xhat2
        out1[(e * 2 + 0) * C + c] = p[0];
        out1[(e * 2 + 1) * C + c] = p[1];
        out2[(e * 2 + 0) * C + c] = sc1 * p[2] + P * Mx1 + Q * M0;
        out2[(e * 2 + 1) * C + c] = sc1 * p[3] + P * Mx12 + Q * Mx2;
    }
}

// forward pool pass: a3 = relu(bn1(relu(bn_c2(z2)))) -> max over the 3x3/s2 window (first max wins), tap kept
template <typename T>
__global__ void __launch_bounds__(256) stem_tail_pool_kernel(const T* __restrict__ z2, T* __restrict__ y,
                                                            uint8_t* __restrict__ am, const float* __restrict__ sc2a,
                                                            const float* __restrict__ sh2a, const float* __restrict__ sc1a,
                                                            const float* __restrict__ sh1a, const float* __restrict__ mu2a,
                                                            const float* __restrict__ mu1a, int N, int ipe, int H, int W,
                                                            int C, int Ho, int Wo) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const long long total = (long long)N * Ho * Wo * CV;
    // the grid stride is a multiple of CV (a power of two <= 256), so a thread keeps ONE channel vector; its 6 constant
    // vectors are reloaded only when the sweep crosses into the next expert
    const int cv = (int)(threadIdx.x & (CV - 1));
    int e_cur = -1;
    float sc2[VE], sh2[VE], sc1[VE], sh1[VE], mu2[VE], mu1[VE];
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const int e = n / ipe;
        if (e != e_cur) {
            e_cur = e;
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                const int c = e * C + cv * VE + q;
                sc2[q] = sc2a[c]; sh2[q] = sh2a[c]; sc1[q] = sc1a[c]; sh1[q] = sh1a[c]; mu2[q] = mu2a[c]; mu1[q] = mu1a[c];
            }
        }
        float best[VE];
        int bi[VE];                      // winning tap (0..8) | 0x80 if the winner's a2 > 0 (used by the pooled backward pass)
#pragma unroll
        for (int q = 0; q < VE; ++q) { best[q] = -INFINITY; bi[q] = 0; }
        bool first = true;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int q3 = 0; q3 < 3; ++q3) {
                const int yy = 2 * oy - 1 + r, xx = 2 * ox - 1 + q3;
                if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
                    float v[VE];
                    unpack16<T>(ldg16(z2 + (((size_t)n * H + yy) * W + xx) * C + cv * VE), v);
#pragma unroll
                    for (int q = 0; q < VE; ++q) {
                        const float a2 = fmaxf((v[q] - mu2[q]) * sc2[q] + sh2[q], 0.f);
                        // round like the unfused path stores a3 (T precision) so ties resolve identically
                        const float a3 = to_f32(from_f32<T>(fmaxf((a2 - mu1[q]) * sc1[q] + sh1[q], 0.f)));
                        if (first || a3 > best[q]) { best[q] = a3; bi[q] = (r * 3 + q3) | (a2 > 0.f ? 0x80 : 0); }
                    }
                    first = false;
                }
            }
        const size_t off = (size_t)i * VE;
        stg16(y + off, pack16<T>(best));
#pragma unroll
        for (int q = 0; q < VE; ++q) am[off + q] = (uint8_t)bi[q];
    }
}

// ---- row-walking variants of the pool pass and of backward phase 3 (the default; PMOE_STEM_WALK=0 = the gather kernels above) ----
// A group of CV lanes (one 16-byte channel vector each) owns 2*KO adjacent input columns = KO pooled columns and walks DOWN a
// strip of rows: every z2 element is loaded once (+ one halo column, + one halo row per strip) and its two BatchNorm+ReLU
// evaluated once, instead of 2.25 gathers and 2.25 evaluations per element; the pooled gradient / winning taps of the <= 2
// pooled rows a pixel row touches stay in registers.  Same arithmetic, same tie rule (first maximum in (row, column) scan
// order of values rounded to T), same summation order as the kernels above: results are bit-identical.
__device__ __forceinline__ void put_byte(unsigned (&w)[2], int i, unsigned code) {
    const int sh = 8 * (i & 3);
    w[i >> 2] = (w[i >> 2] & ~(0xffu << sh)) | (code << sh);
}
__device__ __forceinline__ unsigned get_byte(const unsigned (&w)[2], int i) { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; }

// Pool pass, bf16: a3 >= 0, so the bf16 bit pattern orders like the value and "first maximum in scan order" becomes ONE
// unsigned max over keys  (bf16(a3) << 16) | (8 - tap) << 1 | [a2 > 0]  -- no per-element compare/select chains.
// Horizontal: max over the 3 columns with (2 - column) << 1 in the low bits; vertical: + 6 * (2 - row).
template <int KO>
__global__ void __launch_bounds__(256) stem_tail_pool_walk_kernel(const bf16* __restrict__ z2, bf16* __restrict__ y,
                                                                 uint8_t* __restrict__ am, const float* __restrict__ sc2a,
                                                                 const float* __restrict__ sh2a, const float* __restrict__ sc1a,
                                                                 const float* __restrict__ sh1a, const float* __restrict__ mu2a,
                                                                 const float* __restrict__ mu1a, int N, int ipe, int H, int W,
                                                                 int C, int Ho, int Wo, int G, int NS, int SR, int lcv) {
    constexpr int VE = 8, NP = 2 * KO + 1;
    const int CV = 1 << lcv, cv = threadIdx.x & (CV - 1);
    const long long gi = ((long long)blockIdx.x * 256 + threadIdx.x) >> lcv;
    if (gi >= (long long)N * NS * G) return;
    const int g = (int)(gi % G), strip = (int)((gi / G) % NS), n = (int)(gi / ((long long)G * NS)), e = n / ipe;
    float sc2[VE], sh2[VE], sc1[VE], sh1[VE], mu2[VE], mu1[VE];
#pragma unroll
    for (int q = 0; q < VE; ++q) {
        const int c = e * C + cv * VE + q;
        sc2[q] = sc2a[c]; sh2[q] = sh2a[c]; sc1[q] = sc1a[c]; sh1[q] = sh1a[c]; mu2[q] = mu2a[c]; mu1[q] = mu1a[c];
    }
    const int x0 = 2 * KO * g - 1;                                  // input column of pixel slot 0 (slot p <-> x0 + p)
    const bf16* zim = z2 + (size_t)n * H * W * C + cv * VE;
    auto load_row = [&](int yy, v4i (&raw)[NP]) {
        const bf16* zr = zim + (size_t)yy * W * C;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int xx = x0 + p;
            raw[p] = v4i{0, 0, 0, 0};
            if ((unsigned)xx < (unsigned)W) raw[p] = ldg16(zr + (size_t)xx * C);
        }
    };
    // one row: keys of the KO windows' best column, low bits (2 - column) << 1 | [a2 > 0]
    auto hrow = [&](const v4i (&raw)[NP], unsigned (&hk)[KO][VE]) {
#pragma unroll
        for (int k = 0; k < KO; ++k)
#pragma unroll
            for (int q = 0; q < VE; ++q) hk[k][q] = 0u;
#pragma unroll
        for (int p = 0; p < NP; ++p) {
            const int xx = x0 + p;
            if ((unsigned)xx >= (unsigned)W) continue;           // padding column: no candidate (a valid key is never 0)
            float v[VE];
            unpack16<bf16>(raw[p], v);
            unsigned base[VE];
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                const float a2 = fmaxf((v[q] - mu2[q]) * sc2[q] + sh2[q], 0.f);
                const bf16 a3 = from_f32<bf16>(fmaxf((a2 - mu1[q]) * sc1[q] + sh1[q], 0.f));
                base[q] = ((unsigned)__builtin_bit_cast(unsigned short, a3) << 16) | (a2 > 0.f ? 1u : 0u);
            }
#pragma unroll
            for (int k = 0; k < KO; ++k) {
                const int q3 = p - 2 * k;
                if (q3 < 0 || q3 > 2) continue;
#pragma unroll
                for (int q = 0; q < VE; ++q) hk[k][q] = max(hk[k][q], base[q] | (unsigned)((2 - q3) << 1));
            }
        }
    };
    const int oy0 = strip * SR, oy1 = min(oy0 + SR, Ho);
    unsigned hp[KO][VE];                                            // the row above the current window (row 2*oy - 1)
#pragma unroll
    for (int k = 0; k < KO; ++k)
#pragma unroll
        for (int q = 0; q < VE; ++q) hp[k][q] = 0u;
    if (2 * oy0 - 1 >= 0) {
        v4i raw[NP];
        load_row(2 * oy0 - 1, raw);
        hrow(raw, hp);
    }
    v4i rawA[NP], rawC[NP];
    load_row(2 * oy0, rawA);
    if (2 * oy0 + 1 < H) load_row(2 * oy0 + 1, rawC);
    for (int oy = oy0; oy < oy1; ++oy) {
        const bool haveC = 2 * oy + 1 < H;
        unsigned hA[KO][VE], hC[KO][VE];
        hrow(rawA, hA);
        if (haveC) hrow(rawC, hC);
        if (oy + 1 < oy1) {                                         // next window's rows: in flight during this one's arithmetic
            load_row(2 * oy + 2, rawA);
            if (2 * oy + 3 < H) load_row(2 * oy + 3, rawC);
        }
#pragma unroll
        for (int k = 0; k < KO; ++k) {
            unsigned yw[VE / 2], cw[2] = {0u, 0u};
#pragma unroll
            for (int q = 0; q < VE; ++q) {
                unsigned best = max(hA[k][q] + 6u, hp[k][q] ? hp[k][q] + 12u : 0u);
                if (haveC) best = max(best, hC[k][q]);
                const unsigned low = best & 0xffffu;
                const unsigned code = (8u - (low >> 1)) | ((low & 1u) << 7);
                cw[q >> 2] |= code << (8 * (q & 3));
                if (q & 1) yw[q >> 1] |= best & 0xffff0000u; else yw[q >> 1] = best >> 16;
                hp[k][q] = haveC ? hC[k][q] : 0u;
            }
            const int ox = KO * g + k;
            if (ox >= Wo) continue;
            const size_t off = (((size_t)n * Ho + oy) * Wo + ox) * C + cv * VE;
            stg16(y + off, v4i{(int)yw[0], (int)yw[1], (int)yw[2], (int)yw[3]});
            *reinterpret_cast<uint2*>(am + off) = make_uint2(cw[0], cw[1]);
        }
    }
}

// backward phase 3 (dz2), row-walking: a lane group owns 2*KO input columns and walks SR input rows.  The pooled gradient and the
// winning taps of the two pooled rows a pixel row can belong to stay in registers; per pixel row the taps are re-based once, four
// bytes at a time ((taps | 0x80) - 3*window_row: bit 7 keeps the byte lanes from borrowing), so that a candidate test is one
// byte compare.  The next row's z2 and the next pooled row are loaded while the current row is evaluated.
template <typename T, int KO>
__global__ void __launch_bounds__(256) stem_tail_dz_walk_kernel(const T* __restrict__ z2, const T* __restrict__ dpool,
                                                               const uint8_t* __restrict__ amax, T* __restrict__ dz2,
                                                               TailConsts k, int N, int ipe, int H, int W, int C, int Ho, int Wo,
                                                               int G, int NS, int SR, int lcv) {
    constexpr int VE = 16 / (int)sizeof(T), NX = 2 * KO, NPC = KO + 1, AW = VE / 4;
    const int CV = 1 << lcv, cv = threadIdx.x & (CV - 1);
    const long long gi = ((long long)blockIdx.x * 256 + threadIdx.x) >> lcv;
    if (gi >= (long long)N * NS * G) return;
    const int g = (int)(gi % G), strip = (int)((gi / G) % NS), n = (int)(gi / ((long long)G * NS)), e = n / ipe;
    float sc2[VE], sh2[VE], mu2[VE], sc1[VE], sh1[VE], mu1[VE], is1[VE], P[VE], Q[VE], is2[VE], R[VE], S[VE];
#pragma unroll
    for (int i = 0; i < VE; ++i) {
        const int c = e * C + cv * VE + i;
        sc2[i] = k.sc2[c]; sh2[i] = k.sh2[c]; mu2[i] = k.mu2[c];
        sc1[i] = k.sc1[c]; sh1[i] = k.sh1[c]; mu1[i] = k.mu1[c]; is1[i] = k.is1[c];
        P[i] = -sc1[i] * k.c21[c];
        Q[i] = -sc1[i] * k.c11[c];
        is2[i] = k.is2[c];
        R[i] = -sc2[i] * k.c22[c];
        S[i] = -sc2[i] * k.c12[c];
    }
    const int xb = NX * g, oxb = KO * g;                           // first input column / first pooled column of the group
    struct PRow { v4i d[NPC]; unsigned a[NPC][AW]; };
    auto load_prow = [&](int oy, PRow& r) {
        const size_t base = (((size_t)n * Ho + oy) * Wo) * C + cv * VE;
#pragma unroll
        for (int j = 0; j < NPC; ++j) {
            const int ox = oxb + j;
#pragma unroll
            for (int w = 0; w < AW; ++w) r.a[j][w] = 0xffffffffu;   // tap 0x7f: matches nothing
            r.d[j] = v4i{0, 0, 0, 0};
            if (ox < Wo) {
                const size_t po = base + (size_t)ox * C;
                r.d[j] = ldg16(dpool + po);
                if (VE == 8) { const uint2 w = *reinterpret_cast<const uint2*>(amax + po); r.a[j][0] = w.x; r.a[j][AW - 1] = w.y; }
                else r.a[j][0] = *reinterpret_cast<const unsigned*>(amax + po);
            }
        }
    };
    const int y0 = strip * SR, y1 = min(y0 + SR, H);               // SR even: y0 even
    const T* zim = z2 + (size_t)n * H * W * C + cv * VE;
    T* dim_ = dz2 + (size_t)n * H * W * C + cv * VE;
    auto load_z = [&](int yy, v4i (&zr)[NX]) {
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            zr[p] = v4i{0, 0, 0, 0};
            if (xb + p < W) zr[p] = ldg16(zim + ((size_t)yy * W + xb + p) * C);
        }
    };
    PRow cur, nxt;                                                  // pooled rows yy>>1 and (yy>>1)+1
    load_prow(y0 >> 1, cur);
    nxt = cur;
    v4i zr[NX], zn[NX];
    load_z(y0, zr);
    for (int yy = y0; yy < y1; ++yy) {
        const int oyA = yy >> 1, oyB = (yy + 1) >> 1;
        const bool twoRows = oyB != oyA && oyB < Ho;
        if (!(yy & 1) && oyA + 1 < Ho) load_prow(oyA + 1, nxt);     // used from the next (odd) row on
        if (yy + 1 < y1) load_z(yy + 1, zn);
        const unsigned subA = 0x01010101u * (unsigned)((yy - (2 * oyA - 1)) * 3), subB = 0x01010101u * (unsigned)((yy - (2 * oyB - 1)) * 3);
        unsigned relA[NPC][AW], relB[NPC][AW];
#pragma unroll
        for (int j = 0; j < NPC; ++j)
#pragma unroll
            for (int w = 0; w < AW; ++w) {
                relA[j][w] = (cur.a[j][w] | 0x80808080u) - subA;
                relB[j][w] = (nxt.a[j][w] | 0x80808080u) - subB;
            }
#pragma unroll
        for (int p = 0; p < NX; ++p) {
            const int xx = xb + p;
            if (xx >= W) continue;
            float zv[VE], a2[VE], g3[VE];
            unpack16<T>(zr[p], zv);
#pragma unroll
            for (int i = 0; i < VE; ++i) { a2[i] = fmaxf((zv[i] - mu2[i]) * sc2[i] + sh2[i], 0.f); g3[i] = 0.f; }
            const int jA = p >> 1, jB = (p + 1) >> 1;               // pooled column slots (relative to oxb)
            const unsigned tqA = 0x80u + (unsigned)(p - (2 * jA - 1)), tqB = 0x80u + (unsigned)(p - (2 * jB - 1));
            const bool twoCols = jB != jA && oxb + jB < Wo;
            auto gather = [&](const v4i& dr, const unsigned (&rel)[AW], unsigned want) {
                float d[VE];
                unpack16<T>(dr, d);
#pragma unroll
                for (int i = 0; i < VE; ++i)
                    if (((rel[i >> 2] >> (8 * (i & 3))) & 0xffu) == want) g3[i] += d[i];
            };
            gather(cur.d[jA], relA[jA], tqA);
            if (twoCols) gather(cur.d[jB], relA[jB], tqB);
            if (twoRows) {
                gather(nxt.d[jA], relB[jA], tqA);
                if (twoCols) gather(nxt.d[jB], relB[jB], tqB);
            }
            float o[VE];
#pragma unroll
            for (int i = 0; i < VE; ++i) {
                const float g3m = ((a2[i] - mu1[i]) * sc1[i] + sh1[i]) > 0.f ? g3[i] : 0.f;
                const float da2 = g3m * sc1[i] + (((a2[i] - mu1[i]) * is1[i]) * P[i] + Q[i]);
                const float g2 = a2[i] > 0.f ? da2 : 0.f;
                o[i] = g2 * sc2[i] + (((zv[i] - mu2[i]) * is2[i]) * R[i] + S[i]);
            }
            stg16(dim_ + ((size_t)yy * W + xx) * C, pack16<T>(o));
        }
        if (twoRows) cur = nxt;
#pragma unroll
        for (int p = 0; p < NX; ++p) zr[p] = zn[p];
    }
}

static inline int walk_enabled() {
    const char* ev = getenv("PMOE_STEM_WALK");
    return !ev || atoi(ev);
}
static inline int walk_ko() {                    // pooled columns per lane group (tools/ab_stem_tail.py: 2 vs 4)
    const char* ev = getenv("PMOE_STEM_WALK_KO");
    return ev && atoi(ev) == 4 ? 4 : 2;
}
static inline int ilog2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

static inline bool pow2i(int v) { return v > 0 && !(v & (v - 1)); }

extern "C" {

int pmoe_stem_tail_stats(const void* z2, const float* sc2, const float* sh2, const float* mu2, float* part,
                         int32_t nparts, float* shiftc, float* part_x, int32_t E, int32_t ipe, int32_t H, int32_t W,
                         int32_t C, int32_t dtype, void* stream) {
    TailConsts k{};
    k.sc2 = sc2; k.sh2 = sh2; k.mu2 = mu2;
    const int ve = dtype == PMOE_DT_BF16 ? 8 : 4;
    if (C % ve || !pow2i(C / ve) || C / ve > 256 || nparts < 1) return PMOE_ERR_ARG;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((stem_tail_kernel<bf16, 0>), dim3(nparts, E), dim3(256), 0, (hipStream_t)stream, (const bf16*)z2,
                           nullptr, nullptr, nullptr, k, part, nparts, ipe, H, W, C, shiftc, part_x);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((stem_tail_kernel<float, 0>), dim3(nparts, E), dim3(256), 0, (hipStream_t)stream,
                           (const float*)z2, nullptr, nullptr, nullptr, k, part, nparts, ipe, H, W, C, shiftc, part_x);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_stem_tail_pool(const void* z2, void* y, uint8_t* argmax, const float* sc2, const float* sh2, const float* sc1,
                        const float* sh1, const float* mu2, const float* mu1, int32_t N, int32_t ipe, int32_t H, int32_t W, int32_t C, int32_t dtype,
                        void* stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int ve = dtype == PMOE_DT_BF16 ? 8 : 4;
    if (C % ve || N % ipe || !pow2i(C / ve) || C / ve > 256) return PMOE_ERR_ARG;   // one channel vector per thread
    if (walk_enabled() && dtype == PMOE_DT_BF16) {            // (f32 is the parity mode: gather kernel)
        const int KO = walk_ko(), SR = 8;
        const int G = (Wo + KO - 1) / KO, NS = (Ho + SR - 1) / SR, lcv = ilog2(C / ve);
        const long long groups = (long long)N * NS * G, wgs = (groups * (C / ve) + 255) / 256;
        if (wgs > 0x7fffffffll) return PMOE_ERR_ARG;
#define POOL_WALK(K)                                                                                                        \
        hipLaunchKernelGGL((stem_tail_pool_walk_kernel<K>), dim3((int)wgs), dim3(256), 0, (hipStream_t)stream, (const bf16*)z2, \
                           (bf16*)y, argmax, sc2, sh2, sc1, sh1, mu2, mu1, N, ipe, H, W, C, Ho, Wo, G, NS, SR, lcv)
        if (KO == 4) POOL_WALK(4); else POOL_WALK(2);
#undef POOL_WALK
        return (int)hipGetLastError();
    }
    long long g = ((long long)N * Ho * Wo * (C / ve) + 255) / 256;
    if (g > 16384) g = 16384;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((stem_tail_pool_kernel<bf16>), dim3((int)g), dim3(256), 0, (hipStream_t)stream, (const bf16*)z2,
                           (bf16*)y, argmax, sc2, sh2, sc1, sh1, mu2, mu1, N, ipe, H, W, C, Ho, Wo);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((stem_tail_pool_kernel<float>), dim3((int)g), dim3(256), 0, (hipStream_t)stream,
                           (const float*)z2, (float*)y, argmax, sc2, sh2, sc1, sh1, mu2, mu1, N, ipe, H, W, C, Ho, Wo);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

/* phase 1: part = (sum g3, sum g3*xhat1); phase 2: (sum g2, sum g2*xhat2); phase 3: writes dz2.
 * consts: 12 per-(expert,channel) f32 arrays in the order of struct TailConsts (unused ones may be null). */
int pmoe_stem_tail_bwd(int32_t phase, const void* z2, const void* dpool, const uint8_t* argmax, void* dz2,
                       const float* const* consts, float* part, int32_t nparts, int32_t E, int32_t ipe, int32_t H,
                       int32_t W, int32_t C, int32_t dtype, void* stream) {
    TailConsts k{consts[0], consts[1], consts[2], consts[3], consts[4], consts[5],
                 consts[6], consts[7], consts[8], consts[9], consts[10], consts[11]};
    const int ve = dtype == PMOE_DT_BF16 ? 8 : 4;
    if (C % ve || !pow2i(C / ve) || C / ve > 256 || nparts < 1 || phase < 1 || phase > 3) return PMOE_ERR_ARG;
    dim3 grid(nparts, E), block(256);
    hipStream_t st = (hipStream_t)stream;
    if (phase == 3 && walk_enabled()) {
        const int KO = walk_ko(), SR = 16;
        const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1, N = E * ipe;
        const int G = (W + 2 * KO - 1) / (2 * KO), NS = (H + SR - 1) / SR, lcv = ilog2(C / ve);
        const long long groups = (long long)N * NS * G, wgs = (groups * (C / ve) + 255) / 256;
        if (wgs > 0x7fffffffll || (dtype != PMOE_DT_BF16 && dtype != PMOE_DT_F32)) return PMOE_ERR_ARG;
#define DZ_WALK(TT, K)                                                                                                       \
        hipLaunchKernelGGL((stem_tail_dz_walk_kernel<TT, K>), dim3((int)wgs), block, 0, st, (const TT*)z2, (const TT*)dpool, argmax, \
                           (TT*)dz2, k, N, ipe, H, W, C, Ho, Wo, G, NS, SR, lcv)
        if (dtype == PMOE_DT_BF16) { if (KO == 4) DZ_WALK(bf16, 4); else DZ_WALK(bf16, 2); }
        else { if (KO == 4) DZ_WALK(float, 4); else DZ_WALK(float, 2); }
#undef DZ_WALK
        return (int)hipGetLastError();
    }
#define TAIL_LAUNCH(TT, M)                                                                                           \
    hipLaunchKernelGGL((stem_tail_kernel<TT, M>), grid, block, 0, st, (const TT*)z2, (const TT*)dpool, argmax, (TT*)dz2, k, \
                       part, nparts, ipe, H, W, C, nullptr, nullptr)
    if (dtype == PMOE_DT_BF16) {
        if (phase == 1) TAIL_LAUNCH(bf16, 1); else if (phase == 2) TAIL_LAUNCH(bf16, 2); else TAIL_LAUNCH(bf16, 3);
    } else if (dtype == PMOE_DT_F32) {
        if (phase == 1) TAIL_LAUNCH(float, 1); else if (phase == 2) TAIL_LAUNCH(float, 2); else TAIL_LAUNCH(float, 3);
    } else {
        return PMOE_ERR_ARG;
    }
#undef TAIL_LAUNCH
    return (int)hipGetLastError();
}

/* train-mode replacement of phases 1 and 2: pooled pass (part4 [E][nparts][4][C]) and the closed-form combine
 * (out1 / out2 [E][2][C] = the (sum g, sum g*xhat) rows of bn1 / of the conv2 BatchNorm, for pmoe_bn_bwd_finalize).
 * consts as in pmoe_stem_tail_bwd (entries 0..7 used); part_x [E][npx][3][C] from pmoe_stem_tail_stats. */
int pmoe_stem_tail_pooled(const void* y, const void* dpool, const uint8_t* argmax, const float* const* consts, float* part4,
                          int32_t nparts, int32_t E, int64_t rows_per_expert, int32_t C, int32_t dtype, void* stream) {
    TailConsts k{consts[0], consts[1], consts[2], consts[3], consts[4], consts[5],
                 consts[6], consts[7], consts[8], consts[9], consts[10], consts[11]};
    const int ve = dtype == PMOE_DT_BF16 ? 8 : 4;
    if (C % ve || !pow2i(C / ve) || C / ve > 256 || nparts < 1 || rows_per_expert < 1) return PMOE_ERR_ARG;
    if (dtype == PMOE_DT_BF16)
        hipLaunchKernelGGL((stem_tail_pooled_kernel<bf16>), dim3(nparts, E), dim3(256), 0, (hipStream_t)stream, (const bf16*)y,
                           (const bf16*)dpool, argmax, k, part4, nparts, (long long)rows_per_expert, C);
    else if (dtype == PMOE_DT_F32)
        hipLaunchKernelGGL((stem_tail_pooled_kernel<float>), dim3(nparts, E), dim3(256), 0, (hipStream_t)stream,
                           (const float*)y, (const float*)dpool, argmax, k, part4, nparts, (long long)rows_per_expert, C);
    else
        return PMOE_ERR_ARG;
    return (int)hipGetLastError();
}

int pmoe_stem_tail_combine(const float* part4, int32_t np4, const float* part_x, int32_t npx, const float* const* consts,
                           int64_t count, float* out1, float* out2, int32_t E, int32_t C, void* stream) {
    TailConsts k{consts[0], consts[1], consts[2], consts[3], consts[4], consts[5],
                 consts[6], consts[7], consts[8], consts[9], consts[10], consts[11]};
    if (np4 < 1 || npx < 1 || count < 1) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(stem_tail_combine_kernel, dim3(E), dim3(256), 0, (hipStream_t)stream, part4, np4, part_x, npx, k,
                       (float)count, out1, out2, C);
    return (int)hipGetLastError();
}

}  // extern "C"
