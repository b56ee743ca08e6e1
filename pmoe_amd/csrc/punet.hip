// Kernels that only the PU-Net / PMoE model types need:
//   * MaxPool2d(2,2)                                   (blocks/unet.py:29, forward only: the U-Nets are frozen on this path)
//   * ConvTranspose2d(k=2,s=2) scatter                 (unet.py:34-44): the 4 taps are one 1x1 GEMM with 4*Cout rows, this
//                                                       kernel interleaves them into the 2H x 2W output (a channel window of
//                                                       the skip-concat buffer, unet.py:71-84)
//   * channel-window copy                              (torch.cat along C, punet.py:104,113; view(B,-1,H,W), moe.py:311-313)
//   * tanh action head, L1/MSE losses, PMoE blend      (moe.py:317,353-356; loss.py:135-151)
// All of them are HBM-bound streaming kernels: 16-byte vectors, grid-stride, no LDS.
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
template <typename T>
__global__ void __launch_bounds__(256) maxpool2_kernel(const T* __restrict__ x, T* __restrict__ y, int N, int H, int W,
                                                      int C, int Ho, int Wo, int x_ld, int x_coff) {
    constexpr int VE = 16 / (int)sizeof(T);
    const int CV = C / VE;
    const long long total = (long long)N * Ho * Wo * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        long long t = i / CV;
        const int ox = (int)(t % Wo); t /= Wo;
        const int oy = (int)(t % Ho);
        const int n = (int)(t / Ho);
        const T* p = x + (((size_t)n * H + 2 * oy) * W + 2 * ox) * x_ld + x_coff + cv * VE;
        float a[VE], b[VE], c[VE], d[VE];
        unpack16<T>(ldg16(p), a);
        unpack16<T>(ldg16(p + x_ld), b);
        unpack16<T>(ldg16(p + (size_t)W * x_ld), c);
        unpack16<T>(ldg16(p + (size_t)W * x_ld + x_ld), d);
#pragma unroll
        for (int k = 0; k < VE; ++k) a[k] = fmaxf(fmaxf(a[k], b[k]), fmaxf(c[k], d[k]));
        stg16(y + (size_t)i * VE, pack16<T>(a));
    }
}

// src [N,H,W,4*C] with channel (dy*2+dx)*C + c  ->  dst[n, 2y+dy, 2x+dx, dst_coff + c]
template <typename T>
__global__ void __launch_bounds__(256) pixel_shuffle2_kernel(const T* __restrict__ src, T* __restrict__ dst, int N, int H,
                                                            int W, int C, int src_ld, int dst_ld, int dst_coff) {
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
        const v4i v = ldg16(src + (((size_t)n * H + y) * W + x) * src_ld + q * C + cv * VE);
        stg16(dst + (((size_t)n * 2 * H + 2 * y + (q >> 1)) * 2 * W + 2 * x + (q & 1)) * dst_ld + dst_coff + cv * VE, v);
    }
}

// rows x C elements between two channel windows; VEC = elements per access (1 when an offset is unaligned)
template <typename T, int VEC>
__global__ void __launch_bounds__(256) copy_window_kernel(const T* __restrict__ src, int src_ld, int src_coff,
                                                         T* __restrict__ dst, int dst_ld, int dst_coff, long long rows,
                                                         int C) {
    const int CV = C / VEC;
    const long long total = rows * CV;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int cv = (int)(i % CV);
        const long long r = i / CV;
        const T* s = src + (size_t)r * src_ld + src_coff + cv * VEC;
        T* d = dst + (size_t)r * dst_ld + dst_coff + cv * VEC;
        if (VEC == 1) *d = *s;
        else stg16(d, ldg16(s));
    }
}

// ------------------------------------------------------------------------------------------------
// PUNetExpert tail (moe.py:317): actions = tanh(head[:, 0:2]), pred_speed = spd[:, 0]
template <typename T>
__global__ void __launch_bounds__(256) action_head_fwd_kernel(const T* __restrict__ head, int head_ld,
                                                             const T* __restrict__ spd, int spd_ld,
                                                             float* __restrict__ actions, float* __restrict__ speeds, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    actions[b * 2 + 0] = tanhf(to_f32(head[(size_t)b * head_ld + 0]));
    actions[b * 2 + 1] = tanhf(to_f32(head[(size_t)b * head_ld + 1]));
    speeds[b] = to_f32(spd[(size_t)b * spd_ld]);
}

template <typename T>
__global__ void __launch_bounds__(256) action_head_bwd_kernel(const float* __restrict__ actions,
                                                             const float* __restrict__ dact,
                                                             const float* __restrict__ dspeeds, T* __restrict__ dhead,
                                                             int head_ld, T* __restrict__ dspd, int spd_ld, int B) {
    const int b = blockIdx.x * 256 + threadIdx.x;
    if (b >= B) return;
    T* hr = dhead + (size_t)b * head_ld;
    T* sr = dspd + (size_t)b * spd_ld;
    for (int i = 0; i < head_ld; ++i) {
        float g = 0.f;
        if (i < 2 && dact) {
            const float a = actions[b * 2 + i];
            g = dact[b * 2 + i] * (1.f - a * a);
        }
        hr[i] = from_f32<T>(g);
    }
    sr[0] = from_f32<T>(dspeeds ? dspeeds[b] : 0.f);
    for (int i = 1; i < spd_ld; ++i) sr[i] = from_f32<T>(0.f);
}

// punet_loss / pmoe_loss (loss.py:135-151): c0 * mean|a - a_gt| (+ c1 * mean (s - s_gt)^2), gradients in the same pass
__global__ void __launch_bounds__(256) action_loss_kernel(const float* __restrict__ actions, const float* __restrict__ speeds,
                                                         const float* __restrict__ act_gt, const float* __restrict__ spd_gt,
                                                         float c0, float c1, float* __restrict__ loss,
                                                         float* __restrict__ dact, float* __restrict__ dspeeds, int B) {
    float l1 = 0.f, l2 = 0.f;
    for (int i = threadIdx.x; i < 2 * B; i += 256) {
        const float d = actions[i] - act_gt[i];
        l1 += fabsf(d);
        dact[i] = c0 * (d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f)) / (2.f * (float)B);
    }
    if (speeds)
        for (int b = threadIdx.x; b < B; b += 256) {
            const float d = speeds[b] - spd_gt[b];
            l2 += d * d;
            dspeeds[b] = c1 * 2.f * d / (float)B;
        }
    __shared__ float r1[256], r2[256];
    r1[threadIdx.x] = l1;
    r2[threadIdx.x] = l2;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            r1[threadIdx.x] += r1[threadIdx.x + s];
            r2[threadIdx.x] += r2[threadIdx.x + s];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) loss[0] = c0 * r1[0] / (2.f * (float)B) + (speeds ? c1 * r2[0] / (float)B : 0.f);
}

// PMoE blend (moe.py:353-356): out[b][j] = tanh(w_j[0]*moe[b][j] + w_j[1]*punet[b][j] + bias_j), j=0 lateral, j=1 longitudinal
__global__ void __launch_bounds__(256) blend_fwd_kernel(const float* __restrict__ moe_act, const float* __restrict__ pu_act,
                                                       const float* __restrict__ lat_w, const float* __restrict__ lat_b,
                                                       const float* __restrict__ long_w, const float* __restrict__ long_b,
                                                       float* __restrict__ out, int B) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 2 * B) return;
    const int j = i & 1;
    const float* w = j ? long_w : lat_w;
    const float bias = j ? long_b[0] : lat_b[0];
    out[i] = tanhf(w[0] * moe_act[i] + w[1] * pu_act[i] + bias);
}

// one workgroup: dW_j = sum_b g*[moe, punet], db_j = sum_b g, dpunet = g * w_j[1];  g = dout * (1 - out^2)
__global__ void __launch_bounds__(256) blend_bwd_kernel(const float* __restrict__ moe_act, const float* __restrict__ pu_act,
                                                       const float* __restrict__ lat_w, const float* __restrict__ long_w,
                                                       const float* __restrict__ out, const float* __restrict__ dout,
                                                       float* __restrict__ dlat_w, float* __restrict__ dlat_b,
                                                       float* __restrict__ dlong_w, float* __restrict__ dlong_b,
                                                       float* __restrict__ dpu, int B) {
    float acc[6] = {0, 0, 0, 0, 0, 0};       // lat: w0 w1 b ; long: w0 w1 b
    for (int b = threadIdx.x; b < B; b += 256) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int i = b * 2 + j;
            const float g = dout[i] * (1.f - out[i] * out[i]);
            acc[j * 3 + 0] += g * moe_act[i];
            acc[j * 3 + 1] += g * pu_act[i];
            acc[j * 3 + 2] += g;
            if (dpu) dpu[i] = g * (j ? long_w[1] : lat_w[1]);
        }
    }
    __shared__ float red[6][256];
#pragma unroll
    for (int k = 0; k < 6; ++k) red[k][threadIdx.x] = acc[k];
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s)
#pragma unroll
            for (int k = 0; k < 6; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        dlat_w[0] = red[0][0]; dlat_w[1] = red[1][0]; dlat_b[0] = red[2][0];
        dlong_w[0] = red[3][0]; dlong_w[1] = red[4][0]; dlong_b[0] = red[5][0];
    }
}

// ------------------------------------------------------------------------------------------------
extern "C" {

int pmoe_maxpool2s2_fwd(const void* x, void* y, int32_t N, int32_t H, int32_t W, int32_t C, int32_t x_ld, int32_t x_coff,
                        int32_t dtype, void* stream) {
    if (H < 2 || W < 2 || N < 1) return PMOE_ERR_ARG;
    const int Ho = H / 2, Wo = W / 2;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || x_ld % VE || x_coff % VE || x_coff + C > x_ld) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((maxpool2_kernel<T>), dim3(grid_for((long long)N * Ho * Wo * (C / VE))), dim3(256), 0,
                           (hipStream_t)stream, (const T*)x, (T*)y, N, H, W, C, Ho, Wo, x_ld, x_coff);
        return (int)hipGetLastError();
    });
}

int pmoe_pixel_shuffle2(const void* src, void* dst, int32_t N, int32_t H, int32_t W, int32_t C, int32_t src_ld,
                        int32_t dst_ld, int32_t dst_coff, int32_t dtype, void* stream) {
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        if (C % VE || src_ld % VE || src_ld < 4 * C || dst_ld % VE || dst_coff % VE || dst_coff + C > dst_ld) return PMOE_ERR_ARG;
        hipLaunchKernelGGL((pixel_shuffle2_kernel<T>), dim3(grid_for((long long)N * H * W * 4 * (C / VE))), dim3(256), 0,
                           (hipStream_t)stream, (const T*)src, (T*)dst, N, H, W, C, src_ld, dst_ld, dst_coff);
        return (int)hipGetLastError();
    });
}

int pmoe_copy_window(const void* src, int32_t src_ld, int32_t src_coff, void* dst, int32_t dst_ld, int32_t dst_coff,
                     int64_t rows, int32_t C, int32_t dtype, void* stream) {
    if (rows < 0 || C < 1 || src_coff + C > src_ld || dst_coff + C > dst_ld) return PMOE_ERR_ARG;
    if (rows == 0) return 0;
    DISPATCH_DT(dtype, {
        constexpr int VE = 16 / (int)sizeof(T);
        const bool vec = !(C % VE) && !(src_ld % VE) && !(src_coff % VE) && !(dst_ld % VE) && !(dst_coff % VE);
        if (vec)
            hipLaunchKernelGGL((copy_window_kernel<T, VE>), dim3(grid_for(rows * (C / VE))), dim3(256), 0,
                               (hipStream_t)stream, (const T*)src, src_ld, src_coff, (T*)dst, dst_ld, dst_coff,
                               (long long)rows, C);
        else
            hipLaunchKernelGGL((copy_window_kernel<T, 1>), dim3(grid_for(rows * C)), dim3(256), 0, (hipStream_t)stream,
                               (const T*)src, src_ld, src_coff, (T*)dst, dst_ld, dst_coff, (long long)rows, C);
        return (int)hipGetLastError();
    });
}

int pmoe_action_head_fwd(const void* head, int32_t head_ld, const void* spd, int32_t spd_ld, float* actions,
                         float* speeds, int32_t B, int32_t dtype, void* stream) {
    if (B < 1 || head_ld < 2 || spd_ld < 1) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((action_head_fwd_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                           (const T*)head, head_ld, (const T*)spd, spd_ld, actions, speeds, B);
        return (int)hipGetLastError();
    });
}

int pmoe_action_head_bwd(const float* actions, const float* dactions, const float* dspeeds, void* dhead, int32_t head_ld,
                         void* dspd, int32_t spd_ld, int32_t B, int32_t dtype, void* stream) {
    if (B < 1 || head_ld < 2 || spd_ld < 1) return PMOE_ERR_ARG;
    DISPATCH_DT(dtype, {
        hipLaunchKernelGGL((action_head_bwd_kernel<T>), dim3((B + 255) / 256), dim3(256), 0, (hipStream_t)stream, actions,
                           dactions, dspeeds, (T*)dhead, head_ld, (T*)dspd, spd_ld, B);
        return (int)hipGetLastError();
    });
}

int pmoe_action_loss(const float* actions, const float* speeds, const float* actions_gt, const float* speed_gt, float c0,
                     float c1, float* loss, float* dactions, float* dspeeds, int32_t B, void* stream) {
    if (B < 1 || (speeds && (!speed_gt || !dspeeds))) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(action_loss_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, actions, speeds, actions_gt, speed_gt,
                       c0, c1, loss, dactions, dspeeds, B);
    return (int)hipGetLastError();
}

int pmoe_blend_fwd(const float* moe_actions, const float* punet_actions, const float* lat_w, const float* lat_b,
                   const float* long_w, const float* long_b, float* out, int32_t B, void* stream) {
    if (B < 1) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(blend_fwd_kernel, dim3((2 * B + 255) / 256), dim3(256), 0, (hipStream_t)stream, moe_actions,
                       punet_actions, lat_w, lat_b, long_w, long_b, out, B);
    return (int)hipGetLastError();
}

int pmoe_blend_bwd(const float* moe_actions, const float* punet_actions, const float* lat_w, const float* long_w,
                   const float* out, const float* dout, float* dlat_w, float* dlat_b, float* dlong_w, float* dlong_b,
                   float* dpunet, int32_t B, void* stream) {
    if (B < 1) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(blend_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, moe_actions, punet_actions, lat_w,
                       long_w, out, dout, dlat_w, dlat_b, dlong_w, dlong_b, dpunet, B);
    return (int)hipGetLastError();
}

}  // extern "C"
