// Input pipeline of the hot path's callers: Crop -> Resize -> ToTensor of
// model/data_loader.py:255-275 and autoagents/image_agent.py:71-78,132-136 on the GPU, bit-exact with Pillow's
// ImagingResample (BILINEAR = triangle filter, support stretched by the down-scaling factor): two separable passes over
// uint8 pixels with 22-bit fixed-point coefficients (computed by the host exactly as Resample.c:precompute_coeffs does,
// in double precision) and an 8-bit intermediate image.  Pure integer work, HBM/latency bound: one thread per output
// pixel, all channels.
#include "common.h"

static constexpr int PREC = 32 - 8 - 2;        // Pillow's PRECISION_BITS

__device__ __forceinline__ uint8_t clip8(int v) {
    v >>= PREC;
    return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// src [n][H0][W0][C] (rows row0 .. row0+rows-1 are used) -> dst [n][rows][Wout][C]
__global__ void __launch_bounds__(256) resample_h_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int n_img,
                                                        int H0, int W0, int row0, int rows, int C, int Wout,
                                                        const int32_t* __restrict__ bounds, const int32_t* __restrict__ kk,
                                                        int ksize) {
    const long long total = (long long)n_img * rows * Wout;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % Wout);
        long long t = i / Wout;
        const int r = (int)(t % rows);
        const int n = (int)(t / rows);
        const int xmin = bounds[2 * xx], cnt = bounds[2 * xx + 1];
        const uint8_t* p = src + (((size_t)n * H0 + row0 + r) * W0 + xmin) * C;
        const int32_t* k = kk + (size_t)xx * ksize;
        for (int c = 0; c < C; ++c) {
            int ss = 1 << (PREC - 1);
            for (int x = 0; x < cnt; ++x) ss += (int)p[(size_t)x * C + c] * k[x];
            dst[(size_t)i * C + c] = clip8(ss);
        }
    }
}

// src [n][Hin][W][C] uint8 -> dst [n][C][Hout][W]: f32 = value / 255 (ToTensor: CHW, [0,1]) or int64 = value
// (MaskPILToTensor, model/augmenter.py:52-54: segmentation labels of stage-1 training)
template <typename OUT> __device__ __forceinline__ OUT px_out(uint8_t v);
template <> __device__ __forceinline__ float px_out<float>(uint8_t v) { return (float)v / 255.0f; }
template <> __device__ __forceinline__ long long px_out<long long>(uint8_t v) { return (long long)v; }

template <typename OUT>
__global__ void __launch_bounds__(256) resample_v_kernel(const uint8_t* __restrict__ src, OUT* __restrict__ dst, int n_img,
                                                        int Hin, int W, int C, int Hout, const int32_t* __restrict__ bounds,
                                                        const int32_t* __restrict__ kk, int ksize) {
    const long long total = (long long)n_img * Hout * W;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int xx = (int)(i % W);
        long long t = i / W;
        const int yy = (int)(t % Hout);
        const int n = (int)(t / Hout);
        const int ymin = bounds[2 * yy], cnt = bounds[2 * yy + 1];
        const uint8_t* p = src + (((size_t)n * Hin + ymin) * W + xx) * C;
        const int32_t* k = kk + (size_t)yy * ksize;
        for (int c = 0; c < C; ++c) {
            int ss = 1 << (PREC - 1);
            for (int y = 0; y < cnt; ++y) ss += (int)p[(size_t)y * W * C + c] * k[y];
            dst[(((size_t)n * C + c) * Hout + yy) * W + xx] = px_out<OUT>(clip8(ss));
        }
    }
}

static inline int grid_for(long long n) {
    long long g = (n + 255) / 256;
    return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

extern "C" {

int pmoe_resample_u8_horizontal(const uint8_t* src, uint8_t* dst, int32_t n_img, int32_t H0, int32_t W0, int32_t row0,
                                int32_t rows, int32_t C, int32_t Wout, const int32_t* bounds, const int32_t* coeffs,
                                int32_t ksize, void* stream) {
    if (!src || !dst || !bounds || !coeffs || n_img < 1 || rows < 1 || row0 < 0 || row0 + rows > H0 || W0 < 1 || Wout < 1 ||
        C < 1 || ksize < 1)
        return PMOE_ERR_ARG;
    hipLaunchKernelGGL(resample_h_kernel, dim3(grid_for((long long)n_img * rows * Wout)), dim3(256), 0, (hipStream_t)stream,
                       src, dst, n_img, H0, W0, row0, rows, C, Wout, bounds, coeffs, ksize);
    return (int)hipGetLastError();
}

int pmoe_resample_u8_vertical_to_f32(const uint8_t* src, float* dst_nchw, int32_t n_img, int32_t Hin, int32_t W, int32_t C,
                                     int32_t Hout, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                                     void* stream) {
    if (!src || !dst_nchw || !bounds || !coeffs || n_img < 1 || Hin < 1 || W < 1 || C < 1 || Hout < 1 || ksize < 1)
        return PMOE_ERR_ARG;
    hipLaunchKernelGGL(resample_v_kernel<float>, dim3(grid_for((long long)n_img * Hout * W)), dim3(256), 0,
                       (hipStream_t)stream, src, dst_nchw, n_img, Hin, W, C, Hout, bounds, coeffs, ksize);
    return (int)hipGetLastError();
}

int pmoe_resample_u8_vertical_to_i64(const uint8_t* src, int64_t* dst_nchw, int32_t n_img, int32_t Hin, int32_t W, int32_t C,
                                     int32_t Hout, const int32_t* bounds, const int32_t* coeffs, int32_t ksize,
                                     void* stream) {
    if (!src || !dst_nchw || !bounds || !coeffs || n_img < 1 || Hin < 1 || W < 1 || C < 1 || Hout < 1 || ksize < 1)
        return PMOE_ERR_ARG;
    hipLaunchKernelGGL(resample_v_kernel<long long>, dim3(grid_for((long long)n_img * Hout * W)), dim3(256), 0,
                       (hipStream_t)stream, src, (long long*)dst_nchw, n_img, Hin, W, C, Hout, bounds, coeffs, ksize);
    return (int)hipGetLastError();
}

}  // extern "C"
