// Fused optimizer tail of the stage-2 step (reference trainer/train_2.py:157-165,184):
//   clip_grad_norm_(params, max_norm)  ->  global L2 norm over ~620 gradient tensors + one scale
//   Adam(amsgrad=True).step()          ->  one multi-tensor update
//   AveragedModel.update_parameters()  ->  one multi-tensor running mean
// The reference issues a handful of launches (and a .item() sync) PER parameter tensor; here each stage is one launch
// over a chunk table: tensor t is cut into chunks of PMOE_OPT_CHUNK elements, workgroup i handles chunk
// (chunk_tensor[i], chunk_index[i]).  All of it is HBM-bound streaming work: 16-byte accesses when the chunk base is
// aligned (always, for torch allocations), scalar tail otherwise.  Nothing synchronises with the host: the clip
// coefficient stays in device memory and is consumed by the Adam kernel.
#include "common.h"

static constexpr int CHUNK = PMOE_OPT_CHUNK;

__device__ __forceinline__ float block_sum(float v, float* red) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float s = 0.f;
    if (threadIdx.x == 0)
        for (int w = 0; w < (int)(blockDim.x >> 6); ++w) s += red[w];
    return s;      // valid in thread 0
}

__global__ void __launch_bounds__(256) mt_sqsum_kernel(const pmoe_opt_tensor* __restrict__ tab,
                                                      const int32_t* __restrict__ chunk_tensor,
                                                      const int32_t* __restrict__ chunk_index,
                                                      float* __restrict__ partial) {
    const pmoe_opt_tensor t = tab[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_index[blockIdx.x] * CHUNK;
    long long n = t.numel - base;
    if (n > CHUNK) n = CHUNK;
    const float* g = t.grad + base;
    float s = 0.f;
    if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
        const long long nv = n >> 2;
        for (long long i = threadIdx.x; i < nv; i += 256) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(g + 4 * i);
            s += v[0] * v[0] + v[1] * v[1] + v[2] * v[2] + v[3] * v[3];
        }
        for (long long i = (nv << 2) + threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    } else {
        for (long long i = threadIdx.x; i < n; i += 256) s += g[i] * g[i];
    }
    __shared__ float red[4];
    const float tot = block_sum(s, red);
    if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// norm[0] = sqrt(sum partial) ; norm[1] = clip coefficient min(1, max_norm / (norm + 1e-6))  (torch clip_grad_norm_)
__global__ void __launch_bounds__(256) mt_norm_finish_kernel(const float* __restrict__ partial, int n, float max_norm,
                                                            float* __restrict__ norm) {
    double s = 0.0;            // fixed order, double accumulation: deterministic and exact enough for 1e-6 parity
    for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
    __shared__ double red[256];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int k = 128; k > 0; k >>= 1) {
        if (threadIdx.x < k) red[threadIdx.x] += red[threadIdx.x + k];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const float nr = (float)sqrt(red[0]);
        norm[0] = nr;
        const float c = max_norm > 0.f ? max_norm / (nr + 1e-6f) : 1.f;
        norm[1] = c < 1.f ? c : 1.f;
    }
}

__global__ void __launch_bounds__(256) mt_scale_kernel(const pmoe_opt_tensor* __restrict__ tab,
                                                      const int32_t* __restrict__ chunk_tensor,
                                                      const int32_t* __restrict__ chunk_index,
                                                      const float* __restrict__ norm) {
    const float c = norm[1];
    if (c >= 1.f) return;
    const pmoe_opt_tensor t = tab[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_index[blockIdx.x] * CHUNK;
    long long n = t.numel - base;
    if (n > CHUNK) n = CHUNK;
    float* g = const_cast<float*>(t.grad) + base;
    for (long long i = threadIdx.x; i < n; i += 256) g[i] *= c;
}

// torch.optim.Adam (single-tensor formulas of torch/optim/adam.py, maximize=False, capturable=False):
//   g' = clip * g (+ wd * p);  m = m + (1-b1)(g' - m);  v = b2 v + (1-b2) g'^2;  vmax = max(vmax, v)
//   p -= (lr / bc1) * m / (sqrt(vmax or v) / sqrt(bc2) + eps)
__global__ void __launch_bounds__(256) mt_adam_kernel(const pmoe_opt_tensor* __restrict__ tab,
                                                     const int32_t* __restrict__ chunk_tensor,
                                                     const int32_t* __restrict__ chunk_index, float lr, float beta1,
                                                     float beta2, float eps, float weight_decay, int amsgrad,
                                                     float bc1_all, float bc2s_all, const float* __restrict__ norm) {
    const pmoe_opt_tensor t = tab[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_index[blockIdx.x] * CHUNK;
    long long n = t.numel - base;
    if (n > CHUNK) n = CHUNK;
    const float clip = norm ? norm[1] : 1.f;
    // bias corrections: one value for all tensors (kernel argument, > 0) or the per-tensor entries of the table
    const float step_size = lr / (bc1_all > 0.f ? bc1_all : t.bc1);
    const float inv_bc2s = 1.f / (bc2s_all > 0.f ? bc2s_all : t.bc2_sqrt);
    float* p = t.param + base;
    const float* g = t.grad + base;
    float* m = t.exp_avg + base;
    float* v = t.exp_avg_sq + base;
    float* vm = amsgrad ? t.max_exp_avg_sq + base : nullptr;
    auto upd = [&](float gi, float pi, float& mi, float& vi, float& mx) -> float {
        gi *= clip;
        if (weight_decay != 0.f) gi += weight_decay * pi;
        mi = mi + (1.f - beta1) * (gi - mi);
        vi = vi * beta2 + (1.f - beta2) * gi * gi;
        float d = vi;
        if (amsgrad) {
            mx = fmaxf(mx, vi);
            d = mx;
        }
        return pi - step_size * (mi / (sqrtf(d) * inv_bc2s + eps));
    };
    const bool aligned = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                           reinterpret_cast<uintptr_t>(v) | reinterpret_cast<uintptr_t>(vm)) & 15) == 0;
    long long done = 0;
    if (aligned) {
        const long long nv = n >> 2;
        for (long long i = threadIdx.x; i < nv; i += 256) {
            f32x4 pv = *reinterpret_cast<f32x4*>(p + 4 * i), mv = *reinterpret_cast<f32x4*>(m + 4 * i);
            f32x4 vv = *reinterpret_cast<f32x4*>(v + 4 * i);
            const f32x4 gv = *reinterpret_cast<const f32x4*>(g + 4 * i);
            f32x4 xv = amsgrad ? *reinterpret_cast<f32x4*>(vm + 4 * i) : vv;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float mi = mv[k], vi = vv[k], mx = xv[k];
                pv[k] = upd(gv[k], pv[k], mi, vi, mx);
                mv[k] = mi; vv[k] = vi; xv[k] = mx;
            }
            *reinterpret_cast<f32x4*>(p + 4 * i) = pv;
            *reinterpret_cast<f32x4*>(m + 4 * i) = mv;
            *reinterpret_cast<f32x4*>(v + 4 * i) = vv;
            if (amsgrad) *reinterpret_cast<f32x4*>(vm + 4 * i) = xv;
        }
        done = nv << 2;
    }
    for (long long i = done + threadIdx.x; i < n; i += 256) {
        float mi = m[i], vi = v[i], mx = amsgrad ? vm[i] : 0.f;
        p[i] = upd(g[i], p[i], mi, vi, mx);
        m[i] = mi;
        v[i] = vi;
        if (amsgrad) vm[i] = mx;
    }
}

// AveragedModel.update_parameters (torch/optim/swa_utils.py): first call copies, later p_avg += (p - p_avg) / (n + 1)
__global__ void __launch_bounds__(256) mt_swa_kernel(const pmoe_opt_tensor* __restrict__ tab,
                                                    const int32_t* __restrict__ chunk_tensor,
                                                    const int32_t* __restrict__ chunk_index, float inv_np1) {
    const pmoe_opt_tensor t = tab[chunk_tensor[blockIdx.x]];
    const long long base = (long long)chunk_index[blockIdx.x] * CHUNK;
    long long n = t.numel - base;
    if (n > CHUNK) n = CHUNK;
    const float* p = t.param + base;
    float* a = t.swa + base;
    for (long long i = threadIdx.x; i < n; i += 256) {
        const float ai = a[i];
        a[i] = inv_np1 >= 1.f ? p[i] : ai + (p[i] - ai) * inv_np1;
    }
}

extern "C" {

int pmoe_mt_grad_norm(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                      int32_t n_chunks, float max_norm, float* partial, float* norm, int32_t scale_grads, void* stream) {
    if (n_chunks < 1 || !table || !partial || !norm) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(mt_sqsum_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_index,
                       partial);
    hipLaunchKernelGGL(mt_norm_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, n_chunks, max_norm, norm);
    if (scale_grads)
        hipLaunchKernelGGL(mt_scale_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor,
                           chunk_index, norm);
    return (int)hipGetLastError();
}

int pmoe_mt_adam(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index, int32_t n_chunks,
                 float lr, float beta1, float beta2, float eps, float weight_decay, int32_t amsgrad, float bc1_all,
                 float bc2_sqrt_all, const float* norm, void* stream) {
    if (n_chunks < 1 || !table) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(mt_adam_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_index,
                       lr, beta1, beta2, eps, weight_decay, amsgrad, bc1_all, bc2_sqrt_all, norm);
    return (int)hipGetLastError();
}

int pmoe_mt_swa_update(const pmoe_opt_tensor* table, const int32_t* chunk_tensor, const int32_t* chunk_index,
                       int32_t n_chunks, int64_t n_averaged, void* stream) {
    if (n_chunks < 1 || !table || n_averaged < 0) return PMOE_ERR_ARG;
    hipLaunchKernelGGL(mt_swa_kernel, dim3(n_chunks), dim3(256), 0, (hipStream_t)stream, table, chunk_tensor, chunk_index,
                       1.f / (float)(n_averaged + 1));
    return (int)hipGetLastError();
}

}  // extern "C"
