// Shared device helpers for the PMoE gfx950 kernels.  CDNA4 only: wave = 64 lanes, MFMA 32x32,
// 160 KiB LDS per CU.  No CUDA/compat paths.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16;
typedef int v4i __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

#include "../../include/pmoe_hip.h"   // dtype / error / activation codes shared with the C ABI

// OCP e4m3 ("fp8") storage tag: conv weights quantised by pmoe_pack_conv_weights_fp8, activations converted on the way
// into LDS (conv_common.h); gfx950's v_cvt_pk_fp8_f32 / v_mfma_*_fp8_fp8 use the OCP encoding (not MI300's fnuz).
struct fp8 { unsigned char v; };
#define PMOE_FP8_MAX 448.0f

// 16 bf16 (two 16-byte vectors) -> 16 e4m3 bytes: x * scale, clamped to the finite e4m3 range, round to nearest even
__device__ __forceinline__ v4i cvt16_bf16_to_fp8(const v4i& lo, const v4i& hi, float scale) {
    v4i out;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
        const unsigned w0 = (unsigned)(d < 2 ? lo[2 * d] : hi[2 * d - 4]), w1 = (unsigned)(d < 2 ? lo[2 * d + 1] : hi[2 * d - 3]);
        float f0 = __builtin_bit_cast(float, w0 << 16) * scale, f1 = __builtin_bit_cast(float, w0 & 0xffff0000u) * scale;
        float f2 = __builtin_bit_cast(float, w1 << 16) * scale, f3 = __builtin_bit_cast(float, w1 & 0xffff0000u) * scale;
        f0 = fminf(fmaxf(f0, -PMOE_FP8_MAX), PMOE_FP8_MAX); f1 = fminf(fmaxf(f1, -PMOE_FP8_MAX), PMOE_FP8_MAX);
        f2 = fminf(fmaxf(f2, -PMOE_FP8_MAX), PMOE_FP8_MAX); f3 = fminf(fmaxf(f3, -PMOE_FP8_MAX), PMOE_FP8_MAX);
        int pk = __builtin_amdgcn_cvt_pk_fp8_f32(f0, f1, 0, false);
        pk = __builtin_amdgcn_cvt_pk_fp8_f32(f2, f3, pk, true);
        out[d] = pk;
    }
    return out;
}

template <typename T> struct VecOf;                       // 16-byte vector of T
template <> struct VecOf<bf16> { static constexpr int N = 8; };
template <> struct VecOf<float> { static constexpr int N = 4; };

__device__ __forceinline__ float to_f32(bf16 x) { return (float)x; }
__device__ __forceinline__ float to_f32(float x) { return x; }
template <typename T> __device__ __forceinline__ T from_f32(float x);
template <> __device__ __forceinline__ bf16 from_f32<bf16>(float x) { return (bf16)x; }
template <> __device__ __forceinline__ float from_f32<float>(float x) { return x; }

// unpack a 16-byte vector of T into floats / pack floats back
template <typename T> __device__ __forceinline__ void unpack16(const v4i& v, float* f);
template <> __device__ __forceinline__ void unpack16<bf16>(const v4i& v, float* f) {
    bf16x8 b = __builtin_bit_cast(bf16x8, v);
#pragma unroll
    for (int i = 0; i < 8; ++i) f[i] = (float)b[i];
}
template <> __device__ __forceinline__ void unpack16<float>(const v4i& v, float* f) {
    f32x4 b = __builtin_bit_cast(f32x4, v);
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = b[i];
}
template <typename T> __device__ __forceinline__ v4i pack16(const float* f);
template <> __device__ __forceinline__ v4i pack16<bf16>(const float* f) {
    bf16x8 b;
#pragma unroll
    for (int i = 0; i < 8; ++i) b[i] = (bf16)f[i];
    return __builtin_bit_cast(v4i, b);
}
template <> __device__ __forceinline__ v4i pack16<float>(const float* f) {
    f32x4 b;
#pragma unroll
    for (int i = 0; i < 4; ++i) b[i] = f[i];
    return __builtin_bit_cast(v4i, b);
}

__device__ __forceinline__ v4i ldg16(const void* p) { return *reinterpret_cast<const v4i*>(p); }
__device__ __forceinline__ void stg16(void* p, v4i v) { *reinterpret_cast<v4i*>(p) = v; }

// Activations of make_mlp (model/blocks/basics.py:23-28) and their derivatives expressed through the SAVED OUTPUT y = act(z)
// (the backward pass never keeps z): relu' = [y > 0], elu' = y > 0 ? 1 : y + 1, tanh' = 1 - y^2, sigmoid' = y (1 - y).
__device__ __forceinline__ float act_apply(int act, float v) {
    if (act == PMOE_ACT_RELU) return fmaxf(v, 0.f);
    if (act == PMOE_ACT_ELU) return v > 0.f ? v : expm1f(v);
    if (act == PMOE_ACT_TANH) return tanhf(v);
    if (act == PMOE_ACT_SIGMOID) return 1.f / (1.f + expf(-v));
    return v;
}
__device__ __forceinline__ float act_deriv_from_output(int res_mode, float y) {
    if (res_mode == PMOE_RES_DELU) return y > 0.f ? 1.f : y + 1.f;
    if (res_mode == PMOE_RES_DTANH) return 1.f - y * y;
    return y * (1.f - y);                                // PMOE_RES_DSIGMOID
}

// 32-bit mix for the dropout mask (counter based: seed + element index -> uniform [0,1))
__device__ __forceinline__ float hash_uniform(unsigned long long seed, unsigned long long idx) {
    unsigned long long z = seed + idx * 0x9E3779B97F4A7C15ull;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    return (float)(unsigned)(z >> 40) * (1.0f / 16777216.0f);
}

// XCD-aware block remap (bijective for any grid size): blocks that are neighbours in the logical
// tile order land on one XCD (= share its 4 MiB L2) under the observed round-robin dealing.
__device__ __forceinline__ unsigned xcd_remap(unsigned bid, unsigned nblk) {
    const unsigned q = nblk >> 3, r = nblk & 7u, x = bid & 7u, k = bid >> 3;
    return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + k;
}

static inline int ilog2_exact(int v) {
    int l = 0;
    while ((1 << l) < v) ++l;
    return ((1 << l) == v) ? l : -1;
}

// One-time (per device, thread-safe) opt-in of a kernel to more than 64 KiB of dynamic LDS.  The flag is a bit per device
// ordinal: a process that drives several GPUs sets the attribute on each; two racing first calls both set it (idempotent).
#include <atomic>
template <auto KERNEL> static inline hipError_t ensure_dyn_lds(int bytes) {
    static std::atomic<unsigned long long> done{0};
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return e;
    const unsigned long long bit = 1ull << (dev & 63);
    if (!(done.load(std::memory_order_acquire) & bit)) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(KERNEL), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e != hipSuccess) return e;
        done.fetch_or(bit, std::memory_order_release);
    }
    return hipSuccess;
}

#define HIP_RET(expr)                       \
    do {                                    \
        hipError_t _e = (expr);             \
        if (_e != hipSuccess) return (int)_e; \
    } while (0)
