// Probe for the next conv-tile design ("what is left"): LDS-DMA on gfx950 with this toolchain.
//   global_load_lds_dwordx4 : per-lane global address, LDS destination = wave-uniform base + lane * 16
//   buffer_load_dwordx4 ... offen lds : the same through a buffer resource -- lanes whose offset is beyond num_records
//                                       must deliver ZEROS (the zero fill of a halo patch without a VGPR round trip)
// Build + run on the GPU box:  hipcc --offload-arch=gfx950 -O3 tools/lds_dma_probe.hip -o build/lds_dma_probe && build/lds_dma_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef int v4i __attribute__((ext_vector_type(4)));

__global__ void via_global(const char* src, char* dst) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + threadIdx.x * 16),
                                     (__attribute__((address_space(3))) void*)(smem + (threadIdx.x / 64) * 1024), 16, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    *reinterpret_cast<v4i*>(dst + threadIdx.x * 16) = *reinterpret_cast<v4i*>(smem + threadIdx.x * 16);
}

__global__ void via_buffer(const char* src, char* dst, int nbytes) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 4096 / 4; i += blockDim.x) reinterpret_cast<int*>(smem)[i] = 0x55555555;   // poison
    __syncthreads();
    __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)src, (short)0, nbytes, 0x00020000);
    const int voff = (threadIdx.x & 1) ? 0x40000000 : (int)(threadIdx.x * 16);      // odd lanes: out of range
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(smem + (threadIdx.x / 64) * 1024),
                                             16, voff, 0, 0, 0);
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    *reinterpret_cast<v4i*>(dst + threadIdx.x * 16) = *reinterpret_cast<v4i*>(smem + threadIdx.x * 16);
}

int main() {
    const int n = 256 * 16;
    char *h = (char*)malloc(n), *o = (char*)malloc(n), *ds, *dd;
    for (int i = 0; i < n; ++i) h[i] = (char)(i * 7 + 1);
    hipMalloc(&ds, n); hipMalloc(&dd, n);
    hipMemcpy(ds, h, n, hipMemcpyHostToDevice);
    int bad = 0;
    hipLaunchKernelGGL(via_global, dim3(1), dim3(256), 4096, 0, ds, dd);
    hipMemcpy(o, dd, n, hipMemcpyDeviceToHost);
    for (int i = 0; i < n; ++i) bad += o[i] != h[i];
    printf("global_load_lds_dwordx4: %s (%d mismatching bytes)\n", bad ? "FAIL" : "ok", bad);
    hipLaunchKernelGGL(via_buffer, dim3(1), dim3(256), 4096, 0, ds, dd, n);
    hipMemcpy(o, dd, n, hipMemcpyDeviceToHost);
    int bad_even = 0, nonzero_odd = 0, poison_odd = 0;
    for (int t = 0; t < 256; ++t)
        for (int b = 0; b < 16; ++b) {
            const char v = o[t * 16 + b];
            if (t & 1) { nonzero_odd += v != 0; poison_odd += v == 0x55; }
            else bad_even += v != h[t * 16 + b];
        }
    printf("buffer_load_dwordx4 ... lds: in-range lanes %s (%d bad bytes); out-of-range lanes: %d non-zero bytes (%d still poison)\n",
           bad_even ? "FAIL" : "ok", bad_even, nonzero_odd, poison_odd);
    printf("=> out-of-range lanes %s\n", nonzero_odd == 0 ? "are ZERO-FILLED in LDS" : (poison_odd ? "leave LDS untouched" : "write garbage"));
    return bad || bad_even;
}
