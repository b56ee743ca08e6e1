// Probe (GPU box, `hipcc --offload-arch=gfx950 tools/probe_mfma_f8.hip -o /tmp/probe && /tmp/probe`): operand layout and scale
// encoding of v_mfma_scale_f32_32x32x64_f8f6f4 with e4m3 operands, checked with exact small values against a host product.
//   D[32][32] = A[32][64] * B[64][32]; lane l: row / column r = l & 31, half h = l >> 5; 32 bytes of A row r and of B column r.
// Hypotheses for WHICH 32 of the 64 k's lane half h holds (byte j of the 8 VGPRs):
//   H1: k = 32 h + j                      (contiguous halves)
//   H2: k = 16 h + (j & 15) + 32 (j >> 4) (two interleaved 32-deep steps)
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <math.h>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__global__ void probe(const uint8_t* A, const uint8_t* Bt, float* D, int hyp, int scale) {
    const int l = threadIdx.x, r = l & 31, h = l >> 5;
    union { v8i v; uint8_t b[32]; } a, b;
    for (int j = 0; j < 32; ++j) {
        const int k = hyp == 1 ? 32 * h + j : 16 * h + (j & 15) + 32 * (j >> 4);
        a.b[j] = A[r * 64 + k];
        b.b[j] = Bt[r * 64 + k];
    }
    f32x16 c;
    for (int i = 0; i < 16; ++i) c[i] = 0.f;
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a.v, b.v, c, 0, 0, 0, scale, 0, scale);
    for (int i = 0; i < 16; ++i) D[((i & 3) + 8 * (i >> 2) + 4 * h) * 32 + r] = c[i];
}

static float e4m3_to_f(uint8_t v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float f = e == 0 ? ldexpf(m / 8.f, -6) : ldexpf(1.f + m / 8.f, e - 7);
    return s ? -f : f;
}

int main() {
    // e4m3 codes of exactly representable small values
    const uint8_t codes[] = {0x00, 0x38, 0xB8, 0x40, 0xC0, 0x30, 0xB0, 0x44};   // 0, 1, -1, 2, -2, 0.5, -0.5, 3
    uint8_t hA[32 * 64], hBt[32 * 64];
    srand(3);
    for (int i = 0; i < 32 * 64; ++i) { hA[i] = codes[rand() % 8]; hBt[i] = codes[rand() % 8]; }
    float ref[32 * 32];
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            float s = 0;
            for (int k = 0; k < 64; ++k) s += e4m3_to_f(hA[i * 64 + k]) * e4m3_to_f(hBt[j * 64 + k]);
            ref[i * 32 + j] = s;
        }
    uint8_t *dA, *dB; float* dD;
    hipMalloc(&dA, sizeof hA); hipMalloc(&dB, sizeof hBt); hipMalloc(&dD, sizeof ref);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hBt, sizeof hBt, hipMemcpyHostToDevice);
    const int scales[] = {0x7F7F7F7F, 0, 0x7F, (int)0x80808080u};
    for (int hyp = 1; hyp <= 2; ++hyp)
        for (int si = 0; si < 4; ++si) {
            float out[32 * 32];
            hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dD, hyp, scales[si]);
            hipMemcpy(out, dD, sizeof out, hipMemcpyDeviceToHost);
            double err = 0, ratio = 0; int n = 0;
            for (int i = 0; i < 1024; ++i) { err = fmax(err, fabs(out[i] - ref[i])); if (fabs(ref[i]) > 1) { ratio += out[i] / ref[i]; ++n; } }
            printf("hypothesis %d scale 0x%08x: max |D - ref| = %g, mean D/ref = %g\n", hyp, scales[si], err, ratio / (n ? n : 1));
        }
    return 0;
}
