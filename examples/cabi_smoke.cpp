// The C ABI without Python or torch: a foreign host program includes include/pmoe_hip.h, links libpmoe_hip.so, owns
// every buffer (hipMalloc) and the stream, and checks one grouped 3x3 convolution (+ per-expert bias + ReLU epilogue)
// against a plain CPU loop.  Build + run:  tests/test_cabi_gpu.py  (hipcc --offload-arch=gfx950 examples/cabi_smoke.cpp
// -Iinclude -Lpmoe_amd -lpmoe_hip -Wl,-rpath,pmoe_amd).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "pmoe_hip.h"

#define CK(x)                                                                  \
    do {                                                                       \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 2; } \
    } while (0)
#define PK(x)                                                                  \
    do {                                                                       \
        int r_ = (x);                                                          \
        if (r_ != 0) { printf("pmoe error %d (%s) at %s:%d\n", r_, pmoe_error_string(r_), __FILE__, __LINE__); return 3; } \
    } while (0)

static uint16_t f2bf(float f) {                 // round-to-nearest-even bf16
    uint32_t u;
    memcpy(&u, &f, 4);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (uint16_t)(u >> 16);
}
static float bf2f(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}

int main() {
    const int E = 2, B = 3, H = 9, W = 11, Cin = 16, Cout = 64, KS = 3, N = E * B;
    if (pmoe_abi_sizeof(0) != (int)sizeof(pmoe_conv_desc)) { printf("struct layout mismatch\n"); return 1; }
    srand(7);
    auto rnd = []() { return (float)rand() / (float)RAND_MAX * 2.f - 1.f; };
    // activations NHWC bf16, experts folded into the image index (image n belongs to expert n / B)
    std::vector<uint16_t> x((size_t)N * H * W * Cin);
    for (auto& v : x) v = f2bf(rnd());
    // per-expert parameters in the reference's layout: f32 OIHW + bias
    std::vector<std::vector<float>> w(E, std::vector<float>((size_t)Cout * Cin * KS * KS)), bias(E, std::vector<float>(Cout));
    for (int e = 0; e < E; ++e) {
        for (auto& v : w[e]) v = bf2f(f2bf(rnd() * 0.1f));          // bf16-representable so the pack is exact
        for (auto& v : bias[e]) v = rnd();
    }
    hipStream_t st;
    CK(hipStreamCreate(&st));
    void *dx, *dy, *dwf, *dbias;
    float* dw[2];
    float* db[2];
    void **dwtab, **dbtab;
    CK(hipMalloc(&dx, x.size() * 2));
    CK(hipMalloc(&dy, (size_t)N * H * W * Cout * 2));
    CK(hipMalloc(&dwf, (size_t)E * Cout * KS * KS * Cin * 2));
    CK(hipMalloc(&dbias, (size_t)E * Cout * 4));
    CK(hipMalloc(&dwtab, E * sizeof(void*)));
    CK(hipMalloc(&dbtab, E * sizeof(void*)));
    for (int e = 0; e < E; ++e) {
        CK(hipMalloc(&dw[e], w[e].size() * 4));
        CK(hipMalloc(&db[e], Cout * 4));
        CK(hipMemcpy(dw[e], w[e].data(), w[e].size() * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(db[e], bias[e].data(), Cout * 4, hipMemcpyHostToDevice));
    }
    CK(hipMemcpy(dwtab, dw, E * sizeof(void*), hipMemcpyHostToDevice));
    CK(hipMemcpy(dbtab, db, E * sizeof(void*), hipMemcpyHostToDevice));
    CK(hipMemcpy(dx, x.data(), x.size() * 2, hipMemcpyHostToDevice));
    // one launch repacks all experts' filters to the grouped kernel layout, one the biases
    PK(pmoe_pack_conv_weights((const void* const*)dwtab, dwf, nullptr, E, Cout, Cin, KS, 64, Cin, 64, 64, PMOE_DT_BF16, st));
    PK(pmoe_pack_bias((const void* const*)dbtab, (float*)dbias, E, Cout, 64, st));
    pmoe_conv_desc d;
    memset(&d, 0, sizeof d);
    d.in = dx; d.w = dwf; d.out = dy; d.bias = (const float*)dbias;
    d.n = N; d.h = H; d.w_ = W; d.cin = Cin; d.ho = H; d.wo = W; d.cout = Cout; d.coutp = 64;
    d.in_ld = Cin; d.out_ld = Cout; d.ipe = B; d.ks = KS; d.stride = 1; d.pad = 1;
    d.act = PMOE_ACT_RELU; d.dtype = PMOE_DT_BF16;
    printf("kernel plan code: %d\n", pmoe_conv2d_plan(&d));
    PK(pmoe_conv2d_igemm(&d, st));
    CK(hipStreamSynchronize(st));
    std::vector<uint16_t> y((size_t)N * H * W * Cout);
    CK(hipMemcpy(y.data(), dy, y.size() * 2, hipMemcpyDeviceToHost));
    double worst = 0;
    for (int n = 0; n < N; ++n)
        for (int oy = 0; oy < H; ++oy)
            for (int ox = 0; ox < W; ++ox)
                for (int co = 0; co < Cout; ++co) {
                    const int e = n / B;
                    double s = bias[e][co];
                    for (int ky = 0; ky < KS; ++ky)
                        for (int kx = 0; kx < KS; ++kx) {
                            const int iy = oy + ky - 1, ix = ox + kx - 1;
                            if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
                            for (int ci = 0; ci < Cin; ++ci)
                                s += (double)bf2f(x[(((size_t)n * H + iy) * W + ix) * Cin + ci]) *
                                     w[e][(((size_t)co * Cin + ci) * KS + ky) * KS + kx];
                        }
                    if (s < 0) s = 0;
                    const double got = bf2f(y[(((size_t)n * H + oy) * W + ox) * Cout + co]);
                    const double err = fabs(got - s) / (1.0 + fabs(s));
                    if (err > worst) worst = err;
                }
    printf("worst |got-ref|/(1+|ref|) = %.3e\n", worst);
    if (worst > 1e-2) { printf("FAIL\n"); return 4; }
    printf("CABI OK\n");
    return 0;
}
