// 1x1 convolutions over many pixels, stride 1 or 2, bf16: the three downsample projections of ResNet18 (layer2-4 `.0.downsample`,
// torchvision BasicBlock; stride 2) and the U-Net's transposed-convolution GEMMs (blocks/unet.py: ConvTranspose2d(k=2, s=2) = a 1x1
// convolution into 4*Cout channels + a pixel shuffle).  K = Cin = 64..1024: HBM-bound layers (intensity 40..170 FLOP/B) that the
// LDS-tiled implicit-GEMM kernels ran at 0.6-1.4 TB/s because their per-tile staging, barriers and epilogue are sized for 3x3 taps.
//
// Direct form (as conv_c16.hip): a wave owns 32 consecutive output pixels (flat index over image, row, column); their input rows
// ARE the B operand of v_mfma_f32_32x32x16_bf16 (lane = pixel, lanes 32..63 = channels 8..15 of the 16-channel k-step), loaded
// straight from global memory by buffer loads (the stride is folded into the per-lane offset; pixels past the end are parked out
// of range and read as zeros); the filter slab of the workgroup's MT*32 output channels sits in LDS in A-operand order; the output
// tile is transposed through wave-private LDS so that every store writes whole pixel rows; the accumulators start from the bias; BatchNorm
// partial sums (of the rounded outputs) from the read-back, one [2][CoutP] row per workgroup, fixed order.
// K is walked in chunks of 128 channels; three chunk buffers rotate so that two chunks of loads are in flight under the MFMAs.
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MT: 32-channel output tiles per workgroup slab (4: 128 channels when Cin <= 128; 2: 64 channels for Cin up to 512 -- the slab's
// filters sit in LDS next to the transpose tiles: MT * Cin / 16 KiB; 64-80 KB in all: two workgroups per CU except at Cin = 512)
// INBN (round 4, BASELINE config 4; PMOE_RES_INBN): the input is the pre-activation z of a BatchNorm + ReLU whose output has this launch
// as its only consumer -- the U-Nets' ConvTranspose2d layers and final 1x1 classifier behind a frozen, train-mode block
// (model/blocks/unet.py:28-45,64) -- so the pmoe_bn_apply pass in front of it disappears: the B operand, which this kernel loads
// straight into registers, becomes bf16(max((z - mean) * scale + shift, 0)) there (bn_apply_kernel's arithmetic and rounding:
// bit-identical to the pair), 28 VALU per 16-byte piece in an HBM-bound kernel.  Coefficient rows [mean | scale | shift][Cin] in LDS.
template <int MT, bool INBN = false>
__global__ void __launch_bounds__(256, 2) conv1x1_direct_kernel(ConvArgs a, int tiles_per_expert, int wgs_per_expert, int n_slabs,
                                                               unsigned in_bytes, unsigned out_bytes, int kchunks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int SLAB = MT * 32, CHUNKS = SLAB / 8;               // output channels / 16-byte pieces per output pixel row of the slab
    const int ksteps = kchunks * 8;                                // 16-channel k-steps, padded to whole 128-channel chunks
    v4i* wl = reinterpret_cast<v4i*>(smem);                        // [MT][ksteps][64 lanes]: the A operand of each MFMA
    v4i* tbuf = wl + MT * ksteps * 64;                             // [4 waves][32 pixels][CHUNKS], chunk ^= pixel & (CHUNKS - 1)
    float* red = reinterpret_cast<float*>(tbuf + 4 * 32 * CHUNKS); // [4 waves][64 / CHUNKS lane groups][2][SLAB]
    float* lbias = red + 4 * (64 / CHUNKS) * 2 * SLAB;             // [SLAB]: the accumulators start from the bias (exact: f32, before rounding)
    float* lcoef = lbias + SLAB;                                   // INBN: [3][kchunks * 128]: mean, gamma * invstd, beta (zeros beyond Cin)
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.y;
    const int slab = blockIdx.x % n_slabs, wg = blockIdx.x / n_slabs;
    const int cout0 = slab * SLAB;
    {
        const bf16* w = reinterpret_cast<const bf16*>(a.w) + ((size_t)e * a.CoutP + cout0) * a.Cin;
        for (int idx = tid; idx < MT * ksteps * 64; idx += 256) {
            const int l = idx & 63, ks = (idx >> 6) % ksteps, mt = (idx >> 6) / ksteps;
            const int m = l & 31, kh = l >> 5;
            const int aa = m >> 3, hh = (m >> 2) & 1, b = m & 3;                       // D row m lands in lane half hh, register 4*aa + b
            const int co = 32 * mt + 16 * (aa >> 1) + 8 * hh + 4 * (aa & 1) + b;       // so that a lane's registers are 8-channel groups
            const int ci = ks * 16 + kh * 8;
            wl[idx] = (ci < a.Cin && cout0 + co < a.CoutP) ? ldg16(w + (size_t)co * a.Cin + ci) : v4i{0, 0, 0, 0};
        }
    }
    if (tid < SLAB) lbias[tid] = (a.bias && cout0 + tid < a.Cout) ? a.bias[(size_t)e * a.CoutP + cout0 + tid] : 0.f;
    if constexpr (INBN) {
        const int nset = a.N / a.bn_ipe, set = (e * a.ipe) / a.bn_ipe, cpad = kchunks * 128;
        for (int i = tid; i < 3 * cpad; i += 256) {
            const int k = i / cpad, c = i - k * cpad;
            lcoef[i] = c < a.Cin ? a.bn[((size_t)(k == 0 ? 0 : k + 1) * nset + set) * a.Cin + c] : 0.f;
        }
    }
    __syncthreads();
    const size_t in_img0 = a.in_shared ? 0 : (size_t)e * a.ipe, out_img0 = (size_t)e * a.ipe;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const bf16*>(a.in) + in_img0 * a.H * a.W * a.in_ld + a.in_coff), (short)0, (int)in_bytes, 0x00020000);
    // shuf_c: the destination is the [ipe][2 Ho][2 Wo][out_ld] concatenation buffer (4x the pixels), addressed per store below
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<bf16*>(a.out) + out_img0 * a.Ho * a.Wo * a.out_ld * (a.shuf_c ? 4 : 1) + a.out_coff + (a.shuf_c ? 0 : cout0)),
        (short)0, (int)(out_bytes - (a.shuf_c ? 0u : (unsigned)cout0 * 2u)), 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const int px_l = lane & 31, kh = lane >> 5;
    const unsigned old2 = (unsigned)a.out_ld * 2u;
    const int npix = a.ipe * a.Ho * a.Wo;                          // output pixels of one expert
    const int t0 = (int)((long long)tiles_per_expert * wg / wgs_per_expert);
    const int t1 = (int)((long long)tiles_per_expert * (wg + 1) / wgs_per_expert);
    // read-back mapping: 64 lanes = (64 / CHUNKS) pixels x CHUNKS pieces per pass; the piece (lane % CHUNKS) is fixed per lane
    constexpr int PPP = 64 / CHUNKS, PASSES = 32 / PPP;
    const int rb_c = lane % CHUNKS, rb_p = lane / CHUNKS;
    const bool chunk_ok = cout0 + rb_c * 8 < a.Cout;
    // shuf_c: this lane's 8 output channels belong to one (dy, dx) of the 2x2 scatter and to channels [shuf_cc, +8) of the destination
    const int shuf_q = a.shuf_c ? (cout0 + rb_c * 8) / a.shuf_c : 0;
    const int shuf_cc = a.shuf_c ? (cout0 + rb_c * 8) - shuf_q * a.shuf_c : 0;
    const int lWo = a.shuf_c ? __builtin_ctz(a.Wo) : 0, lHo = a.shuf_c ? __builtin_ctz(a.Ho) : 0;      // powers of two (conv_c1x1_plan)
    f32x2 s1[4], s2[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] = s2[j] = f32x2{0.f, 0.f};

    // per-lane byte offset of output pixel (tile t, lane) in the input: stride folded in, 0xfffffff0 past the expert's last pixel
    auto in_off = [&](int t) -> unsigned {
        const int p = t * 32 + px_l;
        if (p >= npix) return OOB;
        const int ox = p % a.Wo, r = p / a.Wo, oy = r % a.Ho, img = r / a.Ho;
        return (unsigned)(((img * a.H + oy * a.stride) * a.W + ox * a.stride) * a.in_ld) * 2u + (unsigned)kh * 16u;
    };
    // unit u = (tile, k chunk): 8 k-steps of 16 channels
    auto load_unit = [&](int t, int kc, v4u (&raw)[8], bool live) {
        const unsigned vo = live ? in_off(t) : OOB;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int ci = (kc * 8 + j) * 16;
            raw[j] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(ci < a.Cin ? vo : OOB), ci * 2, 0);
        }
    };
    f32x16 acc[MT];
    auto mfma_unit = [&](int kc, const v4u (&raw)[8]) {
        if (kc == 0) {                                               // acc[mt][8g + i] is channel 32 mt + 16 g + 8 kh + i of the slab
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
#pragma unroll
                for (int i = 0; i < 16; ++i) acc[mt][i] = lbias[32 * mt + 16 * (i >> 3) + 8 * kh + (i & 7)];
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            v4u b = raw[j];
            if constexpr (INBN) {
                // this lane's 8 channels of the k-step: (kc * 8 + j) * 16 + kh * 8 .. + 7 (pixels past the end / channels past Cin are
                // zeros times zero weights or unstored rows: whatever relu(bn(0)) is there does not reach an output)
                typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
                typedef short s16x2 __attribute__((ext_vector_type(2)));
                const int cpad = kchunks * 128;
                const float* cm = lcoef + (kc * 8 + j) * 16 + kh * 8;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 mu = *reinterpret_cast<const f32x2*>(cm + 2 * q);
                    const f32x2 sc = *reinterpret_cast<const f32x2*>(cm + cpad + 2 * q);
                    const f32x2 sh = *reinterpret_cast<const f32x2*>(cm + 2 * cpad + 2 * q);
                    const f32x2 z = f32x2{__builtin_bit_cast(float, b[q] << 16), __builtin_bit_cast(float, b[q] & 0xffff0000u)};
                    const f32x2 v = __builtin_elementwise_fma(z - mu, sc, sh);                 // (bn_apply_kernel: centred, one fma)
                    const unsigned pk = __builtin_bit_cast(unsigned, bf16x2{(bf16)v[0], (bf16)v[1]});
                    b[q] = __builtin_bit_cast(unsigned, __builtin_elementwise_max(__builtin_bit_cast(s16x2, pk), s16x2{0, 0}));
                }
            }
#pragma unroll
            for (int mt = 0; mt < MT; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[(mt * ksteps + kc * 8 + j) * 64 + lane]),
                                                                  __builtin_bit_cast(bf16x8, b), acc[mt], 0, 0, 0);
        }
    };
    auto epilogue = [&](int t) {
        v4i* tb = tbuf + wave * 32 * CHUNKS;
#pragma unroll
        for (int mt = 0; mt < MT; ++mt)
#pragma unroll
            for (int g = 0; g < 2; ++g) {
                float v[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) v[i] = acc[mt][8 * g + i];
                const int c = 4 * mt + 2 * g + kh;                                     // 16-byte piece of the pixel's slab row
                tb[px_l * CHUNKS + (c ^ (px_l & (CHUNKS - 1)))] = pack16<bf16>(v);
            }
        const unsigned osoff = (unsigned)t * 32u * old2;
#pragma unroll
        for (int i = 0; i < PASSES; ++i) {
            const int p = i * PPP + rb_p;
            const v4i v = tb[p * CHUNKS + (rb_c ^ (p & (CHUNKS - 1)))];
            const bool ok = t * 32 + p < npix;
            if (a.shuf_c) {
                // pixel (img, oy, ox) of the low-resolution map -> (img, 2 oy + dy, 2 ox + dx) of the destination
                const int P = t * 32 + p;
                const int ox = P & (a.Wo - 1), oy = (P >> lWo) & (a.Ho - 1), img = P >> (lWo + lHo);
                const unsigned dpix = (unsigned)((img * 2 * a.Ho + 2 * oy + (shuf_q >> 1)) * 2 * a.Wo + 2 * ox + (shuf_q & 1));
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs_out,
                                                       (int)(ok && chunk_ok ? dpix * old2 + (unsigned)shuf_cc * 2u : OOB), 0, 0);
            } else
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs_out,
                                                   (int)(ok && chunk_ok ? (unsigned)p * old2 + (unsigned)rb_c * 16u : OOB), (int)osoff, 0);
            if (a.stats && ok) {
                float rr[8];
                unpack16<bf16>(v, rr);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x2 r2 = f32x2{rr[2 * q], rr[2 * q + 1]};
                    s1[q] += r2;
                    s2[q] = __builtin_elementwise_fma(r2, r2, s2[q]);
                }
            }
        }
    };
    // units of this wave in order: tiles t0 + wave, + 4, ...; k chunks innermost.  Three buffers in rotation (loads are never
    // inside a branch: see conv_c16.hip)
    int lt = t0 + wave, lk = 0;                                    // next unit to load
    int ct = t0 + wave, ck = 0;                                    // next unit to compute
    v4u b0[8], b1[8], b2[8];
    auto fetch = [&](v4u (&buf)[8]) {
        load_unit(lt, lk, buf, lt < t1);
        if (++lk == kchunks) { lk = 0; lt += 4; }
    };
    auto compute = [&](const v4u (&buf)[8]) {
        mfma_unit(ck, buf);
        if (++ck == kchunks) { epilogue(ct); ck = 0; ct += 4; }
    };
    fetch(b0);
    fetch(b1);
    while (ct < t1) {
        fetch(b2);
        compute(b0);
        if (ct >= t1) break;
        fetch(b0);
        compute(b1);
        if (ct >= t1) break;
        fetch(b1);
        compute(b2);
    }
    if (!a.stats) return;
    // fold: (64 / CHUNKS) lane groups x 4 waves hold the partial sums of each channel; fixed order
    __syncthreads();
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        red[((wave * PPP + rb_p) * 2 + 0) * SLAB + rb_c * 8 + q] = s1[q >> 1][q & 1];
        red[((wave * PPP + rb_p) * 2 + 1) * SLAB + rb_c * 8 + q] = s2[q >> 1][q & 1];
    }
    __syncthreads();
    for (int i = tid; i < 2 * SLAB; i += 256) {
        const int which = i / SLAB, c = i % SLAB;
        float s = 0.f;
        for (int g = 0; g < 4 * PPP; ++g) s += red[(g * 2 + which) * SLAB + c];
        if (cout0 + c < a.CoutP) a.stats[(((size_t)e * wgs_per_expert + wg) * 2 + which) * a.CoutP + cout0 + c] = s;
    }
}

}  // namespace

// PMOE_CONV_C1X1=0: A/B switch back to the LDS-tiled kernels (read per launch)
bool conv_c1x1_plan(const ConvArgs& a, int dtype, int* wgs_per_expert, int* tiles_per_expert, int* n_slabs, int* mt, size_t* smem) {
    const char* ev = getenv("PMOE_CONV_C1X1");
    if (ev && !atoi(ev)) return false;
    if (dtype != PMOE_DT_BF16 || a.w_fp8 || a.ks != 1 || a.pad != 0 || a.dilate || a.kh != 1 || a.kw != 1) return false;
    if (a.stride != 1 && a.stride != 2) return false;
    if (a.use_tapmap || a.out_step != 1) return false;
    if (a.shuf_c && (a.stride != 1 || a.stats || a.Cout != 4 * a.shuf_c || a.shuf_c % 8 || (a.Ho & (a.Ho - 1)) || (a.Wo & (a.Wo - 1)) ||
                     (long long)a.ipe * a.Ho * a.Wo * 4 * a.out_ld * 2 >= 0xfff00000ll)) return false;
    if (a.Cin % 16 || a.Cin < 64 || a.Cin > 512 || a.CoutP % 64 || a.Cout % 8) return false;
    const bool inbn = a.res_mode == PMOE_RES_INBN;
    if (a.act != PMOE_ACT_NONE || (a.res_mode != PMOE_RES_NONE && !inbn) || a.drop_p > 0.f) return false;
    if (inbn && (!a.bn || a.in_shared || a.stride != 1)) return false;
    if (a.N % a.ipe || a.Ho != (a.H - 1) / a.stride + 1 || a.Wo != (a.W - 1) / a.stride + 1) return false;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8) return false;
    const long long npix = (long long)a.ipe * a.Ho * a.Wo;
    if (npix < 8192) return false;                                   // small maps / MLP layers: the skinny kernel's territory
    if ((long long)a.ipe * a.H * a.W * a.in_ld * 2 >= 0xfff00000ll || npix * a.out_ld * 2 >= 0xfff00000ll) return false;
    const int m = (a.Cin <= 128 && a.CoutP % 128 == 0) ? 4 : 2;
    const int kch = (a.Cin + 127) / 128;
    const size_t sm = (size_t)m * kch * 8 * 1024 + (size_t)4 * 32 * (m * 4) * 16 + (size_t)4 * (64 / (m * 4)) * 2 * (m * 32) * 4 + (size_t)m * 32 * 4 +
                      (inbn ? (size_t)3 * kch * 128 * 4 : 0);
    if (sm > 128 * 1024) return false;
    const int E = a.N / a.ipe, slabs = a.CoutP / (m * 32);
    const long long tpe = (npix + 31) / 32;
    int wpe = 512 / (E * slabs);
    if (wpe < 1) wpe = 1;
    if (wpe > tpe / 16) wpe = (int)(tpe / 16);
    if (wpe < 1) wpe = 1;
    *wgs_per_expert = wpe; *tiles_per_expert = (int)tpe; *n_slabs = slabs; *mt = m; *smem = sm;
    return true;
}

int conv_c1x1_launch(const ConvArgs& a, hipStream_t st) {
    int wpe, tpe, slabs, m;
    size_t sm;
    if (!conv_c1x1_plan(a, PMOE_DT_BF16, &wpe, &tpe, &slabs, &m, &sm)) return PMOE_ERR_ARG;
    const long long in_b = (long long)a.ipe * a.H * a.W * a.in_ld * 2 - (long long)a.in_coff * 2;
    const long long out_b = (long long)a.ipe * a.Ho * a.Wo * a.out_ld * 2 * (a.shuf_c ? 4 : 1) - (long long)a.out_coff * 2;
    const int kch = (a.Cin + 127) / 128;
    dim3 grid(wpe * slabs, a.N / a.ipe), block(256);
    if (a.res_mode == PMOE_RES_INBN && m == 4) {
        HIP_RET((ensure_dyn_lds<conv1x1_direct_kernel<4, true>>(160 * 1024)));
        hipLaunchKernelGGL((conv1x1_direct_kernel<4, true>), grid, block, sm, st, a, tpe, wpe, slabs, (unsigned)in_b, (unsigned)out_b, kch);
    } else if (a.res_mode == PMOE_RES_INBN) {
        HIP_RET((ensure_dyn_lds<conv1x1_direct_kernel<2, true>>(160 * 1024)));
        hipLaunchKernelGGL((conv1x1_direct_kernel<2, true>), grid, block, sm, st, a, tpe, wpe, slabs, (unsigned)in_b, (unsigned)out_b, kch);
    } else if (m == 4) {
        HIP_RET((ensure_dyn_lds<conv1x1_direct_kernel<4>>(160 * 1024)));
        hipLaunchKernelGGL(conv1x1_direct_kernel<4>, grid, block, sm, st, a, tpe, wpe, slabs, (unsigned)in_b, (unsigned)out_b, kch);
    } else {
        HIP_RET((ensure_dyn_lds<conv1x1_direct_kernel<2>>(160 * 1024)));
        hipLaunchKernelGGL(conv1x1_direct_kernel<2>, grid, block, sm, st, a, tpe, wpe, slabs, (unsigned)in_b, (unsigned)out_b, kch);
    }
    return (int)hipGetLastError();
}
