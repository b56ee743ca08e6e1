// 3x3 stride-1 convolution of a 16-channel input into <= 64 channels -- the first stem convolution (12 = 4 frames x RGB input
// channels padded to 16; reference: blocks/basics.py:113-120 conv1 of EfficientConvBlock, applied to the concatenated frames,
// model/moe.py:90-92).  K = 9 x 16 = 144: 2.6 % of the step's FLOPs but 2.1 GB of output -- an HBM-write-bound layer that the
// LDS-tiled kernels ran at 1.8 TB/s (1.17 ms), because their per-tile staging, barriers and epilogue are sized for K >= 576.
//
// Direct form, no LDS on the data path, no barriers in the loop:
//   * a wave owns 32 consecutive pixels of a row; the 16-channel rows ARE the MFMA B operand of v_mfma_f32_32x32x16_bf16
//     (lane = pixel, lanes 32..63 = channels 8..15): 9 fully coalesced 1-KiB loads per tile, shifted by the tap;
//   * the filter bank (18 KB per expert) sits in LDS in A-operand order, rows permuted so that the accumulators of a lane
//     are 4 groups of 8 CONSECUTIVE output channels: D[m][n] -> lane n%32 + 32*h holds channels 16g + 8h + 0..7, g = 0..3;
//     a group is one 16-byte piece of the pixel's 128-byte output row;
//   * the output tile is transposed through 4 KB of wave-private LDS so that every store instruction writes 8 whole pixel rows
//     (direct 16-byte stores from the accumulator layout, two 16-byte pieces per 128-byte line and instruction: 0.85 ms
//     against 0.73 ms, tools/ab_c16.py);
//   * BatchNorm partial sums (of the bf16-rounded outputs, like every other conv epilogue) are taken from the transposed
//     read-back, stay in registers across the wave's tiles and are folded once per workgroup, in fixed order: one
//     [2][CoutP] row per workgroup;
//   * tile coordinates are wave-uniform and advanced incrementally; every address is SGPR offset + per-lane constant.
// Measured and not kept: non-temporal stores (0.70 -> 0.65 ms in isolation, nothing in the step; the same hint on the BatchNorm /
// stem-tail passes' 2 GB outputs changed the step by less than its run-to-run noise).
#include "common.h"
#include "kernels.h"

namespace {

typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256, 2) conv3x3_c16_kernel(ConvArgs a, int tiles_x, int tiles_per_expert, int wgs_per_expert,
                                                            unsigned in_bytes, unsigned out_bytes) {
    __shared__ v4i wl[18][64];                    // [m-tile * 9 + tap][lane]: the A operand of that MFMA
    __shared__ v4i tbuf[4][256];                  // per wave: one output tile [32 pixels][8 chunks of 16 B], chunk ^= pixel & 7
    __shared__ float red[4][8][2][64];            // [wave][lane group][sum | sum of squares][channel]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int e = blockIdx.y, wg = blockIdx.x;
    {
        const bf16* w = reinterpret_cast<const bf16*>(a.w) + (size_t)e * a.CoutP * 9 * 16;
        for (int idx = tid; idx < 18 * 64; idx += 256) {
            const int l = idx & 63, mtap = idx >> 6, mt = mtap / 9, tap = mtap - mt * 9;
            const int m = l & 31, kh = l >> 5;
            const int aa = m >> 3, hh = (m >> 2) & 1, b = m & 3;                       // D row m lands in lane half hh, register 4*aa + b
            const int cout = 16 * (2 * mt + (aa >> 1)) + 8 * hh + 4 * (aa & 1) + b;
            wl[mtap][l] = ldg16(w + ((size_t)cout * 9 + tap) * 16 + kh * 8);
        }
    }
    __syncthreads();
    // all addressing is (uniform 32-bit byte offset in an SGPR) + (per-lane constant): buffer loads / stores whose per-lane
    // offset is pushed out of range where the pixel column or the channel chunk does not exist (reads return 0, writes drop)
    // (the input descriptor starts ONE PIXEL BEFORE the tensor: lane offsets of the left tap column stay non-negative)
    // one descriptor per EXPERT (in_bytes / out_bytes are per-expert sizes): the 32-bit offsets only span one expert's images
    const size_t in_img0 = a.in_shared ? 0 : (size_t)e * a.ipe, out_img0 = (size_t)e * a.ipe;
    const __amdgpu_buffer_rsrc_t rs_in = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<const bf16*>(a.in) + in_img0 * a.H * a.W * a.in_ld + a.in_coff - a.in_ld), (short)0,
        (int)(in_bytes + (unsigned)a.in_ld * 2u), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_out = __builtin_amdgcn_make_buffer_rsrc(
        (void*)(reinterpret_cast<bf16*>(a.out) + out_img0 * a.H * a.W * a.out_ld + a.out_coff), (short)0, (int)out_bytes, 0x00020000);
    constexpr unsigned OOB = 0xfffffff0u;
    const int px_l = lane & 31, kh = lane >> 5;
    const unsigned ld2 = (unsigned)a.in_ld * 2u, old2 = (unsigned)a.out_ld * 2u;
    const unsigned vin = (unsigned)(px_l + 1) * ld2 + (unsigned)kh * 16u;             // tap column 1 (centre) of this lane
    unsigned vout[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) vout[i] = (unsigned)(i * 8 + (lane >> 3)) * old2 + (unsigned)(lane & 7) * 16u;
    const bool chunk_ok = (lane & 7) * 8 < a.Cout;
    const int t0 = (int)((long long)tiles_per_expert * wg / wgs_per_expert);
    const int t1 = (int)((long long)tiles_per_expert * (wg + 1) / wgs_per_expert);
    f32x2 s1[4], s2[4];                                            // BatchNorm sums of this lane's 8-channel chunk (lane & 7)
#pragma unroll
    for (int j = 0; j < 4; ++j) s1[j] = s2[j] = f32x2{0.f, 0.f};

    struct Pos { int xt, y, img; };
    auto advance = [&](Pos& p) {
        p.xt += 4;
        while (p.xt >= tiles_x) { p.xt -= tiles_x; if (++p.y == a.H) { p.y = 0; ++p.img; } }
    };
    // (never inside a branch: the compiler's vmcnt bookkeeping assumes the fewest younger requests of any path, so one skipped
    // prefetch would turn every wait into "everything issued so far"; `live` = false parks all offsets out of range instead)
    auto load_tile = [&](const Pos& p, v4u (&raw)[9], bool live) {
        const unsigned row0 = ((unsigned)(p.img * a.H + p.y) * (unsigned)a.W + (unsigned)p.xt * 32u) * ld2;   // (y, first pixel of the tile)
        const int px = p.xt * 32 + px_l;
        unsigned vq[3];
        vq[0] = live && px >= 1 && px - 1 < a.W ? vin - ld2 : OOB;
        vq[1] = live && px < a.W ? vin : OOB;
        vq[2] = live && px + 1 < a.W ? vin + ld2 : OOB;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int yy = p.y + r - 1;
            const bool row_ok = live && (unsigned)yy < (unsigned)a.H;                  // wave-uniform
            const unsigned soff = row_ok ? row0 + (unsigned)(r - 1) * (unsigned)a.W * ld2 : 0u;
#pragma unroll
            for (int q = 0; q < 3; ++q)
                raw[r * 3 + q] = __builtin_amdgcn_raw_buffer_load_b128(rs_in, (int)(row_ok ? vq[q] : OOB), (int)soff, 0);
        }
    };
    // one tile: 18 MFMAs, transpose through the wave's LDS tile, stores + BatchNorm sums
    auto process = [&](const Pos& pc, const v4u (&cur)[9]) {
        f32x16 acc[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[mt][i] = 0.f;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int mt = 0; mt < 2; ++mt)
                acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl[mt * 9 + tap][lane]),
                                                                  __builtin_bit_cast(bf16x8, cur[tap]), acc[mt], 0, 0, 0);
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float v[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) v[i] = acc[g >> 1][8 * (g & 1) + i];
            tbuf[wave][px_l * 8 + ((2 * g + kh) ^ (px_l & 7))] = pack16<bf16>(v);
        }
        // every store instruction writes 8 whole 128-byte pixel rows, and a lane keeps ONE channel chunk for all pixels and
        // tiles, so the BatchNorm sums need 16 registers instead of 64
        const unsigned osoff = ((unsigned)(pc.img * a.H + pc.y) * (unsigned)a.W + (unsigned)pc.xt * 32u) * old2;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int p = i * 8 + (lane >> 3), c = lane & 7;
            const v4i v = tbuf[wave][p * 8 + (c ^ (p & 7))];
            const bool ok = pc.xt * 32 + p < a.W;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v), rs_out, (int)(ok && chunk_ok ? vout[i] : OOB), (int)osoff, 0);
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
    // three tile buffers in rotation: the loads of tiles t+4 and t+8 are in flight while tile t is computed (two waves per SIMD
    // and ~1 us to the Infinity Cache: one tile of look-ahead left the waves waiting on their operands)
    Pos pl, pc;                                                    // next tile to load / to compute
    {
        const int t = t0 + wave;
        pl.xt = t % tiles_x;
        const int r = t / tiles_x;
        pl.y = r % a.H;
        pl.img = r / a.H;
        pc = pl;
    }
    v4u b0[9], b1[9], b2[9];
    int tl = t0 + wave, t = tl;                                    // tl: next tile to load
    auto fetch = [&](v4u (&buf)[9]) {
        load_tile(pl, buf, tl < t1);
        advance(pl);
        tl += 4;
    };
    fetch(b0);
    fetch(b1);
    while (t < t1) {
        fetch(b2);
        process(pc, b0); advance(pc); t += 4;
        if (t >= t1) break;
        fetch(b0);
        process(pc, b1); advance(pc); t += 4;
        if (t >= t1) break;
        fetch(b1);
        process(pc, b2); advance(pc); t += 4;
    }
    if (!a.stats) return;
    // fold: the 8 lane groups of a wave and the 4 waves hold 32 partial sums per channel; fixed order
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        red[wave][lane >> 3][0][(lane & 7) * 8 + q] = s1[q >> 1][q & 1];
        red[wave][lane >> 3][1][(lane & 7) * 8 + q] = s2[q >> 1][q & 1];
    }
    __syncthreads();
    if (tid < 128) {
        const int which = tid >> 6, c = tid & 63;
        float s = 0.f;
        for (int w = 0; w < 4; ++w)
            for (int pg = 0; pg < 8; ++pg) s += red[w][pg][which][c];
        a.stats[(((size_t)e * wgs_per_expert + wg) * 2 + which) * a.CoutP + c] = s;
    }
}

}  // namespace

// PMOE_CONV_C16=0: A/B switch back to conv3x3_res_kernel<5> (read per launch)
bool conv_c16_plan(const ConvArgs& a, int dtype, int* wgs_per_expert, int* tiles_x, int* tiles_per_expert) {
    const char* ev = getenv("PMOE_CONV_C16");
    if (ev && !atoi(ev)) return false;
    if (dtype != PMOE_DT_BF16 || a.w_fp8 || a.ks != 3 || a.stride != 1 || a.pad != 1 || a.dilate) return false;
    if (a.Cin != 16 || a.CoutP != 64 || a.Cout % 8 || a.Cout > 64) return false;
    if (a.bias || a.act != PMOE_ACT_NONE || a.res_mode != PMOE_RES_NONE || a.drop_p > 0.f) return false;
    if (a.N % a.ipe || a.Ho != a.H || a.Wo != a.W) return false;
    if (a.in_ld % 8 || a.in_coff % 8 || a.out_ld % 8 || a.out_coff % 8) return false;
    // 32-bit byte offsets inside one buffer descriptor per expert (the kernel marks missing pixels with offsets >= 0xfffffff0)
    if ((long long)a.ipe * a.H * a.W * (a.in_ld > a.out_ld ? a.in_ld : a.out_ld) * 2 >= 0xfff00000ll) return false;
    const int E = a.N / a.ipe, tx = (a.W + 31) / 32;
    const long long tpe = (long long)a.ipe * a.H * tx;
    if (tpe > 0x7fffffffll || tpe < 64) return false;                // tiny inputs: the tiled kernels' launch shapes are fine
    int wpe = 512 / E;                                               // two 4-wave workgroups per CU in one round
    if (wpe < 1) wpe = 1;
    if (wpe > tpe / 16) wpe = (int)(tpe / 16);                       // >= 4 tiles per wave: the stat fold is amortised
    if (wpe < 1) wpe = 1;
    *wgs_per_expert = wpe; *tiles_x = tx; *tiles_per_expert = (int)tpe;
    return true;
}

int conv_c16_launch(const ConvArgs& a, hipStream_t st) {
    int wpe, tx, tpe;
    if (!conv_c16_plan(a, PMOE_DT_BF16, &wpe, &tx, &tpe)) return PMOE_ERR_ARG;
    const long long in_b = (long long)a.ipe * a.H * a.W * a.in_ld * 2 - (long long)a.in_coff * 2;
    const long long out_b = (long long)a.ipe * a.H * a.W * a.out_ld * 2 - (long long)a.out_coff * 2;
    hipLaunchKernelGGL(conv3x3_c16_kernel, dim3(wpe, a.N / a.ipe), dim3(256), 0, st, a, tx, tpe, wpe, (unsigned)in_b, (unsigned)out_b);
    return (int)hipGetLastError();
}
