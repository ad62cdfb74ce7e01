// 1x1 convolution of the few-channel, many-pixel layers as a streaming MFMA kernel (the DBlock's skip conv and its data gradient:
// reference implementations/StyleGAN2/model.py:186-212, ``self.skip = Conv2d(in, out, 1)`` and the residual sum).
//
//     y[p, co] = act( sum_ci x[p, ci] * w[co, ci] + bias[co] + residual[p, co] ) * gain          p = pixel of [N, H, W], channels-last bf16
//
// These launches sit far below the MFMA / HBM ridge (2 Cin Cout flops against 2 (Cin + Cout [+ Cout]) bytes per pixel: 13-85 flop per byte):
// all that matters is that x, the residual and y stream at HBM rate.  The generic implicit-GEMM kernel stages a 256-pixel patch per block
// through LDS with a barrier per K chunk -- all prologue and epilogue when K is one or two chunks -- and ran these layers at 0.15-0.25 of
// the HBM roofline (profiles/r04_conv_shapes.txt).  Here there is no patch and no barrier in the pixel loop:
//   * the weights (Cout x Cin <= 8 K elements) sit in LDS for the life of the block, rows padded by 16 bytes;
//   * a wave takes 32 consecutive pixels at a time: the B fragments of v_mfma_f32_32x32x16_bf16 (8 consecutive input channels of pixel
//     lane & 31) are 16-byte loads STRAIGHT FROM GLOBAL MEMORY -- a pixel row of Cin bf16 is covered by Cin/16 such loads, whole sectors;
//   * per 32 output channels: Cin/16 MFMAs, then the lane swap of the 32x32 result layout (v_permlane32_swap) gives every lane 8
//     consecutive channels of its pixel: the residual is read and y is written as 16-byte vectors;
//   * waves walk the pixel tiles with a grid stride; the next tile's fragments and the residual vectors of four output-channel tiles are
//     requested before the MFMAs of the current ones (a first version without that look-ahead ran at 2.2 TB/s, latency-bound).
#include "agf_conv2d_common.h"

namespace {

struct Conv1Params {
    const bf16_t* x; const bf16_t* w; bf16_t* y; const float* bias; const bf16_t* residual;
    int64_t P;                 // pixels
    int Cin, Cout, act;
    float alpha, gain;
};

constexpr int C1_WAVES = 4;

}  // namespace

template <int KSTEPS, int CT>      // Cin / 16, Cout / 32
__global__ void __launch_bounds__(64 * C1_WAVES) conv1x1_stream_kernel(Conv1Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;
    constexpr int Cin = KSTEPS * 16, Cout = CT * 32;
    constexpr int pitch = Cin * 2 + 16;                              // bytes per weight row in LDS
    constexpr int G = CT < 4 ? CT : 4;                               // output-channel tiles whose residual vectors are in flight together
    constexpr bool PREFETCH = KSTEPS <= 8;                           // next pixel tile's fragments loaded under this tile's work
    // weights -> LDS (once per block)
    for (int v = tid; v < Cout * (Cin / 8); v += 64 * C1_WAVES) {
        const int co = v / (Cin / 8), cv = v - co * (Cin / 8);
        *(u32x4*)(smem + co * pitch + cv * 16) = *(const u32x4*)(p.w + (int64_t)co * Cin + cv * 8);
    }
    __syncthreads();
    const int64_t tiles = (p.P + 31) / 32;
    const int64_t stride = (int64_t)gridDim.x * C1_WAVES;
    const bool plain = !p.residual && p.act != 3 && p.gain == 1.f;
    auto load_b = [&](int64_t t, u32x4 (&dst)[KSTEPS]) {
        const int64_t px = t * 32 + l31;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ks++) {
            dst[ks] = u32x4{0u, 0u, 0u, 0u};
            if (t < tiles && px < p.P) dst[ks] = *(const u32x4*)(p.x + px * Cin + ks * 16 + lhi * 8);
        }
    };
    u32x4 bcur[KSTEPS], bnext[PREFETCH ? KSTEPS : 1];
    int64_t t = (int64_t)blockIdx.x * C1_WAVES + wave;
    if constexpr (PREFETCH) load_b(t, bcur);
    for (; t < tiles; t += stride) {
        const int64_t px = t * 32 + l31;
        const bool valid = px < p.P;
        if constexpr (!PREFETCH) load_b(t, bcur);
#pragma unroll
        for (int g0 = 0; g0 < CT; g0 += G) {
            // the residual vectors of this group of output-channel tiles: all in flight before the first MFMA
            u32x4 rv[G][2];
            if (p.residual) {
#pragma unroll
                for (int c = 0; c < G; c++)
#pragma unroll
                    for (int q = 0; q < 2; q++) {
                        rv[c][q] = u32x4{0u, 0u, 0u, 0u};
                        if (valid) rv[c][q] = *(const u32x4*)(p.residual + px * Cout + (g0 + c) * 32 + (2 * q + lhi) * 8);
                    }
            }
            if constexpr (PREFETCH) { if (g0 == 0) load_b(t + stride, bnext); }
#pragma unroll
            for (int c = 0; c < G; c++) {
                const int ct = g0 + c;
                f32x16 acc = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                const unsigned char* wrow = smem + (ct * 32 + l31) * pitch + lhi * 16;
#pragma unroll
                for (int ks = 0; ks < KSTEPS; ks++) {
                    const bf16x8 af = *(const bf16x8*)(wrow + ks * 32);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, bcur[ks]), acc, 0, 0, 0);
                }
                // result layout: acc[rg * 4 + e] = (pixel l31, channel ct*32 + rg*8 + lhi*4 + e); the lane swap (v_permlane32_swap) gives lanes
                // < 32 the 8 channels of group 2q and lanes >= 32 those of group 2q+1
#pragma unroll
                for (int q = 0; q < 2; q++) {
                    const int cb = ct * 32 + (2 * q + lhi) * 8;
                    f32x4 ba = {0.f, 0.f, 0.f, 0.f}, bb = {0.f, 0.f, 0.f, 0.f};
                    if (p.bias) { ba = *(const f32x4*)(p.bias + ct * 32 + (2 * q) * 8 + lhi * 4); bb = *(const f32x4*)(p.bias + ct * 32 + (2 * q + 1) * 8 + lhi * 4); }
                    u32x4 val;
                    if (plain) {
                        // nothing is added after the swap: the values cross it as packed bf16 pairs (2 swaps instead of 4)
                        const uint32_t a0 = Pack16<bf16_t>::pack(acc[(2 * q) * 4 + 0] + ba.x, acc[(2 * q) * 4 + 1] + ba.y);
                        const uint32_t a1 = Pack16<bf16_t>::pack(acc[(2 * q) * 4 + 2] + ba.z, acc[(2 * q) * 4 + 3] + ba.w);
                        const uint32_t b0 = Pack16<bf16_t>::pack(acc[(2 * q + 1) * 4 + 0] + bb.x, acc[(2 * q + 1) * 4 + 1] + bb.y);
                        const uint32_t b1 = Pack16<bf16_t>::pack(acc[(2 * q + 1) * 4 + 2] + bb.z, acc[(2 * q + 1) * 4 + 3] + bb.w);
                        const auto s0 = __builtin_amdgcn_permlane32_swap(a0, b0, false, false);
                        const auto s1 = __builtin_amdgcn_permlane32_swap(a1, b1, false, false);
                        val = u32x4{s0[0], s1[0], s0[1], s1[1]};
                    } else {
                        // fp32 values are swapped, so the sum with the residual is rounded to bf16 exactly once
                        float g[8];
#pragma unroll
                        for (int e = 0; e < 4; e++) {
                            const float a = acc[(2 * q) * 4 + e] + ba[e], b = acc[(2 * q + 1) * 4 + e] + bb[e];
                            const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(a), __float_as_uint(b), false, false);
                            g[e] = __uint_as_float(sw[0]);
                            g[4 + e] = __uint_as_float(sw[1]);
                        }
                        if (p.residual) {
                            float r[8];
                            Pack16<bf16_t>::unpack(rv[c][q].x, r[0], r[1]); Pack16<bf16_t>::unpack(rv[c][q].y, r[2], r[3]);
                            Pack16<bf16_t>::unpack(rv[c][q].z, r[4], r[5]); Pack16<bf16_t>::unpack(rv[c][q].w, r[6], r[7]);
#pragma unroll
                            for (int e = 0; e < 8; e++) g[e] += r[e];
                        }
                        if (p.act == 3) {
#pragma unroll
                            for (int e = 0; e < 8; e++) g[e] = g[e] > 0.f ? g[e] : g[e] * p.alpha;
                        }
#pragma unroll
                        for (int e = 0; e < 8; e++) g[e] *= p.gain;
                        val.x = Pack16<bf16_t>::pack(g[0], g[1]); val.y = Pack16<bf16_t>::pack(g[2], g[3]);
                        val.z = Pack16<bf16_t>::pack(g[4], g[5]); val.w = Pack16<bf16_t>::pack(g[6], g[7]);
                    }
                    if (valid) *(u32x4*)(p.y + px * Cout + cb) = val;
                }
            }
        }
        if constexpr (PREFETCH) {
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ks++) bcur[ks] = bnext[ks];
        }
    }
}

// AGF_ENOKERNEL = shape not covered (the caller falls through to the generic kernel)
int agf_conv1x1_stream_launch(const ConvParams& c, hipStream_t st) {
    const int64_t P = (int64_t)c.N * c.H * c.W;
    if (c.in_scale || c.out_scale || c.noise || c.mask_y || c.res_pooled || c.yMul) return AGF_ENOKERNEL;
    if (c.Cin % 16 || c.Cin < 16 || c.Cin > 256 || c.Cout % 32 || c.Cout > 512 || (int64_t)c.Cin * c.Cout > 8192) return AGF_ENOKERNEL;   // (beyond: register-bound, the generic kernel is as fast)
    if (P < 65536) return AGF_ENOKERNEL;                              // small maps: launch-bound either way
    if (((uintptr_t)c.x % 16) || ((uintptr_t)c.y % 16) || ((uintptr_t)c.w % 16) || (c.residual && ((uintptr_t)c.residual % 16))) return AGF_ENOKERNEL;
    Conv1Params p;
    p.x = c.x; p.w = c.w; p.y = c.y; p.bias = c.bias; p.residual = c.residual; p.P = P; p.Cin = c.Cin; p.Cout = c.Cout;
    p.act = c.act; p.alpha = c.alpha; p.gain = c.gain;
    const size_t lds = (size_t)c.Cout * (c.Cin * 2 + 16);
    const int64_t tiles = (P + 31) / 32;
    int64_t blocks = (tiles + C1_WAVES - 1) / C1_WAVES;
    const int64_t cap = 256 * 8;                                      // 8 blocks (32 waves) per CU at most; grid stride beyond
    if (blocks > cap) blocks = cap;
#define C1_LAUNCH(K, C)                                                                                                           \
    if (c.Cin == 16 * K && c.Cout == 32 * C) {                                                                                    \
        if (lds > 48 * 1024 && hipFuncSetAttribute((const void*)conv1x1_stream_kernel<K, C>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                                   (int)lds) != hipSuccess) return AGF_ENOKERNEL;                                 \
        hipLaunchKernelGGL((conv1x1_stream_kernel<K, C>), dim3((unsigned)blocks), dim3(64 * C1_WAVES), lds, st, p);                \
        return AGF_OK;                                                                                                            \
    }
    // the channel pairs of the StyleGAN2 discriminator's skip convs (C -> 2C, forward) and their data gradients (2C -> C), C = 32, 64
    C1_LAUNCH(2, 2) C1_LAUNCH(4, 4) C1_LAUNCH(4, 1) C1_LAUNCH(8, 2)
    C1_LAUNCH(2, 1) C1_LAUNCH(4, 2) C1_LAUNCH(2, 4) C1_LAUNCH(1, 1) C1_LAUNCH(1, 2)
#undef C1_LAUNCH
    return AGF_ENOKERNEL;
}
