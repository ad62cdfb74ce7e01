// Declarations shared by the MFMA convolution translation units (agf_conv2d.hip, agf_conv2d_pipe.hip).
#pragma once
#include "agf_common.h"
#include <stdlib.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define BLOCK_PIX 256
#define XNONE (-2147483647 - 1)     // "no load" marker of the precomputed patch offsets (real offsets can be negative: halo)

static __device__ __forceinline__ u32x4 scale_vec8_reg(u32x4 val, f32x4 s0, f32x4 s1) {
    float a0, a1;
    Pack16<bf16_t>::unpack(val.x, a0, a1); val.x = Pack16<bf16_t>::pack(a0 * s0.x, a1 * s0.y);
    Pack16<bf16_t>::unpack(val.y, a0, a1); val.y = Pack16<bf16_t>::pack(a0 * s0.z, a1 * s0.w);
    Pack16<bf16_t>::unpack(val.z, a0, a1); val.z = Pack16<bf16_t>::pack(a0 * s1.x, a1 * s1.y);
    Pack16<bf16_t>::unpack(val.w, a0, a1); val.w = Pack16<bf16_t>::pack(a0 * s1.z, a1 * s1.w);
    return val;
}

static __device__ __forceinline__ u32x4 scale_vec8(u32x4 val, const float* sc) {
    f32x4 s0 = *(const f32x4*)sc, s1 = *(const f32x4*)(sc + 4);
    float a0, a1;
    Pack16<bf16_t>::unpack(val.x, a0, a1); val.x = Pack16<bf16_t>::pack(a0 * s0.x, a1 * s0.y);
    Pack16<bf16_t>::unpack(val.y, a0, a1); val.y = Pack16<bf16_t>::pack(a0 * s0.z, a1 * s0.w);
    Pack16<bf16_t>::unpack(val.z, a0, a1); val.z = Pack16<bf16_t>::pack(a0 * s1.x, a1 * s1.y);
    Pack16<bf16_t>::unpack(val.w, a0, a1); val.w = Pack16<bf16_t>::pack(a0 * s1.z, a1 * s1.w);
    return val;
}

// the value of lane ^ 1: DPP quad_perm [1, 0, 3, 2] -- one VALU move (a __shfl_xor is a ds_bpermute through the LDS pipe)
static __device__ __forceinline__ uint32_t agf_swap1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false);
}

struct ConvParams {
    const bf16_t* x;          // [N,H,W,Cin]
    const bf16_t* w;          // [Cout,KS,KS,Cin]
    bf16_t* y;                // [N,H,W,Cout]
    const float* in_scale;    // [N,Cin] or null
    const float* out_scale;   // [N,Cout] or null
    const float* bias;        // [Cout] or null
    const float* noise;       // [N,H,W] or null
    const bf16_t* residual;   // [N,H,W,Cout] or null
    const float* post_scale;  // [N,Cout] or null: the stored output is act(...) * gain * post_scale[n,co] -- the style scale of the NEXT modulated conv,
                              //   this layer's only consumer, which then takes its input unscaled (conv_epilogue / conv_epilogue_pl only)
    uint32_t* pool_mask;      // or null.  Non-null (agf_conv2d_fwd_pool): y is the POOLED tensor [N][H/2][W/2][Cout] = pool_gain * (sum of the 2x2 cell of the
    float pool_gain;          //   bf16-rounded epilogue result), pool_mask [N][H/2][W/2][Cout/8] the 1-bit sign mask of the full-resolution result (the format of
                              //   agf_pool2x2); the full-resolution tensor is never written (conv_epilogue_pl with TW == 32, and the pipe kernel)
    int N, H, W, Cin, Cout;
    int TI, TH, TW;           // pixel tile
    int tilesW, tilesH, tilesN, tilesCo, pixTiles;
    int act;                  // 1 linear, 3 lrelu
    float alpha, gain;
    const bf16_t* mask_y;     // fused lrelu gradient (agf_conv2d_fwd_mask): y *= mask_y > 0 ? 1 : mask_alpha; null = off
    float mask_alpha;
    float* mask_sum;          // [256][Cout] fp32: += sum over pixels of the masked output (nullable)
    const uint32_t* mask_bits;// the same lrelu mask as ONE BIT per element (agf_conv2d_fwd_maskbits): [N][H][W][Cout/32] dwords, bit 8g + e of dword k =
                              //   (y[.., 32k + 8g + e] > 0), written by the producer's launch through bits_out; replaces mask_y (Cout % 32 == 0)
    uint32_t* bits_out;       // or null: this launch also writes the sign bits of its stored output in that format (agf_conv2d_fwd_bits)
    const bf16_t* res_pooled; // [N,H/2,W/2,Cout]: y += res_scale * res_pooled[h/2,w/2] before the mask (the adjoint of a 2x2 average that shares
    float res_scale;          //   this conv's input: the other branch of a residual block); null = off
    int vecStore;             // epilogue: transpose through LDS and store 16-byte vectors (needs Cout % 8 == 0, y 16-byte aligned)
    int hoist;                // A/B switch: hoist the style-scale loads out of the per-vector staging loop
    int wsSlices;             // ping-pong weight-stationary kernel: blocks per image
    int twShift, thShift;     // TW = 1 << twShift, TH = 1 << thShift (both are powers of two)
    uint32_t mPW, mPH;        // magic multipliers for division by PW, PH (operands < 2^16)
    int flat, flatTiles;      // flat tiling: a tile = 256 consecutive pixels (row-major) of one image; flatTiles = tiles per image
    uint32_t mW;              // magic multiplier for division by W (flat tiling)
    int xcdBand;              // pixel tiles per XCD (contiguous bands), 0 = interleaved
    int coXcd;                // conv2d_fwd_kernel: co tiles (not pixel tiles) spread over the XCDs
    int narrow;               // 32 co x 256 px tiling of the small maps (launch_fwd)
    int yMul, yOffH, yOffW, yH, yW;   // conv_epilogue: strided store into a [N, yH, yW, Cout] tensor (yMul = 0: the ordinary [N, H, W, Cout])
    int splitK;               // conv2d_fwd_kernel on the 4x4 / 8x8 maps: blockIdx.y = slice of the input-channel chunks; > 1: partial tiles go to
    float* splitWs;           //   splitWs [tile][slice][register][thread] (fp32, the accumulator layout as it is), the LAST slice of a tile to arrive
    unsigned* splitCnt;       //   (splitCnt [tile], self-resetting) adds them in slice order and runs the epilogue
};

// persistent multi-stage direct-to-LDS kernel (agf_conv2d_pipe.hip); AGF_ENOKERNEL = shape not covered, use the other kernels
int agf_conv2d_pipe_launch(const ConvParams& p, hipStream_t st);

// streaming 1x1 kernel for the few-channel, many-pixel layers (agf_conv1x1.hip); AGF_ENOKERNEL = shape not covered
int agf_conv1x1_stream_launch(const ConvParams& p, hipStream_t st);

// multi-stage ring variant of the 3x3 weight gradient (agf_conv2d_wgrad_ring.hip); AGF_ENOKERNEL = shape not covered.
// workspace (optional): scratch of agf_conv2d_wgrad_ring_workspace() bytes -> two-stage combine, dw is overwritten instead of accumulated into
int agf_conv2d_wgrad_ring_launch(const void* x, const void* dy, float* dw, const float* in_scale, const float* out_scale,
                                 int N, int H, int W, int Cin, int Cout, float scale, float* workspace, int64_t workspaceBytes, int oihw, hipStream_t st);
int64_t agf_conv2d_wgrad_ring_workspace(bool scales, int N, int H, int W, int Cin, int Cout);
