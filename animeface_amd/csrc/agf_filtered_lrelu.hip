// filtered_lrelu for gfx950.
//
//   agf_filtered_lrelu_act : in-place gain -> leaky ReLU -> clamp on an (already upsampled) tensor, writing or
//                            reading the 2-bit sign tensor (reference filtered_lrelu.cu:1099-1210).
//   agf_filtered_lrelu     : the fused op (bias -> up-FIR -> act -> down-FIR in one pass through LDS), see below.
//
// Sign tensor: uint8 [N,C,SH,SW4], element x of a row lives in bits 2*(x&3) of byte x>>2; code 0 = pass,
// 1 = negative (slope applied), 2 = clamped (gradient zero).  The reference packs 16 lanes of a 32-lane warp with
// __shfl_xor_sync masks (filtered_lrelu.cu:1143-1150); on a 64-lane wavefront the packing is two __ballot()s
// (one per code bit) whose 16-bit quarters are bit-interleaved by the quarter's first lane: 4 uint32 stores per wave.
#include "agf_common.h"

struct ActParams {
    void* x;
    uint8_t* s;
    int N, C, H, W;
    int64_t xs[4];
    int SH, SW;          // sign tensor height, width in ELEMENTS (SW % 16 == 0 when writing)
    int ofsx, ofsy;
    float gain, slope, clamp;
};

static __device__ __forceinline__ uint32_t spread16(uint32_t v) {   // bit i -> bit 2i
    v = (v | (v << 8)) & 0x00FF00FFu;
    v = (v | (v << 4)) & 0x0F0F0F0Fu;
    v = (v | (v << 2)) & 0x33333333u;
    v = (v | (v << 1)) & 0x55555555u;
    return v;
}

static __device__ __forceinline__ float clamp_mag(float v, float c) { return fminf(fmaxf(v, -c), c); }

template <class T, int MODE>   // MODE 0 none, 1 write, 2 read
__global__ void __launch_bounds__(256) filtered_lrelu_act_kernel(ActParams p) {
    const int RW = (MODE == 1) ? p.SW : p.W;                    // logical row width of the launch
    const int RH = (MODE == 1) ? p.SH : p.H;
    const int64_t total = (int64_t)p.N * p.C * RH * RW;
    const int64_t stride = (int64_t)gridDim.x * 256;
    // uniform trip count so that __ballot sees whole waves
    const int64_t iters = (total + stride - 1) / stride;
    int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x;
    for (int64_t it = 0; it < iters; it++, id += stride) {
        const bool live = id < total;
        int64_t r = live ? id : 0;
        int x = (int)(r % RW); r /= RW;
        int y = (int)(r % RH); r /= RH;
        int c = (int)(r % p.C);
        int n = (int)(r / p.C);
        uint32_t code = 0;
        const bool inx = live && x < p.W && y < p.H;
        T* pv = (T*)p.x + n * p.xs[0] + c * p.xs[1] + y * p.xs[2] + x * p.xs[3];
        if (inx) {
            float v = (float)Elem<T>::load(pv) * p.gain;
            if (MODE == 2) {
                uint32_t sx = (uint32_t)(x + p.ofsx), sy = (uint32_t)(y + p.ofsy);
                if (sx < (uint32_t)p.SW && sy < (uint32_t)p.SH) {
                    int64_t q = (int64_t)n * p.C + c;
                    uint32_t s = p.s[(sx >> 2) + (int64_t)(p.SW >> 2) * (sy + (int64_t)p.SH * q)];
                    s >>= (sx & 3) << 1;
                    if (s & 1) v *= p.slope;
                    if (s & 2) v = 0.f;
                }
            } else {
                if (v < 0.f) { v *= p.slope; code = 1; }
                if (fabsf(v) > p.clamp) { v = clamp_mag(v, p.clamp); code = 2; }
            }
            Elem<T>::store(pv, v);
        }
        if (MODE == 1) {
            uint64_t b0 = __ballot(code & 1), b1 = __ballot(code >> 1);
            int lane = threadIdx.x & 63;
            if (live && (lane & 15) == 0) {
                int sh = lane & 48;
                uint32_t w = spread16((uint32_t)(b0 >> sh) & 0xffffu) | (spread16((uint32_t)(b1 >> sh) & 0xffffu) << 1);
                int64_t q = (int64_t)n * p.C + c;
                int64_t is = x + (int64_t)p.SW * (y + (int64_t)p.SH * q);      // element index, multiple of 16
                ((uint32_t*)p.s)[is >> 4] = w;
            }
        }
    }
}

template <class T>
static void launch_act(const ActParams& p, int mode, hipStream_t st) {
    const int RW = (mode == 1) ? p.SW : p.W, RH = (mode == 1) ? p.SH : p.H;
    int64_t total = (int64_t)p.N * p.C * RH * RW;
    int64_t blocks = agf_ceil_div(total, 256);
    if (blocks > 256 * 256) blocks = 256 * 256;
    dim3 g((unsigned)blocks), b(256);
    if (mode == 1) hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, 1>), g, b, 0, st, p);
    else if (mode == 2) hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, 2>), g, b, 0, st, p);
    else hipLaunchKernelGGL((filtered_lrelu_act_kernel<T, 0>), g, b, 0, st, p);
}

extern "C" int agf_filtered_lrelu_act(void* x, uint8_t* s, int dtype,
                                      const int32_t x_size[4], const int64_t x_stride[4],
                                      const int32_t s_size[2], const int32_t s_ofs[2], int sign_mode,
                                      float gain, float slope, float clamp, void* stream) {
    // validation mirrors filtered_lrelu.cpp:213-245
    AGF_CHECK(x, "filtered_lrelu_act: null x");
    AGF_CHECK(dtype == AGF_F32 || dtype == AGF_F16 || dtype == AGF_BF16, "x must be float16, bfloat16 or float32");
    AGF_CHECK(sign_mode >= 0 && sign_mode <= 2, "bad sign_mode");
    AGF_CHECK(sign_mode == 0 || s, "signs pointer is null");
    for (int i = 0; i < 4; i++) AGF_CHECK(x_size[i] >= 1, "x is empty");
    ActParams p;
    p.x = x; p.s = s;
    p.N = x_size[0]; p.C = x_size[1]; p.H = x_size[2]; p.W = x_size[3];
    for (int i = 0; i < 4; i++) p.xs[i] = x_stride[i];
    p.SH = sign_mode ? s_size[0] : 0;
    p.SW = sign_mode ? s_size[1] * 4 : 0;        // bytes -> elements
    p.ofsx = s_ofs ? s_ofs[0] : 0; p.ofsy = s_ofs ? s_ofs[1] : 0;
    p.gain = gain; p.slope = slope; p.clamp = clamp;
    if (sign_mode == 1) {
        AGF_CHECK(p.SW % 16 == 0 && p.SW >= p.W && p.SH >= p.H, "sign tensor has the wrong shape for writing");
        AGF_CHECK(((uintptr_t)s % 4) == 0, "signs must be 4-byte aligned");
    }
    hipStream_t st = (hipStream_t)stream;
    switch (dtype) {
        case AGF_F32:  launch_act<float>(p, sign_mode, st); break;
        case AGF_F16:  launch_act<f16_t>(p, sign_mode, st); break;
        default:       launch_act<bf16_t>(p, sign_mode, st); break;
    }
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

// TODO(fused): single-pass LDS kernel.  Until it lands every parameter set reports "no specialised kernel",
// which is the reference's own protocol for falling back to the generic 4-pass composition (filtered_lrelu.py:217-223).
extern "C" int agf_filtered_lrelu(const void* x, const float* fu, const float* fd, const void* b, uint8_t* s, void* y, int dtype,
                                  const int32_t x_size[4], const int64_t x_stride[4],
                                  const int32_t y_size[4], const int64_t y_stride[4],
                                  const int32_t fu_size[2], const int64_t fu_stride[2],
                                  const int32_t fd_size[2], const int64_t fd_stride[2],
                                  const int32_t s_size[2], const int32_t s_ofs[2], int sign_mode,
                                  int up, int down, int px0, int py0,
                                  float gain, float slope, float clamp, int flip, void* stream) {
    (void)x; (void)fu; (void)fd; (void)b; (void)s; (void)y; (void)dtype; (void)x_size; (void)x_stride; (void)y_size; (void)y_stride;
    (void)fu_size; (void)fu_stride; (void)fd_size; (void)fd_stride; (void)s_size; (void)s_ofs; (void)sign_mode;
    (void)up; (void)down; (void)px0; (void)py0; (void)gain; (void)slope; (void)clamp; (void)flip; (void)stream;
    agf_set_error("filtered_lrelu: no specialised kernel for up=%d down=%d", up, down);
    return AGF_ENOKERNEL;
}
