// placeholder (real kernels follow)
#include "agf_common.h"
extern "C" int agf_conv2d_fwd(const void* x, const void* w, void* y,
                   const float* in_scale, const float* out_scale, const float* bias,
                   const float* noise, const void* residual,
                   int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                   int act, float alpha, float act_gain, void* stream) { agf_set_error("conv2d_fwd: not built"); return AGF_ENOKERNEL; }
extern "C" int agf_conv2d_wgrad(const void* x, const void* dy, float* dw,
                     const float* in_scale, const float* out_scale,
                     int32_t N, int32_t H, int32_t W, int32_t Cin, int32_t Cout, int32_t ksize,
                     void* stream) { agf_set_error("conv2d_wgrad: not built"); return AGF_ENOKERNEL; }
