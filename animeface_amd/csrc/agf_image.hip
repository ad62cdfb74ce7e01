// GPU-side input transform: the reference's dataset transform (dataset/_base.py:18-37: torchvision Resize -> CenterCrop ->
// RandomHorizontalFlip -> ToTensor -> Normalize on a PIL image) for a batch of decoded uint8 images resident in HBM.
// torchvision's Resize on a PIL image is Pillow's BILINEAR resampling: separable, anti-aliased (the triangle filter widens with the
// down-scaling factor), 8-bit fixed point with 22 fractional bits, a horizontal pass then a vertical pass, each ROUNDING TO uint8.
// Byte / integer work, reproduced bit-exactly:
//   agf_image_resample_rows : horizontal pass, only the rows the vertical pass will read and only the columns the crop keeps
//   agf_image_finish        : vertical pass + centre crop (already folded into the tables' ranges) + per-image horizontal flip +
//                             v / 255 and (x - 0.5) / 0.5 -> float32 / bf16 NCHW
// The fixed-point tap tables are made on the host in double precision exactly as Pillow makes them (animeface_amd/dataset.py); the kernels
// do the integer accumulation  clip8((sum taps * v + 2^21) >> 22).
#include "agf_common.h"

#define IMG_PRECISION_BITS 22

__global__ void __launch_bounds__(256) image_resample_rows_kernel(const uint8_t* src, uint8_t* dst, const int32_t* first, const int32_t* count,
                                                                  const int32_t* taps, int ksize, int N, int H, int W, int C, int row0, int rows, int OW) {
    const int64_t total = (int64_t)N * rows * OW * C;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int c = (int)(id % C);
        int64_t t = id / C;
        const int ox = (int)(t % OW); t /= OW;
        const int r = (int)(t % rows);
        const int n = (int)(t / rows);
        const uint8_t* p = src + (((int64_t)n * H + row0 + r) * W + first[ox]) * C + c;
        const int32_t* k = taps + (int64_t)ox * ksize;
        int32_t acc = 1 << (IMG_PRECISION_BITS - 1);
        const int cnt = count[ox];
        for (int i = 0; i < cnt; i++) acc += k[i] * (int32_t)p[(int64_t)i * C];
        acc >>= IMG_PRECISION_BITS;
        dst[id] = (uint8_t)(acc < 0 ? 0 : acc > 255 ? 255 : acc);
    }
}

template <class T>
__global__ void __launch_bounds__(256) image_finish_kernel(const uint8_t* tmp, T* out, const int32_t* first, const int32_t* count, const int32_t* taps,
                                                           int ksize, const uint8_t* flip, int N, int rows, int row0, int SW, int C, int SH, int normalize) {
    // one thread per output element [n][c][y][x] (x fastest: coalesced stores; the uint8 reads of a wave cover 64 x C consecutive bytes)
    const int64_t total = (int64_t)N * C * SH * SW;
    for (int64_t id = (int64_t)blockIdx.x * 256 + threadIdx.x; id < total; id += (int64_t)gridDim.x * 256) {
        const int x = (int)(id % SW);
        int64_t t = id / SW;
        const int y = (int)(t % SH); t /= SH;
        const int c = (int)(t % C);
        const int n = (int)(t / C);
        const int sx = (flip && flip[n]) ? SW - 1 - x : x;
        const uint8_t* p = tmp + (((int64_t)n * rows + (first[y] - row0)) * SW + sx) * C + c;
        const int32_t* k = taps + (int64_t)y * ksize;
        int32_t acc = 1 << (IMG_PRECISION_BITS - 1);
        const int cnt = count[y];
        for (int i = 0; i < cnt; i++) acc += k[i] * (int32_t)p[(int64_t)i * SW * C];
        acc >>= IMG_PRECISION_BITS;
        acc = acc < 0 ? 0 : acc > 255 ? 255 : acc;
        float v = (float)acc / 255.0f;                              // ToTensor
        if (normalize) v = (v - 0.5f) / 0.5f;                       // Normalize(0.5, 0.5)
        Elem<T>::store(out + id, v);
    }
}

extern "C" int agf_image_resample_rows(const void* src, void* dst, const int32_t* first, const int32_t* count, const int32_t* taps, int32_t ksize,
                                       int32_t N, int32_t H, int32_t W, int32_t C, int32_t row0, int32_t rows, int32_t OW, void* stream) {
    AGF_CHECK(src && dst && first && count && taps, "image_resample_rows: null pointer");
    AGF_CHECK(N >= 1 && H >= 1 && W >= 1 && C >= 1 && OW >= 1 && ksize >= 1, "image_resample_rows: empty image");
    AGF_CHECK(row0 >= 0 && rows >= 1 && row0 + rows <= H, "image_resample_rows: row range [%d, %d) outside the image height %d", row0, row0 + rows, H);
    const int64_t total = (int64_t)N * rows * OW * C;
    int64_t blocks = agf_ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(image_resample_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)src, (uint8_t*)dst,
                       first, count, taps, ksize, N, H, W, C, row0, rows, OW);
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}

extern "C" int agf_image_finish(const void* tmp, void* out, const int32_t* first, const int32_t* count, const int32_t* taps, int32_t ksize,
                                const uint8_t* flip, int dtype, int32_t N, int32_t rows, int32_t row0, int32_t SW, int32_t C, int32_t SH,
                                int normalize, void* stream) {
    AGF_CHECK(tmp && out && first && count && taps, "image_finish: null pointer");
    AGF_CHECK(dtype == AGF_F32 || dtype == AGF_BF16, "image_finish: output dtype must be f32 or bf16");
    AGF_CHECK(N >= 1 && rows >= 1 && SW >= 1 && SH >= 1 && C >= 1 && ksize >= 1, "image_finish: empty image");
    const int64_t total = (int64_t)N * C * SH * SW;
    int64_t blocks = agf_ceil_div(total, 256);
    if (blocks > 65536) blocks = 65536;
    if (dtype == AGF_F32)
        hipLaunchKernelGGL(image_finish_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp, (float*)out,
                           first, count, taps, ksize, flip, N, rows, row0, SW, C, SH, normalize);
    else
        hipLaunchKernelGGL(image_finish_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (const uint8_t*)tmp, (bf16_t*)out,
                           first, count, taps, ksize, flip, N, rows, row0, SW, C, SH, normalize);
    AGF_LAUNCH_CHECK();
    return AGF_OK;
}
