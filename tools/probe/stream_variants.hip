// Probe: which streaming pattern reaches the highest HBM rate for y = lrelu(x + b[c]) on bf16 / fp32 channels-last data (the bias_act hot
// case)?  Variants: U accesses in flight per lane (1 / 2 / 4), stride-separated vs block-contiguous, non-temporal loads / stores, grid size.
//   hipcc --offload-arch=gfx950 -O3 -o stream_variants stream_variants.hip && ./stream_variants
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#include <vector>
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));

template <bool BF> __device__ __forceinline__ u32x4 act(u32x4 x, u32x4 b) {
    u32x4 r;
    if (BF) {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            uint32_t w = x[i], bw = b[i];
            float lo = __uint_as_float(w << 16) + __uint_as_float(bw << 16), hi = __uint_as_float(w & 0xffff0000u) + __uint_as_float(bw & 0xffff0000u);
            lo = (lo > 0 ? lo : lo * 0.2f) * 1.41421356f; hi = (hi > 0 ? hi : hi * 0.2f) * 1.41421356f;
            typedef __bf16 bf2 __attribute__((ext_vector_type(2)));
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 v; v.x = lo; v.y = hi;
            bf2 h = __builtin_convertvector(v, bf2);
            r[i] = __builtin_bit_cast(uint32_t, h);
        }
    } else {
#pragma unroll
        for (int i = 0; i < 4; i++) {
            float v = __uint_as_float(x[i]) + __uint_as_float(b[i]);
            v = (v > 0 ? v : v * 0.2f) * 1.41421356f;
            r[i] = __float_as_uint(v);
        }
    }
    return r;
}

// MODE 0: accesses of one lane are `stride` apart (grid-stride, U in flight); MODE 1: a block owns U * 256 consecutive vectors per round
template <bool BF, int U, int MODE, bool NTL, bool NTS>
__global__ void __launch_bounds__(256) k(const u32x4* __restrict__ x, u32x4* __restrict__ y, const u32x4* __restrict__ b, uint32_t nvec, uint32_t bmask) {
    const uint32_t stride = gridDim.x * 256u;
    for (uint32_t base = MODE == 0 ? blockIdx.x * 256u + threadIdx.x : blockIdx.x * (256u * U) + threadIdx.x; base < nvec; base += stride * U) {
        u32x4 rx[U], rb[U];
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t v = base + (MODE == 0 ? j * stride : j * 256u);
            if (v < nvec) {
                rx[j] = NTL ? __builtin_nontemporal_load(x + v) : x[v];
                rb[j] = b[v & bmask];
            }
        }
#pragma unroll
        for (int j = 0; j < U; j++) {
            const uint32_t v = base + (MODE == 0 ? j * stride : j * 256u);
            if (v < nvec) {
                const u32x4 r = act<BF>(rx[j], rb[j]);
                if (NTS) __builtin_nontemporal_store(r, y + v); else y[v] = r;
            }
        }
    }
}

template <bool BF, int U, int MODE, bool NTL, bool NTS>
static void run(const char* name, const u32x4* x, u32x4* y, const u32x4* b, uint32_t nvec, uint32_t bmask, int blocks) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int64_t need = ((int64_t)nvec + 256 * U - 1) / (256 * U);
    if (blocks <= 0 || blocks > need) blocks = (int)need;
    float best = 1e9f;
    for (int rep = 0; rep < 4; rep++) {
        for (int w = 0; w < 3; w++) hipLaunchKernelGGL((k<BF, U, MODE, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, b, nvec, bmask);
        hipEventRecord(e0, 0);
        for (int i = 0; i < 20; i++) hipLaunchKernelGGL((k<BF, U, MODE, NTL, NTS>), dim3(blocks), dim3(256), 0, 0, x, y, b, nvec, bmask);
        hipEventRecord(e1, 0);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms / 20 < best) best = ms / 20;
    }
    const double bytes = 2.0 * nvec * 16;
    printf("%-4s U=%d mode=%d ntl=%d nts=%d blocks=%6d : %7.1f us  %5.2f TB/s  %.3f of 8\n", name, U, MODE, (int)NTL, (int)NTS, blocks, best * 1e3, bytes / best / 1e9, bytes / best / 1e9 / 8.0);
}

__global__ void fill_random(uint32_t* p, size_t n, uint32_t seed) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        uint32_t h = (uint32_t)i * 2654435761u ^ seed; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13; h *= 3266489917u; h ^= h >> 16;
        // two bf16 (or one fp32) of magnitude ~1 with random sign and mantissa: exponent field 0x3f
        p[i] = (h & 0x807f807fu) | 0x3f003f00u;
    }
}

int main(int argc, char** argv) {
    const bool randomData = argc > 1;
    for (int pass = 0; pass < 2; pass++) {
        const bool bf = pass == 0;
        const size_t bytes = (size_t)64 * 64 * 256 * 256 * (bf ? 2 : 4);
        const uint32_t nvec = (uint32_t)(bytes / 16);
        u32x4 *x, *y, *b;
        hipMalloc(&x, bytes); hipMalloc(&y, bytes); hipMalloc(&b, 4096);
        hipMemset(x, 0x3c, bytes); hipMemset(b, 0, 4096);
        if (randomData) { hipLaunchKernelGGL(fill_random, dim3(65536), dim3(256), 0, 0, (uint32_t*)x, bytes / 4, 12345u); hipDeviceSynchronize(); }
        printf("# %s data\n", randomData ? "random" : "constant");
        const uint32_t bmask = bf ? 7 : 15;                  // 64 channels = 8 / 16 vectors
        const char* n = bf ? "bf16" : "fp32";
        const int grids[] = {0, 16384};
        for (int g : grids) {
#define RUN(U, M, L, S) if (bf) run<true, U, M, L, S>(n, x, y, b, nvec, bmask, g); else run<false, U, M, L, S>(n, x, y, b, nvec, bmask, g);
            RUN(1, 0, false, false) RUN(1, 0, true, false) RUN(1, 0, false, true) RUN(1, 0, true, true)
            RUN(2, 0, false, false) RUN(2, 1, false, false) RUN(2, 1, true, true)
            RUN(4, 0, false, false) RUN(4, 0, true, false) RUN(4, 1, false, false) RUN(4, 1, true, false) RUN(4, 1, false, true) RUN(4, 1, true, true)
        }
        hipFree(x); hipFree(y); hipFree(b);
    }
    return 0;
}
