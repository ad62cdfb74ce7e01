/* CPU oracle (plain C) of upfirdn2d -- TEST INFRASTRUCTURE ONLY, never linked into the product.
 *
 * Restates the gather form of the reference's native kernel (thirdparty/stylegan3_ops/ops/upfirdn2d.cu:23-86,
 * floor_div at :14-18) for dense NCHW fp32 tensors:
 *   mid = out*down + up - 1 - pad0;  in0 = floor(mid / up);  k0 = (in0 + 1)*up - mid - 1;
 *   y = gain * sum_j sum_i x[in0y + j][in0x + i] * F[k0y + j*upy][k0x + i*upx],  F = f flipped unless `flip`
 * with fp32 accumulation in (ky, kx) order (upfirdn2d.cu:64-75).  Used by tests/ as a second, independent statement
 * next to oracle/upfirdn2d.py, and by tools/bench_kernels.py as the kernel-level CPU baseline.
 * Build: make -C oracle   (gcc -O2 -fopenmp -shared)
 */
#include <stdint.h>

static int floor_div(int a, int b) { int q = a / b; return (a % b != 0 && a < 0) ? q - 1 : q; }

int upfirdn2d_oracle_f32(const float* x, const float* f, float* y,
                         int N, int C, int H, int W, int fh, int fw, int OH, int OW,
                         int upx, int upy, int downx, int downy, int padx0, int pady0, int flip, float gain) {
    long planes = (long)N * C;
#pragma omp parallel for schedule(static)
    for (long pl = 0; pl < planes; pl++) {
        const float* xp = x + pl * H * W;
        float* yp = y + pl * OH * OW;
        for (int oy = 0; oy < OH; oy++) {
            int midy = oy * downy + upy - 1 - pady0;
            int iny0 = floor_div(midy, upy);
            int ky0 = (iny0 + 1) * upy - midy - 1;
            for (int ox = 0; ox < OW; ox++) {
                int midx = ox * downx + upx - 1 - padx0;
                int inx0 = floor_div(midx, upx);
                int kx0 = (inx0 + 1) * upx - midx - 1;
                float v = 0.f;
                for (int ky = ky0, iy = iny0; ky < fh; ky += upy, iy++) {
                    if (iy < 0 || iy >= H) continue;
                    for (int kx = kx0, ix = inx0; kx < fw; kx += upx, ix++) {
                        if (ix < 0 || ix >= W) continue;
                        float tap = flip ? f[ky * fw + kx] : f[(fh - 1 - ky) * fw + (fw - 1 - kx)];
                        v += xp[iy * W + ix] * tap;
                    }
                }
                yp[oy * OW + ox] = v * gain;
            }
        }
    }
    return 0;
}
