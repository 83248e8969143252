/* A plain-C client of the C ABI (include/xmcgan_hip.h): what a non-Python binding of the reference's operators would
 * do.  Without arguments: checks the ABI version (no GPU needed).  With "gpu": runs xmc_conv2d_nhwc (float32, 3x3) on
 * device buffers it allocates through the HIP runtime and compares with a direct CPU convolution. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "xmcgan_hip.h"

/* minimal HIP runtime prototypes (the client needs nothing else from HIP) */
typedef int hipError_t;
hipError_t hipMalloc(void** p, size_t n);
hipError_t hipFree(void* p);
hipError_t hipMemcpy(void* dst, const void* src, size_t n, int kind);
hipError_t hipDeviceSynchronize(void);
enum { H2D = 1, D2H = 2 };

int main(int argc, char** argv) {
    if (xmc_abi_version() != XMC_ABI_VERSION) {
        fprintf(stderr, "ABI version mismatch: library %d, header %d\n", xmc_abi_version(), XMC_ABI_VERSION);
        return 2;
    }
    printf("abi %d\n", xmc_abi_version());
    if (argc < 2 || strcmp(argv[1], "gpu") != 0) return 0;

    const int N = 2, H = 8, W = 8, CI = 16, CO = 24, KS = 3;
    const size_t nx = (size_t)N * H * W * CI, nw = (size_t)CO * 9 * CI, ny = (size_t)N * H * W * CO;
    float *x = malloc(nx * 4), *w = malloc(nw * 4), *b = malloc(CO * 4), *y = malloc(ny * 4), *ref = malloc(ny * 4);
    unsigned s = 12345u;
    for (size_t i = 0; i < nx; ++i) { s = s * 1664525u + 1013904223u; x[i] = (float)(s >> 8) / 8388608.f - 1.f; }
    for (size_t i = 0; i < nw; ++i) { s = s * 1664525u + 1013904223u; w[i] = ((float)(s >> 8) / 8388608.f - 1.f) * 0.1f; }
    for (int i = 0; i < CO; ++i) b[i] = 0.01f * i;
    for (int n = 0; n < N; ++n) for (int yy = 0; yy < H; ++yy) for (int xx = 0; xx < W; ++xx) for (int o = 0; o < CO; ++o) {
        double acc = b[o];
        for (int t = 0; t < 9; ++t) {
            const int sy = yy + t / 3 - 1, sx = xx + t % 3 - 1;
            if (sy < 0 || sy >= H || sx < 0 || sx >= W) continue;
            for (int c = 0; c < CI; ++c) acc += (double)x[((n * H + sy) * W + sx) * CI + c] * w[(o * 9 + t) * CI + c];
        }
        ref[((n * H + yy) * W + xx) * CO + o] = (float)acc;
    }
    void *dx, *dw, *db, *dy;
    if (hipMalloc(&dx, nx * 4) || hipMalloc(&dw, nw * 4) || hipMalloc(&db, CO * 4) || hipMalloc(&dy, ny * 4)) return 3;
    hipMemcpy(dx, x, nx * 4, H2D); hipMemcpy(dw, w, nw * 4, H2D); hipMemcpy(db, b, CO * 4, H2D);
    xmc_conv_desc d;
    memset(&d, 0, sizeof d);
    d.n = N; d.hi = H; d.wi = W; d.cin = CI; d.cout = CO; d.ks = KS; d.dtype = XMC_F32; d.alpha = 1.f; d.res_scale = 1.f;
    const int rc = xmc_conv2d_nhwc(&d, dx, dw, (const float*)db, NULL, NULL, dy, NULL);   /* NULL = default stream */
    if (rc != XMC_OK) { fprintf(stderr, "xmc_conv2d_nhwc rc=%d\n", rc); return 4; }
    hipDeviceSynchronize();
    hipMemcpy(y, dy, ny * 4, D2H);
    double worst = 0;
    for (size_t i = 0; i < ny; ++i) { const double e = fabs((double)y[i] - ref[i]); if (e > worst) worst = e; }
    printf("conv max abs err %.3e\n", worst);
    hipFree(dx); hipFree(dw); hipFree(db); hipFree(dy);
    return worst < 1e-4 ? 0 : 5;
}
