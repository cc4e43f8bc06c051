/* Plain-C client of libtensoir_hip.so: no Python, no torch, no C++.  Shows that the boundary declared in
 * include/tensoir_hip.h is a C ABI: device buffers come from the HIP runtime's C API, the descriptor struct lives on
 * the host, every call returns an int.  It packs a tiny random VM density field, evaluates
 * compute_densityfeature (models/tensoRF_rotated_lights.py:95-110) at a few points through tir_vm_density_fwd and
 * checks the result against a scalar C restatement of the bilinear / linear interpolation.
 * Built by __graft_entry__.build() (gcc); run by tests/test_gpu_c_abi.py on the GPU box. */
#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "tensoir_hip.h"

#define CK(x) do { int rc__ = (int)(x); if (rc__ != 0) { fprintf(stderr, "%s failed: %d\n", #x, rc__); return 2; } } while (0)

static float frand(unsigned* s) { *s = *s * 1664525u + 1013904223u; return ((*s >> 8) & 0xFFFF) / 65535.0f - 0.5f; }

/* F.grid_sample(..., align_corners=True, padding zeros) along one axis */
static void tap(float x, int n, int* i0, int* i1, float* w0, float* w1) {
    float ix = ((x + 1.0f) * 0.5f) * (float)(n - 1), f0 = floorf(ix), t = ix - f0;
    int a = (int)f0, b = a + 1;
    *w0 = (a >= 0 && a < n) ? 1.0f - t : 0.0f; *w1 = (b >= 0 && b < n) ? t : 0.0f;
    *i0 = a < 0 ? 0 : (a >= n ? n - 1 : a); *i1 = b < 0 ? 0 : (b >= n ? n - 1 : b);
}

int main(void) {
    enum { C = 16, N = 257 };
    const int grid[3] = {11, 13, 9};                      /* x, y, z */
    const int mat0[3] = {0, 0, 1}, mat1[3] = {1, 2, 2}, vec[3] = {2, 1, 0};
    unsigned seed = 12345u;
    float *hp[3], *hl[3], *dp_src[3], *dl_src[3], *dp[3], *dl[3];
    TirField f;
    int i, c, n;
    if (tir_version() != TIR_VERSION) { fprintf(stderr, "version mismatch\n"); return 2; }
    CK(tir_device_check());
    memset(&f, 0, sizeof f);
    for (i = 0; i < 3; ++i) {
        const int H = grid[mat1[i]], W = grid[mat0[i]], R = grid[vec[i]];
        size_t np = (size_t)C * H * W, nl = (size_t)C * R;
        hp[i] = (float*)malloc(np * 4); hl[i] = (float*)malloc(nl * 4);
        for (n = 0; n < (int)np; ++n) hp[i][n] = frand(&seed);
        for (n = 0; n < (int)nl; ++n) hl[i][n] = frand(&seed);
        CK(hipMalloc((void**)&dp_src[i], np * 4)); CK(hipMalloc((void**)&dl_src[i], nl * 4));
        CK(hipMalloc((void**)&dp[i], np * 4)); CK(hipMalloc((void**)&dl[i], nl * 4));
        CK(hipMemcpy(dp_src[i], hp[i], np * 4, hipMemcpyHostToDevice));
        CK(hipMemcpy(dl_src[i], hl[i], nl * 4, hipMemcpyHostToDevice));
        CK(tir_pack_plane(dp_src[i], dp[i], C, H, W, NULL));      /* [C,H,W] -> [H,W,C] */
        CK(tir_pack_plane(dl_src[i], dl[i], C, R, 1, NULL));      /* [C,R,1] -> [R,C]   */
        f.dplane[i] = dp[i]; f.dline[i] = dl[i];
        f.grid[i] = grid[i];
    }
    f.n_dcomp = C; f.act = 0; f.density_shift = -10.0f;
    {
        float hx[3 * N], hout[N], *dx, *dout;
        double worst = 0.0;
        for (n = 0; n < 3 * N; ++n) hx[n] = 2.2f * frand(&seed);           /* some points outside [-1,1]: zero padding */
        CK(hipMalloc((void**)&dx, sizeof hx)); CK(hipMalloc((void**)&dout, sizeof hout));
        CK(hipMemcpy(dx, hx, sizeof hx, hipMemcpyHostToDevice));
        CK(tir_vm_density_fwd(&f, dx, dout, NULL, N, NULL));
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(hout, dout, sizeof hout, hipMemcpyDeviceToHost));
        for (n = 0; n < N; ++n) {
            double ref = 0.0;
            for (i = 0; i < 3; ++i) {
                const int H = grid[mat1[i]], W = grid[mat0[i]], R = grid[vec[i]];
                int x0, x1, y0, y1, l0, l1; float wx0, wx1, wy0, wy1, wl0, wl1;
                tap(hx[3 * n + mat0[i]], W, &x0, &x1, &wx0, &wx1);
                tap(hx[3 * n + mat1[i]], H, &y0, &y1, &wy0, &wy1);
                tap(hx[3 * n + vec[i]], R, &l0, &l1, &wl0, &wl1);
                for (c = 0; c < C; ++c) {
                    const float* P = hp[i] + (size_t)c * H * W;
                    const float* L = hl[i] + (size_t)c * R;
                    double pv = (double)P[y0 * W + x0] * wx0 * wy0 + (double)P[y0 * W + x1] * wx1 * wy0 +
                                (double)P[y1 * W + x0] * wx0 * wy1 + (double)P[y1 * W + x1] * wx1 * wy1;
                    ref += pv * ((double)L[l0] * wl0 + (double)L[l1] * wl1);
                }
            }
            if (fabs(ref - hout[n]) > worst) worst = fabs(ref - hout[n]);
        }
        printf("c-abi demo: %d points, max |hip - c| = %.3e\n", N, worst);
        if (!(worst < 1e-5)) return 1;
        /* error path: a NULL descriptor must come back as an error code, not a crash */
        if (tir_vm_density_fwd(NULL, dx, dout, NULL, N, NULL) != TIR_ERR_ARG) return 1;
        printf("c-abi demo ok (%s)\n", tir_error_string(TIR_ERR_ARG));
    }
    return 0;
}
