// Do the matrix pipe and the VALU of ONE SIMD overlap across waves on gfx950?  A 512-thread workgroup puts two waves on every SIMD
// (wave w -> SIMD w % 4).  Waves 0-3 run a loop of independent v_mfma_f32_32x32x16_f16, waves 4-7 a loop of independent v_fma_f32
// chains; modes: matrix waves alone, VALU waves alone, both together, and ONE wave per SIMD doing both in the same loop body.
// If the pipes overlap, "both" takes max(matrix, valu); if the SIMD executes one or the other, it takes the sum.
// (Question behind it: DESIGN 8.1 -- in both fused indirect kernels VALU-issue and matrix-busy fractions ADD UP to ~0.95.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o gpurun_scratch/mfma_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void __launch_bounds__(512) k(float* out, int mode, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    const bool do_m = (mode == 0 || mode == 2) ? wave < 4 : (mode == 3 ? wave < 4 : false);
    const bool do_v = (mode == 1 || mode == 2) ? wave >= 4 : (mode == 3 ? wave < 4 : false);
    f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
    float v0 = threadIdx.x, v1 = v0 + 1, v2 = v0 + 2, v3 = v0 + 3, v4 = v0 + 4, v5 = v0 + 5, v6 = v0 + 6, v7 = v0 + 7;
    const float m = 1.0001f, d = 0.5f;
    if (mode == 3) {                       // one wave per SIMD, both kinds of work in one loop body (8 FMAs per 4 matrix instructions x ratio)
        if (wave < 4) {
            const int per = n_valu / n_mfma;          // VALU groups per matrix group
            for (int i = 0; i < n_mfma; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
                for (int j = 0; j < per; ++j) {
                    v0 = fmaf(v0, m, d); v1 = fmaf(v1, m, d); v2 = fmaf(v2, m, d); v3 = fmaf(v3, m, d);
                    v4 = fmaf(v4, m, d); v5 = fmaf(v5, m, d); v6 = fmaf(v6, m, d); v7 = fmaf(v7, m, d);
                }
            }
        }
    } else {
        if (do_m)
            for (int i = 0; i < n_mfma; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
            }
        if (do_v)
            for (int i = 0; i < n_valu; ++i) {
                v0 = fmaf(v0, m, d); v1 = fmaf(v1, m, d); v2 = fmaf(v2, m, d); v3 = fmaf(v3, m, d);
                v4 = fmaf(v4, m, d); v5 = fmaf(v5, m, d); v6 = fmaf(v6, m, d); v7 = fmaf(v7, m, d);
            }
    }
    float s = v0 + v1 + v2 + v3 + v4 + v5 + v6 + v7;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

int main() {
    float* out;
    hipMalloc(&out, 256 * 512 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int n_mfma = 4000;                 // x 4 matrix instructions x 32 cycles = 512 k cycles
    for (int n_valu : {8000, 16000, 32000}) {   // x 8 FMAs x 4 cycles = 256 k / 512 k / 1024 k cycles
        float t[4];
        for (int mode = 0; mode < 4; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, out, mode, n_mfma, n_valu);
                hipEventRecord(e1); hipEventSynchronize(e1);
                hipEventElapsedTime(&t[mode], e0, e1);
            }
        }
        printf("n_mfma %d x4, n_valu %d x8: matrix waves alone %.3f ms, VALU waves alone %.3f ms, both (two waves per SIMD) %.3f ms, "
               "one wave doing both %.3f ms   [sum %.3f, max %.3f]\n", n_mfma, n_valu, t[0], t[1], t[2], t[3], t[0] + t[1], t[0] > t[1] ? t[0] : t[1]);
    }
    return 0;
}
