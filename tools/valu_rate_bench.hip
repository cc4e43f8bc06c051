// How many cycles does one wave64 VALU instruction take on a gfx950 SIMD, and can several waves share the pipe?
// 8 independent v_fma_f32 chains per lane (no dependency stalls), W waves per SIMD (blocks of 256 threads = one wave per SIMD,
// W blocks per CU by LDS footprint), fixed instruction count per wave.  Prints SIMD cycles per wave-instruction.
// Build: hipcc --offload-arch=gfx950 -O3 tools/valu_rate_bench.hip -o gpurun_scratch/valu_rate_bench
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void __launch_bounds__(256) k_fma(float* out, int iters, float a, float b) {
    extern __shared__ float pad[];
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            x0 = fmaf(x0, a, b); x1 = fmaf(x1, a, b); x2 = fmaf(x2, a, b); x3 = fmaf(x3, a, b);
            x4 = fmaf(x4, a, b); x5 = fmaf(x5, a, b); x6 = fmaf(x6, a, b); x7 = fmaf(x7, a, b);
        }
    }
    float s = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    if (s == 12345.678f) out[0] = s + pad[0];
}

int main() {
    float* out; (void)hipMalloc(&out, 64);
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    const int iters = 2000;
    for (int w = 1; w <= 8; w *= 2) {
        // w blocks per CU: each block asks for 160 KB / w of LDS (minus a little) so exactly w fit
        size_t lds = (size_t)(160 * 1024 / w) - 1024;
        if (lds > 64 * 1024) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fma), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipLaunchKernelGGL(k_fma, dim3(cus * w), dim3(256), lds, 0, out, 10, 1.0001f, 0.5f);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(k_fma, dim3(cus * w), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        const double inst_per_wave = (double)iters * 16 * 8;
        // SIMD-time per wave-instruction, assuming 2.4 GHz (the kernel is light: little throttling)
        printf("waves/SIMD=%d: %.3f ms, %.2f ns per wave-instruction per SIMD = %.2f cycles @2.4GHz, %.1f TFLOP/s\n", w, ms,
               ms * 1e6 / (inst_per_wave * w), ms * 1e6 / (inst_per_wave * w) * 2.4,
               (double)cus * 4 * w * inst_per_wave * 64 * 2 / (ms * 1e-3) / 1e12);
    }
    return 0;
}
