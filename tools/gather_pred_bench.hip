// Micro-benchmark: does a gather instruction with predicated-off lane groups cost less?  Mode A of gather_bench.hip
// (4 lanes x float4 per sample, 16 samples per load instruction, 4 taps x 3 instructions per sample) where every
// (sample, tap) is loaded only with probability keep/8, decided per sample -- i.e. divergent inside the instruction.
// If the time follows the number of ACTIVE quads, skipping redundant taps (samples that share cells with their
// predecessor on the ray) pays even though the instruction count stays the same.
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_pred_bench.hip -o gpurun_scratch/gather_pred_bench
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ tab, unsigned cells, int iters, int coherent, int keep,
                                                float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    const int sl = lane / 4, c = lane % 4;
    for (int it = 0; it < iters; ++it) {
        const unsigned sidx = (wave * 4096u + it) * 16u + sl;
        unsigned cell = coherent ? (hash(wave) + (sidx & 0xffffu) / 2u) % (cells - 400u) : hash(sidx) % (cells - 400u);
        const unsigned offs[4] = {cell, cell + 1u, cell + 300u, cell + 301u};
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const bool on = (int)(hash(sidx * 4u + t + 77u) & 7u) < keep;
            const float* p = tab + (size_t)offs[t] * 48u;
            if (on) {
#pragma unroll
                for (int q = 0; q < 3; ++q) { const float4 v = *reinterpret_cast<const float4*>(p + 16 * q + 4 * c); acc += v.x + v.y + v.z + v.w; }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

int main() {
    const unsigned cells = 300u * 300u * 3u;
    float *tab, *out;
    (void)hipMalloc(&tab, (size_t)cells * 48 * sizeof(float));
    (void)hipMalloc(&out, 64);
    (void)hipMemset(tab, 0, (size_t)cells * 48 * sizeof(float));
    const int blocks = 2048, iters = 96;
    for (int coh = 0; coh < 2; ++coh)
        for (int keep = 8; keep >= 1; keep >>= 1) {
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, cells, 2, coh, keep, out);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(k_gather, dim3(blocks), dim3(256), 0, 0, tab, cells, iters, coh, keep, out);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms = 0;
            (void)hipEventElapsedTime(&ms, e0, e1);
            const double slots = (double)blocks * 4 * iters * 16 * 4;
            printf("coherent=%d keep=%d/8: %8.3f ms   %7.2f G tap-slots/s   %7.2f G active taps/s\n", coh, keep, ms,
                   slots / ms / 1e6, slots * keep / 8 / ms / 1e6);
        }
    return 0;
}
