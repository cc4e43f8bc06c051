// Micro-benchmark: fp32 global atomic-add scatter throughput on gfx950 as a function of how one wave instruction's
// 64 dwords are spread over cache lines.  Build: hipcc --offload-arch=gfx950 -O3 tools/atomic_bench.hip -o /tmp/atomic_bench
// PATTERN p: every wave instruction touches 64/p random 64*... segments of p contiguous dwords (p = 1,4,16,32,64);
// pattern 4s = the k_vm_app_bwd layout: 16 segments of 64 B, 4 lanes each at a 16-B stride.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__device__ __forceinline__ void atomic_add_f32(float* p, float v) {
    __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// per: contiguous dwords per segment; strided: 1 -> lanes of a segment sit 16 B apart (4 lanes cover a 64-B run)
template <int PER, int STRIDED>
__global__ void __launch_bounds__(256) k_scatter(float* buf, unsigned n_seg_mask, int iters, int local) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    const int seg = lane / PER, in = lane % PER;
    for (int it = 0; it < iters; ++it) {
        unsigned h = hash((wave * 8191u + it) * 64u + seg);
        if (local) h = (h & 1023u) + ((wave & 255u) << 10);          // small per-wave working set
        const unsigned base = (h & n_seg_mask) * 64u;                 // 256-B aligned slots, in floats
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const unsigned off = STRIDED ? (in * 4 + u) : (u * PER + in) % 64;
            atomic_add_f32(buf + base + off, 1.0f);
        }
    }
}

template <int PER, int STRIDED>
static void run(const char* name, float* buf, unsigned mask, int local) {
    const int blocks = 2048, iters = 64;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_scatter<PER, STRIDED>), dim3(blocks), dim3(256), 0, 0, buf, mask, 4, local);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_scatter<PER, STRIDED>), dim3(blocks), dim3(256), 0, 0, buf, mask, iters, local);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double atomics = (double)blocks * 256 * iters * 4;
    printf("%-28s local=%d  %8.3f ms  %7.2f G atomics/s  (%.2f G wave-instr-segments/s)\n", name, local, ms,
           atomics / ms / 1e6, atomics / PER / ms / 1e6);
}

int main() {
    const size_t floats = (size_t)1 << 24;        // 64 MB
    float* buf;
    hipMalloc(&buf, floats * sizeof(float));
    hipMemset(buf, 0, floats * sizeof(float));
    const unsigned mask = (unsigned)(floats / 64 - 1);
    for (int local = 0; local < 2; ++local) {
        run<1, 0>("1 dword x 64 segments", buf, mask, local);
        run<4, 1>("4 strided dwords x 16 (app)", buf, mask, local);
        run<4, 0>("4 contiguous dwords x 16", buf, mask, local);
        run<16, 0>("16 contiguous dwords x 4", buf, mask, local);
        run<32, 0>("32 contiguous dwords x 2", buf, mask, local);
        run<64, 0>("64 contiguous dwords x 1", buf, mask, local);
    }
    hipFree(buf);
    return 0;
}
