// Micro-benchmark: how fast can a CU pull 192-B appearance taps (48 fp32 channels of one plane cell) out of L2 / MALL
// as a function of how the lanes of one load instruction are spread over the tap?
//   A  4 lanes x float4 per sample, 3 instructions per tap  (16 samples x 64 B per instruction)   -- k_vm_app_mfma today
//   B  12 of 16 lanes x float4 per sample, 1 instruction per tap (4 samples x 192 B per instruction)
//   C  16 lanes x dword per sample, 3 instructions per tap (4 samples x 64 B per instruction)     -- k_vm_app_bwd today
//   D  8 lanes x float4, 2 instructions: 128 B + 64 B (8 samples per instruction)
// Build: hipcc --offload-arch=gfx950 -O3 tools/gather_bench.hip -o tools/gather_bench.bin
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ __forceinline__ unsigned hash(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
    return x;
}

// cells: number of 192-B cells in the table.  coherent != 0: consecutive samples walk neighbouring cells (ray-like)
template <int MODE>
__global__ void __launch_bounds__(256) k_gather(const float* __restrict__ tab, unsigned cells, int iters, int coherent,
                                                float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    float acc = 0.f;
    constexpr int LPS = MODE == 0 ? 4 : (MODE == 1 ? 16 : (MODE == 2 ? 16 : 8));       // lanes per sample
    constexpr int SPI = 64 / LPS;                                                        // samples per instruction
    const int sl = lane / LPS, c = lane % LPS;
    for (int it = 0; it < iters; ++it) {
        // 16 samples per wave iteration in every mode, 4 taps each (a bilinear footprint: cell, +1, +W, +W+1)
        for (int sb = 0; sb < 16; sb += SPI) {
            const unsigned sidx = (wave * 4096u + it) * 16u + sb + sl;
            unsigned cell = coherent ? (hash(wave) + (sidx & 0xffffu) / 2u) % (cells - 400u) : hash(sidx) % (cells - 400u);
            const unsigned offs[4] = {cell, cell + 1u, cell + 300u, cell + 301u};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const float* p = tab + (size_t)offs[t] * 48u;
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) { const float4 v = *reinterpret_cast<const float4*>(p + 16 * q + 4 * c); acc += v.x + v.y + v.z + v.w; }
                } else if (MODE == 1) {
                    if (c < 12) { const float4 v = *reinterpret_cast<const float4*>(p + 4 * c); acc += v.x + v.y + v.z + v.w; }
                } else if (MODE == 2) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) acc += p[16 * q + c];
                } else {
                    { const float4 v = *reinterpret_cast<const float4*>(p + 4 * c); acc += v.x + v.y + v.z + v.w; }
                    if (c < 4) { const float4 v = *reinterpret_cast<const float4*>(p + 32 + 4 * c); acc += v.x + v.y + v.z + v.w; }
                }
            }
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE>
static void run(const char* name, const float* tab, unsigned cells, int coherent, float* out) {
    const int blocks = 2048, iters = 24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, tab, cells, 2, coherent, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather<MODE>), dim3(blocks), dim3(256), 0, 0, tab, cells, iters, coherent, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double taps = (double)blocks * 4 * iters * 16 * 4;
    printf("%-44s coherent=%d %8.3f ms  %7.2f G taps/s  %7.2f TB/s of tap bytes\n", name, coherent, ms, taps / ms / 1e6,
           taps * 192 / ms / 1e9);
}

// Round 5: the same question for the fp16 shadow taps of the fused indirect kernel (96 B per tap, 48 halves):
//   E  2 lanes x 16 B per sample, 3 instructions per tap (32 samples x 32 B per instruction)   -- k_indirect_fused today
//   F  6 of 8 lanes x 16 B per sample, 1 instruction per tap (8 samples x 96 B per instruction)
// 32 samples per wave iteration, 4 taps each; cells of 96 bytes.
template <int MODE>
__global__ void __launch_bounds__(256) k_gather_h(const uint4* __restrict__ tab, unsigned cells, int iters, int coherent,
                                                  float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const unsigned wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
    unsigned acc = 0;
    //   G  4 lanes x 16 B (64 B of the tap) + lanes 0-1 x 16 B (the remaining 32 B): 16 samples per instruction
    //   H  8 lanes x 12 B (global_load_dwordx3): the whole tap per instruction, every lane active, 8 samples per instruction
    constexpr int LPS = MODE == 0 ? 2 : (MODE == 2 ? 4 : 8), SPI = 64 / LPS;
    const int sl = lane / LPS, c = lane % LPS;
    for (int it = 0; it < iters; ++it) {
        for (int sb = 0; sb < 32; sb += SPI) {
            const unsigned sidx = (wave * 4096u + it) * 32u + sb + sl;
            unsigned cell = coherent ? (hash(wave) + (sidx & 0xffffu) / 2u) % (cells - 400u) : hash(sidx) % (cells - 400u);
            const unsigned offs[4] = {cell, cell + 1u, cell + 300u, cell + 301u};
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const uint4* p = tab + (size_t)offs[t] * 6u;          // 6 x 16 B per cell
                if (MODE == 0) {
#pragma unroll
                    for (int q = 0; q < 3; ++q) { const uint4 v = p[2 * q + c]; acc += v.x ^ v.y ^ v.z ^ v.w; }
                } else if (MODE == 1) {
                    if (c < 6) { const uint4 v = p[c]; acc += v.x ^ v.y ^ v.z ^ v.w; }
                } else if (MODE == 2) {
                    { const uint4 v = p[c]; acc += v.x ^ v.y ^ v.z ^ v.w; }
                    if (c < 2) { const uint4 v = p[4 + c]; acc += v.x ^ v.y ^ v.z ^ v.w; }
                } else {
                    const uint3 v = *reinterpret_cast<const uint3*>(reinterpret_cast<const unsigned*>(p) + 3 * c);
                    acc += v.x ^ v.y ^ v.z;
                }
            }
        }
    }
    if (acc == 0x12345678u) out[0] = (float)acc;
}

template <int MODE>
static void run_h(const char* name, const uint4* tab, unsigned cells, int coherent, float* out) {
    const int blocks = 2048, iters = 24;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k_gather_h<MODE>), dim3(blocks), dim3(256), 0, 0, tab, cells, 2, coherent, out);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((k_gather_h<MODE>), dim3(blocks), dim3(256), 0, 0, tab, cells, iters, coherent, out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double taps = (double)blocks * 4 * iters * 32 * 4;
    printf("%-44s coherent=%d %8.3f ms  %7.2f G taps/s  %7.2f TB/s of tap bytes\n", name, coherent, ms, taps / ms / 1e6,
           taps * 96 / ms / 1e9);
}

int main() {
    const unsigned cells = 300u * 300u * 3u;        // three 300x300 planes of 48 channels = 52 MB
    float *tab, *out;
    (void)hipMalloc(&tab, (size_t)cells * 48 * sizeof(float));
    (void)hipMalloc(&out, 64);
    (void)hipMemset(tab, 0, (size_t)cells * 48 * sizeof(float));
    for (int coh = 0; coh < 2; ++coh) {
        run<0>("A  4 lanes x float4, 3 instr/tap", tab, cells, coh, out);
        run<1>("B  12/16 lanes x float4, 1 instr/tap", tab, cells, coh, out);
        run<2>("C  16 lanes x dword, 3 instr/tap", tab, cells, coh, out);
        run<3>("D  8 lanes x float4, 128 B + 64 B", tab, cells, coh, out);
    }
    for (int coh = 0; coh < 2; ++coh) {            // fp16 taps: the same table read as 96-B cells (twice as many)
        run_h<0>("E  fp16: 2 lanes x 16 B, 3 instr/tap", reinterpret_cast<const uint4*>(tab), cells * 2u, coh, out);
        run_h<1>("F  fp16: 6/8 lanes x 16 B, 1 instr/tap", reinterpret_cast<const uint4*>(tab), cells * 2u, coh, out);
        run_h<2>("G  fp16: 4 lanes x 16 B + 2 lanes x 16 B", reinterpret_cast<const uint4*>(tab), cells * 2u, coh, out);
        run_h<3>("H  fp16: 8 lanes x 12 B, 1 instr/tap", reinterpret_cast<const uint4*>(tab), cells * 2u, coh, out);
    }
    (void)hipFree(tab); (void)hipFree(out);
    return 0;
}
