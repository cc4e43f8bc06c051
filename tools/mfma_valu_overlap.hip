// Do the matrix pipe and the VALU of ONE SIMD overlap across waves on gfx950?  A 512-thread workgroup puts two waves on every SIMD
// (wave w -> SIMD w % 4).  Waves 0-3 run a loop of four independent matrix instructions, waves 4-7 a loop of eight independent VALU
// instructions; per (matrix shape, VALU kind): matrix waves alone, VALU waves alone, both together.  If the pipes overlap, "both" takes
// max(matrix, valu); if the SIMD executes one or the other, it takes the sum.  hidden = (sum - both) / min: the share of the shorter
// stream that ran under the longer one.
// (Question behind it: DESIGN 8.1 -- in both fused indirect kernels VALU-issue and matrix-busy fractions ADD UP to ~0.95.)
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_valu_overlap.hip -o gpurun_scratch/mfma_valu_overlap ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 h2 __attribute__((ext_vector_type(2)));

// MK: 0 = v_mfma_f32_32x32x16_f16 (32 cycles), 1 = v_mfma_f32_16x16x32_f16 (16 cycles, issued twice as often), 2 = v_mfma_f32_32x32x16_bf16,
//     3 = v_mfma_f32_32x32x2_f32 (exact fp32, 64 cycles per 16 passes ... issued half as often)
// VK: 0 = v_fma_f32, 1 = v_pk_fma_f16, 2 = v_pk_fma_f32, 3 = v_sin_f32 (transcendental unit), 4 = v_cvt_pkrtz_f16_f32,
//     5 = v_pk_fma_f16 with SIXTEEN independent chains (is VK 1's overlap only the matrix pipe filling dependency stalls?), 6 = v_fma_f32 x 16,
//     7 = v_mad_u32_u24 (integer), 8 = v_fma_mix_f32 (fp16 operand read in place, fp32 arithmetic)
template <int MK, int VK>
__global__ void __launch_bounds__(512) k(float* out, int mode, int n_mfma, int n_valu) {
    const int wave = threadIdx.x >> 6;
    float s = 0.f;
    if ((mode == 0 || mode == 2) && wave < 4) {
        f16x8 a, b;
        for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(i * 0.5f - threadIdx.x * 0.002f); }
        if constexpr (MK == 0 || MK == 2 || MK == 3) {
            f32x16 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            const bf16x8 ab = __builtin_bit_cast(bf16x8, a), bb = __builtin_bit_cast(bf16x8, b);
            const int n = MK == 3 ? n_mfma / 2 : n_mfma;
            for (int i = 0; i < n; ++i) {
                if constexpr (MK == 0) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c3, 0, 0, 0);
                } else if constexpr (MK == 2) {
                    c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ab, bb, c3, 0, 0, 0);
                } else {
                    const float x = (float)a[0], y = (float)b[0];
                    c0 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c1, 0, 0, 0);
                    c2 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_32x32x2f32(x, y, c3, 0, 0, 0);
                }
            }
            for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        } else {
            f32x4 c0 = {}, c1 = {}, c2 = {}, c3 = {};
            for (int i = 0; i < 2 * n_mfma; ++i) {
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c0, 0, 0, 0); c1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c2, 0, 0, 0); c3 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c3, 0, 0, 0);
            }
            for (int i = 0; i < 4; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
        }
    }
    if ((mode == 1 || mode == 2) && wave >= 4) {
        if constexpr (VK == 0 || VK == 3) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = VK == 0 ? fmaf(v[j], 1.0001f, 0.5f) : __builtin_amdgcn_sinf(v[j]);
            for (int j = 0; j < 8; ++j) s += v[j];
        } else if constexpr (VK == 1) {
            h2 v[8];
            for (int j = 0; j < 8; ++j) v[j] = h2{(_Float16)(threadIdx.x + j), (_Float16)(j)};
            const h2 m = {(_Float16)1.001f, (_Float16)0.999f}, d = {(_Float16)0.5f, (_Float16)0.25f};
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], m, d);
            for (int j = 0; j < 8; ++j) s += (float)v[j].x + (float)v[j].y;
        } else if constexpr (VK == 5) {
            h2 v[16];
            for (int j = 0; j < 16; ++j) v[j] = h2{(_Float16)(threadIdx.x + j), (_Float16)(j)};
            const h2 m = {(_Float16)1.001f, (_Float16)0.999f}, d = {(_Float16)0.5f, (_Float16)0.25f};
            for (int i = 0; i < n_valu / 2; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = __builtin_elementwise_fma(v[j], m, d);
            for (int j = 0; j < 16; ++j) s += (float)v[j].x + (float)v[j].y;
        } else if constexpr (VK == 6) {
            float v[16];
            for (int j = 0; j < 16; ++j) v[j] = threadIdx.x * 0.01f + j;
            for (int i = 0; i < n_valu / 2; ++i)
#pragma unroll
                for (int j = 0; j < 16; ++j) v[j] = fmaf(v[j], 1.0001f, 0.5f);
            for (int j = 0; j < 16; ++j) s += v[j];
        } else if constexpr (VK == 7) {
            unsigned v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x + j;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(v[j]) : "v"(v[j]), "v"(3u), "v"(7u));
            for (int j = 0; j < 8; ++j) s += (float)v[j];
        } else if constexpr (VK == 8) {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
            const unsigned hh = 0x3c003c00u;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(v[j]) : "v"(hh), "v"(v[j]), "v"(0.5f));
            for (int j = 0; j < 8; ++j) s += v[j];
        } else if constexpr (VK == 2) {
            f32x2 v[8];
            for (int j = 0; j < 8; ++j) v[j] = f32x2{threadIdx.x + (float)j, (float)j};
            const f32x2 m = {1.0001f, 0.9999f}, d = {0.5f, 0.25f};
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = __builtin_elementwise_fma(v[j], m, d);
            for (int j = 0; j < 8; ++j) s += v[j].x + v[j].y;
        } else {
            float v[8];
            for (int j = 0; j < 8; ++j) v[j] = threadIdx.x * 0.01f + j;
            for (int i = 0; i < n_valu; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const auto p = __builtin_amdgcn_cvt_pkrtz(v[j], v[(j + 1) & 7]);
                    v[j] = __builtin_bit_cast(float, p);
                }
            for (int j = 0; j < 8; ++j) s += v[j];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

static hipEvent_t e0, e1;
template <int MK, int VK>
static void run(float* out, const char* mname, const char* vname, int n_mfma, int n_valu) {
    float t[3];
    for (int mode = 0; mode < 3; ++mode)
        for (int rep = 0; rep < 3; ++rep) {
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL((k<MK, VK>), dim3(256), dim3(512), 0, 0, out, mode, n_mfma, n_valu);
            (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
            (void)hipEventElapsedTime(&t[mode], e0, e1);
        }
    const float sum = t[0] + t[1], mx = t[0] > t[1] ? t[0] : t[1], mn = t[0] < t[1] ? t[0] : t[1];
    printf("%-28s + %-20s: matrix alone %.3f  VALU alone %.3f  both %.3f ms   [sum %.3f, max %.3f]  hidden %.2f\n", mname, vname, t[0], t[1], t[2], sum, mx,
           (sum - t[2]) / mn);
}

int main() {
    float* out;
    (void)hipMalloc(&out, 256 * 512 * 4);
    (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int nm = 4000, nv = 12000;
    run<0, 0>(out, "v_mfma_f32_32x32x16_f16", "v_fma_f32", nm, nv);
    run<0, 1>(out, "v_mfma_f32_32x32x16_f16", "v_pk_fma_f16", nm, nv);
    run<0, 2>(out, "v_mfma_f32_32x32x16_f16", "v_pk_fma_f32", nm, nv / 2);
    run<0, 3>(out, "v_mfma_f32_32x32x16_f16", "v_sin_f32", nm, nv / 4);
    run<0, 4>(out, "v_mfma_f32_32x32x16_f16", "v_cvt_pkrtz_f16_f32", nm, nv);
    run<1, 0>(out, "v_mfma_f32_16x16x32_f16", "v_fma_f32", nm, nv);
    run<1, 1>(out, "v_mfma_f32_16x16x32_f16", "v_pk_fma_f16", nm, nv);
    run<1, 2>(out, "v_mfma_f32_16x16x32_f16", "v_pk_fma_f32", nm, nv / 2);
    run<2, 0>(out, "v_mfma_f32_32x32x16_bf16", "v_fma_f32", nm, nv);
    run<2, 1>(out, "v_mfma_f32_32x32x16_bf16", "v_pk_fma_f16", nm, nv);
    run<3, 0>(out, "v_mfma_f32_32x32x2_f32", "v_fma_f32", nm, nv);
    run<0, 7>(out, "v_mfma_f32_32x32x16_f16", "v_mad_u32_u24", nm, nv);
    run<0, 8>(out, "v_mfma_f32_32x32x16_f16", "v_fma_mix_f32", nm, nv);
    run<0, 5>(out, "v_mfma_f32_32x32x16_f16", "v_pk_fma_f16 (16 chains)", nm, nv);
    run<0, 6>(out, "v_mfma_f32_32x32x16_f16", "v_fma_f32 (16 chains)", nm, nv);
    run<1, 5>(out, "v_mfma_f32_16x16x32_f16", "v_pk_fma_f16 (16 chains)", nm, nv);
    return 0;
}
