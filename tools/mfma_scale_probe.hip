// Probe of v_mfma_scale_f32_32x32x64_f8f6f4 on gfx950 (fp8 e4m3 x fp8 e4m3, per-lane E8M0 scales): is the operand layout the one
// assumed by tensoir_amd's kernels -- A lane l: row l % 32, k = 32 (l / 32) + byte j; B lane l: column l % 32, same k; D register i
// of lane l: row (i % 4) + 8 (i / 4) + 4 (l / 32), column l % 32 -- and is the product scaled by 2^(sa - 127) 2^(sb - 127)?
// Build: hipcc --offload-arch=gfx950 tools/mfma_scale_probe.hip -o gpurun_scratch/mfma_scale_probe ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef int i32x8 __attribute__((ext_vector_type(8)));

__global__ void k(const i32x8* a, const i32x8* b, float* o, int sa, int sb) {
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a[threadIdx.x], b[threadIdx.x], c, 0, 0, 0, sa, 0, sb);
    for (int i = 0; i < 16; ++i) o[threadIdx.x * 16 + i] = c[i];
}

static float e4m3(unsigned char v) {      // OCP e4m3fn
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x;
    if (e == 0) x = std::ldexp((float)m, -9);
    else if (e == 15 && m == 7) x = NAN;
    else x = std::ldexp(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}

int main() {
    std::vector<unsigned char> A(64 * 32), B(64 * 32);
    srand(3);
    for (auto& v : A) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f || (v & 0x78) > 0x50); }
    for (auto& v : B) { do v = rand() & 0xff; while ((v & 0x7f) == 0x7f || (v & 0x78) > 0x50); }
    void *da, *db; float* dout;
    hipMalloc(&da, 2048); hipMalloc(&db, 2048); hipMalloc(&dout, 64 * 16 * 4);
    hipMemcpy(da, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(db, B.data(), 2048, hipMemcpyHostToDevice);
    for (int sa : {127, 110}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, (const i32x8*)da, (const i32x8*)db, dout, sa, 127);
        std::vector<float> out(64 * 16);
        hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost);
        double worst = 0, scale = std::ldexp(1.0, sa - 127);
        for (int l = 0; l < 64; ++l)
            for (int i = 0; i < 16; ++i) {
                const int m = (i % 4) + 8 * (i / 4) + 4 * (l / 32), n = l % 32;
                double ref = 0;
                for (int kk = 0; kk < 64; ++kk) {
                    const int la = m + 32 * (kk / 32), lb = n + 32 * (kk / 32), j = kk % 32;
                    ref += (double)e4m3(A[la * 32 + j]) * (double)e4m3(B[lb * 32 + j]);
                }
                ref *= scale;
                const double d = std::fabs(out[l * 16 + i] - ref) / std::fmax(1.0 * scale, std::fabs(ref));
                if (d > worst) worst = d;
            }
        printf("scale_a %d: worst relative deviation from the assumed layout %.3e  (sample out %g)\n", sa, worst, out[5]);
    }
    return 0;
}
