// K5: positional encoding + 3-layer decoder (150 -> 128 -> 128 -> out) on the matrix cores.
//
// Exact-fp32 MFMA (v_mfma_f32_32x32x2_f32): bf16 operands miss the 1e-4 parity bar (SURVEY.md
// section 7), and fp32 MFMA runs at the fp32 vector rate while sharing operands across the wave.
//
// Formulation: H^T[neuron, sample] = W[neuron, k] * X^T[k, sample].  A wave owns 32 samples.
//   A operand (weights)   : lane l -> W[row = tile*32 + (l&31)][k = kperm(t, l>>5)]   from LDS
//   B operand (activation): lane l -> X[sample = l&31][k = kperm(t, l>>5)]            from registers
//   C/D                   : lane l holds, for sample (l&31), neurons tile*32 + (r&3) + 8*(r>>2) + 4*(l>>5)
// Because D leaves each lane holding 64 of its own sample's 128 hidden units, those are exactly the B
// operands of the next layer when its contraction order is permuted to k2(t,h) -- no cross-lane traffic
// between layers.  Lane halves h=0/1 split every contraction's k range in two.
#include "tir_common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int F = 27;              // app_dim
constexpr int PE = 2;              // fea_pe == view_pe == pos_pe
constexpr int HID = 128;
constexpr int IN = F + 3 + 2 * PE * F + 2 * PE * 3;   // 150
constexpr int HALF = IN / 2;                          // 75
constexpr int NPF = PE * F;                           // 54
constexpr int R0 = HALF - NPF;                        // 21 raw features handled by half 0

// blob layout (floats)
constexpr int OFF_W0 = 0;                       // [IN][128]   (t*2+h)*128 + i*4 + mt
constexpr int OFF_B0 = OFF_W0 + IN * HID;       // [2][4][16]  (h*4+mt)*16 + r
constexpr int OFF_W1 = OFF_B0 + HID;            // [128][128]
constexpr int OFF_B1 = OFF_W1 + HID * HID;      // [2][4][16]
constexpr int OFF_W2 = OFF_B1 + HID;            // [2][64][4]
constexpr int OFF_B2 = OFF_W2 + 2 * 64 * 4;     // [4]
constexpr int MFMA_FLOATS = OFF_B2 + 4;         // 36356 floats = 145,424 B of LDS
// raw copies for the VALU kernel
constexpr int OFF_RW0 = MFMA_FLOATS;            // [128][150]
constexpr int OFF_RB0 = OFF_RW0 + HID * IN;
constexpr int OFF_RW1 = OFF_RB0 + HID;          // [128][128]
constexpr int OFF_RB1 = OFF_RW1 + HID * HID;
constexpr int OFF_RW2 = OFF_RB1 + HID;          // [4][128]
constexpr int OFF_RB2 = OFF_RW2 + 4 * HID;
constexpr int TOTAL_FLOATS = OFF_RB2 + 4;

// reference input index (models/tensorBase_rotated_lights.py:137-142, :12-17) handled at step t by half h
__host__ __device__ inline int kperm(int t, int h) {
    if (h == 0) return (t < NPF) ? (F + 3 + t) : (t - NPF);
    if (t < NPF) return F + 3 + NPF + t;                  // cos(PE feat)
    int q = t - NPF;
    if (q < F - R0) return R0 + q;                        // feat[21..26]
    q -= F - R0;
    if (q < 3) return F + q;                              // aux
    q -= 3;
    if (q < 3 * PE) return F + 3 + 2 * NPF + q;           // sin(PE aux)
    q -= 3 * PE;
    return F + 3 + 2 * NPF + 3 * PE + q;                  // cos(PE aux)
}

// hidden unit held by accumulator register q (= tile*16 + r) of a lane in half h
__host__ __device__ inline int unit_of(int q, int h) {
    int mt = q >> 4, r = q & 15;
    return mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
}

__global__ void k_pack_mlp(const float* __restrict__ w0, const float* __restrict__ b0,
                           const float* __restrict__ w1, const float* __restrict__ b1,
                           const float* __restrict__ w2, const float* __restrict__ b2, int out_dim,
                           float* __restrict__ p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= TOTAL_FLOATS) return;
    float v = 0.0f;
    if (i < OFF_B0) {
        int j = i - OFF_W0, th = j / 128, rem = j % 128, ii = rem / 4, mt = rem % 4;
        v = w0[(mt * 32 + ii) * IN + kperm(th >> 1, th & 1)];
    } else if (i < OFF_W1) {
        int j = i - OFF_B0, h = j / 64, q = j % 64;
        v = b0[unit_of(q, h)];
    } else if (i < OFF_B1) {
        int j = i - OFF_W1, th = j / 128, rem = j % 128, ii = rem / 4, mt = rem % 4;
        v = w1[(mt * 32 + ii) * HID + unit_of(th >> 1, th & 1)];
    } else if (i < OFF_W2) {
        int j = i - OFF_B1, h = j / 64, q = j % 64;
        v = b1[unit_of(q, h)];
    } else if (i < OFF_B2) {
        int j = i - OFF_W2, h = j / 256, q = (j % 256) / 4, o = j % 4;
        v = (o < out_dim) ? w2[o * HID + unit_of(q, h)] : 0.0f;
    } else if (i < MFMA_FLOATS) {
        int o = i - OFF_B2;
        v = (o < out_dim) ? b2[o] : 0.0f;
    } else if (i < OFF_RB0) v = w0[i - OFF_RW0];
    else if (i < OFF_RW1) v = b0[i - OFF_RB0];
    else if (i < OFF_RB1) v = w1[i - OFF_RW1];
    else if (i < OFF_RW2) v = b1[i - OFF_RB1];
    else if (i < OFF_RB2) { int j = i - OFF_RW2; v = (j / HID < out_dim) ? w2[j] : 0.0f; }
    else { int o = i - OFF_RB2; v = (o < out_dim) ? b2[o] : 0.0f; }
    p[i] = v;
}

__device__ __forceinline__ float act_out(float x, int act) {
    return act == 1 ? tanhf(x) : 1.0f / (1.0f + expf(-x));
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel: 512 threads = 8 waves, 32 samples per wave, persistent over 256-sample tiles
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(512)
k_mlp_mfma(const float* __restrict__ packed, const float* __restrict__ feat, const float* __restrict__ aux,
           const int32_t* __restrict__ aux_map, float* __restrict__ out, int64_t n, int out_dim, int act) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < MFMA_FLOATS; i += 512 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed + i);
    __syncthreads();

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s_raw = tile * 256 + wave * 32 + sl;
        const int64_t s = s_raw < n ? s_raw : n - 1;
        // ---- build this lane's 75 inputs ----
        float x[HALF];
        {
            float ft[F];
            const float* fr = feat + s * F;
#pragma unroll
            for (int d = 0; d < F; ++d) ft[d] = fr[d];
            if (h == 0) {
#pragma unroll
                for (int d = 0; d < F; ++d) {
                    x[2 * d] = sinf(ft[d]);
                    x[2 * d + 1] = sinf(ft[d] * 2.0f);
                }
#pragma unroll
                for (int q = 0; q < R0; ++q) x[NPF + q] = ft[q];
            } else {
                const int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
                const float a0 = aux[3 * ai], a1 = aux[3 * ai + 1], a2 = aux[3 * ai + 2];
#pragma unroll
                for (int d = 0; d < F; ++d) {
                    x[2 * d] = cosf(ft[d]);
                    x[2 * d + 1] = cosf(ft[d] * 2.0f);
                }
#pragma unroll
                for (int q = 0; q < F - R0; ++q) x[NPF + q] = ft[R0 + q];
                const int b = NPF + (F - R0);
                x[b] = a0; x[b + 1] = a1; x[b + 2] = a2;
                x[b + 3] = sinf(a0); x[b + 4] = sinf(a0 * 2.0f);
                x[b + 5] = sinf(a1); x[b + 6] = sinf(a1 * 2.0f);
                x[b + 7] = sinf(a2); x[b + 8] = sinf(a2 * 2.0f);
                x[b + 9] = cosf(a0); x[b + 10] = cosf(a0 * 2.0f);
                x[b + 11] = cosf(a1); x[b + 12] = cosf(a1 * 2.0f);
                x[b + 13] = cosf(a2); x[b + 14] = cosf(a2 * 2.0f);
            }
        }
        // ---- layer 1 ----
        f32x16 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* bp = lds + OFF_B0 + (h * 4 + mt) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = bp[r];
        }
        {
            const float* wp = lds + OFF_W0 + h * 128 + sl * 4;
#pragma unroll
            for (int t = 0; t < HALF; ++t) {
                const float4 a = *reinterpret_cast<const float4*>(wp + t * 256);
                const float b = x[t];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc[3], 0, 0, 0);
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep LDS reads from piling up in VGPRs
            }
        }
        // ---- layer 2 ----
        f32x16 acc2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* bp = lds + OFF_B1 + (h * 4 + mt) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = bp[r];
        }
        {
            const float* wp = lds + OFF_W1 + h * 128 + sl * 4;
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                const float4 a = *reinterpret_cast<const float4*>(wp + t * 256);
                const float b = fmaxf(acc[t >> 4][t & 15], 0.0f);
                acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc2[0], 0, 0, 0);
                acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc2[1], 0, 0, 0);
                acc2[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc2[2], 0, 0, 0);
                acc2[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc2[3], 0, 0, 0);
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- layer 3 (out_dim <= 4): per-lane dot over its 64 hidden units, then add the halves ----
        float o0 = 0.f, o1 = 0.f, o2 = 0.f, o3 = 0.f;
        {
            const float* wp = lds + OFF_W2 + h * 256;
#pragma unroll
            for (int q = 0; q < 64; ++q) {
                const float hv = fmaxf(acc2[q >> 4][q & 15], 0.0f);
                const float4 w = *reinterpret_cast<const float4*>(wp + q * 4);
                o0 = fmaf(hv, w.x, o0); o1 = fmaf(hv, w.y, o1); o2 = fmaf(hv, w.z, o2); o3 = fmaf(hv, w.w, o3);
            }
        }
        o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        o2 += __shfl_xor(o2, 32, 64); o3 += __shfl_xor(o3, 32, 64);
        if (h == 0 && s_raw < n) {
            const float* b2 = lds + OFF_B2;
            float* op = out + s_raw * out_dim;
            op[0] = act_out(o0 + b2[0], act);
            if (out_dim > 1) op[1] = act_out(o1 + b2[1], act);
            if (out_dim > 2) op[2] = act_out(o2 + b2[2], act);
            if (out_dim > 3) op[3] = act_out(o3 + b2[3], act);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// plain VALU kernel (one sample per lane, weights by wave-uniform scalar loads): cross-check only
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_mlp_valu(const float* __restrict__ packed, const float* __restrict__ feat, const float* __restrict__ aux,
           const int32_t* __restrict__ aux_map, float* __restrict__ out, int64_t n, int out_dim, int act) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (s >= n) return;
    float x[IN];
    const int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
    for (int d = 0; d < F; ++d) x[d] = feat[s * F + d];
    for (int d = 0; d < 3; ++d) x[F + d] = aux[3 * ai + d];
    for (int d = 0; d < F; ++d)
        for (int f = 0; f < PE; ++f) {
            float y = x[d] * (float)(1 << f);
            x[F + 3 + d * PE + f] = sinf(y);
            x[F + 3 + NPF + d * PE + f] = cosf(y);
        }
    for (int d = 0; d < 3; ++d)
        for (int f = 0; f < PE; ++f) {
            float y = x[F + d] * (float)(1 << f);
            x[F + 3 + 2 * NPF + d * PE + f] = sinf(y);
            x[F + 3 + 2 * NPF + 3 * PE + d * PE + f] = cosf(y);
        }
    float h1[HID], h2[HID];
    for (int j = 0; j < HID; ++j) {
        float a = packed[OFF_RB0 + j];
        const float* w = packed + OFF_RW0 + j * IN;
        for (int k = 0; k < IN; ++k) a = fmaf(w[k], x[k], a);
        h1[j] = fmaxf(a, 0.0f);
    }
    for (int j = 0; j < HID; ++j) {
        float a = packed[OFF_RB1 + j];
        const float* w = packed + OFF_RW1 + j * HID;
        for (int k = 0; k < HID; ++k) a = fmaf(w[k], h1[k], a);
        h2[j] = fmaxf(a, 0.0f);
    }
    for (int o = 0; o < out_dim; ++o) {
        float a = packed[OFF_RB2 + o];
        const float* w = packed + OFF_RW2 + o * HID;
        for (int k = 0; k < HID; ++k) a = fmaf(w[k], h2[k], a);
        out[s * out_dim + o] = act_out(a, act);
    }
}

int check_mlp(const TirMlp* m) {
    if (!m || !m->packed) return TIR_ERR_ARG;
    if (m->feat_dim != F || m->pe != PE || m->hidden != HID || m->out_dim < 1 || m->out_dim > 4)
        return TIR_ERR_UNSUPPORTED;
    return TIR_OK;
}

}  // namespace

extern "C" int64_t tir_mlp_packed_floats(int32_t feat_dim, int32_t pe, int32_t hidden, int32_t out_dim) {
    if (feat_dim != F || pe != PE || hidden != HID || out_dim < 1 || out_dim > 4) return TIR_ERR_UNSUPPORTED;
    return TOTAL_FLOATS;
}

extern "C" int tir_pack_mlp(const float* w0, const float* b0, const float* w1, const float* b1,
                            const float* w2, const float* b2, int32_t feat_dim, int32_t pe, int32_t hidden,
                            int32_t out_dim, float* packed, void* stream) {
    if (!w0 || !b0 || !w1 || !b1 || !w2 || !b2 || !packed) return TIR_ERR_ARG;
    if (feat_dim != F || pe != PE || hidden != HID || out_dim < 1 || out_dim > 4) return TIR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_pack_mlp, dim3((TOTAL_FLOATS + 255) / 256), dim3(256), 0, tir_stream(stream),
                       w0, b0, w1, b1, w2, b2, out_dim, packed);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_fwd(const TirMlp* m, const float* feat, const float* aux, const int32_t* aux_map,
                           float* out, int64_t n, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!feat || !aux || !out))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    static bool attr_set = false;
    const size_t lds = (size_t)MFMA_FLOATS * sizeof(float);
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(k_mlp_mfma),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return -(int)e;
        attr_set = true;
    }
    int64_t tiles = (n + 255) / 256;
    unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(k_mlp_mfma, dim3(grid), dim3(512), lds, tir_stream(stream), m->packed, feat, aux,
                       aux_map, out, n, m->out_dim, m->act);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_fwd_valu(const TirMlp* m, const float* feat, const float* aux, const int32_t* aux_map,
                                float* out, int64_t n, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!feat || !aux || !out))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_mlp_valu, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, tir_stream(stream), m->packed,
                       feat, aux, aux_map, out, n, m->out_dim, m->act);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
