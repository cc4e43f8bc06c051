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
typedef float f32x4 __attribute__((ext_vector_type(4)));

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
constexpr int FP32_TOTAL = OFF_RB2 + 4;
// split-bf16 image (one contiguous LDS image): fp32 header [b0p 128 | b1p 128 | W2a 544 | b2 4 | pad 4]
// then bf16 operand tiles  W0hi | W0lo | W1hi | W1lo, each [kb][h][mt][i][8]  (8 bf16 = one ds_read_b128).
// W2a = layer-3 weights as the A operand of v_mfma_f32_4x4x1_16b_f32: [h][o][q] with a row stride of 68 floats (the 4
// rows a 16-lane LDS group reads land on disjoint banks).  Layer 1's bias rides in the padded k slot HALF of half 0
// (input 1.0), so the layer-1 accumulators start from the inline constant 0.
constexpr int KB0 = (HALF + 7) / 8;             // 10 k-blocks of 16 for layer 1 (75 -> 80 per half)
constexpr int KB1 = 64 / 8;                     // 8 k-blocks for layer 2
constexpr int W2A_STRIDE = 68;
[[maybe_unused]] constexpr int BH_B0 = 0;
constexpr int BH_B1 = 128, BH_W2 = 256, BH_B2 = BH_W2 + 2 * 4 * W2A_STRIDE, BH_FLOATS = BH_B2 + 8;
constexpr int BW0_ELEMS = KB0 * 2 * 4 * 32 * 8; // 20480 bf16
constexpr int BW1_ELEMS = KB1 * 2 * 4 * 32 * 8; // 16384 bf16
constexpr int BF_BYTES = BH_FLOATS * 4 + (2 * BW0_ELEMS + 2 * BW1_ELEMS) * 2;   // 150,560 B
constexpr int BF_FLOATS = BF_BYTES / 4;
constexpr int OFF_BF = FP32_TOTAL;              // float offset of the bf16 image inside the blob
// Second image for the AUX-TABLE variant of layer 1 (k_mlp_bf16<.., AUXT = true>): when the 3 aux inputs of a launch take few
// distinct values (the view direction of the radiance decoder: one per ray, or one per light direction), their 15 input
// columns (aux + PE(aux)) and the bias are folded into a per-aux-row fp32 table T[a][unit] = b0 + W0[:, aux part] x_aux
// (tir_mlp_aux_table) that INITIALISES the layer-1 accumulators; the matrix product then runs over the remaining
// 27 + 108 = 135 inputs: 68 / 67 per lane half -> 9 k-blocks of 8 instead of 10 (-12 of the 216 MFMAs per 32-sample tile).
// Same layout as the main image: header | W0A hi | W0A lo | W1 hi | W1 lo.
constexpr int R0A = 14;                          // raw features handled by half 0 in the aux-table layout (half 1: 13)
constexpr int KB0A = 9;
constexpr int BW0A_ELEMS = KB0A * 2 * 4 * 32 * 8;                               // 18432 bf16
constexpr int BFA_BYTES = BH_FLOATS * 4 + (2 * BW0A_ELEMS + 2 * BW1_ELEMS) * 2; // 142,368 B
constexpr int BFA_FLOATS = BFA_BYTES / 4;
constexpr int OFF_BFA = OFF_BF + BF_FLOATS;
// Third image: the aux-table layout with ONE fp16 operand plane per layer (header | W0A f16 | W1 f16, 72,864 B) for the
// single-product fp16 decoder (k_mlp_f16_auxt, v_mfma_f32_32x32x16_f16): 11-bit operands, one MFMA per tile and k-block
// instead of three.  Unit roundoff 2^-12 per operand (bf16 single product: 2^-9): ~1e-4 on a decoder output -- NOT parity
// grade on its own; the product path uses it only where the output is averaged down before it reaches a rendered map (the
// radiance of the secondary-ray records, i.e. indirect light: models/relight_utils.py:818-832), see DESIGN section 4.1.
constexpr int FH_BYTES = BH_FLOATS * 4 + (BW0A_ELEMS + BW1_ELEMS) * 2;          // 72,864 B
constexpr int FH_FLOATS = FH_BYTES / 4;
constexpr int OFF_FH = OFF_BFA + BFA_FLOATS;
// Fourth image: the fp16 decoder for the FUSED gather + decoder kernel (k_indirect_fused).  There a lane half does not see all
// 27 features: the gather's 32x32 MFMA leaves half h of a record's lane pair with 16 of the 32 feature rows, and basis_mat's
// rows are dealt so that half 0 owns features 0..13 and half 1 owns 14..26.  Layer 1's contraction order per half is therefore
// [for each own feature j: sin f, sin 2f, cos f, cos 2f, f] (5 inputs per feature, 70 / 65 -> 9 k-blocks); the aux columns and
// the bias come from the aux table as in the other aux-table images.  Same size and header as the fp16 image.
constexpr int FUS_NF0 = 14;                       // features owned by lane half 0 (half 1: F - 14 = 13)
constexpr int OFF_FF = OFF_FH + FH_FLOATS;
// Fifth image: the ROUNDING RESIDUE of the fused layout's fp16 weights, W - fp16(W), scaled by 2^17 and stored as fp8 (e4m3, one
// byte per element, same element order as the fp16 planes: [W0 lo | W1 lo]) for the high-precision fused kernel
// (k_indirect_fused_hp).  On a trained checkpoint the fp16 decoder's deviation on rgb_with_brdf_map is the WEIGHT rounding --
// one fixed perturbation of the function, which the average over a ray's records and 128 light directions does not reduce --
// while the rounding of the activations averages out (profiles/r06_decoder_precision_probe.json: 1.8e-5 with fp16 weights
// whatever the activations' precision, 6.6e-6 with the weights as hi + lo and fp16 activations).  The residue needs only a few
// bits: |W - fp16(W)| <= 2^-11 |W|, so 2^17 x residue < 448 for |W| < 7 (saturating beyond), four significant bits of it give the
// weights 15 bits, and its product with the activations runs on the fp8 matrix pipe -- half the LDS of a second fp16 plane (a full
// hi + lo image does not fit next to the gather's tiles).
// Element order (round 6, second form): the residue product runs on the block-scaled fp8 instruction, K = 64 per issue
// (v_mfma_scale_f32_32x32x64_f8f6f4 with the A scale 2^-17: it accumulates STRAIGHT into the fp16 product's accumulators -- no
// second accumulator set, no fold).  A lane's A operand is 32 consecutive bytes = its 8 slots of four consecutive k-blocks:
// byte ((((j 2 + h) 4 + mt) 32 + row) 32 + (kb % 4) 8 + e) of the layer's section, j = kb / 4 (layer 1: 9 k-blocks padded to 12).
constexpr int F8_G0 = (KB0A + 3) / 4, F8_G1 = KB1 / 4;                  // groups of four k-blocks: 3, 2
constexpr int F8_W0_BYTES = F8_G0 * 2 * 4 * 32 * 32, F8_W1_BYTES = F8_G1 * 2 * 4 * 32 * 32;          // 24,576 + 16,384 B
constexpr int F8_FLOATS = (F8_W0_BYTES + F8_W1_BYTES) / 4;
constexpr int OFF_F8 = OFF_FF + FH_FLOATS;
constexpr float F8_SCALE = 131072.0f;        // 2^17 (undone by the E8M0 scale 110 = 2^-17 of the matrix instruction)
constexpr int TOTAL_FLOATS = OFF_F8 + F8_FLOATS;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

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

// the same for the aux-table layout: half 0 -> sin(PE feat), feat[0..R0A); half 1 -> cos(PE feat), feat[R0A..F); -1 = padding
__host__ __device__ inline int kperm_a(int t, int h) {
    if (t < NPF) return F + 3 + (h ? NPF : 0) + t;
    const int q = t - NPF;
    if (h == 0) return q < R0A ? q : -1;
    return q < F - R0A ? R0A + q : -1;
}

// the fused layout: reference input column of k-slot kk of half h (-1 = padding)
__host__ __device__ inline int kperm_f(int kk, int h) {
    const int j = kk / 5, kind = kk % 5, nf = h ? F - FUS_NF0 : FUS_NF0;
    if (j >= nf) return -1;
    const int f = h ? FUS_NF0 + j : j;
    switch (kind) {
        case 0: return F + 3 + 2 * f;                 // sin(f)
        case 1: return F + 3 + 2 * f + 1;             // sin(2f)
        case 2: return F + 3 + NPF + 2 * f;           // cos(f)
        case 3: return F + 3 + NPF + 2 * f + 1;       // cos(2f)
        default: return f;                            // f itself
    }
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
    else if (i < FP32_TOTAL) { int o = i - OFF_RB2; v = (o < out_dim) ? b2[o] : 0.0f; }
    else if (i >= OFF_F8) {                     // fp8 residue of the fused layout's fp16 planes: four elements per float slot
        float lo[4];
        for (int t = 0; t < 4; ++t) {
            int idx = (i - OFF_F8) * 4 + t;
            const bool l2 = idx >= F8_W0_BYTES;
            if (l2) idx -= F8_W0_BYTES;
            const int e = idx % 8, kl = (idx / 8) % 4, ii = (idx / 32) % 32, mt = (idx / 1024) % 4, h = (idx / 4096) % 2, jg = idx / 8192;
            const int kb = jg * 4 + kl, kk = kb * 8 + e;
            float wv;
            if (l2) wv = w1[(mt * 32 + ii) * HID + unit_of(kk, h)];
            else if (kb >= KB0A) wv = 0.0f;                                   // padding k-blocks of the last group
            else { const int in = kperm_f(kk, h); wv = in >= 0 ? w0[(mt * 32 + ii) * IN + in] : 0.0f; }
            const float r = (wv - (float)(_Float16)wv) * F8_SCALE;
            lo[t] = __builtin_amdgcn_fmed3f(r, -448.0f, 448.0f);
        }
        int w = 0;
        w = __builtin_amdgcn_cvt_pk_fp8_f32(lo[0], lo[1], w, false);
        w = __builtin_amdgcn_cvt_pk_fp8_f32(lo[2], lo[3], w, true);
        v = __builtin_bit_cast(float, w);
    } else {
        const bool fused = i >= OFF_FF;         // the fused kernel's fp16 image: per-half feature ownership (kperm_f)
        const bool f16 = i >= OFF_FH;           // the fp16 images: aux-table layout, one operand plane per layer
        const bool auxt = i >= OFF_BFA;         // the aux-table image: same header and W1 planes, 9-block W0 planes
        const int bw0 = auxt ? BW0A_ELEMS : BW0_ELEMS;
        int j = i - (fused ? OFF_FF : f16 ? OFF_FH : auxt ? OFF_BFA : OFF_BF);  // float slot inside the image
        if (j < BH_FLOATS) {
            if (j < BH_B1) v = b0[unit_of(j % 64, j / 64)];
            else if (j < BH_W2) { int q = j - BH_B1; v = b1[unit_of(q % 64, q / 64)]; }
            else if (j < BH_B2) {
                int q = j - BH_W2, h = q / (4 * W2A_STRIDE), o = (q / W2A_STRIDE) % 4, u = q % W2A_STRIDE;
                v = (o < out_dim && u < 64) ? 0.5f * w2[o * HID + unit_of(u, h)] : 0.0f;     // x 1/2: B operand is 2 relu(h)
            }
            else { int o = j - BH_B2; v = (o < out_dim) ? b2[o] : 0.0f; }
        } else {                                // two bf16 per float slot
            unsigned short hw[2];
            for (int t = 0; t < 2; ++t) {
                int e_all = (j - BH_FLOATS) * 2 + t;
                int sec, idx;
                if (f16) {
                    if (e_all < bw0) { sec = 0; idx = e_all; }
                    else { sec = 2; idx = e_all - bw0; }
                }
                else if (e_all < bw0) { sec = 0; idx = e_all; }
                else if (e_all < 2 * bw0) { sec = 1; idx = e_all - bw0; }
                else if (e_all < 2 * bw0 + BW1_ELEMS) { sec = 2; idx = e_all - 2 * bw0; }
                else { sec = 3; idx = e_all - 2 * bw0 - BW1_ELEMS; }
                int e = idx % 8, ii = (idx / 8) % 32, mt = (idx / 256) % 4, h = (idx / 1024) % 2, kb = idx / 2048;
                int kk = kb * 8 + e;
                float wv;
                if (sec < 2 && fused) { const int in = kperm_f(kk, h); wv = in >= 0 ? w0[(mt * 32 + ii) * IN + in] : 0.0f; }
                else if (sec < 2 && auxt) { const int in = kperm_a(kk, h); wv = in >= 0 ? w0[(mt * 32 + ii) * IN + in] : 0.0f; }
                else if (sec < 2) wv = (kk < HALF) ? w0[(mt * 32 + ii) * IN + kperm(kk, h)]
                                              : ((kk == HALF && h == 0) ? b0[mt * 32 + ii] : 0.0f);   // bias slot (input = 1)
                else wv = w1[(mt * 32 + ii) * HID + unit_of(kk, h)];
                if (f16) {
                    hw[t] = __builtin_bit_cast(unsigned short, (_Float16)wv);
                    continue;
                }
                __bf16 hi = (__bf16)wv;
                __bf16 r = (sec & 1) ? (__bf16)(wv - (float)hi) : hi;
                hw[t] = __builtin_bit_cast(unsigned short, r);
            }
            unsigned int packed2 = (unsigned int)hw[0] | ((unsigned int)hw[1] << 16);
            v = __builtin_bit_cast(float, packed2);
        }
    }
    p[i] = v;
}

__device__ __forceinline__ float act_out(float x, int act) {
    return act == 1 ? tanhf(x) : 1.0f / (1.0f + expf(-x));
}

// ------------------------------------------------------------------------------------------------
// MFMA kernel: 512 threads = 8 waves, 32 samples per wave, persistent over 256-sample tiles
// ------------------------------------------------------------------------------------------------
template <bool SAVE>
__global__ void __launch_bounds__(512)
k_mlp_mfma(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
           const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
           const int32_t* __restrict__ n_dev, int out_dim, int act, float* __restrict__ h1o, float* __restrict__ h2o) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < MFMA_FLOATS; i += 512 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed + i);
    __syncthreads();
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));      // device-side row count (no host sync needed)

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
            const float* fr = feat + s * fstride;
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
                int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
            if (aux_mod > 0) ai %= aux_mod;
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
        if (SAVE && s_raw < n) {      // post-ReLU hidden activations, natural [sample][unit] layout (training)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(h1o + s_raw * HID + mt * 32 + 8 * i + 4 * h) =
                        make_float4(fmaxf(acc[mt][4 * i], 0.f), fmaxf(acc[mt][4 * i + 1], 0.f),
                                    fmaxf(acc[mt][4 * i + 2], 0.f), fmaxf(acc[mt][4 * i + 3], 0.f));
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
        if (SAVE && s_raw < n) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(h2o + s_raw * HID + mt * 32 + 8 * i + 4 * h) =
                        make_float4(fmaxf(acc2[mt][4 * i], 0.f), fmaxf(acc2[mt][4 * i + 1], 0.f),
                                    fmaxf(acc2[mt][4 * i + 2], 0.f), fmaxf(acc2[mt][4 * i + 3], 0.f));
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
// split-bf16 kernel (v_mfma_f32_32x32x16_bf16): x = hi + lo in bf16, products hi*hi + hi*lo + lo*hi
// accumulated in fp32 (NPROD = 3, ~1e-6 abs error on the outputs), or hi*hi only (NPROD = 1, fast mode).
// Same wave/lane decomposition as the fp32 kernel; a lane supplies 8 consecutive k per MFMA.
// ------------------------------------------------------------------------------------------------
// x = hi + lo with hi = bf16(x) (RNE) and lo = bf16(x - hi), two values at a time: one packed conversion for the hi
// pair, shift / mask to get the pair back as floats, one (packed) subtract, one packed conversion for the lo pair --
// 2.5 VALU instructions per value instead of ~4 for the element-wise form.
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

__device__ __forceinline__ void split8(const float* v, bf16x8& hi, bf16x8& lo) {
    u32x4_t H, L;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t x = {v[2 * p], v[2 * p + 1]};
        const unsigned hb = __builtin_bit_cast(unsigned, __builtin_convertvector(x, bf16x2_t));
        const f32x2_t hf = {__builtin_bit_cast(float, hb << 16), __builtin_bit_cast(float, hb & 0xffff0000u)};
        const f32x2_t r = x - hf;
        H[p] = hb;
        L[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    }
    hi = __builtin_bit_cast(bf16x8, H);
    lo = __builtin_bit_cast(bf16x8, L);
}

// sin and cos of x on the hardware transcendental unit.  v_sin_f32 / v_cos_f32 take the angle in REVOLUTIONS and are
// accurate to 1.4e-7 for |angle| <= 1/8 revolution (measured on gfx950) but the naive x * (1/2pi) loses the fraction of
// large arguments (2.8e-6 at |x| = 32).  So the fraction of x / 2pi is formed exactly: k = rint(x * c_hi);
// t = fma(x, c_hi, -k) is the exact fractional part of the product (one rounding, |t| <= 1/2), and fma(x, c_lo, t) adds
// the part of 1/2pi that c_hi misses.  Six instruction slots instead of ~30 for the polynomial form -- the decoder
// kernel is VALU-issue bound -- with <= 4e-7 absolute error for |x| up to ~1e5 (NaN / inf give NaN like sinf).
__device__ __forceinline__ void fast_sincos(float x, float& s, float& c) {
    const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;      // c_hi + c_lo = 1 / (2 pi)
    const float k = rintf(x * c_hi);
    float t = fmaf(x, c_hi, -k);
    t = fmaf(x, c_lo, t);
    s = __builtin_amdgcn_sinf(t);
    c = __builtin_amdgcn_cosf(t);
}

// This lane-half's positional encodings of value v: half 0 -> {sin v, sin 2v}, half 1 -> {cos v, cos 2v}.
// cos x = sin(x + pi/2): with the angle in revolutions, both halves evaluate v_sin_f32 at t + q and 2t + q, q = 0 for
// half 0 and 1/4 for half 1 -- two transcendental issues per value and no select, instead of sin + cos + double-angle
// products + two selects (each half used to compute both functions and keep one).  t is the exact fraction of
// v / 2pi (see fast_sincos), |t| <= 1/2, so 2t + q is exact up to one rounding at magnitude <= 1.25 (6e-8 rev).
__device__ __forceinline__ void pe_pair(float v, float q, float& p1, float& p2) {
    const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;      // c_hi + c_lo = 1 / (2 pi)
    const float k = rintf(v * c_hi);
    float t = fmaf(v, c_hi, -k);
    t = fmaf(v, c_lo, t);
    p1 = __builtin_amdgcn_sinf(t + q);
    p2 = __builtin_amdgcn_sinf(fmaf(t, 2.0f, q));
}

struct AuxPE { float s[3], c[3], s2[3], c2[3]; };   // sin/cos of aux and of 2*aux

// input t (>= NPF) of the tail for lane half h: half 0 -> feat[0..R0), half 1 -> feat[R0..F), aux, sin/cos PE(aux)
template <int T>
__device__ __forceinline__ float tail_input(const float (&ft)[F + 1], const float (&ax)[3], const AuxPE& ap, int h) {
    constexpr int q = T - NPF;
    float a = 0.0f, b = 0.0f;
    if constexpr (T == HALF) a = 1.0f;          // the constant input that carries layer 1's bias (half 0 only)
    if constexpr (q < R0) a = ft[q];
    if constexpr (q < F - R0) b = ft[R0 + q];
    else if constexpr (q < F - R0 + 3) b = ax[q - (F - R0)];
    else if constexpr (q < F - R0 + 3 + 3 * PE) {
        constexpr int u = q - (F - R0 + 3);
        b = (u & 1) ? ap.s2[u >> 1] : ap.s[u >> 1];
    } else if constexpr (q < F - R0 + 3 + 6 * PE) {
        constexpr int u = q - (F - R0 + 3 + 3 * PE);
        b = (u & 1) ? ap.c2[u >> 1] : ap.c[u >> 1];
    }
    return h ? b : a;
}

// the aux-table layout's tail: half 0 -> feat[0..R0A), half 1 -> feat[R0A..F), zero padding (no aux inputs, no bias slot)
template <int T>
__device__ __forceinline__ float tail_input_a(const float (&ft)[F + 1], int h) {
    constexpr int q = T - NPF;
    float a = 0.0f, b = 0.0f;
    if constexpr (q < R0A) a = ft[q];
    if constexpr (q < F - R0A) b = ft[R0A + q];
    return h ? b : a;
}

template <int KB, int E, bool AUXT>
__device__ __forceinline__ void build_pair(const float (&ft)[F + 1], const float (&ax)[3], const AuxPE& ap, int h, float hq,
                                           float (&v)[8]) {
    constexpr int t = KB * 8 + E;
    if constexpr (t + 1 < NPF) pe_pair(ft[t >> 1], hq, v[E], v[E + 1]);
    else if constexpr (AUXT) {
        v[E] = tail_input_a<t>(ft, h);
        v[E + 1] = tail_input_a<t + 1>(ft, h);
    } else {
        v[E] = tail_input<t>(ft, ax, ap, h);
        v[E + 1] = tail_input<t + 1>(ft, ax, ap, h);
    }
}

// ReLU.  fmaxf(x, 0) (and med3(x, 0, inf), which the compiler folds back) lowers to a canonicalising v_max(x, x) plus
// the real v_max, and the decoder kernel is VALU-issue bound.  One instruction instead: the INTEGER maximum of the bit
// pattern and 0 (every float with the sign bit set is a negative integer -> +0.0, every other pattern is returned as it
// is) -- the same function as v_max_f32(0, x) for every non-NaN x, and a NaN stays a NaN as torch.relu keeps it.
// A builtin, not inline asm: these reads come straight behind MFMAs that wrote the operand, and the hazard recogniser
// inserts the MFMA -> VALU wait states only for instructions it can see (inline asm read stale accumulators in the
// first fused kernel: 1e-3 errors).
__device__ __forceinline__ float relu(float x) {
    return __builtin_bit_cast(float, __builtin_elementwise_max(__builtin_bit_cast(int, x), 0));
}

// LDS operand tiles are read through four per-lane byte bases (W0hi / W0lo / W1hi / W1lo section start + this lane's
// offset) that the compiler cannot see through, plus compile-time offsets < 64 KB that fit the 16-bit ds offset field.
// Left to itself (150 KB image, absolute addresses above the 64 KB offset range) the compiler hoists one address VGPR
// PER READ out of the tile loop -- ~80 registers of a 256-register budget.
typedef __attribute__((address_space(3))) const bf16x8 lds_bf16x8;
__device__ __forceinline__ bf16x8 lds_tile(unsigned base, int byte_off) {
    return *reinterpret_cast<lds_bf16x8*>(base + (unsigned)byte_off);
}
__device__ __forceinline__ unsigned opaque(unsigned v) {
    asm volatile("" : "+v"(v));
    return v;
}


// ---- per k-block: [issue the A-tile LDS reads] [build that block's 8 inputs on the VALU] [12 MFMAs].
// The MFMAs of block kb execute on the matrix pipe while the wave's VALU already builds block kb+1 (intra-wave overlap
// instead of "all inputs, then all MFMAs"), the input build covers the LDS latency, and only one block of inputs is live.
// 8 floats -> one fp16 operand vector (round to nearest even: v_cvt_pk_f16_f32), carried in the bf16x8 register type
__device__ __forceinline__ void cvt8_f16(const float* v, bf16x8& x) {
    u32x4_t H;
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const f32x2_t f = {v[2 * p], v[2 * p + 1]};
        H[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(f, f16x2_t));
    }
    x = __builtin_bit_cast(bf16x8, H);
}

template <int NPROD, bool F16 = false>
__device__ __forceinline__ void mfma12(const bf16x8 (&ah)[4], const bf16x8 (&al)[4], const bf16x8& bh, const bf16x8& bl,
                                       f32x16 (&acc)[4]) {
    if constexpr (F16) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
            acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, bh), acc[mt], 0, 0, 0);
        return;
    }
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bh, acc[mt], 0, 0, 0);
    if (NPROD == 3) {
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mt], bh, acc[mt], 0, 0, 0);
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mt], bl, acc[mt], 0, 0, 0);
    }
}

template <int NPROD, int KB, bool AUXT = false, bool F16 = false>
__device__ __forceinline__ void layer1_interleaved(unsigned whi, unsigned wlo, int h,
                                                   const float (&ft)[F + 1], const float (&ax)[3], const AuxPE& ap, float hq,
                                                   f32x16 (&acc)[4]) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
        if (NPROD == 3) al[mt] = lds_tile(wlo, KB * 4096 + mt * 512);
    }
    float v[8];
    build_pair<KB, 0, AUXT>(ft, ax, ap, h, hq, v);
    build_pair<KB, 2, AUXT>(ft, ax, ap, h, hq, v);
    build_pair<KB, 4, AUXT>(ft, ax, ap, h, hq, v);
    build_pair<KB, 6, AUXT>(ft, ax, ap, h, hq, v);
    bf16x8 xh, xl = {};
    if constexpr (F16) cvt8_f16(v, xh); else split8(v, xh, xl);
    mfma12<NPROD, F16>(ah, al, xh, xl, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < (AUXT ? KB0A : KB0)) layer1_interleaved<NPROD, KB + 1, AUXT, F16>(whi, wlo, h, ft, ax, ap, hq, acc);
}

// ReLU that also saturates at the largest finite fp16 (one v_med3_f32, the same issue slot as the plain ReLU): the hidden
// activations of the fp16 decoder are rounded to fp16 operands, and a value above 65504 would turn into +inf there.
__device__ __forceinline__ float relu_sat16(float x) {
    return __builtin_amdgcn_fmed3f(x, 0.0f, 65504.0f);      // (builtin: visible to the hazard recogniser, see relu)
}

template <int NPROD, int KB, bool F16 = false>
__device__ __forceinline__ void layer2_interleaved(unsigned whi, unsigned wlo,
                                                   const f32x16 (&hid)[4], f32x16 (&acc)[4]) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
        if (NPROD == 3) al[mt] = lds_tile(wlo, KB * 4096 + mt * 512);
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        constexpr int q0 = KB * 8;
        v[e] = F16 ? relu_sat16(hid[(q0 + e) >> 4][(q0 + e) & 15]) : relu(hid[(q0 + e) >> 4][(q0 + e) & 15]);
    }
    bf16x8 xh, xl = {};
    if constexpr (F16) cvt8_f16(v, xh); else split8(v, xh, xl);
    mfma12<NPROD, F16>(ah, al, xh, xl, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < KB1) layer2_interleaved<NPROD, KB + 1, F16>(whi, wlo, hid, acc);
}

// one sample's decoder inputs as they come from memory: 27 features (+ zero pad) and the 3 aux values
struct RowIn { float ft[F + 1]; float ax[3]; int64_t ai; };

template <bool VEC>
__device__ __forceinline__ void load_row(const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
                                         int64_t s, int64_t ai, RowIn& r) {
    const float* fr = feat + s * fstride;
    if (VEC) {       // rows are 16-byte aligned and at least 28 floats long: 7 dwordx4 per lane instead of 27 dwords
#pragma unroll
        for (int q = 0; q < 7; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(fr + 4 * q);
            r.ft[4 * q] = v.x; r.ft[4 * q + 1] = v.y; r.ft[4 * q + 2] = v.z; r.ft[4 * q + 3] = v.w;
        }
        r.ft[F] = 0.0f;
    } else {
#pragma unroll
        for (int d = 0; d < F; ++d) r.ft[d] = fr[d];
        r.ft[F] = 0.0f;
    }
    r.ai = ai;
    if (aux) { r.ax[0] = aux[3 * ai]; r.ax[1] = aux[3 * ai + 1]; r.ax[2] = aux[3 * ai + 2]; }
    else { r.ax[0] = r.ax[1] = r.ax[2] = 0.0f; }            // aux-table launches never read the aux values
}

__device__ __forceinline__ int64_t aux_index(const int32_t* __restrict__ aux_map, int aux_mod, int64_t s) {
    int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
    if (aux_mod > 0) ai %= aux_mod;
    return ai;
}

// the decoder for the workgroup `bid` of `nblk` cooperating on one (decoder, row set) job
// AUXT: `aux` is the fp32 table [n_aux][128] of tir_mlp_aux_table (layer-1 accumulators start from its row aux_index(s)
// instead of 0, the matrix product skips the aux columns and the bias slot: 9 k-blocks), not the [n_aux][3] aux values.
// F16 (with AUXT, NPROD = 1): the single-product fp16 operand image (OFF_FH) instead of the bf16 hi / lo planes.
template <int NPROD, bool VEC, bool SAVE, bool AUXT = false, bool F16 = false>
__device__ __forceinline__ void
mlp_bf16_body(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
              const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
              const int32_t* __restrict__ n_dev, int out_dim, int act, float* __restrict__ h1o, float* __restrict__ h2o,
              const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    constexpr int W0_ELEMS = AUXT ? BW0A_ELEMS : BW0_ELEMS;
    static_assert(!F16 || (AUXT && NPROD == 1), "the fp16 image exists for the aux-table layout, one product");
    {
        const float* src = packed + (F16 ? OFF_FH : AUXT ? OFF_BFA : OFF_BF);
        for (int i = threadIdx.x * 4; i < (F16 ? FH_FLOATS : AUXT ? BFA_FLOATS : BF_FLOATS); i += 512 * 4)
            *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(src + i);
    }
    const float* __restrict__ table = AUXT ? aux : nullptr;
    const float* __restrict__ auxv = AUXT ? nullptr : aux;
    __syncthreads();
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));      // device-side row count (no host sync needed)

    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane & 31, h = lane >> 5;
    // tile element (kb, h, mt, i) sits at byte ((kb*2 + h)*128 + mt*32 + i) * 16 of its section
    const unsigned lds0 = (unsigned)(size_t)lds;        // low 32 bits of a flat LDS-aperture address = the LDS byte address
    const unsigned lane_off = lds0 + BH_FLOATS * 4 + (unsigned)(h * 128 + sl) * 16;
    const unsigned w0hi = opaque(lane_off);
    const unsigned w0lo = opaque(lane_off + W0_ELEMS * 2);             // (F16: no lo planes; the lo bases are never read)
    const unsigned w1hi = opaque(lane_off + W0_ELEMS * (F16 ? 2 : 4));
    const unsigned w1lo = opaque(lane_off + W0_ELEMS * 4 + BW1_ELEMS * 2);
    const int64_t n_tiles = (n + 255) / 256;
    const int64_t G = nblk;
    if ((int64_t)bid >= n_tiles) return;
    auto row_of = [&](int64_t tile) { const int64_t sr = tile * 256 + wave * 32 + sl; return sr < n ? sr : n - 1; };
    RowIn cur;
    { const int64_t sc = row_of(bid); load_row<VEC>(feat, fstride, auxv, sc, aux_index(aux_map, aux_mod, sc), cur); }
    f32x16 acc[4], acc2[4];
    // aux-table variant: this lane's 64 layer-1 accumulators start from the table row of its sample's aux index -- units
    // mt*32 + 8 i + 4 h + (0..3) are accumulator registers 4 i .. 4 i + 3 of tile mt: sixteen 16-byte loads per lane
    auto load_acc = [&](int64_t ai) {
        const float* tp = table + ai * HID + 4 * h;
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 t4 = *reinterpret_cast<const float4*>(tp + mt * 32 + 8 * i);
                acc[mt][4 * i] = t4.x; acc[mt][4 * i + 1] = t4.y; acc[mt][4 * i + 2] = t4.z; acc[mt][4 * i + 3] = t4.w;
            }
    };
    if (AUXT) load_acc(cur.ai);
    for (int64_t tile = bid; tile < n_tiles; tile += G) {
        const int64_t s_raw = tile * 256 + wave * 32 + sl;
        AuxPE ap = {};
        if (!AUXT) {
#pragma unroll
            for (int d = 0; d < 3; ++d) {
                fast_sincos(cur.ax[d], ap.s[d], ap.c[d]);
                ap.s2[d] = 2.0f * ap.s[d] * ap.c[d];
                ap.c2[d] = fmaf(-2.0f * ap.s[d], ap.s[d], 1.0f);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[mt][r] = 0.0f;      // bias: the constant-1 input of k slot HALF
        }
        if constexpr (F16) {       // raw features enter layer 1 as fp16 operands: keep them finite there (|f| <= 65504)
#pragma unroll
            for (int d = 0; d < F; ++d) cur.ft[d] = __builtin_amdgcn_fmed3f(cur.ft[d], -65504.0f, 65504.0f);
        }
        layer1_interleaved<NPROD, 0, AUXT, F16>(w0hi, w0lo, h, cur.ft, cur.ax, ap, 0.25f * (float)h, acc);
        // the row registers are dead from here on: fetch the NEXT tile's row into them now, so that the global-load
        // latency (the features were just written by the gather kernel: L2 / HBM) hides behind layers 2 and 3
        if (tile + G < n_tiles) {
            const int64_t sc = row_of(tile + G);
            load_row<VEC>(feat, fstride, auxv, sc, aux_index(aux_map, aux_mod, sc), cur);
        }
        if (SAVE && s_raw < n) {      // post-ReLU hidden activations in natural [sample][unit] order (training)
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(h1o + s_raw * HID + mt * 32 + 8 * i + 4 * h) =
                        make_float4(relu(acc[mt][4 * i]), relu(acc[mt][4 * i + 1]), relu(acc[mt][4 * i + 2]), relu(acc[mt][4 * i + 3]));
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* bp = lds + BH_B1 + (h * 4 + mt) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = bp[r];
        }
        layer2_interleaved<NPROD, 0, F16>(w1hi, w1lo, acc, acc2);
        // layer 2 has consumed the layer-1 values: the NEXT tile's accumulator start (table row of the row fetched above)
        // loads into the same registers now and arrives behind layer 3
        if (AUXT && tile + G < n_tiles) load_acc(cur.ai);
        if (SAVE && s_raw < n) {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    *reinterpret_cast<float4*>(h2o + s_raw * HID + mt * 32 + 8 * i + 4 * h) =
                        make_float4(relu(acc2[mt][4 * i]), relu(acc2[mt][4 * i + 1]), relu(acc2[mt][4 * i + 2]), relu(acc2[mt][4 * i + 3]));
        }
        // ---- layer 3 (128 -> out <= 4) on the matrix pipe, exact fp32: v_mfma_f32_4x4x1_16b_f32 computes 16 independent
        // 4x4 outer products per issue.  Block = 4 consecutive lanes (same half); lane 4b+j supplies B = its own sample's
        // hidden value (column j) and A = W2[row j][that unit]; D register i of lane 4b+j = out_i of lane's sample.
        // 64 issues of 8 cycles replace 256 FMAs + 64 LDS reads per lane; four accumulators break the dependency chain.
        // (register layout probed on gfx950: D reg i of lane 4b+j = A(lane 4b+i) * B(lane 4b+j).)
        f32x4 o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* wp = lds + BH_W2 + (h * 4 + (lane & 3)) * W2A_STRIDE;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(wp + 4 * g);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g + e;
                    const float x = acc2[q >> 4][q & 15];
                    // 2 relu(x) = x + |x| (one VALU op the compiler can schedule and pad: an inline-asm v_max feeding an
                    // MFMA operand would need its VALU->MFMA wait states by hand); W2a carries the exact factor 1/2
                    o4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], x + __builtin_fabsf(x), o4[e], 0, 0, 0);
                }
            }
        }
        float o0 = (o4[0][0] + o4[1][0]) + (o4[2][0] + o4[3][0]);
        float o1 = (o4[0][1] + o4[1][1]) + (o4[2][1] + o4[3][1]);
        float o2 = (o4[0][2] + o4[1][2]) + (o4[2][2] + o4[3][2]);
        float o3 = (o4[0][3] + o4[1][3]) + (o4[2][3] + o4[3][3]);
        o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        o2 += __shfl_xor(o2, 32, 64); o3 += __shfl_xor(o3, 32, 64);
        if (h == 0 && s_raw < n) {
            const float* b2 = lds + BH_B2;
            float* op = out + s_raw * out_dim;
            op[0] = act_out(o0 + b2[0], act);
            if (out_dim > 1) op[1] = act_out(o1 + b2[1], act);
            if (out_dim > 2) op[2] = act_out(o2 + b2[2], act);
            if (out_dim > 3) op[3] = act_out(o3 + b2[3], act);
        }
    }
}


template <int NPROD, bool VEC, bool SAVE>
__global__ void __launch_bounds__(512)
k_mlp_bf16(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
           const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
           const int32_t* __restrict__ n_dev, int out_dim, int act, float* __restrict__ h1o, float* __restrict__ h2o) {
    mlp_bf16_body<NPROD, VEC, SAVE>(packed, feat, fstride, aux, aux_map, aux_mod, out, n, n_dev, out_dim, act, h1o, h2o,
                                    (int)blockIdx.x, (int)gridDim.x);
}

// the aux-table variant (see R0A above): `table` [n_aux][128] from k_mlp_aux_table takes the place of the aux values
template <bool VEC, bool SAVE = false>
__global__ void __launch_bounds__(512)
k_mlp_bf16_auxt(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ table,
                const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
                const int32_t* __restrict__ n_dev, int out_dim, int act, float* __restrict__ h1o, float* __restrict__ h2o) {
    mlp_bf16_body<3, VEC, SAVE, true>(packed, feat, fstride, table, aux_map, aux_mod, out, n, n_dev, out_dim, act, h1o, h2o,
                                      (int)blockIdx.x, (int)gridDim.x);
}

// T[a][u] = b0[u] + sum over the 15 aux-dependent input columns of W0[u][col] x_col(aux_a): exact fp32 FMAs on the raw
// weights, library sin / cos (the table is tiny: one row per ray or per light direction).  Column order of the reference
// input (models/tensorBase_rotated_lights.py:137-142, :12-17): aux at F.., sin(PE aux) at F+3+2*NPF.., cos(PE aux) 3*PE later.
// ------------------------------------------------------------------------------------------------
// FUSED indirect-light kernel: appearance gather (fp16 shadow planes) -> basis_mat contraction -> radiance decoder (fp16
// operands) for the secondary-ray records, in one pass (north_star: "fused into one pass"; models/relight_utils.py:818-829 =
// compute_appfeature -> renderModule).  The feature rows never reach HBM: the gather's v_mfma_f32_32x32x16_f16 leaves lane
// (record = l & 31, half = l >> 5) with 16 of the record's 32 feature rows in its accumulators, and basis_mat's rows are dealt so
// that those are exactly the features whose PE and raw value this lane half feeds into layer 1 (image OFF_FF, kperm_f).
// 512 threads = 8 waves x 32 records, persistent over 256-record tiles; per wave and tile: [gather phase, L1-bound] then
// [decoder phase, VALU / MFMA-bound] -- the two waves of a SIMD drift out of phase and overlap the two.
// LDS: fp16 decoder image (72.9 KB) | basis_mat^T fp16 tiles (9.2 KB) | light rows fp32 | 8 X tiles (3.5 KB each).
// ------------------------------------------------------------------------------------------------
constexpr int FUS_XH = 56;                                  // X tile row stride in halves (48 channels + 8)
constexpr int FUS_WH_BYTES = 3 * 3 * 2 * 32 * 16;           // basis tiles [group][k-step][k-group][row] x 8 halves
constexpr int FUS_DECODER_PRIO = 3;   // s_setprio of k_indirect_fused's decoder phase

__device__ __forceinline__ int fus_feature_of_row(int row) {      // basis_mat row (feature) behind MFMA output row `row`, -1 = none
    const int kg = (row >> 2) & 1, slot = (row >> 3) * 4 + (row & 3);
    if (kg == 0) return slot < FUS_NF0 ? slot : -1;
    return slot < F - FUS_NF0 ? FUS_NF0 + slot : -1;
}

// input slot KK of a lane half in the fused layout: [for each own feature j: sin f, sin 2f, cos f, cos 2f, f].  Each slot evaluates
// its own PE value from fo[j] (the three-instruction range reduction is shared by the slots of a k-block through CSE): no PE
// array exists, so nothing can end up in scratch memory (an array of PE values did: 0.4 GB of HBM writes per launch, measured)
template <int KK>
__device__ __forceinline__ float fus_input(const float (&fo)[16]) {
    constexpr int j = KK / 5, kind = KK % 5;
    if constexpr (j >= FUS_NF0) return 0.0f;
    else if constexpr (kind == 4) return fo[j];
    else {
        const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;      // c_hi + c_lo = 1 / (2 pi), see pe_pair
        const float kr = rintf(fo[j] * c_hi);
        float t = fmaf(fo[j], c_hi, -kr);
        t = fmaf(fo[j], c_lo, t);
        // sin f and cos f on the transcendental unit (quarter rate), the double angle by the addition theorems on the full-rate
        // FMA pipe: sin 2f = 2 sin f cos f, cos 2f = 1 - 2 sin^2 f (28 v_sin_f32 per record instead of 56).  The values are
        // rounded to fp16 operands (2^-11) next: the ~2e-7 the identities add is far below that.  (CSE shares s / c between
        // the slots of a feature.)
        const float sn = __builtin_amdgcn_sinf(t), cs = __builtin_amdgcn_sinf(t + 0.25f);
        if constexpr (kind == 0) return sn;
        else if constexpr (kind == 1) return (sn + sn) * cs;
        else if constexpr (kind == 2) return cs;
        else return fmaf(-(sn + sn), sn, 1.0f);
    }
}

template <int KB>
__device__ __forceinline__ void fus_layer1(unsigned whi, const float (&fo)[16], f32x16 (&acc)[4]) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
    // compile-time slots (a run-time index into pe[][] would send the array to scratch memory)
    const float v[8] = {fus_input<KB * 8 + 0>(fo), fus_input<KB * 8 + 1>(fo), fus_input<KB * 8 + 2>(fo), fus_input<KB * 8 + 3>(fo),
                        fus_input<KB * 8 + 4>(fo), fus_input<KB * 8 + 5>(fo), fus_input<KB * 8 + 6>(fo), fus_input<KB * 8 + 7>(fo)};
    bf16x8 xh, xl = {};
    cvt8_f16(v, xh);
    mfma12<1, true>(ah, al, xh, xl, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < KB0A) fus_layer1<KB + 1>(whi, fo, acc);
}

typedef _Float16 fus_f16x2 __attribute__((ext_vector_type(2)));

// layer 2 from PACKED layer-1 activations: hp[p] = (relu(h[2p]), relu(h[2p+1])) as fp16 pairs in accumulator order -- the B operand of
// k-block KB is hp[4 KB .. 4 KB + 3] as it stands.  With the activations packed (32 registers) the 64 fp32 layer-1 accumulators
// are dead before layer 2 starts: acc and acc2 never coexist, which is what lets three waves share a SIMD (<= 168 VGPRs).
template <int KB>
__device__ __forceinline__ void fus_layer2p(unsigned whi, const unsigned (&hp)[32], f32x16 (&acc)[4]) {
    bf16x8 ah[4], al[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
    const u32x4_t H = {hp[4 * KB], hp[4 * KB + 1], hp[4 * KB + 2], hp[4 * KB + 3]};
    const bf16x8 xh = __builtin_bit_cast(bf16x8, H), xl = {};
    mfma12<1, true>(ah, al, xh, xl, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < KB1) fus_layer2p<KB + 1>(whi, hp, acc);
}

// one VM group of the fp16-shadow gather for one record: byte offsets of the six taps (this lane's 16-byte half-chunk) and the
// interpolation weights as fp16 pairs.  k is a constant after unrolling.
struct FusGroup { unsigned o00, o01, o10, o11, l0, l1; tir::tir_h2 w00, w01, w10, w11, wl0, wl1; };

__device__ __forceinline__ FusGroup fus_group(const TirField& f, const tir::TapQ (&ax)[3], const unsigned (&xb)[3][2], const int k) {
    constexpr int CA = 48;
    const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
    const tir::TapQ &tx = ax[m0], &ty = ax[m1], &tl = ax[vi];
    FusGroup g;
    const float w00 = tx.w.x * ty.w.x, w01 = tx.w.y * ty.w.x, w10 = tx.w.x * ty.w.y, w11 = tx.w.y * ty.w.y;
    g.w00 = tir::tir_h2{(_Float16)w00, (_Float16)w00}; g.w01 = tir::tir_h2{(_Float16)w01, (_Float16)w01};
    g.w10 = tir::tir_h2{(_Float16)w10, (_Float16)w10}; g.w11 = tir::tir_h2{(_Float16)w11, (_Float16)w11};
    g.wl0 = tir::tir_h2{(_Float16)tl.w.x, (_Float16)tl.w.x}; g.wl1 = tir::tir_h2{(_Float16)tl.w.y, (_Float16)tl.w.y};
    // row starts in bytes on the full-rate 24-bit multiplier (fp16 row pitch W * 96 B < 2^24, plane < 2^32 B: checked by the launcher)
    const unsigned row_bytes = (unsigned)f.grid[m0] * (2 * CA);
    const unsigned r0 = tir::mul_u24(ty.i0, row_bytes), r1 = tir::mul_u24(ty.i1, row_bytes);
    g.o00 = r0 + xb[m0][0]; g.o01 = r0 + xb[m0][1]; g.o10 = r1 + xb[m0][0]; g.o11 = r1 + xb[m0][1];
    g.l0 = xb[vi][0]; g.l1 = xb[vi][1];
    return g;
}

struct FusChunk { uint4 a, b, c, d, e, g; };           // the six taps of one 16-channel chunk (this lane's 8 channels)

__device__ __forceinline__ uint4 ld_u4b(const void* base, unsigned byte_off) {
    return *reinterpret_cast<const uint4*>(reinterpret_cast<const char*>(base) + byte_off);
}

__device__ __forceinline__ void fus_issue(FusChunk& c, const void* __restrict__ pl, const void* __restrict__ ln, const FusGroup& g, const int q) {
    const unsigned o = 32u * (unsigned)q;              // 16 channels x 2 B (a constant after unrolling: the immediate offset)
    c.a = ld_u4b(pl, g.o00 + o); c.b = ld_u4b(pl, g.o01 + o); c.c = ld_u4b(pl, g.o10 + o); c.d = ld_u4b(pl, g.o11 + o);
    c.e = ld_u4b(ln, g.l0 + o);  c.g = ld_u4b(ln, g.l1 + o);
}

// NW waves per workgroup (tile = 32 NW records).  PACK: the three-waves-per-SIMD form -- the aux-table row is loaded at the START of
// the decoder phase (no prefetch registers during the gather; the other two waves of the SIMD cover the latency) and layer 2 runs on
// packed activations (fus_layer2p).
template <int NW, bool PACK>
__global__ void __launch_bounds__(NW * 64)
k_indirect_fused(TirField f, TirFieldHalf fh, const float* __restrict__ packed, const float* __restrict__ xyz,
                 const int32_t* __restrict__ light_idx, const int32_t* __restrict__ rec_map, int idx_div, int aux_mod,
                 const float* __restrict__ table, float* __restrict__ out, int64_t n, const int32_t* __restrict__ n_dev,
                 int out_dim, int act, int lt_rows) {
    using namespace tir;
    constexpr int CA = 48;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < FH_FLOATS; i += NW * 64 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed + OFF_FF + i);
    f16x8* Wh = reinterpret_cast<f16x8*>(lds + FH_FLOATS);
    float* LT = lds + FH_FLOATS + FUS_WH_BYTES / 4;
    const int n_lt = lt_rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    _Float16* X = reinterpret_cast<_Float16*>(LT + (n_lt + 1) * (3 * CA)) + wave * (32 * FUS_XH);
    for (int e = threadIdx.x; e < 3 * 3 * 2 * 32; e += NW * 64) {
        const int row = e & 31, kg = (e >> 5) & 1, t = (e >> 6) % 3, k = e / 192;
        const int feat = fus_feature_of_row(row);
        f16x8 hv;
#pragma unroll
        for (int q = 0; q < 8; ++q) hv[q] = feat >= 0 ? sat_half(f.basis_t[(size_t)(k * CA + 16 * t + 8 * kg + q) * 32 + feat]) : (_Float16)0.0f;
        Wh[e] = hv;
    }
    _Float16* LT16 = reinterpret_cast<_Float16*>(LT);
    for (int i = threadIdx.x; i < n_lt * 3 * CA; i += NW * 64) LT16[i] = sat_half(f.light_line[i]);
    for (int i = threadIdx.x * 4; i < 3 * CA; i += NW * 64 * 4)
        *reinterpret_cast<float4*>(LT + n_lt * 3 * CA + i) = *reinterpret_cast<const float4*>(f.light_mean + i);
    __syncthreads();
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    const int sl = lane & 31, h = lane >> 5;               // decoder / MFMA role: record column, lane half (= k group)
    const int gj = lane >> 1, gc = lane & 1;               // gather role: record slot, which of every two 16-byte chunks
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned lane_off = lds0 + BH_FLOATS * 4 + (unsigned)(h * 128 + sl) * 16;
    const unsigned w0hi = opaque(lane_off);
    const unsigned w1hi = opaque(lane_off + BW0A_ELEMS * 2);
    constexpr int TILE = NW * 32;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const UDiv by_div = make_udiv(idx_div), by_mod = make_udiv(aux_mod);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE + wave * 32;
        // ---------------- gather phase: record gj of this wave's 32, two lanes per record
        const int64_t sg = r0 + gj, sgc = sg < n ? sg : n - 1;
        const float p[3] = {xyz[3 * sgc], xyz[3 * sgc + 1], xyz[3 * sgc + 2]};
        const float* lrow;
        const _Float16* lrow16;
        {
            unsigned lsel = rec_map ? (unsigned)rec_map[sgc] : (unsigned)sgc, rem_;      // record -> ray: / idx_div (n < 2^31, launcher)
            lsel = udiv(lsel, by_div, rem_);
            int li = light_idx[lsel];
            li = min(max(li, 0), f.n_lights - 1);
            lrow = n_lt ? LT + li * (3 * CA) : f.light_line + (size_t)li * (3 * CA);
            lrow16 = LT16 + li * (3 * CA);
        }
        // decoder role: this lane's record and its aux-table row
        const int64_t sd = r0 + sl, sdc = sd < n ? sd : n - 1;
        unsigned ai = rec_map ? (unsigned)rec_map[sdc] : (unsigned)sdc;
        if (aux_mod > 0) udiv(ai, by_mod, ai);
        f32x16 facc;
#pragma unroll
        for (int r = 0; r < 16; ++r) facc[r] = 0.0f;
        f32x16 acc[4];
        auto load_table = [&]() {       // the layer-1 accumulators' start values: aux-table row of the record's direction
            const float* tp = table + (size_t)ai * HID + 4 * h;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 t4 = *reinterpret_cast<const float4*>(tp + mt * 32 + 8 * i);
                    acc[mt][4 * i] = t4.x; acc[mt][4 * i + 1] = t4.y; acc[mt][4 * i + 2] = t4.z; acc[mt][4 * i + 3] = t4.w;
                }
        };
        {
            // Nine 16-channel chunks (3 VM groups x 3), software-pipelined THREE deep (round 6): the six 16-byte taps of chunks
            // i + 1 .. i + 3 are in flight while chunk i is interpolated on the packed fp16 pipe and contracted -- as many loads
            // in flight as the group-at-a-time schedule of round 5 (18), but no load -> wait -> compute -> MFMA sequence per
            // group any more (waiting was 0.42 of the wave cycles).  The taps of the three axes are computed once per record.
            TapQ ax[3];
            unsigned xb[3][2];                                  // byte offset of this lane's 16-byte half-chunk inside fp16 row `index`
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                ax[a] = make_tap_q(p[a], f.grid[a]);
                xb[a][0] = mul_u24(ax[a].i0, 2 * CA) + 16u * (unsigned)gc;
                xb[a][1] = mul_u24(ax[a].i1, 2 * CA) + 16u * (unsigned)gc;
            }
            FusGroup G[3];
            FusChunk ck[3];
            G[0] = fus_group(f, ax, xb, 0);
            fus_issue(ck[0], fh.aplane[0], fh.aline[0], G[0], 0);
            fus_issue(ck[1], fh.aplane[0], fh.aline[0], G[0], 1);
            fus_issue(ck[2], fh.aplane[0], fh.aline[0], G[0], 2);
#pragma unroll
            for (int i = 0; i < 9; ++i) {       // (unrolled: k, q and the ring slot are constants)
                const int k = i / 3, q = i % 3;
                FusChunk& c = ck[i % 3];
                const int ch0 = 16 * q + 8 * gc;
                *reinterpret_cast<uint4*>(X + gj * FUS_XH + ch0) =
                    h16_chunk_pk(c.a, c.b, c.c, c.d, c.e, c.g, G[k].w00, G[k].w01, G[k].w10, G[k].w11, G[k].wl0, G[k].wl1,
                                 n_lt ? *reinterpret_cast<const uint4*>(lrow16 + k * CA + ch0) : pack8_half(lrow + k * CA + ch0));
                if (i + 3 < 9) {
                    const int k2 = (i + 3) / 3, q2 = (i + 3) % 3;
                    if (q2 == 0) G[k2] = fus_group(f, ax, xb, k2);
                    fus_issue(c, fh.aplane[k2], fh.aline[k2], G[k2], q2);
                }
                __builtin_amdgcn_wave_barrier();
                const f16x8 a = Wh[((k * 3 + q) * 2 + h) * 32 + sl];
                const f16x8 b = *reinterpret_cast<const f16x8*>(X + sl * FUS_XH + 16 * q + 8 * h);
                facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, facc, 0, 0, 0);
                __builtin_amdgcn_wave_barrier();
            }
        }
        // The decoder phase runs at a raised wave priority: the waves of a SIMD tend to run the same phase at the same time, and with
        // equal priorities the arbiter lets the gather-phase VALU work of one wave delay the matrix instructions of another whose
        // accumulators everything downstream waits for.  Measured: alone on the stream 0.377-0.384 -> 0.370-0.371 ms (priority 3;
        // 1: nothing), in the bench step 0.941 -> 0.936 ms (three alternating runs each).  The hp kernel gains 3 % alone on the stream
        // with priority 1 and nothing in the step (two batches in flight): left without.
        __builtin_amdgcn_s_setprio(FUS_DECODER_PRIO);
        // ---------------- decoder phase: facc[4 i + r] = this half's feature slot 4 i + r  (half 0: features 0..13, half 1: 14..26)
        float fo[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) fo[r] = __builtin_amdgcn_fmed3f(facc[r], -65504.0f, 65504.0f);
        if (PACK) load_table();
        fus_layer1<0>(w0hi, fo, acc);
        f32x16 acc2[4];
        if constexpr (PACK) {
            unsigned hp[32];
#pragma unroll
            for (int q = 0; q < 64; q += 2) {
                // (the builtin, not the inline-asm relu_sat16: these reads come straight behind layer 1's last MFMAs, and the hazard
                //  recogniser does not insert the MFMA -> VALU wait states for inline asm -- that read stale accumulators: 1e-3 errors)
                // Convert first, then ReLU and saturation on the PACKED fp16 pipe (v_pk_max_f16 / v_pk_min_f16): the same values for every
                // finite input (rounding is monotonic, rnd(0) = 0, an overflow becomes inf and is cut to 65504), 32 fp32-class
                // instructions instead of 96 -- and packed fp16 instructions run under other waves' matrix instructions, fp32 ones
                // do not (profiles/r06_mfma_valu_overlap.txt).
                const f32x2_t v2 = {acc[q >> 4][q & 15], acc[(q + 1) >> 4][(q + 1) & 15]};
                fus_f16x2 hv = __builtin_convertvector(v2, fus_f16x2);
                hv = __builtin_elementwise_min(__builtin_elementwise_max(hv, fus_f16x2{(_Float16)0.0f, (_Float16)0.0f}),
                                               fus_f16x2{(_Float16)65504.0f, (_Float16)65504.0f});
                hp[q >> 1] = __builtin_bit_cast(unsigned, hv);
            }
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* bp = lds + BH_B1 + (h * 4 + mt) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] = bp[r];
            }
            fus_layer2p<0>(w1hi, hp, acc2);
        } else {
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) {
                const float* bp = lds + BH_B1 + (h * 4 + mt) * 16;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc2[mt][r] = bp[r];
            }
            layer2_interleaved<1, 0, true>(w1hi, w1hi, acc, acc2);
        }
        f32x4 o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* wp = lds + BH_W2 + (h * 4 + (lane & 3)) * W2A_STRIDE;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(wp + 4 * g);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g + e;
                    const float x = acc2[q >> 4][q & 15];
                    o4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], x + __builtin_fabsf(x), o4[e], 0, 0, 0);
                }
            }
        }
        float o0 = (o4[0][0] + o4[1][0]) + (o4[2][0] + o4[3][0]);
        float o1 = (o4[0][1] + o4[1][1]) + (o4[2][1] + o4[3][1]);
        float o2 = (o4[0][2] + o4[1][2]) + (o4[2][2] + o4[3][2]);
        float o3 = (o4[0][3] + o4[1][3]) + (o4[2][3] + o4[3][3]);
        o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        o2 += __shfl_xor(o2, 32, 64); o3 += __shfl_xor(o3, 32, 64);
        __builtin_amdgcn_s_setprio(0);
        if (h == 0 && sd < n) {
            const float* b2 = lds + BH_B2;
            float* op = out + sd * out_dim;
            op[0] = act_out(o0 + b2[0], act);
            if (out_dim > 1) op[1] = act_out(o1 + b2[1], act);
            if (out_dim > 2) op[2] = act_out(o2 + b2[2], act);
            if (out_dim > 3) op[3] = act_out(o3 + b2[3], act);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// HIGH-PRECISION fused indirect-light kernel (round 6): what the auto policy runs when a checkpoint's self-check rejects the fp16
// kernel above (a field trained to 300^3 does: profiles/r06_precision_trained_300.json) instead of two launches with the feature
// rows through HBM.  Same stage (models/relight_utils.py:818-829), same tile / lane roles for the decoder, but
//   * taps are the fp32 parameters themselves (3456 B per record), gathered with FOUR adjacent lanes per record reading whole
//     64-byte runs (the texture path serves a quad's 64 contiguous bytes in one cycle; two lanes x 16 B is half that rate, one lane
//     a quarter), interpolated in fp32; a wave's 32 records are gathered as two passes of 16;
//   * the plane x line x light products go through a per-wave LDS tile of ONE 16-channel chunk at a time, already split
//     x = hi + lo in fp16, and are contracted with basis_mat (hi + lo as well) by three v_mfma_f32_32x32x16_f16 per chunk
//     (hi hi + lo hi + hi lo: ~2^-21 relative per product) -- the features are fp32-grade;
//   * the decoder reads its activations as fp16 (their rounding is random and averages out over a ray's records and the light
//     directions) but its WEIGHTS as fp16 + an fp8 residue (F8 image, see OFF_F8): the residue product runs on the block-scaled fp8
//     matrix instruction (K = 64, A scale 2^-17) straight into the same accumulators.  Layer 3 exact as everywhere.
// LDS: fp16 decoder image (72.9 KB) | fp8 residue image (41.0 KB) | basis_mat^T hi / lo fp16 tiles (18.4 KB) | light rows fp32 |
// NW x 3 KB product tiles = 157.9 KB with 8 waves (two per SIMD, <= 256 VGPRs).
// ------------------------------------------------------------------------------------------------
constexpr int HP_XS = 24;                                    // product-tile row stride in halves (16 channels + 8 pad: 48 B)
constexpr int HP_X_HALVES = 2 * 32 * HP_XS;                  // hi tile + lo tile of one wave

typedef __attribute__((address_space(3))) const long lds_i64;
__device__ __forceinline__ long lds_tile8(unsigned base, int byte_off) {
    return *reinterpret_cast<lds_i64*>(base + (unsigned)byte_off);
}

// 8 floats (|v| <= 448, the largest finite e4m3: the caller clamps) -> 8 fp8, byte e = value e: the B operand of
// v_mfma_f32_32x32x16_fp8_fp8
__device__ __forceinline__ long cvt8_fp8(const float (&c)[8]) {
    int lo = 0, hi = 0;
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[0], c[1], lo, false);
    lo = __builtin_amdgcn_cvt_pk_fp8_f32(c[2], c[3], lo, true);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[4], c[5], hi, false);
    hi = __builtin_amdgcn_cvt_pk_fp8_f32(c[6], c[7], hi, true);
    return (long)(((unsigned long)(unsigned)hi << 32) | (unsigned long)(unsigned)lo);
}

// input slot KK for the fp8 residue product: the PE slots are in [-1, 1] as they are; the raw feature comes from fo8 = fo clamped to
// +-448 (16 clamps per record instead of one per input)
template <int KK>
__device__ __forceinline__ float fus_input8(const float (&fo)[16], const float (&fo8)[16]) {
    if constexpr (KK % 5 == 4 && KK / 5 < FUS_NF0) return fo8[KK / 5];
    else return fus_input<KK>(fo);
}

typedef int hp_i32x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) const u32x4_t lds_u32x4;

// the residue product of one group of four k-blocks: A = 32 bytes per lane from the F8 image (two 16-byte LDS reads), B = the
// lane's 32 fp8 activations of those k-blocks, A scaled by 2^-17 (E8M0 exponent 110), accumulated into the main accumulators
__device__ __forceinline__ void hp_residue(unsigned w8, int byte_off, const hp_i32x8& xs, f32x16 (&acc)[4]) {
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) {
        const u32x4_t lo = *reinterpret_cast<lds_u32x4*>(w8 + (unsigned)(byte_off + mt * 1024));
        const u32x4_t hi = *reinterpret_cast<lds_u32x4*>(w8 + (unsigned)(byte_off + mt * 1024 + 16));
        const hp_i32x8 a = {(int)lo[0], (int)lo[1], (int)lo[2], (int)lo[3], (int)hi[0], (int)hi[1], (int)hi[2], (int)hi[3]};
        acc[mt] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, xs, acc[mt], 0, 0, 0, 110, 0, 127);
    }
}

// layer 1 of the high-precision decoder, k-block KB: fp16 product into acc; the fp8 activations of the block are parked in xs and
// every fourth block (and the last) the residue product of the group follows into the same accumulators
template <int KB>
__device__ __forceinline__ void hp_layer1(unsigned whi, unsigned w8, const float (&fo)[16], const float (&fo8)[16], f32x16 (&acc)[4],
                                          hp_i32x8& xs) {
    bf16x8 ah[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
    const float v[8] = {fus_input<KB * 8 + 0>(fo), fus_input<KB * 8 + 1>(fo), fus_input<KB * 8 + 2>(fo), fus_input<KB * 8 + 3>(fo),
                        fus_input<KB * 8 + 4>(fo), fus_input<KB * 8 + 5>(fo), fus_input<KB * 8 + 6>(fo), fus_input<KB * 8 + 7>(fo)};
    const float v8[8] = {fus_input8<KB * 8 + 0>(fo, fo8), fus_input8<KB * 8 + 1>(fo, fo8), fus_input8<KB * 8 + 2>(fo, fo8), fus_input8<KB * 8 + 3>(fo, fo8),
                         fus_input8<KB * 8 + 4>(fo, fo8), fus_input8<KB * 8 + 5>(fo, fo8), fus_input8<KB * 8 + 6>(fo, fo8), fus_input8<KB * 8 + 7>(fo, fo8)};
    bf16x8 xh;
    cvt8_f16(v, xh);
    const long x8 = cvt8_fp8(v8);
    if constexpr (KB % 4 == 0) xs = hp_i32x8{0, 0, 0, 0, 0, 0, 0, 0};       // (padding blocks of the last group: zero activations)
    xs[2 * (KB % 4)] = (int)(unsigned)(unsigned long)x8;
    xs[2 * (KB % 4) + 1] = (int)(unsigned)((unsigned long)x8 >> 32);
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, xh), acc[mt], 0, 0, 0);
    if constexpr (KB % 4 == 3 || KB + 1 == KB0A) hp_residue(w8, (KB / 4) * 8192, xs, acc);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < KB0A) hp_layer1<KB + 1>(whi, w8, fo, fo8, acc, xs);
}

// layer 2 from packed activations: hp[p] = fp16 pair, h8[p] = fp8 quad of relu(h) in accumulator order
template <int KB>
__device__ __forceinline__ void hp_layer2(unsigned whi, unsigned w8, const unsigned (&hp)[32], const unsigned (&h8)[16], f32x16 (&acc)[4]) {
    bf16x8 ah[4];
#pragma unroll
    for (int mt = 0; mt < 4; ++mt) ah[mt] = lds_tile(whi, KB * 4096 + mt * 512);
    const u32x4_t H = {hp[4 * KB], hp[4 * KB + 1], hp[4 * KB + 2], hp[4 * KB + 3]};
#pragma unroll
    for (int mt = 0; mt < 4; ++mt)
        acc[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ah[mt]), __builtin_bit_cast(f16x8, H), acc[mt], 0, 0, 0);
    if constexpr (KB % 4 == 3) {
        constexpr int g = KB / 4;
        const hp_i32x8 xs = {(int)h8[8 * g], (int)h8[8 * g + 1], (int)h8[8 * g + 2], (int)h8[8 * g + 3],
                             (int)h8[8 * g + 4], (int)h8[8 * g + 5], (int)h8[8 * g + 6], (int)h8[8 * g + 7]};
        hp_residue(w8, g * 8192, xs, acc);
    }
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (KB + 1 < KB1) hp_layer2<KB + 1>(whi, w8, hp, h8, acc);
}

// Per record: the taps of the three axes (computed ONCE, shared by the two planes and the line that use each axis -- the
// density march's scheme, tir_common.hpp make_tap_q) and the byte offsets of this lane's 16-byte quarter inside row `index`.
// A texel / line row of the appearance field is 48 channels x 4 B = 192 B.
constexpr unsigned HP_TB = 192;
struct HpAxes { tir::TapQ t[3]; unsigned ob[3][2]; };

__device__ __forceinline__ HpAxes hp_axes(const TirField& f, const float (&p)[3], int c) {
    HpAxes A;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        A.t[a] = tir::make_tap_q(p[a], f.grid[a]);
        A.ob[a][0] = tir::mul_u24(A.t[a].i0, HP_TB) + 16u * (unsigned)c;
        A.ob[a][1] = tir::mul_u24(A.t[a].i1, HP_TB) + 16u * (unsigned)c;
    }
    return A;
}

// the six tap offsets (bytes) and the packed weights of VM group k for one record.  k is a constant after unrolling (a run-time
// k would turn the axis selects into a scratch table, see k_indirect_fused).
struct HpGroup { unsigned o00, o01, o10, o11, l0, l1; tir::tir_f2 wa, wb, wl; };

__device__ __forceinline__ HpGroup hp_group(const TirField& f, const HpAxes& A, const int k) {
    const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
    const tir::TapQ &tx = A.t[m0], &ty = A.t[m1], &tl = A.t[vi];
    HpGroup g;
    g.wa = tx.w * tir::tir_f2{ty.w.x, ty.w.x};       // (w00, w01)
    g.wb = tx.w * tir::tir_f2{ty.w.y, ty.w.y};       // (w10, w11)
    g.wl = tl.w;
    // row starts in bytes on the full-rate 24-bit multiplier (row bytes < 2^24 and plane bytes < 2^32: checked by the launcher)
    const unsigned row_bytes = (unsigned)f.grid[m0] * HP_TB;
    const unsigned r0 = tir::mul_u24(ty.i0, row_bytes), r1 = tir::mul_u24(ty.i1, row_bytes);
    g.o00 = r0 + A.ob[m0][0]; g.o01 = r0 + A.ob[m0][1]; g.o10 = r1 + A.ob[m0][0]; g.o11 = r1 + A.ob[m0][1];
    g.l0 = A.ob[vi][0]; g.l1 = A.ob[vi][1];
    return g;
}

// the 12 loads of one 16-channel chunk: six taps x two passes (records gj and 16 + gj), this lane's 16-byte quarter
struct HpChunk { float4 a0, b0, c0, d0, e0, g0, a1, b1, c1, d1, e1, g1; };

__device__ __forceinline__ void hp_issue(HpChunk& c, const float* __restrict__ pl, const float* __restrict__ ln, const HpGroup& tA,
                                         const HpGroup& tB, const int q) {
    const unsigned o = 64u * (unsigned)q;            // (a constant after unrolling: the instruction's immediate offset)
    c.a0 = tir::ld4b(pl, tA.o00 + o); c.b0 = tir::ld4b(pl, tA.o01 + o); c.c0 = tir::ld4b(pl, tA.o10 + o);
    c.d0 = tir::ld4b(pl, tA.o11 + o); c.e0 = tir::ld4b(ln, tA.l0 + o);  c.g0 = tir::ld4b(ln, tA.l1 + o);
    c.a1 = tir::ld4b(pl, tB.o00 + o); c.b1 = tir::ld4b(pl, tB.o01 + o); c.c1 = tir::ld4b(pl, tB.o10 + o);
    c.d1 = tir::ld4b(pl, tB.o11 + o); c.e1 = tir::ld4b(ln, tB.l0 + o);  c.g1 = tir::ld4b(ln, tB.l1 + o);
}

// plane x line x light of 4 channels -> fp16 hi / lo halves (8 B each) of the product-tile row.  Scalar fp32 chains (the
// arithmetic of k_vm_app_mfma).  NOT the packed-fp32 form of the density march: with v_pk_mul_f32 / v_pk_fma_f32 here, results of
// the last 16 lanes of a wave changed from run to run (a few records in 10^4, 1e-4 off; found with tools/hp_debug.py, bisected to
// exactly this choice by build variants -- waits, issue order, inline-asm helpers and the LDS light rows all ruled out); the scalar
// form is bit-reproducible.
__device__ __forceinline__ void hp_products(const float4& a, const float4& b, const float4& cc, const float4& d, const float4& e,
                                            const float4& g, const HpGroup& t, const float4& lr, _Float16* __restrict__ xh,
                                            _Float16* __restrict__ xl) {
    float val[4];
    val[0] = fmaf(d.x, t.wb.y, fmaf(cc.x, t.wb.x, fmaf(b.x, t.wa.y, a.x * t.wa.x))) * fmaf(g.x, t.wl.y, e.x * t.wl.x) * lr.x;
    val[1] = fmaf(d.y, t.wb.y, fmaf(cc.y, t.wb.x, fmaf(b.y, t.wa.y, a.y * t.wa.x))) * fmaf(g.y, t.wl.y, e.y * t.wl.x) * lr.y;
    val[2] = fmaf(d.z, t.wb.y, fmaf(cc.z, t.wb.x, fmaf(b.z, t.wa.y, a.z * t.wa.x))) * fmaf(g.z, t.wl.y, e.z * t.wl.x) * lr.z;
    val[3] = fmaf(d.w, t.wb.y, fmaf(cc.w, t.wb.x, fmaf(b.w, t.wa.y, a.w * t.wa.x))) * fmaf(g.w, t.wl.y, e.w * t.wl.x) * lr.w;
    unsigned hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = __builtin_amdgcn_fmed3f(val[2 * q], -65504.0f, 65504.0f), x1 = __builtin_amdgcn_fmed3f(val[2 * q + 1], -65504.0f, 65504.0f);
        const f32x2_t x = {x0, x1};
        hi[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(x, f16x2_t));
        // residue x - float(hi) in one v_fma_mix_f32 per value (the fp16 operand is read in place)
        const f32x2_t r = {tir::fma_mix_lo(hi[q], -1.0f, x0), tir::fma_mix_hi(hi[q], -1.0f, x1)};
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    }
    *reinterpret_cast<uint2*>(xh) = make_uint2(hi[0], hi[1]);
    *reinterpret_cast<uint2*>(xl) = make_uint2(lo[0], lo[1]);
}

template <int NW>
__global__ void __launch_bounds__(NW * 64)
k_indirect_fused_hp(TirField f, const float* __restrict__ packed, const float* __restrict__ xyz,
                    const int32_t* __restrict__ light_idx, const int32_t* __restrict__ rec_map, int idx_div, int aux_mod,
                    const float* __restrict__ table, float* __restrict__ out, int64_t n, const int32_t* __restrict__ n_dev,
                    int out_dim, int act, int lt_rows) {
    using namespace tir;
    constexpr int CA = 48;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < FH_FLOATS; i += NW * 64 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed + OFF_FF + i);
    for (int i = threadIdx.x * 4; i < F8_FLOATS; i += NW * 64 * 4)
        *reinterpret_cast<float4*>(lds + FH_FLOATS + i) = *reinterpret_cast<const float4*>(packed + OFF_F8 + i);
    f16x8* Wh = reinterpret_cast<f16x8*>(lds + FH_FLOATS + F8_FLOATS);
    f16x8* Wl = Wh + 3 * 3 * 2 * 32;
    float* LT = lds + FH_FLOATS + F8_FLOATS + 2 * (FUS_WH_BYTES / 4);
    const int n_lt = lt_rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    _Float16* Xh = reinterpret_cast<_Float16*>(LT + n_lt * (3 * CA)) + wave * HP_X_HALVES;
    _Float16* Xl = Xh + 32 * HP_XS;
    for (int e = threadIdx.x; e < 3 * 3 * 2 * 32; e += NW * 64) {
        const int row = e & 31, kg = (e >> 5) & 1, t = (e >> 6) % 3, k = e / 192;
        const int feat = fus_feature_of_row(row);
        f16x8 hv, lv;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const float b = feat >= 0 ? f.basis_t[(size_t)(k * CA + 16 * t + 8 * kg + q) * 32 + feat] : 0.0f;
            hv[q] = sat_half(b);
            lv[q] = (_Float16)(b - (float)hv[q]);
        }
        Wh[e] = hv; Wl[e] = lv;
    }
    for (int i = threadIdx.x * 4; i < n_lt * 3 * CA; i += NW * 64 * 4)
        *reinterpret_cast<float4*>(LT + i) = *reinterpret_cast<const float4*>(f.light_line + i);
    __syncthreads();
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    const int sl = lane & 31, h = lane >> 5;               // decoder / MFMA role: record column, lane half (= k group)
    const int gj = lane >> 2, gc = lane & 3;               // gather role: record slot of a 16-record pass, 16-byte quarter of a 64-byte run
    const unsigned lds0 = (unsigned)(size_t)lds;
    const unsigned lane_off = lds0 + BH_FLOATS * 4 + (unsigned)(h * 128 + sl) * 16;
    const unsigned w0hi = opaque(lane_off);
    const unsigned w1hi = opaque(lane_off + BW0A_ELEMS * 2);
    const unsigned w0f8 = opaque(lds0 + FH_BYTES + (unsigned)(h * 4096 + sl * 32));
    const unsigned w1f8 = opaque(lds0 + FH_BYTES + F8_W0_BYTES + (unsigned)(h * 4096 + sl * 32));
    constexpr int TILE = NW * 32;
    const int64_t n_tiles = (n + TILE - 1) / TILE;
    const UDiv by_div = make_udiv(idx_div), by_mod = make_udiv(aux_mod);
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t r0 = tile * TILE + wave * 32;
        // ---------------- gather phase: records r0 + gj (pass A) and r0 + 16 + gj (pass B), four lanes per record
        float pA[3], pB[3];
        const float *lrA, *lrB;
        {
            const int64_t sa = r0 + gj, sb = r0 + 16 + gj;
            const int64_t sac = sa < n ? sa : n - 1, sbc = sb < n ? sb : n - 1;
            pA[0] = xyz[3 * sac]; pA[1] = xyz[3 * sac + 1]; pA[2] = xyz[3 * sac + 2];
            pB[0] = xyz[3 * sbc]; pB[1] = xyz[3 * sbc + 1]; pB[2] = xyz[3 * sbc + 2];
            unsigned la = rec_map ? (unsigned)rec_map[sac] : (unsigned)sac, lb = rec_map ? (unsigned)rec_map[sbc] : (unsigned)sbc, rem_;
            la = udiv(la, by_div, rem_); lb = udiv(lb, by_div, rem_);
            int lia = light_idx[la], lib = light_idx[lb];
            lia = min(max(lia, 0), f.n_lights - 1); lib = min(max(lib, 0), f.n_lights - 1);
            lrA = LT + lia * (3 * CA);         // (the launcher stages every light row in LDS: n_lights <= 16)
            lrB = LT + lib * (3 * CA);
        }
        // decoder role: this lane's record and its aux-table row
        const int64_t sd = r0 + sl, sdc = sd < n ? sd : n - 1;
        unsigned ai = rec_map ? (unsigned)rec_map[sdc] : (unsigned)sdc;
        if (aux_mod > 0) udiv(ai, by_mod, ai);
        f32x16 facc;
#pragma unroll
        for (int r = 0; r < 16; ++r) facc[r] = 0.0f;
        f32x16 acc[4];
        auto load_table = [&]() {       // layer 1 starts from the aux-table row of the record's direction (sixteen 16-byte loads per lane)
            const float* tp = table + (size_t)ai * HID + 4 * h;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 t4 = *reinterpret_cast<const float4*>(tp + mt * 32 + 8 * i);
                    acc[mt][4 * i] = t4.x; acc[mt][4 * i + 1] = t4.y; acc[mt][4 * i + 2] = t4.z; acc[mt][4 * i + 3] = t4.w;
                }
        };
        {
            // nine 16-channel chunks (3 VM groups x 3), software-pipelined two deep: while chunk i is interpolated and contracted,
            // the 12 loads of chunks i + 1 (and, once i is consumed, i + 2) are in flight -- no bubble at the group boundaries
            const HpAxes axA = hp_axes(f, pA, gc), axB = hp_axes(f, pB, gc);
            HpGroup GA[3], GB[3];
            HpChunk ck[2];
            GA[0] = hp_group(f, axA, 0); GB[0] = hp_group(f, axB, 0);
            hp_issue(ck[0], f.aplane[0], f.aline[0], GA[0], GB[0], 0);
            hp_issue(ck[1], f.aplane[0], f.aline[0], GA[0], GB[0], 1);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                const int k = i / 3, q = i % 3;
                HpChunk& c = ck[i & 1];
                const int ch = k * CA + 16 * q + 4 * gc;
                hp_products(c.a0, c.b0, c.c0, c.d0, c.e0, c.g0, GA[k], tir::ld4(lrA + ch), Xh + gj * HP_XS + 4 * gc, Xl + gj * HP_XS + 4 * gc);
                hp_products(c.a1, c.b1, c.c1, c.d1, c.e1, c.g1, GB[k], tir::ld4(lrB + ch), Xh + (16 + gj) * HP_XS + 4 * gc, Xl + (16 + gj) * HP_XS + 4 * gc);
                if (i + 2 < 9) {
                    const int k2 = (i + 2) / 3, q2 = (i + 2) % 3;
                    if (q2 == 0) { GA[k2] = hp_group(f, axA, k2); GB[k2] = hp_group(f, axB, k2); }
                    hp_issue(c, f.aplane[k2], f.aline[k2], GA[k2], GB[k2], q2);
                } else if (i == 7) {
                    load_table();       // the tap pipeline is draining: the decoder's start values travel behind the last chunk
                }
                __builtin_amdgcn_wave_barrier();
                const f16x8 ah = Wh[((k * 3 + q) * 2 + h) * 32 + sl], al = Wl[((k * 3 + q) * 2 + h) * 32 + sl];
                const f16x8 bh = *reinterpret_cast<const f16x8*>(Xh + sl * HP_XS + 8 * h);
                const f16x8 bl = *reinterpret_cast<const f16x8*>(Xl + sl * HP_XS + 8 * h);
                facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, facc, 0, 0, 0);
                facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, facc, 0, 0, 0);
                facc = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, facc, 0, 0, 0);
                __builtin_amdgcn_wave_barrier();
            }
        }
        // ---------------- decoder phase: facc[4 i + r] = this half's feature slot 4 i + r  (half 0: features 0..13, half 1: 14..26)
        float fo[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) fo[r] = __builtin_amdgcn_fmed3f(facc[r], -65504.0f, 65504.0f);
        float fo8[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) fo8[r] = __builtin_amdgcn_fmed3f(fo[r], -448.0f, 448.0f);
        {
            hp_i32x8 xs;
            hp_layer1<0>(w0hi, w0f8, fo, fo8, acc, xs);
        }
        unsigned hp[32], h8[16];
#pragma unroll
        for (int q = 0; q < 64; q += 4) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = __builtin_amdgcn_fmed3f(acc[(q + e) >> 4][(q + e) & 15], 0.0f, 65504.0f);
            const f32x2_t v01 = {v[0], v[1]}, v23 = {v[2], v[3]};
            hp[q >> 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v01, fus_f16x2));
            hp[(q >> 1) + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector(v23, fus_f16x2));
            int w = 0;
            w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(v[0], 448.0f), fminf(v[1], 448.0f), w, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32(fminf(v[2], 448.0f), fminf(v[3], 448.0f), w, true);
            h8[q >> 2] = (unsigned)w;
        }
        f32x16 acc2[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt) {
            const float* bp = lds + BH_B1 + (h * 4 + mt) * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc2[mt][r] = bp[r];
        }
        hp_layer2<0>(w1hi, w1f8, hp, h8, acc2);
        f32x4 o4[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) o4[c] = f32x4{0.f, 0.f, 0.f, 0.f};
        {
            const float* wp = lds + BH_W2 + (h * 4 + (lane & 3)) * W2A_STRIDE;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                const float4 w = *reinterpret_cast<const float4*>(wp + 4 * g);
                const float wv[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int q = 4 * g + e;
                    const float x = acc2[q >> 4][q & 15];
                    o4[e] = __builtin_amdgcn_mfma_f32_4x4x1f32(wv[e], x + __builtin_fabsf(x), o4[e], 0, 0, 0);
                }
            }
        }
        float o0 = (o4[0][0] + o4[1][0]) + (o4[2][0] + o4[3][0]);
        float o1 = (o4[0][1] + o4[1][1]) + (o4[2][1] + o4[3][1]);
        float o2 = (o4[0][2] + o4[1][2]) + (o4[2][2] + o4[3][2]);
        float o3 = (o4[0][3] + o4[1][3]) + (o4[2][3] + o4[3][3]);
        o0 += __shfl_xor(o0, 32, 64); o1 += __shfl_xor(o1, 32, 64);
        o2 += __shfl_xor(o2, 32, 64); o3 += __shfl_xor(o3, 32, 64);
        if (h == 0 && sd < n) {
            const float* b2 = lds + BH_B2;
            float* op = out + sd * out_dim;
            op[0] = act_out(o0 + b2[0], act);
            if (out_dim > 1) op[1] = act_out(o1 + b2[1], act);
            if (out_dim > 2) op[2] = act_out(o2 + b2[2], act);
            if (out_dim > 3) op[3] = act_out(o3 + b2[3], act);
        }
    }
}

// single-product fp16 form of the aux-table decoder (see FH_BYTES above): same lane decomposition, same layer 3 (exact fp32)
template <bool VEC>
__global__ void __launch_bounds__(512)
k_mlp_f16_auxt(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ table,
               const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
               const int32_t* __restrict__ n_dev, int out_dim, int act) {
    mlp_bf16_body<1, VEC, false, true, true>(packed, feat, fstride, table, aux_map, aux_mod, out, n, n_dev, out_dim, act, nullptr, nullptr,
                                             (int)blockIdx.x, (int)gridDim.x);
}

__global__ void k_mlp_aux_table(const float* __restrict__ packed, const float* __restrict__ aux, int64_t n_aux,
                                float* __restrict__ table) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_aux * HID) return;
    const int64_t a = i / HID;
    const int u = (int)(i % HID);
    const float* w = packed + OFF_RW0 + u * IN;
    float acc = packed[OFF_RB0 + u];
#pragma unroll
    for (int d = 0; d < 3; ++d) {
        const float x = aux[3 * a + d];
        acc = fmaf(w[F + d], x, acc);
#pragma unroll
        for (int f = 0; f < PE; ++f) {
            const float y = x * (float)(1 << f);
            acc = fmaf(w[F + 3 + 2 * NPF + d * PE + f], sinf(y), acc);
            acc = fmaf(w[F + 3 + 2 * NPF + 3 * PE + d * PE + f], cosf(y), acc);
        }
    }
    table[i] = acc;
}

// Several decoders over the same number of rows in ONE launch (the primary stage runs rgb / brdf / jittered brdf / normal
// on the same records): the grid is split evenly, a workgroup loads ITS decoder's operand image once and walks that
// decoder's tiles.  Against one launch per decoder: one 150 KB LDS fill per workgroup instead of four, one tail instead
// of four (at 230 k rows a launch is only 3.5 tiles per workgroup).
struct TirMlpJob { const float* packed; const float* feat; const float* aux; const int32_t* aux_map; float* out; int out_dim, act;
                   float* h1; float* h2; const float* table; };      // table != NULL: the aux-table variant for this job
struct TirMlpJobs { TirMlpJob j[4]; int n_jobs; };

template <int NPROD, bool SAVE = false>
__global__ void __launch_bounds__(512)
k_mlp_bf16_multi(TirMlpJobs jobs, int fstride, int64_t n, const int32_t* __restrict__ n_dev) {
    const int per = (int)gridDim.x / jobs.n_jobs;
    const int ji = (int)blockIdx.x / per;
    if (ji >= jobs.n_jobs) return;
    const TirMlpJob& jb = jobs.j[ji];
    if (!SAVE && jb.table)
        mlp_bf16_body<NPROD, true, false, true>(jb.packed, jb.feat, fstride, jb.table, jb.aux_map, 0, jb.out, n, n_dev, jb.out_dim,
                                                jb.act, nullptr, nullptr, (int)blockIdx.x - ji * per, per);
    else
        mlp_bf16_body<NPROD, true, SAVE>(jb.packed, jb.feat, fstride, jb.aux, jb.aux_map, 0, jb.out, n, n_dev, jb.out_dim, jb.act,
                                         jb.h1, jb.h2, (int)blockIdx.x - ji * per, per);
}

// ------------------------------------------------------------------------------------------------
// plain VALU kernel (one sample per lane, weights by wave-uniform scalar loads): cross-check only
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(64)
k_mlp_valu(const float* __restrict__ packed, const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
           const int32_t* __restrict__ aux_map, int aux_mod, float* __restrict__ out, int64_t n,
           const int32_t* __restrict__ n_dev, int out_dim, int act) {
    int64_t s = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (s >= n) return;
    float x[IN];
    int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
            if (aux_mod > 0) ai %= aux_mod;
    for (int d = 0; d < F; ++d) x[d] = feat[s * fstride + d];
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


// ------------------------------------------------------------------------------------------------
// Training: decoder input rows, backward-data kernel.
// ------------------------------------------------------------------------------------------------
constexpr int XPAD = 160;                        // input row stride of tir_mlp_inputs / W0^T row count
// backward blob (floats): W2 [2][64][4] | W1^T [64*2][128] | W0^T [64*2][160]
constexpr int OFFB_W2 = 0;
constexpr int OFFB_W1T = OFFB_W2 + 2 * 64 * 4;
constexpr int OFFB_W0T = OFFB_W1T + 128 * HID;
constexpr int BWD_FLOATS = OFFB_W0T + 128 * XPAD;          // 37376 floats = 149,504 B of LDS
// split-bf16 image appended to the blob (tir_mlp_bwd_bf16x3): W1^T hi | W1^T lo | W0^T hi | W0^T lo as bf16x8 operand
// vectors [(kb*2 + h)*ROWS + row] -- the forward image's layout with the transposed matrices; k slot (kb, h, e) is the
// hidden unit unit_of(8 kb + e, h), rows of W0^T permuted by bwd_inrow.
constexpr int BB_W1T_ELEMS = HID * HID;                    // 16384 bf16 per plane
constexpr int BB_W0T_ELEMS = XPAD * HID;                   // 20480
constexpr int BB_ELEMS = 2 * BB_W1T_ELEMS + 2 * BB_W0T_ELEMS;
constexpr int BWD_BF_FLOATS = BB_ELEMS / 2;                // 36864 float slots
constexpr int BWD_TOTAL_FLOATS = BWD_FLOATS + BWD_BF_FLOATS;
constexpr int BWD_BF_LDS_BYTES = (2 * 64 * 4) * 4 + BB_ELEMS * 2;      // W2 (fp32) + the bf16 image = 149,504 B

// MFMA output row R (0..159) of d x^T = W0^T dz1^T  ->  decoder input index it carries (or -1).
// Rows are permuted so that lane half h ends up holding, for feature d = 16 h + u/5 (u = 16 mt + r its accumulator
// slot), the 5 cotangents {raw, sin f0, sin f1, cos f0, cos f1} the positional-encoding chain rule needs.
__host__ __device__ inline int bwd_inrow(int R) {
    const int mt = R >> 5, within = R & 31;
    const int hh = (within >> 2) & 1, r = (within & 3) + 4 * (within >> 3);
    const int u = mt * 16 + r;
    const int d = hh * 16 + u / 5, comp = u % 5;
    if (d >= F) return -1;
    switch (comp) {
        case 0: return d;
        case 1: return F + 3 + d * PE;
        case 2: return F + 3 + d * PE + 1;
        case 3: return F + 3 + NPF + d * PE;
        default: return F + 3 + NPF + d * PE + 1;
    }
}

__global__ void k_pack_mlp_bwd(const float* __restrict__ w0, const float* __restrict__ w1, const float* __restrict__ w2,
                               int out_dim, float* __restrict__ p) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= BWD_TOTAL_FLOATS) return;
    float v = 0.0f;
    if (i >= BWD_FLOATS) {                          // two bf16 per float slot of the split-bf16 image
        unsigned short hw[2];
        for (int t = 0; t < 2; ++t) {
            int e_all = (i - BWD_FLOATS) * 2 + t;
            const bool is_w1 = e_all < 2 * BB_W1T_ELEMS;
            const int plane_elems = is_w1 ? BB_W1T_ELEMS : BB_W0T_ELEMS;
            const int rel = is_w1 ? e_all : e_all - 2 * BB_W1T_ELEMS;
            const bool lo = rel >= plane_elems;
            const int idx = lo ? rel - plane_elems : rel;
            const int ntile = is_w1 ? 4 : 5;
            const int e = idx % 8, ii = (idx / 8) % 32, mt = (idx / 256) % ntile, hh = (idx / (256 * ntile)) % 2,
                      kb = idx / (512 * ntile);
            const int unit = unit_of(kb * 8 + e, hh);
            float wv;
            if (is_w1) wv = w1[unit * HID + (mt * 32 + ii)];                      // W1^T[row][k = unit]
            else { const int in = bwd_inrow(mt * 32 + ii); wv = (in >= 0) ? w0[unit * IN + in] : 0.0f; }
            const __bf16 hi = (__bf16)wv;
            const __bf16 r = lo ? (__bf16)(wv - (float)hi) : hi;
            hw[t] = __builtin_bit_cast(unsigned short, r);
        }
        p[i] = __builtin_bit_cast(float, (unsigned int)hw[0] | ((unsigned int)hw[1] << 16));
        return;
    }
    if (i < OFFB_W1T) {
        int j = i - OFFB_W2, h = j / 256, q = (j % 256) / 4, o = j % 4;
        v = (o < out_dim) ? w2[o * HID + unit_of(q, h)] : 0.0f;
    } else if (i < OFFB_W0T) {
        int j = i - OFFB_W1T, th = j / 128, rem = j % 128, ii = rem / 4, mt = rem % 4;
        v = w1[unit_of(th >> 1, th & 1) * HID + (mt * 32 + ii)];          // W1^T[row = mt*32+ii][k = unit]
    } else {
        int j = i - OFFB_W0T, th = j / XPAD, rem = j % XPAD, ii = rem / 5, mt = rem % 5;
        int in = bwd_inrow(mt * 32 + ii);
        v = (in >= 0) ? w0[unit_of(th >> 1, th & 1) * IN + in] : 0.0f;
    }
    p[i] = v;
}

// one decoder-input column of one row (see the forward's input build for the layout)
__device__ __forceinline__ float mlp_input_col(const float* __restrict__ frow, const float* __restrict__ arow, int col) {
    if (col < F) return frow[col];
    if (col < F + 3) return arow[col - F];
    if (col >= IN) return 0.0f;
    const bool from_aux = col >= F + 3 + 2 * NPF;
    const int q = from_aux ? col - (F + 3 + 2 * NPF) : col - (F + 3);
    const int half = from_aux ? 3 * PE : NPF;            // sin block, then cos block
    const bool is_cos = q >= half;
    const int qq = is_cos ? q - half : q;
    const float base = from_aux ? arow[qq / PE] : frow[qq / PE];
    float sv, cv;
    fast_sincos(base * (float)(1 << (qq % PE)), sv, cv);  // transcendental unit, <= 4e-7 abs (these rows only feed d W0 = dz1^T x)
    return is_cos ? cv : sv;
}

// 320 threads = 8 rows x 40 float4 granules: every row leaves as 640 contiguous bytes
__global__ void __launch_bounds__(320)
k_mlp_inputs(const float* __restrict__ feat, int fstride, const float* __restrict__ aux, const int32_t* __restrict__ aux_map,
             int aux_mod, float* __restrict__ x, int64_t n) {
    const int r = threadIdx.x / (XPAD / 4), g = threadIdx.x % (XPAD / 4);
    const int64_t s = (int64_t)blockIdx.x * 8 + r;
    if (s >= n) return;
    int64_t ai = aux_map ? (int64_t)aux_map[s] : s;
    if (aux_mod > 0) ai %= aux_mod;
    const float* frow = feat + s * fstride;
    const float* arow = aux + 3 * ai;
    float4 v;
    v.x = mlp_input_col(frow, arow, 4 * g);     v.y = mlp_input_col(frow, arow, 4 * g + 1);
    v.z = mlp_input_col(frow, arow, 4 * g + 2); v.w = mlp_input_col(frow, arow, 4 * g + 3);
    *reinterpret_cast<float4*>(x + s * XPAD + 4 * g) = v;
}

// Backward-data: 8 waves x 32 samples per 256-sample tile, same lane decomposition as the forward kernels
// (lane = sample l&31, half h = l>>5 holds hidden units unit_of(q, h)).  d h^T = W^T dz^T has exactly the forward's
// shape with the transposed weights as the A operand, so the lane that holds h[unit] receives d h[unit].
__global__ void __launch_bounds__(512)
k_mlp_bwd(const float* __restrict__ packed_bwd, const float* __restrict__ feat, int fstride, const float* __restrict__ out,
          const float* __restrict__ g_out, const float* __restrict__ h1, const float* __restrict__ h2, int64_t n,
          int out_dim, int act, float* __restrict__ g_feat, float* __restrict__ dz1o, float* __restrict__ dz2o,
          float* __restrict__ dz3o) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    for (int i = threadIdx.x * 4; i < BWD_FLOATS; i += 512 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed_bwd + i);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 255) / 256;
    for (int64_t tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
        const int64_t s_raw = tile * 256 + wave * 32 + sl;
        const bool on = s_raw < n;
        const int64_t s = on ? s_raw : n - 1;
        // ---- output layer ----
        float dz3[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float y = 0.f, gy = 0.f;
            if (o < out_dim) { y = out[s * out_dim + o]; gy = on ? g_out[s * out_dim + o] : 0.f; }
            dz3[o] = gy * (act == 1 ? (1.0f - y * y) : y * (1.0f - y));
        }
        if (on && h == 0) *reinterpret_cast<float4*>(dz3o + s * 4) = make_float4(dz3[0], dz3[1], dz3[2], dz3[3]);
        float dz[64];
        {
            const float* wp = lds + OFFB_W2 + h * 256;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 hv = *reinterpret_cast<const float4*>(h2 + s * HID + mt * 32 + 8 * i + 4 * h);
                    const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
                    float o4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = mt * 16 + 4 * i + j;
                        const float4 w = *reinterpret_cast<const float4*>(wp + q * 4);
                        const float d = w.x * dz3[0] + w.y * dz3[1] + w.z * dz3[2] + w.w * dz3[3];
                        o4[j] = hvv[j] > 0.f ? d : 0.f;
                        dz[q] = o4[j];
                    }
                    if (on) *reinterpret_cast<float4*>(dz2o + s * HID + mt * 32 + 8 * i + 4 * h) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                }
        }
        // ---- d h1^T = W1^T dz2^T ----
        f32x16 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
        {
            const float* wp = lds + OFFB_W1T + h * 128 + sl * 4;
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                const float4 a = *reinterpret_cast<const float4*>(wp + t * 256);
                const float b = dz[t];
                acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b, acc[0], 0, 0, 0);
                acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b, acc[1], 0, 0, 0);
                acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b, acc[2], 0, 0, 0);
                acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b, acc[3], 0, 0, 0);
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 hv = *reinterpret_cast<const float4*>(h1 + s * HID + mt * 32 + 8 * i + 4 * h);
                const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o4[j] = hvv[j] > 0.f ? acc[mt][4 * i + j] : 0.f;
                    dz[mt * 16 + 4 * i + j] = o4[j];
                }
                if (on) *reinterpret_cast<float4*>(dz1o + s * HID + mt * 32 + 8 * i + 4 * h) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
        // ---- d x^T = W0^T dz1^T (rows permuted, see bwd_inrow) ----
        f32x16 ax[5];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ax[mt][r] = 0.f;
        {
            const float* wp = lds + OFFB_W0T + h * XPAD + sl * 5;
#pragma unroll
            for (int t = 0; t < 64; ++t) {
                const float* a = wp + t * (2 * XPAD);
                const float b = dz[t];
                ax[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b, ax[0], 0, 0, 0);
                ax[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b, ax[1], 0, 0, 0);
                ax[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b, ax[2], 0, 0, 0);
                ax[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b, ax[3], 0, 0, 0);
                ax[4] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[4], b, ax[4], 0, 0, 0);
                if ((t & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
        // ---- positional-encoding chain rule: x = [f, sin f, sin 2f, cos f, cos 2f] ----
        float gf[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = h * 16 + j;
            float v = 0.f;
            if (d < F) {
                const float xv = feat[s * fstride + d];
                float s1, c1;
                fast_sincos(xv, s1, c1);
                const float s2 = 2.0f * s1 * c1, c2 = fmaf(-2.0f * s1, s1, 1.0f);      // double-angle identities
                const int u0 = 5 * j;
                const float g_raw = ax[(u0) >> 4][(u0) & 15], g_s0 = ax[(u0 + 1) >> 4][(u0 + 1) & 15];
                const float g_s1 = ax[(u0 + 2) >> 4][(u0 + 2) & 15], g_c0 = ax[(u0 + 3) >> 4][(u0 + 3) & 15];
                const float g_c1 = ax[(u0 + 4) >> 4][(u0 + 4) & 15];
                v = g_raw + c1 * g_s0 + 2.0f * c2 * g_s1 - s1 * g_c0 - 2.0f * s2 * g_c1;
            }
            gf[j] = v;
        }
        if (on) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(g_feat + s * 32 + h * 16 + 4 * i) = make_float4(gf[4 * i], gf[4 * i + 1], gf[4 * i + 2], gf[4 * i + 3]);
        }
    }
}

// The same on the bf16 matrix pipe: both W^T dz^T products with split operands (x = hi + lo, 3 products, fp32
// accumulation) on v_mfma_f32_32x32x16_bf16 -- 216 MFMAs of 32 cycles per 32-sample tile instead of 576 of 64.
// Backward-data: 8 waves x 32 samples per 256-sample tile, same lane decomposition as the forward kernels
// (lane = sample l&31, half h = l>>5 holds hidden units unit_of(q, h)).  d h^T = W^T dz^T has exactly the forward's
// shape with the transposed weights as the A operand, so the lane that holds h[unit] receives d h[unit].
__device__ __forceinline__ void
mlp_bwd_bf16_body(const float* __restrict__ packed_bwd, const float* __restrict__ feat, int fstride, const float* __restrict__ out,
          const float* __restrict__ g_out, const float* __restrict__ h1, const float* __restrict__ h2, int64_t n,
          int out_dim, int act, float* __restrict__ g_feat, float* __restrict__ dz1o, float* __restrict__ dz2o,
          float* __restrict__ dz3o, const int bid, const int nblk) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    // LDS: W2 (fp32, 512 floats) | bf16 image (W1^T hi, lo, W0^T hi, lo)
    for (int i = threadIdx.x * 4; i < 2 * 64 * 4; i += 512 * 4)
        *reinterpret_cast<float4*>(lds + i) = *reinterpret_cast<const float4*>(packed_bwd + OFFB_W2 + i);
    for (int i = threadIdx.x * 4; i < BWD_BF_FLOATS; i += 512 * 4)
        *reinterpret_cast<float4*>(lds + 2 * 64 * 4 + i) = *reinterpret_cast<const float4*>(packed_bwd + BWD_FLOATS + i);
    __syncthreads();
    const bf16x8* w1hi = reinterpret_cast<const bf16x8*>(lds + 2 * 64 * 4);
    const bf16x8* w1lo = w1hi + BB_W1T_ELEMS / 8;
    const bf16x8* w0hi = w1lo + BB_W1T_ELEMS / 8;
    const bf16x8* w0lo = w0hi + BB_W0T_ELEMS / 8;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int sl = lane & 31, h = lane >> 5;
    const int64_t n_tiles = (n + 255) / 256;
    for (int64_t tile = bid; tile < n_tiles; tile += nblk) {
        const int64_t s_raw = tile * 256 + wave * 32 + sl;
        const bool on = s_raw < n;
        const int64_t s = on ? s_raw : n - 1;
        // ---- output layer ----
        float dz3[4];
#pragma unroll
        for (int o = 0; o < 4; ++o) {
            float y = 0.f, gy = 0.f;
            if (o < out_dim) { y = out[s * out_dim + o]; gy = on ? g_out[s * out_dim + o] : 0.f; }
            dz3[o] = gy * (act == 1 ? (1.0f - y * y) : y * (1.0f - y));
        }
        if (on && h == 0) *reinterpret_cast<float4*>(dz3o + s * 4) = make_float4(dz3[0], dz3[1], dz3[2], dz3[3]);
        float dz[64];
        {
            const float* wp = lds + OFFB_W2 + h * 256;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt)
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float4 hv = *reinterpret_cast<const float4*>(h2 + s * HID + mt * 32 + 8 * i + 4 * h);
                    const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
                    float o4[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int q = mt * 16 + 4 * i + j;
                        const float4 w = *reinterpret_cast<const float4*>(wp + q * 4);
                        const float d = w.x * dz3[0] + w.y * dz3[1] + w.z * dz3[2] + w.w * dz3[3];
                        o4[j] = hvv[j] > 0.f ? d : 0.f;
                        dz[q] = o4[j];
                    }
                    if (on) *reinterpret_cast<float4*>(dz2o + s * HID + mt * 32 + 8 * i + 4 * h) = make_float4(o4[0], o4[1], o4[2], o4[3]);
                }
        }
        // ---- d h1^T = W1^T dz2^T ----
        f32x16 acc[4];
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mt][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            bf16x8 ah[4], al[4], xh, xl;
            const int wi = (kb * 2 + h) * 128 + sl;
#pragma unroll
            for (int mt = 0; mt < 4; ++mt) { ah[mt] = w1hi[wi + mt * 32]; al[mt] = w1lo[wi + mt * 32]; }
            split8(dz + kb * 8, xh, xl);
            mfma12<3>(ah, al, xh, xl, acc);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int mt = 0; mt < 4; ++mt)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4 hv = *reinterpret_cast<const float4*>(h1 + s * HID + mt * 32 + 8 * i + 4 * h);
                const float hvv[4] = {hv.x, hv.y, hv.z, hv.w};
                float o4[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o4[j] = hvv[j] > 0.f ? acc[mt][4 * i + j] : 0.f;
                    dz[mt * 16 + 4 * i + j] = o4[j];
                }
                if (on) *reinterpret_cast<float4*>(dz1o + s * HID + mt * 32 + 8 * i + 4 * h) = make_float4(o4[0], o4[1], o4[2], o4[3]);
            }
        // ---- d x^T = W0^T dz1^T (rows permuted, see bwd_inrow) ----
        f32x16 ax[5];
#pragma unroll
        for (int mt = 0; mt < 5; ++mt)
#pragma unroll
            for (int r = 0; r < 16; ++r) ax[mt][r] = 0.f;
#pragma unroll
        for (int kb = 0; kb < 8; ++kb) {
            bf16x8 xh, xl;
            const int wi = (kb * 2 + h) * 160 + sl;
            split8(dz + kb * 8, xh, xl);
            // two groups of row tiles (3 + 2) keep the operand registers at 6 vectors; product-major inside a group:
            // never two consecutive MFMAs on one accumulator
#pragma unroll
            for (int g0 = 0; g0 < 5; g0 += 3) {
                const int gn = (g0 == 0) ? 3 : 2;
                bf16x8 ah[3], al[3];
#pragma unroll
                for (int u = 0; u < 3; ++u)
                    if (u < gn) { ah[u] = w0hi[wi + (g0 + u) * 32]; al[u] = w0lo[wi + (g0 + u) * 32]; }
#pragma unroll
                for (int u = 0; u < 3; ++u) if (u < gn) ax[g0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], xh, ax[g0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 3; ++u) if (u < gn) ax[g0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], xh, ax[g0 + u], 0, 0, 0);
#pragma unroll
                for (int u = 0; u < 3; ++u) if (u < gn) ax[g0 + u] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], xl, ax[g0 + u], 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- positional-encoding chain rule: x = [f, sin f, sin 2f, cos f, cos 2f] ----
        float gf[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int d = h * 16 + j;
            float v = 0.f;
            if (d < F) {
                const float xv = feat[s * fstride + d];
                float s1, c1;
                fast_sincos(xv, s1, c1);
                const float s2 = 2.0f * s1 * c1, c2 = fmaf(-2.0f * s1, s1, 1.0f);      // double-angle identities
                const int u0 = 5 * j;
                const float g_raw = ax[(u0) >> 4][(u0) & 15], g_s0 = ax[(u0 + 1) >> 4][(u0 + 1) & 15];
                const float g_s1 = ax[(u0 + 2) >> 4][(u0 + 2) & 15], g_c0 = ax[(u0 + 3) >> 4][(u0 + 3) & 15];
                const float g_c1 = ax[(u0 + 4) >> 4][(u0 + 4) & 15];
                v = g_raw + c1 * g_s0 + 2.0f * c2 * g_s1 - s1 * g_c0 - 2.0f * s2 * g_c1;
            }
            gf[j] = v;
        }
        if (on) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<float4*>(g_feat + s * 32 + h * 16 + 4 * i) = make_float4(gf[4 * i], gf[4 * i + 1], gf[4 * i + 2], gf[4 * i + 3]);
        }
    }
}

__global__ void __launch_bounds__(512)
k_mlp_bwd_bf16(const float* __restrict__ packed_bwd, const float* __restrict__ feat, int fstride, const float* __restrict__ out,
          const float* __restrict__ g_out, const float* __restrict__ h1, const float* __restrict__ h2, int64_t n,
          int out_dim, int act, float* __restrict__ g_feat, float* __restrict__ dz1o, float* __restrict__ dz2o,
          float* __restrict__ dz3o) {    mlp_bwd_bf16_body(packed_bwd, feat, fstride, out, g_out, h1, h2, n, out_dim, act, g_feat, dz1o, dz2o, dz3o, (int)blockIdx.x,
                      (int)gridDim.x);
}

// Backward-data of several decoder invocations over the same number of rows in ONE launch (the primary stage: rgb, brdf,
// jittered brdf, normal): the grid is split evenly, a workgroup loads ITS job's operand image once and walks that job's
// tiles -- one 150 KB LDS fill and one tail per workgroup instead of four short launches (see k_mlp_bf16_multi).
struct TirMlpBwdJob { const float* packed_bwd; const float* feat; const float* out; const float* g_out; const float* h1; const float* h2;
                      float* g_feat; float* dz1; float* dz2; float* dz3; int out_dim, act; };
struct TirMlpBwdJobs { TirMlpBwdJob j[4]; int n_jobs; };

__global__ void __launch_bounds__(512)
k_mlp_bwd_bf16_multi(TirMlpBwdJobs jobs, int fstride, int64_t n) {
    const int per = (int)gridDim.x / jobs.n_jobs;
    const int ji = (int)blockIdx.x / per;
    if (ji >= jobs.n_jobs) return;
    const TirMlpBwdJob& jb = jobs.j[ji];
    mlp_bwd_bf16_body(jb.packed_bwd, jb.feat, fstride, jb.out, jb.g_out, jb.h1, jb.h2, n, jb.out_dim, jb.act, jb.g_feat, jb.dz1, jb.dz2,
                      jb.dz3, (int)blockIdx.x - ji * per, per);
}

// ------------------------------------------------------------------------------------------------
// Weight gradients of up to four decoder invocations in ONE pass over their rows (the leaves of the training backward):
//     dW0 += dz1^T X  [128 x 150]     dW1 += dz2^T H1  [128 x 128]     dW2 += dz3^T H2  [4 x 128]     db_l += 1^T dz_l
// One launch replaces, per invocation, tir_mlp_inputs (the 160-wide input rows written to HBM and read back: 0.6 GB per
// step) and three tir_gemm_tn launches: X is rebuilt in registers from the 27 features + 3 aux values the forward kernel
// reads, and every operand row is fetched once per workgroup straight into the matrix-core layout -- a lane of
// v_mfma_f32_32x32x16_bf16 supplies 8 CONSECUTIVE k (= rows) of ONE column, and for a fixed row the 32 lanes of a half-wave
// read 32 consecutive columns (one 128-B line): the "transposed" operand of a TN product is a plain coalesced load, no LDS
// staging and no workgroup barrier anywhere.  Split-bf16 operands (hi + lo, three products, fp32 accumulation) as in the
// forward.  16 waves per workgroup: wave (mt = w & 3, g = w >> 2) owns rows mt*32.. of the products' left operands and
//     g = 0: dW0 column tiles 0,1,2          (A = dz1; also db0)        g = 2: dW1 column tiles 0,1  (A = dz2; also db1)
//     g = 1: dW0 column tiles 3,4 + dW2 tile mt (A = dz1, dz3; db2)     g = 3: dW1 column tiles 2,3  (A = dz2)
// so the four waves of a SIMD (same mt) carry the heavy (X-building) and the light (H1-loading) tiles together.
// A workgroup walks a contiguous chunk of its job's rows, 16 rows per step, and adds its partial sums atomically at the end.
// ------------------------------------------------------------------------------------------------
struct TirWgradJob { const float *dz1, *dz2, *dz3, *h1, *h2, *feat, *aux; const int32_t* aux_map;
                     float *dW0, *db0, *dW1, *db1, *dW2, *db2; };
struct TirWgradJobs { TirWgradJob j[4]; int n_jobs; };

struct WgXCol { int off; float mul, phase; bool is_aux, is_pe, zero; };

// decoder input column c (reference order, models/tensorBase_rotated_lights.py:137-142 / :199-204, :12-17) -> how to build it
__device__ __forceinline__ WgXCol wg_xcol(int c) {
    WgXCol x{0, 1.0f, 0.0f, false, false, true};
    if (c < F) { x.off = c; x.zero = false; }
    else if (c < F + 3) { x.off = c - F; x.is_aux = true; x.zero = false; }
    else if (c < F + 3 + 2 * NPF) {
        int q = c - (F + 3);
        const bool cs = q >= NPF;
        if (cs) q -= NPF;
        x.off = q / PE; x.mul = (float)(1 << (q % PE)); x.phase = cs ? 0.25f : 0.0f; x.is_pe = true; x.zero = false;
    } else if (c < IN) {
        int q = c - (F + 3 + 2 * NPF);
        const bool cs = q >= 3 * PE;
        if (cs) q -= 3 * PE;
        x.off = q / PE; x.mul = (float)(1 << (q % PE)); x.phase = cs ? 0.25f : 0.0f; x.is_pe = true; x.is_aux = true; x.zero = false;
    }
    return x;
}

// 8 consecutive rows s0.. of column `col` of a row-major [.][ld] matrix (rows >= s0 + nv read as 0)
template <bool FULL>
__device__ __forceinline__ void wg_col8(const float* __restrict__ p, int ld, int col, int64_t s0, int nv, float (&v)[8]) {
    const float* b = p + s0 * ld + col;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = (FULL || j < nv) ? b[(int64_t)j * ld] : 0.0f;
}

// the same for decoder-input column xc of X (never materialised)
template <bool FULL>
__device__ __forceinline__ void wg_x8(const WgXCol& xc, const float* __restrict__ feat, int fstride, const float* __restrict__ aux,
                                      const int64_t (&ai)[8], int64_t s0, int nv, float (&v)[8]) {
    const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;      // c_hi + c_lo = 1 / (2 pi), see pe_pair
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float r = 0.0f;
        if (FULL || j < nv) {
            const float* src = xc.is_aux ? aux + 3 * ai[j] + xc.off : feat + (s0 + j) * fstride + xc.off;
            const float b = *src;
            const float k = rintf(b * c_hi);
            float t = fmaf(b, c_hi, -k);
            t = fmaf(b, c_lo, t);
            const float pe = __builtin_amdgcn_sinf(fmaf(t, xc.mul, xc.phase));
            r = xc.zero ? 0.0f : (xc.is_pe ? pe : b);
        }
        v[j] = r;
    }
}

__device__ __forceinline__ void wg_atomic(float* p, float v) { unsafeAtomicAdd(p, v); }

// acc tile (rows mt*32.., columns nt*32..) -> C[row][col] += acc, row < M (the D layout of the 32x32 MFMA)
__device__ __forceinline__ void wg_flush(const f32x16& acc, float* __restrict__ C, int ldc, int row0, int M, int col, int N) {
    if (col >= N) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = row0 + (r & 3) + 8 * (r >> 2);
        if (row < M) wg_atomic(C + (int64_t)row * ldc + col, acc[r]);
    }
}

template <int NT>
__device__ __forceinline__ void wg_mma(const bf16x8& ah, const bf16x8& al, const bf16x8 (&bh)[NT], const bf16x8 (&bl)[NT],
                                       f32x16 (&acc)[NT]) {
    // product-major: never two consecutive MFMAs on one accumulator
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
}

// one 16-row step of wave group G (see the table above)
template <int G, bool FULL>
__device__ __forceinline__ void wg_step(const TirWgradJob& jb, int fstride, int mt, int li, int h, int64_t s0, int nv,
                                        const WgXCol (&xc)[3], f32x16 (&acc)[3], float& bias) {
    float v[8];
    bf16x8 ah, al;
    if (G <= 1) {                                           // left operand dz1, right operand X
        wg_col8<FULL>(jb.dz1, HID, mt * 32 + li, s0, nv, v);
        if (G == 0) bias += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        split8(v, ah, al);
        int64_t ai[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int64_t s = s0 + ((FULL || j < nv) ? j : 0);
            ai[j] = jb.aux_map ? (int64_t)jb.aux_map[s] : s;
        }
        constexpr int NX = (G == 0) ? 3 : 2;
        bf16x8 bh[NX], bl[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            float x[8];
            wg_x8<FULL>(xc[t], jb.feat, fstride, jb.aux, ai, s0, nv, x);
            split8(x, bh[t], bl[t]);
        }
        f32x16 (&a)[NX] = reinterpret_cast<f32x16 (&)[NX]>(acc);
        wg_mma<NX>(ah, al, bh, bl, a);
        if (G == 1) {                                       // dW2 tile: left operand dz3 (4 columns), right operand H2 tile mt
            float z[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = (li < 4 && (FULL || j < nv)) ? jb.dz3[(s0 + j) * 4 + li] : 0.0f;
            bias += ((z[0] + z[1]) + (z[2] + z[3])) + ((z[4] + z[5]) + (z[6] + z[7]));
            wg_col8<FULL>(jb.h2, HID, mt * 32 + li, s0, nv, y);
            bf16x8 zh, zl, yh[1], yl[1];
            split8(z, zh, zl);
            split8(y, yh[0], yl[0]);
            f32x16 (&a2)[1] = reinterpret_cast<f32x16 (&)[1]>(acc[2]);
            wg_mma<1>(zh, zl, yh, yl, a2);
        }
    } else {                                                // left operand dz2, right operand H1 tiles 2(G-2), 2(G-2)+1
        wg_col8<FULL>(jb.dz2, HID, mt * 32 + li, s0, nv, v);
        if (G == 2) bias += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        split8(v, ah, al);
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[8];
            wg_col8<FULL>(jb.h1, HID, (2 * (G - 2) + t) * 32 + li, s0, nv, y);
            split8(y, bh[t], bl[t]);
        }
        f32x16 (&a)[2] = reinterpret_cast<f32x16 (&)[2]>(acc);
        wg_mma<2>(ah, al, bh, bl, a);
    }
}

template <int G>
__device__ __forceinline__ void wg_path(const TirWgradJob& jb, int fstride, int mt, int li, int h, int64_t r0, int64_t r1) {
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    WgXCol xc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) xc[t] = wg_xcol((G == 0 ? t : 3 + t) * 32 + li);
    float bias = 0.0f;
    int64_t kb = r0;
    for (; kb + 16 <= r1; kb += 16) wg_step<G, true>(jb, fstride, mt, li, h, kb + 8 * h, 8, xc, acc, bias);
    if (kb < r1) {                                          // ragged tail of the chunk
        const int64_t s0 = kb + 8 * h;
        const int nv = (int)max((int64_t)0, min((int64_t)8, r1 - s0));
        wg_step<G, false>(jb, fstride, mt, li, h, min(s0, r1 - 1), s0 < r1 ? nv : 0, xc, acc, bias);
    }
    const int row0 = mt * 32 + 4 * h;
    if (G == 0) {
#pragma unroll
        for (int t = 0; t < 3; ++t) wg_flush(acc[t], jb.dW0, IN, row0, HID, t * 32 + li, IN);
    } else if (G == 1) {
#pragma unroll
        for (int t = 0; t < 2; ++t) wg_flush(acc[t], jb.dW0, IN, row0, HID, (3 + t) * 32 + li, IN);
        wg_flush(acc[2], jb.dW2, HID, 4 * h, 4, mt * 32 + li, HID);        // rows = the 4 outputs (lanes of half 0, r < 4)
    } else {
#pragma unroll
        for (int t = 0; t < 2; ++t) wg_flush(acc[t], jb.dW1, HID, row0, HID, (2 * (G - 2) + t) * 32 + li, HID);
    }
    if (G != 3) {                                           // column sums of the left operand = the bias gradients
        bias += __shfl_xor(bias, 32, 64);
        if (h == 0) {
            if (G == 0) wg_atomic(jb.db0 + mt * 32 + li, bias);
            if (G == 2) wg_atomic(jb.db1 + mt * 32 + li, bias);
            if (G == 1 && mt == 0 && li < 4) wg_atomic(jb.db2 + li, bias);
        }
    }
}

__global__ void __launch_bounds__(1024)
k_mlp_wgrad(TirWgradJobs jobs, int fstride, int64_t n) {
    const int per = (int)gridDim.x / jobs.n_jobs;
    const int ji = (int)blockIdx.x / per;
    if (ji >= jobs.n_jobs) return;
    const TirWgradJob& jb = jobs.j[ji];
    const int bid = (int)blockIdx.x - ji * per;
    int64_t chunk = (n + per - 1) / per;
    chunk = (chunk + 15) / 16 * 16;
    const int64_t r0 = (int64_t)bid * chunk, r1 = min(n, r0 + chunk);
    if (r0 >= r1) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5, mt = w & 3;
    switch (w >> 2) {
        case 0: wg_path<0>(jb, fstride, mt, li, h, r0, r1); break;
        case 1: wg_path<1>(jb, fstride, mt, li, h, r0, r1); break;
        case 2: wg_path<2>(jb, fstride, mt, li, h, r0, r1); break;
        default: wg_path<3>(jb, fstride, mt, li, h, r0, r1); break;
    }
}

// ---- the same products with the operand rows staged through LDS, double buffered -------------------------------------------
// The direct-load kernel above exposes one memory latency per 16-row step (its 128-register budget has no room to prefetch),
// and every dz row is fetched by two wave groups.  Here half of the workgroup (the waves with registers to spare) fetches the NEXT step's 16 rows of
// dz1, dz2, h1, h2 (coalesced float4 loads), features, dz3 and aux into registers while the waves work on the current
// step out of LDS, then park them in the other buffer: one barrier per step, each row read from L2 / HBM once per workgroup,
// LDS addresses are immediates (no 64-bit address arithmetic in the loop).  Row strides 132 / 36 floats: the two half-waves
// of an operand read (rows r and r + 8) land 32 banks apart.
constexpr int WL_LD = 132, WL_LDF = 36;
// a staged row of `feat` is [0..31] the feature row as stored (27 used), [32..34] the row's aux triple, [35] zero
struct WgStage { float dz1[16 * WL_LD], dz2[16 * WL_LD], h1[16 * WL_LD], h2[16 * WL_LD], feat[16 * WL_LDF], dz3[16 * 4]; };
struct WlXCol { int off; float mul, phase; };        // mul == 0: the raw value at `off`; else sin(2 pi (mul x + phase))

__device__ __forceinline__ WlXCol wl_xcol(int c) {
    const WgXCol x = wg_xcol(c);
    WlXCol r;
    r.off = x.zero ? 35 : (x.is_aux ? 32 + x.off : x.off);
    r.mul = (x.zero || !x.is_pe) ? 0.0f : x.mul;
    r.phase = x.phase;
    return r;
}
constexpr int WL_XCOLS = 160;                          // X column descriptors (3 floats each) live behind the two stages
constexpr int WL_LDS_BYTES = 2 * (int)sizeof(WgStage) + WL_XCOLS * 3 * (int)sizeof(float);

struct WgFetch { float4 z1, z2, a1, a2; };

// The fetching is done by the waves with registers to spare while they compute: the 512 threads of wave groups 2 and 3 (two
// accumulator tiles, no X columns) bring in the four [16][128] operands, u = threadIdx.x - 512; the 256 threads of group 1
// (two X tiles + the small dW2 tile) the feature rows, dz3 and the aux triples, u1 = threadIdx.x - 256.  Rows >= r1 read as 0.
// Uniform 64-bit row bases + one 32-bit lane offset: scalar-base addressing, no per-lane 64-bit arithmetic.
__device__ __forceinline__ WgFetch wl_fetch(const TirWgradJob& jb, int64_t s0, int64_t r1, int u) {
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    WgFetch f{zero, zero, zero, zero};
    const unsigned row = (unsigned)u >> 5, o = row * HID + 4u * ((unsigned)u & 31u);
    if ((int64_t)row < r1 - s0) {
        f.z1 = *reinterpret_cast<const float4*>(jb.dz1 + s0 * HID + o);
        f.z2 = *reinterpret_cast<const float4*>(jb.dz2 + s0 * HID + o);
        f.a1 = *reinterpret_cast<const float4*>(jb.h1 + s0 * HID + o);
        f.a2 = *reinterpret_cast<const float4*>(jb.h2 + s0 * HID + o);
    }
    return f;
}

__device__ __forceinline__ void wl_park(WgStage& st, const WgFetch& f, int u) {
    const int o = (u >> 5) * WL_LD + 4 * (u & 31);
    *reinterpret_cast<float4*>(st.dz1 + o) = f.z1;
    *reinterpret_cast<float4*>(st.dz2 + o) = f.z2;
    *reinterpret_cast<float4*>(st.h1 + o) = f.a1;
    *reinterpret_cast<float4*>(st.h2 + o) = f.a2;
}

// ai = the aux row of an aux-role thread (looked up one step earlier)
__device__ __forceinline__ float4 wl_fetch_side(const TirWgradJob& jb, int fstride, int64_t s0, int64_t r1, int ai, int u1) {
    float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
    const int left = (int)min((int64_t)16, r1 - s0);
    if (u1 < 128) {
        const unsigned fr = (unsigned)u1 >> 3;
        if ((int)fr < left) c = *reinterpret_cast<const float4*>(jb.feat + s0 * fstride + (fr * (unsigned)fstride + 4u * ((unsigned)u1 & 7u)));
    } else if (u1 < 144) {
        if (u1 - 128 < left) c = *reinterpret_cast<const float4*>(jb.dz3 + s0 * 4 + 4u * (unsigned)(u1 - 128));
    } else if (u1 < 192) {
        const int q = u1 - 144;
        if (q / 3 < left) c.x = jb.aux[3 * (int64_t)ai + q % 3];
    }
    return c;
}

__device__ __forceinline__ void wl_park_side(WgStage& st, const float4& c, int u1) {
    if (u1 < 128) *reinterpret_cast<float4*>(st.feat + (u1 >> 3) * WL_LDF + 4 * (u1 & 7)) = c;
    else if (u1 < 144) *reinterpret_cast<float4*>(st.dz3 + (u1 - 128) * 4) = c;
    else if (u1 < 192) st.feat[((u1 - 144) / 3) * WL_LDF + 32 + (u1 - 144) % 3] = c.x;
}

// rows 8h .. 8h+7 of column col of a staged [16][WL_LD] operand
__device__ __forceinline__ void wl_col8(const float* __restrict__ m, int col, int h, float (&v)[8]) {
    const float* b = m + 8 * h * WL_LD + col;
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = b[j * WL_LD];
}

// column descriptor c is re-read from the LDS table every step: three registers per column for the duration of its eight
// values instead of nine held across the loop (the 3-tile wave groups sit exactly at the 128-register budget)
__device__ __forceinline__ void wl_x8(const float* __restrict__ xtab, int c, const WgStage& st, int h, float (&v)[8]) {
    const float c_hi = 0.15915493667125702f, c_lo = 6.4206382432985265e-09f;      // c_hi + c_lo = 1 / (2 pi), see pe_pair
    const int off = __float_as_int(xtab[3 * c]);
    const float mul = xtab[3 * c + 1], phase = xtab[3 * c + 2];
    const float* src = st.feat + 8 * h * WL_LDF + off;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float b = src[j * WL_LDF];
        const float k = rintf(b * c_hi);
        float t = fmaf(b, c_hi, -k);
        t = fmaf(b, c_lo, t);
        const float pe = __builtin_amdgcn_sinf(fmaf(t, mul, phase));
        v[j] = mul == 0.0f ? b : pe;
    }
}

template <int G>
__device__ __forceinline__ void wl_step(const WgStage& st, const float* __restrict__ xtab, int mt, int li, int h, f32x16 (&acc)[3], float& bias) {
    float v[8];
    bf16x8 ah, al;
    if (G <= 1) {
        wl_col8(st.dz1, mt * 32 + li, h, v);
        if (G == 0) bias += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        split8(v, ah, al);
        constexpr int NX = (G == 0) ? 3 : 2;
        bf16x8 bh[NX], bl[NX];
#pragma unroll
        for (int t = 0; t < NX; ++t) {
            float x[8];
            wl_x8(xtab, (G == 0 ? t : 3 + t) * 32 + li, st, h, x);
            split8(x, bh[t], bl[t]);
        }
        f32x16 (&a)[NX] = reinterpret_cast<f32x16 (&)[NX]>(acc);
        wg_mma<NX>(ah, al, bh, bl, a);
        if (G == 1) {
            float z[8], y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) z[j] = li < 4 ? st.dz3[(8 * h + j) * 4 + li] : 0.0f;
            bias += ((z[0] + z[1]) + (z[2] + z[3])) + ((z[4] + z[5]) + (z[6] + z[7]));
            wl_col8(st.h2, mt * 32 + li, h, y);
            bf16x8 zh, zl, yh[1], yl[1];
            split8(z, zh, zl);
            split8(y, yh[0], yl[0]);
            f32x16 (&a2)[1] = reinterpret_cast<f32x16 (&)[1]>(acc[2]);
            wg_mma<1>(zh, zl, yh, yl, a2);
        }
    } else {
        wl_col8(st.dz2, mt * 32 + li, h, v);
        if (G == 2) bias += ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
        split8(v, ah, al);
        bf16x8 bh[2], bl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float y[8];
            wl_col8(st.h1, (2 * (G - 2) + t) * 32 + li, h, y);
            split8(y, bh[t], bl[t]);
        }
        f32x16 (&a)[2] = reinterpret_cast<f32x16 (&)[2]>(acc);
        wg_mma<2>(ah, al, bh, bl, a);
    }
}

template <int G>
__device__ __forceinline__ void wl_path(const TirWgradJob& jb, int fstride, WgStage* stage, int mt, int li, int h, int64_t r0, int64_t r1) {
    f32x16 acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;
    float* xtab = reinterpret_cast<float*>(stage + 2);
    float bias = 0.0f;
    if (threadIdx.x < 32) stage[threadIdx.x >> 4].feat[(threadIdx.x & 15) * WL_LDF + 35] = 0.0f;      // the zero column of both buffers
    if (threadIdx.x >= 64 && threadIdx.x < 64 + WL_XCOLS) {
        const WlXCol x = wl_xcol((int)threadIdx.x - 64);
        float* e = xtab + 3 * ((int)threadIdx.x - 64);
        e[0] = __int_as_float(x.off); e[1] = x.mul; e[2] = x.phase;
    }
    // aux rows are looked up through aux_map: the index of step k + 2 is fetched while step k + 1's data is, so the dependent
    // load never sits inside one step's window
    constexpr bool FETCH = G >= 2, SIDE = G == 1;
    const int u = (int)threadIdx.x - 512, u1 = (int)threadIdx.x - 256;
    const bool aux_role = SIDE && u1 >= 144 && u1 < 192;
    const int arow = aux_role ? (u1 - 144) / 3 : 0;
    auto aux_index = [&](int64_t s0) -> int {              // rows < 2^31 (checked by the launcher)
        const int64_t s = s0 + arow;
        if (!aux_role || s >= r1) return 0;
        return jb.aux_map ? jb.aux_map[s] : (int)s;
    };
    int ai = 0, ai_next = 0;
    WgFetch f;
    float4 side;
    if (FETCH) {
        f = wl_fetch(jb, r0, r1, u);
        wl_park(stage[0], f, u);
    }
    if (SIDE) {
        ai = aux_index(r0); ai_next = aux_index(r0 + 16);
        side = wl_fetch_side(jb, fstride, r0, r1, ai, u1);
        wl_park_side(stage[0], side, u1);
    }
    __syncthreads();
    int cur = 0;
    for (int64_t kb = r0; kb < r1; kb += 16) {
        const bool more = kb + 16 < r1;
        if (FETCH && more) f = wl_fetch(jb, kb + 16, r1, u);
        if (SIDE && more) {
            ai = ai_next;
            ai_next = aux_index(kb + 32);
            side = wl_fetch_side(jb, fstride, kb + 16, r1, ai, u1);
        }
        wl_step<G>(stage[cur], xtab, mt, li, h, acc, bias);
        if (FETCH && more) wl_park(stage[cur ^ 1], f, u);
        if (SIDE && more) wl_park_side(stage[cur ^ 1], side, u1);
        __syncthreads();
        cur ^= 1;
    }
    const int row0 = mt * 32 + 4 * h;
    if (G == 0) {
#pragma unroll
        for (int t2 = 0; t2 < 3; ++t2) wg_flush(acc[t2], jb.dW0, IN, row0, HID, t2 * 32 + li, IN);
    } else if (G == 1) {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) wg_flush(acc[t2], jb.dW0, IN, row0, HID, (3 + t2) * 32 + li, IN);
        wg_flush(acc[2], jb.dW2, HID, 4 * h, 4, mt * 32 + li, HID);
    } else {
#pragma unroll
        for (int t2 = 0; t2 < 2; ++t2) wg_flush(acc[t2], jb.dW1, HID, row0, HID, (2 * (G - 2) + t2) * 32 + li, HID);
    }
    if (G != 3) {
        bias += __shfl_xor(bias, 32, 64);
        if (h == 0) {
            if (G == 0) wg_atomic(jb.db0 + mt * 32 + li, bias);
            if (G == 2) wg_atomic(jb.db1 + mt * 32 + li, bias);
            if (G == 1 && mt == 0 && li < 4) wg_atomic(jb.db2 + li, bias);
        }
    }
}

__global__ void __launch_bounds__(1024)
k_mlp_wgrad_lds(TirWgradJobs jobs, int fstride, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wl_lds[];
    WgStage* stage = reinterpret_cast<WgStage*>(wl_lds);
    const int per = (int)gridDim.x / jobs.n_jobs;
    const int ji = (int)blockIdx.x / per;
    if (ji >= jobs.n_jobs) return;
    const TirWgradJob& jb = jobs.j[ji];
    const int bid = (int)blockIdx.x - ji * per;
    int64_t chunk = (n + per - 1) / per;
    chunk = (chunk + 15) / 16 * 16;
    const int64_t r0 = (int64_t)bid * chunk, r1 = min(n, r0 + chunk);
    if (r0 >= r1) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5, mt = w & 3;
    switch (w >> 2) {
        case 0: wl_path<0>(jb, fstride, stage, mt, li, h, r0, r1); break;
        case 1: wl_path<1>(jb, fstride, stage, mt, li, h, r0, r1); break;
        case 2: wl_path<2>(jb, fstride, stage, mt, li, h, r0, r1); break;
        default: wl_path<3>(jb, fstride, stage, mt, li, h, r0, r1); break;
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

extern "C" int tir_mlp_fwd(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                           float* out, int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !out))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const size_t lds = (size_t)MFMA_FLOATS * sizeof(float);
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_mfma<false>), (int)lds)) return rc;
    int64_t tiles = (n + 255) / 256;
    unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(k_mlp_mfma<false>, dim3(grid), dim3(512), lds, tir_stream(stream), m->packed, feat, feat_stride, aux,
                       aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act, (float*)nullptr, (float*)nullptr);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_train_fwd(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                                 const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2,
                                 int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !out || !h1 || !h2))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const size_t lds = (size_t)MFMA_FLOATS * sizeof(float);
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_mfma<true>), (int)lds)) return rc;
    int64_t tiles = (n + 255) / 256;
    unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(k_mlp_mfma<true>, dim3(grid), dim3(512), lds, tir_stream(stream), m->packed, feat, feat_stride, aux,
                       aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act, h1, h2);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

template <int NPROD, bool VEC, bool SAVE>
static int launch_bf16_v(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                         float* out, int64_t n, const int32_t* n_dev, void* stream, float* h1 = nullptr, float* h2 = nullptr) {
    const size_t lds = (size_t)BF_BYTES;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bf16<NPROD, VEC, SAVE>), (int)lds)) return rc;
    int64_t tiles = (n + 255) / 256;
    const int grid_max = m->tune_grid > 0 ? m->tune_grid : 256;           // launch option carried by the descriptor
    unsigned grid = (unsigned)(tiles < grid_max ? tiles : grid_max);
    hipLaunchKernelGGL((k_mlp_bf16<NPROD, VEC, SAVE>), dim3(grid), dim3(512), lds, tir_stream(stream), m->packed, feat, feat_stride, aux,
                       aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act, h1, h2);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

template <int NPROD>
static int launch_bf16(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                       float* out, int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !out))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    // 16-byte aligned rows of >= 28 floats take the dwordx4 row loads
    const bool vec = (feat_stride % 4 == 0) && feat_stride >= F + 1 && (reinterpret_cast<uintptr_t>(feat) % 16 == 0);
    return vec ? launch_bf16_v<NPROD, true, false>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream)
               : launch_bf16_v<NPROD, false, false>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream);
}

extern "C" int tir_mlp_train_fwd_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                                        const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2,
                                        int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !out || !h1 || !h2))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const bool vec = (feat_stride % 4 == 0) && feat_stride >= F + 1 && (reinterpret_cast<uintptr_t>(feat) % 16 == 0);
    return vec ? launch_bf16_v<3, true, true>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream, h1, h2)
               : launch_bf16_v<3, false, true>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream, h1, h2);
}

extern "C" int tir_mlp_fwd_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                                  float* out, int64_t n, const int32_t* n_dev, void* stream) {
    return launch_bf16<3>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream);
}

template <bool SAVE>
static int launch_multi(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride, const float* const* auxs,
                        const int32_t* const* aux_maps, float* const* outs, float* const* h1s, float* const* h2s,
                        int32_t n_jobs, int64_t n, const int32_t* n_dev, void* stream, const float* const* tables = nullptr) {
    if (n_jobs < 1 || n_jobs > 4 || !mlps || !feats || !auxs || !outs || n < 0) return TIR_ERR_ARG;
    if (SAVE && (!h1s || !h2s)) return TIR_ERR_ARG;
    if (feat_stride % 4 != 0 || feat_stride < F + 1) return TIR_ERR_ARG;          // rows must take the dwordx4 loads
    TirMlpJobs jobs;
    jobs.n_jobs = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        int rc = check_mlp(mlps[i]);
        if (rc) return rc;
        const float* tab = (!SAVE && tables) ? tables[i] : nullptr;
        if (n > 0 && (!feats[i] || (!auxs[i] && !tab) || !outs[i])) return TIR_ERR_ARG;
        if (SAVE && n > 0 && (!h1s[i] || !h2s[i])) return TIR_ERR_ARG;
        if (reinterpret_cast<uintptr_t>(feats[i]) % 16 != 0 || reinterpret_cast<uintptr_t>(tab) % 16 != 0) return TIR_ERR_ARG;
        jobs.j[i] = TirMlpJob{mlps[i]->packed, feats[i], auxs[i], aux_maps ? aux_maps[i] : nullptr, outs[i], mlps[i]->out_dim,
                              mlps[i]->act, SAVE ? h1s[i] : nullptr, SAVE ? h2s[i] : nullptr, tab};
    }
    if (n == 0) return TIR_OK;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bf16_multi<3, SAVE>), (int)BF_BYTES)) return rc;
    const int64_t tiles = (n + 255) / 256;
    int per = 256 / n_jobs;
    if (tiles < per) per = (int)tiles;
    hipLaunchKernelGGL((k_mlp_bf16_multi<3, SAVE>), dim3((unsigned)(per * n_jobs)), dim3(512), (size_t)BF_BYTES, tir_stream(stream),
                       jobs, feat_stride, n, n_dev);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_fwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                                        const float* const* auxs, const int32_t* const* aux_maps, float* const* outs,
                                        int32_t n_jobs, int64_t n, const int32_t* n_dev, void* stream) {
    return launch_multi<false>(mlps, feats, feat_stride, auxs, aux_maps, outs, nullptr, nullptr, n_jobs, n, n_dev, stream);
}

extern "C" int tir_mlp_fwd_multi_auxtab_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                                               const float* const* auxs, const int32_t* const* aux_maps,
                                               const float* const* tables, float* const* outs, int32_t n_jobs, int64_t n,
                                               const int32_t* n_dev, void* stream) {
    return launch_multi<false>(mlps, feats, feat_stride, auxs, aux_maps, outs, nullptr, nullptr, n_jobs, n, n_dev, stream, tables);
}

extern "C" int tir_mlp_aux_table(const TirMlp* m, const float* aux, int64_t n_aux, float* table, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n_aux < 0 || (n_aux > 0 && (!aux || !table))) return TIR_ERR_ARG;
    if (n_aux == 0) return TIR_OK;
    const int64_t total = n_aux * HID;
    hipLaunchKernelGGL(k_mlp_aux_table, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, tir_stream(stream), m->packed, aux, n_aux,
                       table);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

template <bool SAVE>
static int launch_auxtab(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table, const int32_t* aux_map,
                         int32_t aux_mod, float* out, float* h1, float* h2, int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !table || !out || (SAVE && (!h1 || !h2))))) return TIR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(table) % 16 != 0) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const bool vec = (feat_stride % 4 == 0) && feat_stride >= F + 1 && (reinterpret_cast<uintptr_t>(feat) % 16 == 0);
    const void* kfn = vec ? reinterpret_cast<const void*>(k_mlp_bf16_auxt<true, SAVE>) : reinterpret_cast<const void*>(k_mlp_bf16_auxt<false, SAVE>);
    if (int r2 = tir_allow_dynamic_lds(kfn, (int)BFA_BYTES)) return r2;
    const int64_t tiles = (n + 255) / 256;
    const int grid_max = m->tune_grid > 0 ? m->tune_grid : 256;
    const unsigned grid = (unsigned)(tiles < grid_max ? tiles : grid_max);
    if (vec) hipLaunchKernelGGL((k_mlp_bf16_auxt<true, SAVE>), dim3(grid), dim3(512), (size_t)BFA_BYTES, tir_stream(stream), m->packed,
                                feat, feat_stride, table, aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act, h1, h2);
    else     hipLaunchKernelGGL((k_mlp_bf16_auxt<false, SAVE>), dim3(grid), dim3(512), (size_t)BFA_BYTES, tir_stream(stream), m->packed,
                                feat, feat_stride, table, aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act, h1, h2);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_fwd_auxtab_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                                         const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
                                         void* stream) {
    return launch_auxtab<false>(m, feat, feat_stride, table, aux_map, aux_mod, out, nullptr, nullptr, n, n_dev, stream);
}

extern "C" int tir_mlp_fwd_auxtab_f16(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                                      const int32_t* aux_map, int32_t aux_mod, float* out, int64_t n, const int32_t* n_dev,
                                      void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !table || !out))) return TIR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(table) % 16 != 0) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const bool vec = (feat_stride % 4 == 0) && feat_stride >= F + 1 && (reinterpret_cast<uintptr_t>(feat) % 16 == 0);
    const void* kfn = vec ? reinterpret_cast<const void*>(k_mlp_f16_auxt<true>) : reinterpret_cast<const void*>(k_mlp_f16_auxt<false>);
    if (int r2 = tir_allow_dynamic_lds(kfn, (int)FH_BYTES)) return r2;
    const int64_t tiles = (n + 255) / 256;
    const int grid_max = m->tune_grid > 0 ? m->tune_grid : 256;
    const unsigned grid = (unsigned)(tiles < grid_max ? tiles : grid_max);
    if (vec) hipLaunchKernelGGL((k_mlp_f16_auxt<true>), dim3(grid), dim3(512), (size_t)FH_BYTES, tir_stream(stream), m->packed, feat,
                                feat_stride, table, aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act);
    else     hipLaunchKernelGGL((k_mlp_f16_auxt<false>), dim3(grid), dim3(512), (size_t)FH_BYTES, tir_stream(stream), m->packed, feat,
                                feat_stride, table, aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_indirect_fused_fwd(const TirField* f, const TirFieldHalf* fh, const TirMlp* m, const float* xyz,
                                      const int32_t* light_idx, const int32_t* rec_map, int32_t idx_div, int32_t aux_mod,
                                      const float* table, float* out, int64_t n, const int32_t* n_dev, void* stream) {
    if (!f || !fh) return TIR_ERR_ARG;
    int rc = check_mlp(m);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !fh->aplane[i] || !fh->aline[i] || reinterpret_cast<uintptr_t>(fh->aplane[i]) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(fh->aline[i]) % 16 != 0) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->n_acomp != 48 || f->app_dim != F || !tir_app_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    if (n < 0 || (n > 0 && (!xyz || !light_idx || !table || !out))) return TIR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(table) % 16 != 0) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (n >= (int64_t)1 << 31 || idx_div < 0 || aux_mod < 0) return TIR_ERR_UNSUPPORTED;       // 32-bit record / ray arithmetic in the kernel
    const int lt_rows = f->n_lights <= 16 ? f->n_lights : 0;
    constexpr int NW = 12; constexpr bool PACK = true;
    const size_t lds = (size_t)FH_BYTES + FUS_WH_BYTES + (size_t)(lt_rows + 1) * 144 * sizeof(float) + (size_t)NW * 32 * FUS_XH * 2;
    if (int r2 = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_indirect_fused<NW, PACK>), (int)lds)) return r2;
    const int64_t tiles = (n + NW * 32 - 1) / (NW * 32);
    const int grid_max = m->tune_grid > 0 ? m->tune_grid : 256;
    const unsigned grid = (unsigned)(tiles < grid_max ? tiles : grid_max);
    hipLaunchKernelGGL((k_indirect_fused<NW, PACK>), dim3(grid), dim3(NW * 64), lds, tir_stream(stream), *f, *fh, m->packed, xyz, light_idx, rec_map,
                       idx_div, aux_mod, table, out, n, n_dev, m->out_dim, m->act, lt_rows);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_indirect_fused_hp_fwd(const TirField* f, const TirMlp* m, const float* xyz, const int32_t* light_idx,
                                         const int32_t* rec_map, int32_t idx_div, int32_t aux_mod, const float* table, float* out,
                                         int64_t n, const int32_t* n_dev, void* stream) {
    if (!f) return TIR_ERR_ARG;
    int rc = check_mlp(m);
    if (rc) return rc;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i] || reinterpret_cast<uintptr_t>(f->aplane[i]) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(f->aline[i]) % 16 != 0) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_line) return TIR_ERR_ARG;
    if (f->n_acomp != 48 || f->app_dim != F || !tir_app_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    for (int i = 0; i < 3; ++i)           // 32-bit byte offsets on the 24-bit multiplier: row bytes < 2^24, plane bytes < 2^32
        for (int j = i + 1; j < 3; ++j)
            if ((int64_t)f->grid[i] * HP_TB >= (1 << 24) || (int64_t)f->grid[j] * HP_TB >= (1 << 24) ||
                (int64_t)f->grid[i] * f->grid[j] * HP_TB >= ((int64_t)1 << 32)) return TIR_ERR_UNSUPPORTED;
    if (n < 0 || (n > 0 && (!xyz || !light_idx || !table || !out))) return TIR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(table) % 16 != 0) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (n >= (int64_t)1 << 31 || idx_div < 0 || aux_mod < 0) return TIR_ERR_UNSUPPORTED;       // 32-bit record / ray arithmetic in the kernel
    if (f->n_lights < 1 || f->n_lights > 8) return TIR_ERR_UNSUPPORTED;        // every light row is staged in LDS (576 B each; 8 rows: 158.2 KB in all)
    const int lt_rows = f->n_lights;
    constexpr int NW = 8;
    const size_t lds = (size_t)FH_BYTES + (size_t)F8_FLOATS * 4 + 2 * (size_t)FUS_WH_BYTES + (size_t)lt_rows * 144 * sizeof(float) +
                       (size_t)NW * HP_X_HALVES * 2;
    if (int r2 = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_indirect_fused_hp<NW>), (int)lds)) return r2;
    const int64_t tiles = (n + NW * 32 - 1) / (NW * 32);
    const int grid_max = m->tune_grid > 0 ? m->tune_grid : 256;
    const unsigned grid = (unsigned)(tiles < grid_max ? tiles : grid_max);
    hipLaunchKernelGGL((k_indirect_fused_hp<NW>), dim3(grid), dim3(NW * 64), lds, tir_stream(stream), *f, m->packed, xyz, light_idx, rec_map,
                       idx_div, aux_mod, table, out, n, n_dev, m->out_dim, m->act, lt_rows);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_train_fwd_auxtab_bf16x3(const TirMlp* m, const float* feat, int32_t feat_stride, const float* table,
                                               const int32_t* aux_map, int32_t aux_mod, float* out, float* h1, float* h2,
                                               int64_t n, const int32_t* n_dev, void* stream) {
    return launch_auxtab<true>(m, feat, feat_stride, table, aux_map, aux_mod, out, h1, h2, n, n_dev, stream);
}

extern "C" int tir_mlp_train_fwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* feats, int32_t feat_stride,
                                              const float* const* auxs, const int32_t* const* aux_maps, float* const* outs,
                                              float* const* h1s, float* const* h2s, int32_t n_jobs, int64_t n,
                                              const int32_t* n_dev, void* stream) {
    return launch_multi<true>(mlps, feats, feat_stride, auxs, aux_maps, outs, h1s, h2s, n_jobs, n, n_dev, stream);
}

extern "C" int tir_mlp_fwd_bf16(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                                float* out, int64_t n, const int32_t* n_dev, void* stream) {
    return launch_bf16<1>(m, feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, stream);
}

extern "C" int tir_mlp_fwd_valu(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux, const int32_t* aux_map, int32_t aux_mod,
                                float* out, int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !out))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_mlp_valu, dim3((unsigned)((n + 63) / 64)), dim3(64), 0, tir_stream(stream), m->packed,
                       feat, feat_stride, aux, aux_map, aux_mod, out, n, n_dev, m->out_dim, m->act);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_inputs(const TirMlp* m, const float* feat, int32_t feat_stride, const float* aux,
                              const int32_t* aux_map, int32_t aux_mod, float* x, int64_t n, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (n < 0 || feat_stride < F || (n > 0 && (!feat || !aux || !x))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_mlp_inputs, dim3((unsigned)((n + 7) / 8)), dim3(8 * (XPAD / 4)), 0, tir_stream(stream), feat,
                       feat_stride, aux, aux_map, aux_mod, x, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int64_t tir_mlp_bwd_packed_floats(int32_t feat_dim, int32_t pe, int32_t hidden, int32_t out_dim) {
    if (feat_dim != F || pe != PE || hidden != HID || out_dim < 1 || out_dim > 4) return TIR_ERR_UNSUPPORTED;
    return BWD_TOTAL_FLOATS;
}

extern "C" int tir_pack_mlp_bwd(const float* w0, const float* w1, const float* w2, int32_t feat_dim, int32_t pe,
                                int32_t hidden, int32_t out_dim, float* packed, void* stream) {
    if (!w0 || !w1 || !w2 || !packed) return TIR_ERR_ARG;
    if (feat_dim != F || pe != PE || hidden != HID || out_dim < 1 || out_dim > 4) return TIR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_pack_mlp_bwd, dim3((BWD_TOTAL_FLOATS + 255) / 256), dim3(256), 0, tir_stream(stream), w0, w1, w2,
                       out_dim, packed);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_bwd(const TirMlp* m, const float* packed_bwd, const float* feat, int32_t feat_stride,
                           const float* out, const float* g_out, const float* h1, const float* h2, int64_t n,
                           float* g_feat, float* dz1, float* dz2, float* dz3, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (!packed_bwd || n < 0 || feat_stride < F) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (!feat || !out || !g_out || !h1 || !h2 || !g_feat || !dz1 || !dz2 || !dz3) return TIR_ERR_ARG;
    const size_t lds = (size_t)BWD_FLOATS * sizeof(float);
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd), (int)lds)) return rc;
    int64_t tiles = (n + 255) / 256;
    unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(k_mlp_bwd, dim3(grid), dim3(512), lds, tir_stream(stream), packed_bwd, feat, feat_stride, out,
                       g_out, h1, h2, n, m->out_dim, m->act, g_feat, dz1, dz2, dz3);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_bwd_multi_bf16x3(const TirMlp* const* mlps, const float* const* packed_bwds, const float* const* feats,
                                       int32_t feat_stride, const float* const* outs, const float* const* g_outs,
                                       const float* const* h1s, const float* const* h2s, int32_t n_jobs, int64_t n,
                                       float* const* g_feats, float* const* dz1s, float* const* dz2s, float* const* dz3s, void* stream) {
    if (n_jobs < 1 || n_jobs > 4 || !mlps || !packed_bwds || !feats || !outs || !g_outs || !h1s || !h2s || !g_feats || !dz1s || !dz2s ||
        !dz3s || n < 0 || feat_stride < F)
        return TIR_ERR_ARG;
    TirMlpBwdJobs jobs;
    jobs.n_jobs = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        int rc = check_mlp(mlps[i]);
        if (rc) return rc;
        if (!packed_bwds[i]) return TIR_ERR_ARG;
        if (n > 0 && (!feats[i] || !outs[i] || !g_outs[i] || !h1s[i] || !h2s[i] || !g_feats[i] || !dz1s[i] || !dz2s[i] || !dz3s[i]))
            return TIR_ERR_ARG;
        jobs.j[i] = TirMlpBwdJob{packed_bwds[i], feats[i], outs[i], g_outs[i], h1s[i], h2s[i], g_feats[i], dz1s[i], dz2s[i], dz3s[i],
                                 mlps[i]->out_dim, mlps[i]->act};
    }
    if (n == 0) return TIR_OK;
    const size_t lds = (size_t)BWD_BF_LDS_BYTES;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_bf16_multi), (int)lds)) return rc;
    const int64_t tiles = (n + 255) / 256;
    int per = 256 / n_jobs;
    if (tiles < per) per = (int)tiles;
    hipLaunchKernelGGL(k_mlp_bwd_bf16_multi, dim3((unsigned)(per * n_jobs)), dim3(512), lds, tir_stream(stream), jobs, feat_stride, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_bwd_bf16x3(const TirMlp* m, const float* packed_bwd, const float* feat, int32_t feat_stride,
                           const float* out, const float* g_out, const float* h1, const float* h2, int64_t n,
                           float* g_feat, float* dz1, float* dz2, float* dz3, void* stream) {
    int rc = check_mlp(m);
    if (rc) return rc;
    if (!packed_bwd || n < 0 || feat_stride < F) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (!feat || !out || !g_out || !h1 || !h2 || !g_feat || !dz1 || !dz2 || !dz3) return TIR_ERR_ARG;
    const size_t lds = (size_t)BWD_BF_LDS_BYTES;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_bwd_bf16), (int)lds)) return rc;
    int64_t tiles = (n + 255) / 256;
    unsigned grid = (unsigned)(tiles < 256 ? tiles : 256);
    hipLaunchKernelGGL(k_mlp_bwd_bf16, dim3(grid), dim3(512), lds, tir_stream(stream), packed_bwd, feat, feat_stride, out,
                       g_out, h1, h2, n, m->out_dim, m->act, g_feat, dz1, dz2, dz3);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_mlp_wgrad_multi(const float* const* dz1s, const float* const* dz2s, const float* const* dz3s,
                                   const float* const* h1s, const float* const* h2s, const float* const* feats,
                                   int32_t feat_stride, const float* const* auxs, const int32_t* const* aux_maps,
                                   float* const* dW0s, float* const* db0s, float* const* dW1s, float* const* db1s,
                                   float* const* dW2s, float* const* db2s, int32_t n_jobs, int64_t n, int32_t max_workgroups,
                                   void* stream) {
    if (n_jobs < 1 || n_jobs > 4 || n < 0 || feat_stride < F || max_workgroups < 0) return TIR_ERR_ARG;
    if (!dz1s || !dz2s || !dz3s || !h1s || !h2s || !feats || !auxs || !dW0s || !db0s || !dW1s || !db1s || !dW2s || !db2s)
        return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    TirWgradJobs jobs;
    jobs.n_jobs = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        if (!dz1s[i] || !dz2s[i] || !dz3s[i] || !h1s[i] || !h2s[i] || !feats[i] || !auxs[i] || !dW0s[i] || !db0s[i] ||
            !dW1s[i] || !db1s[i] || !dW2s[i] || !db2s[i])
            return TIR_ERR_ARG;
        jobs.j[i] = TirWgradJob{dz1s[i], dz2s[i], dz3s[i], h1s[i], h2s[i], feats[i], auxs[i], aux_maps ? aux_maps[i] : nullptr,
                                dW0s[i], db0s[i], dW1s[i], db1s[i], dW2s[i], db2s[i]};
    }
    // A workgroup of 16 waves at 128 registers fills a CU's register file: with one per CU nothing else runs on the chip until
    // the launch ends.  These products are leaves of the backward (they run on a second stream beside the chain that feeds
    // the optimizer): max_workgroups (0 = 256) lets the caller leave CUs to that chain.
    const int total = max_workgroups > 0 ? (max_workgroups < 256 ? max_workgroups : 256) : 256;
    int per = total / n_jobs;                                // the grid split between the jobs
    if (per < 1) per = 1;
    const int64_t steps = (n + 15) / 16;
    if (steps < per) per = (int)steps;
    bool vec_ok = !(feat_stride & 3) && feat_stride >= 32 && n < ((int64_t)1 << 31);
    for (int i = 0; i < n_jobs && vec_ok; ++i)                                  // the staged kernel fetches rows as float4
        for (const float* q : {dz1s[i], dz2s[i], dz3s[i], h1s[i], h2s[i], feats[i]})
            if (reinterpret_cast<uintptr_t>(q) % 16 != 0) vec_ok = false;
    if (!vec_ok) {                                                              // rows not float4-addressable: the direct-load kernel
        hipLaunchKernelGGL(k_mlp_wgrad, dim3((unsigned)(per * n_jobs)), dim3(1024), 0, tir_stream(stream), jobs, feat_stride, n);
    } else {
        if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_mlp_wgrad_lds), WL_LDS_BYTES)) return rc;
        hipLaunchKernelGGL(k_mlp_wgrad_lds, dim3((unsigned)(per * n_jobs)), dim3(1024), WL_LDS_BYTES, tir_stream(stream), jobs,
                           feat_stride, n);
    }
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
