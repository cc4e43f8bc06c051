// Backward kernels of the training step: the reference gets these from
// torch.autograd over its ATen op chain (train_tensoIR.py:315-317); here each chain has a closed-form
// backward.  Scatter-adds into the (channel-last) parameter gradients use hardware fp32 atomics.
#include "tir_common.hpp"

using namespace tir;

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ void atomic_add_f32(float* p, float v) { unsafeAtomicAdd(p, v); }

// ------------------------------------------------------------------------------------------------
// Scatter of one VM group (plane i + line i) for ONE channel `c` of every tap.
// Value model:  feat = sum_ch P L,   du = sum_ch Pu L,  dv = sum_ch Pv L,  dw = sum_ch P Lw
// with P the bilinear plane value, Pu/Pv its derivatives per texel, L the linear line value, Lw = l1 - l0.
// Cotangents: F (feat) and -- NORMAL only -- Gu, Gv, Gw (of du, dv, dw; texel-scale factors already applied).
// NORMAL == false uses grid_sample's zero-padding weights (compute_densityfeature, F.grid_sample);
// NORMAL == true uses clamped indices + unclamped weights (models/relight_utils.py:82-92).
//
// Lane layout: min(CH, 16) ADJACENT lanes share a sample and own consecutive channels, so one wave atomic instruction
// lands 16 consecutive dwords in each 64-B segment it touches.  The L2 atomic rate on gfx950 is per 64-B segment per
// instruction (~20 G segments/s chip-wide, tools/atomic_bench.hip) whether 1 or 16 of its dwords are written: this
// layout moves 4x the gradient per segment of the float4-per-lane layout the forward gathers use.
//
// Run-length combining: a lane group walks CONSECUTIVE samples of a ray (half a voxel apart), which usually stay in
// one plane cell / line row for a few steps.  The tap gradients are summed in registers (VmRun) while the cell does not
// change and leave as atomics only when it does.
// ------------------------------------------------------------------------------------------------
template <int NIT>                      // NIT = channels per lane = CH / min(CH, 16)
struct VmRun {
    int r[4];                           // float offsets of the open plane cell's taps (r[0] < 0: closed)
    int q[2];                           // ... of the open line rows (q[0] < 0: closed)
    float a[NIT][4], b[NIT][2];
};

template <int NIT>
__device__ __forceinline__ void run_init(VmRun<NIT>& u) {
    u.r[0] = -1; u.q[0] = -1;
#pragma unroll
    for (int it = 0; it < NIT; ++it) { u.a[it][0] = u.a[it][1] = u.a[it][2] = u.a[it][3] = 0.f; u.b[it][0] = u.b[it][1] = 0.f; }
}
template <int CH, int NIT>
__device__ __forceinline__ void run_flush_plane(VmRun<NIT>& u, float* __restrict__ gplane, int c) {
    constexpr int LPS = CH < 16 ? CH : 16;
    if (u.r[0] < 0) return;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int t = 0; t < 4; ++t) { atomic_add_f32(gplane + u.r[t] + c + it * LPS, u.a[it][t]); u.a[it][t] = 0.f; }
}
template <int CH, bool LL, int NIT>
__device__ __forceinline__ void run_flush_line(VmRun<NIT>& u, float* __restrict__ gline, int c) {
    constexpr int LPS = CH < 16 ? CH : 16;
    if (u.q[0] < 0) return;
#pragma unroll
    for (int it = 0; it < NIT; ++it)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (LL) atomicAdd(gline + u.q[t] + c + it * LPS, u.b[it][t]);      // block-local LDS copy of the line gradient
            else atomic_add_f32(gline + u.q[t] + c + it * LPS, u.b[it][t]);
            u.b[it][t] = 0.f;
        }
}

template <int CH, bool NORMAL, bool LL>
__device__ __forceinline__ void scatter_group(VmRun<(CH + 15) / 16>& run, const float* __restrict__ plane,
                                              const float* __restrict__ line, float* __restrict__ gplane,
                                              float* __restrict__ gline, int H, int W, int R, float u, float v, float w,
                                              int c, float F, float Gu, float Gv, float Gw) {
    Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
    float wx0, wx1, wy0, wy1, wl0, wl1;
    if (NORMAL) { wx0 = 1.0f - tx.t; wx1 = tx.t; wy0 = 1.0f - ty.t; wy1 = ty.t; wl0 = 1.0f - tl.t; wl1 = tl.t; }
    else { wx0 = tx.w0; wx1 = tx.w1; wy0 = ty.w0; wy1 = ty.w1; wl0 = tl.w0; wl1 = tl.w1; }
    const float w00 = wx0 * wy0, w01 = wx1 * wy0, w10 = wx0 * wy1, w11 = wx1 * wy1;
    const int r0 = (ty.i0 * W + tx.i0) * CH, r1 = (ty.i0 * W + tx.i1) * CH;      // < 2^31 floats (check_grad_field)
    const int r2 = (ty.i1 * W + tx.i0) * CH, r3 = (ty.i1 * W + tx.i1) * CH;
    const int q0 = tl.i0 * CH, q1 = tl.i1 * CH;
    if (r0 != run.r[0] || r3 != run.r[3]) {                    // left the cell: the open run goes out
        run_flush_plane<CH>(run, gplane, c);
        run.r[0] = r0; run.r[1] = r1; run.r[2] = r2; run.r[3] = r3;
    }
    if (q0 != run.q[0] || q1 != run.q[1]) {
        run_flush_line<CH, LL>(run, gline, c);
        run.q[0] = q0; run.q[1] = q1;
    }
    // per-tap cotangent coefficients (Pu = (b-a) wy0 + (d-c) wy1,  Pv = (c-a) wx0 + (d-b) wx1)
    float a00 = F * w00, a01 = F * w01, a10 = F * w10, a11 = F * w11;
    if (NORMAL) {
        a00 += -Gu * wy0 - Gv * wx0; a01 += Gu * wy0 - Gv * wx1;
        a10 += -Gu * wy1 + Gv * wx0; a11 += Gu * wy1 + Gv * wx1;
    }
    constexpr int LPS = CH < 16 ? CH : 16;
#pragma unroll
    for (int it = 0; it < (CH + 15) / 16; ++it) {
        const int ch = c + it * LPS;
        const float av = plane[r0 + ch], bv = plane[r1 + ch], cv = plane[r2 + ch], dv = plane[r3 + ch];
        const float ev = line[q0 + ch], gv = line[q1 + ch];
        const float L = fmaf(gv, wl1, ev * wl0);
        const float P = fmaf(dv, w11, fmaf(cv, w10, fmaf(bv, w01, av * w00)));
        float t00 = L * a00, t01 = L * a01, t10 = L * a10, t11 = L * a11;
        float S = F * P;
        if (NORMAL) {
            const float Lw = gv - ev;
            t00 = fmaf(Lw, Gw * w00, t00); t01 = fmaf(Lw, Gw * w01, t01);
            t10 = fmaf(Lw, Gw * w10, t10); t11 = fmaf(Lw, Gw * w11, t11);
            const float Pu = fmaf(dv - cv, wy1, (bv - av) * wy0);
            const float Pv = fmaf(dv - bv, wx1, (cv - av) * wx0);
            S = fmaf(Gv, Pv, fmaf(Gu, Pu, S));
        }
        run.a[it][0] += t00; run.a[it][1] += t01; run.a[it][2] += t10; run.a[it][3] += t11;
        float s0 = S * wl0, s1 = S * wl1;
        if (NORMAL) { s0 = fmaf(-Gw, P, s0); s1 = fmaf(Gw, P, s1); }
        run.b[it][0] += s0; run.b[it][1] += s1;
    }
}

// the open runs of the three VM groups of the density field
template <int C4> struct DensityRuns { VmRun<(C4 * 4 + 15) / 16> g[3]; };

template <int C4, bool LL>
__device__ __forceinline__ float* density_line_grad(const TirField& f, const TirFieldGrad& g, float* lds_lines, int i) {
    // LDS layout of the three line gradients: line i at offset sum_{j<i} R_j * CH  (R_0 = grid z, R_1 = grid y)
    return LL ? lds_lines + (size_t)(i == 0 ? 0 : (i == 1 ? f.grid[2] : f.grid[2] + f.grid[1])) * (C4 * 4) : g.dline[i];
}

// the three VM groups of the density field for one sample; c = this lane's first channel (lane % min(CH, 16))
template <int C4, bool NORMAL, bool LL>
__device__ __forceinline__ void scatter_density(DensityRuns<C4>& runs, const TirField& f, const TirFieldGrad& g, float* lds_lines,
                                                float x, float y, float z, int c, float F, float G0, float G1, float G2) {
    const float p[3] = {x, y, z};
    const float G[3] = {G0, G1, G2};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int m0 = (i == 2) ? 1 : 0, m1 = (i == 0) ? 1 : 2, vi = 2 - i;
        const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
        float Gu = 0.f, Gv = 0.f, Gw = 0.f;
        if (NORMAL) { Gu = G[m0] * (0.5f * (float)(W - 1)); Gv = G[m1] * (0.5f * (float)(H - 1)); Gw = G[vi] * (0.5f * (float)(R - 1)); }
        scatter_group<C4 * 4, NORMAL, LL>(runs.g[i], f.dplane[i], f.dline[i], g.dplane[i],
                                          density_line_grad<C4, LL>(f, g, lds_lines, i), H, W, R, p[m0], p[m1], p[vi],
                                          c, F, Gu, Gv, Gw);
    }
}
template <int C4>
__device__ __forceinline__ void density_runs_init(DensityRuns<C4>& runs) {
#pragma unroll
    for (int i = 0; i < 3; ++i) run_init(runs.g[i]);
}
template <int C4, bool LL>
__device__ __forceinline__ void density_runs_flush(DensityRuns<C4>& runs, const TirField& f, const TirFieldGrad& g, float* lds_lines, int c) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        run_flush_plane<C4 * 4>(runs.g[i], g.dplane[i], c);
        run_flush_line<C4 * 4, LL>(runs.g[i], density_line_grad<C4, LL>(f, g, lds_lines, i), c);
        run_init(runs.g[i]);
    }
}

// block-local LDS copy of the three density-line gradients ([R_z + R_y + R_x][CH]): zero / flush helpers
template <int CH>
__device__ __forceinline__ void lines_zero(const TirField& f, float* lds_lines) {
    const int n = (f.grid[0] + f.grid[1] + f.grid[2]) * CH;
    for (int i = threadIdx.x; i < n; i += blockDim.x) lds_lines[i] = 0.0f;
}
template <int CH>
__device__ __forceinline__ void lines_flush(const TirField& f, const TirFieldGrad& g, const float* lds_lines) {
    const int n0 = f.grid[2] * CH, n1 = f.grid[1] * CH, n2 = f.grid[0] * CH;
    for (int i = threadIdx.x; i < n0 + n1 + n2; i += blockDim.x) {
        const float v = lds_lines[i];
        if (v == 0.0f) continue;
        if (i < n0) atomic_add_f32(g.dline[0] + i, v);
        else if (i < n0 + n1) atomic_add_f32(g.dline[1] + (i - n0), v);
        else atomic_add_f32(g.dline[2] + (i - n0 - n1), v);
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the primary march: one wave64 per ray, persistent blocks of 4 waves.
// ------------------------------------------------------------------------------------------------
#define TIR_MAX_CHUNKS 64      // S <= 4096

template <int C4, bool LL>
__global__ void __launch_bounds__(256)
k_march_primary_bwd(TirField f, TirFieldGrad g, const float* __restrict__ rays, const float* __restrict__ ray_jitter,
                    int B, int S, const float* __restrict__ sigma, const float* __restrict__ weight,
                    const float* __restrict__ gw, const float* __restrict__ g_acc, const float* __restrict__ g_depth,
                    float* __restrict__ g_feat_out) {
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    __shared__ float tstart[4][TIR_MAX_CHUNKS];
    __shared__ __attribute__((aligned(16))) float wl_all[4][256];
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    float* wl = wl_all[wv];
    if (LL) { lines_zero<C4 * 4>(f, lds_lines); __syncthreads(); }
    constexpr int LPS = C4 * 4 < 16 ? C4 * 4 : 16, PER = 64 / LPS;
    const int slot_in = lane / LPS, c = lane % LPS;
  for (int ray = blockIdx.x * 4 + wv; ray < B; ray += gridDim.x * 4) {
    RaySetup rs = ray_setup(f, rays, ray);
    DensityRuns<C4> runs;
    density_runs_init<C4>(runs);
    const bool hj = ray_jitter != nullptr;
    const float jit = hj ? ray_jitter[ray] : 0.0f;
    const int nch = (S + 63) / 64;
    // pass 1: transmittance at the start of every 64-sample chunk, with the forward's exact arithmetic
    float T = 1.0f;
    for (int ch = 0; ch < nch; ++ch) {
        const int k = ch * 64 + lane;
        float v = 1.0f;
        if (k < S) {
            const float z = sample_z(f, rs.t_min, k, jit, hj);
            const float dist = (k + 1 < S) ? sub_rn(sample_z(f, rs.t_min, k + 1, jit, hj), z) : 0.0f;
            const float alpha = 1.0f - expf(-sigma[(size_t)ray * S + k] * mul_rn(dist, f.distance_scale));
            v = add_rn(sub_rn(1.0f, alpha), 1e-10f);
        }
        if (lane == 0) tstart[wv][ch] = T;
        const float incl = scan_prod<64>(v, lane);
        T = T * __shfl(incl, 63, 64);
    }
    __builtin_amdgcn_wave_barrier();
    // pass 2: chunks in reverse, carrying  suffix = sum_{j in later chunks} g_j w_j
    const float ga = g_acc[ray], gd = g_depth[ray];
    float suffix = 0.0f;
    for (int ch = nch - 1; ch >= 0; --ch) {
        const int k = ch * 64 + lane;
        float gk = 0.f, wk = 0.f, sig = 0.f, z = 0.f, dist = 0.f, alpha = 0.f, v = 1.0f;
        if (k < S) {
            z = sample_z(f, rs.t_min, k, jit, hj);
            dist = (k + 1 < S) ? mul_rn(sub_rn(sample_z(f, rs.t_min, k + 1, jit, hj), z), f.distance_scale) : 0.0f;
            sig = sigma[(size_t)ray * S + k];
            wk = weight[(size_t)ray * S + k];
            gk = gw[(size_t)ray * S + k] + ga + z * gd;
            alpha = 1.0f - expf(-sig * dist);
            v = add_rn(sub_rn(1.0f, alpha), 1e-10f);
        }
        const float a = gk * wk;
        // sum of a over the lanes AFTER this one: a reverse scan of the shifted values (never `total - prefix`:
        // deep inside an opaque region both are ~1e-8 while the true suffix is 0, and it is divided by v ~ 1e-10)
        float rsum = __shfl_down(a, 1, 64);
        if (lane == 63) rsum = 0.0f;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const float o = __shfl_down(rsum, d, 64);
            if (lane + d < 64) rsum += o;
        }
        const float total = __shfl(rsum, 0, 64) + __shfl(a, 0, 64);
        const float sfx = suffix + rsum;
        const float incl = scan_prod<64>(v, lane);
        const float excl = shift_up1<64>(incl, lane);
        const float Tk = tstart[wv][ch] * excl;
        // w_k = alpha_k T_k, w_j (j > k) is proportional to (1 - alpha_k + 1e-10)
        const float dalpha = gk * Tk - sfx / v;
        const float dsig = dalpha * dist * expf(-sig * dist);              // d alpha / d sigma = dist exp(-sigma dist)
        const float df = (f.act == 1) ? (sig > 0.f ? dsig : 0.f) : dsig * (-expm1f(-sig));      // softplus' = sigmoid = 1 - e^-sigma (expm1: no cancellation for small sigma)
        suffix += total;
        if (g_feat_out && k < S) g_feat_out[(size_t)ray * S + k] = df;
        const bool on = (k < S) && (df != 0.0f);
        float x = 0.f, y = 0.f, zz = 0.f;
        if (on) {
            x = norm_coord(add_rn(rs.o[0], mul_rn(rs.d[0], z)), f.aabb_min[0], f.inv_aabb[0]);
            y = norm_coord(add_rn(rs.o[1], mul_rn(rs.d[1], z)), f.aabb_min[1], f.inv_aabb[1]);
            zz = norm_coord(add_rn(rs.o[2], mul_rn(rs.d[2], z)), f.aabb_min[2], f.inv_aabb[2]);
        }
        // wave-collective scatter: compact the active samples, min(CH, 16) lanes per sample, one channel per lane
        const unsigned long long m = __ballot(on);
        const int n = __popcll(m);
        if (n == 0) continue;
        const int rank = __popcll(m & ((1ull << lane) - 1ull));
        if (on) { wl[rank * 4] = x; wl[rank * 4 + 1] = y; wl[rank * 4 + 2] = zz; wl[rank * 4 + 3] = df; }
        __builtin_amdgcn_wave_barrier();
        // lane group g walks the CONSECUTIVE slots [g per, (g + 1) per): its open runs (VmRun) carry over from sample
        // to sample and from chunk to chunk of the ray
        const int per = (n + PER - 1) / PER;
        const int lo = slot_in * per, hi = min(n, lo + per);
        for (int t = 0; t < per; ++t) {
            const int slot = lo + t;
            if (slot < hi) {
                const float4 p = *reinterpret_cast<const float4*>(wl + slot * 4);
                scatter_density<C4, false, LL>(runs, f, g, lds_lines, p.x, p.y, p.z, c, p.w, 0.f, 0.f, 0.f);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }
    density_runs_flush<C4, LL>(runs, f, g, lds_lines, c);
  }
    if (LL) { __syncthreads(); lines_flush<C4 * 4>(f, g, lds_lines); }
}

// ------------------------------------------------------------------------------------------------
// Backward of the derived normal n = -g / max(|g|, 1e-6), g = softplus'(feat + shift) * grad feat
// min(CH, 16) adjacent lanes per sample, one channel per lane (the scatter layout, see scatter_group).
// ------------------------------------------------------------------------------------------------
template <int C4, bool LL>
__global__ void __launch_bounds__(256)
k_density_grad_bwd(TirField f, TirFieldGrad g, const float* __restrict__ xyz, const float* __restrict__ g_normal, int64_t n) {
    extern __shared__ __attribute__((aligned(16))) float lds_lines[];
    if (LL) { lines_zero<C4 * 4>(f, lds_lines); __syncthreads(); }
    constexpr int CH = C4 * 4, LPS = CH < 16 ? CH : 16;
    constexpr int RUN = 8;                 // consecutive samples per lane group (run-length combining, see VmRun)
  const int64_t n_grp = (n + RUN - 1) / RUN;
  const int64_t n_lanes = (n_grp * LPS + 255) / 256 * 256;     // whole blocks: the shuffles below need full lane groups
  for (int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; tid < n_lanes; tid += (int64_t)gridDim.x * blockDim.x) {
    const int64_t grp = tid / LPS;
    const int c = (int)(tid % LPS);
    DensityRuns<C4> runs;
    density_runs_init<C4>(runs);
#pragma unroll 1
   for (int step = 0; step < RUN; ++step) {
    const int64_t i = grp * RUN + step;
    const bool on = i < n;
    const int64_t ic = on ? i : n - 1;
    const float p[3] = {xyz[3 * ic], xyz[3 * ic + 1], xyz[3 * ic + 2]};
    // forward recompute: this lane's partial sums over its channel(s)
    float feat = 0.f, gr[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
        const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
        Tap1 tx = make_tap(p[m0], W), ty = make_tap(p[m1], H), tl = make_tap(p[vi], R);
        const float wx0 = 1.0f - tx.t, wx1 = tx.t, wy0 = 1.0f - ty.t, wy1 = ty.t;
        const float* pl = f.dplane[k];
        const float* ln = f.dline[k];
        const int r0 = (ty.i0 * W + tx.i0) * CH, r1 = (ty.i0 * W + tx.i1) * CH;
        const int r2 = (ty.i1 * W + tx.i0) * CH, r3 = (ty.i1 * W + tx.i1) * CH;
        const int q0 = tl.i0 * CH, q1 = tl.i1 * CH;
        float s_val = 0.f, s_du = 0.f, s_dv = 0.f, s_dw = 0.f;
#pragma unroll
        for (int ch = c; ch < CH; ch += LPS) {
            const float av = pl[r0 + ch], bv = pl[r1 + ch], cv = pl[r2 + ch], dv = pl[r3 + ch];
            const float ev = ln[q0 + ch], gv = ln[q1 + ch];
            const float P = fmaf(dv, wx1 * wy1, fmaf(cv, wx0 * wy1, fmaf(bv, wx1 * wy0, av * (wx0 * wy0))));
            const float Pu = fmaf(dv - cv, wy1, (bv - av) * wy0);
            const float Pv = fmaf(dv - bv, wx1, (cv - av) * wx0);
            const float L = fmaf(gv, tl.t, ev * (1.0f - tl.t));
            s_val = fmaf(P, L, s_val); s_du = fmaf(Pu, L, s_du); s_dv = fmaf(Pv, L, s_dv);
            s_dw = fmaf(P, gv - ev, s_dw);
        }
        feat += s_val;
        gr[m0] += s_du * (0.5f * (float)(W - 1));
        gr[m1] += s_dv * (0.5f * (float)(H - 1));
        gr[vi] += s_dw * (0.5f * (float)(R - 1));
    }
#pragma unroll
    for (int d = 1; d < LPS; d <<= 1) {
        feat += __shfl_xor(feat, d, 64);
        gr[0] += __shfl_xor(gr[0], d, 64); gr[1] += __shfl_xor(gr[1], d, 64); gr[2] += __shfl_xor(gr[2], d, 64);
    }
    if (!on) continue;
    float ds, dds;         // softplus' and softplus''
    if (f.act == 1) { ds = feat > 0.f ? 1.f : 0.f; dds = 0.f; }
    else {
        const float x = feat + f.density_shift;
        if (x > 20.f) { ds = 1.f; dds = 0.f; }
        else { ds = 1.0f / (1.0f + expf(-x)); dds = ds * (1.0f - ds); }
    }
    const float gx = ds * gr[0], gy = ds * gr[1], gz = ds * gr[2];
    const float nrm = sqrtf(gx * gx + gy * gy + gz * gz);
    const float dn[3] = {g_normal[3 * i], g_normal[3 * i + 1], g_normal[3 * i + 2]};
    float dg[3];
    if (nrm > 1e-6f) {                     // n = -g/|g|:  dg = -(dn - n (n . dn)) / |g|
        const float nx = -gx / nrm, ny = -gy / nrm, nz = -gz / nrm;
        const float dot = nx * dn[0] + ny * dn[1] + nz * dn[2];
        dg[0] = -(dn[0] - nx * dot) / nrm; dg[1] = -(dn[1] - ny * dot) / nrm; dg[2] = -(dn[2] - nz * dot) / nrm;
    } else { dg[0] = -dn[0] / 1e-6f; dg[1] = -dn[1] / 1e-6f; dg[2] = -dn[2] / 1e-6f; }
    const float F = (dg[0] * gr[0] + dg[1] * gr[1] + dg[2] * gr[2]) * dds;
    scatter_density<C4, true, LL>(runs, f, g, lds_lines, p[0], p[1], p[2], c, F, ds * dg[0], ds * dg[1], ds * dg[2]);
   }
    density_runs_flush<C4, LL>(runs, f, g, lds_lines, c);
  }
    if (LL) { __syncthreads(); lines_flush<C4 * 4>(f, g, lds_lines); }
}

// ------------------------------------------------------------------------------------------------
// Backward of compositing + tone mapping: one wave per ray, lanes stride over the ray's records (as the forward does).
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float srgb_grad(float x) {       // d linear2srgb / dx incl. the [0,1] clip (pass-through inclusive)
    if (!(x >= 0.0f && x <= 1.0f)) return 0.0f;
    if (x <= 0.0031308f) return 12.92f;
    return 1.055f * 0.41666666666666667f * powf(x + 1e-6f, 0.41666666666666667f - 1.0f);
}
__device__ __forceinline__ float clip01_grad(float x) { return (x >= 0.0f && x <= 1.0f) ? 1.0f : 0.0f; }

// relative smoothness term ((a - b) / max(max(a, b), 1e-6))^2  (models/tensorBase_rotated_lights.py:858-863)
__device__ __forceinline__ void rel_smooth_grad(float a, float b, float& da, float& db) {
    const float mx = fmaxf(a, b);
    const float base = fmaxf(mx, 1e-6f);
    const float dlt = (a - b) / base;
    // d base: through the clip only when mx >= 1e-6; torch.maximum sends the gradient to the larger (half each on ties)
    float ba = 0.f, bb = 0.f;
    if (mx >= 1e-6f) { if (a > b) ba = 1.f; else if (b > a) bb = 1.f; else { ba = 0.5f; bb = 0.5f; } }
    const float ddlt = 2.0f * dlt;
    const float dbase = -ddlt * (a - b) / (base * base);
    da = ddlt / base + dbase * ba;
    db = -ddlt / base + dbase * bb;
}

__global__ void __launch_bounds__(256)
k_composite_primary_bwd(const float* __restrict__ rays, const int32_t* __restrict__ offsets,
                        const int32_t* __restrict__ rec_k, const float* __restrict__ rec_w,
                        const float* __restrict__ rgb, const float* __restrict__ brdf,
                        const float* __restrict__ brdf_jit, const float* __restrict__ pred_n,
                        const float* __restrict__ der_n, const float* __restrict__ acc_in,
                        const float* __restrict__ depth_in, int B, int S, int white_bg, int is_relight,
                        float fixed_fresnel, const float* __restrict__ g_maps, float* __restrict__ g_rgb,
                        float* __restrict__ g_brdf, float* __restrict__ g_brdf_jit, float* __restrict__ g_pred,
                        float* __restrict__ g_der, float* __restrict__ g_weight, float* __restrict__ g_acc_out,
                        float* __restrict__ g_depth_out) {
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (r >= B) return;
    const int b = offsets[r], e = offsets[r + 1];
    const float* go = g_maps + (size_t)r * TIR_MAP_STRIDE;
    const float vd[3] = {rays[6 * (size_t)r + 3], rays[6 * (size_t)r + 4], rays[6 * (size_t)r + 5]};
    // ---- recompute the forward sums (same order as k_composite_primary) ----
    float c[3] = {0, 0, 0}, nm[3] = {0, 0, 0}, al[3] = {0, 0, 0}, rough = 0;
    for (int i = b + lane; i < e; i += 64) {
        const float w = rec_w[i];
        if (rgb) { c[0] = fmaf(w, rgb[3 * (size_t)i], c[0]); c[1] = fmaf(w, rgb[3 * (size_t)i + 1], c[1]); c[2] = fmaf(w, rgb[3 * (size_t)i + 2], c[2]); }
        if (!is_relight) continue;
        if (brdf) {
            al[0] = fmaf(w, brdf[4 * (size_t)i], al[0]); al[1] = fmaf(w, brdf[4 * (size_t)i + 1], al[1]); al[2] = fmaf(w, brdf[4 * (size_t)i + 2], al[2]);
            rough = fmaf(w, brdf[4 * (size_t)i + 3] * 0.9f + 0.09f, rough);
        }
        if (pred_n) { nm[0] = fmaf(w, pred_n[3 * (size_t)i], nm[0]); nm[1] = fmaf(w, pred_n[3 * (size_t)i + 1], nm[1]); nm[2] = fmaf(w, pred_n[3 * (size_t)i + 2], nm[2]); }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { c[q] = group_sum<64>(c[q]); nm[q] = group_sum<64>(nm[q]); al[q] = group_sum<64>(al[q]); }
    rough = group_sum<64>(rough);
    const float acc = acc_in[r];
    const float bg = 1.0f - acc;
    float g_acc = go[14];
    const float g_depth = go[3];
    float gc[3], gnm[3] = {0, 0, 0}, gal[3] = {0, 0, 0}, grough = 0.f;
    float g15 = 0.f, g16 = 0.f, g17 = 0.f, g18 = 0.f;
    if (white_bg) g_acc -= g_depth * rays[6 * (size_t)r + 5];
    if (!is_relight) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { gc[q] = go[q]; if (white_bg) g_acc -= gc[q]; }
    } else {
        float fr = fixed_fresnel;
        if (white_bg) { c[0] += bg; c[1] += bg; c[2] += bg; nm[2] += bg; al[0] += bg; al[1] += bg; al[2] += bg; rough += bg; fr += bg; }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            gc[q] = go[q] * srgb_grad(c[q]);
            gal[q] = go[7 + q] * clip01_grad(al[q]);
        }
        grough = go[10] * clip01_grad(rough);
        const float gfr = (go[11] + go[12] + go[13]) * clip01_grad(fr);
        // safe_l2_normalize(v) = v / max(|v|, 1e-6)
        const float nn = sqrtf(nm[0] * nm[0] + nm[1] * nm[1] + nm[2] * nm[2]);
        if (nn > 1e-6f) {
            const float n0 = nm[0] / nn, n1 = nm[1] / nn, n2 = nm[2] / nn;
            const float dot = n0 * go[4] + n1 * go[5] + n2 * go[6];
            gnm[0] = (go[4] - n0 * dot) / nn; gnm[1] = (go[5] - n1 * dot) / nn; gnm[2] = (go[6] - n2 * dot) / nn;
        } else { gnm[0] = go[4] / 1e-6f; gnm[1] = go[5] / 1e-6f; gnm[2] = go[6] / 1e-6f; }
        if (white_bg) g_acc -= gc[0] + gc[1] + gc[2] + gnm[2] + gal[0] + gal[1] + gal[2] + grough + gfr;
        g15 = go[15]; g16 = go[16]; g17 = go[17]; g18 = go[18];
    }
    if (lane == 0) { g_acc_out[r] = g_acc; g_depth_out[r] = g_depth; }
    // ---- per-record gradients ----
    for (int i = b + lane; i < e; i += 64) {
        const float w = rec_w[i];
        float gw = 0.f;
        if (rgb) {
#pragma unroll
            for (int q = 0; q < 3; ++q) { gw = fmaf(rgb[3 * (size_t)i + q], gc[q], gw); g_rgb[3 * (size_t)i + q] = w * gc[q]; }
        }
        if (is_relight) {
            float a3[3] = {0, 0, 0}, rg = 0.f;
            float gb[4] = {0, 0, 0, 0}, gbj[4] = {0, 0, 0, 0};
            if (brdf) {
                a3[0] = brdf[4 * (size_t)i]; a3[1] = brdf[4 * (size_t)i + 1]; a3[2] = brdf[4 * (size_t)i + 2];
                rg = brdf[4 * (size_t)i + 3] * 0.9f + 0.09f;
                gw += a3[0] * gal[0] + a3[1] * gal[1] + a3[2] * gal[2] + rg * grough;
                gb[0] = w * gal[0]; gb[1] = w * gal[1]; gb[2] = w * gal[2]; gb[3] = w * grough * 0.9f;
            }
            if (brdf && brdf_jit) {
                float cost = 0.f;
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float aj = brdf_jit[4 * (size_t)i + q];
                    const float base = fmaxf(fmaxf(a3[q], aj), 1e-6f);
                    const float dlt = (a3[q] - aj) / base;
                    cost = fmaf(dlt, dlt, cost);
                    float da, db;
                    rel_smooth_grad(a3[q], aj, da, db);
                    gb[q] = fmaf(w * g17, da, gb[q]);
                    gbj[q] = w * g17 * db;
                }
                const float rj = brdf_jit[4 * (size_t)i + 3] * 0.9f + 0.09f;
                const float base = fmaxf(fmaxf(rg, rj), 1e-6f);
                const float dlt = (rg - rj) / base;
                float da, db;
                rel_smooth_grad(rg, rj, da, db);
                gb[3] = fmaf(w * g18 * 0.9f, da, gb[3]);
                gbj[3] = w * g18 * 0.9f * db;
                gw += cost * g17 + dlt * dlt * g18;
            }
            if (g_brdf && brdf) { g_brdf[4 * (size_t)i] = gb[0]; g_brdf[4 * (size_t)i + 1] = gb[1]; g_brdf[4 * (size_t)i + 2] = gb[2]; g_brdf[4 * (size_t)i + 3] = gb[3]; }
            if (g_brdf_jit && brdf_jit) { g_brdf_jit[4 * (size_t)i] = gbj[0]; g_brdf_jit[4 * (size_t)i + 1] = gbj[1]; g_brdf_jit[4 * (size_t)i + 2] = gbj[2]; g_brdf_jit[4 * (size_t)i + 3] = gbj[3]; }
            if (pred_n) {
                const float p3[3] = {pred_n[3 * (size_t)i], pred_n[3 * (size_t)i + 1], pred_n[3 * (size_t)i + 2]};
                float gp[3] = {w * gnm[0], w * gnm[1], w * gnm[2]};
                gw += p3[0] * gnm[0] + p3[1] * gnm[1] + p3[2] * gnm[2];
                if (der_n) {
                    const float d0 = p3[0] - der_n[3 * (size_t)i], d1 = p3[1] - der_n[3 * (size_t)i + 1], d2 = p3[2] - der_n[3 * (size_t)i + 2];
                    gw += (d0 * d0 + d1 * d1 + d2 * d2) * g15;
                    const float s = 2.0f * w * g15;
                    gp[0] = fmaf(s, d0, gp[0]); gp[1] = fmaf(s, d1, gp[1]); gp[2] = fmaf(s, d2, gp[2]);
                    if (g_der) { g_der[3 * (size_t)i] = -s * d0; g_der[3 * (size_t)i + 1] = -s * d1; g_der[3 * (size_t)i + 2] = -s * d2; }
                }
                const float dot = vd[0] * p3[0] + vd[1] * p3[1] + vd[2] * p3[2];
                if (dot > 0.f) {                       // clamp(min=0): gradient passes for dot > 0
                    gw += dot * g16;
                    gp[0] = fmaf(w * g16, vd[0], gp[0]); gp[1] = fmaf(w * g16, vd[1], gp[1]); gp[2] = fmaf(w * g16, vd[2], gp[2]);
                }
                if (g_pred) { g_pred[3 * (size_t)i] = gp[0]; g_pred[3 * (size_t)i + 1] = gp[1]; g_pred[3 * (size_t)i + 2] = gp[2]; }
            }
        }
        g_weight[(size_t)r * S + rec_k[i]] = gw;
    }
}

// ------------------------------------------------------------------------------------------------
// Backward of the appearance feature (compute_bothfeature / compute_appfeature / compute_intrinfeature), two kernels:
//
//  k_app_dy        dY[n][3CA] = g_feat[n][app_dim] . basis_mat  -- the cotangent of the (plane (.) line (.) light) products,
//                  an fp32-MFMA GEMM, written INTO the y_rad / y_int output buffers (no extra workspace);
//  k_vm_app_bwd    the scatter: 16 adjacent lanes per sample, lane c owning channels c, c+16, ... of every tap (the
//                  layout of scatter_group: every plane atomic instruction lands 16 consecutive dwords per 64-B
//                  segment), reading dY in place and overwriting it with y = (plane (.) line) (.) light row.
//                  The gathers of item i+1 (one 16-channel run of one sample group) are issued before the atomics of
//                  item i, and every atomic is unconditional (zero-weight taps add 0.0), so the loop body is
//                  straight-line code whose loads complete while the previous atomics are still in flight.
// ------------------------------------------------------------------------------------------------
#define TIR_APP_MAX_L 16
#define TIR_APP_BWD_THREADS 1024

template <int C4>
__global__ void __launch_bounds__(256)
k_app_dy(const float* __restrict__ basis_t, int app_dim, const float* __restrict__ gfeat, int stride, int64_t n,
         float* __restrict__ y) {
    // dY tile [32 samples][32 columns] = G[32][28] W[28][32] on v_mfma_f32_32x32x2_f32 (exact fp32), k = 2 t + h.
    // A wave keeps ITS B operands -- the 14 x NT basis values of its lane (column li of every column tile, k parity h) -- in
    // registers for all the tiles it walks: no LDS image of basis_mat (round 2 filled a 21 KB LDS copy per workgroup, through
    // stride-32 reads, for ONE tile per wave: 0.14-0.19 ms per launch for a kernel whose traffic takes 35 us).
    constexpr int CA3 = 12 * C4, NT = (CA3 + 31) / 32;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, h = lane >> 5;
    float w[NT][14];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int t = 0; t < 14; ++t) {
            const int col = 32 * nt + li, kk = 2 * t + h;
            w[nt][t] = (col < CA3 && kk < app_dim) ? basis_t[(size_t)col * 32 + kk] : 0.0f;
        }
    const int64_t n_tile = (n + 31) / 32;
    for (int64_t tile = (int64_t)blockIdx.x * 4 + wave; tile < n_tile; tile += (int64_t)gridDim.x * 4) {
        const int64_t s0 = tile * 32, row = s0 + li;
        float a[14];
#pragma unroll
        for (int t = 0; t < 14; ++t) a[t] = (row < n && 2 * t + h < app_dim) ? gfeat[row * stride + 2 * t + h] : 0.0f;
        // column tiles in groups of <= 3: 48 accumulator registers live at a time (all NT at once put the kernel at 268
        // registers = one wave per SIMD); k-major inside a group: consecutive MFMAs go to different accumulators
#pragma unroll
        for (int g0 = 0; g0 < NT; g0 += 3) {
            constexpr int GMAX = 3;
            f32x16 acc[GMAX];
#pragma unroll
            for (int u = 0; u < GMAX; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
            for (int t = 0; t < 14; ++t)
#pragma unroll
                for (int u = 0; u < GMAX; ++u)
                    if (g0 + u < NT) acc[u] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t], w[g0 + u][t], acc[u], 0, 0, 0);
#pragma unroll
            for (int u = 0; u < GMAX; ++u) {
                if (g0 + u >= NT) continue;
                const int col = 32 * (g0 + u) + li;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int64_t rr = s0 + (r & 3) + 8 * (r >> 2) + 4 * h;
                    if (rr < n && col < CA3) y[rr * CA3 + col] = acc[u][r];
                }
            }
        }
    }
}

struct AppSamp { float x, y, z; int li; int64_t sc; bool on; };

__device__ __forceinline__ float ld_b(const float* base, unsigned byte_off) {          // saddr + 32-bit voffset form
    return *reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + byte_off);
}
__device__ __forceinline__ void atomic_b(float* base, unsigned byte_off, float v) {
    atomic_add_f32(reinterpret_cast<float*>(reinterpret_cast<char*>(base) + byte_off), v);
}

template <int C4, bool RAD, bool INTR>
__global__ void __launch_bounds__(TIR_APP_BWD_THREADS)
k_vm_app_bwd(TirField f, TirFieldGrad g, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
             const int32_t* __restrict__ idx_map, int64_t n, float* __restrict__ y_rad, float* __restrict__ y_int,
             int line_lds) {
    constexpr int CA = C4 * 4, CA3 = 3 * CA, NCQ = (CA + 15) / 16;
    extern __shared__ __attribute__((aligned(16))) float lds_ab[];
    const int nl = min(f.n_lights, TIR_APP_MAX_L);
    const bool lds_light = f.n_lights <= TIR_APP_MAX_L;
    float* Lt = lds_ab;                              // [(nl + 1)][3*CA] light_line rows, then light_mean
    float* gl = Lt + (nl + 1) * CA3;                 // same shape: block-local light_line / light_mean gradient
    // block-local gradient of ONE appearance line ([R][CA], the current VM group's): the line has only R rows, so
    // every sample of the batch lands on a few hundred addresses -- in L2 those atomics serialise; in LDS they are cheap
    float* lg = gl + (nl + 1) * CA3;
    const int nthr = blockDim.x, nwave = nthr >> 6;
    for (int i = threadIdx.x; i < (nl + 1) * CA3; i += nthr) {
        Lt[i] = (i < nl * CA3) ? (lds_light ? f.light_line[i] : 0.0f) : f.light_mean[i - nl * CA3];
        gl[i] = 0.0f;
    }
    const int L = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = L >> 4, c = L & 15;
    // A wave owns chunks of 4 x RUN consecutive samples; lane group j walks samples [RUN j, RUN j + RUN) of the chunk
    // one after the other.  Consecutive records are consecutive samples of a ray (half a voxel apart), so a group
    // usually stays in the same plane cell for 2-3 samples: the four tap gradients are summed in registers while
    // the cell does not change and leave as atomics only when it does (run-length combining).
    constexpr int RUN = 8;
    const int64_t n_chunk = (n + 4 * RUN - 1) / (4 * RUN), cstride = (int64_t)gridDim.x * nwave;
    // sample of this lane group at step i of a chunk (clamped to n-1 past the end), and its row in light_idx
    auto samp_of = [&](int64_t chunk, int i) { const int64_t s = chunk * (4 * RUN) + RUN * j + i; return s < n ? s : n - 1; };
    auto lsel_of = [&](int64_t sc) { return (RAD && idx_map) ? (int64_t)idx_map[sc] : sc; };
    // position + light of a sample; lsel comes from lsel_of() one step earlier so the two loads do not chain
    auto load_samp = [&](int64_t chunk, int i, int64_t lsel) {
        AppSamp sm;
        sm.on = chunk * (4 * RUN) + RUN * j + i < n;
        sm.sc = samp_of(chunk, i);
        sm.x = xyz[3 * sm.sc]; sm.y = xyz[3 * sm.sc + 1]; sm.z = xyz[3 * sm.sc + 2];
        sm.li = RAD ? min(max(light_idx[lsel], 0), f.n_lights - 1) : 0;
        return sm;
    };
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {                  // one VM group at a time over ALL of the block's samples
        const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
        const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
        if (line_lds)
            for (int i = threadIdx.x; i < R * CA; i += nthr) lg[i] = 0.0f;
        __syncthreads();
        const float* pl = f.aplane[k];
        const float* ln = f.aline[k];
        float* gp = g.aplane[k];
        float* gln = g.aline[k];
        for (int64_t chunk = (int64_t)blockIdx.x * nwave + wave; chunk < n_chunk; chunk += cstride) {
            AppSamp sm = load_samp(chunk, 0, lsel_of(samp_of(chunk, 0)));
            int64_t lsel_n = lsel_of(samp_of(chunk, 1));
            // the open run: tap byte offsets and the summed tap gradients of every 16-channel run of this lane
            unsigned r00 = 0xffffffffu, r01 = 0, r10 = 0, r11 = 0;
            float a00[NCQ], a01[NCQ], a10[NCQ], a11[NCQ];
            // ... the same for the line rows (runs are longer: one axis) and for the light row of the samples' light
            unsigned rq0 = 0xffffffffu, rq1 = 0;
            int rli = -1;
            float b0[NCQ], b1[NCQ], lr_acc[NCQ], lm_acc[NCQ];
#pragma unroll
            for (int cq = 0; cq < NCQ; ++cq) {
                a00[cq] = 0.f; a01[cq] = 0.f; a10[cq] = 0.f; a11[cq] = 0.f;
                b0[cq] = 0.f; b1[cq] = 0.f; lr_acc[cq] = 0.f; lm_acc[cq] = 0.f;
            }
            auto flush = [&]() {
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq) {
                    const unsigned cb = (16 * cq + c < CA) ? 64u * cq : 0u;     // dead lanes of a partial run add 0.0
                    atomic_b(gp, r00 + cb, a00[cq]); atomic_b(gp, r01 + cb, a01[cq]);
                    atomic_b(gp, r10 + cb, a10[cq]); atomic_b(gp, r11 + cb, a11[cq]);
                    a00[cq] = 0.f; a01[cq] = 0.f; a10[cq] = 0.f; a11[cq] = 0.f;
                }
            };
            auto flush_line = [&]() {
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq) {
                    const unsigned cb = (16 * cq + c < CA) ? 64u * cq : 0u;
                    if (line_lds) { atomicAdd(lg + (rq0 + cb) / 4u, b0[cq]); atomicAdd(lg + (rq1 + cb) / 4u, b1[cq]); }
                    else { atomic_b(gln, rq0 + cb, b0[cq]); atomic_b(gln, rq1 + cb, b1[cq]); }
                    b0[cq] = 0.f; b1[cq] = 0.f;
                }
            };
            auto flush_light = [&]() {
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq) {
                    const int ch = k * CA + ((16 * cq + c < CA) ? 16 * cq : 0) + c;
                    if (RAD) {
                        if (lds_light) atomicAdd(gl + rli * CA3 + ch, lr_acc[cq]);
                        else atomic_add_f32(g.light_line + (size_t)rli * CA3 + ch, lr_acc[cq]);
                    }
                    lr_acc[cq] = 0.f;
                }
            };
#pragma unroll 1
            for (int i = 0; i < RUN; ++i) {
                const float p[3] = {sm.x, sm.y, sm.z};
                Tap1 tx = make_tap(p[m0], W), ty = make_tap(p[m1], H), tl = make_tap(p[vi], R);
                const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
                // byte offsets of this lane's first channel in each tap: < 2^32 (check_grad_field)
                const unsigned o00 = ((unsigned)(ty.i0 * W + tx.i0) * CA + c) * 4u, o01 = ((unsigned)(ty.i0 * W + tx.i1) * CA + c) * 4u;
                const unsigned o10 = ((unsigned)(ty.i1 * W + tx.i0) * CA + c) * 4u, o11 = ((unsigned)(ty.i1 * W + tx.i1) * CA + c) * 4u;
                const unsigned q0 = ((unsigned)tl.i0 * CA + c) * 4u, q1 = ((unsigned)tl.i1 * CA + c) * 4u;
                const int li = sm.li;
                const bool on = sm.on;
                float* yr = RAD ? y_rad + sm.sc * CA3 + k * CA + c : nullptr;
                float* yi = INTR ? y_int + sm.sc * CA3 + k * CA + c : nullptr;
                // every gather of the step (and the next sample's position) in flight before anything is consumed: a
                // wave waits on ALL of its outstanding loads and atomics at once (vmcnt is shared), so one round
                // trip per step is what this costs
                float ga[NCQ], gb[NCQ], gc[NCQ], gd[NCQ], ge[NCQ], gg[NCQ], dyr[NCQ], dyi[NCQ];
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq) {
                    const bool chan = 16 * cq + c < CA;            // CA = 24: the second run is half empty
                    const unsigned cb = chan ? 64u * cq : 0u;
                    ga[cq] = ld_b(pl, o00 + cb); gb[cq] = ld_b(pl, o01 + cb); gc[cq] = ld_b(pl, o10 + cb); gd[cq] = ld_b(pl, o11 + cb);
                    ge[cq] = ld_b(ln, q0 + cb); gg[cq] = ld_b(ln, q1 + cb);
                    dyr[cq] = RAD ? yr[chan ? 16 * cq : 0] : 0.0f;
                    dyi[cq] = INTR ? yi[chan ? 16 * cq : 0] : 0.0f;
                }
                if (i + 1 < RUN) {
                    sm = load_samp(chunk, i + 1, lsel_n);
                    lsel_n = lsel_of(samp_of(chunk, i + 2 < RUN ? i + 2 : i + 1));
                }
                if (o00 != r00 || o11 != r11) {                    // left the cell: the open run goes out
                    if (r00 != 0xffffffffu) flush();
                    r00 = o00; r01 = o01; r10 = o10; r11 = o11;
                }
                if (q0 != rq0 || q1 != rq1) {
                    if (rq0 != 0xffffffffu) flush_line();
                    rq0 = q0; rq1 = q1;
                }
                if (RAD && li != rli) {
                    if (rli >= 0) flush_light();
                    rli = li;
                }
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq) {
                    const bool chan = 16 * cq + c < CA;
                    const bool live = on && chan;
                    const int ch = k * CA + (chan ? 16 * cq : 0) + c;      // channel of the 3*CA product vector
                    float lrv = 0.f, lmv = 0.f;
                    if (RAD) lrv = lds_light ? Lt[li * CA3 + ch] : f.light_line[(size_t)li * CA3 + ch];
                    if (INTR) lmv = Lt[nl * CA3 + ch];
                    const float er = live ? dyr[cq] : 0.f, ei = live ? dyi[cq] : 0.f;
                    const float P = fmaf(gd[cq], w11, fmaf(gc[cq], w10, fmaf(gb[cq], w01, ga[cq] * w00)));
                    const float Ln = fmaf(gg[cq], tl.w1, ge[cq] * tl.w0);
                    const float plv = P * Ln;
                    // lanes past CA redo run 0 of the same sample (same y); lanes past n redo sample n-1 and must
                    // not touch its y row (another group may not have read dY there yet); both add zeros below
                    if (on) {
                        if (RAD) yr[chan ? 16 * cq : 0] = plv * lrv;
                        if (INTR) yi[chan ? 16 * cq : 0] = plv * lmv;
                    }
                    if (RAD) lr_acc[cq] = fmaf(er, plv, lr_acc[cq]);
                    if (INTR) lm_acc[cq] = fmaf(ei, plv, lm_acc[cq]);
                    const float dpl = er * lrv + ei * lmv;
                    const float dP = dpl * Ln, dL = dpl * P;
                    a00[cq] = fmaf(dP, w00, a00[cq]); a01[cq] = fmaf(dP, w01, a01[cq]);
                    a10[cq] = fmaf(dP, w10, a10[cq]); a11[cq] = fmaf(dP, w11, a11[cq]);
                    b0[cq] = fmaf(dL, tl.w0, b0[cq]); b1[cq] = fmaf(dL, tl.w1, b1[cq]);
                }
            }
            if (r00 != 0xffffffffu) flush();
            if (rq0 != 0xffffffffu) flush_line();
            if (RAD && rli >= 0) flush_light();
            if (INTR) {
#pragma unroll
                for (int cq = 0; cq < NCQ; ++cq)
                    atomicAdd(gl + nl * CA3 + k * CA + ((16 * cq + c < CA) ? 16 * cq : 0) + c, lm_acc[cq]);
            }
        }
        __syncthreads();
        if (line_lds)
            for (int i = threadIdx.x; i < R * CA; i += nthr) {
                const float v = lg[i];
                if (v != 0.f) atomic_add_f32(g.aline[k] + i, v);
            }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < (nl + 1) * CA3; i += nthr) {
        const float v = gl[i];
        if (v == 0.f) continue;
        if (i < nl * CA3) { if (RAD && lds_light) atomic_add_f32(g.light_line + i, v); }
        else if (INTR) atomic_add_f32(g.light_mean + (i - nl * CA3), v);
    }
}

// ------------------------------------------------------------------------------------------------
// C[M][ldc] += A^T B  (A [n][lda], B [n][ldb]), fp32 MFMA 32x32x2, split over n, atomics into C.
// 4 waves; the (<= 4 x 5) 32x32 tiles of C are dealt round-robin to the waves (tile t = w + 4i, row tile t % MT:
// with M = 128 a wave keeps one row tile; with M <= 32 the column tiles spread over all four waves).  The next
// 32-row slab of A and B is fetched into registers while the current one is multiplied out of LDS.
// ------------------------------------------------------------------------------------------------
#define GT_AS 160      // LDS row strides == 32 mod 64 banks: the two k rows of an MFMA step hit disjoint banks
#define GT_BS 224

template <int TPW>                    // 32x32 tiles per wave: 4 * TPW >= MT * NT
__global__ void __launch_bounds__(256)
k_gemm_tn(const float* __restrict__ A, int lda, int M, const float* __restrict__ Bm, int ldb, int N, int ones_col,
          int64_t n, float* __restrict__ C, int ldc, int64_t chunk, float* __restrict__ bias_out) {
    __shared__ __attribute__((aligned(16))) float As[32 * GT_AS];
    __shared__ __attribute__((aligned(16))) float Bs[32 * GT_BS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int MT = (M + 31) / 32, NT = (N + ones_col + 31) / 32, T = MT * NT;
    const int64_t s0 = (int64_t)blockIdx.x * chunk;
    const int64_t s1 = min(n, s0 + chunk);
    int ta[TPW], tb[TPW];                 // LDS column offsets of this wave's tiles (slots past T redo tile 0, unused)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = (w + 4 * i < T) ? w + 4 * i : 0;
        ta[i] = 32 * (t % MT) + li; tb[i] = 32 * (t / MT) + li;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int Np = (N + 3) & ~3;
    const int a4 = (M + 3) >> 2, b4 = (N + ones_col + 3) >> 2;     // float4 granules per slab row actually needed
    // slab-invariant part of this thread's granules: row / column, global offsets, LDS offsets (-1: none)
    int a_lds[4], b_lds[5], a_row[4], b_row[5], b_col[5];
    int64_t a_off[4], b_off[5];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int e = threadIdx.x + 256 * q, row = e / a4, c4 = (e % a4) * 4;
        a_row[q] = row; a_lds[q] = row < 32 ? row * GT_AS + c4 : -1; a_off[q] = (int64_t)row * lda + c4;
    }
#pragma unroll
    for (int q = 0; q < 5; ++q) {
        const int e = threadIdx.x + 256 * q, row = e / b4, c4 = (e % b4) * 4;
        b_row[q] = row; b_col[q] = c4; b_lds[q] = row < 32 ? row * GT_BS + c4 : -1; b_off[q] = (int64_t)row * ldb + c4;
    }
    float4 ra[4], rb[5];
    auto fetch = [&](int64_t s) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (a_lds[q] >= 0 && s + a_row[q] < s1) ra[q] = *reinterpret_cast<const float4*>(A + s * lda + a_off[q]);
        }
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            rb[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (b_lds[q] >= 0 && s + b_row[q] < s1 && b_col[q] < Np) rb[q] = *reinterpret_cast<const float4*>(Bm + s * ldb + b_off[q]);
        }
    };
    auto stage = [&](int64_t s) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (a_lds[q] >= 0) *reinterpret_cast<float4*>(As + a_lds[q]) = ra[q];
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            if (b_lds[q] < 0) continue;
            float4 v = rb[q];
            const int d = N - b_col[q];             // columns >= N of the granule: padding -> 0, the ones column -> 1
            const float one = (ones_col && s + b_row[q] < s1) ? 1.0f : 0.0f;
            if (d <= 0) v.x = (d == 0) ? one : 0.f;
            if (d <= 1) v.y = (d == 1) ? one : 0.f;
            if (d <= 2) v.z = (d == 2) ? one : 0.f;
            if (d <= 3) v.w = (d == 3) ? one : 0.f;
            *reinterpret_cast<float4*>(Bs + b_lds[q]) = v;
        }
    };
    // columns of the LDS tiles beyond a4 / b4 granules are read by the MFMAs of partial tiles: zero them once
    for (int e = threadIdx.x; e < 32 * (GT_AS / 4); e += 256) *reinterpret_cast<float4*>(As + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < 32 * (GT_BS / 4); e += 256) *reinterpret_cast<float4*>(Bs + 4 * e) = make_float4(0.f, 0.f, 0.f, 0.f);
    fetch(s0);
    __syncthreads();
    for (int64_t s = s0; s < s1; s += 32) {
        stage(s);
        __syncthreads();
        if (s + 32 < s1) fetch(s + 32);
#pragma unroll 2
        for (int t = 0; t < 16; ++t) {
            const int kk = 2 * t + h;
#pragma unroll
            for (int i = 0; i < TPW; ++i)
                acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk * GT_AS + ta[i]], Bs[kk * GT_BS + tb[i]], acc[i], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = w + 4 * i;
        const int jcol = 32 * (t / MT) + li;
        const int row0 = 32 * (t % MT) + 4 * h;
        const bool col_ok = t < T && jcol < N + ones_col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            if (col_ok && row < M) {
                if (bias_out && jcol == N) atomic_add_f32(bias_out + row, acc[i][r]);      // the ones column, kept apart
                else atomic_add_f32(C + (size_t)row * ldc + jcol, acc[i][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// The same product on the bf16 matrix pipe: every element x = hi + lo (two bf16), three products hi*hi + lo*hi + hi*lo
// with fp32 accumulation (|error| ~ 2^-16 per product, the scheme of the split-bf16 decoders).
// v_mfma_f32_32x32x16_bf16 wants 8 CONSECUTIVE k (= rows of A / B) per lane, so the 32-row slab is transposed on its
// way into LDS: At[col][k], Bt[col][k] as bf16, hi and lo planes, 80-B column stride (conflict-free ds_read_b128).
// A thread stages (row pair, 4 columns) items -- one packed u32 per column and plane; lanes run over 8 row pairs x 8
// column granules, so global loads fetch whole 128-B lines and the transposing LDS writes are at most 2-way conflicted.
// 120 MFMAs of 32 cycles per slab and block instead of 320 of 64.
// ------------------------------------------------------------------------------------------------
#define GB_RS 40          // bf16 elements per LDS column (32 k + 8 pad)

typedef __bf16 gb_bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 gb_bf16x2 __attribute__((ext_vector_type(2)));
typedef float gb_f32x2 __attribute__((ext_vector_type(2)));

__device__ __forceinline__ void gb_split2(float x0, float x1, unsigned& hi, unsigned& lo) {
    const gb_f32x2 x = {x0, x1};
    hi = __builtin_bit_cast(unsigned, __builtin_convertvector(x, gb_bf16x2));
    const gb_f32x2 hf = {__builtin_bit_cast(float, hi << 16), __builtin_bit_cast(float, hi & 0xffff0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(x - hf, gb_bf16x2));
}

template <int TPW, bool SHARE_A>      // SHARE_A: MT == 4, a wave's tiles all sit in row tile w
__global__ void __launch_bounds__(256)
k_gemm_tn_bf16(const float* __restrict__ A, int lda, int M, const float* __restrict__ Bm, int ldb, int N, int ones_col,
               int64_t n, float* __restrict__ C, int ldc, int64_t chunk, float* __restrict__ bias_out) {
    __shared__ __attribute__((aligned(16))) unsigned short At[2][128 * GB_RS];      // [hi / lo][column][k]
    __shared__ __attribute__((aligned(16))) unsigned short Bt[2][160 * GB_RS];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    const int MT = (M + 31) / 32, NT = (N + ones_col + 31) / 32, T = MT * NT;
    const int64_t s0 = (int64_t)blockIdx.x * chunk;
    const int64_t s1 = min(n, s0 + chunk);
    int ta[TPW], tb[TPW];                 // LDS columns of this wave's tiles (slots past T redo tile 0, unused)
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = (w + 4 * i < T) ? w + 4 * i : 0;
        ta[i] = 32 * (t % MT) + li; tb[i] = 32 * (t / MT) + li;
    }
    f32x16 acc[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const int a4 = (M + 3) >> 2, b4 = (N + ones_col + 3) >> 2, Np4 = (N + 3) >> 2;
    // staging items of this wave: item w + 4 q of 8 (A: 2 row-pair halves x 4 granule blocks) + 10 (B: 2 x 5)
    const int rp_l = lane & 7, g_l = lane >> 3;
    float4 r0[5], r1[5];
    auto item = [&](int q, bool& isA, int& rp, int& g) {
        const int wi = w + 4 * q;
        isA = wi < 8;
        const int v = isA ? wi : wi - 8;
        rp = 8 * (v & 1) + rp_l;                     // row pair 0..15 -> slab rows 2 rp, 2 rp + 1
        g = 8 * (v >> 1) + g_l;                      // float4 granule (4 columns)
        return wi < 18 && (isA ? g < a4 : g < b4);
    };
    auto fetch = [&](int64_t s) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            bool isA; int rp, g;
            r0[q] = make_float4(0.f, 0.f, 0.f, 0.f); r1[q] = r0[q];
            if (!item(q, isA, rp, g)) continue;
            if (!isA && g >= Np4) continue;          // the granule that only holds the ones column
            const float* src = isA ? A + (s + 2 * rp) * lda + 4 * g : Bm + (s + 2 * rp) * ldb + 4 * g;
            const int ld = isA ? lda : ldb;
            if (s + 2 * rp < s1) r0[q] = *reinterpret_cast<const float4*>(src);
            if (s + 2 * rp + 1 < s1) r1[q] = *reinterpret_cast<const float4*>(src + ld);
        }
    };
    auto stage = [&](int64_t s) {
#pragma unroll
        for (int q = 0; q < 5; ++q) {
            bool isA; int rp, g;
            if (!item(q, isA, rp, g)) continue;
            const float x0[4] = {r0[q].x, r0[q].y, r0[q].z, r0[q].w}, x1[4] = {r1[q].x, r1[q].y, r1[q].z, r1[q].w};
            const float one0 = (ones_col && s + 2 * rp < s1) ? 1.0f : 0.0f, one1 = (ones_col && s + 2 * rp + 1 < s1) ? 1.0f : 0.0f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int col = 4 * g + j;
                float v0 = x0[j], v1 = x1[j];
                if (!isA && col >= N) { v0 = (col == N) ? one0 : 0.f; v1 = (col == N) ? one1 : 0.f; }   // padding / ones column
                if (isA && col >= M) { v0 = 0.f; v1 = 0.f; }
                unsigned hi, lo;
                gb_split2(v0, v1, hi, lo);
                unsigned* dh = reinterpret_cast<unsigned*>(isA ? At[0] : Bt[0]) + (col * GB_RS) / 2 + rp;
                unsigned* dl = reinterpret_cast<unsigned*>(isA ? At[1] : Bt[1]) + (col * GB_RS) / 2 + rp;
                *dh = hi; *dl = lo;
            }
        }
    };
    // columns never staged (beyond a4 / b4 granules) are read by the MFMAs of partial tiles: zero everything once
    for (int e = threadIdx.x; e < 2 * 128 * GB_RS / 8; e += 256) reinterpret_cast<float4*>(&At[0][0])[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int e = threadIdx.x; e < 2 * 160 * GB_RS / 8; e += 256) reinterpret_cast<float4*>(&Bt[0][0])[e] = make_float4(0.f, 0.f, 0.f, 0.f);
    fetch(s0);
    __syncthreads();
    for (int64_t s = s0; s < s1; s += 32) {
        stage(s);
        __syncthreads();
        if (s + 32 < s1) fetch(s + 32);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int kb = 16 * ks + 8 * h;
            gb_bf16x8 ah[TPW], al[TPW], bh[TPW], bl[TPW];
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                if (!SHARE_A || i == 0) {
                    ah[i] = *reinterpret_cast<const gb_bf16x8*>(&At[0][ta[i] * GB_RS + kb]);
                    al[i] = *reinterpret_cast<const gb_bf16x8*>(&At[1][ta[i] * GB_RS + kb]);
                } else { ah[i] = ah[0]; al[i] = al[0]; }
                bh[i] = *reinterpret_cast<const gb_bf16x8*>(&Bt[0][tb[i] * GB_RS + kb]);
                bl[i] = *reinterpret_cast<const gb_bf16x8*>(&Bt[1][tb[i] * GB_RS + kb]);
            }
            // product-major: never two consecutive MFMAs on one accumulator
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < TPW; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[i], acc[i], 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < TPW; ++i) {
        const int t = w + 4 * i;
        const int jcol = 32 * (t / MT) + li;
        const int row0 = 32 * (t % MT) + 4 * h;
        const bool col_ok = t < T && jcol < N + ones_col;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = row0 + (r & 3) + 8 * (r >> 2);
            if (col_ok && row < M) {
                if (bias_out && jcol == N) atomic_add_f32(bias_out + row, acc[i][r]);      // the ones column, kept apart
                else atomic_add_f32(C + (size_t)row * ldc + jcol, acc[i][r]);
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// C[M <= 32][N <= 160] += sum over up to three (A_j, B_j) pairs of A_j[n][lda]^T B_j[n][ldb] -- the basis-matrix gradient
// d basis_mat = g_feat^T y (27 x 144) of the stage's three appearance gathers in one launch.  Same operand path as
// tir_mlp_wgrad_multi: a lane of v_mfma_f32_32x32x16_bf16 supplies 8 consecutive k (= rows) of one column, and for a fixed
// row the 32 lanes of a half-wave read 32 consecutive columns, so the transposed operand is a plain coalesced load -- no
// LDS staging, no barrier; split-bf16 (3 products, fp32 accumulation).  A wave owns ALL column tiles of the single row tile
// and walks every fourth 16-row step of its workgroup's chunk; partial sums leave through fp32 atomics.
// (k_gemm_tn_bf16 above stages 32-row slabs through LDS with two barriers per slab: 0.18 ms for this 230 k x (27, 144)
// product, ~5 x its HBM time.)
// ------------------------------------------------------------------------------------------------
struct TirSmallGemmJob { const float* A; const float* B; };
struct TirSmallGemmJobs { TirSmallGemmJob j[3]; int n_jobs; };

__device__ __forceinline__ void gs_split8(const float (&v)[8], gb_bf16x8& hi, gb_bf16x8& lo) {
    unsigned H[4], L[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) gb_split2(v[2 * p], v[2 * p + 1], H[p], L[p]);
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    const u4 hv = {H[0], H[1], H[2], H[3]}, lv = {L[0], L[1], L[2], L[3]};
    hi = __builtin_bit_cast(gb_bf16x8, hv);
    lo = __builtin_bit_cast(gb_bf16x8, lv);
}

template <int NT>
__global__ void __launch_bounds__(256)
k_gemm_tn_small(TirSmallGemmJobs jobs, int lda, int M, int ldb, int N, int64_t n, float* __restrict__ C, int ldc) {
    const int per = (int)gridDim.x / jobs.n_jobs;
    const int ji = (int)blockIdx.x / per;
    if (ji >= jobs.n_jobs) return;
    const float* __restrict__ A = jobs.j[ji].A;
    const float* __restrict__ Bm = jobs.j[ji].B;
    const int bid = (int)blockIdx.x - ji * per;
    int64_t chunk = (n + per - 1) / per;
    chunk = (chunk + 15) / 16 * 16;
    const int64_t r0 = (int64_t)bid * chunk, r1 = min(n, r0 + chunk);
    if (r0 >= r1) return;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int li = lane & 31, h = lane >> 5;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const bool a_ok = li < M;
    for (int64_t kb = r0 + 16 * w; kb < r1; kb += 64) {
        const int64_t s0 = kb + 8 * h;
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = (a_ok && s0 + j < r1) ? A[(s0 + j) * lda + li] : 0.f;
        gb_bf16x8 ah, al, bh[NT], bl[NT];
        gs_split8(v, ah, al);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int col = t * 32 + li;
            float y[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) y[j] = (col < N && s0 + j < r1) ? Bm[(s0 + j) * ldb + col] : 0.f;
            gs_split8(y, bh[t], bl[t]);
        }
        // product-major: never two consecutive MFMAs on one accumulator
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[t], acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[t], acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int col = t * 32 + li;
        if (col >= N) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = 4 * h + (r & 3) + 8 * (r >> 2);
            if (row < M) atomic_add_f32(C + (size_t)row * ldc + col, acc[t][r]);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// Shading backward: one wave per surface point, lanes over light directions.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void normalize3(float& x, float& y, float& z, float eps) {
    const float n = fmaxf(sqrtf(x * x + y * y + z * z), eps);
    x /= n; y /= n; z /= n;
}
__device__ __forceinline__ float clamp_pass(float raw, float lo, float hi) { return (raw >= lo && raw <= hi) ? 1.0f : 0.0f; }

// One surface point on one wave.  genv: where the environment-radiance gradient [n_lights][D][3] accumulates -- the
// block's LDS copy (lds_env) or, when that does not fit, global memory.
__device__ __forceinline__ void
shade_bwd_point(int m, int lane, const float* __restrict__ maps, const float* __restrict__ rays, const float* __restrict__ dirs,
                const int32_t* __restrict__ light_idx, const float* __restrict__ vis,
                const float* __restrict__ indirect, const float* __restrict__ env,
                const float* __restrict__ weight_d, int D, int n_lights, int equal_area, int use_srgb,
                float acc_thres, const float* __restrict__ g_out, float* __restrict__ g_maps,
                float* __restrict__ g_env, float* lds_env) {
    const float* mp = maps + (size_t)m * TIR_MAP_STRIDE;
    float* gm = g_maps + (size_t)m * TIR_MAP_STRIDE;
    if (lane < TIR_MAP_STRIDE) gm[lane] = 0.0f;
    if (!(mp[14] > acc_thres)) return;                       // background row: constant output
    const float PI = 3.14159265358979323846f, four_pi = 4.0f * PI;
    const float* r = rays + 6 * (size_t)m;
    float V[3] = {-r[3], -r[4], -r[5]};
    normalize3(V[0], V[1], V[2], 1e-6f);                      // safe_l2_normalize(-rays_d)
    normalize3(V[0], V[1], V[2], 1e-12f);                     // F.normalize inside GGX_specular
    const float nraw[3] = {mp[4], mp[5], mp[6]};
    const float nlen = fmaxf(sqrtf(nraw[0] * nraw[0] + nraw[1] * nraw[1] + nraw[2] * nraw[2]), 1e-12f);
    const float Nn[3] = {nraw[0] / nlen, nraw[1] / nlen, nraw[2] / nlen};
    const float nov0 = V[0] * Nn[0] + V[1] * Nn[1] + V[2] * Nn[2];
    const float sg = (nov0 > 0.f) ? 1.f : ((nov0 < 0.f) ? -1.f : 0.f);
    const float N[3] = {Nn[0] * sg, Nn[1] * sg, Nn[2] * sg};
    const float nov_raw = N[0] * V[0] + N[1] * V[1] + N[2] * V[2];
    const float NoV = fminf(fmaxf(nov_raw, 1e-6f), 1.f);
    const float rr = mp[10];
    const float a1 = rr * rr, a2 = a1 * a1, kk = (a1 + 2.f * rr + 1.0f) / 8.0f;
    const float F0[3] = {mp[11], mp[12], mp[13]};
    const float alb_pi[3] = {mp[7] / PI, mp[8] / PI, mp[9] / PI};
    int li = light_idx ? light_idx[m] : 0;
    li = min(max(li, 0), n_lights - 1);
    const float* envl = env + (size_t)li * D * 3;
    // pass 1: forward totals (for the clip / sRGB derivative)
    float c[3] = {0.f, 0.f, 0.f};
    for (int d = lane; d < D; d += 64) {
        float lx = dirs[3 * d], ly = dirs[3 * d + 1], lz = dirs[3 * d + 2];
        const float cosine = fmaxf(lx * nraw[0] + ly * nraw[1] + lz * nraw[2], 0.f);
        normalize3(lx, ly, lz, 1e-12f);
        float hx = (lx + V[0]) / 2.0f, hy = (ly + V[1]) / 2.0f, hz = (lz + V[2]) / 2.0f;
        normalize3(hx, hy, hz, 1e-12f);
        const float NoL = fminf(fmaxf(N[0] * lx + N[1] * ly + N[2] * lz, 1e-6f), 1.f);
        const float NoH = fminf(fmaxf(N[0] * hx + N[1] * hy + N[2] * hz, 1e-6f), 1.f);
        const float VoH = fminf(fmaxf(V[0] * hx + V[1] * hy + V[2] * hz, 1e-6f), 1.f);
        const float p2 = exp2f(((-5.55473f) * VoH - 6.98316f) * VoH);
        const size_t md = (size_t)m * D + d;
        const float v = vis[md], wd = equal_area ? four_pi : weight_d[d];
        const float nom0 = NoH * NoH * (a2 - 1.f) + 1.f, nom1 = NoV * (1.f - kk) + kk, nom2 = NoL * (1.f - kk) + kk;
        const float nom = fminf(fmaxf(four_pi * nom0 * nom0 * nom1 * nom2, 1e-6f), four_pi);
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float spec = (F0[q] + (1.f - F0[q]) * p2) * a2 / nom;
            const float light = v * envl[3 * d + q] + (indirect ? indirect[3 * md + q] : 0.f);
            c[q] += (alb_pi[q] + spec) * light * cosine * wd;
        }
    }
    float gt[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float t = group_sum<64>(c[q]);
        if (equal_area) t /= (float)D;
        const float go = g_out[3 * (size_t)m + q];
        gt[q] = use_srgb ? go * srgb_grad(t) : go * clip01_grad(t);
        if (equal_area) gt[q] /= (float)D;
    }
    // pass 2: per-direction chain rule
    float gN[3] = {0, 0, 0}, gnraw[3] = {0, 0, 0}, galb[3] = {0, 0, 0}, gF0[3] = {0, 0, 0};
    float gNoV = 0.f, grough = 0.f;
    for (int d = lane; d < D; d += 64) {
        const float l0 = dirs[3 * d], l1 = dirs[3 * d + 1], l2 = dirs[3 * d + 2];
        float lx = l0, ly = l1, lz = l2;
        const float cos_raw = l0 * nraw[0] + l1 * nraw[1] + l2 * nraw[2];
        const float cosine = fmaxf(cos_raw, 0.f);
        normalize3(lx, ly, lz, 1e-12f);
        float hx = (lx + V[0]) / 2.0f, hy = (ly + V[1]) / 2.0f, hz = (lz + V[2]) / 2.0f;
        normalize3(hx, hy, hz, 1e-12f);
        const float nol_raw = N[0] * lx + N[1] * ly + N[2] * lz, noh_raw = N[0] * hx + N[1] * hy + N[2] * hz;
        const float NoL = fminf(fmaxf(nol_raw, 1e-6f), 1.f), NoH = fminf(fmaxf(noh_raw, 1e-6f), 1.f);
        const float VoH = fminf(fmaxf(V[0] * hx + V[1] * hy + V[2] * hz, 1e-6f), 1.f);
        const float p2 = exp2f(((-5.55473f) * VoH - 6.98316f) * VoH);
        const size_t md = (size_t)m * D + d;
        const float v = vis[md], wd = equal_area ? four_pi : weight_d[d];
        const float nom0 = NoH * NoH * (a2 - 1.f) + 1.f, nom1 = NoV * (1.f - kk) + kk, nom2 = NoL * (1.f - kk) + kk;
        const float nom_raw = four_pi * nom0 * nom0 * nom1 * nom2;
        const float nom = fminf(fmaxf(nom_raw, 1e-6f), four_pi);
        const float nom_pass = clamp_pass(nom_raw, 1e-6f, four_pi);
        float dcos = 0.f, dNoH = 0.f, dNoL = 0.f, dNoVd = 0.f, da2 = 0.f, dk = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float frac0 = F0[q] + (1.f - F0[q]) * p2;
            const float frac = frac0 * a2;
            const float spec = frac / nom;
            const float direct = envl[3 * d + q];
            const float light = v * direct + (indirect ? indirect[3 * md + q] : 0.f);
            const float brdf = alb_pi[q] + spec;
            const float gterm = gt[q] * wd;                       // cotangent of brdf * light * cosine
            galb[q] += gterm * light * cosine / PI;
            dcos += gterm * brdf * light;
            if (g_env && v != 0.f) {
                const float ge = gterm * brdf * cosine * v;
                if (lds_env) atomicAdd(lds_env + (li * D + d) * 3 + q, ge);
                else atomic_add_f32(g_env + ((size_t)li * D + d) * 3 + q, ge);
            }
            const float gs = gterm * light * cosine;               // cotangent of spec
            const float dfrac = gs / nom;
            const float dnomr = -gs * frac / (nom * nom) * nom_pass;
            const float dnom0 = dnomr * four_pi * 2.f * nom0 * nom1 * nom2;
            const float dnom1 = dnomr * four_pi * nom0 * nom0 * nom2;
            const float dnom2 = dnomr * four_pi * nom0 * nom0 * nom1;
            da2 += dfrac * frac0 + dnom0 * NoH * NoH;
            gF0[q] += dfrac * a2 * (1.f - p2);
            dNoH += dnom0 * 2.f * NoH * (a2 - 1.f);
            dNoVd += dnom1 * (1.f - kk);
            dNoL += dnom2 * (1.f - kk);
            dk += dnom1 * (1.f - NoV) + dnom2 * (1.f - NoL);
        }
        grough += da2 * 4.f * a1 * rr + dk * (2.f * rr + 2.f) / 8.0f;
        gNoV += dNoVd;
        dNoH *= clamp_pass(noh_raw, 1e-6f, 1.f);
        dNoL *= clamp_pass(nol_raw, 1e-6f, 1.f);
        gN[0] += dNoH * hx + dNoL * lx; gN[1] += dNoH * hy + dNoL * ly; gN[2] += dNoH * hz + dNoL * lz;
        if (cos_raw > 0.f) { gnraw[0] += dcos * l0; gnraw[1] += dcos * l1; gnraw[2] += dcos * l2; }
    }
    gNoV = group_sum<64>(gNoV) * clamp_pass(nov_raw, 1e-6f, 1.f);
    grough = group_sum<64>(grough);
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        gN[q] = group_sum<64>(gN[q]) + gNoV * V[q];
        gnraw[q] = group_sum<64>(gnraw[q]);
        galb[q] = group_sum<64>(galb[q]);
        gF0[q] = group_sum<64>(gF0[q]);
    }
    if (lane == 0) {
        // N = sg * normalize(nraw)
        const float dNn[3] = {gN[0] * sg, gN[1] * sg, gN[2] * sg};
        const float dot = Nn[0] * dNn[0] + Nn[1] * dNn[1] + Nn[2] * dNn[2];
        gm[4] = gnraw[0] + (dNn[0] - Nn[0] * dot) / nlen;
        gm[5] = gnraw[1] + (dNn[1] - Nn[1] * dot) / nlen;
        gm[6] = gnraw[2] + (dNn[2] - Nn[2] * dot) / nlen;
        gm[7] = galb[0]; gm[8] = galb[1]; gm[9] = galb[2];
        gm[10] = grough;
        gm[11] = gF0[0]; gm[12] = gF0[1]; gm[13] = gF0[2];
    }
}

// Persistent blocks of 4 waves, one surface point per wave at a time.  Every point adds to the SAME n_lights x D x 3
// environment-gradient entries: left as global atomics that is an M-way collision per address (4096 waves serialise
// on 384 words), so the block sums its points in LDS and flushes once.
__global__ void __launch_bounds__(256)
k_shade_integrate_bwd(const float* __restrict__ maps, const float* __restrict__ rays, const float* __restrict__ dirs,
                      const int32_t* __restrict__ light_idx, const float* __restrict__ vis,
                      const float* __restrict__ indirect, const float* __restrict__ env,
                      const float* __restrict__ weight_d, int M, int D, int n_lights, int equal_area, int use_srgb,
                      float acc_thres, const float* __restrict__ g_out, float* __restrict__ g_maps,
                      float* __restrict__ g_env, int env_in_lds) {
    extern __shared__ float lds_env_buf[];
    float* lds_env = (g_env && env_in_lds) ? lds_env_buf : nullptr;
    const int n_env = n_lights * D * 3;
    if (lds_env) {
        for (int i = threadIdx.x; i < n_env; i += blockDim.x) lds_env[i] = 0.0f;
        __syncthreads();
    }
    const int lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    for (int m = blockIdx.x * nw + (threadIdx.x >> 6); m < M; m += gridDim.x * nw)
        shade_bwd_point(m, lane, maps, rays, dirs, light_idx, vis, indirect, env, weight_d, D, n_lights, equal_area,
                        use_srgb, acc_thres, g_out, g_maps, g_env, lds_env);
    if (lds_env) {
        __syncthreads();
        for (int i = threadIdx.x; i < n_env; i += blockDim.x) {
            const float v = lds_env[i];
            if (v != 0.0f) atomic_add_f32(g_env + i, v);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// SG environment backward: one block per SG, threads over (light, direction) pairs.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_env_sg_bwd(TirEnvSG e, const float* __restrict__ dirs, int D, const float* __restrict__ g_env, float* __restrict__ g_sgs) {
    const int k = blockIdx.x;
    const float* s = e.sgs + 7 * k;
    const float nrm = sqrtf(s[0] * s[0] + s[1] * s[1] + s[2] * s[2]);
    const float u0 = s[0] / nrm, u1 = s[1] / nrm, u2 = s[2] / nrm;
    const float lam = fabsf(s[3]);
    const float sl = (s[3] > 0.f) ? 1.f : ((s[3] < 0.f) ? -1.f : 0.f);
    float acc[7] = {0, 0, 0, 0, 0, 0, 0};
    for (int ld = threadIdx.x; ld < e.n_lights * D; ld += blockDim.x) {
        const int l = ld / D, d = ld % D;
        const float v0 = dirs[3 * d], v1 = dirs[3 * d + 1], v2 = dirs[3 * d + 2];
        const float* R = e.rot + 9 * l;
        const float r0 = v0 * R[0] + v1 * R[3] + v2 * R[6];
        const float r1 = v0 * R[1] + v1 * R[4] + v2 * R[7];
        const float r2 = v0 * R[2] + v1 * R[5] + v2 * R[8];
        const float dot = r0 * u0 + r1 * u1 + r2 * u2;
        const float ex = expf(lam * (dot - 1.0f));
        const float* go = g_env + 3 * (size_t)ld;
        float G = 0.f;
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float mu = s[4 + q];
            const float sm = (mu > 0.f) ? 1.f : ((mu < 0.f) ? -1.f : 0.f);
            acc[4 + q] += sm * ex * go[q];
            G += fabsf(mu) * go[q];
        }
        const float gex = G * ex;
        acc[3] += sl * (dot - 1.0f) * gex;
        const float gdot = lam * gex;
        // dot = r . lobe / |lobe|
        acc[0] += gdot * (r0 - u0 * dot) / nrm; acc[1] += gdot * (r1 - u1 * dot) / nrm; acc[2] += gdot * (r2 - u2 * dot) / nrm;
    }
    __shared__ float red[4][7];
#pragma unroll
    for (int q = 0; q < 7; ++q) acc[q] = group_sum<64>(acc[q]);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int q = 0; q < 7; ++q) red[threadIdx.x >> 6][q] = acc[q];
    __syncthreads();
    if (threadIdx.x < 7) g_sgs[7 * k + threadIdx.x] += red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

int check_grad_field(const TirField* f, const TirFieldGrad* g, bool density, bool app) {
    if (!f || !g) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i) {
        if (f->grid[i] < 2) return TIR_ERR_ARG;
        if (density && (!f->dplane[i] || !f->dline[i] || !g->dplane[i] || !g->dline[i])) return TIR_ERR_ARG;
        if (app && (!f->aplane[i] || !f->aline[i] || !g->aplane[i] || !g->aline[i])) return TIR_ERR_ARG;
    }
    // the kernels index planes with 32-bit float offsets
    const int64_t g0 = f->grid[0], g1 = f->grid[1], g2 = f->grid[2];
    int64_t cells = g0 * g1;
    if (g0 * g2 > cells) cells = g0 * g2;
    if (g1 * g2 > cells) cells = g1 * g2;
    if (density && cells * f->n_dcomp >= (1ll << 31)) return TIR_ERR_UNSUPPORTED;
    if (app && cells * f->n_acomp >= (1ll << 30)) return TIR_ERR_UNSUPPORTED;      // 32-bit BYTE offsets
    if (!tir_occ_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    return TIR_OK;
}

}  // namespace

extern "C" int tir_march_primary_bwd(const TirField* f, const TirFieldGrad* g, const float* rays,
                                     const float* ray_jitter, int32_t B, int32_t S, const float* sigma,
                                     const float* weight, const float* g_weight, const float* g_acc,
                                     const float* g_depth, float* g_feature, void* stream) {
    int rc = check_grad_field(f, g, true, false);
    if (rc) return rc;
    if (B < 0 || S <= 0) return TIR_ERR_ARG;
    if (S > 64 * TIR_MAX_CHUNKS) return TIR_ERR_UNSUPPORTED;
    if (B == 0) return TIR_OK;
    if (!rays || !sigma || !weight || !g_weight || !g_acc || !g_depth) return TIR_ERR_ARG;
    hipStream_t s = tir_stream(stream);
    const size_t line_bytes = (size_t)(f->grid[0] + f->grid[1] + f->grid[2]) * f->n_dcomp * sizeof(float);
    const bool ll = line_bytes <= 96 * 1024;          // density-line gradients block-local in LDS when they fit
    const size_t lds = ll ? line_bytes : 0;
    int blocks = (B + 3) / 4;
    if (ll && blocks > 512) blocks = 512;             // persistent blocks amortise the LDS zero / flush
    dim3 grid(blocks), blk(256);
#define TIR_LAUNCH_MB(C4)                                                                                             \
    do {                                                                                                              \
        if (ll) {                                                                                                     \
            if (int rc_ = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_march_primary_bwd<C4, true>), 96 * 1024)) return rc_; \
            hipLaunchKernelGGL((k_march_primary_bwd<C4, true>), grid, blk, lds, s, *f, *g, rays, ray_jitter, B, S, sigma, weight, g_weight, g_acc, g_depth, g_feature); \
        } else                                                                                                        \
            hipLaunchKernelGGL((k_march_primary_bwd<C4, false>), grid, blk, 0, s, *f, *g, rays, ray_jitter, B, S, sigma, weight, g_weight, g_acc, g_depth, g_feature); \
    } while (0)
    switch (f->n_dcomp) {
        case 16: TIR_LAUNCH_MB(4); break;
        case 8:  TIR_LAUNCH_MB(2); break;
        case 32: TIR_LAUNCH_MB(8); break;
        case 4:  TIR_LAUNCH_MB(1); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
#undef TIR_LAUNCH_MB
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_density_grad_bwd(const TirField* f, const TirFieldGrad* g, const float* xyz,
                                    const float* g_normal, int64_t n, void* stream) {
    int rc = check_grad_field(f, g, true, false);
    if (rc) return rc;
    if (n < 0 || (n > 0 && (!xyz || !g_normal))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipStream_t s = tir_stream(stream);
    const size_t line_bytes = (size_t)(f->grid[0] + f->grid[1] + f->grid[2]) * f->n_dcomp * sizeof(float);
    const bool ll = line_bytes <= 96 * 1024;
    const size_t lds = ll ? line_bytes : 0;
    const int lps = f->n_dcomp < 16 ? f->n_dcomp : 16;       // lanes per sample, 8 consecutive samples per lane group
    int64_t blocks = ((n + 7) / 8 * lps + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    dim3 grid((unsigned)blocks), blk(256);
#define TIR_LAUNCH_DG(C4)                                                                                             \
    do {                                                                                                              \
        if (ll) {                                                                                                     \
            if (int rc_ = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_density_grad_bwd<C4, true>), 96 * 1024)) return rc_; \
            hipLaunchKernelGGL((k_density_grad_bwd<C4, true>), grid, blk, lds, s, *f, *g, xyz, g_normal, n);          \
        } else                                                                                                        \
            hipLaunchKernelGGL((k_density_grad_bwd<C4, false>), grid, blk, 0, s, *f, *g, xyz, g_normal, n);           \
    } while (0)
    switch (f->n_dcomp) {
        case 16: TIR_LAUNCH_DG(4); break;
        case 8:  TIR_LAUNCH_DG(2); break;
        case 32: TIR_LAUNCH_DG(8); break;
        case 4:  TIR_LAUNCH_DG(1); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
#undef TIR_LAUNCH_DG
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_composite_primary_bwd(const float* rays, const int32_t* offsets, const int32_t* rec_k,
                                         const float* rec_w, const float* rgb, const float* brdf,
                                         const float* brdf_jit, const float* pred_normal,
                                         const float* derived_normal, const float* acc, const float* depth,
                                         int32_t B, int32_t S, int32_t white_bg, int32_t is_relight,
                                         float fixed_fresnel, const float* g_maps, float* g_rgb, float* g_brdf,
                                         float* g_brdf_jit, float* g_pred, float* g_der, float* g_weight,
                                         float* g_acc, float* g_depth, void* stream) {
    if (B < 0 || S <= 0) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !offsets || !acc || !depth || !g_maps || !g_weight || !g_acc || !g_depth) return TIR_ERR_ARG;
    if (rgb && !g_rgb) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_composite_primary_bwd, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), rays, offsets,
                       rec_k, rec_w, rgb, brdf, brdf_jit, pred_normal, derived_normal, acc, depth, B, S, white_bg,
                       is_relight, fixed_fresnel, g_maps, g_rgb, g_brdf, g_brdf_jit, g_pred, g_der, g_weight, g_acc,
                       g_depth);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

template <int C4>
static int launch_app_bwd(const TirField* f, const TirFieldGrad* g, const float* xyz, const int32_t* li,
                          const int32_t* map, const float* g_rad, const float* g_int, int stride, int64_t n,
                          float* y_rad, float* y_int, hipStream_t s) {
    constexpr int CA = C4 * 4;
    const int nl = f->n_lights < TIR_APP_MAX_L ? f->n_lights : TIR_APP_MAX_L;
    const size_t base = (size_t)(2 * (nl + 1) * 3 * CA) * sizeof(float);
    int rmax = f->grid[0] > f->grid[1] ? f->grid[0] : f->grid[1];
    if (f->grid[2] > rmax) rmax = f->grid[2];
    const size_t line = (size_t)rmax * CA * sizeof(float);
    const int line_lds = base + line <= 160 * 1024 ? 1 : 0;     // one line's gradient block-local in LDS when it fits
    const size_t lds = base + (line_lds ? line : 0);
    // dY = g_feat . basis_mat into the y buffers
    int64_t dyb = (n + 127) / 128;                              // 4 waves x 32 samples per workgroup and pass; persistent beyond 3 per CU
    if (dyb > 768) dyb = 768;
    if (g_rad) hipLaunchKernelGGL((k_app_dy<C4>), dim3((unsigned)dyb), dim3(256), 0, s, f->basis_t, f->app_dim, g_rad, stride, n, y_rad);
    if (g_int) hipLaunchKernelGGL((k_app_dy<C4>), dim3((unsigned)dyb), dim3(256), 0, s, f->basis_t, f->app_dim, g_int, stride, n, y_int);
    const int threads = TIR_APP_BWD_THREADS;
    int64_t blocks = (n * 2 + threads - 1) / threads;            // 32 samples per wave chunk
    const int64_t per_cu = (lds <= 80 * 1024) ? 2 : 1;           // persistent blocks amortise the LDS zero / flush
    if (blocks > 256 * per_cu) blocks = 256 * per_cu;
    dim3 grid((unsigned)blocks), blk(threads);
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_bwd<C4, true, true>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_bwd<C4, true, false>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_bwd<C4, false, true>), 160 * 1024)) return rc;
    if (g_rad && g_int) hipLaunchKernelGGL((k_vm_app_bwd<C4, true, true>), grid, blk, lds, s, *f, *g, xyz, li, map, n, y_rad, y_int, line_lds);
    else if (g_rad)     hipLaunchKernelGGL((k_vm_app_bwd<C4, true, false>), grid, blk, lds, s, *f, *g, xyz, li, map, n, y_rad, y_int, line_lds);
    else                hipLaunchKernelGGL((k_vm_app_bwd<C4, false, true>), grid, blk, lds, s, *f, *g, xyz, li, map, n, y_rad, y_int, line_lds);
    return TIR_OK;
}

extern "C" int tir_vm_app_bwd(const TirField* f, const TirFieldGrad* g, const float* xyz,
                              const int32_t* light_idx, const int32_t* idx_map, const float* g_rad,
                              const float* g_int, int32_t stride, int64_t n, float* y_rad, float* y_int,
                              void* stream) {
    int rc = check_grad_field(f, g, false, true);
    if (rc) return rc;
    if (!f->basis_t || !f->light_mean || !f->light_line || !g->light_line || !g->light_mean) return TIR_ERR_ARG;
    if (f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (stride < f->app_dim) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz) || (!g_rad && !g_int) || (g_rad && !light_idx)) return TIR_ERR_ARG;
    if ((g_rad && !y_rad) || (g_int && !y_int)) return TIR_ERR_ARG;      // the y buffers double as the dY workspace
    if (n == 0) return TIR_OK;
    hipStream_t s = tir_stream(stream);
    switch (f->n_acomp) {
        case 48: rc = launch_app_bwd<12>(f, g, xyz, light_idx, idx_map, g_rad, g_int, stride, n, y_rad, y_int, s); break;
        case 24: rc = launch_app_bwd<6>(f, g, xyz, light_idx, idx_map, g_rad, g_int, stride, n, y_rad, y_int, s); break;
        case 16: rc = launch_app_bwd<4>(f, g, xyz, light_idx, idx_map, g_rad, g_int, stride, n, y_rad, y_int, s); break;
        case 96: rc = launch_app_bwd<24>(f, g, xyz, light_idx, idx_map, g_rad, g_int, stride, n, y_rad, y_int, s); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

static int gemm_tn_launch(bool bf16, const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t N,
                          int32_t ones_col, int64_t n, float* C, int32_t ldc, float* bias_out, void* stream) {
    if (!A || !B || !C || M <= 0 || N <= 0 || n < 0) return TIR_ERR_ARG;
    ones_col = (ones_col || bias_out) ? 1 : 0;
    if (M > 128 || N + ones_col > 160) return TIR_ERR_UNSUPPORTED;
    if ((lda & 3) || (ldb & 3) || lda < ((M + 3) & ~3) || ldb < ((N + 3) & ~3) || ldc < N + (bias_out ? 0 : ones_col)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    int64_t chunk = (n + 511) / 512;
    chunk = (chunk + 31) / 32 * 32;
    if (chunk < 128) chunk = 128;
    const unsigned blocks = (unsigned)((n + chunk - 1) / chunk);
    const int mt = (M + 31) / 32, tiles = mt * ((N + ones_col + 31) / 32);
    hipStream_t s = tir_stream(stream);
#define TIR_GEMM_ARGS A, lda, M, B, ldb, N, ones_col, n, C, ldc, chunk, bias_out
    if (bf16) {
        if (tiles <= 4)      hipLaunchKernelGGL((k_gemm_tn_bf16<1, false>), dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
        else if (tiles <= 8) hipLaunchKernelGGL((k_gemm_tn_bf16<2, false>), dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
        else if (mt == 4)    hipLaunchKernelGGL((k_gemm_tn_bf16<5, true>), dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
        else                 hipLaunchKernelGGL((k_gemm_tn_bf16<5, false>), dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
    } else {
        if (tiles <= 4)      hipLaunchKernelGGL(k_gemm_tn<1>, dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
        else if (tiles <= 8) hipLaunchKernelGGL(k_gemm_tn<2>, dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
        else                 hipLaunchKernelGGL(k_gemm_tn<5>, dim3(blocks), dim3(256), 0, s, TIR_GEMM_ARGS);
    }
#undef TIR_GEMM_ARGS
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_gemm_tn(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t N,
                           int32_t ones_col, int64_t n, float* C, int32_t ldc, float* bias_out, void* stream) {
    return gemm_tn_launch(false, A, lda, M, B, ldb, N, ones_col, n, C, ldc, bias_out, stream);
}

extern "C" int tir_gemm_tn_bf16x3(const float* A, int32_t lda, int32_t M, const float* B, int32_t ldb, int32_t N,
                                  int32_t ones_col, int64_t n, float* C, int32_t ldc, float* bias_out, void* stream) {
    return gemm_tn_launch(true, A, lda, M, B, ldb, N, ones_col, n, C, ldc, bias_out, stream);
}

extern "C" int tir_shade_integrate_bwd(const float* maps, const float* rays, const float* dirs,
                                       const int32_t* light_idx, const float* vis, const float* indirect,
                                       const float* env, const float* weight_d, int32_t M, int32_t D,
                                       int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                                       const float* g_out, float* g_maps, float* g_env, void* stream) {
    if (M < 0 || D <= 0 || n_lights <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!maps || !rays || !dirs || !vis || !env || !g_out || !g_maps || (!equal_area && !weight_d)) return TIR_ERR_ARG;
    const size_t env_bytes = (size_t)n_lights * D * 3 * sizeof(float);
    const int env_in_lds = env_bytes <= 64 * 1024 ? 1 : 0;
    int blocks = (M + 3) / 4;
    if (env_in_lds && blocks > 512) blocks = 512;             // persistent: one LDS flush per block
    hipLaunchKernelGGL(k_shade_integrate_bwd, dim3(blocks), dim3(256), env_in_lds ? env_bytes : 0, tir_stream(stream),
                       maps, rays, dirs, light_idx, vis, indirect, env, weight_d, M, D, n_lights, equal_area, use_srgb,
                       acc_thres, g_out, g_maps, g_env, env_in_lds);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_sg_bwd(const TirEnvSG* e, const float* dirs, int32_t D, const float* g_env, float* g_sgs,
                              void* stream) {
    if (!e || !e->sgs || !e->rot || e->n_sg <= 0 || e->n_lights <= 0 || D < 0) return TIR_ERR_ARG;
    if (D == 0) return TIR_OK;
    if (!dirs || !g_env || !g_sgs) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_env_sg_bwd, dim3(e->n_sg), dim3(256), 0, tir_stream(stream), *e, dirs, D, g_env, g_sgs);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// Adam over a list of tensors in ONE launch (the caller of the training step: optimizer.step(), train_tensoIR.py:317,
// torch.optim.Adam(grad_vars, betas=(0.9, 0.99)), :197).  The framework's multi-tensor Adam is ~90 launches of ~9 us per
// step over these ~35 tensors (0.75 ms of a 6.5 ms step); one pass over parameter, gradient and both moments is 28 B per
// element -- 0.08 ms for the 17.4 M parameters of the 300^3 field at HBM speed.  Same arithmetic as torch's default
// (non-amsgrad, no weight decay) update, element for element:
//   m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;  p -= (lr / bc1) * m / (sqrt(v) / sqrt(bc2) + eps).
// Operands are raw storage: all four tensors of an entry must be dense with the same element order.
// ------------------------------------------------------------------------------------------------
#define ADAM_MAX_T 40
#define ADAM_CHUNK 8192
struct AdamEntry { float* p; const float* g; float* m; float* v; long long n; float step_size, inv_bc2_sqrt; int first_chunk; };
struct AdamTable { AdamEntry e[ADAM_MAX_T]; int n_tensors; };

__global__ void __launch_bounds__(256)
k_adam(AdamTable tab, float b1, float b2, float eps) {
    __shared__ int t_sh;
    if (threadIdx.x == 0) {
        int t = 0;
        while (t + 1 < tab.n_tensors && (int)blockIdx.x >= tab.e[t + 1].first_chunk) ++t;
        t_sh = t;
    }
    __syncthreads();
    const AdamEntry& e = tab.e[t_sh];
    const long long base = (long long)((int)blockIdx.x - e.first_chunk) * ADAM_CHUNK;
    const long long end = min(e.n, base + ADAM_CHUNK);
    const float w1 = 1.0f - b1, w2 = 1.0f - b2;
    auto upd = [&](float& p, float g, float& m, float& v) {
        m = m + (g - m) * w1;
        v = v * b2 + w2 * g * g;
        const float denom = sqrtf(v) * e.inv_bc2_sqrt + eps;
        p = p - e.step_size * (m / denom);
    };
    const bool vec = ((((size_t)e.p | (size_t)e.g | (size_t)e.m | (size_t)e.v) & 15) == 0);
    if (vec) {
        for (long long i = base + 4 * (long long)threadIdx.x; i < end; i += 4 * 256) {
            if (i + 4 <= end) {
                float4 p = *reinterpret_cast<float4*>(e.p + i), m = *reinterpret_cast<float4*>(e.m + i), v = *reinterpret_cast<float4*>(e.v + i);
                const float4 g = *reinterpret_cast<const float4*>(e.g + i);
                upd(p.x, g.x, m.x, v.x); upd(p.y, g.y, m.y, v.y); upd(p.z, g.z, m.z, v.z); upd(p.w, g.w, m.w, v.w);
                *reinterpret_cast<float4*>(e.p + i) = p; *reinterpret_cast<float4*>(e.m + i) = m; *reinterpret_cast<float4*>(e.v + i) = v;
            } else {
                for (long long j = i; j < end; ++j) upd(e.p[j], e.g[j], e.m[j], e.v[j]);
            }
        }
    } else {
        for (long long i = base + threadIdx.x; i < end; i += 256) upd(e.p[i], e.g[i], e.m[i], e.v[i]);
    }
}

extern "C" int tir_adam_step(int32_t n_tensors, float* const* p, const float* const* g, float* const* m, float* const* v,
                             const int64_t* count, const float* lr, const float* bias_correction1,
                             const float* bias_correction2, float beta1, float beta2, float eps, void* stream) {
    if (n_tensors < 0 || (n_tensors > 0 && (!p || !g || !m || !v || !count || !lr || !bias_correction1 || !bias_correction2)))
        return TIR_ERR_ARG;
    hipStream_t s = tir_stream(stream);
    int t = 0;
    while (t < n_tensors) {
        AdamTable tab;
        int k = 0, chunks = 0;
        for (; t < n_tensors && k < ADAM_MAX_T; ++t) {
            if (count[t] < 0 || !(bias_correction1[t] > 0.0f) || !(bias_correction2[t] > 0.0f)) return TIR_ERR_ARG;
            if (count[t] == 0) continue;
            if (!p[t] || !g[t] || !m[t] || !v[t]) return TIR_ERR_ARG;
            const int64_t c = (count[t] + ADAM_CHUNK - 1) / ADAM_CHUNK;
            if (chunks + c > (1 << 30)) break;
            tab.e[k] = AdamEntry{p[t], g[t], m[t], v[t], (long long)count[t], lr[t] / bias_correction1[t],
                                 1.0f / sqrtf(bias_correction2[t]), chunks};
            chunks += (int)c;
            ++k;
        }
        tab.n_tensors = k;
        if (k == 0) {                   // nothing fitted: either only empty tensors were left, or ONE tensor needs more
            if (t < n_tensors) return TIR_ERR_ARG;   // than 2^30 chunks (it can never fit a launch: do not spin on it)
            continue;
        }
        hipLaunchKernelGGL(k_adam, dim3((unsigned)chunks), dim3(256), 0, s, tab, beta1, beta2, eps);
        TIR_CHECK_LAUNCH();
    }
    return TIR_OK;
}

extern "C" int tir_gemm_tn_small_bf16x3(const float* const* As, int32_t lda, int32_t M, const float* const* Bs, int32_t ldb,
                                        int32_t N, int32_t n_jobs, int64_t n, float* C, int32_t ldc, void* stream) {
    if (n_jobs < 1 || n_jobs > 3 || !As || !Bs || !C || n < 0 || M < 1 || M > 32 || N < 1 || N > 160 || lda < M || ldb < N || ldc < N)
        return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    TirSmallGemmJobs jobs;
    jobs.n_jobs = n_jobs;
    for (int i = 0; i < n_jobs; ++i) {
        if (!As[i] || !Bs[i]) return TIR_ERR_ARG;
        jobs.j[i] = TirSmallGemmJob{As[i], Bs[i]};
    }
    int per = 256 / n_jobs;
    const int64_t steps = (n + 63) / 64;                     // a workgroup's four waves take one 16-row step each
    if (steps < per) per = (int)steps;
    const dim3 g((unsigned)(per * n_jobs)), b(256);
    hipStream_t s = tir_stream(stream);
    switch ((N + 31) / 32) {
        case 1: hipLaunchKernelGGL(k_gemm_tn_small<1>, g, b, 0, s, jobs, lda, M, ldb, N, n, C, ldc); break;
        case 2: hipLaunchKernelGGL(k_gemm_tn_small<2>, g, b, 0, s, jobs, lda, M, ldb, N, n, C, ldc); break;
        case 3: hipLaunchKernelGGL(k_gemm_tn_small<3>, g, b, 0, s, jobs, lda, M, ldb, N, n, C, ldc); break;
        case 4: hipLaunchKernelGGL(k_gemm_tn_small<4>, g, b, 0, s, jobs, lda, M, ldb, N, n, C, ldc); break;
        default: hipLaunchKernelGGL(k_gemm_tn_small<5>, g, b, 0, s, jobs, lda, M, ldb, N, n, C, ldc); break;
    }
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
