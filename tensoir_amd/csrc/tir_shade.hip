// K8: environment light (spherical Gaussians), GGX specular, rendering-equation integration,
// importance-sampled HDR relighting.
#include "tir_common.hpp"

using namespace tir;

namespace {

// F.normalize(x, dim=-1): x / max(||x||, 1e-12)
__device__ __forceinline__ void normalize3(float& x, float& y, float& z, float eps) {
    float n = fmaxf(sqrtf(x * x + y * y + z * z), eps);
    x /= n; y /= n; z /= n;
}

struct Surface {     // per-surface-point quantities of GGX_specular (models/relight_utils.py:22-37)
    float N[3], V[3];
    float NoV;
    float alpha2[3], k[3];
    float fres[3];
    float alb_pi[3];
};

__device__ __forceinline__ Surface make_surface(const float* normal, const float* view, const float* rough3,
                                                const float* fres3, const float* albedo) {
    Surface s;
    s.N[0] = normal[0]; s.N[1] = normal[1]; s.N[2] = normal[2];
    s.V[0] = view[0]; s.V[1] = view[1]; s.V[2] = view[2];
    normalize3(s.V[0], s.V[1], s.V[2], 1e-12f);
    normalize3(s.N[0], s.N[1], s.N[2], 1e-12f);
    float nov = s.V[0] * s.N[0] + s.V[1] * s.N[1] + s.V[2] * s.N[2];
    float sg = (nov > 0.f) ? 1.f : ((nov < 0.f) ? -1.f : 0.f);           // N = N * NoV.sign()  (:30)
    s.N[0] *= sg; s.N[1] *= sg; s.N[2] *= sg;
    nov = s.N[0] * s.V[0] + s.N[1] * s.V[1] + s.N[2] * s.V[2];
    s.NoV = fminf(fmaxf(nov, 1e-6f), 1.f);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float r = rough3[c];
        float a = r * r;
        s.alpha2[c] = a * a;
        s.k[c] = (a + 2.f * r + 1.0f) / 8.0f;
        s.fres[c] = fres3[c];
        s.alb_pi[c] = albedo ? albedo[c] / 3.14159265358979323846f : 0.f;
    }
    return s;
}

// specular BRDF for one light direction (un-normalised l allowed), three channels (:22-49)
__device__ __forceinline__ void ggx_dir(const Surface& s, float lx, float ly, float lz, float spec[3]) {
    normalize3(lx, ly, lz, 1e-12f);
    float hx = (lx + s.V[0]) / 2.0f, hy = (ly + s.V[1]) / 2.0f, hz = (lz + s.V[2]) / 2.0f;
    normalize3(hx, hy, hz, 1e-12f);
    float NoL = fminf(fmaxf(s.N[0] * lx + s.N[1] * ly + s.N[2] * lz, 1e-6f), 1.f);
    float NoH = fminf(fmaxf(s.N[0] * hx + s.N[1] * hy + s.N[2] * hz, 1e-6f), 1.f);
    float VoH = fminf(fmaxf(s.V[0] * hx + s.V[1] * hy + s.V[2] * hz, 1e-6f), 1.f);
    float FMi = ((-5.55473f) * VoH - 6.98316f) * VoH;
    float p2 = exp2f(FMi);
    const float four_pi = 4.0f * 3.14159265358979323846f;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float frac0 = s.fres[c] + (1.f - s.fres[c]) * p2;
        float frac = frac0 * s.alpha2[c];
        float nom0 = NoH * NoH * (s.alpha2[c] - 1.f) + 1.f;
        float nom1 = s.NoV * (1.f - s.k[c]) + s.k[c];
        float nom2 = NoL * (1.f - s.k[c]) + s.k[c];
        float nom = fminf(fmaxf(four_pi * nom0 * nom0 * nom1 * nom2, 1e-6f), four_pi);
        spec[c] = frac / nom;
    }
}

// ---- a15: SG environment radiance per (light rotation, direction) ------------------------------
__global__ void __launch_bounds__(128)
k_env_sg(TirEnvSG e, const float* __restrict__ dirs, int D, float* __restrict__ out) {
    // one block of 128 threads per (light, dir): thread k evaluates SG k, block-reduces
    const int ld = blockIdx.x;
    const int l = ld / D, d = ld % D;
    const float v0 = dirs[3 * d], v1 = dirs[3 * d + 1], v2 = dirs[3 * d + 2];
    const float* R = e.rot + 9 * l;
    // torch.matmul(dirs[1,D,3], R[L,3,3]): row-vector times matrix  (models/tensorBase_rotated_lights.py:586)
    const float r0 = v0 * R[0] + v1 * R[3] + v2 * R[6];
    const float r1 = v0 * R[1] + v1 * R[4] + v2 * R[7];
    const float r2 = v0 * R[2] + v1 * R[5] + v2 * R[8];
    float c0 = 0.f, c1 = 0.f, c2 = 0.f;
    for (int k = threadIdx.x; k < e.n_sg; k += blockDim.x) {
        const float* g = e.sgs + 7 * k;
        float nrm = sqrtf(g[0] * g[0] + g[1] * g[1] + g[2] * g[2]);
        float dot = r0 * (g[0] / nrm) + r1 * (g[1] / nrm) + r2 * (g[2] / nrm);
        float ex = expf(fabsf(g[3]) * (dot - 1.0f));
        c0 += fabsf(g[4]) * ex; c1 += fabsf(g[5]) * ex; c2 += fabsf(g[6]) * ex;
    }
    __shared__ float red[3][2];
    c0 = group_sum<64>(c0); c1 = group_sum<64>(c1); c2 = group_sum<64>(c2);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][wv] = c0; red[1][wv] = c1; red[2][wv] = c2; }
    __syncthreads();
    if (threadIdx.x == 0) {
        float* o = out + 3 * (size_t)ld;
        o[0] = red[0][0] + red[0][1]; o[1] = red[1][0] + red[1][1]; o[2] = red[2][0] + red[2][1];
    }
}

// ---- render_with_BRDF geometry (models/relight_utils.py:417-435) -----------------------------------
__global__ void __launch_bounds__(1024)
k_shade_setup(const float* __restrict__ maps, const float* __restrict__ rays, const float* __restrict__ dirs,
              int M, int D, float acc_thres, float* __restrict__ surf, uint8_t* __restrict__ active,
              int32_t* __restrict__ pair_ids, int32_t* __restrict__ n_active, float* __restrict__ vis,
              int32_t* __restrict__ rec_cnt, int dir_major) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const bool in = t < (int64_t)M * D;
    // thread order = order of the compacted pair list.  dir_major: consecutive threads are consecutive surface points
    // (neighbouring pixels) with the SAME light direction, so the 32 rays a march block takes are a bundle of nearly
    // parallel rays from neighbouring points: they walk neighbouring plane cells in step (L1 / L2 hits) and end
    // after similar step counts.  Point-major order hands a block one point's whole fan of directions.
    int64_t i = t;
    if (dir_major && in) i = (t % M) * D + t / M;
    bool act = false;
    if (in) {
        const int m = (int)(i / D), d = (int)(i % D);
        const float* mp = maps + (size_t)m * TIR_MAP_STRIDE;
        const float* r = rays + 6 * (size_t)m;
        if (d == 0) {
            const float depth = mp[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) surf[3 * (size_t)m + a] = add_rn(r[a], mul_rn(depth, r[3 + a]));   // :422
        }
        // cosine = clamp(einsum(surf2l, normal_map), 0); mask = cosine > 1e-6   (:433-435)
        float cs = dirs[3 * d] * mp[4] + dirs[3 * d + 1] * mp[5] + dirs[3 * d + 2] * mp[6];
        act = fmaxf(cs, 0.f) > 1e-6f && mp[14] > acc_thres;       // acc_mask = acc > 0.5 (:1031, renderer.py:86)
        active[i] = act ? 1 : 0;
    }
    if (!pair_ids) return;
    // compacted list of the (surface point, direction) pairs that get a secondary ray -- the boolean-mask indexing
    // surf2l[cosine_mask] of the reference (:440-441) -- so that the march kernel spends no half-wave on a masked pair;
    // masked pairs get their zero visibility / empty record range here (:437-438)
    // one reservation per 1024-pair block (not per wave: 8192 atomics on one address cost 100 us), wave offsets by a
    // small scan of the per-wave counts in LDS
    __shared__ int s_wcnt[16];
    __shared__ int s_base;
    const unsigned long long mask = __ballot(act);
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if (lane == 0) s_wcnt[wv] = __popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
        for (int q = 0; q < nw; ++q) { const int c = s_wcnt[q]; s_wcnt[q] = tot; tot += c; }
        s_base = tot ? atomicAdd(n_active, tot) : 0;
    }
    __syncthreads();
    const int base = s_base + s_wcnt[wv];
    if (act) {
        // the shared pair counter must have been zero at entry (re-armed by the integration kernel); a stale count left by
        // an interrupted earlier call must never turn into a write past the M*D slots
        const int64_t slot = (int64_t)base + __popcll(mask & ((1ull << lane) - 1ull));
        if (slot < (int64_t)M * D) pair_ids[slot] = (int32_t)i;
    } else if (in) {
        if (vis) vis[i] = 0.0f;
        if (rec_cnt) rec_cnt[i] = 0;
    }
}

// ---- K8: one wave per surface point, lanes over light directions -----------------------------------
__global__ void __launch_bounds__(256)
k_shade_integrate(const float* __restrict__ maps, const float* __restrict__ rays, const float* __restrict__ dirs,
                  const int32_t* __restrict__ light_idx, const float* __restrict__ vis,
                  const float* __restrict__ indirect, const float* __restrict__ env,
                  const float* __restrict__ weight_d, int M, int D, int n_lights, int equal_area, int use_srgb,
                  float acc_thres, float* __restrict__ out, const int32_t* __restrict__ rec_off,
                  const int32_t* __restrict__ rec_cnt, const float* __restrict__ rec_w, const float* __restrict__ rec_rgb,
                  int32_t* __restrict__ reset_counter) {
    const int lane = threadIdx.x & 63;
    // the last consumer of a step re-arms the caller's pair counter for the next step (also under HIP-graph replay)
    if (reset_counter && blockIdx.x == 0 && threadIdx.x == 0) *reset_counter = 0;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    const float* mp = maps + (size_t)m * TIR_MAP_STRIDE;
    if (!(mp[14] > acc_thres)) {              // background ray: white (renderer.py:105-106)
        if (lane < 3) out[3 * (size_t)m + lane] = 1.0f;
        return;
    }
    const float* r = rays + 6 * (size_t)m;
    float view[3] = {-r[3], -r[4], -r[5]};
    normalize3(view[0], view[1], view[2], 1e-6f);                   // safe_l2_normalize(-rays_d)  (:429-430)
    const float rough3[3] = {mp[10], mp[10], mp[10]};                // roughness.repeat(1,3)  (renderer.py:91)
    Surface s = make_surface(mp + 4, view, rough3, mp + 11, mp + 7);
    int li = light_idx ? light_idx[m] : 0;
    li = min(max(li, 0), n_lights - 1);
    const float* envl = env + (size_t)li * D * 3;
    float c[3] = {0.f, 0.f, 0.f};
    for (int d = lane; d < D; d += 64) {
        const float lx = dirs[3 * d], ly = dirs[3 * d + 1], lz = dirs[3 * d + 2];
        const float cosine = fmaxf(lx * mp[4] + ly * mp[5] + lz * mp[6], 0.f);
        float spec[3];
        ggx_dir(s, lx, ly, lz, spec);
        const size_t md = (size_t)m * D + d;
        const float v = vis[md];
        const float wd = equal_area ? 1.0f : weight_d[d];
        float ind[3] = {0.f, 0.f, 0.f};
        if (indirect) { ind[0] = indirect[3 * md]; ind[1] = indirect[3 * md + 1]; ind[2] = indirect[3 * md + 2]; }
        else if (rec_off) {        // indirect radiance straight from the ray's records, in sample order (models/relight_utils.py:832)
            const int b = rec_off[md], e = b + rec_cnt[md];
            for (int i = b; i < e; ++i) {
                const float w = rec_w[i];
                ind[0] = fmaf(w, rec_rgb[3 * (size_t)i], ind[0]);
                ind[1] = fmaf(w, rec_rgb[3 * (size_t)i + 1], ind[1]);
                ind[2] = fmaf(w, rec_rgb[3 * (size_t)i + 2], ind[2]);
            }
        }
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            float light = v * envl[3 * d + q] + ind[q];                                       // :462
            float brdf = s.alb_pi[q] + spec[q];                                               // :455
            if (equal_area) c[q] += 4.0f * 3.14159265358979323846f * brdf * light * cosine;   // :470-471
            else c[q] += brdf * light * cosine * wd;                                          // :474-475
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float t = group_sum<64>(c[q]);
        if (equal_area) t /= (float)D;
        t = fminf(fmaxf(t, 0.f), 1.f);                                                        // :477
        if (use_srgb) t = linear2srgb(t);
        if (lane == 0) out[3 * (size_t)m + q] = t;
    }
}

// ---- K9: importance-sampled relighting (scripts/relight_importance.py:154-170) ----------------------
__global__ void __launch_bounds__(256)
k_relight_importance(const float* __restrict__ normal, const float* __restrict__ albedo,
                     const float* __restrict__ rough, const float* __restrict__ fresnel,
                     const float* __restrict__ rays_d, const float* __restrict__ ldir,
                     const float* __restrict__ lrgb, const float* __restrict__ lpdf,
                     const float* __restrict__ vis, int M, int Ns, float* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m >= M) return;
    float view[3] = {-rays_d[3 * (size_t)m], -rays_d[3 * (size_t)m + 1], -rays_d[3 * (size_t)m + 2]};
    normalize3(view[0], view[1], view[2], 1e-6f);
    const float rough3[3] = {rough[m], rough[m], rough[m]};          // [M,1] broadcast against [M,3]
    const float* nm = normal + 3 * (size_t)m;
    Surface s = make_surface(nm, view, rough3, fresnel + 3 * (size_t)m, albedo + 3 * (size_t)m);
    float c[3] = {0.f, 0.f, 0.f};
    for (int j = lane; j < Ns; j += 64) {
        const size_t mj = (size_t)m * Ns + j;
        const float lx = ldir[3 * mj], ly = ldir[3 * mj + 1], lz = ldir[3 * mj + 2];
        const float cosine = lx * nm[0] + ly * nm[1] + lz * nm[2];   // not clamped (:125); vis is 0 below 1e-6
        float spec[3];
        ggx_dir(s, lx, ly, lz, spec);
        const float v = vis[mj], pdf = lpdf[mj];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            c[q] += (s.alb_pi[q] + spec[q]) * (v * lrgb[3 * mj + q]) * cosine / pdf;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float t = group_sum<64>(c[q]) / (float)Ns;
        t = linear2srgb(fminf(fmaxf(t, 0.f), 1.f));
        if (lane == 0) out[3 * (size_t)m + q] = t;
    }
}

// ---- K9 on the device: importance sampler, cell-indexed integration, background lookup ---------------------------
// Environment_Light.sample_light (models/relight_utils.py:150-188) draws torch.multinomial over the H*W cells of
// pdf ~ (R+G+B) sin(theta) -- an inverse-CDF search.  Same distribution here with a two-level CDF (row marginal, then the
// row-conditional column CDF: short fp32 prefix sums instead of one over 2 M cells), one Philox draw per (point, sample),
// fused with the cosine mask of scripts/relight_importance.py:125-127: the [M][Ns][3] direction / radiance / pdf tensors
// of the reference (100 MB per chunk at 4096 x 512) are replaced by one int32 cell index per sample.
__device__ __forceinline__ int cdf_upper_bound(const float* __restrict__ cdf, int n, float u) {
    int lo = 0, hi = n - 1;                       // first index with cdf[i] > u (clamped to n-1)
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}

// The same search started from a guide table: guide[k] = cdf_upper_bound(cdf, n, k / G) for k = 0 .. G (G a power of two, so
// k / G and u * G are exact in fp32 and floor(u G) = k means k / G <= u < (k + 1) / G).  upper_bound is monotone in u, so the
// answer lies in [guide[k], guide[k + 1]] and every entry before guide[k] is <= k / G <= u: the search restricted to that
// range returns the same index as the full one, after ~2 dependent loads + log2(range) instead of log2(n).
template <typename G>
__device__ __forceinline__ int cdf_upper_bound_guided(const float* __restrict__ cdf, int n, float u, const G* __restrict__ guide, int g) {
    const int k = min((int)(u * (float)g), g - 1);
    int lo = (int)guide[k], hi = (int)guide[k + 1];
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] > u) hi = mid; else lo = mid + 1;
    }
    return lo;
}

__global__ void __launch_bounds__(256)
k_env_sample_setup(const float* __restrict__ row_cdf, const float* __restrict__ col_cdf, int H, int W,
                   const float* __restrict__ env_dir, const float* __restrict__ normal, int M, int Ns,
                   unsigned long long seed, unsigned long long offset, int32_t* __restrict__ cell,
                   uint8_t* __restrict__ active) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * Ns) return;
    const int m = (int)(i / Ns);
    float u[4];
    tir::philox_uniform4(seed, offset, (uint64_t)i, u);
    const int row = cdf_upper_bound(row_cdf, H, u[0]);
    const int col = cdf_upper_bound(col_cdf + (size_t)row * W, W, u[1]);
    const int c = row * W + col;
    cell[i] = c;
    const float* d = env_dir + 3 * (size_t)c;
    const float* nm = normal + 3 * (size_t)m;
    const float cosine = d[0] * nm[0] + d[1] * nm[1] + d[2] * nm[2];          // einsum('ijk,ik->ij') (:125)
    active[i] = cosine > 1e-6f ? 1 : 0;                                       // cosine_mask (:127)
}

// The same draws + mask, and the unmasked (point, cell) pairs as a compacted list for the visibility march.  About half of
// the pairs fail the cosine mask (:127 drops them before the visibility query); marched as a masked list they left half-empty
// waves behind (the march gives half a wave to a ray): the list alone makes the march 21 % faster (C5 view, 400^3 field:
// 0.848 -> 0.673 s of march per view, profiles/r04_c5_pair_lists.json).  A block owns `block_pairs` consecutive pairs, counts
// its active pairs per coarse direction bin (bins_r x bins_c cells of the map: LDS histogram), reserves its segment of the
// list with ONE atomic and writes the pair ids bin by bin (counting sort in LDS; a global `argsort` of a chunk's 2 M keys cost
// what the ordering saved, HISTORY 8.4).  Measured: one surface point per block (512 pairs) and 15 x 17 bins -- the two rays
// a wave marches together then share the origin and point within ~10 degrees of each other -- is worth another 1.7 % of the
// march; blocks of several points (8 x 8 bins over 2 / 8 points) are slower than no bins at all.  vis of the masked pairs is
// written here (0), so the march touches only listed pairs.  List order does not enter any result: every per-pair output is
// addressed by pair id.  The inverse-CDF search starts from guide tables (cdf_upper_bound_guided).
__global__ void __launch_bounds__(256)
k_env_sample_list(const float* __restrict__ row_cdf, const float* __restrict__ col_cdf, int H, int W,
                    const float* __restrict__ env_dir, int dir_stride, const float* __restrict__ normal, int M, int Ns,
                    unsigned long long seed, unsigned long long offset, int bins_r, int bins_c, int block_pairs,
                    const int32_t* __restrict__ row_guide, const uint16_t* __restrict__ col_guide, int g_rows, int g_cols,
                    int32_t* __restrict__ cell, float* __restrict__ vis, int32_t* __restrict__ pair_ids,
                    int32_t* __restrict__ n_active, const int32_t* __restrict__ m_dev) {
    extern __shared__ __attribute__((aligned(16))) unsigned char sb_lds[];
    if (m_dev) M = min(M, max(*m_dev, 0));        // device-side point count (tir_surface_compact): blocks past it have nothing to do
    uint16_t* const list = reinterpret_cast<uint16_t*>(sb_lds);                  // [block_pairs] local pair index, bin-major
    uint8_t* const keys = sb_lds + (size_t)block_pairs * 2;                      // [block_pairs] bin, 255 = masked
    __shared__ int s_hist[256], s_cursor[256], s_base;
    const int64_t n = (int64_t)M * Ns;
    const int64_t base = (int64_t)blockIdx.x * block_pairs;
    if (base >= n) return;                        // (block-uniform)
    const int cnt = (int)min((int64_t)block_pairs, n - base);
    const int n_bins = bins_r * bins_c;
    s_hist[threadIdx.x] = 0;
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
        const int64_t i = base + j;
        const int m = (int)(i / Ns);
        float u[4];
        tir::philox_uniform4(seed, offset, (uint64_t)i, u);
        const int row = row_guide ? cdf_upper_bound_guided(row_cdf, H, u[0], row_guide, g_rows) : cdf_upper_bound(row_cdf, H, u[0]);
        const int col = col_guide ? cdf_upper_bound_guided(col_cdf + (size_t)row * W, W, u[1], col_guide + (size_t)row * (g_cols + 2), g_cols)
                                  : cdf_upper_bound(col_cdf + (size_t)row * W, W, u[1]);
        const int c = row * W + col;
        cell[i] = c;
        const float* d = env_dir + (size_t)dir_stride * (size_t)c;
        const float* nm = normal + 3 * (size_t)m;
        const float cosine = d[0] * nm[0] + d[1] * nm[1] + d[2] * nm[2];          // einsum('ijk,ik->ij') (:125)
        int key = 255;
        if (cosine > 1e-6f) {                                                     // cosine_mask (:127)
            key = (int)(((int64_t)row * bins_r) / H) * bins_c + (int)(((int64_t)col * bins_c) / W);
            atomicAdd(&s_hist[key], 1);
        } else {
            vis[i] = 0.0f;
        }
        keys[j] = (uint8_t)key;
    }
    __syncthreads();
    {   // exclusive scan of the bin counts (256 entries, one per thread; unused bins hold 0)
        const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
        const int c = s_hist[threadIdx.x];
        int incl = c;
#pragma unroll
        for (int dd = 1; dd < 64; dd <<= 1) {
            const int oth = __shfl_up(incl, dd, 64);
            if (lane >= dd) incl += oth;
        }
        __shared__ int s_wsum[4];
        if (lane == 63) s_wsum[wv] = incl;
        __syncthreads();
        int before = 0;
        for (int q = 0; q < wv; ++q) before += s_wsum[q];
        s_cursor[threadIdx.x] = before + incl - c;
        if (threadIdx.x == 255) {
            const int total = before + incl;
            s_hist[0] = total;                                                    // (the counts are consumed: reuse slot 0)
            s_base = total > 0 ? atomicAdd(n_active, total) : 0;
        }
    }
    __syncthreads();
    for (int j = threadIdx.x; j < cnt; j += 256) {
        const int key = keys[j];
        if (key < n_bins) list[atomicAdd(&s_cursor[key], 1)] = (uint16_t)j;
    }
    __syncthreads();
    const int total = s_hist[0];
    const int64_t ob = s_base;
    for (int j = threadIdx.x; j < total; j += 256) pair_ids[ob + j] = (int32_t)(base + list[j]);
}

// PACKED: direction, pdf and radiance of a cell come from ONE 32-byte record [dir.xyz, pdf, rgb, 0] (env_dir = that table) instead
// of three tables (12 + 12 + 4 bytes at unrelated addresses: three sectors per sample of a 2 M-cell map, every one a cache miss).
template <bool PACKED>
__global__ void __launch_bounds__(256)
k_relight_importance_cells(const float* __restrict__ normal, const float* __restrict__ albedo,
                           const float* __restrict__ rough, const float* __restrict__ fresnel,
                           const float* __restrict__ rays_d, const int32_t* __restrict__ cell,
                           const float* __restrict__ env_dir, const float* __restrict__ env_rgb,
                           const float* __restrict__ env_pdf, const float* __restrict__ vis, int M, int Ns,
                           float* __restrict__ out, const int32_t* __restrict__ m_dev) {
    const int lane = threadIdx.x & 63;
    const int m = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (m_dev) M = min(M, max(*m_dev, 0));
    if (m >= M) return;
    float view[3] = {-rays_d[3 * (size_t)m], -rays_d[3 * (size_t)m + 1], -rays_d[3 * (size_t)m + 2]};
    normalize3(view[0], view[1], view[2], 1e-6f);
    const float rough3[3] = {rough[m], rough[m], rough[m]};
    const float* nm = normal + 3 * (size_t)m;
    Surface s = make_surface(nm, view, rough3, fresnel + 3 * (size_t)m, albedo + 3 * (size_t)m);
    float c[3] = {0.f, 0.f, 0.f};
    for (int j = lane; j < Ns; j += 64) {
        const size_t mj = (size_t)m * Ns + j;
        const size_t ce = (size_t)cell[mj];
        float lx, ly, lz, pdf, er[3];
        if (PACKED) {
            const float4 a = *reinterpret_cast<const float4*>(env_dir + 8 * ce), b = *reinterpret_cast<const float4*>(env_dir + 8 * ce + 4);
            lx = a.x; ly = a.y; lz = a.z; pdf = a.w; er[0] = b.x; er[1] = b.y; er[2] = b.z;
        } else {
            lx = env_dir[3 * ce]; ly = env_dir[3 * ce + 1]; lz = env_dir[3 * ce + 2]; pdf = env_pdf[ce];
            er[0] = env_rgb[3 * ce]; er[1] = env_rgb[3 * ce + 1]; er[2] = env_rgb[3 * ce + 2];
        }
        const float cosine = lx * nm[0] + ly * nm[1] + lz * nm[2];
        float spec[3];
        ggx_dir(s, lx, ly, lz, spec);
        const float v = vis[mj];
#pragma unroll
        for (int q = 0; q < 3; ++q)
            c[q] += (s.alb_pi[q] + spec[q]) * (v * er[q]) * cosine / pdf;
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        float t = group_sum<64>(c[q]) / (float)Ns;
        t = linear2srgb(fminf(fmaxf(t, 0.f), 1.f));
        if (lane == 0) out[3 * (size_t)m + q] = t;
    }
}

// Environment_Light.get_light (models/relight_utils.py:191-205): bilinear lookup of the map at a direction,
// F.grid_sample(align_corners=True, zero padding) of qx = -theta / pi, qy = 2 (acos(z) - 1e-6) / pi - 1
__global__ void __launch_bounds__(256)
k_env_lookup(const float* __restrict__ env_rgb, int H, int W, const float* __restrict__ dirs, int dir_stride, int64_t n,
             float* __restrict__ out, int out_stride, const int32_t* __restrict__ slot, const float* __restrict__ fg) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (slot) {        // tir_env_compose: a foreground row takes its relit colour (the index_put_ of scripts/relight_importance.py:166-171)
        const int sl = slot[i];
        if (sl >= 0) {
            out[(size_t)out_stride * i] = fg[3 * (size_t)sl]; out[(size_t)out_stride * i + 1] = fg[3 * (size_t)sl + 1];
            out[(size_t)out_stride * i + 2] = fg[3 * (size_t)sl + 2];
            return;
        }
    }
    const float dx = dirs[(size_t)dir_stride * i], dy = dirs[(size_t)dir_stride * i + 1], dz = dirs[(size_t)dir_stride * i + 2];
    // |dz| can exceed 1 by an ulp after the rotation / normalisation: acosf would return NaN and the (int) conversion of the
    // NaN row index is undefined; the reference has the same NaN at the source, reproducing it buys nothing
    const float phi = acosf(fminf(fmaxf(dz, -1.0f), 1.0f)) - 1e-6f;
    const float theta = atan2f(dy, dx);
    const float qy = (phi / 3.14159265358979323846f) * 2.0f - 1.0f;
    const float qx = -theta / 3.14159265358979323846f;
    const float ix = ((qx + 1.0f) * 0.5f) * (float)(W - 1), iy = ((qy + 1.0f) * 0.5f) * (float)(H - 1);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int xx = x0 + (t & 1), yy = y0 + (t >> 1);
        const float w = ((t & 1) ? tx : 1.0f - tx) * ((t >> 1) ? ty : 1.0f - ty);
        if (xx >= 0 && xx < W && yy >= 0 && yy < H) {
            const float* p = env_rgb + 3 * ((size_t)yy * W + xx);
            c[0] = fmaf(w, p[0], c[0]); c[1] = fmaf(w, p[1], c[1]); c[2] = fmaf(w, p[2], c[2]);
        }
    }
    out[(size_t)out_stride * i] = c[0]; out[(size_t)out_stride * i + 1] = c[1]; out[(size_t)out_stride * i + 2] = c[2];
}

// The acc > thres rows of a chunk's primary maps as compacted surface-point arrays, in ascending row order (= the boolean-mask
// indexing of scripts/relight_importance.py:99-113 without a host round trip): surf = o + depth d (:104), normal / albedo /
// roughness / fresnel / ray direction rows, slot[row] = compacted index or -1, n_hit[0] = their count.  One block walks the
// chunk in slabs of 1024 rows (ordered: wave ballots + a 16-entry scan per slab); rows >= n_hit of the outputs are not written.
__global__ void __launch_bounds__(1024)
k_surface_compact(const float* __restrict__ maps, const float* __restrict__ rays, int B, float acc_thres, float* __restrict__ surf,
                  float* __restrict__ nrm, float* __restrict__ alb, float* __restrict__ rgh, float* __restrict__ fr,
                  float* __restrict__ rd, int32_t* __restrict__ slot, int32_t* __restrict__ n_hit) {
    __shared__ int s_wcnt[16];
    __shared__ int s_run;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    if (threadIdx.x == 0) s_run = 0;
    __syncthreads();
    for (int base = 0; base < B; base += 1024) {
        const int row = base + threadIdx.x;
        const float* mp = maps + (size_t)(row < B ? row : 0) * TIR_MAP_STRIDE;
        const bool hit = row < B && mp[14] > acc_thres;
        const unsigned long long mask = __ballot(hit);
        if (lane == 0) s_wcnt[wv] = __popcll(mask);
        __syncthreads();
        int before = s_run;
        for (int q = 0; q < wv; ++q) before += s_wcnt[q];
        if (hit) {
            const int sl = before + __popcll(mask & ((1ull << lane) - 1ull));
            const float* r = rays + 6 * (size_t)row;
            const float depth = mp[3];
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                surf[3 * (size_t)sl + a] = add_rn(r[a], mul_rn(depth, r[3 + a]));
                nrm[3 * (size_t)sl + a] = mp[4 + a];
                alb[3 * (size_t)sl + a] = mp[7 + a];
                fr[3 * (size_t)sl + a] = mp[11 + a];
                rd[3 * (size_t)sl + a] = r[3 + a];
            }
            rgh[sl] = mp[10];
            slot[row] = sl;
        } else if (row < B) {
            slot[row] = -1;
        }
        __syncthreads();
        if (threadIdx.x == 0) { int t = 0; for (int q = 0; q < 16; ++q) t += s_wcnt[q]; s_run += t; }
        __syncthreads();
    }
    if (threadIdx.x == 0) n_hit[0] = s_run;
}

// get_light_rgbs for light_kind == 'pixel' (models/tensorBase_rotated_lights.py:585-605): the environment map is a learnable
// [H][W][3] image passed through softplus(beta = 5); a direction, rotated into light l's frame (dirs . R_l), is looked up with
// F.grid_sample(align_corners=False, zero padding) at qx = -theta / pi, qy = 2 (acos(z) - 1e-6) / pi - 1.
// One thread per (light, direction).  PixTap: the four taps of a lookup (index or -1 when outside, weight).
struct PixTap { int idx[4]; float w[4]; };

__device__ __forceinline__ PixTap pixel_taps(const float* __restrict__ rot, const float* __restrict__ dirs, int l, int64_t d,
                                             int H, int W) {
    const float* R = rot + 9 * (size_t)l;
    const float x = dirs[3 * d], y = dirs[3 * d + 1], z = dirs[3 * d + 2];
    const float rx = x * R[0] + y * R[3] + z * R[6], ry = x * R[1] + y * R[4] + z * R[7], rz = x * R[2] + y * R[5] + z * R[8];
    const float PI = 3.14159265358979323846f;
    // clamp: see k_env_lookup -- here a NaN tap weight would be atomically added into the light image's gradient and stay there
    const float phi = acosf(fminf(fmaxf(rz, -1.0f), 1.0f)) - 1e-6f, theta = atan2f(ry, rx);
    const float qy = (phi / PI) * 2.0f - 1.0f, qx = -theta / PI;
    const float ix = ((qx + 1.0f) * (float)W - 1.0f) * 0.5f, iy = ((qy + 1.0f) * (float)H - 1.0f) * 0.5f;   // align_corners=False
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy;
    const float tx = ix - fx, ty = iy - fy;
    PixTap t;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int xx = x0 + (k & 1), yy = y0 + (k >> 1);
        t.w[k] = ((k & 1) ? tx : 1.0f - tx) * ((k >> 1) ? ty : 1.0f - ty);
        t.idx[k] = (xx >= 0 && xx < W && yy >= 0 && yy < H) ? yy * W + xx : -1;
    }
    return t;
}

__device__ __forceinline__ float softplus5(float x) {        // torch softplus(beta=5, threshold=20)
    const float bx = 5.0f * x;
    return bx > 20.0f ? x : log1pf(expf(bx)) * 0.2f;
}

__global__ void __launch_bounds__(256)
k_env_pixel(const float* __restrict__ light_rgbs, int H, int W, const float* __restrict__ rot, const float* __restrict__ dirs,
            int L, int64_t D, int softplus, float* __restrict__ env) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)L * D) return;
    const int l = (int)(i / D);
    const PixTap t = pixel_taps(rot, dirs, l, i % D, H, W);
    float c[3] = {0.f, 0.f, 0.f};
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.idx[k] >= 0) {
            const float* p = light_rgbs + 3 * (size_t)t.idx[k];
            c[0] = fmaf(t.w[k], softplus ? softplus5(p[0]) : p[0], c[0]); c[1] = fmaf(t.w[k], softplus ? softplus5(p[1]) : p[1], c[1]);
            c[2] = fmaf(t.w[k], softplus ? softplus5(p[2]) : p[2], c[2]);
        }
    env[3 * i] = c[0]; env[3 * i + 1] = c[1]; env[3 * i + 2] = c[2];
}

// d loss / d light_rgbs[h][w][c] += g_env . tap weight . sigmoid(5 x)   (caller zero-fills g_light)
__global__ void __launch_bounds__(256)
k_env_pixel_bwd(const float* __restrict__ light_rgbs, int H, int W, const float* __restrict__ rot,
                const float* __restrict__ dirs, int L, int64_t D, const float* __restrict__ g_env, float* __restrict__ g_light) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)L * D) return;
    const int l = (int)(i / D);
    const PixTap t = pixel_taps(rot, dirs, l, i % D, H, W);
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (t.idx[k] >= 0) {
#pragma unroll
            for (int ch = 0; ch < 3; ++ch) {
                const float x = light_rgbs[3 * (size_t)t.idx[k] + ch];
                const float ds = (5.0f * x > 20.0f) ? 1.0f : 1.0f / (1.0f + expf(-5.0f * x));
                unsafeAtomicAdd(g_light + 3 * (size_t)t.idx[k] + ch, g_env[3 * i + ch] * t.w[k] * ds);
            }
        }
}

__global__ void __launch_bounds__(256)
k_ggx(const float* __restrict__ normal, const float* __restrict__ v, const float* __restrict__ l,
      const float* __restrict__ rough, const float* __restrict__ fresnel, int M, int D, float* __restrict__ spec) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)M * D) return;
    const int m = (int)(i / D);
    Surface s = make_surface(normal + 3 * (size_t)m, v + 3 * (size_t)m, rough + 3 * (size_t)m,
                             fresnel + 3 * (size_t)m, nullptr);
    float sp[3];
    ggx_dir(s, l[3 * i], l[3 * i + 1], l[3 * i + 2], sp);
    spec[3 * i] = sp[0]; spec[3 * i + 1] = sp[1]; spec[3 * i + 2] = sp[2];
}

}  // namespace

extern "C" int tir_env_sg_fwd(const TirEnvSG* e, const float* dirs, int32_t D, float* out, void* stream) {
    if (!e || !e->sgs || !e->rot || e->n_sg <= 0 || e->n_lights <= 0 || D < 0) return TIR_ERR_ARG;
    if (D == 0) return TIR_OK;
    if (!dirs || !out) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_env_sg, dim3(e->n_lights * D), dim3(128), 0, tir_stream(stream), *e, dirs, D, out);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_shade_setup(const float* maps, const float* rays, const float* dirs, int32_t M,
                               int32_t D, float acc_thres, float* surf, uint8_t* active, void* stream) {
    if (M < 0 || D <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!maps || !rays || !dirs || !surf || !active) return TIR_ERR_ARG;
    int64_t n = (int64_t)M * D;
    hipLaunchKernelGGL(k_shade_setup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), maps,
                       rays, dirs, M, D, acc_thres, surf, active, (int32_t*)nullptr, (int32_t*)nullptr, (float*)nullptr,
                       (int32_t*)nullptr, 0);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_shade_setup_compact(const float* maps, const float* rays, const float* dirs, int32_t M,
                                       int32_t D, float acc_thres, float* surf, uint8_t* active, int32_t* pair_ids,
                                       int32_t* n_active, float* vis, int32_t* ray_rec_cnt, int32_t pair_order, void* stream) {
    if (M < 0 || D <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!maps || !rays || !dirs || !surf || !active || !pair_ids || !n_active) return TIR_ERR_ARG;
    int64_t n = (int64_t)M * D;
    if (n >= ((int64_t)1 << 31)) return TIR_ERR_UNSUPPORTED;
    // pair-list order: direction-major by default (+3.7 % on the whole step, -9 % on the secondary march);
    // pair_order = 2 gives point-major for A/B runs
    if (pair_order < 0 || pair_order > 2) return TIR_ERR_ARG;
    const int dir_major = pair_order == 2 ? 0 : 1;
    hipLaunchKernelGGL(k_shade_setup, dim3((unsigned)((n + 1023) / 1024)), dim3(1024), 0, tir_stream(stream), maps,
                       rays, dirs, M, D, acc_thres, surf, active, pair_ids, n_active, vis, ray_rec_cnt, dir_major);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_shade_integrate(const float* maps, const float* rays, const float* dirs,
                                   const int32_t* light_idx, const float* vis, const float* indirect,
                                   const float* env, const float* weight_d, int32_t M, int32_t D,
                                   int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                                   float* out_rgb, void* stream) {
    if (M < 0 || D <= 0 || n_lights <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!maps || !rays || !dirs || !vis || !env || !out_rgb || (!equal_area && !weight_d)) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_shade_integrate, dim3((M + 3) / 4), dim3(256), 0, tir_stream(stream), maps, rays, dirs,
                       light_idx, vis, indirect, env, weight_d, M, D, n_lights, equal_area, use_srgb, acc_thres, out_rgb,
                       (const int32_t*)nullptr, (const int32_t*)nullptr, (const float*)nullptr, (const float*)nullptr,
                       (int32_t*)nullptr);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_shade_integrate_records(const float* maps, const float* rays, const float* dirs,
                                           const int32_t* light_idx, const float* vis, const int32_t* ray_rec_off,
                                           const int32_t* ray_rec_cnt, const float* rec_w, const float* rec_rgb,
                                           const float* env, const float* weight_d, int32_t M, int32_t D,
                                           int32_t n_lights, int32_t equal_area, int32_t use_srgb, float acc_thres,
                                           float* out_rgb, int32_t* reset_counter, void* stream) {
    if (M < 0 || D <= 0 || n_lights <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!maps || !rays || !dirs || !vis || !env || !out_rgb || (!equal_area && !weight_d)) return TIR_ERR_ARG;
    if (!ray_rec_off || !ray_rec_cnt || !rec_w || !rec_rgb) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_shade_integrate, dim3((M + 3) / 4), dim3(256), 0, tir_stream(stream), maps, rays, dirs,
                       light_idx, vis, (const float*)nullptr, env, weight_d, M, D, n_lights, equal_area, use_srgb,
                       acc_thres, out_rgb, ray_rec_off, ray_rec_cnt, rec_w, rec_rgb, reset_counter);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_relight_importance(const float* normal, const float* albedo, const float* rough,
                                      const float* fresnel, const float* rays_d, const float* light_dir,
                                      const float* light_rgb, const float* light_pdf, const float* vis,
                                      int32_t M, int32_t Ns, float* out_rgb, void* stream) {
    if (M < 0 || Ns <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!normal || !albedo || !rough || !fresnel || !rays_d || !light_dir || !light_rgb || !light_pdf || !vis || !out_rgb)
        return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_relight_importance, dim3((M + 3) / 4), dim3(256), 0, tir_stream(stream), normal, albedo,
                       rough, fresnel, rays_d, light_dir, light_rgb, light_pdf, vis, M, Ns, out_rgb);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_sample_setup(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                                    const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                                    int32_t* cell, uint8_t* active, void* stream) {
    if (M < 0 || Ns <= 0 || H <= 0 || W <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!row_cdf || !col_cdf || !env_dir || !normal || !cell || !active) return TIR_ERR_ARG;
    const int64_t n = (int64_t)M * Ns;
    hipLaunchKernelGGL(k_env_sample_setup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), row_cdf,
                       col_cdf, H, W, env_dir, normal, M, Ns, (unsigned long long)seed, (unsigned long long)offset, cell, active);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_sample_setup_list_n(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                                           int32_t dir_stride, const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                                           int32_t bins_r, int32_t bins_c, int32_t block_pairs, const int32_t* row_guide,
                                           const uint16_t* col_guide, int32_t guide_rows, int32_t guide_cols, int32_t* cell,
                                           float* vis, int32_t* pair_ids, int32_t* n_active, const int32_t* m_dev, void* stream) {
    if (M < 0 || Ns <= 0 || H <= 0 || W <= 0 || bins_r <= 0 || bins_c <= 0 || dir_stride < 3) return TIR_ERR_ARG;
    if ((row_guide == nullptr) != (col_guide == nullptr)) return TIR_ERR_ARG;
    if (row_guide) {        // power-of-two guide sizes (exact k / G thresholds), column indices in 16 bits
        if (guide_rows <= 0 || guide_cols <= 0 || (guide_rows & (guide_rows - 1)) || (guide_cols & (guide_cols - 1))) return TIR_ERR_ARG;
        if (W > 65535 || guide_rows > (1 << 20) || guide_cols > (1 << 20)) return TIR_ERR_UNSUPPORTED;
    }
    if ((int64_t)bins_r * bins_c > 255 || block_pairs < 256 || block_pairs > 32768) return TIR_ERR_UNSUPPORTED;
    if (M == 0) return TIR_OK;
    if (!row_cdf || !col_cdf || !env_dir || !normal || !cell || !vis || !pair_ids || !n_active) return TIR_ERR_ARG;
    const int64_t n = (int64_t)M * Ns;
    if (n >= (int64_t)1 << 31) return TIR_ERR_UNSUPPORTED;
    const size_t lds = (size_t)block_pairs * 3;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_env_sample_list), 100 * 1024)) return rc;
    hipLaunchKernelGGL(k_env_sample_list, dim3((unsigned)((n + block_pairs - 1) / block_pairs)), dim3(256), lds,
                       tir_stream(stream), row_cdf, col_cdf, H, W, env_dir, dir_stride, normal, M, Ns, (unsigned long long)seed,
                       (unsigned long long)offset, bins_r, bins_c, block_pairs, row_guide, col_guide, guide_rows, guide_cols, cell,
                       vis, pair_ids, n_active, m_dev);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_sample_setup_list(const float* row_cdf, const float* col_cdf, int32_t H, int32_t W, const float* env_dir,
                                         int32_t dir_stride, const float* normal, int32_t M, int32_t Ns, uint64_t seed, uint64_t offset,
                                         int32_t bins_r, int32_t bins_c, int32_t block_pairs, const int32_t* row_guide,
                                         const uint16_t* col_guide, int32_t guide_rows, int32_t guide_cols, int32_t* cell,
                                         float* vis, int32_t* pair_ids, int32_t* n_active, void* stream) {
    return tir_env_sample_setup_list_n(row_cdf, col_cdf, H, W, env_dir, dir_stride, normal, M, Ns, seed, offset, bins_r, bins_c, block_pairs,
                                       row_guide, col_guide, guide_rows, guide_cols, cell, vis, pair_ids, n_active, nullptr, stream);
}

extern "C" int tir_relight_importance_cells(const float* normal, const float* albedo, const float* rough,
                                            const float* fresnel, const float* rays_d, const int32_t* cell,
                                            const float* env_dir, const float* env_rgb, const float* env_pdf,
                                            const float* vis, int32_t M, int32_t Ns, float* out_rgb, void* stream) {
    if (M < 0 || Ns <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!normal || !albedo || !rough || !fresnel || !rays_d || !cell || !env_dir || !env_rgb || !env_pdf || !vis || !out_rgb)
        return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_relight_importance_cells<false>, dim3((M + 3) / 4), dim3(256), 0, tir_stream(stream), normal, albedo,
                       rough, fresnel, rays_d, cell, env_dir, env_rgb, env_pdf, vis, M, Ns, out_rgb, nullptr);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_relight_importance_cells_packed_n(const float* normal, const float* albedo, const float* rough,
                                                     const float* fresnel, const float* rays_d, const int32_t* cell,
                                                     const float* env_cell, const float* vis, int32_t M, int32_t Ns,
                                                     float* out_rgb, const int32_t* m_dev, void* stream) {
    if (M < 0 || Ns <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!normal || !albedo || !rough || !fresnel || !rays_d || !cell || !env_cell || !vis || !out_rgb) return TIR_ERR_ARG;
    if (reinterpret_cast<uintptr_t>(env_cell) % 16 != 0) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_relight_importance_cells<true>, dim3((M + 3) / 4), dim3(256), 0, tir_stream(stream), normal, albedo,
                       rough, fresnel, rays_d, cell, env_cell, nullptr, nullptr, vis, M, Ns, out_rgb, m_dev);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_relight_importance_cells_packed(const float* normal, const float* albedo, const float* rough,
                                                   const float* fresnel, const float* rays_d, const int32_t* cell,
                                                   const float* env_cell, const float* vis, int32_t M, int32_t Ns,
                                                   float* out_rgb, void* stream) {
    return tir_relight_importance_cells_packed_n(normal, albedo, rough, fresnel, rays_d, cell, env_cell, vis, M, Ns, out_rgb, nullptr, stream);
}

extern "C" int tir_env_lookup(const float* env_rgb, int32_t H, int32_t W, const float* dirs, int64_t n, float* out,
                              void* stream) {
    if (n < 0 || H <= 0 || W <= 0) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (!env_rgb || !dirs || !out) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_env_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), env_rgb, H, W,
                       dirs, 3, n, out, 3, nullptr, nullptr);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_compose(const float* env_rgb, int32_t H, int32_t W, const float* dirs, int32_t dir_stride, int64_t n,
                               const int32_t* slot, const float* fg_rgb, float* out, int32_t out_stride, void* stream) {
    if (n < 0 || H <= 0 || W <= 0 || dir_stride < 3 || out_stride < 3) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    if (!env_rgb || !dirs || !out || !slot || !fg_rgb) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_env_lookup, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), env_rgb, H, W,
                       dirs, dir_stride, n, out, out_stride, slot, fg_rgb);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_surface_compact(const float* maps, const float* rays, int32_t B, float acc_thres, float* surf, float* normal,
                                   float* albedo, float* rough, float* fresnel, float* rays_d, int32_t* slot, int32_t* n_hit,
                                   void* stream) {
    if (B < 0) return TIR_ERR_ARG;
    if (!n_hit || (B > 0 && (!maps || !rays || !surf || !normal || !albedo || !rough || !fresnel || !rays_d || !slot))) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_surface_compact, dim3(1), dim3(1024), 0, tir_stream(stream), maps, rays, B, acc_thres, surf, normal, albedo,
                       rough, fresnel, rays_d, slot, n_hit);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_pixel_fwd(const float* light_rgbs, int32_t H, int32_t W, const float* rot, const float* dirs,
                                 int32_t n_lights, int64_t n_dirs, int32_t softplus, float* env, void* stream) {
    if (H <= 0 || W <= 0 || n_lights <= 0 || n_dirs < 0 || (softplus != 0 && softplus != 1)) return TIR_ERR_ARG;
    if (n_dirs == 0) return TIR_OK;
    if (!light_rgbs || !rot || !dirs || !env) return TIR_ERR_ARG;
    const int64_t n = (int64_t)n_lights * n_dirs;
    hipLaunchKernelGGL(k_env_pixel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), light_rgbs, H, W, rot,
                       dirs, n_lights, n_dirs, softplus, env);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_env_pixel_bwd(const float* light_rgbs, int32_t H, int32_t W, const float* rot, const float* dirs,
                                 int32_t n_lights, int64_t n_dirs, const float* g_env, float* g_light, void* stream) {
    if (H <= 0 || W <= 0 || n_lights <= 0 || n_dirs < 0) return TIR_ERR_ARG;
    if (n_dirs == 0) return TIR_OK;
    if (!light_rgbs || !rot || !dirs || !g_env || !g_light) return TIR_ERR_ARG;
    const int64_t n = (int64_t)n_lights * n_dirs;
    hipLaunchKernelGGL(k_env_pixel_bwd, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), light_rgbs, H, W,
                       rot, dirs, n_lights, n_dirs, g_env, g_light);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_ggx_specular(const float* normal, const float* v, const float* l, const float* rough,
                                const float* fresnel, int32_t M, int32_t D, float* spec, void* stream) {
    if (M < 0 || D <= 0) return TIR_ERR_ARG;
    if (M == 0) return TIR_OK;
    if (!normal || !v || !l || !rough || !fresnel || !spec) return TIR_ERR_ARG;
    int64_t n = (int64_t)M * D;
    hipLaunchKernelGGL(k_ggx, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), normal, v, l,
                       rough, fresnel, M, D, spec);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ---- misc API ---------------------------------------------------------------------------------------
extern "C" int tir_version(void) { return TIR_VERSION; }

extern "C" const char* tir_error_string(int code) {
    switch (code) {
        case TIR_OK: return "ok";
        case TIR_ERR_ARG: return "invalid argument (null pointer, negative size or inconsistent descriptor)";
        case TIR_ERR_UNSUPPORTED: return "shape not supported by the gfx950 kernels";
        case TIR_ERR_NO_DEVICE: return "no usable HIP device";
        default: return code < 0 ? hipGetErrorString((hipError_t)(-code)) : "unknown";
    }
}

__global__ void k_probe(int* x) { if (threadIdx.x == 0) *x = 950; }

extern "C" int tir_device_check(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return TIR_ERR_NO_DEVICE;
    int* d = nullptr;
    if (hipMalloc(&d, sizeof(int)) != hipSuccess) return TIR_ERR_NO_DEVICE;
    hipLaunchKernelGGL(k_probe, dim3(1), dim3(64), 0, 0, d);
    int h = 0;
    hipError_t e = hipMemcpy(&h, d, sizeof(int), hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return (e == hipSuccess && h == 950) ? TIR_OK : TIR_ERR_NO_DEVICE;
}
