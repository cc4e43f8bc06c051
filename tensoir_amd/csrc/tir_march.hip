// Ray marching kernels: primary (K1+K2+K3), record compaction, compositing, secondary (K7).
#include "tir_common.hpp"

using namespace tir;

// ------------------------------------------------------------------------------------------------
// primary march: one wave64 per ray, 64 consecutive samples per step, transmittance carried
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ void march_one_ray(const TirField& f, const float* __restrict__ rays, const float* __restrict__ ray_jitter,
                                              int B, int S, float t_stop, float* __restrict__ weight, float* __restrict__ acc_out,
                                              float* __restrict__ depth_out, float* __restrict__ tend_out,
                                              int32_t* __restrict__ app_count, unsigned long long* __restrict__ stats,
                                              float* __restrict__ sigma_out, float* __restrict__ viewdirs, bool coh, int ray, int lane) {
    RaySetup rs = ray_setup(f, rays, ray);
    if (viewdirs && lane < 3) viewdirs[3 * (size_t)ray + lane] = rays[6 * (size_t)ray + 3 + lane];

    const bool hj = ray_jitter != nullptr;
    const float jit = hj ? ray_jitter[ray] : 0.0f;

    __shared__ __attribute__((aligned(16))) float wl_all[4][256];
    float* wl = wl_all[threadIdx.x >> 6];
    float T = 1.0f;          // transmittance before the current step (wave-uniform)
    float acc = 0.0f, depth = 0.0f;
    int cnt = 0;
    int k0 = 0;
    unsigned n_gather = 0;
    for (; k0 < S; k0 += 64) {
        const int k = k0 + lane;
        float z = 0.0f, x = 0.f, y = 0.f, zz = 0.f;
        bool valid = false;
        if (k < S) {
            z = sample_z(f, rs.t_min, k, jit, hj);
            float px = add_rn(rs.o[0], mul_rn(rs.d[0], z));
            float py = add_rn(rs.o[1], mul_rn(rs.d[1], z));
            float pz = add_rn(rs.o[2], mul_rn(rs.d[2], z));
            valid = sample_valid(f, px, py, pz, x, y, zz);
        }
        const float sigma = wave_sigma(f, valid, x, y, zz, wl);
        float w = 0.0f, v = 1.0f;
        if (k < S) {
            // dists: z[k+1]-z[k], last 0 (:887); raw2alpha (:21-28) with dist * distance_scale (:921)
            float dist = (k + 1 < S) ? sub_rn(sample_z(f, rs.t_min, k + 1, jit, hj), z) : 0.0f;
            float alpha = 1.0f - expf(-sigma * mul_rn(dist, f.distance_scale));
            v = add_rn(sub_rn(1.0f, alpha), 1e-10f);
            w = alpha;   // multiplied by the exclusive transmittance below
        }
        float incl = scan_prod<64>(v, lane);
        float excl = shift_up1<64>(incl, lane);
        w = w * (T * excl);
        if (k < S) weight[(size_t)ray * S + k] = w;
        if (sigma_out && k < S) sigma_out[(size_t)ray * S + k] = sigma;
        acc += w;
        depth = fmaf(w, z, depth);
        cnt += __popcll(__ballot(w > f.weight_thres));
        if (stats) n_gather += __popcll(__ballot(valid));
        T = T * __shfl(incl, 63, 64);
        if (T < t_stop) { k0 += 64; break; }
    }
    // early stop: remaining weights are zero
    for (; k0 < S; k0 += 64) {
        const int k = k0 + lane;
        if (k < S) weight[(size_t)ray * S + k] = 0.0f;
        if (sigma_out && k < S) sigma_out[(size_t)ray * S + k] = 0.0f;
    }
    acc = group_sum<64>(acc);
    depth = group_sum<64>(depth);
    if (lane == 0) {
        acc_out[ray] = acc;
        depth_out[ray] = depth;
        if (tend_out) tend_out[ray] = T;
        // coh: the count is consumed by ANOTHER workgroup of this launch (the last one scans): write-through store
        if (coh) __hip_atomic_store(app_count + ray, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else app_count[ray] = cnt;
        if (stats) atomicAdd(stats, (unsigned long long)n_gather);
    }
}


// Optional by-products of the primary march that used to be separate (framework) launches of a step:
//   viewdirs  [B,3]   the ray directions as a contiguous table (aux input of the radiance decoder);
//   zero_words        n_zero int32 counters of LATER kernels of the same pass (secondary record counter) re-armed here;
//   ticket/offsets    the capped exclusive scan of the per-ray record counts, done by the block that finishes last
//                     (agent-scope release / acquire around a ticket counter that re-arms itself) -- deterministic.
struct TirMarchExtras {
    float* viewdirs;
    int32_t* zero_words; int n_zero;
    int32_t* ticket; int32_t* offsets; int cap; int32_t* total_out;
};

__device__ __forceinline__ void block_exclusive_scan_capped(const int32_t* __restrict__ counts, int32_t* __restrict__ offsets,
                                                             int n, int cap, int32_t* __restrict__ total_out) {
    // all 256 threads of the calling block; chunked, carry in LDS (same arithmetic as k_exclusive_scan)
    __shared__ int32_t wsum[4];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 256) {
        const int i = base + tid;
        const int32_t v = (i < n) ? __hip_atomic_load(counts + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0;
        int32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int32_t woff = 0;
        for (int q = 0; q < wv; ++q) woff += wsum[q];
        const int32_t carry = carry_s;
        if (i < n) offsets[i] = min(carry + woff + incl - v, cap);
        __syncthreads();
        if (tid == 255) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        offsets[n] = min(carry_s, cap);
        if (total_out) *total_out = carry_s;
    }
}

__global__ void __launch_bounds__(256)
k_march_primary(TirField f, const float* __restrict__ rays, const float* __restrict__ ray_jitter,
                int B, int S, float t_stop, float* __restrict__ weight, float* __restrict__ acc_out,
                float* __restrict__ depth_out, float* __restrict__ tend_out, int32_t* __restrict__ app_count,
                unsigned long long* __restrict__ stats, float* __restrict__ sigma_out, TirMarchExtras ex) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ex.zero_words && blockIdx.x == 0 && threadIdx.x < ex.n_zero) ex.zero_words[threadIdx.x] = 0;
    if (ray < B) march_one_ray(f, rays, ray_jitter, B, S, t_stop, weight, acc_out, depth_out, tend_out, app_count, stats,
                               sigma_out, ex.viewdirs, ex.ticket != nullptr, ray, lane);
    if (!ex.ticket) return;
    // last workgroup done -> scan.  The counts are published with write-through (sc1) stores, each wave drains its own
    // stores, then one ticket per workgroup; the holder of the last ticket reads all B counts with sc1 loads.  (An
    // agent-scope release fence per workgroup would write back the XCD's whole dirty L2 -- the 8 MB of weights this
    // kernel has just produced -- 1024 times: measured 0.035 -> 0.066 ms.)
    __shared__ int s_last;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ex.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (int)gridDim.x - 1);
        if (s_last) __hip_atomic_store(ex.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // re-armed
    }
    __syncthreads();
    if (s_last) block_exclusive_scan_capped(app_count, ex.offsets, B, ex.cap, ex.total_out);
}

extern "C" int tir_march_primary_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                                     int32_t B, int32_t S, float t_stop, float* weight, float* acc,
                                     float* depth, float* t_end, int32_t* app_count,
                                     unsigned long long* stats, void* stream) {
    if (!f || B < 0 || S <= 0) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !weight || !acc || !depth || !app_count) return TIR_ERR_ARG;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    if (!tir_occ_index_ok(f) || !tir_plane_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_march_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), *f, rays,
                       ray_jitter, B, S, t_stop, weight, acc, depth, t_end, app_count, stats, (float*)nullptr, TirMarchExtras{});
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// tir_march_primary_fwd + by-products (see TirMarchExtras): viewdirs table, re-armed counters, capped exclusive scan of
// the record counts (ticket: one int32, zero before the first use; offsets [B+1]; total = uncapped record count).
extern "C" int tir_march_primary_fused_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                                           int32_t B, int32_t S, float t_stop, float* weight, float* acc,
                                           float* depth, float* t_end, int32_t* app_count, unsigned long long* stats,
                                           float* viewdirs, int32_t* zero_words, int32_t n_zero, int32_t* ticket,
                                           int32_t* offsets, int32_t cap, int32_t* total, void* stream) {
    if (!f || B < 0 || S <= 0 || n_zero < 0 || n_zero > 256 || cap < 0) return TIR_ERR_ARG;
    if ((ticket == nullptr) != (offsets == nullptr)) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !weight || !acc || !depth || !app_count) return TIR_ERR_ARG;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    if (!tir_occ_index_ok(f) || !tir_plane_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    TirMarchExtras ex{viewdirs, zero_words, n_zero, ticket, offsets, cap, total};
    hipLaunchKernelGGL(k_march_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), *f, rays,
                       ray_jitter, B, S, t_stop, weight, acc, depth, t_end, app_count, stats, (float*)nullptr, ex);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_march_primary_train_fwd(const TirField* f, const float* rays, const float* ray_jitter,
                                           int32_t B, int32_t S, float t_stop, float* weight, float* sigma,
                                           float* acc, float* depth, float* t_end, int32_t* app_count, void* stream) {
    if (!f || B < 0 || S <= 0) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !weight || !sigma || !acc || !depth || !app_count) return TIR_ERR_ARG;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    if (!tir_occ_index_ok(f) || !tir_plane_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(k_march_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), *f, rays,
                       ray_jitter, B, S, t_stop, weight, acc, depth, t_end, app_count,
                       (unsigned long long*)nullptr, sigma, TirMarchExtras{});
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// exclusive scan of small int arrays (ray counts): single workgroup, chunked
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(1024)
k_exclusive_scan(const int32_t* __restrict__ counts, int32_t* __restrict__ offsets, int n, int cap,
                 int32_t* __restrict__ total_out) {
    __shared__ int32_t wsum[16];
    __shared__ int32_t carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        int i = base + tid;
        int32_t v = (i < n) ? counts[i] : 0;
        int32_t incl = v;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            int32_t o = __shfl_up(incl, d, 64);
            if (lane >= d) incl += o;
        }
        if (lane == 63) wsum[wv] = incl;
        __syncthreads();
        int32_t woff = 0;
        for (int q = 0; q < wv; ++q) woff += wsum[q];
        int32_t carry = carry_s;
        if (i < n) offsets[i] = min(carry + woff + incl - v, cap);
        __syncthreads();
        if (tid == 1023) carry_s = carry + woff + incl;
        __syncthreads();
    }
    if (tid == 0) {
        offsets[n] = min(carry_s, cap);
        if (total_out) *total_out = carry_s;
    }
}

extern "C" int tir_exclusive_scan(const int32_t* counts, int32_t* offsets, int32_t n, void* stream) {
    if (!offsets || n < 0 || (n > 0 && !counts)) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, tir_stream(stream), counts, offsets, n, 0x7fffffff,
                       (int32_t*)nullptr);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_exclusive_scan_capped(const int32_t* counts, int32_t* offsets, int32_t n, int32_t cap,
                                         int32_t* total, void* stream) {
    if (!offsets || n < 0 || cap < 0 || (n > 0 && !counts)) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_exclusive_scan, dim3(1), dim3(1024), 0, tir_stream(stream), counts, offsets, n, cap, total);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// Record-capacity check of a captured step (tensoir_amd/graph.py).  Device state (int64[5], kept by the caller, never
// cleared here): running maximum of each counter over all launches [0..3] and a STICKY overflow flag [4], set by any
// launch whose counter exceeded its capacity.  Every launch mirrors {this launch's counters [0..3], the running maxima
// [4..7], the flag [8]} into pinned, device-mapped host memory.  One single-thread kernel instead of a handful of
// framework ops per replay; a caller can queue many replays and later learn whether any overflowed and by how much.
// ------------------------------------------------------------------------------------------------
struct TirCheck4 { const int32_t* cnt[4]; int64_t cap[4]; int n; };

__global__ void k_record_check(TirCheck4 c, int64_t* __restrict__ state, int64_t* __restrict__ host_out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    int64_t over = state[4];
    for (int i = 0; i < 4; ++i) {
        int64_t v = 0;
        if (i < c.n) {
            v = *c.cnt[i];
            if (v > c.cap[i]) over = 1;
            if (v > state[i]) state[i] = v;
        }
        host_out[i] = v;
        host_out[4 + i] = state[i];
    }
    state[4] = over;
    host_out[8] = over;
    __threadfence_system();
}

extern "C" int tir_record_check(const int32_t* const* counters, const int64_t* caps, int32_t n, int64_t* state,
                                int64_t* host_out, void* stream) {
    if (n < 0 || n > 4 || !state || !host_out || (n > 0 && (!counters || !caps))) return TIR_ERR_ARG;
    TirCheck4 c;
    c.n = n;
    for (int i = 0; i < 4; ++i) { c.cnt[i] = i < n ? counters[i] : nullptr; c.cap[i] = i < n ? caps[i] : 0; }
    for (int i = 0; i < n; ++i) if (!c.cnt[i]) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_record_check, dim3(1), dim3(64), 0, tir_stream(stream), c, state, host_out);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// compaction of weight > thres samples into (ray, sample)-ordered records
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_compact_primary(TirField f, const float* __restrict__ rays, const float* __restrict__ ray_jitter,
                  const float* __restrict__ weight, const int32_t* __restrict__ offsets, int B, int S,
                  int32_t* __restrict__ rec_ray, int32_t* __restrict__ rec_k, float* __restrict__ rec_w,
                  float* __restrict__ rec_xyz) {
    const int lane = threadIdx.x & 63;
    const int ray = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ray >= B) return;
    int base = offsets[ray];
    const int end = offsets[ray + 1];       // with capped offsets the records past the capacity are dropped
    if (end == base) return;
    RaySetup rs = ray_setup(f, rays, ray);
    const bool hj = ray_jitter != nullptr;
    const float jit = hj ? ray_jitter[ray] : 0.0f;
    for (int k0 = 0; k0 < S; k0 += 64) {
        const int k = k0 + lane;
        float w = (k < S) ? weight[(size_t)ray * S + k] : 0.0f;
        bool keep = w > f.weight_thres;
        unsigned long long m = __ballot(keep);
        const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && slot < end) {
            float z = sample_z(f, rs.t_min, k, jit, hj);
            rec_ray[slot] = ray;
            rec_k[slot] = k;
            rec_w[slot] = w;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                float p = add_rn(rs.o[a], mul_rn(rs.d[a], z));
                rec_xyz[3 * (size_t)slot + a] = norm_coord(p, f.aabb_min[a], f.inv_aabb[a]);
            }
        }
        base += __popcll(m);
    }
}

extern "C" int tir_compact_primary(const TirField* f, const float* rays, const float* ray_jitter,
                                   const float* weight, const int32_t* offsets, int32_t B, int32_t S,
                                   int32_t* rec_ray, int32_t* rec_k, float* rec_w, float* rec_xyz,
                                   void* stream) {
    if (!f || B < 0 || S <= 0) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !weight || !offsets || !rec_ray || !rec_k || !rec_w || !rec_xyz) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_compact_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), *f, rays,
                       ray_jitter, weight, offsets, B, S, rec_ray, rec_k, rec_w, rec_xyz);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// compositing + tone mapping (models/tensorBase_rotated_lights.py:973-1031): one wave per ray, lanes stride
// over the ray's records, butterfly reduction (deterministic sums).
// ------------------------------------------------------------------------------------------------
struct TirCompositeExtras {
    int32_t* ticket; float* smooth_out;      // smoothness means (both or neither)
    long long* rng_bump; long long rng_step; // device-side {seed, offset} of the jitter noise: offset += step per pass
};

__device__ __forceinline__ void composite_one_ray(const float* __restrict__ rays, const int32_t* __restrict__ offsets,
                    const float* __restrict__ rec_w, const float* __restrict__ rgb,
                    const float* __restrict__ brdf, const float* __restrict__ brdf_jit,
                    const float* __restrict__ pred_n, const float* __restrict__ der_n,
                    const float* __restrict__ acc_in, const float* __restrict__ depth_in, int white_bg,
                    int is_relight, float fixed_fresnel, float* __restrict__ out, bool coh, int r, int lane) {
    const int b = offsets[r], e = offsets[r + 1];
    float c[3] = {0, 0, 0}, nm[3] = {0, 0, 0}, al[3] = {0, 0, 0};
    float rough = 0, ndiff = 0, norient = 0, albc = 0, rghc = 0;
    const float vd[3] = {rays[6 * (size_t)r + 3], rays[6 * (size_t)r + 4], rays[6 * (size_t)r + 5]};
    for (int i = b + lane; i < e; i += 64) {
        const float w = rec_w[i];
        if (rgb) { c[0] = fmaf(w, rgb[3 * (size_t)i], c[0]); c[1] = fmaf(w, rgb[3 * (size_t)i + 1], c[1]); c[2] = fmaf(w, rgb[3 * (size_t)i + 2], c[2]); }
        if (!is_relight) continue;
        float a3[3] = {0, 0, 0}, rg = 0;
        if (brdf) {
            const float4 bv = *reinterpret_cast<const float4*>(brdf + 4 * (size_t)i);
            a3[0] = bv.x; a3[1] = bv.y; a3[2] = bv.z;
            rg = bv.w * 0.9f + 0.09f;                                            // :933
            al[0] = fmaf(w, a3[0], al[0]); al[1] = fmaf(w, a3[1], al[1]); al[2] = fmaf(w, a3[2], al[2]);
            rough = fmaf(w, rg, rough);
        }
        if (brdf && brdf_jit) {                                                  // :937-943, :858-863
            const float4 jv = *reinterpret_cast<const float4*>(brdf_jit + 4 * (size_t)i);
            const float ajv[3] = {jv.x, jv.y, jv.z};
            float cost = 0.f;
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                float base = fmaxf(fmaxf(a3[q], ajv[q]), 1e-6f);
                float dlt = (a3[q] - ajv[q]) / base;
                cost = fmaf(dlt, dlt, cost);
            }
            albc = fmaf(w, cost, albc);
            float rj = jv.w * 0.9f + 0.09f;
            float base = fmaxf(fmaxf(rg, rj), 1e-6f);
            float dlt = (rg - rj) / base;
            rghc = fmaf(w, dlt * dlt, rghc);
        }
        if (pred_n) {                                                            // :953-960
            float p3[3] = {pred_n[3 * (size_t)i], pred_n[3 * (size_t)i + 1], pred_n[3 * (size_t)i + 2]};
            nm[0] = fmaf(w, p3[0], nm[0]); nm[1] = fmaf(w, p3[1], nm[1]); nm[2] = fmaf(w, p3[2], nm[2]);
            if (der_n) {
                float d0 = p3[0] - der_n[3 * (size_t)i], d1 = p3[1] - der_n[3 * (size_t)i + 1], d2 = p3[2] - der_n[3 * (size_t)i + 2];
                ndiff = fmaf(w, d0 * d0 + d1 * d1 + d2 * d2, ndiff);
            }
            float dot = vd[0] * p3[0] + vd[1] * p3[1] + vd[2] * p3[2];
            norient = fmaf(w, fmaxf(dot, 0.f), norient);
        }
    }
#pragma unroll
    for (int q = 0; q < 3; ++q) { c[q] = group_sum<64>(c[q]); }
    if (is_relight) {
#pragma unroll
        for (int q = 0; q < 3; ++q) { nm[q] = group_sum<64>(nm[q]); al[q] = group_sum<64>(al[q]); }
        rough = group_sum<64>(rough); ndiff = group_sum<64>(ndiff); norient = group_sum<64>(norient);
        albc = group_sum<64>(albc); rghc = group_sum<64>(rghc);
    }
    if (lane != 0) return;
    const float acc = acc_in[r];
    float depth = depth_in[r];
    float* o = out + (size_t)r * TIR_MAP_STRIDE;
    const float bg = 1.0f - acc;
    if (!is_relight) {                                                           // :978-986
        if (white_bg) { depth = depth + bg * rays[6 * (size_t)r + 5]; c[0] += bg; c[1] += bg; c[2] += bg; }
        o[0] = c[0]; o[1] = c[1]; o[2] = c[2]; o[3] = depth;
        for (int q = 4; q < TIR_MAP_STRIDE; ++q) o[q] = 0.f;
        o[14] = acc;
        return;
    }
    float fr = fixed_fresnel;
    if (white_bg) {                                                              // :1004-1014
        depth = depth + bg * rays[6 * (size_t)r + 5];                            // quirk: rays_d.z
        c[0] += bg; c[1] += bg; c[2] += bg;
        nm[2] += bg;                                                             // background normal (0,0,1)
        al[0] += bg; al[1] += bg; al[2] += bg;
        rough += bg;
        fr += bg;
    }
    o[0] = linear2srgb(c[0]); o[1] = linear2srgb(c[1]); o[2] = linear2srgb(c[2]);  // :1017-1023
    o[3] = depth;
    float nn = fmaxf(sqrtf(nm[0] * nm[0] + nm[1] * nm[1] + nm[2] * nm[2]), 1e-6f); // :1028
    o[4] = nm[0] / nn; o[5] = nm[1] / nn; o[6] = nm[2] / nn;
    o[7] = fminf(fmaxf(al[0], 0.f), 1.f); o[8] = fminf(fmaxf(al[1], 0.f), 1.f); o[9] = fminf(fmaxf(al[2], 0.f), 1.f);
    o[10] = fminf(fmaxf(rough, 0.f), 1.f);
    fr = fminf(fmaxf(fr, 0.f), 1.f);
    o[11] = fr; o[12] = fr; o[13] = fr;
    o[14] = acc;
    o[15] = ndiff; o[16] = norient; o[19] = 0.f;
    if (coh) {      // read back by the last workgroup of this launch (smoothness means): write-through stores
        __hip_atomic_store(o + 17, albc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(o + 18, rghc, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    } else { o[17] = albc; o[18] = rghc; }
}


__global__ void __launch_bounds__(256)
k_composite_primary(const float* __restrict__ rays, const int32_t* __restrict__ offsets,
                    const float* __restrict__ rec_w, const float* __restrict__ rgb,
                    const float* __restrict__ brdf, const float* __restrict__ brdf_jit,
                    const float* __restrict__ pred_n, const float* __restrict__ der_n,
                    const float* __restrict__ acc_in, const float* __restrict__ depth_in, int B, int white_bg,
                    int is_relight, float fixed_fresnel, float* __restrict__ out, TirCompositeExtras ex) {
    // one wave64 per ray: lane l takes records b+l, b+l+64, ...; fixed-shape butterfly reduction -> the sums
    // depend only on the ray's own records (deterministic, invariant under ray sharding)
    const int lane = threadIdx.x & 63;
    const int r = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (ex.rng_bump && blockIdx.x == 0 && threadIdx.x == 0) ex.rng_bump[1] += ex.rng_step;   // next pass: fresh jitter noise
    if (r < B) composite_one_ray(rays, offsets, rec_w, rgb, brdf, brdf_jit, pred_n, der_n, acc_in, depth_in, white_bg,
                                 is_relight, fixed_fresnel, out, ex.ticket != nullptr, r, lane);
    if (!ex.ticket || !is_relight) return;
    // the two smoothness losses = means over ALL rays of map columns 17 / 18 (models/tensorBase_rotated_lights.py:999-1000):
    // the block that finishes last sums the columns in a fixed order (deterministic; no extra reduction launch)
    __shared__ int s_last;
    __shared__ float s_part[2][4];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's write-through stores have landed
    __syncthreads();
    if (threadIdx.x == 0) {
        const int t = __hip_atomic_fetch_add(ex.ticket, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_last = (t == (int)gridDim.x - 1);
        if (s_last) __hip_atomic_store(ex.ticket, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __syncthreads();
    if (!s_last) return;
    float a = 0.0f, g = 0.0f;
    for (int i = threadIdx.x; i < B; i += 256) {
        a += __hip_atomic_load(out + (size_t)i * TIR_MAP_STRIDE + 17, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        g += __hip_atomic_load(out + (size_t)i * TIR_MAP_STRIDE + 18, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    a = group_sum<64>(a); g = group_sum<64>(g);
    if (lane == 0) { s_part[0][threadIdx.x >> 6] = a; s_part[1][threadIdx.x >> 6] = g; }
    __syncthreads();
    if (threadIdx.x == 0) {
        ex.smooth_out[0] = ((s_part[0][0] + s_part[0][1]) + (s_part[0][2] + s_part[0][3])) / (float)B;
        ex.smooth_out[1] = ((s_part[1][0] + s_part[1][1]) + (s_part[1][2] + s_part[1][3])) / (float)B;
    }
}

extern "C" int tir_composite_primary(const float* rays, const int32_t* offsets, const float* rec_w,
                                     const float* rgb, const float* brdf, const float* brdf_jit,
                                     const float* pred_normal, const float* derived_normal,
                                     const float* acc, const float* depth, int32_t B, int32_t white_bg,
                                     int32_t is_relight, float fixed_fresnel, float* out_maps, void* stream) {
    if (B < 0) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !offsets || !acc || !depth || !out_maps) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_composite_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), rays, offsets,
                       rec_w, rgb, brdf, brdf_jit, pred_normal, derived_normal, acc, depth, B, white_bg,
                       is_relight, fixed_fresnel, out_maps, TirCompositeExtras{});
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// tir_composite_primary + the two smoothness-loss means (smooth_out[2], ticket: one int32, zero before the first use) and
// the per-pass bump of the device-side jitter RNG offset (rng_state = int64 {seed, offset}, may be null).
extern "C" int tir_composite_primary_fused(const float* rays, const int32_t* offsets, const float* rec_w,
                                           const float* rgb, const float* brdf, const float* brdf_jit,
                                           const float* pred_normal, const float* derived_normal,
                                           const float* acc, const float* depth, int32_t B, int32_t white_bg,
                                           int32_t is_relight, float fixed_fresnel, float* out_maps, int32_t* ticket,
                                           float* smooth_out, int64_t* rng_state, int64_t rng_step, void* stream) {
    if (B < 0 || (ticket == nullptr) != (smooth_out == nullptr)) return TIR_ERR_ARG;
    if (B == 0) return TIR_OK;
    if (!rays || !offsets || !acc || !depth || !out_maps) return TIR_ERR_ARG;
    TirCompositeExtras ex{ticket, smooth_out, reinterpret_cast<long long*>(rng_state), (long long)rng_step};
    hipLaunchKernelGGL(k_composite_primary, dim3((B + 3) / 4), dim3(256), 0, tir_stream(stream), rays, offsets,
                       rec_w, rgb, brdf, brdf_jit, pred_normal, derived_normal, acc, depth, B, white_bg,
                       is_relight, fixed_fresnel, out_maps, ex);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// K7 secondary march: half a wave (32 lanes) per (surface point, direction) ray; n_sample/32 steps.
// Weights of the ray are staged in LDS so that the w > thres records can be reserved with one atomic
// and written contiguously in sample order.
// ------------------------------------------------------------------------------------------------
#define TIR_SEC_MAX_SAMPLES 256
#define TIR_SEC_RPB 32     // rays per block: 4 groups of 8 half-waves; ONE record reservation (atomic) per block

// LDS layout (floats): [z table n_sample] [wave scratch 4 x 256] [ints: cnt[32], kstop[32], base[32]]
//                      [weights 32 x n_sample (only when records are requested)]
__global__ void __launch_bounds__(256)
k_march_secondary(TirField f, const float* __restrict__ origins, const int32_t* __restrict__ org_map,
                  const float* __restrict__ dirs, const int32_t* __restrict__ dir_map,
                  const uint8_t* __restrict__ active, int64_t n_rays, int n_dirs, int n_sample,
                  const float* __restrict__ z_vals, float t_stop, float* __restrict__ vis,
                  float* __restrict__ one_minus_acc, int32_t* __restrict__ rec_counter, int64_t rec_cap,
                  int32_t* __restrict__ rec_ray, float* __restrict__ rec_w, float* __restrict__ rec_xyz,
                  int32_t* __restrict__ ray_rec_off, int32_t* __restrict__ ray_rec_cnt,
                  unsigned long long* __restrict__ stats, int xcd_on, const int32_t* __restrict__ ray_ids,
                  const int32_t* __restrict__ n_ids_dev) {
    extern __shared__ __attribute__((aligned(16))) float w_lds[];
    // XCD-aware block order: XCD x (= blockIdx % 8) takes the x-th contiguous eighth of the ray blocks
    int64_t bid = blockIdx.x;
    if (xcd_on && (gridDim.x & 7) == 0) bid = (int64_t)(blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    // ray_ids: the rays of this launch are the listed pair ids (a compacted active list, *n_ids_dev of them); every
    // per-ray input and output is addressed by the pair id, the loop index only spreads the work
    if (ray_ids && n_ids_dev) n_rays = min(n_rays, (int64_t)max(*n_ids_dev, 0));
    if (bid * TIR_SEC_RPB >= n_rays) return;
    const int hl = threadIdx.x & 31;                 // lane within the half-wave
    const int hw = threadIdx.x >> 5;                 // half-wave within the block
    const int nz = (n_sample + 3) & ~3;
    float* zt = w_lds;
    float* ws = w_lds + nz + (threadIdx.x >> 6) * 256;            // this wave's gather scratch (16-B aligned)
    int* s_cnt = reinterpret_cast<int*>(w_lds + nz + 4 * 256);
    int* s_kstop = s_cnt + TIR_SEC_RPB;
    int* s_base = s_kstop + TIR_SEC_RPB;
    int* s_pid = s_base + TIR_SEC_RPB;
    float* w_all = w_lds + nz + 4 * 256 + 4 * TIR_SEC_RPB;        // [32][n_sample]
    for (int i = threadIdx.x; i < n_sample; i += blockDim.x) zt[i] = z_vals[i];
    __syncthreads();
    const bool want_rec = rec_counter != nullptr;
    unsigned n_gather = 0;

    for (int g = 0; g < TIR_SEC_RPB / 8; ++g) {
        const int rl = g * 8 + hw;
        const int64_t slot_id = bid * TIR_SEC_RPB + rl;
        const bool in_range = slot_id < n_rays;
        const int64_t ray = in_range ? (ray_ids ? (int64_t)ray_ids[slot_id] : slot_id) : 0;      // pair id
        const bool live = in_range && !(active && !active[ray]);
        int cnt = 0, k_stop = n_sample;    // k_stop: first sample this ray did not march (zero weight from there)
        if (__any(live)) {
            float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
            if (live) {
                const size_t oi = org_map ? (size_t)org_map[ray] : (n_dirs > 0 ? (size_t)((unsigned)ray / (unsigned)n_dirs) : (size_t)ray);
                const size_t di = dir_map ? (size_t)dir_map[ray] : (n_dirs > 0 ? (size_t)((unsigned)ray % (unsigned)n_dirs) : (size_t)ray);
#pragma unroll
                for (int a = 0; a < 3; ++a) { o[a] = origins[3 * oi + a]; d[a] = dirs[3 * di + a]; }
            }
            float* wl = w_all + (size_t)rl * n_sample;
            float T = 1.0f, acc = 0.0f;
            bool done = !live;                 // per half-wave (uniform inside a half)
            for (int k0 = 0; k0 < n_sample; k0 += 32) {   // wave-uniform loop: the gather below is wave-collective
                const int k = k0 + hl;
                float z = 0.f, x = 0.f, y = 0.f, zz = 0.f;
                bool valid = false;
                const bool on = !done && k < n_sample;
                if (on) {
                    z = zt[k];
                    float px = add_rn(o[0], mul_rn(d[0], z));
                    float py = add_rn(o[1], mul_rn(d[1], z));
                    float pz = add_rn(o[2], mul_rn(d[2], z));
                    valid = sample_valid(f, px, py, pz, x, y, zz);
                }
                const float sigma = wave_sigma(f, valid, x, y, zz, ws);
                float w = 0.0f, v = 1.0f;
                if (on) {
                    float dist = (k + 1 < n_sample) ? sub_rn(zt[k + 1], z) : 0.0f;
                    float alpha = 1.0f - exp_neg_fast(sigma * mul_rn(dist, f.distance_scale));
                    v = add_rn(sub_rn(1.0f, alpha), 1e-10f);
                    w = alpha;
                }
                float incl = scan_prod<32>(v, hl);
                float excl = shift_up1<32>(incl, hl);
                w = w * (T * excl);
                if (want_rec && on) wl[k] = w;
                acc += w;
                const unsigned long long m = __ballot(w > f.weight_thres);
                cnt += __popc((unsigned)(m >> ((threadIdx.x & 32) ? 32 : 0)));
                if (stats) n_gather += __popc((unsigned)(__ballot(valid) >> ((threadIdx.x & 32) ? 32 : 0)));
                T = T * __shfl(incl, 31, 32);
                if (!done && T < t_stop) { done = true; k_stop = min(k0 + 32, n_sample); }
                if (__all(done)) break;
            }
            acc = group_sum<32>(acc);
            if (live && hl == 0) {
                if (vis) vis[ray] = T;
                if (one_minus_acc) one_minus_acc[ray] = 1.0f - acc;
            }
        }
        if (hl == 0) {
            if (in_range && !live) {
                if (vis) vis[ray] = 0.0f;
                if (one_minus_acc) one_minus_acc[ray] = 0.0f;
            }
            s_cnt[rl] = live ? cnt : 0;
            s_kstop[rl] = k_stop;
            s_pid[rl] = in_range ? (int)ray : -1;
        }
    }
    if (stats && hl == 0 && n_gather) atomicAdd(stats, (unsigned long long)n_gather);
    if (!want_rec) return;
    __syncthreads();
    // one reservation for the block's 32 rays: inclusive scan of the counts in the first half-wave
    if (threadIdx.x < 32) {
        const int c = s_cnt[hl];
        int incl = c;
#pragma unroll
        for (int dd = 1; dd < 32; dd <<= 1) {
            int oth = __shfl_up(incl, dd, 32);
            if (hl >= dd) incl += oth;
        }
        const int total = __shfl(incl, 31, 32);
        int base = 0;
        if (hl == 31 && total > 0) base = atomicAdd(rec_counter, total);
        base = __shfl(base, 31, 32);
        const bool fits = (int64_t)base + total <= rec_cap;
        // rec_counter[1]: length of the fully written record prefix.  Reservations are handed out in increasing order,
        // so once one block does not fit no later one does: the written slots are exactly [0, max fitting base+total).
        if (hl == 31 && total > 0 && fits) atomicMax(rec_counter + 1, base + total);
        const int pid = s_pid[hl];
        if (pid >= 0) {
            ray_rec_off[pid] = base + incl - c;
            ray_rec_cnt[pid] = fits ? c : 0;
        }
        s_base[hl] = (fits && c > 0) ? base + incl - c : -1;
    }
    __syncthreads();
    for (int g = 0; g < TIR_SEC_RPB / 8; ++g) {
        const int rl = g * 8 + hw;
        int base = s_base[rl];
        if (base < 0) continue;                    // uniform inside the half-wave
        const int64_t ray = s_pid[rl];
        const size_t oi = org_map ? (size_t)org_map[ray] : (n_dirs > 0 ? (size_t)((unsigned)ray / (unsigned)n_dirs) : (size_t)ray);
        const size_t di = dir_map ? (size_t)dir_map[ray] : (n_dirs > 0 ? (size_t)((unsigned)ray % (unsigned)n_dirs) : (size_t)ray);
        float o[3], d[3];
#pragma unroll
        for (int a = 0; a < 3; ++a) { o[a] = origins[3 * oi + a]; d[a] = dirs[3 * di + a]; }
        const float* wl = w_all + (size_t)rl * n_sample;
        const int k_end = s_kstop[rl];
        const unsigned half_shift = (threadIdx.x & 32) ? 32 : 0;
        for (int q0 = 0; q0 < k_end; q0 += 32) {
            const int k = q0 + hl;
            float w = (k < k_end) ? wl[k] : 0.0f;
            const bool keep = w > f.weight_thres;
            const unsigned hm = (unsigned)(__ballot(keep) >> half_shift);
            if (keep) {
                const int slot = base + __popc(hm & ((1u << hl) - 1u));
                const float z = zt[k];
                rec_ray[slot] = (int32_t)ray;
                rec_w[slot] = w;
#pragma unroll
                for (int a = 0; a < 3; ++a) {
                    float p = add_rn(o[a], mul_rn(d[a], z));
                    rec_xyz[3 * (size_t)slot + a] = norm_coord(p, f.aabb_min[a], f.inv_aabb[a]);
                }
            }
            base += __popc(hm);
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K7 with LDS-staged line factors (north_star).  Persistent 512-thread blocks (two per CU): the block copies the three
// density lines (3 x R x 16 floats = 57.6 KB at R = 300) into LDS once and then walks batches of 64 rays -- 16 half-waves
// x 4 rays each.  A lane keeps the weights of its 4 x NSTEP samples in registers (n_sample <= 32 * NSTEP), so the
// record pass needs no weight staging and the block's LDS is lines + 8 KB of gather scratch: 2 blocks = 16 waves per CU
// (the plain kernel: 134 VGPRs -> 12 waves).  One record reservation (atomic) per 64 rays.  Same arithmetic as
// k_march_secondary in the same order: bit-identical results.
// ------------------------------------------------------------------------------------------------

// NT = 512 (two blocks per CU while lines + scratch fit 80 KB: R <= ~370) or 1024 (one block per CU, e.g. the 400^3 field
// of the ficus config: 76.8 KB of lines); RPB = NT / 8 rays per batch.
template <int C4, int NSTEP, int NT, bool REC>
__global__ void __launch_bounds__(NT, 4)
k_march_secondary_lds(TirField f, const float* __restrict__ origins_a, const int32_t* __restrict__ org_map_a,
                      const float* __restrict__ dirs_a, const int32_t* __restrict__ dir_map_a,
                      const uint8_t* __restrict__ active_a, int64_t n_rays, int n_dirs, int n_sample,
                      const float* __restrict__ z_vals, float t_stop, float* __restrict__ vis_a,
                      float* __restrict__ one_minus_acc_a, int32_t* __restrict__ rec_counter_a, int64_t rec_cap_a,
                      int32_t* __restrict__ rec_ray_a, float* __restrict__ rec_w_a, float* __restrict__ rec_xyz_a,
                      int32_t* __restrict__ ray_rec_off_a, int32_t* __restrict__ ray_rec_cnt_a,
                      unsigned long long* __restrict__ stats_a, int xcd_on, const int32_t* __restrict__ ray_ids_a,
                      const int32_t* __restrict__ n_ids_dev, int line_floats) {
    constexpr int RPB = NT / 8, NHW = NT / 32;                  // rays per batch, half-waves per block
    extern __shared__ __attribute__((aligned(16))) float sl_lds[];
    float* ll = sl_lds;                                         // [line 0 | line 1 | line 2]
    const int nz = (n_sample + 3) & ~3;
    float* zt = sl_lds + line_floats;
    float* ws = zt + nz + (threadIdx.x >> 6) * (64 * TIR_TAPREC); // this wave's gather scratch: 64 tap records
    int* s_cnt = reinterpret_cast<int*>(zt + nz + (NT / 64) * (64 * TIR_TAPREC));
    int* s_base = s_cnt + RPB;
    int* s_pid = s_base + RPB;
    __shared__ int s_wtot[2], s_bb, s_fits;
    // The seven record-phase arguments are used once per batch of 64 / 128 rays, but as kernel arguments they would sit in 14
    // SGPRs through the whole march -- whose hot loop already spills ~100 SGPRs to VGPR lanes and pays a v_readlane (VALU, the
    // binding pipe) for each use.  They are parked in LDS here and read back (as per-lane values) where the records are written.
    // (the per-ray arguments -- pair list, origins, directions, outputs: used four times per batch -- are parked with them)
    struct Parked { int32_t* rec_counter; int64_t rec_cap; int32_t* rec_ray; float* rec_w; float* rec_xyz; int32_t* ray_rec_off; int32_t* ray_rec_cnt;
                    const float* origins; const int32_t* org_map; const float* dirs; const int32_t* dir_map; const uint8_t* active;
                    float* vis; float* one_minus_acc; const int32_t* ray_ids; unsigned long long* stats; float occ_lo[3], occ_hi[3]; };
    __shared__ Parked s_park;
    __shared__ __attribute__((aligned(16))) float s_setup[NT / 64][8][12];     // per wave: set-up slots of its 8 rays of a batch
    // REC = false: the visibility-only instantiation (a C5 view's launches) carries no weight registers, counts or record phase
    if (threadIdx.x == 0) s_park = Parked{rec_counter_a, rec_cap_a, rec_ray_a, rec_w_a, rec_xyz_a, ray_rec_off_a, ray_rec_cnt_a,
                                          origins_a, org_map_a, dirs_a, dir_map_a, active_a, vis_a, one_minus_acc_a, ray_ids_a, stats_a,
                                          {f.occ_lo[0], f.occ_lo[1], f.occ_lo[2]}, {f.occ_hi[0], f.occ_hi[1], f.occ_hi[2]}};
    const bool stats = stats_a != nullptr;
    {   // stage the line factors (coalesced 16-B copies; lines are [R][16] rows, contiguous)
        int off = 0;
        for (int i = 0; i < 3; ++i) {
            const int nf = f.grid[2 - i] * (C4 * 4);
            for (int e = threadIdx.x * 4; e < nf; e += NT * 4)
                *reinterpret_cast<float4*>(ll + off + e) = *reinterpret_cast<const float4*>(f.dline[i] + e);
            off += nf;
        }
        for (int i = threadIdx.x; i < n_sample; i += NT) zt[i] = z_vals[i];
    }
    __syncthreads();
    if (ray_ids_a && n_ids_dev) n_rays = min(n_rays, (int64_t)max(*n_ids_dev, 0));
    const int64_t n_batches = (n_rays + RPB - 1) / RPB;
    const XcdRange xr = xcd_range(n_batches, 1, xcd_on != 0);
    const int hl = threadIdx.x & 31, hw = threadIdx.x >> 5;     // lane in the half-wave, half-wave in the block (0..15)
    const unsigned half_shift = (threadIdx.x & 32) ? 32 : 0;
    unsigned n_gather = 0;

    // this wave's 8 rays of a batch (4 groups x 2 half-waves): set-up slots of 12 floats [o | d | t_in t_out | pair id, flags]
    float* const sw = &s_setup[threadIdx.x >> 6][0][0];
    const float* const sh = sw + ((threadIdx.x >> 5) & 1) * 12;   // group g of this half-wave: sh + 24 g

    for (int64_t batch = xr.first; batch < xr.end; batch += xr.stride) {
        float wreg[REC ? 4 : 1][REC ? NSTEP : 1];
        int cnts[4];
        // Per-ray set-up ONCE per batch with one lane per ray (lanes 0..7 of the wave), handed to the half-waves through LDS:
        // pair id -> (point, direction) division, the six loads and the slab test were ~160 VALU instructions executed by all
        // 64 lanes for 2 rays, four times per batch (17 % of the kernel's VALU work), and again in the record pass.
        if ((threadIdx.x & 63) < 8) {
            const int q = threadIdx.x & 7;
            const int rl = (q >> 1) * NHW + ((threadIdx.x >> 6) << 1) + (q & 1);
            const float* const origins = s_park.origins; const int32_t* const org_map = s_park.org_map;
            const float* const dirs = s_park.dirs; const int32_t* const dir_map = s_park.dir_map;
            const uint8_t* const active = s_park.active; const int32_t* const ray_ids = s_park.ray_ids;
            const int64_t slot_id = batch * RPB + rl;
            const bool in_range = slot_id < n_rays;
            const int64_t ray = in_range ? (ray_ids ? (int64_t)ray_ids[slot_id] : slot_id) : 0;
            const bool live = in_range && !(active && !active[ray]);
            float o[3] = {0.f, 0.f, 0.f}, d[3] = {0.f, 0.f, 0.f};
            if (live) {
                // pair id -> (point, direction) with 32-bit division (pair ids are < 2^31, checked at launch; the 64-bit
                // form costs ~200 instructions per ray)
                const unsigned ru = (unsigned)ray, nd = (unsigned)n_dirs;
                const size_t oi = org_map ? (size_t)org_map[ray] : (n_dirs > 0 ? (size_t)(ru / nd) : (size_t)ray);
                const size_t di = dir_map ? (size_t)dir_map[ray] : (n_dirs > 0 ? (size_t)(ru % nd) : (size_t)ray);
#pragma unroll
                for (int a = 0; a < 3; ++a) { o[a] = origins[3 * oi + a]; d[a] = dirs[3 * di + a]; }
            }
            float t_in, t_out;                                  // where the ray can meet occupied space at all
            occ_t_range(s_park.occ_lo, s_park.occ_hi, o, d, t_in, t_out);
            float4* const dst = reinterpret_cast<float4*>(sw + q * 12);
            dst[0] = make_float4(o[0], o[1], o[2], d[0]);
            dst[1] = make_float4(d[1], d[2], t_in, t_out);
            dst[2] = make_float4(__int_as_float((int)ray), __int_as_float((in_range ? 1 : 0) | (live ? 2 : 0)), 0.f, 0.f);
        }
        __builtin_amdgcn_wave_barrier();      // LDS ops of one wave complete in order; keep the compiler from reordering
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            float* const vis = s_park.vis; float* const one_minus_acc = s_park.one_minus_acc;
            const int rl = g * NHW + hw;
            const float4 s0 = *reinterpret_cast<const float4*>(sh + 24 * g);
            const float4 s1 = *reinterpret_cast<const float4*>(sh + 24 * g + 4);
            const float4 s2 = *reinterpret_cast<const float4*>(sh + 24 * g + 8);
            const int64_t ray = (int64_t)__float_as_int(s2.x);
            const bool in_range = (__float_as_int(s2.y) & 1) != 0, live = (__float_as_int(s2.y) & 2) != 0;
            int cnt = 0;
            if constexpr (REC) {
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) wreg[g][st] = 0.0f;
            }
            if (__any(live)) {
                const float o[3] = {s0.x, s0.y, s0.z}, d[3] = {s0.w, s1.x, s1.y};
                const float t_in = s1.z, t_out = s1.w;
                float T = 1.0f, acc = 0.0f;
                bool done = !live;
                bool all_done = false;
#pragma unroll
                for (int st = 0; st < NSTEP; ++st) {
                    const int k = st * 32 + hl;
                    if (!all_done && st * 32 < n_sample) {          // wave-uniform
                        // both rays of the wave are past (or not yet at) the occupied box for this whole step: every
                        // sample would be culled -- T, the sums and the record count stay as they are
                        const bool touches = !done && zt[st * 32] <= t_out && zt[min(st * 32 + 31, n_sample - 1)] >= t_in;
                        if (!__any(touches)) continue;
                        float z = 0.f, x = 0.f, y = 0.f, zz = 0.f;
                        bool valid = false;
                        const bool on = !done && k < n_sample;
                        if (on) {
                            z = zt[k];
                            float px = add_rn(o[0], mul_rn(d[0], z));
                            float py = add_rn(o[1], mul_rn(d[1], z));
                            float pz = add_rn(o[2], mul_rn(d[2], z));
                            valid = sample_valid(f, px, py, pz, x, y, zz);
                        }
                        const float sigma = wave_sigma_lds<C4>(f, ll, valid, x, y, zz, ws);
                        float w = 0.0f, v = 1.0f;
                        if (on) {
                            float dist = (k + 1 < n_sample) ? sub_rn(zt[k + 1], z) : 0.0f;
                            float alpha = 1.0f - exp_neg_fast(sigma * mul_rn(dist, f.distance_scale));
                            v = add_rn(sub_rn(1.0f, alpha), 1e-10f);
                            w = alpha;
                        }
                        float incl = scan_prod<32>(v, hl);
                        float excl = shift_up1<32>(incl, hl);
                        w = w * (T * excl);
                        acc += w;
                        if constexpr (REC) {
                            wreg[g][st] = on ? w : 0.0f;
                            const unsigned long long m = __ballot(w > f.weight_thres);
                            cnt += __popc((unsigned)(m >> half_shift));
                        }
                        if (stats) n_gather += __popc((unsigned)(__ballot(valid) >> half_shift));
                        T = T * __shfl(incl, 31, 32);
                        if (!done && T < t_stop) done = true;
                        all_done = __all(done);
                    }
                }
                acc = group_sum<32>(acc);
                if (live && hl == 0) {
                    if (vis) vis[ray] = T;
                    if (one_minus_acc) one_minus_acc[ray] = 1.0f - acc;
                }
            }
            cnts[g] = live ? cnt : 0;
            if (hl == 0) {
                if (in_range && !live) {
                    if (vis) vis[ray] = 0.0f;
                    if (one_minus_acc) one_minus_acc[ray] = 0.0f;
                }
                s_cnt[rl] = cnts[g];
                s_pid[rl] = in_range ? (int)ray : -1;
            }
        }
        if constexpr (!REC) continue;
        __syncthreads();
        // one reservation for the batch: inclusive scan of the RPB ray counts (one or two waves), one atomic
        int c = 0, incl = 0;
        if (threadIdx.x < RPB) {
            const int lane = threadIdx.x & 63;
            c = s_cnt[threadIdx.x];
            incl = c;
#pragma unroll
            for (int dd = 1; dd < 64; dd <<= 1) {
                int oth = __shfl_up(incl, dd, 64);
                if (lane >= dd) incl += oth;
            }
            if (lane == 63) s_wtot[threadIdx.x >> 6] = incl;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            int32_t* rec_counter = s_park.rec_counter;
            const int64_t rec_cap = s_park.rec_cap;
            const int total = s_wtot[0] + (RPB > 64 ? s_wtot[1] : 0);
            int base = 0;
            if (total > 0) base = atomicAdd(rec_counter, total);
            const bool fits = (int64_t)base + total <= rec_cap;
            // rec_counter[1]: length of the fully written record prefix (reservations are handed out in increasing order)
            if (total > 0 && fits) atomicMax(rec_counter + 1, base + total);
            s_bb = base; s_fits = fits ? 1 : 0;
        }
        __syncthreads();
        if (threadIdx.x < RPB) {
            if (RPB > 64 && threadIdx.x >= 64) incl += s_wtot[0];
            const int pid = s_pid[threadIdx.x];
            const bool fits = s_fits != 0;
            if (pid >= 0) {
                s_park.ray_rec_off[pid] = s_bb + incl - c;
                s_park.ray_rec_cnt[pid] = fits ? c : 0;
            }
            s_base[threadIdx.x] = (fits && c > 0) ? s_bb + incl - c : -1;
        }
        __syncthreads();
        int32_t* const rec_ray = s_park.rec_ray;
        float* const rec_w = s_park.rec_w;
        float* const rec_xyz = s_park.rec_xyz;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rl = g * NHW + hw;
            int base = s_base[rl];
            if (base < 0) continue;                    // uniform inside the half-wave
            const int64_t ray = s_pid[rl];
            const float4 s0 = *reinterpret_cast<const float4*>(sh + 24 * g);          // this batch's set-up slot of the ray
            const float2 s1 = *reinterpret_cast<const float2*>(sh + 24 * g + 4);
            const float o[3] = {s0.x, s0.y, s0.z}, d[3] = {s0.w, s1.x, s1.y};
#pragma unroll
            for (int st = 0; st < NSTEP; ++st) {
                const int k = st * 32 + hl;
                const float w = wreg[REC ? g : 0][REC ? st : 0];
                const bool keep = w > f.weight_thres;
                const unsigned hm = (unsigned)(__ballot(keep) >> half_shift);
                if (keep) {
                    const int slot = base + __popc(hm & ((1u << hl) - 1u));
                    const float z = zt[k];
                    rec_ray[slot] = (int32_t)ray;
                    rec_w[slot] = w;
#pragma unroll
                    for (int a = 0; a < 3; ++a) {
                        float p = add_rn(o[a], mul_rn(d[a], z));
                        rec_xyz[3 * (size_t)slot + a] = norm_coord(p, f.aabb_min[a], f.inv_aabb[a]);
                    }
                }
                base += __popc(hm);
            }
        }
        // the next batch's s_cnt / s_pid writes come after this batch's readers: s_base / s_bb are only rewritten behind
        // the next batch's first __syncthreads, s_pid[rl] / s_cnt[rl] belong to the half-wave that reads them here
    }
    if (stats && hl == 0 && n_gather) atomicAdd(s_park.stats, (unsigned long long)n_gather);
}


extern "C" int tir_march_secondary_fwd(const TirField* f, const float* origins, const int32_t* org_map,
                                       const float* dirs, const int32_t* dir_map, const uint8_t* active,
                                       int64_t n_rays, int32_t n_dirs, int32_t n_sample, const float* z_vals,
                                       float t_stop, float* vis, float* one_minus_acc,
                                       int32_t* rec_counter, int64_t rec_cap, int32_t* rec_ray,
                                       float* rec_w, float* rec_xyz, int32_t* ray_rec_off,
                                       int32_t* ray_rec_cnt, unsigned long long* stats, void* stream) {
    return tir_march_secondary_ids_fwd(f, origins, org_map, dirs, dir_map, active, n_rays, n_dirs, n_sample, z_vals, t_stop,
                                       vis, one_minus_acc, rec_counter, rec_cap, rec_ray, rec_w, rec_xyz, ray_rec_off,
                                       ray_rec_cnt, stats, nullptr, nullptr, stream);
}

extern "C" int tir_march_secondary_ids_fwd(const TirField* f, const float* origins, const int32_t* org_map,
                                           const float* dirs, const int32_t* dir_map, const uint8_t* active,
                                           int64_t n_rays, int32_t n_dirs, int32_t n_sample, const float* z_vals,
                                           float t_stop, float* vis, float* one_minus_acc,
                                           int32_t* rec_counter, int64_t rec_cap, int32_t* rec_ray,
                                           float* rec_w, float* rec_xyz, int32_t* ray_rec_off,
                                           int32_t* ray_rec_cnt, unsigned long long* stats, const int32_t* ray_ids,
                                           const int32_t* n_ids_dev, void* stream) {
    if (!f || n_rays < 0 || n_sample <= 0) return TIR_ERR_ARG;
    if ((ray_ids == nullptr) != (n_ids_dev == nullptr)) return TIR_ERR_ARG;
    if (n_sample > TIR_SEC_MAX_SAMPLES) return TIR_ERR_UNSUPPORTED;
    if (n_rays == 0) return TIR_OK;
    if (!origins || !dirs || !z_vals) return TIR_ERR_ARG;
    if (rec_counter && (!rec_ray || !rec_w || !rec_xyz || !ray_rec_off || !ray_rec_cnt || rec_cap < 0)) return TIR_ERR_ARG;
    if (n_rays >= (int64_t)1 << 31) return TIR_ERR_UNSUPPORTED;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    if (!tir_occ_index_ok(f) || !tir_plane_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    const int xcd_on = tir_xcd_mapping(f);
    // LDS-staged line factors: 16 density components, <= 96 samples per ray, lines + scratch within half a CU's LDS
    {
        const int64_t line_floats = (int64_t)(f->grid[0] + f->grid[1] + f->grid[2]) * f->n_dcomp;
        const size_t fixed = ((size_t)line_floats + ((n_sample + 3) & ~3)) * sizeof(float);
        const size_t lds512 = fixed + (8 * 64 * TIR_TAPREC + 3 * 64) * sizeof(float), lds1024 = fixed + (16 * 64 * TIR_TAPREC + 3 * 128) * sizeof(float);
        if (f->tune_lds_lines != 2 && f->n_dcomp == 16 && n_sample <= 96 && f->grid[0] < 65536 && f->grid[1] < 65536 && f->grid[2] < 65536 && lds1024 + 8192 <= 158 * 1024) {
            const bool small = lds512 + 4096 <= 80 * 1024;          // two 512-thread blocks per CU (4 KB: the kernel's static LDS), else one of 1024
            const int rpb = small ? 64 : 128;
            const int64_t n_batches = (n_rays + rpb - 1) / rpb;
            unsigned nblk = (unsigned)std::min<int64_t>(n_batches, small ? 2 * 256 : 256);
            if (xcd_on) nblk = (nblk + 7) / 8 * 8;
            // (the visibility-only launches of a C5 view run the instantiation without weight registers, counts and record phase)
#define TIR_LAUNCH_SEC_LDS(NT_, REC_, LDS_, CAP_)                                                                                          \
            do {                                                                                                                          \
                if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_march_secondary_lds<4, 3, NT_, REC_>), CAP_)) return rc; \
                hipLaunchKernelGGL((k_march_secondary_lds<4, 3, NT_, REC_>), dim3(nblk), dim3(NT_), LDS_, tir_stream(stream),             \
                                   *f, origins, org_map, dirs, dir_map, active, n_rays, n_dirs, n_sample, z_vals, t_stop, vis,            \
                                   one_minus_acc, rec_counter, rec_cap, rec_ray, rec_w, rec_xyz, ray_rec_off, ray_rec_cnt, stats,         \
                                   xcd_on, ray_ids, n_ids_dev, (int)line_floats);                                                         \
            } while (0)
            if (small) { if (rec_counter) TIR_LAUNCH_SEC_LDS(512, true, lds512, 80 * 1024); else TIR_LAUNCH_SEC_LDS(512, false, lds512, 80 * 1024); }
            else       { if (rec_counter) TIR_LAUNCH_SEC_LDS(1024, true, lds1024, 150 * 1024); else TIR_LAUNCH_SEC_LDS(1024, false, lds1024, 150 * 1024); }
#undef TIR_LAUNCH_SEC_LDS
            TIR_CHECK_LAUNCH();
            return TIR_OK;
        }
    }
    size_t lds = ((size_t)((n_sample + 3) & ~3) + 4 * 256 + 4 * TIR_SEC_RPB +
                  (rec_counter ? (size_t)TIR_SEC_RPB * n_sample : 0)) * sizeof(float);
    unsigned nblk = (unsigned)((n_rays + TIR_SEC_RPB - 1) / TIR_SEC_RPB);
    if (xcd_on) nblk = (nblk + 7) / 8 * 8;
    hipLaunchKernelGGL(k_march_secondary, dim3(nblk), dim3(256), lds, tir_stream(stream),
                       *f, origins, org_map, dirs, dir_map, active, n_rays, n_dirs, n_sample, z_vals, t_stop, vis,
                       one_minus_acc, rec_counter, rec_cap, rec_ray, rec_w, rec_xyz, ray_rec_off, ray_rec_cnt, stats, xcd_on,
                       ray_ids, n_ids_dev);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// indirect[p] = sum_k w * rgb over ray p's records, in sample order (models/relight_utils.py:832)
__global__ void __launch_bounds__(256)
k_accumulate_records(const int32_t* __restrict__ off, const int32_t* __restrict__ cnt,
                     const float* __restrict__ rec_w, const float* __restrict__ rec_rgb, int64_t n,
                     float* __restrict__ indirect) {
    int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    float c0 = 0, c1 = 0, c2 = 0;
    const int b = off[p], e = b + cnt[p];
    for (int i = b; i < e; ++i) {
        float w = rec_w[i];
        c0 = fmaf(w, rec_rgb[3 * (size_t)i], c0);
        c1 = fmaf(w, rec_rgb[3 * (size_t)i + 1], c1);
        c2 = fmaf(w, rec_rgb[3 * (size_t)i + 2], c2);
    }
    indirect[3 * p] = c0; indirect[3 * p + 1] = c1; indirect[3 * p + 2] = c2;
}

extern "C" int tir_accumulate_records(const int32_t* ray_rec_off, const int32_t* ray_rec_cnt,
                                      const float* rec_w, const float* rec_rgb, int64_t n_rays,
                                      float* indirect, void* stream) {
    if (n_rays < 0) return TIR_ERR_ARG;
    if (n_rays == 0) return TIR_OK;
    if (!ray_rec_off || !ray_rec_cnt || !indirect) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_accumulate_records, dim3((unsigned)((n_rays + 255) / 256)), dim3(256), 0,
                       tir_stream(stream), ray_rec_off, ray_rec_cnt, rec_w, rec_rgb, n_rays, indirect);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
