// Shared device helpers for libtensoir_hip (gfx950 / CDNA4 only; wave = 64 lanes).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "tensoir_hip.h"

#define TIR_WAVE 64

#define TIR_CHECK_LAUNCH()                                   \
    do {                                                     \
        hipError_t e__ = hipGetLastError();                  \
        if (e__ != hipSuccess) return -(int)e__;             \
    } while (0)

static inline hipStream_t tir_stream(void* s) { return reinterpret_cast<hipStream_t>(s); }

// XCD-aware work mapping switch (see tir::xcd_range).  Measured on the bench scene (same box, A/B): contiguous
// per-XCD ranges leave the appearance gather unchanged (0.615 ms) and slow the secondary march by 7 % (0.509 -> 0.544 ms:
// the eighths of the image are not equally expensive, and the 70 MB field is Infinity-Cache resident anyway), so the
// interleaved order is the default; TirField::tune_xcd_order = 1 enables the partitioned order.  (A per-call option carried
// by the descriptor: the library reads no environment variable.)
static inline int tir_xcd_mapping(const TirField* f) { return f->tune_xcd_order == 1 ? 1 : 0; }

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-DEVICE property of a kernel: remember (kernel, device ordinal) pairs,
// so that a second GPU driven from the same process gets the attribute too, and hand the runtime's error code back.
#include <mutex>
#include <set>
#include <utility>
static inline int tir_allow_dynamic_lds(const void* kernel, int bytes) {
    static std::mutex mu;
    static std::set<std::pair<const void*, int>> done;
    int dev = 0;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) return -(int)e;
    std::lock_guard<std::mutex> lock(mu);
    if (done.count({kernel, dev})) return TIR_OK;
    e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return -(int)e;
    done.insert({kernel, dev});
    return TIR_OK;
}

// the occupancy neighbourhood bytes are indexed with 32-bit arithmetic (tir::occupancy_hit)
static inline bool tir_occ_index_ok(const TirField* f) {
    if (!f->occ_nbr) return true;
    const int64_t W = f->occ_dim[0], H = f->occ_dim[1], D = f->occ_dim[2];
    return W > 0 && H > 0 && D > 0 && W + 1 < (1 << 24) && (H + 1) * (D + 1) < (1 << 24) && (W + 1) * (H + 1) * (D + 1) < ((int64_t)1 << 31);
}

// the density gathers address plane taps with 32-bit BYTE offsets (tir::density_chunk_impl): every plane must be < 4 GB
static inline bool tir_plane_index_ok(const TirField* f) {
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if ((int64_t)f->grid[i] * f->grid[j] * f->n_dcomp * 4 >= ((int64_t)1 << 32) || (int64_t)f->grid[i] * f->n_dcomp * 4 >= (1 << 24) || (int64_t)f->grid[j] * f->n_dcomp * 4 >= (1 << 24)) return false;
    return true;
}

// the fp16 appearance gathers (k_vm_app_h16, k_indirect_fused) address plane taps with 32-bit ELEMENT offsets
static inline bool tir_app_index_ok(const TirField* f) {
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if ((int64_t)f->grid[i] * f->grid[j] * f->n_acomp >= ((int64_t)1 << 31)) return false;
    for (int i = 0; i < 3; ++i)      // 24-bit index multiplies: every index and every row pitch in elements below 2^24
        if ((int64_t)f->grid[i] * f->n_acomp >= (1 << 24)) return false;
    return true;
}

// matMode / vecMode of the reference (models/tensorBase_rotated_lights.py:398-399)
__device__ __constant__ const int kMat0[3] = {0, 0, 1};
__device__ __constant__ const int kMat1[3] = {1, 2, 2};
__device__ __constant__ const int kVec[3] = {2, 1, 0};

namespace tir {

// ---- exactly-rounded fp32 steps where the reference's decision points (floor / compare) depend on
// them; everything else may contract to FMA.
// `#pragma clang fp contract(off)` inside each helper, NOT __fmul_rn / __fadd_rn: hipcc lowers those intrinsics to plain
// `a * b` / `a + b` that still carry the `contract` flag, and add_rn(o, mul_rn(d, z)) became ONE v_fma_f32 (rounds 1-4 shipped
// that).  The reference rounds the product and the sum separately (rays_o + rays_d * z: two ATen kernels,
// models/tensorBase_rotated_lights.py:722); a sample position that differs in its last bit flips the in-box test of a sample
// sitting exactly on the box -- after shrink() the box IS aligned with voxel boundaries and every ray's first sample sits on
// it -- and moved rendered maps of a trained checkpoint by up to 1.6e-3 (profiles/r05_fma_contraction.json).  An operation
// without the flag cannot be fused with its neighbours, whatever the caller's contraction mode is.
__device__ __forceinline__ float mul_rn(float a, float b) {
#pragma clang fp contract(off)
    return a * b;
}
__device__ __forceinline__ float add_rn(float a, float b) {
#pragma clang fp contract(off)
    return a + b;
}
__device__ __forceinline__ float sub_rn(float a, float b) {
#pragma clang fp contract(off)
    return a - b;
}

// normalize_coord: (x - aabb0) * invaabbSize - 1   (models/tensorBase_rotated_lights.py:640-641)
__device__ __forceinline__ float norm_coord(float x, float mn, float inv) {
    return sub_rn(mul_rn(sub_rn(x, mn), inv), 1.0f);
}

// grid_sample unnormalise, align_corners=True: ((x+1)/2)*(size-1)  (models/relight_utils.py:64-65)
__device__ __forceinline__ float unnorm(float x, int size) {
    return mul_rn(mul_rn(add_rn(x, 1.0f), 0.5f), (float)(size - 1));
}

struct Tap1 {   // linear interpolation along one axis, zero padding
    int i0, i1;     // clamped indices (always loadable)
    float w0, w1;   // weights, 0 for out-of-range taps
    float t;        // fractional position
    float m0, m1;   // in-range masks (1/0) for derivative taps
};

__device__ __forceinline__ Tap1 make_tap(float x, int size) {
    Tap1 r;
    float ix = unnorm(x, size);
    float f0 = floorf(ix);
    r.t = ix - f0;
    int i0 = (int)f0, i1 = i0 + 1;
    bool ok0 = (i0 >= 0) & (i0 < size);
    bool ok1 = (i1 >= 0) & (i1 < size);
    r.m0 = ok0 ? 1.0f : 0.0f;
    r.m1 = ok1 ? 1.0f : 0.0f;
    r.w0 = ok0 ? (1.0f - r.t) : 0.0f;
    r.w1 = ok1 ? r.t : 0.0f;
    r.i0 = min(max(i0, 0), size - 1);
    r.i1 = min(max(i1, 0), size - 1);
    return r;
}

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// a * b for a, b < 2^24 on the full-rate 24-bit multiplier.  Opaque to the optimiser on purpose: written as __umul24 the
// product is re-associated with the following constant factor into one v_mul_lo_u32 (quarter rate).
__device__ __forceinline__ unsigned mul_u24(unsigned a, unsigned b) {
    unsigned r;
    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// sum_c bilinear(plane)[c] * linear(line)[c] for one VM component group
// (one term of compute_densityfeature, models/tensoRF_rotated_lights.py:103-108)
template <int C4>
__device__ __forceinline__ float plane_line_dot(const float* __restrict__ plane,
                                                const float* __restrict__ line, int H, int W,
                                                int R, float u, float v, float w) {
    Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
    const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
    const float* p00 = plane + ((size_t)ty.i0 * W + tx.i0) * (C4 * 4);
    const float* p01 = plane + ((size_t)ty.i0 * W + tx.i1) * (C4 * 4);
    const float* p10 = plane + ((size_t)ty.i1 * W + tx.i0) * (C4 * 4);
    const float* p11 = plane + ((size_t)ty.i1 * W + tx.i1) * (C4 * 4);
    const float* l0 = line + (size_t)tl.i0 * (C4 * 4);
    const float* l1 = line + (size_t)tl.i1 * (C4 * 4);
    float acc = 0.0f;
#pragma unroll
    for (int c = 0; c < C4; ++c) {
        float4 a = ld4(p00 + 4 * c), b = ld4(p01 + 4 * c), cc = ld4(p10 + 4 * c), d = ld4(p11 + 4 * c);
        float4 e = ld4(l0 + 4 * c), g = ld4(l1 + 4 * c);
        float px = fmaf(d.x, w11, fmaf(cc.x, w10, fmaf(b.x, w01, a.x * w00)));
        float py = fmaf(d.y, w11, fmaf(cc.y, w10, fmaf(b.y, w01, a.y * w00)));
        float pz = fmaf(d.z, w11, fmaf(cc.z, w10, fmaf(b.z, w01, a.z * w00)));
        float pw = fmaf(d.w, w11, fmaf(cc.w, w10, fmaf(b.w, w01, a.w * w00)));
        acc = fmaf(px, fmaf(g.x, tl.w1, e.x * tl.w0), acc);
        acc = fmaf(py, fmaf(g.y, tl.w1, e.y * tl.w0), acc);
        acc = fmaf(pz, fmaf(g.z, tl.w1, e.z * tl.w0), acc);
        acc = fmaf(pw, fmaf(g.w, tl.w1, e.w * tl.w0), acc);
    }
    return acc;
}

// density feature at a normalised point (compute_densityfeature, models/tensoRF_rotated_lights.py:95-110)
template <int C4>
__device__ __forceinline__ float density_feature(const TirField& f, float x, float y, float z) {
    const float p[3] = {x, y, z};
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int m0 = (i == 2) ? 1 : 0, m1 = (i == 0) ? 1 : 2, vi = 2 - i;
        acc += plane_line_dot<C4>(f.dplane[i], f.dline[i], f.grid[m1], f.grid[m0], f.grid[vi],
                                  p[m0], p[m1], p[vi]);
    }
    return acc;
}

__device__ __forceinline__ float density_feature_dyn(const TirField& f, float x, float y, float z) {
    switch (f.n_dcomp) {
        case 16: return density_feature<4>(f, x, y, z);
        case 8:  return density_feature<2>(f, x, y, z);
        case 32: return density_feature<8>(f, x, y, z);
        default: return density_feature<1>(f, x, y, z);   // n_dcomp == 4
    }
}

// torch softplus (beta 1, threshold 20): log1p(exp(x)).  The library log1pf is ~110 VALU instructions (double-float
// arithmetic) and sat in every step of the VALU-bound march kernels; here log1p(e) = log(u) * e / (u - 1) with u = fl(1 + e)
// (the rounding of 1 + e cancels in the ratio; u == 1 means e < 2^-24 and log1p(e) = e to that precision), the logarithm
// on the transcendental unit (v_log_f32, 1 ulp): <= 3.5e-7 relative error over x in [-40, 20] against fp64 (torch's own
// fp32 softplus: 1.2e-7), ~12 instructions.
__device__ __forceinline__ float softplus20(float x) {
    if (x > 20.0f) return x;
    const float e = expf(x);
    const float u = 1.0f + e;
    const float d = u - 1.0f;
    const float lg = __builtin_amdgcn_logf(u) * 0.69314718055994531f;
    return (d == 0.0f) ? e : lg * (e * __builtin_amdgcn_rcpf(d));
}

// feature2density (models/tensorBase_rotated_lights.py:813-817)
__device__ __forceinline__ float feature2density(const TirField& f, float feat) {
    if (f.act == 1) return fmaxf(feat, 0.0f);
    return softplus20(feat + f.density_shift);
}

// the same through the library's log1pf: occupancy-mask building (tir_dense_alpha) thresholds alpha against 1e-3 and is
// compared voxel by voxel with masks the reference built -- off the hot path, it keeps the last-ulp behaviour it had
__device__ __forceinline__ float feature2density_ref(const TirField& f, float feat) {
    if (f.act == 1) return fmaxf(feat, 0.0f);
    const float x = feat + f.density_shift;
    return (x > 20.0f) ? x : log1pf(expf(x));
}

// AlphaGridMask.sample_alpha(...) > 0  (models/tensorBase_rotated_lights.py:112-119, :893-894).
// The volume is 0/1 and trilinear weights are >= 0, so "interpolated value > 0" == "some corner with
// non-zero weight is set".  occ_nbr holds, per base voxel (x0,y0,z0) in [-1,W-1]x[-1,H-1]x[-1,D-1], one byte
// whose bit dx+2dy+4dz is the occupancy of corner (x0+dx,y0+dy,z0+dz) (0 outside the grid): ONE scattered
// byte load per sample instead of eight word loads.
__device__ __forceinline__ bool occupancy_hit(const TirField& f, float px, float py, float pz) {
    const int W = f.occ_dim[0], H = f.occ_dim[1], D = f.occ_dim[2];
    float qx = sub_rn(mul_rn(sub_rn(px, f.occ_aabb_min[0]), f.occ_inv[0]), 1.0f);
    float qy = sub_rn(mul_rn(sub_rn(py, f.occ_aabb_min[1]), f.occ_inv[1]), 1.0f);
    float qz = sub_rn(mul_rn(sub_rn(pz, f.occ_aabb_min[2]), f.occ_inv[2]), 1.0f);
    float ix = unnorm(qx, W), iy = unnorm(qy, H), iz = unnorm(qz, D);
    float fx = floorf(ix), fy = floorf(iy), fz = floorf(iz);
    // clamp before the int conversion (far-away points would overflow); out-of-range cells hold 0 anyway
    fx = fminf(fmaxf(fx, -2.0f), (float)W); fy = fminf(fmaxf(fy, -2.0f), (float)H); fz = fminf(fmaxf(fz, -2.0f), (float)D);
    const int x0 = (int)fx, y0 = (int)fy, z0 = (int)fz;
    if ((x0 < -1) | (x0 >= W) | (y0 < -1) | (y0 >= H) | (z0 < -1) | (z0 >= D)) return false;
    // 32-bit index (launchers check (W+1)(H+1)(D+1) < 2^31 and (D+1)(H+1), W+1 < 2^24, tir_occ_index_ok): two full-rate 24-bit
    // multiplies instead of two 64-bit ones (the 32-bit v_mul_lo_u32 issues at a quarter of the rate)
    const uint32_t b = f.occ_nbr[mul_u24(mul_u24((unsigned)(z0 + 1), (unsigned)(H + 1)) + (unsigned)(y0 + 1), (unsigned)(W + 1)) + (unsigned)(x0 + 1)];
    // corners with a zero interpolation weight (fraction exactly 0) do not count
    const uint32_t mx = (ix - fx) > 0.0f ? 0xFFu : 0x55u;
    const uint32_t my = (iy - fy) > 0.0f ? 0xFFu : 0x33u;
    const uint32_t mz = (iz - fz) > 0.0f ? 0xFFu : 0x0Fu;
    return (b & mx & my & mz) != 0u;
}

// in-bbox test of sample_ray / sample_ray_equally: ~((aabb0 > p) | (p > aabb1)).any()
__device__ __forceinline__ bool in_bbox(const TirField& f, float px, float py, float pz) {
    return !((f.aabb_min[0] > px) | (px > f.aabb_max[0]) | (f.aabb_min[1] > py) |
             (py > f.aabb_max[1]) | (f.aabb_min[2] > pz) | (pz > f.aabb_max[2]));
}

// ---- the density gather's arithmetic, written for the VALU-issue-bound march kernels (ISA of a 16-sample gather pass before /
// after: profiles/r04_march_isa.txt).
// * make_tap_q = make_tap's indices and weights (bit for bit, for every input incl. out-of-range points) with the range tests
//   as one unsigned compare each and the clamps as v_med3_i32: 6 instead of 10 instructions per axis.
// * the three axes' taps are computed once per sample and shared by the three planes / lines that use them.
// * tap addresses are 32-bit BYTE offsets from a wave-uniform base (global_load ... v_off, s[base]: no 64-bit address
//   arithmetic per lane; planes are < 4 GB, tir_plane_index_ok).
// * the bilinear / linear interpolations run two channels per instruction (v_pk_mul_f32 / v_pk_fma_f32: the same IEEE
//   operations in the same order per channel, so every result is bit-identical to the scalar chain); the plane x line
//   product keeps its scalar x, y, z, w accumulation order.
typedef float tir_f2 __attribute__((ext_vector_type(2)));

struct TapQ { unsigned i0, i1; tir_f2 w; };          // clamped indices; (w0, w1) masked weights as one register pair

__device__ __forceinline__ unsigned clamp_index(int i, int size_m1) {
    int r;
    asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "v"(size_m1));
    return (unsigned)r;
}

__device__ __forceinline__ TapQ make_tap_q(float x, int size) {
    const float ix = unnorm(x, size);
    const float f0 = floorf(ix);
    const float t = ix - f0;
    const int i0 = (int)f0, i1 = i0 + 1;
    TapQ r;
    r.w.x = ((unsigned)i0 < (unsigned)size) ? (1.0f - t) : 0.0f;
    r.w.y = ((unsigned)i1 < (unsigned)size) ? t : 0.0f;
    r.i0 = clamp_index(i0, size - 1);
    r.i1 = clamp_index(i1, size - 1);
    return r;
}

__device__ __forceinline__ float4 ld4b(const float* base, unsigned byte_off) {
    return *reinterpret_cast<const float4*>(reinterpret_cast<const char*>(base) + byte_off);
}

// sum over this lane's 4 channels of bilinear(plane) * linear(line), added to acc in x, y, z, w order.
// wa = (w00, w01) = tx.w * ty.w0, wb = (w10, w11) = tx.w * ty.w1, wl = (line w0, w1)
__device__ __forceinline__ float plane_line_4ch(const float4& a, const float4& b, const float4& cc, const float4& d,
                                                const float4& e, const float4& g, tir_f2 wa, tir_f2 wb, tir_f2 wl, float acc) {
    tir_f2 pxy = tir_f2{a.x, a.y} * tir_f2{wa.x, wa.x}, pzw = tir_f2{a.z, a.w} * tir_f2{wa.x, wa.x};
    pxy = __builtin_elementwise_fma(tir_f2{b.x, b.y}, tir_f2{wa.y, wa.y}, pxy);
    pzw = __builtin_elementwise_fma(tir_f2{b.z, b.w}, tir_f2{wa.y, wa.y}, pzw);
    pxy = __builtin_elementwise_fma(tir_f2{cc.x, cc.y}, tir_f2{wb.x, wb.x}, pxy);
    pzw = __builtin_elementwise_fma(tir_f2{cc.z, cc.w}, tir_f2{wb.x, wb.x}, pzw);
    pxy = __builtin_elementwise_fma(tir_f2{d.x, d.y}, tir_f2{wb.y, wb.y}, pxy);
    pzw = __builtin_elementwise_fma(tir_f2{d.z, d.w}, tir_f2{wb.y, wb.y}, pzw);
    tir_f2 lxy = tir_f2{e.x, e.y} * tir_f2{wl.x, wl.x}, lzw = tir_f2{e.z, e.w} * tir_f2{wl.x, wl.x};
    lxy = __builtin_elementwise_fma(tir_f2{g.x, g.y}, tir_f2{wl.y, wl.y}, lxy);
    lzw = __builtin_elementwise_fma(tir_f2{g.z, g.w}, tir_f2{wl.y, wl.y}, lzw);
    acc = fmaf(pxy.x, lxy.x, acc);
    acc = fmaf(pxy.y, lxy.y, acc);
    acc = fmaf(pzw.x, lzw.x, acc);
    acc = fmaf(pzw.y, lzw.y, acc);
    return acc;
}

// partial density feature: this lane's 16-byte chunk `c` of every tap (4 of the 4*C4 channels).  LDSL: the three density
// line factors come from an LDS image `ll` = [line 0 | line 1 | line 2], each [R_i][4*C4] floats (north_star: "LDS-staged
// factor tiles": a third of every sample's taps (6 of 18 x 64 B) then never reaches the vector L1 / L2; measured cost of
// those taps in the march: ~24 % of the kernel, profiles/r02_line_tap_experiment.txt).  Same arithmetic either way.
template <int C4, bool LDSL>
__device__ __forceinline__ float density_chunk_impl(const TirField& f, const float* __restrict__ ll, float x, float y, float z, int c) {
    constexpr unsigned TB = C4 * 16;                     // bytes per texel / line row
    const TapQ ta[3] = {make_tap_q(x, f.grid[0]), make_tap_q(y, f.grid[1]), make_tap_q(z, f.grid[2])};
    const unsigned cb = 16u * (unsigned)c;
    // byte offset of this lane's chunk within row `index` of an axis: the x offset of a plane tap AND the offset of the line
    // row of the same axis (both tables have TB bytes per row): 6 values serve all 18 taps
    unsigned ob[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ob[a][0] = ta[a].i0 * TB + cb; ob[a][1] = ta[a].i1 * TB + cb; }
    float acc = 0.0f;
    unsigned loff = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int m0 = (i == 2) ? 1 : 0, m1 = (i == 0) ? 1 : 2, vi = 2 - i;
        const TapQ &tx = ta[m0], &ty = ta[m1], &tl = ta[vi];
        const tir_f2 wa = tx.w * tir_f2{ty.w.x, ty.w.x}, wb = tx.w * tir_f2{ty.w.y, ty.w.y};
        // row starts in bytes with the full-rate 24-bit multiply (v_mul_u32_u24: indices < 2^24 and row bytes < 2^24,
        // tir_plane_index_ok; the 32-bit v_mul_lo_u32 issues at a quarter of the rate)
        const unsigned row_bytes = (unsigned)f.grid[m0] * TB;
        const unsigned r0 = mul_u24(ty.i0, row_bytes), r1 = mul_u24(ty.i1, row_bytes);
        const float* pl = f.dplane[i];
        const float4 a = ld4b(pl, r0 + ob[m0][0]);
        const float4 b = ld4b(pl, r0 + ob[m0][1]);
        const float4 cc = ld4b(pl, r1 + ob[m0][0]);
        const float4 d = ld4b(pl, r1 + ob[m0][1]);
        float4 e, g;
        if (LDSL) {
            const char* lb = reinterpret_cast<const char*>(ll + loff);
            e = *reinterpret_cast<const float4*>(lb + ob[vi][0]);
            g = *reinterpret_cast<const float4*>(lb + ob[vi][1]);
            loff += (unsigned)f.grid[vi] * (C4 * 4);
        } else {
            e = ld4b(f.dline[i], ob[vi][0]);
            g = ld4b(f.dline[i], ob[vi][1]);
        }
        acc = plane_line_4ch(a, b, cc, d, e, g, wa, wb, tl.w, acc);
    }
    return acc;
}

template <int C4>
__device__ __forceinline__ float density_feature_chunk(const TirField& f, float x, float y, float z, int c) {
    return density_chunk_impl<C4, false>(f, nullptr, x, y, z, c);
}

template <int C4>
__device__ __forceinline__ float density_feature_chunk_lds(const TirField& f, const float* __restrict__ ll,
                                                           float x, float y, float z, int c) {
    return density_chunk_impl<C4, true>(f, ll, x, y, z, c);
}

// ---- fp16-shadow appearance taps (indirect-light precision policy): d = float(half of h2) * w + acc in ONE instruction
// (v_fma_mix_f32 reads an fp16 operand in place, fp32 arithmetic, one rounding -- what fmaf((float)h, w, acc) computes).  Left
// to the compiler the same source becomes v_cvt_f32_f16 per tap-channel plus packed FMAs: 1.5 instructions per tap-channel.
__device__ __forceinline__ float fma_mix_lo(unsigned h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}
__device__ __forceinline__ float fma_mix_hi(unsigned h2, float w, float acc) {
    float d;
    asm("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(d) : "v"(h2), "v"(w), "v"(acc));
    return d;
}

// fp32 -> fp16 with saturation (|x| > 65504 -> +-65504, never inf)
__device__ __forceinline__ _Float16 sat_half(float x) { return (_Float16)__builtin_amdgcn_fmed3f(x, -65504.0f, 65504.0f); }

// 8 channels of one VM group: bilinear(plane taps a, b, c, d) * linear(line taps e, g) * light row lr[0..8) -> 8 halves (4 dwords)
typedef _Float16 tir_h2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint4 h16_chunk(const uint4& a, const uint4& b, const uint4& c, const uint4& d, const uint4& e,
                                           const uint4& g, float w00, float w01, float w10, float w11, float l0, float l1,
                                           const float4& lr0, const float4& lr1) {
    const unsigned A[4] = {a.x, a.y, a.z, a.w}, B[4] = {b.x, b.y, b.z, b.w}, Cc[4] = {c.x, c.y, c.z, c.w}, D[4] = {d.x, d.y, d.z, d.w};
    const unsigned E[4] = {e.x, e.y, e.z, e.w}, G[4] = {g.x, g.y, g.z, g.w};
    const float lr[8] = {lr0.x, lr0.y, lr0.z, lr0.w, lr1.x, lr1.y, lr1.z, lr1.w};
    unsigned pk[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        float pl = fma_mix_lo(A[p], w00, 0.0f), ph = fma_mix_hi(A[p], w00, 0.0f);
        pl = fma_mix_lo(B[p], w01, pl);  ph = fma_mix_hi(B[p], w01, ph);
        pl = fma_mix_lo(Cc[p], w10, pl); ph = fma_mix_hi(Cc[p], w10, ph);
        pl = fma_mix_lo(D[p], w11, pl);  ph = fma_mix_hi(D[p], w11, ph);
        float ll = fma_mix_lo(E[p], l0, 0.0f), lh = fma_mix_hi(E[p], l0, 0.0f);
        ll = fma_mix_lo(G[p], l1, ll);   lh = fma_mix_hi(G[p], l1, lh);
        const tir_f2 v2 = {(pl * ll) * lr[2 * p], (ph * lh) * lr[2 * p + 1]};
        pk[p] = __builtin_bit_cast(unsigned, __builtin_convertvector(v2, tir_h2));
    }
    return make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

// The same chunk entirely on the PACKED fp16 pipe (v_pk_mul_f16 / v_pk_fma_f16: two channels per instruction, fp16 rounding after
// every step instead of once): 8 VALU instructions per channel pair instead of 15 (12 v_fma_mix_f32 + 2 v_pk_mul_f32 + 1
// conversion) -- the gather arithmetic is half of the fused indirect kernel's VALU work.  Weights and light row arrive as fp16
// pairs.  ~2x the rounding error of h16_chunk on a product (measured on the maps: profiles/r05_fused_pk16.json); the indirect
// policy's self-check covers it like every other fp16 effect.  Range: (plane x line) is formed BEFORE the light row is applied,
// so the range guard bounds max|plane| max|line| max(1, max|light row|) (ops.HalfRange).
__device__ __forceinline__ uint4 h16_chunk_pk(const uint4& a, const uint4& b, const uint4& c, const uint4& d, const uint4& e,
                                              const uint4& g, tir_h2 w00, tir_h2 w01, tir_h2 w10, tir_h2 w11, tir_h2 l0, tir_h2 l1,
                                              const uint4& lr) {
    const unsigned A[4] = {a.x, a.y, a.z, a.w}, B[4] = {b.x, b.y, b.z, b.w}, Cc[4] = {c.x, c.y, c.z, c.w}, D[4] = {d.x, d.y, d.z, d.w};
    const unsigned E[4] = {e.x, e.y, e.z, e.w}, G[4] = {g.x, g.y, g.z, g.w}, LR[4] = {lr.x, lr.y, lr.z, lr.w};
    unsigned pk[4];
    auto H = [](unsigned u) { return __builtin_bit_cast(tir_h2, u); };
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        tir_h2 pl = H(A[p]) * w00;
        pl = __builtin_elementwise_fma(H(B[p]), w01, pl);
        pl = __builtin_elementwise_fma(H(Cc[p]), w10, pl);
        pl = __builtin_elementwise_fma(H(D[p]), w11, pl);
        tir_h2 ll = H(E[p]) * l0;
        ll = __builtin_elementwise_fma(H(G[p]), l1, ll);
        pk[p] = __builtin_bit_cast(unsigned, (pl * ll) * H(LR[p]));
    }
    return make_uint4(pk[0], pk[1], pk[2], pk[3]);
}

// 8 consecutive fp32 values as 8 saturating fp16 (the light row of a record when the rows are not staged in LDS)
__device__ __forceinline__ uint4 pack8_half(const float* __restrict__ p) {
    const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
    const tir_h2 h0 = {sat_half(a.x), sat_half(a.y)}, h1 = {sat_half(a.z), sat_half(a.w)}, h2 = {sat_half(b.x), sat_half(b.y)},
                 h3 = {sat_half(b.z), sat_half(b.w)};
    return make_uint4(__builtin_bit_cast(unsigned, h0), __builtin_bit_cast(unsigned, h1), __builtin_bit_cast(unsigned, h2),
                      __builtin_bit_cast(unsigned, h3));
}

// DPP helper: value of lane (l - shift) within a 16-lane row, `ident` where that lane is outside the row
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_f(float ident, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, ident),
                                                                 __builtin_bit_cast(int, v), CTRL, ROW_MASK, 0xf, false));
}

// inclusive product scan across `width` (32 or 64) consecutive lanes, VALU only (DPP row shifts +
// row broadcasts; no LDS round trips): row_shr:1/2/4/8 inside 16-lane rows, row_bcast:15 into rows 1 and 3,
// row_bcast:31 into rows 2-3 (64-wide only).
template <int WIDTH>
__device__ __forceinline__ float scan_prod(float v, int /*lane_in_group*/) {
    v *= dpp_f<0x111, 0xf>(1.0f, v);
    v *= dpp_f<0x112, 0xf>(1.0f, v);
    v *= dpp_f<0x114, 0xf>(1.0f, v);
    v *= dpp_f<0x118, 0xf>(1.0f, v);
    v *= dpp_f<0x142, 0xa>(1.0f, v);
    if (WIDTH == 64) v *= dpp_f<0x143, 0xc>(1.0f, v);
    return v;
}

// exclusive version: value of the previous lane of the group (1 for the group's first lane)
template <int WIDTH>
__device__ __forceinline__ float shift_up1(float incl, int lane_in_group) {
    float o = dpp_f<0x138, 0xf>(1.0f, incl);        // wave_shr:1
    return lane_in_group == 0 ? 1.0f : o;
}

template <int WIDTH>
__device__ __forceinline__ float group_sum(float v) {
#pragma unroll
    for (int d = WIDTH / 2; d > 0; d >>= 1) v += __shfl_xor(v, d, WIDTH);
    return v;
}

// Wave-collective density evaluation.  Every lane brings one sample (valid flag + normalised coordinates);
// the valid samples of the wave are compacted (ballot rank) through a small per-wave LDS list and gathered
// with C4 adjacent lanes per sample, each lane owning 16 bytes of every 16*C4-byte tap, so a lane group
// reads whole runs (the L1/TA handles about one lane-address per clock: one-sample-per-lane costs 18*C4
// address cycles per sample, this costs 18) and no gather instruction is spent on culled samples.
// All 64 lanes must call it together.  wl: this wave's LDS scratch, 64*4 floats.
template <int C4>
__device__ __forceinline__ float wave_sigma_t(const TirField& f, bool valid, float x, float y, float z, float* wl) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(valid);
    const int n = __popcll(m);
    if (n == 0) return 0.0f;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    if (valid) { wl[rank * 4] = x; wl[rank * 4 + 1] = y; wl[rank * 4 + 2] = z; }
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = 64 / C4;               // samples per pass
    const int slot_in = lane / C4, c = lane % C4;
    for (int base = 0; base < n; base += PER) {
        const int slot = base + slot_in;
        float part = 0.0f;
        if (slot < n) {
            const float4 p = *reinterpret_cast<const float4*>(wl + slot * 4);
            part = density_feature_chunk<C4>(f, p.x, p.y, p.z, c);
        }
        if (C4 >= 2) part += dpp_f<0xB1, 0xf>(0.0f, part);        // quad_perm [1,0,3,2]
        if (C4 >= 4) part += dpp_f<0x4E, 0xf>(0.0f, part);        // quad_perm [2,3,0,1]
        if (C4 >= 8) part += dpp_f<0x141, 0xf>(0.0f, part);       // row_half_mirror (values are quad-uniform)
        if (slot < n && c == 0) wl[slot * 4 + 3] = part;
    }
    __builtin_amdgcn_wave_barrier();
    float sig = 0.0f;
    if (valid) sig = feature2density(f, wl[rank * 4 + 3]);
    __builtin_amdgcn_wave_barrier();
    return sig;
}

// ---- wave_sigma_t with the density lines in LDS (ll) and the tap SET-UP once per sample.  The lane that brought a sample builds
// its three axis taps (make_tap_q: ~45 VALU instructions) and leaves a 9-dword record in the wave's LDS scratch; the C4 gather
// lanes of a slot read the record and form their 18 addresses with one v_mad_u32_u24 each.  Before, every gather lane repeated
// the set-up for its slot -- C4 times per sample, and the pass ran at 134 VALU instructions of which 78 were taps and addresses
// (profiles/r06_march_isa.txt).  Same taps, same addresses, same arithmetic in the same order: bit-identical results.
// Record: [x: i0 | i1 << 16] [y] [z] [x: w0 w1] [y: w0 w1] [z: w0 w1]; dword 0 receives the slot's feature sum afterwards.
// Grid sizes < 2^16 (checked where the kernel is chosen).  wl: this wave's scratch, 64 * TIR_TAPREC dwords.
#define TIR_TAPREC 9

__device__ __forceinline__ unsigned mad_u24(unsigned a, unsigned b, unsigned c) {
    unsigned r;
    asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 32-bit LDS addresses (device code only: the host pass of hipcc parses these bodies too and has no address space 3)
__device__ __forceinline__ unsigned lds_addr(const float* p) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (unsigned)(uintptr_t)(const __attribute__((address_space(3))) float*)p;
#else
    return 0;
#endif
}
__device__ __forceinline__ float4 lds_ld4(unsigned addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    return *(const __attribute__((address_space(3))) float4*)(uintptr_t)addr;
#else
    return float4{};
#endif
}

template <int C4>
__device__ __forceinline__ float density_chunk_rec(const TirField& f, const float* __restrict__ ll, const unsigned* __restrict__ rec, int c) {
    constexpr unsigned TB = C4 * 16;                     // bytes per texel / line row
    const unsigned pk[3] = {rec[0], rec[1], rec[2]};
    unsigned ix[3][2];
#pragma unroll
    for (int a = 0; a < 3; ++a) { ix[a][0] = pk[a] & 0xffffu; ix[a][1] = pk[a] >> 16; }
    const unsigned cb = 16u * (unsigned)c;
    unsigned col[2][2];                                  // this lane's chunk in column x / y of a plane row
#pragma unroll
    for (int a = 0; a < 2; ++a) { col[a][0] = mad_u24(ix[a][0], TB, cb); col[a][1] = mad_u24(ix[a][1], TB, cb); }
    float acc = 0.0f;
    // LDS byte address of this lane's chunk in row 0 of the current line (a 32-bit LDS address, so that the row offset is the only
    // thing added per tap: v_mad_u32_u24 straight into ds_read_b128)
    unsigned laddr = lds_addr(ll) + cb;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const int m0 = (i == 2) ? 1 : 0, m1 = (i == 0) ? 1 : 2, vi = 2 - i;
        const unsigned row_bytes = (unsigned)f.grid[m0] * TB;
        const float* pl = f.dplane[i];
        const float4 a = ld4b(pl, mad_u24(ix[m1][0], row_bytes, col[m0][0]));
        const float4 b = ld4b(pl, mad_u24(ix[m1][0], row_bytes, col[m0][1]));
        const float4 cc = ld4b(pl, mad_u24(ix[m1][1], row_bytes, col[m0][0]));
        const float4 d = ld4b(pl, mad_u24(ix[m1][1], row_bytes, col[m0][1]));
        const float4 e = lds_ld4(mad_u24(ix[vi][0], TB, laddr));
        const float4 g = lds_ld4(mad_u24(ix[vi][1], TB, laddr));
        laddr += (unsigned)f.grid[vi] * TB;
        // (the weights are read behind the taps: they are not needed before the taps arrive)
        const tir_f2 w0 = {__uint_as_float(rec[3 + 2 * m0]), __uint_as_float(rec[4 + 2 * m0])};
        const tir_f2 w1 = {__uint_as_float(rec[3 + 2 * m1]), __uint_as_float(rec[4 + 2 * m1])};
        const tir_f2 wv = {__uint_as_float(rec[3 + 2 * vi]), __uint_as_float(rec[4 + 2 * vi])};
        const tir_f2 wa = w0 * tir_f2{w1.x, w1.x}, wb = w0 * tir_f2{w1.y, w1.y};
        acc = plane_line_4ch(a, b, cc, d, e, g, wa, wb, wv, acc);
    }
    return acc;
}

template <int C4>
__device__ __forceinline__ float wave_sigma_lds(const TirField& f, const float* __restrict__ ll, bool valid, float x, float y,
                                                float z, float* wl) {
    const int lane = threadIdx.x & 63;
    const unsigned long long m = __ballot(valid);
    const int n = __popcll(m);
    if (n == 0) return 0.0f;
    const int rank = __popcll(m & ((1ull << lane) - 1ull));
    unsigned* const wr = reinterpret_cast<unsigned*>(wl);
    if (valid) {
        const TapQ tx = make_tap_q(x, f.grid[0]), ty = make_tap_q(y, f.grid[1]), tz = make_tap_q(z, f.grid[2]);
        unsigned* r = wr + rank * TIR_TAPREC;
        r[0] = tx.i0 | (tx.i1 << 16); r[1] = ty.i0 | (ty.i1 << 16); r[2] = tz.i0 | (tz.i1 << 16);
        r[3] = __float_as_uint(tx.w.x); r[4] = __float_as_uint(tx.w.y);
        r[5] = __float_as_uint(ty.w.x); r[6] = __float_as_uint(ty.w.y);
        r[7] = __float_as_uint(tz.w.x); r[8] = __float_as_uint(tz.w.y);
    }
    __builtin_amdgcn_wave_barrier();
    constexpr int PER = 64 / C4;
    const int slot_in = lane / C4, c = lane % C4;
    for (int base = 0; base < n; base += PER) {
        const int slot = base + slot_in;
        float part = 0.0f;
        if (slot < n) part = density_chunk_rec<C4>(f, ll, wr + slot * TIR_TAPREC, c);
        if (C4 >= 2) part += dpp_f<0xB1, 0xf>(0.0f, part);
        if (C4 >= 4) part += dpp_f<0x4E, 0xf>(0.0f, part);
        if (C4 >= 8) part += dpp_f<0x141, 0xf>(0.0f, part);
        if (slot < n && c == 0) wl[slot * TIR_TAPREC] = part;      // (the slot's lanes have read the record: LDS ops of a wave complete in order)
    }
    __builtin_amdgcn_wave_barrier();
    float sig = 0.0f;
    if (valid) sig = feature2density(f, wl[rank * TIR_TAPREC]);
    __builtin_amdgcn_wave_barrier();
    return sig;
}

__device__ __forceinline__ float wave_sigma(const TirField& f, bool valid, float x, float y, float z, float* wl) {
    switch (f.n_dcomp) {
        case 16: return wave_sigma_t<4>(f, valid, x, y, z, wl);
        case 8:  return wave_sigma_t<2>(f, valid, x, y, z, wl);
        case 32: return wave_sigma_t<8>(f, valid, x, y, z, wl);
        default: return wave_sigma_t<1>(f, valid, x, y, z, wl);
    }
}

// exp(-x), x >= 0, for the SECONDARY marches (visibility / indirect light; models/relight_utils.py:690-697, :811-817): one
// multiply + v_exp_f32 (exp2, 1 ulp) instead of libm's expf (~18 VALU instructions with its range handling) in a kernel that is
// VALU-issue bound.  |x| 2^-24 of argument rounding -> a relative error of the result <= 6e-8 (1 + |x|): three orders of magnitude
// below what the transmittance needs (T < 1e-4 ends a ray's contribution).  The primary march keeps expf: its weights decide the
// record set the decoders see.
__device__ __forceinline__ float exp_neg_fast(float x) { return __builtin_amdgcn_exp2f(x * -1.4426950408889634f); }

// cull of one world-space sample (bbox + occupancy, models/tensorBase_rotated_lights.py:892-897) and its
// normalised coordinates (:916)
__device__ __forceinline__ bool sample_valid(const TirField& f, float px, float py, float pz, float& x, float& y, float& z) {
    x = norm_coord(px, f.aabb_min[0], f.inv_aabb[0]);
    y = norm_coord(py, f.aabb_min[1], f.inv_aabb[1]);
    z = norm_coord(pz, f.aabb_min[2], f.inv_aabb[2]);
    if (!in_bbox(f, px, py, pz)) return false;
    if (f.occ_nbr != nullptr && !occupancy_hit(f, px, py, pz)) return false;
    return true;
}

// Parameter range [t0, t1] of the ray o + t d inside the box that contains everything the occupancy mask can report as
// occupied (TirField::occ_lo / occ_hi; built with 1.25 cells of margin).  Samples outside of it are culled by
// sample_valid in any case: the march kernels use the range to skip whole 32-sample steps without touching them, which
// changes no result (a culled sample contributes alpha = 0: T * (1 + 1e-10) == T in fp32, weight 0).  Box not given
// (occ_lo >= occ_hi) -> (-inf, +inf); the ray misses the box -> t0 > t1.
__device__ __forceinline__ void occ_t_range(const float (&occ_lo)[3], const float (&occ_hi)[3], const float (&o)[3], const float (&d)[3],
                                            float& t0, float& t1) {
    t0 = -INFINITY; t1 = INFINITY;
    if (!((occ_lo[0] < occ_hi[0]) & (occ_lo[1] < occ_hi[1]) & (occ_lo[2] < occ_hi[2]))) return;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        if (fabsf(d[a]) < 1e-20f) {
            if ((o[a] < occ_lo[a]) | (o[a] > occ_hi[a])) { t0 = INFINITY; t1 = -INFINITY; }
        } else {
            // v_rcp_f32 (1 ulp) instead of the IEEE division (~10 instructions per axis): the range only has to be conservative,
            // and the 1e-4 slack below is ~1000 x the error of the fast reciprocal at these magnitudes (|t| < 10)
            const float inv = __builtin_amdgcn_rcpf(d[a]);
            const float ta = (occ_lo[a] - o[a]) * inv, tb = (occ_hi[a] - o[a]) * inv;
            t0 = fmaxf(t0, fminf(ta, tb));
            t1 = fminf(t1, fmaxf(ta, tb));
        }
    }
    t0 -= 1e-4f; t1 += 1e-4f;            // fp32 slack of the slab arithmetic itself
}

// x / d and x % d for a launch-invariant divisor and x < 2^32: m = floor((2^32 - 1) / d) gives q0 = mulhi(x, m) in {q - 1, q}
// (m d = 2^32 - e with 1 <= e <= d, so x m / 2^32 = x / d - x e / (d 2^32) > x / d - 1), one correction makes it exact.  The
// compiler's 64-bit division sequence costs ~30 VALU instructions even on its 32-bit fast path; this is 2 quarter-rate
// multiplies + 4.
struct UDiv { unsigned d, m; };
__device__ __forceinline__ UDiv make_udiv(int d) { return UDiv{d > 1 ? (unsigned)d : 1u, d > 1 ? 0xFFFFFFFFu / (unsigned)d : 0u}; }
__device__ __forceinline__ unsigned udiv(unsigned x, const UDiv& u, unsigned& rem) {
    if (u.d == 1) { rem = 0; return x; }
    unsigned q = __umulhi(x, u.m), r = x - q * u.d;
    if (r >= u.d) { q += 1; r -= u.d; }
    rem = r;
    return q;
}

// linear2srgb_torch after the [0,1] clip (models/relight_utils.py:489-515)
__device__ __forceinline__ float linear2srgb(float x) {
    x = fminf(fmaxf(x, 0.0f), 1.0f);
    float lin = x * 12.92f;
    float nonlin = 1.055f * powf(x + 1e-6f, 0.41666666666666667f) - 0.055f;   // python: 1/2.4, (1.055-1) as doubles -> fp32
    return (x <= 0.0031308f) ? lin : nonlin;
}

// ------------------------------------------------------------------------------------------------
// per-ray setup of sample_ray (models/tensorBase_rotated_lights.py:705-713)
// ------------------------------------------------------------------------------------------------
struct RaySetup {
    float o[3], d[3];
    float t_min;
};

__device__ __forceinline__ RaySetup ray_setup(const TirField& f, const float* __restrict__ rays, int r) {
    RaySetup s;
    float tm = -INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        s.o[a] = rays[6 * (size_t)r + a];
        s.d[a] = rays[6 * (size_t)r + 3 + a];
        float vec = (s.d[a] == 0.0f) ? 1e-6f : s.d[a];
        float ra = __fdiv_rn(sub_rn(f.aabb_max[a], s.o[a]), vec);
        float rb = __fdiv_rn(sub_rn(f.aabb_min[a], s.o[a]), vec);
        tm = fmaxf(tm, fminf(ra, rb));
    }
    s.t_min = fminf(fmaxf(tm, f.near_), f.far_);
    return s;
}

// z of sample k: t_min + stepSize * (k [+ jitter])   (:714-719)
__device__ __forceinline__ float sample_z(const TirField& f, float t_min, int k, float jitter, bool has_jitter) {
    float rng = (float)k;
    if (has_jitter) rng = add_rn(rng, jitter);
    return add_rn(t_min, mul_rn(f.step_size, rng));
}



// ---- counter-based RNG (Philox4x32-10): BRDF-jitter noise in the gather kernel, importance samples of the HDR map ----
__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
}

// three independent N(0,1) values for point `idx` (Box-Muller on the four Philox words)
__device__ __forceinline__ void jitter_normals(unsigned long long seed, unsigned long long offset, uint64_t idx, float (&nrm)[3]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
    const float u0 = ((float)c[0] + 0.5f) * 2.3283064365386963e-10f, u1 = ((float)c[1] + 0.5f) * 2.3283064365386963e-10f;
    const float u2 = ((float)c[2] + 0.5f) * 2.3283064365386963e-10f, u3 = ((float)c[3] + 0.5f) * 2.3283064365386963e-10f;
    const float r0 = sqrtf(-2.0f * logf(fminf(u0, 0.99999994f))), r1 = sqrtf(-2.0f * logf(fminf(u2, 0.99999994f)));
    float s0, c0, s1, c1;
    sincospif(2.0f * u1, &s0, &c0);
    sincospif(2.0f * u3, &s1, &c1);
    nrm[0] = r0 * c0; nrm[1] = r0 * s0; nrm[2] = r1 * c1;
    (void)s1;
}

// four uniforms in (0,1) for counter idx
__device__ __forceinline__ void philox_uniform4(unsigned long long seed, unsigned long long offset, uint64_t idx, float (&u)[4]) {
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), (uint32_t)offset, (uint32_t)(offset >> 32)};
    uint32_t k0 = (uint32_t)seed, k1 = (uint32_t)(seed >> 32);
#pragma unroll
    for (int r = 0; r < 10; ++r) { philox_round(c, k0, k1); k0 += 0x9E3779B9u; k1 += 0xBB67AE85u; }
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = fminf(((float)c[q] + 0.5f) * 2.3283064365386963e-10f, 0.99999994f);
}

// XCD-aware work mapping.  The dispatcher places block b on XCD b % 8 (observed, not contractual: a wrong guess is only
// slower), and each XCD has its own 4 MiB L2.  Records / rays are ordered along the image, so giving every XCD ONE
// contiguous eighth of the items (instead of interleaving items across XCDs) keeps each L2's working set to the part
// of the factor planes its own image region touches.  Returns this block's first item and its stride so that
//     for (i = first; i < end; i += stride)      visits every item of [0, n_items) exactly once over the whole grid.
// `per` = items a block consumes per iteration (e.g. 4 waves x 1 pass).  gridDim.x must be a multiple of 8 when on.
struct XcdRange { long long first, end, stride; };
// bid / nblk: this workgroup's index inside, and the size of, the group of workgroups that shares the items (the whole grid,
// or a slice of it that starts at a multiple of 8 when one launch carries several jobs)
__device__ __forceinline__ XcdRange xcd_range_at(long long n_items, int per, bool on, int bid, int nblk) {
    XcdRange r;
    if (!on || (nblk & 7) != 0) {
        r.first = (long long)bid * per; r.end = n_items; r.stride = (long long)nblk * per;
        return r;
    }
    const int xcd = bid & 7, local = bid >> 3, per_xcd = nblk >> 3;
    const long long chunk = (n_items + 7) / 8;
    const long long lo = chunk * xcd;
    r.first = lo + (long long)local * per;
    r.end = lo + chunk < n_items ? lo + chunk : n_items;
    r.stride = (long long)per_xcd * per;
    return r;
}
__device__ __forceinline__ XcdRange xcd_range(long long n_items, int per, bool on) {
    return xcd_range_at(n_items, per, on, (int)blockIdx.x, (int)gridDim.x);
}

}  // namespace tir
