// Packing kernels + standalone VM field evaluation (K2 density, K6 analytic gradient, K4 appearance).
#include "tir_common.hpp"

using namespace tir;

// ------------------------------------------------------------------------------------------------
// packing
// ------------------------------------------------------------------------------------------------
// [C,H,W] -> [H,W,C]: one thread per (texel, channel); reads strided by H*W, writes coalesced.
__global__ void k_pack_plane(const float* __restrict__ src, float* __restrict__ dst, int C, int HW) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)C * HW) return;
    int c = (int)(i % C);
    int64_t t = i / C;
    dst[i] = src[(int64_t)c * HW + t];
}

__global__ void k_pack_occ(const float* __restrict__ vol, uint32_t* __restrict__ bits, int64_t n) {
    int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    int64_t base = w * 32;
    if (base >= n) return;
    uint32_t m = 0;
    for (int b = 0; b < 32 && base + b < n; ++b) m |= (vol[base + b] > 0.5f ? 1u : 0u) << b;
    bits[w] = m;
}

__global__ void k_pack_basis(const float* __restrict__ w, float* __restrict__ dst, int app_dim, int n_in) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n_in * 32) return;
    int c = i / 32, j = i % 32;
    dst[i] = (j < app_dim) ? w[j * n_in + c] : 0.0f;
}

// torch.mean(light_line(arange(L)), dim=0)  (models/tensoRF_rotated_lights.py:160-161)
__global__ void k_light_mean(const float* __restrict__ ll, float* __restrict__ mean, int L, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float s = 0.0f;
    for (int l = 0; l < L; ++l) s += ll[l * n + i];
    mean[i] = s / (float)L;
}

extern "C" int tir_pack_plane(const float* src, float* dst, int32_t C, int32_t H, int32_t W, void* stream) {
    if (!src || !dst || C <= 0 || H <= 0 || W <= 0) return TIR_ERR_ARG;
    int64_t n = (int64_t)C * H * W;
    hipLaunchKernelGGL(k_pack_plane, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       src, dst, C, H * W);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_pack_occupancy(const float* vol, uint32_t* bits, int64_t n, void* stream) {
    if (!vol || !bits || n <= 0) return TIR_ERR_ARG;
    int64_t words = (n + 31) / 32;
    hipLaunchKernelGGL(k_pack_occ, dim3((unsigned)((words + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       vol, bits, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_pack_basis(const float* w, float* dst, int32_t app_dim, int32_t n_in, void* stream) {
    if (!w || !dst || app_dim <= 0 || app_dim > 32 || n_in <= 0) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_pack_basis, dim3((n_in * 32 + 255) / 256), dim3(256), 0, tir_stream(stream),
                       w, dst, app_dim, n_in);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_light_mean(const float* ll, float* mean, int32_t L, int32_t n, void* stream) {
    if (!ll || !mean || L <= 0 || n <= 0) return TIR_ERR_ARG;
    hipLaunchKernelGGL(k_light_mean, dim3((n + 255) / 256), dim3(256), 0, tir_stream(stream), ll, mean, L, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

static int check_field(const TirField* f) {
    if (!f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->dplane[i] || !f->dline[i]) return TIR_ERR_ARG;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// K2: density feature (+ activation) at normalised points, one point per lane
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256)
k_vm_density(TirField f, const float* __restrict__ xyz, float* __restrict__ feat,
             float* __restrict__ sigma, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = xyz[3 * i], y = xyz[3 * i + 1], z = xyz[3 * i + 2];
    float v = density_feature_dyn(f, x, y, z);
    if (feat) feat[i] = v;
    if (sigma) sigma[i] = feature2density(f, v);
}

extern "C" int tir_vm_density_fwd(const TirField* f, const float* xyz, float* feat, float* sigma,
                                  int64_t n, void* stream) {
    int rc = check_field(f);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !xyz)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_vm_density, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       *f, xyz, feat, sigma, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// AlphaGridMask.sample_alpha(xyz) > 0 at world-space points (models/tensorBase_rotated_lights.py:112-119)
__global__ void __launch_bounds__(256)
k_occupancy_query(TirField f, const float* __restrict__ xyz, uint8_t* __restrict__ hit, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    hit[i] = occupancy_hit(f, xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]) ? 1 : 0;
}

extern "C" int tir_occupancy_query(const TirField* f, const float* xyz, uint8_t* hit, int64_t n, void* stream) {
    if (!f || !f->occ_bits || n < 0 || (n > 0 && (!xyz || !hit))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_occupancy_query, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       *f, xyz, hit, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// K6: analytic gradient of sigma w.r.t. normalised xyz
// ------------------------------------------------------------------------------------------------
template <int C4>
__device__ __forceinline__ void plane_line_grad(const float* __restrict__ plane, const float* __restrict__ line,
                                                int H, int W, int R, float u, float v, float w,
                                                float& val, float& du, float& dv, float& dw) {
    Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
    // The reference's differentiable grid_sample clamps tap indices to the border instead of zero
    // padding (models/relight_utils.py:82-92); inside [-1,1]^3 both agree.  Values here use the clamped
    // taps with the unclamped weights, exactly like that code.
    const float wx0 = 1.0f - tx.t, wx1 = tx.t, wy0 = 1.0f - ty.t, wy1 = ty.t;
    const float* p00 = plane + ((size_t)ty.i0 * W + tx.i0) * (C4 * 4);
    const float* p01 = plane + ((size_t)ty.i0 * W + tx.i1) * (C4 * 4);
    const float* p10 = plane + ((size_t)ty.i1 * W + tx.i0) * (C4 * 4);
    const float* p11 = plane + ((size_t)ty.i1 * W + tx.i1) * (C4 * 4);
    const float* l0 = line + (size_t)tl.i0 * (C4 * 4);
    const float* l1 = line + (size_t)tl.i1 * (C4 * 4);
    const float sx = 0.5f * (float)(W - 1), sy = 0.5f * (float)(H - 1), sl = 0.5f * (float)(R - 1);
    float a_val = 0.f, a_du = 0.f, a_dv = 0.f, a_dw = 0.f;
#pragma unroll
    for (int c = 0; c < C4 * 4; ++c) {
        float a = p00[c], b = p01[c], cc = p10[c], d = p11[c], e = l0[c], g = l1[c];
        float P = fmaf(d, wx1 * wy1, fmaf(cc, wx0 * wy1, fmaf(b, wx1 * wy0, a * (wx0 * wy0))));
        float Pu = fmaf(d - cc, wy1, (b - a) * wy0);
        float Pv = fmaf(d - b, wx1, (cc - a) * wx0);
        float L = fmaf(g, tl.t, e * (1.0f - tl.t));
        float Lw = g - e;
        a_val = fmaf(P, L, a_val);
        a_du = fmaf(Pu, L, a_du);
        a_dv = fmaf(Pv, L, a_dv);
        a_dw = fmaf(P, Lw, a_dw);
    }
    val = a_val; du = a_du * sx; dv = a_dv * sy; dw = a_dw * sl;
}

template <int C4>
__global__ void __launch_bounds__(256)
k_density_grad(TirField f, const float* __restrict__ xyz, float* __restrict__ sigma,
               float* __restrict__ grad, float* __restrict__ normal, int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float px = xyz[3 * i], py = xyz[3 * i + 1], pz = xyz[3 * i + 2];
    float g0 = 0.f, g1 = 0.f, g2 = 0.f;
    float feat = 0.f;
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {       // one component group at a time keeps the tap registers bounded
        // (m0,m1,vi): k=0 -> (x,y,z), k=1 -> (x,z,y), k=2 -> (y,z,x)
        const float u = (k == 2) ? py : px, v = (k == 0) ? py : pz, w = (k == 0) ? pz : ((k == 1) ? py : px);
        const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
        float val, du, dv, dw;
        plane_line_grad<C4>(f.dplane[k], f.dline[k], f.grid[m1], f.grid[m0], f.grid[vi], u, v, w, val, du, dv, dw);
        feat += val;
        if (k == 0) { g0 += du; g1 += dv; g2 += dw; }
        else if (k == 1) { g0 += du; g2 += dv; g1 += dw; }
        else { g1 += du; g2 += dv; g0 += dw; }
    }
    const float g[3] = {g0, g1, g2};
    float ds, sg;
    if (f.act == 1) { sg = fmaxf(feat, 0.f); ds = feat > 0.f ? 1.f : 0.f; }
    else {
        float x = feat + f.density_shift;
        sg = (x > 20.f) ? x : log1pf(expf(x));
        ds = (x > 20.f) ? 1.f : 1.0f / (1.0f + expf(-x));
    }
    float gx = ds * g[0], gy = ds * g[1], gz = ds * g[2];
    if (sigma) sigma[i] = sg;
    if (grad) { grad[3 * i] = gx; grad[3 * i + 1] = gy; grad[3 * i + 2] = gz; }
    if (normal) {
        float nrm = fmaxf(sqrtf(gx * gx + gy * gy + gz * gz), 1e-6f);   // safe_l2_normalize eps
        normal[3 * i] = -gx / nrm; normal[3 * i + 1] = -gy / nrm; normal[3 * i + 2] = -gz / nrm;
    }
}

extern "C" int tir_density_grad_fwd(const TirField* f, const float* xyz, float* sigma, float* grad,
                                    float* normal, int64_t n, void* stream) {
    int rc = check_field(f);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !xyz)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    switch (f->n_dcomp) {
        case 16: hipLaunchKernelGGL(k_density_grad<4>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n); break;
        case 8:  hipLaunchKernelGGL(k_density_grad<2>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n); break;
        case 32: hipLaunchKernelGGL(k_density_grad<8>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n); break;
        default: hipLaunchKernelGGL(k_density_grad<1>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n); break;
    }
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// K4: appearance feature gather + light modulation + basis contraction, one point per lane.
// The 3*Ca plane*line products never leave registers; basis_mat^T rows are wave-uniform -> scalar loads.
// ------------------------------------------------------------------------------------------------
template <int C4, bool RAD, bool INTR>
__global__ void __launch_bounds__(256)
k_vm_app(TirField f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
         const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
         int64_t n) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int CA = C4 * 4;
    const float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float* lrow = nullptr;
    if (RAD) {
        int li = light_idx[idx_map ? idx_map[i] : i];
        li = min(max(li, 0), f.n_lights - 1);
        lrow = f.light_line + (size_t)li * (3 * CA);
    }
    float accr[27], acci[27];
#pragma unroll
    for (int j = 0; j < 27; ++j) { accr[j] = 0.f; acci[j] = 0.f; }
    const int AD = f.app_dim;   // <= 27 here (checked on the host)
#pragma unroll 1
    for (int k = 0; k < 3; ++k) {
        const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
        const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
        Tap1 tx = make_tap(p[m0], W), ty = make_tap(p[m1], H), tl = make_tap(p[vi], R);
        const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
        const float* pl = f.aplane[k];
        const float* p00 = pl + ((size_t)ty.i0 * W + tx.i0) * CA;
        const float* p01 = pl + ((size_t)ty.i0 * W + tx.i1) * CA;
        const float* p10 = pl + ((size_t)ty.i1 * W + tx.i0) * CA;
        const float* p11 = pl + ((size_t)ty.i1 * W + tx.i1) * CA;
        const float* l0 = f.aline[k] + (size_t)tl.i0 * CA;
        const float* l1 = f.aline[k] + (size_t)tl.i1 * CA;
#pragma unroll 2
        for (int c = 0; c < C4; ++c) {
            float4 a = ld4(p00 + 4 * c), b = ld4(p01 + 4 * c), cc = ld4(p10 + 4 * c), d = ld4(p11 + 4 * c);
            float4 e = ld4(l0 + 4 * c), g = ld4(l1 + 4 * c);
            float v[4];
            v[0] = fmaf(d.x, w11, fmaf(cc.x, w10, fmaf(b.x, w01, a.x * w00))) * fmaf(g.x, tl.w1, e.x * tl.w0);
            v[1] = fmaf(d.y, w11, fmaf(cc.y, w10, fmaf(b.y, w01, a.y * w00))) * fmaf(g.y, tl.w1, e.y * tl.w0);
            v[2] = fmaf(d.z, w11, fmaf(cc.z, w10, fmaf(b.z, w01, a.z * w00))) * fmaf(g.z, tl.w1, e.z * tl.w0);
            v[3] = fmaf(d.w, w11, fmaf(cc.w, w10, fmaf(b.w, w01, a.w * w00))) * fmaf(g.w, tl.w1, e.w * tl.w0);
            const int ch = k * CA + 4 * c;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float* brow = f.basis_t + (size_t)(ch + q) * 32;    // wave-uniform address
                float vr = 0.f, vi_ = 0.f;
                if (RAD) vr = v[q] * lrow[ch + q];
                if (INTR) vi_ = v[q] * f.light_mean[ch + q];
#pragma unroll
                for (int j = 0; j < 27; ++j) {
                    float bw = brow[j];
                    if (RAD) accr[j] = fmaf(vr, bw, accr[j]);
                    if (INTR) acci[j] = fmaf(vi_, bw, acci[j]);
                }
            }
        }
    }
    if (RAD) for (int j = 0; j < AD; ++j) rad_feat[i * AD + j] = accr[j];
    if (INTR) for (int j = 0; j < AD; ++j) int_feat[i * AD + j] = acci[j];
}

template <int C4>
static void launch_app(const TirField* f, const float* xyz, const int32_t* li, const int32_t* map,
                       float* rad, float* intr, int64_t n, hipStream_t s) {
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    if (rad && intr) hipLaunchKernelGGL((k_vm_app<C4, true, true>), g, b, 0, s, *f, xyz, li, map, rad, intr, n);
    else if (rad)    hipLaunchKernelGGL((k_vm_app<C4, true, false>), g, b, 0, s, *f, xyz, li, map, rad, intr, n);
    else             hipLaunchKernelGGL((k_vm_app<C4, false, true>), g, b, 0, s, *f, xyz, li, map, rad, intr, n);
}

extern "C" int tir_vm_app_fwd(const TirField* f, const float* xyz, const int32_t* light_idx,
                              const int32_t* idx_map, float* rad_feat, float* int_feat,
                              int64_t n, void* stream) {
    if (!f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i]) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (n < 0 || (n > 0 && !xyz) || (!rad_feat && !int_feat) || (rad_feat && !light_idx)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipStream_t s = tir_stream(stream);
    switch (f->n_acomp) {
        case 48: launch_app<12>(f, xyz, light_idx, idx_map, rad_feat, int_feat, n, s); break;
        case 24: launch_app<6>(f, xyz, light_idx, idx_map, rad_feat, int_feat, n, s); break;
        case 16: launch_app<4>(f, xyz, light_idx, idx_map, rad_feat, int_feat, n, s); break;
        case 96: launch_app<24>(f, xyz, light_idx, idx_map, rad_feat, int_feat, n, s); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
