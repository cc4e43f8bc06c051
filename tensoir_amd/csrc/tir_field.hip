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

// float 0/1 volume [D][H][W] -> neighbourhood bytes [(D+1)][(H+1)][(W+1)]: cell (z0+1,y0+1,x0+1) holds, in bit
// dx+2dy+4dz, whether voxel (x0+dx,y0+dy,z0+dz) is inside the grid and occupied (> 0.5)
__global__ void k_pack_occ(const float* __restrict__ vol, uint8_t* __restrict__ nbr, int W, int H, int D) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t n = (int64_t)(W + 1) * (H + 1) * (D + 1);
    if (i >= n) return;
    const int x0 = (int)(i % (W + 1)) - 1, y0 = (int)((i / (W + 1)) % (H + 1)) - 1, z0 = (int)(i / ((int64_t)(W + 1) * (H + 1))) - 1;
    uint32_t m = 0;
    for (int c = 0; c < 8; ++c) {
        const int xx = x0 + (c & 1), yy = y0 + ((c >> 1) & 1), zz = z0 + (c >> 2);
        if (xx >= 0 && xx < W && yy >= 0 && yy < H && zz >= 0 && zz < D && vol[((int64_t)zz * H + yy) * W + xx] > 0.5f) m |= 1u << c;
    }
    nbr[i] = (uint8_t)m;
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

extern "C" int tir_pack_occupancy(const float* vol, uint8_t* nbr, int32_t W, int32_t H, int32_t D, void* stream) {
    if (!vol || !nbr || W <= 0 || H <= 0 || D <= 0) return TIR_ERR_ARG;
    int64_t n = (int64_t)(W + 1) * (H + 1) * (D + 1);
    hipLaunchKernelGGL(k_pack_occ, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       vol, nbr, W, H, D);
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
    // the gathers index planes with 32-bit element offsets
    for (int i = 0; i < 3; ++i)
        for (int j = i + 1; j < 3; ++j)
            if ((int64_t)f->grid[i] * f->grid[j] * (f->n_acomp > f->n_dcomp ? f->n_acomp : f->n_dcomp) >= ((int64_t)1 << 31)) return TIR_ERR_UNSUPPORTED;
    if (!(f->n_dcomp == 4 || f->n_dcomp == 8 || f->n_dcomp == 16 || f->n_dcomp == 32)) return TIR_ERR_UNSUPPORTED;
    if (!tir_occ_index_ok(f) || !tir_plane_index_ok(f)) return TIR_ERR_UNSUPPORTED;
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
    if (!f || !f->occ_nbr || n < 0 || (n > 0 && (!xyz || !hit))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_occupancy_query, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream),
                       *f, xyz, hit, n);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// ------------------------------------------------------------------------------------------------
// Occupancy-grid maintenance on the device: getDenseAlpha / compute_alpha,
// the 3x3x3 max-pool + threshold of updateAlphaMask, filtering_rays
// (models/tensorBase_rotated_lights.py:737-811, :819-837).
// ------------------------------------------------------------------------------------------------
// alpha[ix][iy][iz] = 1 - exp(-sigma(p) * length),  p = aabb0 * (1 - s) + aabb1 * s with s = (lin_x[ix], lin_y[iy], lin_z[iz])
// (the caller's torch.linspace tables, so positions are bit-identical to the reference's meshgrid);
// sigma = 0 where the current occupancy mask (if any) culls the point.
__global__ void __launch_bounds__(256)
k_dense_alpha(TirField f, const float* __restrict__ lin_x, const float* __restrict__ lin_y,
              const float* __restrict__ lin_z, int gx, int gy, int gz, float length, float* __restrict__ alpha) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)gx * gy * gz) return;
    const int iz = (int)(i % gz), iy = (int)((i / gz) % gy), ix = (int)(i / ((int64_t)gz * gy));
    const float s[3] = {lin_x[ix], lin_y[iy], lin_z[iz]};
    float p[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) p[a] = add_rn(mul_rn(f.aabb_min[a], sub_rn(1.0f, s[a])), mul_rn(f.aabb_max[a], s[a]));
    float sigma = 0.0f;
    if (f.occ_nbr == nullptr || occupancy_hit(f, p[0], p[1], p[2])) {
        const float x = norm_coord(p[0], f.aabb_min[0], f.inv_aabb[0]);
        const float y = norm_coord(p[1], f.aabb_min[1], f.inv_aabb[1]);
        const float z = norm_coord(p[2], f.aabb_min[2], f.inv_aabb[2]);
        sigma = feature2density_ref(f, density_feature_dyn(f, x, y, z));
    }
    alpha[i] = 1.0f - expf(-sigma * length);
}

// vol[z][y][x] = max over the 3x3x3 neighbourhood of clamp(alpha[x'][y'][z'], 0, 1) >= thres ? 1 : 0
// (alpha.clamp(0,1).transpose(0,2) -> F.max_pool3d(k=3, pad=1, stride=1) -> threshold, :757-765);
// also the index bounding box of the occupied voxels (6 ints: min x,y,z, max x,y,z) for the new aabb (:770-777).
__global__ void __launch_bounds__(256)
k_alpha_pool(const float* __restrict__ alpha, int gx, int gy, int gz, float thres, float* __restrict__ vol,
             int32_t* __restrict__ bbox) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (int64_t)gx * gy * gz) return;
    const int x = (int)(i % gx), y = (int)((i / gx) % gy), z = (int)(i / ((int64_t)gx * gy));
    float m = -INFINITY;
    for (int dx = -1; dx <= 1; ++dx)
        for (int dy = -1; dy <= 1; ++dy)
            for (int dz = -1; dz <= 1; ++dz) {
                const int xx = x + dx, yy = y + dy, zz = z + dz;
                if (xx < 0 || xx >= gx || yy < 0 || yy >= gy || zz < 0 || zz >= gz) continue;
                const float a = alpha[((int64_t)xx * gy + yy) * gz + zz];
                m = fmaxf(m, fminf(fmaxf(a, 0.0f), 1.0f));
            }
    const bool occ = m >= thres;
    vol[i] = occ ? 1.0f : 0.0f;
    if (occ && bbox) {
        atomicMin(bbox + 0, x); atomicMin(bbox + 1, y); atomicMin(bbox + 2, z);
        atomicMax(bbox + 3, x); atomicMax(bbox + 4, y); atomicMax(bbox + 5, z);
    }
}

// filtering_rays (:781-811): one wave per ray.  bbox_only: t_max > t_min of the slab test; otherwise "some
// sample of the ray (eval sampling, no jitter) has a positive occupancy lookup".
__global__ void __launch_bounds__(256)
k_filter_rays(TirField f, const float* __restrict__ rays, int64_t n, int n_samples, int bbox_only,
              uint8_t* __restrict__ mask) {
    const int lane = threadIdx.x & 63;
    const int64_t ray = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (ray >= n) return;
    float o[3], d[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) { o[a] = rays[6 * ray + a]; d[a] = rays[6 * ray + 3 + a]; }
    float tmin = -INFINITY, tmax = INFINITY;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float vec = (d[a] == 0.0f) ? 1e-6f : d[a];
        const float ra = __fdiv_rn(sub_rn(f.aabb_max[a], o[a]), vec), rb = __fdiv_rn(sub_rn(f.aabb_min[a], o[a]), vec);
        tmin = fmaxf(tmin, fminf(ra, rb));
        tmax = fminf(tmax, fmaxf(ra, rb));
    }
    if (bbox_only) {
        if (lane == 0) mask[ray] = tmax > tmin ? 1 : 0;
        return;
    }
    const float t0 = fminf(fmaxf(tmin, f.near_), f.far_);
    bool any = false;
    for (int k0 = 0; k0 < n_samples && !any; k0 += 64) {
        const int k = k0 + lane;
        bool hit = false;
        if (k < n_samples) {
            const float z = add_rn(t0, mul_rn(f.step_size, (float)k));
            hit = occupancy_hit(f, add_rn(o[0], mul_rn(d[0], z)), add_rn(o[1], mul_rn(d[1], z)), add_rn(o[2], mul_rn(d[2], z)));
        }
        any = __any(hit);
    }
    if (lane == 0) mask[ray] = any ? 1 : 0;
}

extern "C" int tir_dense_alpha(const TirField* f, const float* lin_x, const float* lin_y, const float* lin_z,
                               int32_t gx, int32_t gy, int32_t gz, float length, float* alpha, void* stream) {
    int rc = check_field(f);
    if (rc) return rc;
    if (!lin_x || !lin_y || !lin_z || !alpha || gx <= 0 || gy <= 0 || gz <= 0) return TIR_ERR_ARG;
    const int64_t n = (int64_t)gx * gy * gz;
    hipLaunchKernelGGL(k_dense_alpha, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), *f, lin_x,
                       lin_y, lin_z, gx, gy, gz, length, alpha);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_alpha_pool(const float* alpha, int32_t gx, int32_t gy, int32_t gz, float thres, float* vol,
                              int32_t* bbox, void* stream) {
    if (!alpha || !vol || gx <= 0 || gy <= 0 || gz <= 0) return TIR_ERR_ARG;
    const int64_t n = (int64_t)gx * gy * gz;
    hipLaunchKernelGGL(k_alpha_pool, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, tir_stream(stream), alpha, gx, gy,
                       gz, thres, vol, bbox);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_filter_rays(const TirField* f, const float* rays, int64_t n, int32_t n_samples, int32_t bbox_only,
                               uint8_t* mask, void* stream) {
    if (!f || n < 0 || (n > 0 && (!rays || !mask))) return TIR_ERR_ARG;
    if (!bbox_only && (!f->occ_nbr || n_samples <= 0)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipLaunchKernelGGL(k_filter_rays, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, tir_stream(stream), *f, rays, n,
                       n_samples, bbox_only, mask);
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
               float* __restrict__ grad, float* __restrict__ normal, int64_t n, const int32_t* __restrict__ n_dev) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
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
        sg = softplus20(x);
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
                                    float* normal, int64_t n, const int32_t* n_dev, void* stream) {
    int rc = check_field(f);
    if (rc) return rc;
    if (n < 0 || (n > 0 && !xyz)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    dim3 g((unsigned)((n + 255) / 256)), b(256);
    switch (f->n_dcomp) {
        case 16: hipLaunchKernelGGL(k_density_grad<4>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n, n_dev); break;
        case 8:  hipLaunchKernelGGL(k_density_grad<2>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n, n_dev); break;
        case 32: hipLaunchKernelGGL(k_density_grad<8>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n, n_dev); break;
        default: hipLaunchKernelGGL(k_density_grad<1>, g, b, 0, tir_stream(stream), *f, xyz, sigma, grad, normal, n, n_dev); break;
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
k_vm_app_valu(TirField f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
         const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
         int out_stride, int idx_div, int64_t n, const int32_t* __restrict__ n_dev) {
    int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    if (i >= n) return;
    constexpr int CA = C4 * 4;
    const float p[3] = {xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]};
    const float* lrow = nullptr;
    if (RAD) {
        int64_t lsel = idx_map ? (int64_t)idx_map[i] : i;
        if (idx_div > 1) lsel /= idx_div;
        int li = light_idx[lsel];
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
#pragma unroll
    for (int j = 0; j < 32; ++j) {
        if (j < out_stride) {
            const float vr = (j < 27) ? accr[j < 27 ? j : 0] : 0.f, vi2 = (j < 27) ? acci[j < 27 ? j : 0] : 0.f;
            if (RAD) rad_feat[i * out_stride + j] = (j < AD) ? vr : 0.f;
            if (INTR) int_feat[i * out_stride + j] = (j < AD) ? vi2 : 0.f;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4 on the matrix cores.  Gather: 4 adjacent lanes share one sample, each owning 16 B of every 64 B
// run of a tap, so a quad reads whole lines (the L1/TA handles ~1 lane-address per clock: a lane-per-sample
// gather costs 216 address cycles per sample, this mapping 54).  Contraction: the plane*line*light products
// of 16 samples go through a per-wave LDS tile [channel][sample] into v_mfma_f32_16x16x4_f32 (exact fp32)
// against basis_mat^T held in LDS:  F^T[32 x 16] += W^T[32 x 4] * X[4 x 16], one plane group (CA channels)
// at a time.
// ------------------------------------------------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
#define TIR_XLD 17

// In-kernel BRDF-jitter noise (models/tensorBase_rotated_lights.py:937: xyz + randn_like(xyz) * 0.01): the gather of the
// jittered points draws its own N(0,1) triple per point -- Philox4x32-10 keyed by the framework generator's (seed,
// offset), counter = point index -- instead of a framework randn + add pair of launches and an [A,3] round trip.
// rng_dev (int64[2] = {seed, offset} on the device) overrides the by-value pair: a captured graph replays with fresh
// noise because the compositing kernel of the pass bumps the device-side offset.
struct TirJitter {
    float scale;                       // 0 = no jitter
    unsigned long long seed, offset;
    const long long* rng_dev;
    float* xyz_out;                    // [n,3] jittered points (the decoder's aux input)
};

   // padded sample stride of the X tile (bank spread for the quad-strided writes)

// HX (round 6): the basis_mat contraction on v_mfma_f32_16x16x32_f16 with BOTH operands split x = hi + lo in fp16 (hi hi + lo hi +
// hi lo, fp32 accumulate: ~2^-21 relative per product, fp32-grade features: 2e-6 of the feature scale measured) instead of
// v_mfma_f32_16x16x4_f32, which runs at the vector rate: 12 matrix instructions per VM group and feature vector instead of 24,
// each half as long -- a quarter of the matrix-pipe time.  Measured on the bench step: 99 -> 96 us for the merged primary gather
// (the launch is bound by its 1.8 passes per wave and the tap latency, not by the matrix pipe, whatever the static count said);
// and the fp16 residue of a product below 0.125 is a subnormal: features good to ~1e-6 of their scale, which the BRDF decoder of a
// field trained to 300^3 amplifies to 1e-4 on the albedo map (profiles/r06m_precision_trained_300_x3_gather.json).  OPT-IN
// (TENSOIR_APP_CONTRACTION=x3); the primary stage keeps the exact instruction.  The product tile is [16 samples][64 + 8 halves] hi and lo (channels >= CA stay zero), basis_mat^T is pre-split into
// operand tiles at kernel start (the layout of k_vm_app_bf16 below).  Products beyond +-65504 saturate (no field does that:
// the fp16 kernels' range guard bounds them).  The exact route (HX = false) stays for TENSOIR_DECODER=mfma / the training forward.
typedef _Float16 app_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 app_f16x2 __attribute__((ext_vector_type(2)));
typedef float app_f32x2 __attribute__((ext_vector_type(2)));
#define TIR_XHX 72      // sample stride of the half-split product tile in halves (64 channels + 8: 144 B rows)

// 4 channel values -> saturating fp16 hi and the fp16 residue, 8 bytes each
__device__ __forceinline__ void app_split4(const float (&v)[4], _Float16* __restrict__ xh, _Float16* __restrict__ xl) {
    unsigned hi[2], lo[2];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const float x0 = __builtin_amdgcn_fmed3f(v[2 * q], -65504.0f, 65504.0f), x1 = __builtin_amdgcn_fmed3f(v[2 * q + 1], -65504.0f, 65504.0f);
        const app_f32x2 x = {x0, x1};
        const app_f16x2 h2 = __builtin_convertvector(x, app_f16x2);
        const app_f32x2 r = x - __builtin_convertvector(h2, app_f32x2);
        hi[q] = __builtin_bit_cast(unsigned, h2);
        lo[q] = __builtin_bit_cast(unsigned, __builtin_convertvector(r, app_f16x2));
    }
    *reinterpret_cast<uint2*>(xh) = make_uint2(hi[0], hi[1]);
    *reinterpret_cast<uint2*>(xl) = make_uint2(lo[0], lo[1]);
}

template <int C4, bool RAD, bool INTR, bool JIT, bool HX = false>
__device__ __forceinline__ void
app_mfma_body(const TirField& f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
              const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
              int out_stride, int idx_div, int64_t n, const int32_t* __restrict__ n_dev, int xcd_on, TirJitter jt, int lt_rows,
              const int bid, const int nblk) {
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));        // device-side point count (no host sync needed)
    constexpr int CA = C4 * 4;
    constexpr int NX = (RAD ? 1 : 0) + (INTR ? 1 : 0);
    static_assert(!HX || CA <= 64, "half-split contraction: one VM group must fit two 32-wide k-steps");
    constexpr int W_FLOATS = HX ? (2 * 3 * 2 * 2 * 64 * 16) / 4 : 3 * CA * 32;          // operand tiles hi + lo | fp32 basis_mat^T
    constexpr int X_FLOATS = HX ? (NX * 2 * 16 * TIR_XHX * 2) / 4 : NX * CA * TIR_XLD;   // per wave
    extern __shared__ __attribute__((aligned(16))) float lds_app[];
    float* Wt = lds_app;                                   // [3*CA][32]
    app_f16x8* Whi = reinterpret_cast<app_f16x8*>(lds_app);    // HX: [group 3][k-step 2][row tile 2][k-group 4][row 16] x 8 halves, hi then lo
    app_f16x8* Wlo = Whi + 3 * 2 * 2 * 64;
    if (JIT && jt.rng_dev) { jt.seed = (unsigned long long)jt.rng_dev[0]; jt.offset = (unsigned long long)jt.rng_dev[1]; }
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    // light rows in LDS: [n_lt rows of light_line | light_mean] x 3*CA floats (576 B per row).  Their taps were 9 (RAD) +
    // 9 (INTR) of the 63-72 load instructions of a pass, and the ray-coherent gather is bound by the RATE of load
    // instructions through the texture addresser (profiles/r02_gather_pred_bench.txt), not by bytes.
    float* LT = lds_app + W_FLOATS;
    const int n_lt = lt_rows;                              // light_line rows staged (0: read them from memory)
    float* X = LT + (n_lt + 1) * (3 * CA) + wave * X_FLOATS;   // [NX][CA][17]  |  HX: [NX][hi, lo][16][TIR_XHX halves]
    _Float16* XH = reinterpret_cast<_Float16*>(X);
    if constexpr (HX) {
        for (int e = threadIdx.x; e < 3 * 2 * 2 * 64; e += 256) {
            const int row = e & 15, kg = (e >> 4) & 3, mt = (e >> 6) & 1, t = (e >> 7) & 1, k = e >> 8;
            app_f16x8 hi, lo;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int ch = 32 * t + 8 * kg + q;
                const float w = (ch < CA) ? f.basis_t[(size_t)(k * CA + ch) * 32 + mt * 16 + row] : 0.0f;
                hi[q] = (_Float16)__builtin_amdgcn_fmed3f(w, -65504.0f, 65504.0f);
                lo[q] = (_Float16)(w - (float)hi[q]);
            }
            Whi[e] = hi; Wlo[e] = lo;
        }
        for (int e = L; e < X_FLOATS; e += 64) X[e] = 0.0f;            // incl. the never-written pad channels
    } else {
        for (int i = threadIdx.x * 4; i < 3 * CA * 32; i += 256 * 4)
            *reinterpret_cast<float4*>(Wt + i) = *reinterpret_cast<const float4*>(f.basis_t + i);
    }
    for (int i = threadIdx.x * 4; i < n_lt * 3 * CA; i += 256 * 4)
        *reinterpret_cast<float4*>(LT + i) = *reinterpret_cast<const float4*>(f.light_line + i);
    for (int i = threadIdx.x * 4; i < 3 * CA; i += 256 * 4)
        *reinterpret_cast<float4*>(LT + n_lt * 3 * CA + i) = *reinterpret_cast<const float4*>(f.light_mean + i);
    __syncthreads();
    const int j = L >> 2, c = L & 3;          // gather role: sample slot, 16-byte quarter
    const int jj = L & 15, kq = L >> 4;       // MFMA role: sample column, k quarter / output row quarter
    const int64_t n_pass = (n + 15) / 16;
    const XcdRange xr = xcd_range_at(n_pass, 4, xcd_on != 0, bid, nblk);
    for (int64_t pass = xr.first + wave; pass < xr.end; pass += xr.stride) {
        const int64_t s = pass * 16 + j;
        const int64_t sc = s < n ? s : n - 1;
        float p[3] = {xyz[3 * sc], xyz[3 * sc + 1], xyz[3 * sc + 2]};
        if (JIT) {                   // all four lanes of the sample derive the same triple from the same counter
            float nrm[3];
            jitter_normals(jt.seed, jt.offset, (uint64_t)sc, nrm);
#pragma unroll
            for (int a = 0; a < 3; ++a) p[a] = add_rn(p[a], mul_rn(nrm[a], jt.scale));
            if (jt.xyz_out && c == 0 && s < n) { jt.xyz_out[3 * s] = p[0]; jt.xyz_out[3 * s + 1] = p[1]; jt.xyz_out[3 * s + 2] = p[2]; }
        }
        const float* lrow = nullptr;
        if (RAD) {
            int64_t lsel = idx_map ? (int64_t)idx_map[sc] : sc;
            if (idx_div > 1) lsel /= idx_div;
            int li = light_idx[lsel];
            li = min(max(li, 0), f.n_lights - 1);
            lrow = n_lt ? LT + li * (3 * CA) : f.light_line + (size_t)li * (3 * CA);
        }
        const float* lmean = LT + n_lt * (3 * CA);
        f32x4 accr[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 acci[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
            const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
            const float u = (k == 2) ? p[1] : p[0], v = (k == 0) ? p[1] : p[2], w = (k == 0) ? p[2] : ((k == 1) ? p[1] : p[0]);
            Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
            const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
            const float* pl = f.aplane[k];
            const unsigned r0 = (unsigned)(ty.i0 * W) * CA, r1 = (unsigned)(ty.i1 * W) * CA;     // 32-bit element offsets
            const unsigned x0 = (unsigned)tx.i0 * CA, x1 = (unsigned)tx.i1 * CA;
            const float* p00 = pl + (r0 + x0);
            const float* p01 = pl + (r0 + x1);
            const float* p10 = pl + (r1 + x0);
            const float* p11 = pl + (r1 + x1);
            const float* l0 = f.aline[k] + (unsigned)tl.i0 * CA;
            const float* l1 = f.aline[k] + (unsigned)tl.i1 * CA;
            // all of this group's taps are requested before the first one is used: the gather is bound by how many loads a
            // wave keeps in flight (a schedule that saves registers by serialising them measured 10-20 % slower)
            constexpr int NQ = (C4 + 3) / 4;
            float4 ta[NQ], tb[NQ], tc[NQ], td[NQ], te[NQ], tg[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ch4 = 4 * q + c;
                if (ch4 < C4) {
                    ta[q] = ld4(p00 + 4 * ch4); tb[q] = ld4(p01 + 4 * ch4); tc[q] = ld4(p10 + 4 * ch4); td[q] = ld4(p11 + 4 * ch4);
                    te[q] = ld4(l0 + 4 * ch4); tg[q] = ld4(l1 + 4 * ch4);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ch4 = 4 * q + c;                 // this lane's 16-byte chunk of the 64-byte run q
                if (ch4 < C4) {
                    const float4 a = ta[q], b = tb[q], cc = tc[q], d = td[q], e = te[q], g = tg[q];
                    float val[4];
                    val[0] = fmaf(d.x, w11, fmaf(cc.x, w10, fmaf(b.x, w01, a.x * w00))) * fmaf(g.x, tl.w1, e.x * tl.w0);
                    val[1] = fmaf(d.y, w11, fmaf(cc.y, w10, fmaf(b.y, w01, a.y * w00))) * fmaf(g.y, tl.w1, e.y * tl.w0);
                    val[2] = fmaf(d.z, w11, fmaf(cc.z, w10, fmaf(b.z, w01, a.z * w00))) * fmaf(g.z, tl.w1, e.z * tl.w0);
                    val[3] = fmaf(d.w, w11, fmaf(cc.w, w10, fmaf(b.w, w01, a.w * w00))) * fmaf(g.w, tl.w1, e.w * tl.w0);
                    if (RAD) {
                        const float4 lr = ld4(lrow + k * CA + 4 * ch4);
                        if constexpr (HX) {
                            const float vr[4] = {val[0] * lr.x, val[1] * lr.y, val[2] * lr.z, val[3] * lr.w};
                            app_split4(vr, XH + j * TIR_XHX + 4 * ch4, XH + 16 * TIR_XHX + j * TIR_XHX + 4 * ch4);
                        } else {
                            float* xr = X + (4 * ch4) * TIR_XLD + j;
                            xr[0] = val[0] * lr.x; xr[TIR_XLD] = val[1] * lr.y; xr[2 * TIR_XLD] = val[2] * lr.z; xr[3 * TIR_XLD] = val[3] * lr.w;
                        }
                    }
                    if (INTR) {
                        const float4 lm = ld4(lmean + k * CA + 4 * ch4);
                        if constexpr (HX) {
                            const float vi[4] = {val[0] * lm.x, val[1] * lm.y, val[2] * lm.z, val[3] * lm.w};
                            _Float16* xb = XH + (RAD ? 2 * 16 * TIR_XHX : 0);
                            app_split4(vi, xb + j * TIR_XHX + 4 * ch4, xb + 16 * TIR_XHX + j * TIR_XHX + 4 * ch4);
                        } else {
                            float* xi = X + ((RAD ? CA : 0) + 4 * ch4) * TIR_XLD + j;
                            xi[0] = val[0] * lm.x; xi[TIR_XLD] = val[1] * lm.y; xi[2 * TIR_XLD] = val[2] * lm.z; xi[3 * TIR_XLD] = val[3] * lm.w;
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();      // LDS ops of one wave complete in order; keep the compiler from reordering
            if constexpr (HX) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (32 * t >= CA) break;
                    const int wi = ((k * 2 + t) * 2) * 64 + kq * 16 + jj;
                    const app_f16x8 ah0 = Whi[wi], ah1 = Whi[wi + 64], al0 = Wlo[wi], al1 = Wlo[wi + 64];
                    if (RAD) {
                        const _Float16* xr = XH + jj * TIR_XHX + 32 * t + 8 * kq;
                        const app_f16x8 bh = *reinterpret_cast<const app_f16x8*>(xr), bl = *reinterpret_cast<const app_f16x8*>(xr + 16 * TIR_XHX);
                        accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, accr[0], 0, 0, 0);
                        accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, accr[1], 0, 0, 0);
                        accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, accr[0], 0, 0, 0);
                        accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, accr[1], 0, 0, 0);
                        accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, accr[0], 0, 0, 0);
                        accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, accr[1], 0, 0, 0);
                    }
                    if (INTR) {
                        const _Float16* xi = XH + (RAD ? 2 * 16 * TIR_XHX : 0) + jj * TIR_XHX + 32 * t + 8 * kq;
                        const app_f16x8 bh = *reinterpret_cast<const app_f16x8*>(xi), bl = *reinterpret_cast<const app_f16x8*>(xi + 16 * TIR_XHX);
                        acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bh, acci[0], 0, 0, 0);
                        acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bh, acci[1], 0, 0, 0);
                        acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al0, bh, acci[0], 0, 0, 0);
                        acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al1, bh, acci[1], 0, 0, 0);
                        acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah0, bl, acci[0], 0, 0, 0);
                        acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah1, bl, acci[1], 0, 0, 0);
                    }
                }
            }
            const float* wk = Wt + (size_t)(k * CA) * 32;
#pragma unroll 4
            for (int t = 0; t < (HX ? 0 : CA / 4); ++t) {
                const float a0 = wk[(4 * t + kq) * 32 + jj], a1 = wk[(4 * t + kq) * 32 + 16 + jj];
                if (RAD) {
                    const float br = X[(4 * t + kq) * TIR_XLD + jj];
                    accr[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, br, accr[0], 0, 0, 0);
                    accr[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, br, accr[1], 0, 0, 0);
                }
                if (INTR) {
                    const float bi = X[((RAD ? CA : 0) + 4 * t + kq) * TIR_XLD + jj];
                    acci[0] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0, bi, acci[0], 0, 0, 0);
                    acci[1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1, bi, acci[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        // D layout: column = sample jj, rows = kq*4 + reg within each 16-row tile
        const int64_t so = pass * 16 + jj;
        if (so < n) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row0 = mt * 16 + kq * 4;
                if (RAD) {
                    float* o = rad_feat + so * out_stride + row0;
                    if (row0 + 4 <= out_stride && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(accr[mt][0], accr[mt][1], accr[mt][2], accr[mt][3]);
                    else for (int r = 0; r < 4; ++r) if (row0 + r < out_stride) o[r] = accr[mt][r];
                }
                if (INTR) {
                    float* o = int_feat + so * out_stride + row0;
                    if (row0 + 4 <= out_stride && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(acci[mt][0], acci[mt][1], acci[mt][2], acci[mt][3]);
                    else for (int r = 0; r < 4; ++r) if (row0 + r < out_stride) o[r] = acci[mt][r];
                }
            }
        }
    }
}

template <int C4, bool RAD, bool INTR, bool JIT = false, bool HX = false>
__global__ void __launch_bounds__(256)
k_vm_app_mfma(TirField f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
              const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
              int out_stride, int idx_div, int64_t n, const int32_t* __restrict__ n_dev, int xcd_on, TirJitter jt, int lt_rows) {
    app_mfma_body<C4, RAD, INTR, JIT, HX>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, xcd_on, jt,
                                          lt_rows, (int)blockIdx.x, (int)gridDim.x);
}

// The two appearance gathers of the primary stage in ONE launch: workgroups [0, nb0) compute the radiance + intrinsic
// features of the records (models/tensoRF_rotated_lights.py:132-165), workgroups [nb0, nb0 + nb1) the intrinsic features of
// the JITTERED records (:937-938, noise drawn in the kernel).  Both are 230 k-row launches of 1.8 passes per wave on
// their own: merged they fill the chip once instead of twice.
template <int C4, bool HX = false>
__global__ void __launch_bounds__(256)
k_vm_app_primary(TirField f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
                 const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
                 float* __restrict__ int_feat_jit, int out_stride, int64_t n, const int32_t* __restrict__ n_dev, int xcd_on,
                 TirJitter jt, int lt_rows, int nb0) {
    if ((int)blockIdx.x < nb0) {
        TirJitter none{0.0f, 0ull, 0ull, nullptr, nullptr};
        app_mfma_body<C4, true, true, false, HX>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, 0, n, n_dev, xcd_on, none,
                                                 lt_rows, (int)blockIdx.x, nb0);
    } else {
        app_mfma_body<C4, false, true, true, HX>(f, xyz, nullptr, nullptr, nullptr, int_feat_jit, out_stride, 0, n, n_dev, xcd_on, jt,
                                                 lt_rows, (int)blockIdx.x - nb0, (int)gridDim.x - nb0);
    }
}

// ------------------------------------------------------------------------------------------------
// K4 with the contraction on v_mfma_f32_16x16x32_bf16 and every operand split x = hi + lo in bf16 (hi*hi + lo*hi +
// hi*lo, fp32 accumulate; ~2^-17 relative per product): 12 matrix instructions per VM group and feature instead of 24
// four-wide fp32 ones (the exact-fp32 kernel above spends > 50 % of its wave time in MFMA issue stalls).
// Same gather; the per-group X tile is [16 samples][64 channels] fp32 in LDS (channels >= CA stay zero), read back
// 8 consecutive channels per lane and split on the fly; basis_mat^T is pre-split into bf16 operand tiles at kernel start.
// ------------------------------------------------------------------------------------------------
typedef __bf16 app_bf16x8 __attribute__((ext_vector_type(8)));
#define TIR_XS 68      // sample stride of the X tile in floats (64 channels + 4: spreads the 16 rows over the banks)

__device__ __forceinline__ void app_split8(const float4& v0, const float4& v1, app_bf16x8& hi, app_bf16x8& lo) {
    const float v[8] = {v0.x, v0.y, v0.z, v0.w, v1.x, v1.y, v1.z, v1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        hi[e] = (__bf16)v[e];
        lo[e] = (__bf16)(v[e] - (float)hi[e]);
    }
}

template <int C4, bool RAD, bool INTR>
__global__ void __launch_bounds__(256)
k_vm_app_bf16(TirField f, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
              const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, float* __restrict__ int_feat,
              int out_stride, int idx_div, int64_t n, const int32_t* __restrict__ n_dev, int xcd_on) {
    static_assert(C4 * 4 <= 64, "one VM group must fit two 32-wide k-steps");
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    constexpr int CA = C4 * 4;
    constexpr int NX = (RAD ? 1 : 0) + (INTR ? 1 : 0);
    extern __shared__ __attribute__((aligned(16))) float lds_app[];
    // W tiles: [group 3][k-step 2][row tile 2][k-group 4][row 16] x 8 bf16, hi then lo
    app_bf16x8* Whi = reinterpret_cast<app_bf16x8*>(lds_app);
    app_bf16x8* Wlo = Whi + 3 * 2 * 2 * 64;
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    float* X = reinterpret_cast<float*>(Wlo + 3 * 2 * 2 * 64) + wave * (NX * 16 * TIR_XS);      // [NX][16][TIR_XS]
    for (int e = threadIdx.x; e < 3 * 2 * 2 * 64; e += 256) {
        const int row = e & 15, kg = (e >> 4) & 3, mt = (e >> 6) & 1, t = (e >> 7) & 1, k = e >> 8;
        app_bf16x8 hi, lo;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int ch = 32 * t + 8 * kg + q;
            const float w = (ch < CA) ? f.basis_t[(size_t)(k * CA + ch) * 32 + mt * 16 + row] : 0.0f;
            hi[q] = (__bf16)w;
            lo[q] = (__bf16)(w - (float)hi[q]);
        }
        Whi[e] = hi; Wlo[e] = lo;
    }
    for (int e = L; e < NX * 16 * TIR_XS; e += 64) X[e] = 0.0f;          // incl. the never-written pad channels
    __syncthreads();
    const int j = L >> 2, c = L & 3;          // gather role: sample slot, 16-byte quarter
    const int jj = L & 15, kg = L >> 4;       // MFMA role: sample column, k-group / output row quarter
    const int64_t n_pass = (n + 15) / 16;
    const XcdRange xr = xcd_range(n_pass, 4, xcd_on != 0);
    for (int64_t pass = xr.first + wave; pass < xr.end; pass += xr.stride) {
        const int64_t s = pass * 16 + j;
        const int64_t sc = s < n ? s : n - 1;
        const float p[3] = {xyz[3 * sc], xyz[3 * sc + 1], xyz[3 * sc + 2]};
        const float* lrow = nullptr;
        if (RAD) {
            int64_t lsel = idx_map ? (int64_t)idx_map[sc] : sc;
            if (idx_div > 1) lsel /= idx_div;
            int li = light_idx[lsel];
            li = min(max(li, 0), f.n_lights - 1);
            lrow = f.light_line + (size_t)li * (3 * CA);
        }
        f32x4 accr[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        f32x4 acci[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const int m0 = (k == 2) ? 1 : 0, m1 = (k == 0) ? 1 : 2, vi = 2 - k;
            const int H = f.grid[m1], W = f.grid[m0], R = f.grid[vi];
            const float u = (k == 2) ? p[1] : p[0], v = (k == 0) ? p[1] : p[2], w = (k == 0) ? p[2] : ((k == 1) ? p[1] : p[0]);
            Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
            const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
            const float* pl = f.aplane[k];
            const unsigned r0 = (unsigned)(ty.i0 * W) * CA, r1 = (unsigned)(ty.i1 * W) * CA;     // 32-bit element offsets
            const unsigned x0 = (unsigned)tx.i0 * CA, x1 = (unsigned)tx.i1 * CA;
            const float* p00 = pl + (r0 + x0);
            const float* p01 = pl + (r0 + x1);
            const float* p10 = pl + (r1 + x0);
            const float* p11 = pl + (r1 + x1);
            const float* l0 = f.aline[k] + (unsigned)tl.i0 * CA;
            const float* l1 = f.aline[k] + (unsigned)tl.i1 * CA;
#pragma unroll
            for (int q = 0; q < (C4 + 3) / 4; ++q) {
                const int ch4 = 4 * q + c;
                if (ch4 < C4) {
                    const float4 a = ld4(p00 + 4 * ch4), b = ld4(p01 + 4 * ch4), cc = ld4(p10 + 4 * ch4), d = ld4(p11 + 4 * ch4);
                    const float4 e = ld4(l0 + 4 * ch4), g = ld4(l1 + 4 * ch4);
                    float4 val;
                    val.x = fmaf(d.x, w11, fmaf(cc.x, w10, fmaf(b.x, w01, a.x * w00))) * fmaf(g.x, tl.w1, e.x * tl.w0);
                    val.y = fmaf(d.y, w11, fmaf(cc.y, w10, fmaf(b.y, w01, a.y * w00))) * fmaf(g.y, tl.w1, e.y * tl.w0);
                    val.z = fmaf(d.z, w11, fmaf(cc.z, w10, fmaf(b.z, w01, a.z * w00))) * fmaf(g.z, tl.w1, e.z * tl.w0);
                    val.w = fmaf(d.w, w11, fmaf(cc.w, w10, fmaf(b.w, w01, a.w * w00))) * fmaf(g.w, tl.w1, e.w * tl.w0);
                    if (RAD) {
                        const float4 lr = ld4(lrow + k * CA + 4 * ch4);
                        *reinterpret_cast<float4*>(X + j * TIR_XS + 4 * ch4) = make_float4(val.x * lr.x, val.y * lr.y, val.z * lr.z, val.w * lr.w);
                    }
                    if (INTR) {
                        const float4 lm = ld4(f.light_mean + k * CA + 4 * ch4);
                        *reinterpret_cast<float4*>(X + (RAD ? 16 * TIR_XS : 0) + j * TIR_XS + 4 * ch4) =
                            make_float4(val.x * lm.x, val.y * lm.y, val.z * lm.z, val.w * lm.w);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                if (32 * t >= CA) break;
                const int wi = ((k * 2 + t) * 2) * 64 + kg * 16 + jj;
                const app_bf16x8 ah0 = Whi[wi], ah1 = Whi[wi + 64], al0 = Wlo[wi], al1 = Wlo[wi + 64];
                if (RAD) {
                    const float* xr = X + jj * TIR_XS + 32 * t + 8 * kg;
                    app_bf16x8 bh, bl;
                    app_split8(*reinterpret_cast<const float4*>(xr), *reinterpret_cast<const float4*>(xr + 4), bh, bl);
                    accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bh, accr[0], 0, 0, 0);
                    accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bh, accr[1], 0, 0, 0);
                    accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, bh, accr[0], 0, 0, 0);
                    accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, bh, accr[1], 0, 0, 0);
                    accr[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bl, accr[0], 0, 0, 0);
                    accr[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bl, accr[1], 0, 0, 0);
                }
                if (INTR) {
                    const float* xi = X + (RAD ? 16 * TIR_XS : 0) + jj * TIR_XS + 32 * t + 8 * kg;
                    app_bf16x8 bh, bl;
                    app_split8(*reinterpret_cast<const float4*>(xi), *reinterpret_cast<const float4*>(xi + 4), bh, bl);
                    acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bh, acci[0], 0, 0, 0);
                    acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bh, acci[1], 0, 0, 0);
                    acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al0, bh, acci[0], 0, 0, 0);
                    acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al1, bh, acci[1], 0, 0, 0);
                    acci[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah0, bl, acci[0], 0, 0, 0);
                    acci[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah1, bl, acci[1], 0, 0, 0);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        const int64_t so = pass * 16 + jj;
        if (so < n) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const int row0 = mt * 16 + kg * 4;
                if (RAD) {
                    float* o = rad_feat + so * out_stride + row0;
                    if (row0 + 4 <= out_stride && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(accr[mt][0], accr[mt][1], accr[mt][2], accr[mt][3]);
                    else for (int r = 0; r < 4; ++r) if (row0 + r < out_stride) o[r] = accr[mt][r];
                }
                if (INTR) {
                    float* o = int_feat + so * out_stride + row0;
                    if (row0 + 4 <= out_stride && (out_stride & 3) == 0) *reinterpret_cast<float4*>(o) = make_float4(acci[mt][0], acci[mt][1], acci[mt][2], acci[mt][3]);
                    else for (int r = 0; r < 4; ++r) if (row0 + r < out_stride) o[r] = acci[mt][r];
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// K4 for the SECONDARY-ray records (indirect light, models/relight_utils.py:818-829) on fp16 storage and fp16 matrix
// operands, fp32 arithmetic and accumulation (precision policy).
//
// The fp32 gather above is bound by the L1 fill rate (64 B/clk/CU: 3 456 B per record = 12 GB per bench step through the
// vector L1s, 0.63 of the chip's 34.5 TB/s) with the exact-fp32 contraction (72 v_mfma_f32_16x16x4_f32 per 16 records) as the
// second bound.  Here the taps come from an fp16 shadow of the appearance planes / lines (tir_pack_half: same channel-last
// layout, 96 B per texel -> half the bytes through the L1), two lanes per record each owning every other 16-byte chunk of
// a tap (32 records per wave pass, the same 54 load instructions per lane), interpolation and the light-row product in fp32
// (v_fma_mix_f32 reads the half operands in place: no conversion instructions), the products rounded once to fp16 into a
// per-wave LDS tile [record][channel] and contracted against an fp16 image of basis_mat^T with v_mfma_f32_32x32x16_f16:
// 9 matrix instructions of 8 passes per 32 records instead of 144 of 8.  Unit roundoff 2^-12 on the stored taps, on the
// products and on basis_mat; the features feed the single-product fp16 decoder, which rounds its inputs the same way.
// NOT parity grade on a feature by itself (~2e-4 relative): used only where the radiance is averaged over a secondary ray's
// records and 128+ light directions before it reaches rgb_with_brdf_map (measured there: profiles/r04_precision_policy.json).
// ------------------------------------------------------------------------------------------------
typedef _Float16 app_f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 app_f16x2 __attribute__((ext_vector_type(2)));
typedef float app_f32x16 __attribute__((ext_vector_type(16)));
typedef float app_f32x2 __attribute__((ext_vector_type(2)));
#define TIR_XH 56      // record stride of the fp16 X tile in halves (48 channels + 8: 112-B rows keep ds_read_b128 conflict-free)

// fp16 shadow copy with SATURATING casts (|x| > 65504 -> +-65504, never inf) and, when jb.absmax is given, the table's
// abs-max (bit pattern of |x|: non-negative floats order like unsigned integers, a NaN sorts above inf and is reported as
// such) -- the range guard of the indirect-light precision policy reads it (ops.pack_half, relight._indirect_mode).
// dst == NULL: scan only (light rows, basis_mat: tables the gather reads as fp32 / casts in-kernel).
// A workgroup of 256 threads owns TIR_PACK_CHUNKS x 2048 consecutive elements of one table (four 16-byte stores per thread:
// the reduction / atomic tail is paid once per 8192 elements); workgroups past the end of a shorter table leave at once.
#define TIR_PACK_CHUNKS 4
__global__ void __launch_bounds__(256) k_pack_half(TirHalfJobs jobs) {
    const TirHalfJob jb = jobs.job[blockIdx.y];
    const int64_t base = (int64_t)blockIdx.x * (2048 * TIR_PACK_CHUNKS);
    if (base >= jb.n) return;                                   // (block-uniform)
    _Float16* dst = reinterpret_cast<_Float16*>(jb.dst);
    unsigned m = 0;
#pragma unroll
    for (int ch = 0; ch < TIR_PACK_CHUNKS; ++ch) {
        const int64_t i = base + (int64_t)ch * 2048 + (int64_t)threadIdx.x * 8;
        if (i + 8 <= jb.n) {
            const float4 a = ld4(jb.src + i), b = ld4(jb.src + i + 4);
            const float v[8] = {a.x, a.y, a.z, a.w, b.x, b.y, b.z, b.w};
#pragma unroll
            for (int q = 0; q < 8; ++q) m = max(m, __builtin_bit_cast(unsigned, v[q]) & 0x7fffffffu);
            if (dst) {
                app_f16x8 h = {sat_half(a.x), sat_half(a.y), sat_half(a.z), sat_half(a.w), sat_half(b.x), sat_half(b.y), sat_half(b.z), sat_half(b.w)};
                *reinterpret_cast<app_f16x8*>(dst + i) = h;
            }
        } else {
            for (int64_t e = i; e < jb.n; ++e) {
                const float x = jb.src[e];
                m = max(m, __builtin_bit_cast(unsigned, x) & 0x7fffffffu);
                if (dst) dst[e] = sat_half(x);
            }
        }
    }
    if (jb.absmax) {      // (block-uniform branch)
        // one atomic per WORKGROUP at most, and only when it would raise the maximum: a plain read first (the value only ever
        // grows, so a stale read can only cause a redundant atomic, never a missed one).  One atomic per wave on eight addresses
        // serialised 35 000 of them per launch: 209 us instead of 23 (profiles/r05_script_trace.txt).
        __shared__ unsigned s_m[4];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = max(m, (unsigned)__shfl_xor((int)m, o, 64));
        if ((threadIdx.x & 63) == 0) s_m[threadIdx.x >> 6] = m;
        __syncthreads();
        if (threadIdx.x == 0) {
            const unsigned mm = max(max(s_m[0], s_m[1]), max(s_m[2], s_m[3]));
            if (mm > __hip_atomic_load(jb.absmax, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) atomicMax(jb.absmax, mm);
        }
    }
}

__global__ void __launch_bounds__(256)
k_vm_app_h16(TirField f, TirFieldHalf fh, const float* __restrict__ xyz, const int32_t* __restrict__ light_idx,
             const int32_t* __restrict__ idx_map, float* __restrict__ rad_feat, int out_stride, int idx_div, int64_t n,
             const int32_t* __restrict__ n_dev, int xcd_on, int lt_rows) {
    if (n_dev) n = min(n, (int64_t)max(*n_dev, 0));
    constexpr int CA = 48, NQ = 3;                          // 6 16-byte chunks per tap, two lanes per record
    extern __shared__ __attribute__((aligned(16))) float lds_app[];
    // basis_mat^T as A-operand tiles: [group 3][k-step 3][k-group 2][row 32] x 8 halves
    app_f16x8* Wh = reinterpret_cast<app_f16x8*>(lds_app);
    float* LT = lds_app + (3 * 3 * 2 * 32 * 8) / 2;         // light rows, fp32: [n_lt rows of light_line | light_mean] x 144
    const int n_lt = lt_rows;
    const int wave = threadIdx.x >> 6, L = threadIdx.x & 63;
    _Float16* X = reinterpret_cast<_Float16*>(LT + (n_lt + 1) * (3 * CA)) + wave * (32 * TIR_XH);
    for (int e = threadIdx.x; e < 3 * 3 * 2 * 32; e += 256) {
        const int row = e & 31, kg = (e >> 5) & 1, t = (e >> 6) % 3, k = e / 192;
        app_f16x8 h;
#pragma unroll
        for (int q = 0; q < 8; ++q) h[q] = sat_half(f.basis_t[(size_t)(k * CA + 16 * t + 8 * kg + q) * 32 + row]);
        Wh[e] = h;
    }
    _Float16* LT16 = reinterpret_cast<_Float16*>(LT);
    for (int i = threadIdx.x; i < n_lt * 3 * CA; i += 256) LT16[i] = sat_half(f.light_line[i]);
    for (int i = threadIdx.x * 4; i < 3 * CA; i += 256 * 4)
        *reinterpret_cast<float4*>(LT + n_lt * 3 * CA + i) = *reinterpret_cast<const float4*>(f.light_mean + i);
    __syncthreads();
    const int j = L >> 1, c = L & 1;          // gather role: record slot, which of every two 16-byte chunks
    const int col = L & 31, kg = L >> 5;      // MFMA role: record column / feature row, k group
    const int64_t n_pass = (n + 31) / 32;
    const XcdRange xr = xcd_range_at(n_pass, 4, xcd_on != 0, (int)blockIdx.x, (int)gridDim.x);
    for (int64_t pass = xr.first + wave; pass < xr.end; pass += xr.stride) {
        const int64_t s = pass * 32 + j;
        const int64_t sc = s < n ? s : n - 1;
        const float p[3] = {xyz[3 * sc], xyz[3 * sc + 1], xyz[3 * sc + 2]};
        int64_t lsel = idx_map ? (int64_t)idx_map[sc] : sc;
        if (idx_div > 1) lsel /= idx_div;
        int li = light_idx[lsel];
        li = min(max(li, 0), f.n_lights - 1);
        const float* lrow = n_lt ? LT + li * (3 * CA) : f.light_line + (size_t)li * (3 * CA);
        app_f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll 1
        for (int k = 0; k < 3; ++k) {
            const int H = f.grid[(k == 0) ? 1 : 2], W = f.grid[(k == 2) ? 1 : 0], R = f.grid[2 - k];
            const float u = (k == 2) ? p[1] : p[0], v = (k == 0) ? p[1] : p[2], w = (k == 0) ? p[2] : ((k == 1) ? p[1] : p[0]);
            Tap1 tx = make_tap(u, W), ty = make_tap(v, H), tl = make_tap(w, R);
            const float w00 = tx.w0 * ty.w0, w01 = tx.w1 * ty.w0, w10 = tx.w0 * ty.w1, w11 = tx.w1 * ty.w1;
            const _Float16* pl = reinterpret_cast<const _Float16*>(fh.aplane[k]);
            const _Float16* ln = reinterpret_cast<const _Float16*>(fh.aline[k]);
            const unsigned r0 = (unsigned)(ty.i0 * W) * CA, r1 = (unsigned)(ty.i1 * W) * CA;     // 32-bit element offsets
            const unsigned x0 = (unsigned)tx.i0 * CA + 8 * c, x1 = (unsigned)tx.i1 * CA + 8 * c;
            const _Float16* p00 = pl + (r0 + x0);
            const _Float16* p01 = pl + (r0 + x1);
            const _Float16* p10 = pl + (r1 + x0);
            const _Float16* p11 = pl + (r1 + x1);
            const _Float16* l0 = ln + ((unsigned)tl.i0 * CA + 8 * c);
            const _Float16* l1 = ln + ((unsigned)tl.i1 * CA + 8 * c);
            // all 18 taps of the group in flight before the first is used (as in the fp32 gather)
            uint4 ta[NQ], tb[NQ], tc[NQ], td[NQ], te[NQ], tg[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                ta[q] = *reinterpret_cast<const uint4*>(p00 + 16 * q); tb[q] = *reinterpret_cast<const uint4*>(p01 + 16 * q);
                tc[q] = *reinterpret_cast<const uint4*>(p10 + 16 * q); td[q] = *reinterpret_cast<const uint4*>(p11 + 16 * q);
                te[q] = *reinterpret_cast<const uint4*>(l0 + 16 * q);  tg[q] = *reinterpret_cast<const uint4*>(l1 + 16 * q);
            }
            __builtin_amdgcn_sched_barrier(0);
            const tir_h2 hw00 = {(_Float16)w00, (_Float16)w00}, hw01 = {(_Float16)w01, (_Float16)w01}, hw10 = {(_Float16)w10, (_Float16)w10},
                         hw11 = {(_Float16)w11, (_Float16)w11}, hl0 = {(_Float16)tl.w0, (_Float16)tl.w0}, hl1 = {(_Float16)tl.w1, (_Float16)tl.w1};
            const _Float16* lrow16 = LT16 + li * (3 * CA);
#pragma unroll
            for (int q = 0; q < NQ; ++q) {
                const int ch0 = 16 * q + 8 * c;            // this lane's 8 channels of chunk pair q
                *reinterpret_cast<uint4*>(X + j * TIR_XH + ch0) =
                    h16_chunk_pk(ta[q], tb[q], tc[q], td[q], te[q], tg[q], hw00, hw01, hw10, hw11, hl0, hl1,
                                 n_lt ? *reinterpret_cast<const uint4*>(lrow16 + k * CA + ch0) : pack8_half(lrow + k * CA + ch0));
            }
            __builtin_amdgcn_wave_barrier();      // LDS ops of one wave complete in order; keep the compiler from reordering
#pragma unroll
            for (int t = 0; t < 3; ++t) {
                const app_f16x8 a = Wh[((k * 3 + t) * 2 + kg) * 32 + col];
                const app_f16x8 b = *reinterpret_cast<const app_f16x8*>(X + col * TIR_XH + 16 * t + 8 * kg);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc, 0, 0, 0);
            }
            __builtin_amdgcn_wave_barrier();
        }
        // D layout: column = record col, rows (features) 8 i + 4 kg + (0..3) in registers 4 i .. 4 i + 3
        const int64_t so = pass * 32 + col;
        if (so < n) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row0 = 8 * i + 4 * kg;
                float* o = rad_feat + so * out_stride + row0;
                if (row0 + 4 <= out_stride && (out_stride & 3) == 0)
                    *reinterpret_cast<float4*>(o) = make_float4(acc[4 * i], acc[4 * i + 1], acc[4 * i + 2], acc[4 * i + 3]);
                else for (int r = 0; r < 4; ++r) if (row0 + r < out_stride) o[r] = acc[4 * i + r];
            }
        }
    }
}

template <int C4>
static int launch_app_bf16(const TirField* f, const float* xyz, const int32_t* li, const int32_t* map,
                           float* rad, float* intr, int stride, int idx_div, int64_t n, const int32_t* n_dev, hipStream_t s) {
    const int nx = (rad ? 1 : 0) + (intr ? 1 : 0);
    const size_t lds = (size_t)2 * 3 * 2 * 2 * 64 * 16 + (size_t)4 * nx * 16 * TIR_XS * sizeof(float);
    int64_t blocks = (n + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    const int xcd_on = tir_xcd_mapping(f);
    if (xcd_on) blocks = (blocks + 7) / 8 * 8;
    dim3 g((unsigned)blocks), b(256);
    if (rad && intr) hipLaunchKernelGGL((k_vm_app_bf16<C4, true, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on);
    else if (rad)    hipLaunchKernelGGL((k_vm_app_bf16<C4, true, false>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on);
    else             hipLaunchKernelGGL((k_vm_app_bf16<C4, false, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on);
    return TIR_OK;
}

// the half-split contraction (app_mfma_body HX) for the plain gather: radiance only, both, or intrinsic only
template <int C4>
static int launch_app_hx(const TirField* f, const float* xyz, const int32_t* li, const int32_t* map,
                         float* rad, float* intr, int stride, int idx_div, int64_t n, const int32_t* n_dev, hipStream_t s) {
    constexpr int CA = C4 * 4;
    const int nx = (rad ? 1 : 0) + (intr ? 1 : 0);
    const int lt_rows = (rad && f->n_lights <= 16) ? f->n_lights : 0;
    const size_t lds = ((size_t)(2 * 3 * 2 * 2 * 64 * 16) / 4 + (size_t)(lt_rows + 1) * 3 * CA + 4 * (size_t)(nx * 2 * 16 * TIR_XHX * 2) / 4) * sizeof(float);
    int64_t blocks = (n + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    const int xcd_on = tir_xcd_mapping(f);
    if (xcd_on) blocks = (blocks + 7) / 8 * 8;
    dim3 g((unsigned)blocks), b(256);
    const TirJitter jt{0.0f, 0ull, 0ull, nullptr, nullptr};
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, true, true, false, true>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, true, false, false, true>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, false, true, false, true>), 160 * 1024)) return rc;
    if (rad && intr) hipLaunchKernelGGL((k_vm_app_mfma<C4, true, true, false, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    else if (rad)    hipLaunchKernelGGL((k_vm_app_mfma<C4, true, false, false, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    else             hipLaunchKernelGGL((k_vm_app_mfma<C4, false, true, false, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    return TIR_OK;
}

template <int C4>
static int launch_app(const TirField* f, const float* xyz, const int32_t* li, const int32_t* map,
                      float* rad, float* intr, int stride, int idx_div, int64_t n, const int32_t* n_dev, hipStream_t s, bool valu,
                      const TirJitter& jt = TirJitter{0.0f, 0ull, 0ull, nullptr, nullptr}) {
    if (valu) {
        dim3 g((unsigned)((n + 255) / 256)), b(256);
        if (rad && intr) hipLaunchKernelGGL((k_vm_app_valu<C4, true, true>), g, b, 0, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev);
        else if (rad)    hipLaunchKernelGGL((k_vm_app_valu<C4, true, false>), g, b, 0, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev);
        else             hipLaunchKernelGGL((k_vm_app_valu<C4, false, true>), g, b, 0, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev);
        return TIR_OK;
    }
    constexpr int CA = C4 * 4;
    const int nx = (rad ? 1 : 0) + (intr ? 1 : 0);
    const int lt_rows = (rad && f->n_lights <= 16) ? f->n_lights : 0;         // light_line rows staged in LDS (576 B each)
    const size_t lds = (size_t)(3 * CA * 32 + (lt_rows + 1) * 3 * CA + 4 * nx * CA * TIR_XLD) * sizeof(float);
    int64_t blocks = (n + 63) / 64;
    if (blocks > 2048) blocks = 2048;
    const int xcd_on = tir_xcd_mapping(f);
    if (xcd_on) blocks = (blocks + 7) / 8 * 8;
    dim3 g((unsigned)blocks), b(256);
    if (lds > 160 * 1024) return TIR_ERR_UNSUPPORTED;
    // > 64 KB only for unusually wide fields; harmless otherwise
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, true, true>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, true, false>), 160 * 1024)) return rc;
    if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, false, true>), 160 * 1024)) return rc;
    if (jt.scale != 0.0f) {
        if (rad || !intr) return TIR_ERR_ARG;
        if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_mfma<C4, false, true, true>), 160 * 1024)) return rc;
        hipLaunchKernelGGL((k_vm_app_mfma<C4, false, true, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
        return TIR_OK;
    }
    if (rad && intr) hipLaunchKernelGGL((k_vm_app_mfma<C4, true, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    else if (rad)    hipLaunchKernelGGL((k_vm_app_mfma<C4, true, false>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    else             hipLaunchKernelGGL((k_vm_app_mfma<C4, false, true>), g, b, lds, s, *f, xyz, li, map, rad, intr, stride, idx_div, n, n_dev, xcd_on, jt, lt_rows);
    return TIR_OK;
}

static int app_fwd(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map,
                   float* rad_feat, float* int_feat, int32_t out_stride, int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream,
                   bool valu, bool split_bf16 = false, const TirJitter& jt = TirJitter{0.0f, 0ull, 0ull, nullptr, nullptr}) {
    if (!f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i]) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (out_stride < f->app_dim || out_stride > 32) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz) || (!rad_feat && !int_feat) || (rad_feat && !light_idx)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    hipStream_t s = tir_stream(stream);
    int rc;
    if (split_bf16) {
        switch (f->n_acomp) {
            case 48: rc = launch_app_bf16<12>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s); break;
            case 24: rc = launch_app_bf16<6>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s); break;
            case 16: rc = launch_app_bf16<4>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s); break;
            default: return TIR_ERR_UNSUPPORTED;
        }
        if (rc) return rc;
        TIR_CHECK_LAUNCH();
        return TIR_OK;
    }
    switch (f->n_acomp) {
        case 48: rc = launch_app<12>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s, valu, jt); break;
        case 24: rc = launch_app<6>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s, valu, jt); break;
        case 16: rc = launch_app<4>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s, valu, jt); break;
        case 96: rc = launch_app<24>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, s, valu, jt); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_vm_app_fwd(const TirField* f, const float* xyz, const int32_t* light_idx,
                              const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                              int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream) {
    return app_fwd(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, stream, false);
}

// Intrinsic feature of the JITTERED points xyz + scale * N(0,1) (models/tensorBase_rotated_lights.py:937-938) with the
// noise drawn inside the gather kernel; xyz_out receives the jittered points.
extern "C" int tir_vm_app_jitter_fwd(const TirField* f, const float* xyz, int64_t n, const int32_t* n_dev, float scale,
                                     uint64_t seed, uint64_t offset, const int64_t* rng_dev, float* xyz_out,
                                     float* int_feat, int32_t out_stride, void* stream) {
    if (!xyz_out || !int_feat || scale == 0.0f) return TIR_ERR_ARG;
    TirJitter jt{scale, (unsigned long long)seed, (unsigned long long)offset, reinterpret_cast<const long long*>(rng_dev), xyz_out};
    return app_fwd(f, xyz, nullptr, nullptr, nullptr, int_feat, out_stride, 0, n, n_dev, stream, false, false, jt);
}

static int app_primary_launch(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map, float* rad_feat,
                              float* int_feat, int32_t out_stride, int64_t n, const int32_t* n_dev, float scale, uint64_t seed,
                              uint64_t offset, const int64_t* rng_state, float* xyz_out, float* int_feat_jit, void* stream, bool hx) {
    constexpr int C4 = 12, CA = 48;
    const int lt_rows = f->n_lights <= 16 ? f->n_lights : 0;
    const size_t w_floats = hx ? (2 * 3 * 2 * 2 * 64 * 16) / 4 : 3 * CA * 32;
    const size_t x_floats = hx ? (2 * 2 * 16 * TIR_XHX * 2) / 4 : 2 * CA * TIR_XLD;         // per wave, two feature sets
    const size_t lds = (w_floats + (size_t)(lt_rows + 1) * 3 * CA + 4 * x_floats) * sizeof(float);
    int64_t nb = (n + 63) / 64;
    if (nb > 1024) nb = 1024;
    nb = (nb + 7) / 8 * 8;                                   // both slices start at a multiple of 8 (XCD mapping)
    const int xcd_on = tir_xcd_mapping(f);
    TirJitter jt{scale, (unsigned long long)seed, (unsigned long long)offset, reinterpret_cast<const long long*>(rng_state), xyz_out};
    if (hx) {
        if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_primary<C4, true>), 160 * 1024)) return rc;
        hipLaunchKernelGGL((k_vm_app_primary<C4, true>), dim3((unsigned)(2 * nb)), dim3(256), lds, tir_stream(stream), *f, xyz, light_idx, idx_map,
                           rad_feat, int_feat, int_feat_jit, out_stride, n, n_dev, xcd_on, jt, lt_rows, (int)nb);
    } else {
        if (int rc = tir_allow_dynamic_lds(reinterpret_cast<const void*>(k_vm_app_primary<C4, false>), 160 * 1024)) return rc;
        hipLaunchKernelGGL((k_vm_app_primary<C4, false>), dim3((unsigned)(2 * nb)), dim3(256), lds, tir_stream(stream), *f, xyz, light_idx, idx_map,
                           rad_feat, int_feat, int_feat_jit, out_stride, n, n_dev, xcd_on, jt, lt_rows, (int)nb);
    }
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// tir_vm_app_fwd (both features) + tir_vm_app_jitter_fwd on the same points in one launch (48 appearance components).
extern "C" int tir_vm_app_primary_fwd(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map,
                                      float* rad_feat, float* int_feat, int32_t out_stride, int64_t n, const int32_t* n_dev,
                                      float scale, uint64_t seed, uint64_t offset, const int64_t* rng_state, float* xyz_out,
                                      float* int_feat_jit, void* stream) {
    if (!f || !rad_feat || !int_feat || !xyz_out || !int_feat_jit || !light_idx || scale == 0.0f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i]) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->n_acomp != 48 || f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (out_stride < f->app_dim || out_stride > 32) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    return app_primary_launch(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, n, n_dev, scale, seed, offset, rng_state, xyz_out,
                              int_feat_jit, stream, false);
}

// ... with the basis_mat contraction on fp16 hi + lo operands (three products; app_mfma_body HX): the default primary stage of
// inference under the split-bf16 decoders
extern "C" int tir_vm_app_primary_x3_fwd(const TirField* f, const float* xyz, const int32_t* light_idx, const int32_t* idx_map,
                                         float* rad_feat, float* int_feat, int32_t out_stride, int64_t n, const int32_t* n_dev,
                                         float scale, uint64_t seed, uint64_t offset, const int64_t* rng_state, float* xyz_out,
                                         float* int_feat_jit, void* stream) {
    if (!f || !rad_feat || !int_feat || !xyz_out || !int_feat_jit || !light_idx || scale == 0.0f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i]) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->n_acomp != 48 || f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (out_stride < f->app_dim || out_stride > 32) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    return app_primary_launch(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, n, n_dev, scale, seed, offset, rng_state, xyz_out,
                              int_feat_jit, stream, true);
}

extern "C" int tir_vm_app_fwd_x3(const TirField* f, const float* xyz, const int32_t* light_idx,
                                 const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                                 int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream) {
    if (!f) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !f->aplane[i] || !f->aline[i]) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->app_dim < 1 || f->app_dim > 27) return TIR_ERR_UNSUPPORTED;
    if (out_stride < f->app_dim || out_stride > 32) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && !xyz) || (!rad_feat && !int_feat) || (rad_feat && !light_idx)) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    int rc;
    switch (f->n_acomp) {
        case 48: rc = launch_app_hx<12>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, tir_stream(stream)); break;
        case 24: rc = launch_app_hx<6>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, tir_stream(stream)); break;
        case 16: rc = launch_app_hx<4>(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, tir_stream(stream)); break;
        default: return TIR_ERR_UNSUPPORTED;
    }
    if (rc) return rc;
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

extern "C" int tir_vm_app_fwd_valu(const TirField* f, const float* xyz, const int32_t* light_idx,
                                   const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                                   int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream) {
    return app_fwd(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, stream, true);
}

extern "C" int tir_vm_app_fwd_bf16x3(const TirField* f, const float* xyz, const int32_t* light_idx,
                                     const int32_t* idx_map, float* rad_feat, float* int_feat, int32_t out_stride,
                                     int32_t idx_div, int64_t n, const int32_t* n_dev, void* stream) {
    return app_fwd(f, xyz, light_idx, idx_map, rad_feat, int_feat, out_stride, idx_div, n, n_dev, stream, false, true);
}

// fp32 -> fp16 copies of up to 8 tables in one launch (same element order): the shadow planes / lines of tir_vm_app_fwd_h16
static int pack_half_launch(const float* const* srcs, void* const* dsts, const int64_t* counts, int32_t n_tables, float* absmax, void* stream) {
    if (n_tables < 0 || n_tables > TIR_HALF_MAX_JOBS || (n_tables > 0 && (!srcs || !dsts || !counts))) return TIR_ERR_ARG;
    if (n_tables == 0) return TIR_OK;
    TirHalfJobs jobs;
    int64_t most = 0;
    for (int i = 0; i < n_tables; ++i) {
        if (counts[i] < 0 || (counts[i] > 0 && !srcs[i])) return TIR_ERR_ARG;
        if (counts[i] > 0 && !dsts[i] && !absmax) return TIR_ERR_ARG;           // a scan-only table needs somewhere to report
        if (reinterpret_cast<uintptr_t>(srcs[i]) % 16 != 0 || reinterpret_cast<uintptr_t>(dsts[i]) % 16 != 0) return TIR_ERR_ARG;
        jobs.job[i] = TirHalfJob{srcs[i], dsts[i], counts[i], absmax ? reinterpret_cast<unsigned*>(absmax) + i : nullptr};
        most = counts[i] > most ? counts[i] : most;
    }
    if (most == 0) return TIR_OK;
    const int64_t per_block = 2048 * TIR_PACK_CHUNKS;
    hipLaunchKernelGGL(k_pack_half, dim3((unsigned)((most + per_block - 1) / per_block), (unsigned)n_tables), dim3(256), 0, tir_stream(stream), jobs);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}

// fp32 -> fp16 copies of up to 8 tables in one launch (same element order, saturating): the shadow planes / lines of tir_vm_app_fwd_h16
extern "C" int tir_pack_half(const float* const* srcs, void* const* dsts, const int64_t* counts, int32_t n_tables, void* stream) {
    return pack_half_launch(srcs, dsts, counts, n_tables, nullptr, stream);
}

// ... and the abs-max of every table (absmax[i], device floats the CALLER has zeroed; dsts[i] == NULL: scan table i only)
extern "C" int tir_pack_half_checked(const float* const* srcs, void* const* dsts, const int64_t* counts, int32_t n_tables, float* absmax,
                                     void* stream) {
    if (!absmax) return TIR_ERR_ARG;
    return pack_half_launch(srcs, dsts, counts, n_tables, absmax, stream);
}

extern "C" int tir_vm_app_fwd_h16(const TirField* f, const TirFieldHalf* fh, const float* xyz, const int32_t* light_idx,
                                  const int32_t* idx_map, float* rad_feat, int32_t out_stride, int32_t idx_div, int64_t n,
                                  const int32_t* n_dev, void* stream) {
    if (!f || !fh) return TIR_ERR_ARG;
    for (int i = 0; i < 3; ++i)
        if (f->grid[i] < 2 || !fh->aplane[i] || !fh->aline[i] || reinterpret_cast<uintptr_t>(fh->aplane[i]) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(fh->aline[i]) % 16 != 0) return TIR_ERR_ARG;
    if (!f->basis_t || !f->light_mean || !f->light_line) return TIR_ERR_ARG;
    if (f->n_acomp != 48 || f->app_dim < 1 || f->app_dim > 27 || !tir_app_index_ok(f)) return TIR_ERR_UNSUPPORTED;
    if (out_stride < f->app_dim || out_stride > 32) return TIR_ERR_ARG;
    if (n < 0 || (n > 0 && (!xyz || !rad_feat || !light_idx))) return TIR_ERR_ARG;
    if (n == 0) return TIR_OK;
    const int lt_rows = f->n_lights <= 16 ? f->n_lights : 0;
    const size_t lds = (size_t)(3 * 3 * 2 * 32) * 16 + (size_t)(lt_rows + 1) * 144 * sizeof(float) + (size_t)4 * 32 * TIR_XH * 2;
    int64_t blocks = (n + 127) / 128;
    if (blocks > 2048) blocks = 2048;
    const int xcd_on = tir_xcd_mapping(f);
    if (xcd_on) blocks = (blocks + 7) / 8 * 8;
    hipLaunchKernelGGL(k_vm_app_h16, dim3((unsigned)blocks), dim3(256), lds, tir_stream(stream), *f, *fh, xyz, light_idx, idx_map,
                       rad_feat, out_stride, idx_div, n, n_dev, xcd_on, lt_rows);
    TIR_CHECK_LAUNCH();
    return TIR_OK;
}
