// Zephyr featurizer for gfx950: frame staging (blur + RGB-D interleave), model table, projection,
// free-space-violation count, and the per-(hypothesis, point) error features.
//
// Reference interfaces stood behind (paths under /root/reference/python/ossid):
//   utils/zephyr_utils.py:13-14  cv2.GaussianBlur(5x5, sigma 0) + /255
//   utils/zephyr_utils.py:58     zephyr.utils.projectPointsUv
//   utils/zephyr_utils.py:31     ScoreDataset.getPointNetData (dataset="HSVD_diff_uv_norm",
//                                scripts/online_learning.py:191-196)
// Arithmetic follows SPEC.md / oracle/zephyr_oracle.c operation for operation (the file is
// compiled with -ffp-contract=off; IEEE divide and sqrt are hipcc defaults).
//
// Layout in HBM: rgbd [H][W] float4 (r,g,b,depth) -- one 16-B gather per projected point, the
// 4.9 MB frame stays L2/Infinity-Cache resident across all hypotheses; model table [M] 3xfloat4;
// point_x [N'][M] 2xfloat4, written once with 16-B stores. The kernel is store-bound:
// 32 B (+8 B uv) per (hypothesis, point) against ~150 flops.
#include "common.h"

namespace {

struct Cam3 {
    float x, y, z;
};

__device__ __forceinline__ Cam3 rot3(const float* __restrict__ T, float x, float y, float z) {
    Cam3 o;
    o.x = (T[0] * x + T[1] * y) + T[2] * z;
    o.y = (T[4] * x + T[5] * y) + T[6] * z;
    o.z = (T[8] * x + T[9] * y) + T[10] * z;
    return o;
}

struct Proj {
    Cam3 cam;
    float uf, vf;
    int u, v;
};

__device__ __forceinline__ Proj project1(const float* __restrict__ T, float px, float py, float pz, float fx,
                                         float fy, float cx, float cy) {
    Proj r;
    Cam3 c = rot3(T, px, py, pz);
    c.x = c.x + T[3];
    c.y = c.y + T[7];
    c.z = c.z + T[11];
    r.cam = c;
    bool ok = c.z > 1e-6f;
    float a = 0.0f, b = 0.0f;
    if (ok) {
        a = (c.x / c.z) * fx + cx;
        b = (c.y / c.z) * fy + cy;
        ok = isfinite(a) && isfinite(b) && fabsf(a) < 1.0e9f && fabsf(b) < 1.0e9f;
    }
    r.uf = a;
    r.vf = b;
    r.u = ok ? (int)a : -1;
    r.v = ok ? (int)b : -1;
    return r;
}

__device__ __forceinline__ void rgb2hsv(float r, float g, float b, float& h, float& s, float& v) {
    float mx = fmaxf(r, fmaxf(g, b));
    float mn = fminf(r, fminf(g, b));
    float delta = mx - mn;
    float ss = 0.0f, hh = 0.0f;
    if (mx > 0.0f) ss = delta / mx;
    if (delta > 0.0f) {
        if (r == mx)
            hh = (g - b) / delta;
        else if (g == mx)
            hh = 2.0f + (b - r) / delta;
        else
            hh = 4.0f + (r - g) / delta;
        hh = hh / 6.0f;
        if (hh < 0.0f) hh = hh + 1.0f;
    }
    h = hh;
    s = ss;
    v = mx;
}

__device__ __forceinline__ int clampi(int x, int lo, int hi) { return x < lo ? lo : (x > hi ? hi : x); }

__device__ __forceinline__ float4 observe(const float4* __restrict__ rgbd, int H, int W, int u, int v, float uf,
                                          float vf, bool interp) {
    float4 c = rgbd[(size_t)v * W + u];
    if (!interp) return c;
    float xf = uf - 0.5f, yf = vf - 0.5f;
    float x0f = floorf(xf), y0f = floorf(yf);
    float wx = xf - x0f, wy = yf - y0f;
    int x0 = clampi((int)x0f, 0, W - 1), x1 = clampi((int)x0f + 1, 0, W - 1);
    int y0 = clampi((int)y0f, 0, H - 1), y1 = clampi((int)y0f + 1, 0, H - 1);
    float4 a = rgbd[(size_t)y0 * W + x0], b = rgbd[(size_t)y0 * W + x1];
    float4 cc = rgbd[(size_t)y1 * W + x0], d = rgbd[(size_t)y1 * W + x1];
    float w00 = (1.0f - wx) * (1.0f - wy), w10 = wx * (1.0f - wy);
    float w01 = (1.0f - wx) * wy, w11 = wx * wy;
    float4 o;
    o.x = ((a.x * w00 + b.x * w10) + cc.x * w01) + d.x * w11;
    o.y = ((a.y * w00 + b.y * w10) + cc.y * w01) + d.y * w11;
    o.z = ((a.z * w00 + b.z * w10) + cc.z * w01) + d.z * w11;
    if (a.w > 0.0f && b.w > 0.0f && cc.w > 0.0f && d.w > 0.0f)
        o.w = ((a.w * w00 + b.w * w10) + cc.w * w01) + d.w * w11;
    else
        o.w = c.w;
    return o;
}

__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) {
        if (i < 0) i = -i;
        if (i >= n) i = 2 * n - 2 - i;
    }
    return i;
}

// ---- frame staging ------------------------------------------------------------------------------
// 32x8 output pixels per workgroup; the (32+4)x(8+4) u8 RGB halo tile is staged in LDS once and each
// thread forms its 5x5 binomial sum from LDS (exact integer arithmetic, one rounding).
constexpr int TX = 32, TY = 8, HALO = 2;
__global__ __launch_bounds__(TX* TY) void prep_frame_u8_kernel(const uint8_t* __restrict__ img,
                                                                 const float* __restrict__ depth, int H, int W,
                                                                 int blur, float4* __restrict__ rgbd) {
    __shared__ uint8_t tile[(TY + 2 * HALO) * (TX + 2 * HALO) * 3];
    const int tx = threadIdx.x % TX, ty = threadIdx.x / TX;
    const int x0 = blockIdx.x * TX - HALO, y0 = blockIdx.y * TY - HALO;
    constexpr int TW = TX + 2 * HALO, TH = TY + 2 * HALO;
    for (int i = threadIdx.x; i < TW * TH; i += TX * TY) {
        int lx = i % TW, ly = i / TW;
        int gx = reflect101(x0 + lx, W), gy = reflect101(y0 + ly, H);
        const uint8_t* s = img + ((size_t)gy * W + gx) * 3;
        tile[i * 3 + 0] = s[0];
        tile[i * 3 + 1] = s[1];
        tile[i * 3 + 2] = s[2];
    }
    __syncthreads();
    const int x = blockIdx.x * TX + tx, y = blockIdx.y * TY + ty;
    if (x >= W || y >= H) return;
    int out[3];
    if (blur) {
        const int k[5] = {1, 4, 6, 4, 1};
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            int S = 0;
#pragma unroll
            for (int dy = 0; dy < 5; ++dy) {
                int rs = 0;
#pragma unroll
                for (int dx = 0; dx < 5; ++dx) rs += k[dx] * (int)tile[((ty + dy) * TW + tx + dx) * 3 + c];
                S += k[dy] * rs;
            }
            out[c] = (S + 128) >> 8;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 3; ++c) out[c] = tile[((ty + HALO) * TW + tx + HALO) * 3 + c];
    }
    float4 o;
    o.x = (float)out[0] / 255.0f;
    o.y = (float)out[1] / 255.0f;
    o.z = (float)out[2] / 255.0f;
    o.w = depth[(size_t)y * W + x];
    rgbd[(size_t)y * W + x] = o;
}

__global__ __launch_bounds__(256) void prep_frame_f32_kernel(const float* __restrict__ rgb,
                                                              const float* __restrict__ depth, int n,
                                                              float4* __restrict__ rgbd) {
    int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float4 o;
    o.x = rgb[3 * (size_t)i];
    o.y = rgb[3 * (size_t)i + 1];
    o.z = rgb[3 * (size_t)i + 2];
    o.w = depth[i];
    rgbd[i] = o;
}

__global__ __launch_bounds__(256) void prep_model_kernel(const float* __restrict__ pts,
                                                          const float* __restrict__ nrm,
                                                          const float* __restrict__ rgb, int M,
                                                          float4* __restrict__ tab) {
    int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    float h, s, v;
    rgb2hsv(rgb[3 * m], rgb[3 * m + 1], rgb[3 * m + 2], h, s, v);
    tab[3 * (size_t)m + 0] = make_float4(pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], nrm[3 * m]);
    tab[3 * (size_t)m + 1] = make_float4(nrm[3 * m + 1], nrm[3 * m + 2], h, s);
    tab[3 * (size_t)m + 2] = make_float4(v, 0.0f, 0.0f, 0.0f);
}

// ---- Z1 -----------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void project_uv_kernel(const float* __restrict__ T,
                                                          const float* __restrict__ pts, int M, float fx,
                                                          float fy, float cx, float cy, int2* __restrict__ uv) {
    const float* Tn = T + 16 * (size_t)blockIdx.y;
    int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    Proj p = project1(Tn, pts[3 * m], pts[3 * m + 1], pts[3 * m + 2], fx, fy, cx, cy);
    uv[(size_t)blockIdx.y * M + m] = make_int2(p.u, p.v);
}

// ---- Z2 (a): free-space-violation count, one workgroup per hypothesis -----------------------------
__global__ __launch_bounds__(256) void inconst_count_kernel(const float4* __restrict__ rgbd, int H, int W,
                                                             const float* __restrict__ T,
                                                             const float4* __restrict__ tab, int M, float fx,
                                                             float fy, float cx, float cy, float margin,
                                                             int* __restrict__ count) {
    __shared__ int red[4];
    const float* Tn = T + 16 * (size_t)blockIdx.x;
    int cnt = 0;
    for (int m = threadIdx.x; m < M; m += 256) {
        float4 t0 = tab[3 * (size_t)m];
        Proj p = project1(Tn, t0.x, t0.y, t0.z, fx, fy, cx, cy);
        bool inb = (p.u >= 0) && (p.u < W) && (p.v >= 0) && (p.v < H);
        if (inb) {
            float d = rgbd[(size_t)p.v * W + p.u].w;
            if (d > 0.0f && (d - p.cam.z) > margin) ++cnt;
        }
    }
    cnt = wave_sum_i32(cnt);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = cnt;
    __syncthreads();
    if (threadIdx.x == 0) count[blockIdx.x] = red[0] + red[1] + red[2] + red[3];
}

// ---- Z2 (b): features, one workgroup per selected hypothesis --------------------------------------
// Three sweeps over the M points. The projection is recomputed in each (~40 flops) instead of being
// parked in LDS: the kernel is bound by its 40 B/point of stores, the recompute is free, and no M limit
// or LDS budget constrains occupancy. Sweep 1: integer pixel sums (mean). Sweep 2: extent (needs the
// mean). Sweep 3: gather + features + the two 16-B row stores.
__global__ __launch_bounds__(256) void featurize_kernel(const float4* __restrict__ rgbd, int H, int W,
                                                         const float* __restrict__ T,
                                                         const int* __restrict__ sel,
                                                         const float4* __restrict__ tab, int M, float fx,
                                                         float fy, float cx, float cy, int interp,
                                                         float4* __restrict__ point_x, int2* __restrict__ uv_out) {
    __shared__ int red_u[4], red_v[4];
    __shared__ float red_e[4];
    const int hyp = sel ? sel[blockIdx.x] : blockIdx.x;
    const float* Tn = T + 16 * (size_t)hyp;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;

    int su = 0, sv = 0;
    for (int m = threadIdx.x; m < M; m += 256) {
        float4 t0 = tab[3 * (size_t)m];
        Proj p = project1(Tn, t0.x, t0.y, t0.z, fx, fy, cx, cy);
        bool inb = (p.u >= 0) && (p.u < W) && (p.v >= 0) && (p.v < H);
        su += inb ? p.u : 0;
        sv += inb ? p.v : 0;
    }
    su = wave_sum_i32(su);
    sv = wave_sum_i32(sv);
    if (lane == 0) {
        red_u[wave] = su;
        red_v[wave] = sv;
    }
    __syncthreads();
    su = (red_u[0] + red_u[1]) + (red_u[2] + red_u[3]);
    sv = (red_v[0] + red_v[1]) + (red_v[2] + red_v[3]);
    const float mu = (float)su / (float)M, mv = (float)sv / (float)M;

    float ext = 0.0f;
    for (int m = threadIdx.x; m < M; m += 256) {
        float4 t0 = tab[3 * (size_t)m];
        Proj p = project1(Tn, t0.x, t0.y, t0.z, fx, fy, cx, cy);
        bool inb = (p.u >= 0) && (p.u < W) && (p.v >= 0) && (p.v < H);
        int u = inb ? p.u : 0, v = inb ? p.v : 0;
        ext = fmaxf(ext, fabsf((float)u - mu));
        ext = fmaxf(ext, fabsf((float)v - mv));
    }
    ext = wave_max_f32(ext);
    if (lane == 0) red_e[wave] = ext;
    __syncthreads();
    ext = fmaxf(fmaxf(red_e[0], red_e[1]), fmaxf(red_e[2], red_e[3]));
    if (!(ext > 0.0f)) ext = 1.0f;

    float4* out = point_x + (size_t)blockIdx.x * M * 2;
    for (int m = threadIdx.x; m < M; m += 256) {
        float4 t0 = tab[3 * (size_t)m], t1 = tab[3 * (size_t)m + 1], t2 = tab[3 * (size_t)m + 2];
        Proj p = project1(Tn, t0.x, t0.y, t0.z, fx, fy, cx, cy);
        if (uv_out) uv_out[(size_t)blockIdx.x * M + m] = make_int2(p.u, p.v);
        bool inb = (p.u >= 0) && (p.u < W) && (p.v >= 0) && (p.v < H);
        int u = inb ? p.u : 0, v = inb ? p.v : 0;
        float4 o = observe(rgbd, H, W, u, v, p.uf, p.vf, interp && inb);
        float oh, os, ov;
        rgb2hsv(o.x, o.y, o.z, oh, os, ov);
        float dh = fabsf(oh - t1.z);
        dh = fminf(dh, 1.0f - dh);
        Cam3 nr = rot3(Tn, t0.w, t1.x, t1.y);
        float dot = (nr.x * p.cam.x + nr.y * p.cam.y) + nr.z * p.cam.z;
        float len = sqrtf((p.cam.x * p.cam.x + p.cam.y * p.cam.y) + p.cam.z * p.cam.z);
        float4 a, b;
        a.x = ((float)u - mu) / ext;
        a.y = ((float)v - mv) / ext;
        a.z = 0.0f;
        a.w = dh;
        b.x = os - t1.w;
        b.y = ov - t2.x;
        b.z = (o.w > 0.0f) ? (o.w - p.cam.z) : 0.0f;
        b.w = (len > 0.0f) ? dot / len : 0.0f;
        out[2 * (size_t)m] = a;
        out[2 * (size_t)m + 1] = b;
    }
}

}  // namespace

extern "C" {

int ossid_abi_version(char* arch_out_host, int len) {
    if (arch_out_host && len > 0) {
        hipDeviceProp_t prop;
        int dev = 0;
        arch_out_host[0] = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) {
            int i = 0;
            for (; i < len - 1 && prop.gcnArchName[i]; ++i) arch_out_host[i] = prop.gcnArchName[i];
            arch_out_host[i] = 0;
        }
    }
    return OSSID_ABI_VERSION;
}

int ossid_zephyr_prep_frame_u8(const uint8_t* img_rgb, const float* depth, int H, int W, int blur, float* rgbd,
                               void* stream) {
    if (!img_rgb || !depth || !rgbd || H <= 0 || W <= 0) return OSSID_EINVAL;
    dim3 grid((W + TX - 1) / TX, (H + TY - 1) / TY);
    hipLaunchKernelGGL(prep_frame_u8_kernel, grid, dim3(TX * TY), 0, (hipStream_t)stream, img_rgb, depth, H, W,
                       blur, (float4*)rgbd);
    return ossid_launch_status();
}

int ossid_zephyr_prep_frame_f32(const float* img_rgb, const float* depth, int H, int W, float* rgbd,
                                void* stream) {
    if (!img_rgb || !depth || !rgbd || H <= 0 || W <= 0) return OSSID_EINVAL;
    int n = H * W;
    hipLaunchKernelGGL(prep_frame_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, img_rgb,
                       depth, n, (float4*)rgbd);
    return ossid_launch_status();
}

int ossid_zephyr_prep_model(const float* points, const float* normals, const float* colors_rgb, int M, float* tab,
                            void* stream) {
    if (!points || !normals || !colors_rgb || !tab || M <= 0) return OSSID_EINVAL;
    hipLaunchKernelGGL(prep_model_kernel, dim3((M + 255) / 256), dim3(256), 0, (hipStream_t)stream, points, normals,
                       colors_rgb, M, (float4*)tab);
    return ossid_launch_status();
}

int ossid_zephyr_project_uv(const float* transforms, const float* points, int N, int M, float fx, float fy,
                            float cx, float cy, int32_t* uv, void* stream) {
    if (N < 0 || M < 0 || N > 65535) return OSSID_EINVAL;
    if (N == 0 || M == 0) return OSSID_OK;
    if (!transforms || !points || !uv) return OSSID_EINVAL;
    hipLaunchKernelGGL(project_uv_kernel, dim3((M + 255) / 256, N), dim3(256), 0, (hipStream_t)stream, transforms,
                       points, M, fx, fy, cx, cy, (int2*)uv);
    return ossid_launch_status();
}

int ossid_zephyr_inconst_count(const float* rgbd, int H, int W, const float* transforms, int N, const float* tab,
                               int M, float fx, float fy, float cx, float cy, float margin, int32_t* count,
                               void* stream) {
    if (N < 0 || M <= 0 || H <= 0 || W <= 0) return OSSID_EINVAL;
    if (N == 0) return OSSID_OK;
    if (!rgbd || !transforms || !tab || !count) return OSSID_EINVAL;
    hipLaunchKernelGGL(inconst_count_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const float4*)rgbd, H, W,
                       transforms, (const float4*)tab, M, fx, fy, cx, cy, margin, count);
    return ossid_launch_status();
}

int ossid_zephyr_featurize(const float* rgbd, int H, int W, const float* transforms, const int32_t* sel, int Nsel,
                           const float* tab, int M, float fx, float fy, float cx, float cy, int interp,
                           float* point_x, int32_t* uv_original, void* stream) {
    if (Nsel < 0 || M <= 0 || H <= 0 || W <= 0) return OSSID_EINVAL;
    if (Nsel == 0) return OSSID_OK;
    if (!rgbd || !transforms || !tab || !point_x) return OSSID_EINVAL;
    hipLaunchKernelGGL(featurize_kernel, dim3(Nsel), dim3(256), 0, (hipStream_t)stream, (const float4*)rgbd, H, W,
                       transforms, sel, (const float4*)tab, M, fx, fy, cx, cy, interp, (float4*)point_x,
                       (int2*)uv_original);
    return ossid_launch_status();
}

}  // extern "C"

// ---- per-hypothesis pose error (the step right before Z0) ------------------------------------------
// Stands behind the list comprehension of scripts/online_learning.py:452,
//   pp_err = [err_func(R, t, R_gt, t_gt, model_points) for mat in poses_all],  err_func = zephyr.utils.metrics.add / adi
// (BOP definitions: ADD = mean_i |(R p_i + t) - (R_gt p_i + t_gt)|, ADI = mean_i min_j |(R p_i + t) - (R_gt p_j + t_gt)|).
// float64 like the numpy reference. One workgroup per hypothesis; the ground-truth cloud is transformed once per
// workgroup into LDS for ADI (M^2 distance evaluations per hypothesis), sums by wave butterfly + LDS combine.
namespace {

__device__ __forceinline__ double wave_sum_f64(double v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}

__device__ __forceinline__ void xform(const double* __restrict__ T, double x, double y, double z, double& ox, double& oy,
                                      double& oz) {
    ox = T[0] * x + T[1] * y + T[2] * z + T[3];
    oy = T[4] * x + T[5] * y + T[6] * z + T[7];
    oz = T[8] * x + T[9] * y + T[10] * z + T[11];
}

template <bool SYM>
__global__ __launch_bounds__(256) void pose_error_kernel(const double* __restrict__ T, const double* __restrict__ Tgt,
                                                         const double* __restrict__ pts, int M,
                                                         double* __restrict__ err) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* gt = (double*)smem_raw;   // [M][3] ground-truth points (ADI only)
    __shared__ double red[4];
    const double* Tn = T + 16 * (size_t)blockIdx.x;
    if (SYM) {
        for (int j = threadIdx.x; j < M; j += 256)
            xform(Tgt, pts[3 * j], pts[3 * j + 1], pts[3 * j + 2], gt[3 * j], gt[3 * j + 1], gt[3 * j + 2]);
        __syncthreads();
    }
    double s = 0.0;
    for (int i = threadIdx.x; i < M; i += 256) {
        double ex, ey, ez;
        xform(Tn, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], ex, ey, ez);
        if (SYM) {
            double best = 1e300;
            for (int j = 0; j < M; ++j) {
                const double dx = ex - gt[3 * j], dy = ey - gt[3 * j + 1], dz = ez - gt[3 * j + 2];
                best = fmin(best, dx * dx + dy * dy + dz * dz);
            }
            s += sqrt(best);
        } else {
            double gx, gy, gz;
            xform(Tgt, pts[3 * i], pts[3 * i + 1], pts[3 * i + 2], gx, gy, gz);
            const double dx = ex - gx, dy = ey - gy, dz = ez - gz;
            s += sqrt(dx * dx + dy * dy + dz * dz);
        }
    }
    s = wave_sum_f64(s);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) err[blockIdx.x] = ((red[0] + red[1]) + (red[2] + red[3])) / (double)M;
}

}  // namespace

extern "C" int ossid_pose_errors(const double* transforms, const double* transform_gt, const double* points, int N, int M,
                                 int symmetric, double* err, void* stream) {
    if (N < 0 || M <= 0) return OSSID_EINVAL;
    if (N == 0) return OSSID_OK;
    if (!transforms || !transform_gt || !points || !err) return OSSID_EINVAL;
    if (symmetric) {
        const size_t lds = (size_t)M * 24;
        if (lds > 150 * 1024) return OSSID_EINVAL;
        OSSID_ENSURE_LDS(pose_error_kernel<true>, lds);
        hipLaunchKernelGGL(pose_error_kernel<true>, dim3(N), dim3(256), lds, (hipStream_t)stream, transforms, transform_gt,
                           points, M, err);
    } else {
        hipLaunchKernelGGL(pose_error_kernel<false>, dim3(N), dim3(256), 0, (hipStream_t)stream, transforms, transform_gt,
                           points, M, err);
    }
    return ossid_launch_status();
}
