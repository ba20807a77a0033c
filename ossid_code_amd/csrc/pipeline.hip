// The steps either side of the hot path that SURVEY.md 8(f) ranks next, as device kernels (gfx950). All are HBM-bound
// elementwise / reduction work over one 640x480 frame: coalesced rows, wave butterflies for the reductions.
//   (1) DTOID batch producer: processData + bbox + Gaussian heat map   datasets/dtoid_bop_dataset.py:240-338,
//       utils/data.py:7-83, utils/__init__.py:241-255 (depth2xyz), :354-367 (heatmapGaussain)
//   (3) post-score step: visibility mask (bop_toolkit estimate_visib_mask_gt, bop19 mode), mask IoUs, pseudo-label box
//       scripts/online_learning.py:485-500, :557-558; and a depth-only point-splat renderer standing in for pyrender
//       (:485 renderer.render(depth_only=True)) so that the predicted depth never leaves the GPU.
#include "common.h"

namespace {

__device__ __forceinline__ int wave_min_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = min(v, __shfl_xor(v, m));
    return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = max(v, __shfl_xor(v, m));
    return v;
}

// ---- (1a) processData: depth -> xyz map, resize (bilinear, pixel-centre aligned like cv2.INTER_LINEAR), /255, CHW ----
// img u8 [Ho][Wo][3], depth f32 [Ho][Wo], mask f32 [Ho][Wo] in [0,1] -> img_out [3][H][W], xyz_out [3][H][W], mask_out [H][W].
// Same size in and out (the LM-O / YCB-V case: 480x640 native) is an exact copy.
__device__ __forceinline__ void bilin(int d, int n_dst, int n_src, int& i0, int& i1, float& w) {
    if (n_dst == n_src) {
        i0 = i1 = d;
        w = 0.0f;
        return;
    }
    const float s = ((float)d + 0.5f) * ((float)n_src / (float)n_dst) - 0.5f;
    const float f = floorf(s);
    i0 = (int)f;
    w = s - f;
    i1 = i0 + 1;
    if (i0 < 0) { i0 = 0; i1 = 0; w = 0.0f; }
    if (i1 > n_src - 1) { i1 = n_src - 1; if (i0 > n_src - 1) i0 = n_src - 1; }
}

__global__ __launch_bounds__(256) void prep_sample_kernel(const uint8_t* __restrict__ img, const float* __restrict__ depth,
                                                          const float* __restrict__ mask, int Ho, int Wo, float fx,
                                                          float fy, float cx, float cy, int H, int W,
                                                          float* __restrict__ img_out, float* __restrict__ xyz_out,
                                                          float* __restrict__ mask_out) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    int x0, x1, y0, y1;
    float wx, wy;
    bilin(x, W, Wo, x0, x1, wx);
    bilin(y, H, Ho, y0, y1, wy);
    const float w00 = (1.f - wx) * (1.f - wy), w10 = wx * (1.f - wy), w01 = (1.f - wx) * wy, w11 = wx * wy;
    const size_t p00 = (size_t)y0 * Wo + x0, p10 = (size_t)y0 * Wo + x1, p01 = (size_t)y1 * Wo + x0, p11 = (size_t)y1 * Wo + x1;
    const size_t o = (size_t)y * W + x, plane = (size_t)H * W;
    const bool same = (W == Wo) && (H == Ho);
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        float v;
        if (same) {
            v = (float)img[p00 * 3 + c];
        } else {   // interpolate, round to the uint8 the reference's cv2.resize returns, then scale
            v = ((float)img[p00 * 3 + c] * w00 + (float)img[p10 * 3 + c] * w10) +
                ((float)img[p01 * 3 + c] * w01 + (float)img[p11 * 3 + c] * w11);
            v = floorf(v + 0.5f);
        }
        img_out[c * plane + o] = v / 255.0f;
    }
    // depth2xyz at the ORIGINAL resolution and intrinsics (utils/data.py:29), then resized like the image
    auto xyz_at = [&](int yy, int xx, float& X, float& Y, float& Z) {
        const float z = depth[(size_t)yy * Wo + xx];
        X = ((float)xx - cx) * z / fx;
        Y = ((float)yy - cy) * z / fy;
        Z = z;
    };
    float X, Y, Z;
    if (same) {
        xyz_at(y0, x0, X, Y, Z);
        mask_out[o] = mask[p00];
    } else {
        float a[3], b[3], c4[3], d[3];
        xyz_at(y0, x0, a[0], a[1], a[2]);
        xyz_at(y0, x1, b[0], b[1], b[2]);
        xyz_at(y1, x0, c4[0], c4[1], c4[2]);
        xyz_at(y1, x1, d[0], d[1], d[2]);
        X = (a[0] * w00 + b[0] * w10) + (c4[0] * w01 + d[0] * w11);
        Y = (a[1] * w00 + b[1] * w10) + (c4[1] * w01 + d[1] * w11);
        Z = (a[2] * w00 + b[2] * w10) + (c4[2] * w01 + d[2] * w11);
        mask_out[o] = (mask[p00] * w00 + mask[p10] * w10) + (mask[p01] * w01 + mask[p11] * w11);
    }
    xyz_out[o] = X;
    xyz_out[plane + o] = Y;
    xyz_out[2 * plane + o] = Z;
}

// ---- (1b) mask -> bounding box (min/max of the non-zero pixels), one workgroup, wave butterflies ---------------------
__global__ __launch_bounds__(1024) void mask_bbox_kernel(const float* __restrict__ mask, int H, int W, int* __restrict__ box) {
    __shared__ int red[4][16];
    int x1 = 1 << 30, y1 = 1 << 30, x2 = -1, y2 = -1;
    for (int i = threadIdx.x; i < H * W; i += 1024)
        if (mask[i] != 0.0f) {
            const int y = i / W, x = i - y * W;
            x1 = min(x1, x), y1 = min(y1, y), x2 = max(x2, x), y2 = max(y2, y);
        }
    x1 = wave_min_i32(x1), y1 = wave_min_i32(y1), x2 = wave_max_i32(x2), y2 = wave_max_i32(y2);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[0][wv] = x1, red[1][wv] = y1, red[2][wv] = x2, red[3][wv] = y2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 16; ++k) {
            red[0][0] = min(red[0][0], red[0][k]), red[1][0] = min(red[1][0], red[1][k]);
            red[2][0] = max(red[2][0], red[2][k]), red[3][0] = max(red[3][0], red[3][k]);
        }
        box[0] = red[0][0], box[1] = red[1][0], box[2] = red[2][0], box[3] = red[3][0];
        box[4] = red[2][0] >= 0 ? 1 : -1;                  // label 1 = object; -1 = empty mask, no box (padding label)
    }
}

// ---- (1c) Gaussian heat map around the box centre, float64 like the numpy reference -----------------------------------
__global__ __launch_bounds__(256) void heatmap_kernel(const int* __restrict__ box, double scale, double sigma, int hh, int hw,
                                                      double* __restrict__ heat) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= hh * hw) return;
    const int y = i / hw, x = i - y * hw;
    const double cx = ((double)box[0] + (double)box[2]) / 2.0 * scale, cy = ((double)box[1] + (double)box[3]) / 2.0 * scale;
    const double dx = (double)x - cx, dy = (double)y - cy;
    const double dst = sqrt(dx * dx + dy * dy);
    heat[i] = box[4] > 0 ? exp(-(dst * dst / (2.0 * sigma * sigma))) : 0.0;
}

// ---- (3a) depth-only point-splat renderer: z-buffer by atomicMin on the float bits (positive floats order like uints) ----
__global__ __launch_bounds__(256) void splat_clear_kernel(unsigned* __restrict__ zbuf, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) zbuf[i] = 0x7f800000u;   // +inf
}
__global__ __launch_bounds__(256) void splat_points_kernel(const float* __restrict__ T, const float* __restrict__ pts, int M,
                                                           float fx, float fy, float cx, float cy, int H, int W, int radius,
                                                           unsigned* __restrict__ zbuf) {
    const int m = blockIdx.x * 256 + threadIdx.x;
    if (m >= M) return;
    const float x = pts[3 * m], y = pts[3 * m + 1], z = pts[3 * m + 2];
    const float X = (T[0] * x + T[1] * y) + T[2] * z + T[3];
    const float Y = (T[4] * x + T[5] * y) + T[6] * z + T[7];
    const float Z = (T[8] * x + T[9] * y) + T[10] * z + T[11];
    if (!(Z > 1e-6f)) return;
    const float uf = (X / Z) * fx + cx, vf = (Y / Z) * fy + cy;
    if (!(fabsf(uf) < 1e9f) || !(fabsf(vf) < 1e9f)) return;
    const int u = (int)floorf(uf), v = (int)floorf(vf);
    const unsigned zb = __float_as_uint(Z);
    for (int dv = -radius; dv <= radius; ++dv)
        for (int du = -radius; du <= radius; ++du) {
            const int uu = u + du, vv = v + dv;
            if (uu >= 0 && uu < W && vv >= 0 && vv < H) atomicMin(&zbuf[(size_t)vv * W + uu], zb);
        }
}
__global__ __launch_bounds__(256) void splat_resolve_kernel(const unsigned* __restrict__ zbuf, int n, float* __restrict__ depth) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) depth[i] = zbuf[i] == 0x7f800000u ? 0.0f : __uint_as_float(zbuf[i]);
}

// ---- (3b) visibility mask (bop19) + the four set sizes of the two IoUs + pseudo-label box ------------------------------
// visib = (d_pred - d_obs <= delta  or  d_obs == 0) and d_pred > 0        (bop_toolkit_lib.visibility, mode 'bop19')
// counts[0..3] = |pred & gt|, |pred | gt|, |visib & gt_visib|, |visib | gt_visib|   (online_learning.py:557-558)
__global__ __launch_bounds__(256) void visib_iou_kernel(const float* __restrict__ d_obs, const float* __restrict__ d_pred,
                                                        const uint8_t* __restrict__ gt, const uint8_t* __restrict__ gt_visib,
                                                        int n, float delta, uint8_t* __restrict__ pred_mask,
                                                        uint8_t* __restrict__ visib_mask, int* __restrict__ counts) {
    __shared__ int red[4][4];
    int c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float dp = d_pred[i], dob = d_obs[i];
        const bool pm = dp > 0.0f;
        const bool vm = ((dp - dob) <= delta || dob == 0.0f) && pm;
        pred_mask[i] = pm, visib_mask[i] = vm;
        const bool g = gt && gt[i] > 0, gv = gt_visib && gt_visib[i] > 0;
        c0 += pm && g, c1 += pm || g, c2 += vm && gv, c3 += vm || gv;
    }
    c0 = wave_sum_i32(c0), c1 = wave_sum_i32(c1), c2 = wave_sum_i32(c2), c3 = wave_sum_i32(c3);
    const int wv = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) red[0][wv] = c0, red[1][wv] = c1, red[2][wv] = c2, red[3][wv] = c3;
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(&counts[threadIdx.x], red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

}  // namespace

extern "C" {

int ossid_dtoid_prep_sample(const uint8_t* img, const float* depth, const float* mask, int Ho, int Wo, float fx, float fy,
                            float cx, float cy, int H, int W, float* img_out, float* xyz_out, float* mask_out, void* stream) {
    if (!img || !depth || !mask || !img_out || !xyz_out || !mask_out || Ho <= 0 || Wo <= 0 || H <= 0 || W <= 0) return OSSID_EINVAL;
    hipLaunchKernelGGL(prep_sample_kernel, dim3((W + 255) / 256, H), dim3(256), 0, (hipStream_t)stream, img, depth, mask, Ho, Wo,
                       fx, fy, cx, cy, H, W, img_out, xyz_out, mask_out);
    return ossid_launch_status();
}

int ossid_mask_bbox_heatmap(const float* mask, int H, int W, int heat_h, int heat_w, double heat_scale, double sigma,
                            int32_t* bbox5, double* heatmap, void* stream) {
    if (!mask || !bbox5 || H <= 0 || W <= 0) return OSSID_EINVAL;
    hipLaunchKernelGGL(mask_bbox_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, mask, H, W, bbox5);
    if (heatmap) {
        if (heat_h <= 0 || heat_w <= 0) return OSSID_EINVAL;
        hipLaunchKernelGGL(heatmap_kernel, dim3((heat_h * heat_w + 255) / 256), dim3(256), 0, (hipStream_t)stream, bbox5,
                           heat_scale, sigma, heat_h, heat_w, heatmap);
    }
    return ossid_launch_status();
}

int ossid_render_depth_points(const float* transform, const float* points, int M, float fx, float fy, float cx, float cy,
                              int H, int W, int radius, void* zbuf_workspace, float* depth_out, void* stream) {
    if (!transform || !points || !zbuf_workspace || !depth_out || M < 0 || H <= 0 || W <= 0 || radius < 0 || radius > 8) return OSSID_EINVAL;
    const int n = H * W;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(splat_clear_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (unsigned*)zbuf_workspace, n);
    if (M > 0)
        hipLaunchKernelGGL(splat_points_kernel, dim3((M + 255) / 256), dim3(256), 0, s, transform, points, M, fx, fy, cx, cy, H,
                           W, radius, (unsigned*)zbuf_workspace);
    hipLaunchKernelGGL(splat_resolve_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const unsigned*)zbuf_workspace, n, depth_out);
    return ossid_launch_status();
}

int ossid_visib_mask_iou(const float* depth_obs, const float* depth_pred, const uint8_t* gt_mask, const uint8_t* gt_mask_visib,
                         int H, int W, float delta, uint8_t* pred_mask, uint8_t* pred_mask_visib, int32_t* counts4,
                         void* stream) {
    if (!depth_obs || !depth_pred || !pred_mask || !pred_mask_visib || !counts4 || H <= 0 || W <= 0) return OSSID_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (hipMemsetAsync(counts4, 0, 16, s) != hipSuccess) return OSSID_ELAUNCH;
    const int n = H * W;
    int blocks = (n + 255) / 256;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(visib_iou_kernel, dim3(blocks), dim3(256), 0, s, depth_obs, depth_pred, gt_mask, gt_mask_visib, n, delta,
                       pred_mask, pred_mask_visib, counts4);
    return ossid_launch_status();
}

}  // extern "C"
