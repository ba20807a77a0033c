// Kernels of the DTOID finetune step (scripts/online_learning.py:650-679: model.train(); loss.backward(); optimizer.step())
// that are not a forward convolution: everything here works on channels-last activations [rows = B*H*W][channels].
//
//   chan_op_kernel      one generic pass  out (+)= (alpha[c]*g + beta[c]*x + kappa[c]) * mask(x)  with optional per-channel
//                       column sums -- BatchNorm batch statistics, BatchNorm / ReLU / ELU backward, the dense block's
//                       gradient accumulation: all are instances (see include/ossid_hip.h, ossid_chan_op)
//   bn_fold_*           training-mode BatchNorm folded to a per-channel (scale, shift) that the NEXT convolution applies
//                       while staging its input (csrc/conv.hip prologue) + its backward
//   avgpool2_*          DenseNet transitions (network.py:165: the third one with stride 1)
//   upsample_bwd        gradient of F.interpolate(mode="nearest") (network.py:354-357): window sums
//   wgrad_kernel        weight gradient of the 3x3 / 1x1 convolutions on the f32 matrix cores, operands staged through
//                       LDS, the input's BatchNorm(+ReLU) prologue re-applied on the fly, deterministic split-K
//   pack_dgrad_kernel   conv weights -> MFMA layout of the TRANSPOSED, 180-degree-rotated kernel: the data gradient is
//                       ossid_conv_nhwc_fwd on dy with these
#include <stdlib.h>

#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// =====================================================================================================================
// generic per-channel elementwise pass with column sums
struct ChanOpArgs {
    const float *g, *x;
    float* out;
    const float *alpha, *beta, *kappa, *ms, *mt, *pivot;
    long long n_rows;
    int C, g_cs, x_cs, out_cs, mask_mode, accumulate, sum_mode, rows_per_block;
    float* partials;      // [gridDim.y][2][C]
};


template <int QX>
__global__ __launch_bounds__(256) void chan_op_kernel(const ChanOpArgs A) {
    constexpr int RY = 256 / QX;
    __shared__ float red[RY][QX][8];
    const int tx = threadIdx.x % QX, ty = threadIdx.x / QX;
    const int q = blockIdx.x * QX + tx;
    const bool valid = q * 4 < A.C;
    const int c0 = q * 4;
    // The six per-channel vectors with UNCONDITIONAL loads (an absent one reads g's first row -- any valid address -- and is
    // replaced by its default afterwards): a load under a condition is followed by a wait for it before the next one is issued,
    // and this launch -- ~390 of them on the step's chains -- began with six memory round trips in a row.
    const int c0s = valid ? c0 : 0;
    const float* cp[6] = {A.alpha, A.beta, A.kappa, A.ms, A.mt, A.pivot};
    float4 cv[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) cv[k] = *(const float4*)((cp[k] ? cp[k] : A.g) + c0s);
    __builtin_amdgcn_sched_barrier(0);                      // (all six issued before the first select waits)
    auto sel4 = [&](int k, float dflt) { return (cp[k] && valid) ? cv[k] : make_float4(dflt, dflt, dflt, dflt); };
    const float4 al = sel4(0, 1.f), be = sel4(1, 0.f), ka = sel4(2, 0.f), ms = sel4(3, 1.f), mt = sel4(4, 0.f), pv4 = sel4(5, 0.f);
    const float pv[4] = {pv4.x, pv4.y, pv4.z, pv4.w};
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const long long r0 = (long long)blockIdx.y * A.rows_per_block;
    const long long r1 = r0 + A.rows_per_block < A.n_rows ? r0 + A.rows_per_block : A.n_rows;
    if (valid) {
        const float a4[4] = {al.x, al.y, al.z, al.w}, b4[4] = {be.x, be.y, be.z, be.w}, k4[4] = {ka.x, ka.y, ka.z, ka.w};
        const float ms4[4] = {ms.x, ms.y, ms.z, ms.w}, mt4[4] = {mt.x, mt.y, mt.z, mt.w};
        // U rows per trip with every load of the trip issued before the first use: most launches of the finetune step are a
        // few MB (a dense layer at 30 x 40: 9 600 rows), i.e. a handful of rows per thread -- their time is the number of
        // DEPENDENT memory round trips, not bandwidth
        constexpr int U = 4;
        // (the rows' loads likewise: rows past the block's last re-read it, an absent x / accumulator reads g instead; what is
        // not there is zeroed by a select behind the load, and rows past the end are neither summed nor stored)
        const bool has_x = A.x != nullptr, has_o = A.out && A.accumulate;
        const float* xp = has_x ? A.x : A.g;
        const int xcs = has_x ? A.x_cs : A.g_cs;
        const float* op = has_o ? A.out : A.g;
        const int ocs = has_o ? A.out_cs : A.g_cs;
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
        for (long long rb = r0 + ty; rb < r1; rb += (long long)RY * U) {
            float4 gv[U], xv[U], ov[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = rb + (long long)u * RY;
                const long long rc = r < r1 ? r : r1 - 1;
                gv[u] = *(const float4*)(A.g + rc * A.g_cs + c0);
                const float4 xl = *(const float4*)(xp + rc * xcs + c0), ol = *(const float4*)(op + rc * ocs + c0);
                xv[u] = has_x ? xl : z4;
                ov[u] = has_o ? ol : z4;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long long r = rb + (long long)u * RY;
                if (r >= r1) break;
                const float g[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w}, x[4] = {xv[u].x, xv[u].y, xv[u].z, xv[u].w};
                float res[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float m = 1.0f;
                    if (A.mask_mode == 1) m = (ms4[i] * x[i] + mt4[i] > 0.0f) ? 1.0f : 0.0f;
                    else if (A.mask_mode == 2) m = x[i] > 0.0f ? 1.0f : x[i] + 1.0f;
                    else if (A.mask_mode == 3) m = x[i] > 0.0f ? 1.0f : 0.0f;
                    const float gm = g[i] * m;
                    res[i] = (a4[i] * g[i] + b4[i] * x[i] + k4[i]) * m;
                    if (A.sum_mode == 1) s1[i] += gm, s2[i] += gm * x[i];
                    else if (A.sum_mode == 2) s1[i] += res[i], s2[i] += res[i] * x[i];
                    else if (A.sum_mode == 3) {      // statistics about a per-channel pivot: no E[x^2] - E[x]^2 cancellation
                        const float dlt = g[i] - pv[i];
                        s1[i] += dlt, s2[i] += dlt * dlt;
                    }
                }
                if (A.out) {
                    if (A.accumulate) res[0] += ov[u].x, res[1] += ov[u].y, res[2] += ov[u].z, res[3] += ov[u].w;
                    *(float4*)(A.out + r * A.out_cs + c0) = make_float4(res[0], res[1], res[2], res[3]);
                }
            }
        }
    }
    if (A.sum_mode == 0) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) red[ty][tx][i] = s1[i], red[ty][tx][4 + i] = s2[i];
    __syncthreads();
    if (ty == 0 && valid) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = red[0][tx][i];
        for (int y = 1; y < RY; ++y)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += red[y][tx][i];
        float* p = A.partials + (size_t)blockIdx.y * 2 * A.C + c0;
        *(float4*)p = make_float4(t[0], t[1], t[2], t[3]);
        *(float4*)(p + A.C) = make_float4(t[4], t[5], t[6], t[7]);
    }
}

// =====================================================================================================================
// training-mode BatchNorm as a folded affine. The two column sums of a channel come either finished (sums[c],
// sums[row_stride + c]) or as the P per-block partials an ossid_chan_op launch with defer_finalize left behind
// ([P][2][C]); in the second case this kernel does the finalize itself (same fixed order, in double): one launch less
// per BatchNorm. Block = 64 channels x 4 quarter-sums.
template <int PARTS>
__device__ __forceinline__ void column_sums(const float* sums, int row_stride, const float* partials, int P, int C, int c,
                                            int part, double (&red)[PARTS][256 / PARTS][2], double& s0, double& s1) {
    constexpr int CPB = 256 / PARTS;
    s0 = s1 = 0.0;
    const int col = threadIdx.x % CPB;
    if (P <= 0) {
        if (part == 0 && c < C) s0 = (double)sums[c], s1 = (double)sums[row_stride + c];
        return;
    }
    if (c < C) {
        const int per = (P + PARTS - 1) / PARTS, p0 = part * per, p1 = min(P, p0 + per);
        const float* src = partials + c;
        int p = p0;
        for (; p + 8 <= p1; p += 8) {
            float a[8], b[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) a[k] = src[(size_t)(p + k) * 2 * C], b[k] = src[(size_t)(p + k) * 2 * C + C];
#pragma unroll
            for (int k = 0; k < 8; ++k) s0 += (double)a[k], s1 += (double)b[k];
        }
        for (; p < p1; ++p) s0 += (double)src[(size_t)p * 2 * C], s1 += (double)src[(size_t)p * 2 * C + C];
    }
    red[part][col][0] = s0, red[part][col][1] = s1;
    __syncthreads();
    if (part == 0) {
        s0 = red[0][col][0], s1 = red[0][col][1];
#pragma unroll
        for (int k = 1; k < PARTS; ++k) s0 += red[k][col][0], s1 += red[k][col][1];
    }
}

template <int PARTS>
__global__ __launch_bounds__(256) void bn_fold_fwd_kernel(const float* __restrict__ sums, int sums_row_stride,
                                                          const float* __restrict__ partials, int P,
                                                          const float* __restrict__ pivot, int C, double n,
                                                          const float* __restrict__ gamma, const float* __restrict__ beta,
                                                          float eps, float momentum, float* __restrict__ running_mean,
                                                          float* __restrict__ running_var, float* __restrict__ scale,
                                                          float* __restrict__ shift, float* __restrict__ mean_out,
                                                          float* __restrict__ rstd_out) {
    constexpr int CPB = 256 / PARTS;
    __shared__ double red[PARTS][CPB][2];
    const int c = blockIdx.x * CPB + (threadIdx.x % CPB), part = threadIdx.x / CPB;
    double s0, s1;
    column_sums<PARTS>(sums, sums_row_stride, partials, P, C, c, part, red, s0, s1);
    if (part != 0 || c >= C) return;
    // the sums are about pivot[c] (0 without one): mean = pivot + S1/n, var = S2/n - (S1/n)^2
    const double dm = s0 / n;
    const double mean = (pivot ? (double)pivot[c] : 0.0) + dm;
    double var = s1 / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const float s = (float)((double)g * rstd);
    scale[c] = s;
    shift[c] = (float)((double)b - mean * (double)s);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
    if (running_mean) {   // torch: running = (1 - momentum) * running + momentum * batch (unbiased variance)
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

// The same fold for a dense block's norm1, whose statistics table is finished for the first tail_c0 channels while the LAST
// tail_c channels -- the slab the previous layer just appended -- still are partial rows [P][2][tail_c] of a deferred
// ossid_chan_op (sum_mode 3, pivot = the slab's first row): this launch finalizes them INTO the table (what
// colsum_finalize_p_kernel did as a launch of its own, 58 times per step on the critical chain) and folds all channels.
template <int PARTS>
__global__ __launch_bounds__(256) void bn_fold_fwd_tail_kernel(float* __restrict__ table, int row_stride, int tail_c0,
                                                               const float* __restrict__ partials, int P,
                                                               const float* __restrict__ tail_pivot, int C, double n,
                                                               const float* __restrict__ gamma, const float* __restrict__ beta,
                                                               float eps, float momentum, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ scale,
                                                               float* __restrict__ shift, float* __restrict__ mean_out,
                                                               float* __restrict__ rstd_out) {
    constexpr int CPB = 256 / PARTS;
    __shared__ double red[PARTS][CPB][2];
    const int c = blockIdx.x * CPB + (threadIdx.x % CPB), part = threadIdx.x / CPB;
    const int first = blockIdx.x * CPB;
    double s0, s1;
    if (first + CPB <= tail_c0) {                             // (uniform per block) finished channels: the table as it stands
        if (part != 0 || c >= C) return;
        s0 = (double)table[c], s1 = (double)table[row_stride + c];
    } else {
        // a block never straddles the boundary: tail_c0 is a multiple of 32 >= CPB (host-checked)
        column_sums<PARTS>(nullptr, 0, partials, P, C - tail_c0, c - tail_c0, part, red, s0, s1);
        if (part != 0 || c >= C) return;
        const float f0 = (float)s0, f1 = (float)s1;
        table[c] = f0, table[row_stride + c] = f1, table[2 * row_stride + c] = tail_pivot[c - tail_c0];
        s0 = (double)f0, s1 = (double)f1;                     // (the values every later reader of the table sees)
    }
    const double dm = s0 / n;
    const double mean = (double)table[2 * row_stride + c] + dm;
    double var = s1 / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const float s = (float)((double)g * rstd);
    scale[c] = s;
    shift[c] = (float)((double)b - mean * (double)s);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
    if (running_mean) {
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

// (d scale, d shift) -> d gamma, d beta and the two per-channel coefficients of the statistics' own gradient:
//   dx += cb[c] * x + ck[c]     (= d mean / n + 2 (x - mean) d var / n)
// With P > 0 the pair comes as partial rows: row 0 = d shift (sum of g*m), row 1 = d scale (sum of g*m*x).
template <int PARTS>
__global__ __launch_bounds__(256) void bn_fold_bwd_kernel(const float* __restrict__ dscale, const float* __restrict__ dshift,
                                                          const float* __restrict__ partials, int P,
                                                          const float* __restrict__ gamma, const float* __restrict__ mean,
                                                          const float* __restrict__ rstd, int C, double n,
                                                          float* __restrict__ dgamma, float* __restrict__ dbeta,
                                                          float* __restrict__ cb, float* __restrict__ ck, int accumulate,
                                                          float* __restrict__ zero_row) {
    constexpr int CPB = 256 / PARTS;
    __shared__ double red[PARTS][CPB][2];
    const int c = blockIdx.x * CPB + (threadIdx.x % CPB), part = threadIdx.x / CPB;
    double s0 = 0.0, s1 = 0.0;
    if (P > 0) {
        column_sums<PARTS>(nullptr, 0, partials, P, C, c, part, red, s0, s1);
    } else if (part == 0 && c < C) {
        s0 = (double)dshift[c], s1 = (double)dscale[c];
    }
    if (part != 0 || c >= C) return;
    const double g = gamma ? (double)gamma[c] : 1.0, mu = (double)mean[c], r = (double)rstd[c];
    const double dt = s0, ds = s1 - dt * mu;     // shift = beta - mean * scale
    const double s = g * r;
    const double dmean = -dt * s, dvar = -0.5 * ds * g * r * r * r;
    if (dgamma) dgamma[c] = (float)(ds * r);
    if (dbeta) dbeta[c] = (float)dt;
    const double b = 2.0 * dvar / n, k = dmean / n - 2.0 * mu * dvar / n;
    if (zero_row) zero_row[c] = 0.0f;
    if (accumulate) {
        cb[c] += (float)b;
        ck[c] += (float)k;
    } else {
        cb[c] = (float)b;
        ck[c] = (float)k;
    }
}

// stand-alone finalize of partial rows into a (possibly strided) sums table: 256 / PARTS channels per block, PARTS threads per
// channel. These launches sit on the step's critical chain ~300 times: their time is the number of dependent rounds of loads
// a thread makes (P / PARTS / 8), so PARTS follows P -- 512 partial rows: 64 threads per channel, one round of 8 loads each.
template <int PARTS>
__global__ __launch_bounds__(256) void colsum_finalize_p_kernel(const float* __restrict__ partials, int P, int C,
                                                                float* __restrict__ sums, int row_stride,
                                                                const float* __restrict__ pivot) {
    constexpr int CPB = 256 / PARTS;
    __shared__ double red[PARTS][CPB][2];
    const int c = blockIdx.x * CPB + (threadIdx.x % CPB), part = threadIdx.x / CPB;
    double s0, s1;
    column_sums<PARTS>(nullptr, 0, partials, P, C, c, part, red, s0, s1);
    if (part != 0 || c >= C) return;
    sums[c] = (float)s0;
    sums[row_stride + c] = (float)s1;
    if (pivot) sums[2 * row_stride + c] = pivot[c];          // third row: the pivot the sums are about
}
static int launch_colsum_finalize(const float* partials, int P, int C, float* sums, int row_stride, const float* pivot,
                                  hipStream_t s) {
    if (P > 128)
        hipLaunchKernelGGL(colsum_finalize_p_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, s, partials, P, C, sums, row_stride, pivot);
    else if (P > 16)
        hipLaunchKernelGGL(colsum_finalize_p_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, s, partials, P, C, sums, row_stride, pivot);
    else
        hipLaunchKernelGGL(colsum_finalize_p_kernel<4>, dim3((C + 63) / 64), dim3(256), 0, s, partials, P, C, sums, row_stride, pivot);
    return ossid_launch_status();
}

// =====================================================================================================================
// 2x2 average pooling, stride 1 or 2, channels-last
__global__ __launch_bounds__(256) void avgpool2_fwd_kernel(const float4* __restrict__ x, int H, int W, int C4, int stride,
                                                           int Ho, int Wo, size_t total, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    const float4* p = x + (((size_t)b * H + (size_t)yo * stride) * W + (size_t)xo * stride) * C4 + c;
    const float4 a = p[0], bb = p[C4], cc = p[(size_t)W * C4], d = p[(size_t)W * C4 + C4];
    out[i] = make_float4(0.25f * ((a.x + bb.x) + (cc.x + d.x)), 0.25f * ((a.y + bb.y) + (cc.y + d.y)),
                         0.25f * ((a.z + bb.z) + (cc.z + d.z)), 0.25f * ((a.w + bb.w) + (cc.w + d.w)));
}

__global__ __launch_bounds__(256) void avgpool2_bwd_kernel(const float4* __restrict__ dout, int H, int W, int C4, int stride,
                                                           int Ho, int Wo, size_t total, float4* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int X = (int)(r % W);
    r /= W;
    const int Y = (int)(r % H), b = (int)(r / H);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int dy = 0; dy < 2; ++dy) {
        const int yy = Y - dy;
        if (yy < 0 || yy % stride) continue;
        const int yo = yy / stride;
        if (yo >= Ho) continue;
#pragma unroll
        for (int dxx = 0; dxx < 2; ++dxx) {
            const int xx = X - dxx;
            if (xx < 0 || xx % stride) continue;
            const int xo = xx / stride;
            if (xo >= Wo) continue;
            const float4 v = dout[(((size_t)b * Ho + yo) * Wo + xo) * C4 + c];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    }
    dx[i] = make_float4(0.25f * s.x, 0.25f * s.y, 0.25f * s.z, 0.25f * s.w);
}

// gradient of nearest-neighbour up-sampling [Hs][Ws] -> [H][W]: source pixel (sy, sx) sums the destination pixels that
// read it: rows row_start[sy] .. row_start[sy+1]-1, columns col_start[sx] .. col_start[sx+1]-1 (tables built by the caller
// with the forward's own index formula)
__global__ __launch_bounds__(256) void upsample_bwd_kernel(const float4* __restrict__ dup, int Hs, int Ws, int H, int W, int C4,
                                                           const int* __restrict__ row_start, const int* __restrict__ col_start,
                                                           size_t total, float4* __restrict__ dsrc) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)(i % C4);
    size_t r = i / C4;
    const int sx = (int)(r % Ws);
    r /= Ws;
    const int sy = (int)(r % Hs), b = (int)(r / Hs);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int y = row_start[sy]; y < row_start[sy + 1]; ++y)
        for (int x = col_start[sx]; x < col_start[sx + 1]; ++x) {
            const float4 v = dup[(((size_t)b * H + y) * W + x) * C4 + c];
            s.x += v.x, s.y += v.y, s.z += v.z, s.w += v.w;
        }
    dsrc[i] = s;
}

// every convolution weight of the training step packed in ONE launch (forward and data-gradient layouts): `table` has
// one row per (layer, layout): {w, wpk, first block, Cout, Cin, taps, kind}; a block finds its row by binary search
struct PackRow {
    const float* w;
    float4* wpk;
    long long first_block;      // prefix sum of blocks (256 float4 each)
    int Cout, Cin, taps, kind;  // kind 0: forward layout, 1: data-gradient layout, 2 / 3: their Winograd forms, 4 / 5: 0 / 1 for exact-f32 launches, 6 / 7: for three-way-split launches
};

// A thread packs EVERYTHING that derives from one lane's group of reduction channels of one output row: all taps and all
// bf16 pieces (direct layouts), all 16 transform positions and both pieces (Winograd layouts). Round 3 had one thread per
// 16-byte output unit, each gathering its 8 (direct) or 72 (Winograd) source weights again: 2 reads per weight for the direct
// layouts, 32 for the Winograd ones, 4 bytes at a time -- 1.5 ms per step for 0.5 GB of traffic. Here a weight is read once per
// layout (a thread's 8 x taps source values are contiguous in the forward layouts, 8 runs of `taps` in the data-gradient ones),
// and a wave's stores are whole 1 KB units. The grid and the table are unchanged (first_block counts 256 output units per
// block): a row simply needs fewer of its blocks, the rest return at once.
__global__ __launch_bounds__(256) void pack_all_kernel(const PackRow* __restrict__ table, int n_rows) {
    int lo = 0, hi = n_rows - 1;
    const long long b = blockIdx.x;
    while (lo < hi) {
        const int mid = (lo + hi + 1) >> 1;
        if (table[mid].first_block <= b) lo = mid; else hi = mid - 1;
    }
    const PackRow R = table[lo];
    const size_t i = (size_t)(b - R.first_block) * 256 + threadIdx.x;
    const float* __restrict__ w = R.w;
    const int Cout = R.Cout, Cin = R.Cin;
    if (R.kind == 2 || R.kind == 3) {      // Winograd layouts (csrc/wino.hip)
#ifndef OSSID_WINO_F32
        const int dgrad = R.kind == 3;
        const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, KC = K / 16, M32 = (M + 31) / 32;
        if (i >= (size_t)M32 * KC * 64) return;
        const int lane = (int)(i & 63);
        const size_t r = i >> 6;
        const int ch = (int)(r % KC), mt = (int)(r / KC);
        const int m = mt * 32 + (lane & 31), k0 = ch * 16 + 8 * (lane >> 5);
        union Oct {
            __bf16 hv[8];
            float4 f;
        } hi_o[16], lo_o[16];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float g[9];
            if (m < M) {
                const float* src = dgrad ? w + ((size_t)(k0 + e) * Cin + m) * 9 : w + ((size_t)m * Cin + k0 + e) * 9;
#pragma unroll
                for (int t = 0; t < 9; ++t) g[t] = src[t];
            }
#pragma unroll
            for (int xi = 0; xi < 16; ++xi) {
                float u = 0.0f;
                if (m < M) {                                   // (the arithmetic of ossid_wino_u, on the values loaded above)
                    const int ti = xi >> 2, tj = xi & 3;
                    float t3[3];
#pragma unroll
                    for (int bb = 0; bb < 3; ++bb) {
                        const float g0 = dgrad ? g[8 - bb] : g[bb], g1 = dgrad ? g[5 - bb] : g[3 + bb], g2 = dgrad ? g[2 - bb] : g[6 + bb];
                        t3[bb] = ti == 0 ? g0 : (ti == 1 ? 0.5f * (g0 + g1 + g2) : (ti == 2 ? 0.5f * (g0 - g1 + g2) : g2));
                    }
                    u = tj == 0 ? t3[0] : (tj == 1 ? 0.5f * (t3[0] + t3[1] + t3[2]) : (tj == 2 ? 0.5f * (t3[0] - t3[1] + t3[2]) : t3[2]));
                }
                const __bf16 h = (__bf16)u;
                hi_o[xi].hv[e] = h;
                lo_o[xi].hv[e] = (__bf16)(u - (float)h);
            }
        }
        float4* out = R.wpk + (((size_t)mt * KC + ch) * 16) * 2 * 64 + lane;
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            out[(size_t)(xi * 2 + 0) * 64] = hi_o[xi].f;
            out[(size_t)(xi * 2 + 1) * 64] = lo_o[xi].f;
        }
#else
        const int K8 = (R.kind == 2 ? R.Cin : R.Cout) / 8, M32 = ((R.kind == 2 ? R.Cout : R.Cin) + 31) / 32;
        if (i < (size_t)M32 * K8 * 16 * 64) R.wpk[i] = ossid_wino_pack_quad(R.w, R.Cout, R.Cin, R.kind == 3, i);
#endif
        return;
    }
    if (R.kind < 0 || R.kind > 7) return;
    const int dgrad = R.kind & 1, exact = R.kind >= 6 ? 2 : (R.kind >= 4 ? 1 : 0);
    const int taps = R.taps;
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout, MT = (M + 31) / 32;
    auto at = [&](int m, int k, int tap) {
        return dgrad ? w[((size_t)k * Cin + m) * taps + (taps - 1 - tap)] : w[((size_t)m * Cin + k) * taps + tap];
    };
    if (OSSID_CONV_SB && exact != 1) {     // split forms: [mt][K/16][taps][parts][64 lanes] x 8 bf16
        const int parts = exact == 2 ? 3 : 2, KU = K / 16;
        if (i >= (size_t)MT * KU * 64) return;
        const int lane = (int)(i & 63);
        const size_t r = i >> 6;
        const int u = (int)(r % KU), mt = (int)(r / KU);
        const int m = mt * 32 + (lane & 31), k0 = u * 16 + 8 * (lane >> 5);
        float4* out = R.wpk + (((size_t)mt * KU + u) * taps) * parts * 64 + lane;
        for (int tap = 0; tap < taps; ++tap) {
            union {
                __bf16 hv[8];
                float4 f;
            } o[3];
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                float v = m < M ? at(m, k0 + e, tap) : 0.0f;
#pragma unroll
                for (int p = 0; p < 3; ++p) {
                    const __bf16 pc = (__bf16)v;
                    o[p].hv[e] = pc;
                    v -= (float)pc;
                }
            }
            for (int p = 0; p < parts; ++p) out[((size_t)tap * parts + p) * 64] = o[p].f;
        }
        return;
    }
    {                                      // exact-f32 form: [mt][K/8][taps][64 lanes] x 4 floats
        const int KB = K / 8;
        if (i >= (size_t)MT * KB * 64) return;
        const int lane = (int)(i & 63);
        const size_t r = i >> 6;
        const int kb = (int)(r % KB), mt = (int)(r / KB);
        const int m = mt * 32 + (lane & 31), k0 = kb * 8 + 4 * (lane >> 5);
        float4* out = R.wpk + (((size_t)mt * KB + kb) * taps) * 64 + lane;
        for (int tap = 0; tap < taps; ++tap) {
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = m < M ? at(m, k0 + e, tap) : 0.0f;
            out[(size_t)tap * 64] = make_float4(v[0], v[1], v[2], v[3]);
        }
    }
}

// =====================================================================================================================
// weight gradient: dW[co][ci][tap] = sum_px dY[px][co] * P(X)[px + tap][ci],  P = the forward's input prologue
// (per-channel affine (+ReLU) on real pixels, zero outside the image). GEMM with the PIXELS on K:
// v_mfma_f32_32x32x2_f32, A = dY (rows = output channels, k = two consecutive pixels, one per lane half), B = P(X)
// shifted by the tap (columns = input channels). A workgroup owns a (WM*TM*32) x (WN*TN*32) tile of (co, ci) for one
// kernel row and a share of the row-chunks (split-K); a chunk = KT consecutive pixels
// of one image row, staged through double-buffered LDS with 16-byte global loads ([px][channel]: an operand read is one
// conflict-free ds_read_b32 per lane), next chunk's loads in flight under the current chunk's MFMAs. Partial results go
// to slabs [split][tap][co][ci], summed in a fixed order by wgrad_reduce2_kernel: bit-reproducible, no float atomics.
struct WgradArgs {
    const float *x, *dy, *pre_scale, *pre_shift;
    float* slabs;
    int B, H, W, Cin, Cout, in_cs, dy_cs, pre_relu, KT, chunks_per_row, nsplit, co_blocks, ci_blocks, ntiles;
    int Hs, Ws;             // x is [B][Hs][Ws][..], nearest-neighbour up-sampled to [H][W] on the fly (Hs == H: plain)
    float scale_h, scale_w;
    long long n_chunks;     // B * H * chunks_per_row
};

template <int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
__device__ __forceinline__ void wgrad_body(const WgradArgs& A, const int L) {
    constexpr int CO_T = WM * TM * 32, CI_T = WN * TN * 32;
    constexpr int KX = TAPS == 9 ? 3 : 1;
    constexpr int NT = TM * TN * KX * KYB;                   // accumulator tiles per wave
    constexpr int HALO = TAPS == 9 ? 2 : 0;
    constexpr int KTMAX = 48;
    constexpr int DY4 = CO_T / 4, X4 = CI_T / 4;              // float4 per staged pixel
    constexpr int NLD_DY = (KTMAX * DY4 + 255) / 256, NLD_X = (KYB * (KTMAX + HALO) * X4 + 255) / 256;
    static_assert(WM * WN * WK == 4 && 256 % DY4 == 0 && 256 % X4 == 0, "bad tiling");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KT = A.KT;
    const int dy_buf = KT * CO_T, x_buf = KYB * (KT + HALO) * CI_T;      // floats per buffer
    float* dyl = lds;                                // [2][KT][CO_T]
    float* xl = lds + 2 * dy_buf;                    // [2][KYB][KT + HALO][CI_T]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    // WK > 1 (layers with few channels): the waves of a workgroup share ONE (co, ci) tile and take every WK-th k-step of
    // a chunk; each writes its own slab (slab index split * WK + wk), so there is no cross-wave reduction in the kernel
    const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);

    // ---- which tile / split: consecutive workgroup ids walk the tiles of ONE split (they read the same dY / X rows,
    // which then hit in L2), splits follow each other (placement only -- any mapping computes the same sums)
    const int split = L / A.ntiles;
    int tile = L - split * A.ntiles;
    int ky0 = 0;
    if (TAPS == 9 && KYB == 1) {
        ky0 = tile % 3;
        tile /= 3;
    }
    const int cib = tile % A.ci_blocks, cob = tile / A.ci_blocks;
    const int co0 = cob * CO_T, ci0 = cib * CI_T;

    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // staging maps (fixed per thread): dY element e -> (pixel, co quad); X element e -> (ky row, position, ci quad)
    const int dyq = tid % DY4, xq = tid % X4;
    const bool dyq_ok = co0 + 4 * dyq < A.Cout, xq_ok = ci0 + 4 * xq < A.Cin;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.pre_scale && xq_ok) {
        ps = *(const float4*)(A.pre_scale + ci0 + 4 * xq);
        pt = *(const float4*)(A.pre_shift + ci0 + 4 * xq);
    }
    float4 sdy[NLD_DY], sx[NLD_X];

    const int H = A.H, W = A.W;
    // Chunk geometry (image b, row y, chunk-in-row cr) is stepped, not divided out: a workgroup visits chunks split,
    // split + nsplit, ... and the three quotients of that stride are computed once. Likewise every thread's staging map
    // (which (kernel row, position) of the patch its e-th float4 is) is fixed: computed once, not per chunk. (The first
    // version did a 64-bit divide + modulo per chunk and a runtime divide per staged element per chunk.)
    const int cpr = A.chunks_per_row;
    const int d_cr = A.nsplit % cpr, d_row = A.nsplit / cpr, d_y = d_row % H, d_b = d_row / H;
    auto advance = [&](int& b, int& y, int& cr) {
        cr += d_cr;
        int carry = cr >= cpr ? 1 : 0;
        cr -= carry * cpr;
        y += d_y + carry;
        carry = y >= H ? 1 : 0;
        y -= carry * H;
        b += d_b + carry;
    };
    int xkr[NLD_X], xp[NLD_X];
#pragma unroll
    for (int e = 0; e < NLD_X; ++e) {
        const int idx = (tid + e * 256) / X4;              // (ky row, position)
        xkr[e] = idx / (KT + HALO);
        xp[e] = idx - xkr[e] * (KT + HALO);
    }
    auto stage_load = [&](int b, int y, int cr) {
        const int x0 = cr * KT;
        const float* dyr = A.dy + ((size_t)(b * H + y) * W) * A.dy_cs + co0 + 4 * dyq;
#pragma unroll
        for (int e = 0; e < NLD_DY; ++e) {
            const int p = (tid + e * 256) / DY4;               // pixel within the chunk
            const bool ok = p < KT && x0 + p < W && dyq_ok;
            sdy[e] = ok ? *(const float4*)(dyr + (size_t)(x0 + p) * A.dy_cs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int e = 0; e < NLD_X; ++e) {
            const int kr = xkr[e], p = xp[e];
            const int yy = y + (TAPS == 9 ? ky0 + kr - 1 : 0), xx = x0 + p - (TAPS == 9 ? 1 : 0);
            const bool ok = kr < KYB && yy >= 0 && yy < H && xx >= 0 && xx < W && xq_ok;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                // same index formula as the forward's fused up-sampling (csrc/conv.hip stage map)
                const int sy = (A.Hs == H) ? yy : min((int)floorf((float)yy * A.scale_h), A.Hs - 1);
                const int sxx = (A.Ws == W) ? xx : min((int)floorf((float)xx * A.scale_w), A.Ws - 1);
                v = *(const float4*)(A.x + ((size_t)(b * A.Hs + sy) * A.Ws + sxx) * A.in_cs + ci0 + 4 * xq);
                if (A.pre_scale) {
                    v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                    if (A.pre_relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
            }
            sx[e] = v;
        }
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NLD_DY; ++e) {
            const int idx = tid + e * 256;
            if (idx < KT * DY4) *(float4*)(dyl + (size_t)buf * dy_buf + (size_t)idx * 4) = sdy[e];
        }
#pragma unroll
        for (int e = 0; e < NLD_X; ++e) {
            const int idx = tid + e * 256;
            if (idx < KYB * (KT + HALO) * X4) *(float4*)(xl + (size_t)buf * x_buf + (size_t)idx * 4) = sx[e];
        }
    };

    long long ch = split;
    int gb, gy, gcr;                                          // geometry of the chunk being PREFETCHED
    {
        gcr = split % cpr;
        const int row = split / cpr;
        gy = row % H, gb = row / H;
    }
    if (ch < A.n_chunks) {
        stage_load(gb, gy, gcr);
        stage_write(0);
    }
    __syncthreads();
    int buf = 0;
    const int ksteps = KT / 2;
#pragma unroll 1
    for (; ch < A.n_chunks; ch += A.nsplit) {
        const long long nxt = ch + A.nsplit;
        advance(gb, gy, gcr);
        if (nxt < A.n_chunks) stage_load(gb, gy, gcr);
        const float* da = dyl + (size_t)buf * dy_buf + (size_t)h * CO_T + wm * TM * 32 + c;
        const float* xb = xl + (size_t)buf * x_buf + (size_t)h * CI_T + wn * TN * 32 + c;
        float a[2][TM], bv[2][KYB][TN][KX];
        auto fetch = [&](int k, float(&aa)[TM], float(&bb)[KYB][TN][KX]) {
#pragma unroll
            for (int m = 0; m < TM; ++m) aa[m] = da[(size_t)(2 * k) * CO_T + 32 * m];
#pragma unroll
            for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int kx = 0; kx < KX; ++kx)
                        bb[kr][n][kx] = xb[((size_t)kr * (KT + HALO) + 2 * k + kx) * CI_T + 32 * n];
        };
        auto run = [&](const float(&aa)[TM], const float(&bb)[KYB][TN][KX]) {
#pragma unroll
            for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int kx = 0; kx < KX; ++kx) {
                            const int t = ((kr * TM + m) * TN + n) * KX + kx;
                            acc[t] = mfma(aa[m], bb[kr][n][kx], acc[t]);
                        }
        };
        // operands of this wave's next k-step are read from LDS while the MFMAs of the current one run (two per trip)
        if (wk < ksteps) fetch(wk, a[0], bv[0]);
#pragma unroll 1
        for (int k = wk; k < ksteps; k += 2 * WK) {
            if (k + WK < ksteps) fetch(k + WK, a[1], bv[1]);
            __builtin_amdgcn_sched_barrier(0);
            run(a[0], bv[0]);
            __builtin_amdgcn_sched_barrier(0);
            if (k + WK < ksteps) {
                if (k + 2 * WK < ksteps) fetch(k + 2 * WK, a[0], bv[0]);
                __builtin_amdgcn_sched_barrier(0);
                run(a[1], bv[1]);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (nxt < A.n_chunks) stage_write(buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- slab [split][tap][Cout][Cin]: rows of an accumulator tile = output channels (registers), columns = input
    // channels (lanes): 128-byte rows
    float* slab = A.slabs + ((size_t)split * WK + wk) * TAPS * A.Cout * A.Cin;
#pragma unroll
    for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int ci = ci0 + (wn * TN + n) * 32 + c;
                if (ci >= A.Cin) continue;
#pragma unroll
                for (int kx = 0; kx < KX; ++kx) {
                    const int t = ((kr * TM + m) * TN + n) * KX + kx;
                    const int tap = TAPS == 9 ? (ky0 + kr) * 3 + kx : 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + (wm * TM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (co < A.Cout) slab[((size_t)tap * A.Cout + co) * A.Cin + ci] = acc[t][r];
                    }
                }
            }
}


// ---------------------------------------------------------------------------------------------------------------------
// The same weight gradient on split-bf16 products (the default for the tilings without a wave split of the k-steps;
// -DOSSID_WGRAD_F32 keeps every launch on the exact kernel above). Both operands are split when they are staged
// (x = hi + lo, bf16 each) and kept in LDS as [pixel][channel] bf16 images, one per part -- the natural order of the
// channels-last tensors, so a thread's float4 becomes two 8-byte writes. The matrix cores want the REDUCTION index (pixels)
// contiguous per lane: ds_read_b64_tr_b16 (cdna_hip_programming.md T10) reads a block of 4 pixels x 16 channels and hands
// lane i the 4 pixels of channel i, two of them make the 8-pixel operand of v_mfma_f32_32x32x16_bf16; the 3x3 taps'
// column shift is just a different first row of the block. Per 16 pixels and pair of 32x32 tiles three MFMAs
// (dy_lo*x_hi + dy_hi*x_lo + dy_hi*x_hi) instead of eight f32 ones. Row pitch of an image: channels * 2 bytes padded so that
// (pitch mod 256) is 64 or 192 -- the four rows of a block then sit on four different quarter-sets of the 64 banks.
// One LDS buffer (the images of one chunk), the next chunk's global loads in flight in registers under the MFMAs.
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));

__host__ __device__ constexpr int wgrad_sb_pitch(int channels) {      // bytes
    return channels == 128 ? 320 : (channels == 64 ? 192 : (channels == 32 ? 64 : channels * 2 + 64));
}

template <int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
__device__ __forceinline__ void wgrad_sb_body(const WgradArgs& A, const int L) {
    constexpr int CO_T = WM * TM * 32, CI_T = WN * TN * 32;
    constexpr int KX = TAPS == 9 ? 3 : 1;
    constexpr int NT = TM * TN * KX * KYB;
    constexpr int HALO = TAPS == 9 ? 2 : 0;
    constexpr int KTMAX = WK == 4 ? 64 : 48;                 // pixels per chunk at most (the staging registers are sized by it)
    constexpr int DY4 = CO_T / 4, X4 = CI_T / 4;
    constexpr int NLD_DY = (KTMAX * DY4 + 255) / 256, NLD_X = (KYB * (KTMAX + HALO) * X4 + 255) / 256;
    constexpr int PD = wgrad_sb_pitch(CO_T), PX = wgrad_sb_pitch(CI_T);
    static_assert(WM * WN * WK == 4 && 256 % DY4 == 0 && 256 % X4 == 0, "bad tiling");
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int KT = A.KT;                                     // a multiple of 16 here
    const int dy_img = KT * PD, x_img = KYB * (KT + HALO) * PX;          // bytes per part
    char* dyl = (char*)lds;                                  // [2 parts][KT][PD]
    char* xl = dyl + 2 * dy_img;                             // [2 parts][KYB][KT + HALO][PX]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    // WK > 1 (few-channel layers: one or two 32 x 32 tiles of (co, ci) per kernel row): the waves share the tiles and take
    // every WK-th 16-pixel k-step of a chunk, each writes its own slab (index split * WK + wk)
    const int wm = wave % WM, wn = (wave / WM) % WN, wk = wave / (WM * WN);

    const int split = L / A.ntiles;
    int tile = L - split * A.ntiles;
    int ky0 = 0;
    if (TAPS == 9 && KYB == 1) {
        ky0 = tile % 3;
        tile /= 3;
    }
    const int cib = tile % A.ci_blocks, cob = tile / A.ci_blocks;
    const int co0 = cob * CO_T, ci0 = cib * CI_T;

    v16f acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int dyq = tid % DY4, xq = tid % X4;
    const bool dyq_ok = co0 + 4 * dyq < A.Cout, xq_ok = ci0 + 4 * xq < A.Cin;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (A.pre_scale && xq_ok) {
        ps = *(const float4*)(A.pre_scale + ci0 + 4 * xq);
        pt = *(const float4*)(A.pre_shift + ci0 + 4 * xq);
    }
    float4 sdy[NLD_DY], sx[NLD_X];

    const int H = A.H, W = A.W;
    const int cpr = A.chunks_per_row;
    const int d_cr = A.nsplit % cpr, d_row = A.nsplit / cpr, d_y = d_row % H, d_b = d_row / H;
    auto advance = [&](int& b, int& y, int& cr) {
        cr += d_cr;
        int carry = cr >= cpr ? 1 : 0;
        cr -= carry * cpr;
        y += d_y + carry;
        carry = y >= H ? 1 : 0;
        y -= carry * H;
        b += d_b + carry;
    };
    int xkr[NLD_X], xp[NLD_X];
#pragma unroll
    for (int e = 0; e < NLD_X; ++e) {
        const int idx = (tid + e * 256) / X4;
        xkr[e] = idx / (KT + HALO);
        xp[e] = idx - xkr[e] * (KT + HALO);
    }
    auto stage_load = [&](int b, int y, int cr) {
        const int x0 = cr * KT;
        const float* dyr = A.dy + ((size_t)(b * H + y) * W) * A.dy_cs + co0 + 4 * dyq;
#pragma unroll
        for (int e = 0; e < NLD_DY; ++e) {
            const int p = (tid + e * 256) / DY4;
            const bool ok = p < KT && x0 + p < W && dyq_ok;
            sdy[e] = ok ? *(const float4*)(dyr + (size_t)(x0 + p) * A.dy_cs) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
#pragma unroll
        for (int e = 0; e < NLD_X; ++e) {
            const int kr = xkr[e], p = xp[e];
            const int yy = y + (TAPS == 9 ? ky0 + kr - 1 : 0), xx = x0 + p - (TAPS == 9 ? 1 : 0);
            const bool ok = kr < KYB && yy >= 0 && yy < H && xx >= 0 && xx < W && xq_ok;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (ok) {
                const int sy = (A.Hs == H) ? yy : min((int)floorf((float)yy * A.scale_h), A.Hs - 1);
                const int sxx = (A.Ws == W) ? xx : min((int)floorf((float)xx * A.scale_w), A.Ws - 1);
                v = *(const float4*)(A.x + ((size_t)(b * A.Hs + sy) * A.Ws + sxx) * A.in_cs + ci0 + 4 * xq);
                if (A.pre_scale) {
                    v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                    if (A.pre_relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
            }
            sx[e] = v;
        }
    };
    auto split_store = [&](const float4& f, char* hi_at, int part_stride) {
        const float v[4] = {f.x, f.y, f.z, f.w};
        union {
            __bf16 b[4];
            uint2 u;
        } hi, lo;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            hi.b[i] = (__bf16)v[i];
            lo.b[i] = (__bf16)(v[i] - (float)hi.b[i]);
        }
        *(uint2*)hi_at = hi.u;
        *(uint2*)(hi_at + part_stride) = lo.u;
    };
    auto stage_write = [&]() {
#pragma unroll
        for (int e = 0; e < NLD_DY; ++e) {
            const int idx = tid + e * 256;
            if (idx < KT * DY4) split_store(sdy[e], dyl + (size_t)(idx / DY4) * PD + 8 * dyq, dy_img);
        }
#pragma unroll
        for (int e = 0; e < NLD_X; ++e) {
            const int idx = tid + e * 256;
            if (idx < KYB * (KT + HALO) * X4) split_store(sx[e], xl + (size_t)(idx / X4) * PX + 8 * xq, x_img);
        }
    };

    // transposed-read addresses (T10): within its group of 16 lanes, lane 4q+p supplies the address of block row q (a pixel),
    // channels 4p..4p+3 of the block's 16; groups 0/1 take channels 0-15 / 16-31 of a 32-channel tile, the wave's halves the
    // pixels 8h..8h+7 of the k-step (two blocks of 4 pixels each)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const char* a_base = dyl + (size_t)(8 * h + tq) * PD + ((wm * TM) * 32 + 16 * tg + 4 * tp) * 2;
    const char* b_base = xl + (size_t)(8 * h + tq) * PX + ((wn * TN) * 32 + 16 * tg + 4 * tp) * 2;
    auto tr8 = [&](const char* at, int pitch) {
        const v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)at);
        const v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(at + 4 * pitch));
        typedef short v8i16 __attribute__((ext_vector_type(8)));
        const v8i16 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(v8bf16, v);
    };

    long long ch = split;
    int gb, gy, gcr;
    {
        gcr = split % cpr;
        const int row = split / cpr;
        gy = row % H, gb = row / H;
    }
    if (ch < A.n_chunks) stage_load(gb, gy, gcr);
    const int ksteps = KT / 16;
#pragma unroll 1
    for (; ch < A.n_chunks; ch += A.nsplit) {
        stage_write();
        __syncthreads();
        const long long nxt = ch + A.nsplit;
        advance(gb, gy, gcr);
        if (nxt < A.n_chunks) stage_load(gb, gy, gcr);       // in flight under this chunk's MFMAs
#pragma unroll 1
        for (int k = wk; k < ksteps; k += WK) {
            v8bf16 a[TM][2], bv[KYB][TN][KX][2];
#pragma unroll
            for (int m = 0; m < TM; ++m)
#pragma unroll
                for (int part = 0; part < 2; ++part)
                    a[m][part] = tr8(a_base + (size_t)part * dy_img + (size_t)(16 * k) * PD + m * 64, PD);
#pragma unroll
            for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
                for (int n = 0; n < TN; ++n)
#pragma unroll
                    for (int kx = 0; kx < KX; ++kx)
#pragma unroll
                        for (int part = 0; part < 2; ++part)
                            bv[kr][n][kx][part] = tr8(b_base + (size_t)part * x_img +
                                                      (size_t)(kr * (KT + HALO) + 16 * k + kx) * PX + n * 64, PX);
#pragma unroll
            for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
                for (int m = 0; m < TM; ++m)
#pragma unroll
                    for (int n = 0; n < TN; ++n)
#pragma unroll
                        for (int kx = 0; kx < KX; ++kx) {
                            const int t = ((kr * TM + m) * TN + n) * KX + kx;
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][1], bv[kr][n][kx][0], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bv[kr][n][kx][1], acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[m][0], bv[kr][n][kx][0], acc[t], 0, 0, 0);
                        }
        }
        __syncthreads();                                      // every wave has read this chunk's images
    }

    float* slab = A.slabs + ((size_t)split * WK + wk) * TAPS * A.Cout * A.Cin;
#pragma unroll
    for (int kr = 0; kr < KYB; ++kr)
#pragma unroll
        for (int m = 0; m < TM; ++m)
#pragma unroll
            for (int n = 0; n < TN; ++n) {
                const int ci = ci0 + (wn * TN + n) * 32 + c;
                if (ci >= A.Cin) continue;
#pragma unroll
                for (int kx = 0; kx < KX; ++kx) {
                    const int t = ((kr * TM + m) * TN + n) * KX + kx;
                    const int tap = TAPS == 9 ? (ky0 + kr) * 3 + kx : 0;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int co = co0 + (wm * TM + m) * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                        if (co < A.Cout) slab[((size_t)tap * A.Cout + co) * A.Cin + ci] = acc[t][r];
                    }
                }
            }
}

template <bool SB, int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
__device__ __forceinline__ void wgrad_any_body(const WgradArgs& A, const int L) {
    if constexpr (SB) wgrad_sb_body<TAPS, KYB, TM, TN, WM, WN, WK>(A, L);
    else wgrad_body<TAPS, KYB, TM, TN, WM, WN, WK>(A, L);
}

template <bool SB, int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
__global__ __launch_bounds__(256, (SB && WK == 1) ? 2 : 1) void wgrad_kernel(const WgradArgs A) {
    wgrad_any_body<SB, TAPS, KYB, TM, TN, WM, WN, WK>(A, blockIdx.x);
}

// Several independent weight gradients of ONE tiling variant in one launch (the 2 x L small problems of a dense block,
// each too small to fill the chip and too short to hide a launch): the descriptors travel in the kernel arguments.
#define OSSID_WGRAD_GROUP_MAX 24
struct WgradGroup {
    WgradArgs a[OSSID_WGRAD_GROUP_MAX];
    int first_block[OSSID_WGRAD_GROUP_MAX + 1];
    int n;
};

template <bool SB, int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
__global__ __launch_bounds__(256, (SB && WK == 1) ? 2 : 1) void wgrad_group_kernel(const WgradGroup G) {
    int i = 0;
    const int b = blockIdx.x;
    while (i + 1 < G.n && G.first_block[i + 1] <= b) ++i;
    wgrad_any_body<SB, TAPS, KYB, TM, TN, WM, WN, WK>(G.a[i], b - G.first_block[i]);
}

struct ReduceRow {
    const float* slabs;
    float* dw;
    int nslabs, taps, Cout, Cin, accumulate, first_block;
};
struct ReduceGroup {
    ReduceRow r[2 * OSSID_WGRAD_GROUP_MAX];
    int n;
};

// grouped form of wgrad_reduce2_kernel<4>: 64 elements x 4 partial sums per block
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(const ReduceGroup R) {
    __shared__ float red[4][64];
    int g = 0;
    const int b = blockIdx.x;
    while (g + 1 < R.n && R.r[g + 1].first_block <= b) ++g;
    const ReduceRow& row = R.r[g];
    const size_t n = (size_t)row.taps * row.Cout * row.Cin;
    const int e = threadIdx.x & 63, part = threadIdx.x >> 6;
    const size_t i = (size_t)(b - row.first_block) * 64 + e;
    float s = 0.0f;
    if (i < n) {
        int k = part;
        for (; k + 28 < row.nslabs; k += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = row.slabs[(size_t)(k + 4 * u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < row.nslabs; k += 4) s += row.slabs[(size_t)k * n + i];
    }
    red[part][e] = s;
    __syncthreads();
    if (part != 0 || i >= n) return;
    s = (red[0][e] + red[1][e]) + (red[2][e] + red[3][e]);
    const int ci = (int)(i % row.Cin);
    const size_t r = i / row.Cin;
    const int co = (int)(r % row.Cout), tap = (int)(r / row.Cout);
    float* o = row.dw + ((size_t)co * row.Cin + ci) * row.taps + tap;
    *o = row.accumulate ? *o + s : s;
}

// dw[co][ci][tap] (+)= sum over splits of slabs[split][tap][co][ci], in a fixed order. PARTS threads share one output
// element (each sums every PARTS-th slab with independent, unrolled loads; the partial sums meet in LDS): with hundreds
// of slabs and a small dW (layers with few channels and millions of pixels) one thread per element would be a serial
// chain of L2 latencies.
template <int PARTS>
__global__ __launch_bounds__(256) void wgrad_reduce2_kernel(const float* __restrict__ slabs, int nsplit, int taps, int Cout,
                                                            int Cin, float* __restrict__ dw, int accumulate) {
    constexpr int EPB = 256 / PARTS;                                    // elements per block
    __shared__ float red[PARTS][EPB];
    const size_t n = (size_t)taps * Cout * Cin;
    const int e = threadIdx.x % EPB, part = threadIdx.x / EPB;
    const size_t i = (size_t)blockIdx.x * EPB + e;                      // index in [tap][co][ci] order: coalesced reads
    float s = 0.0f;
    if (i < n) {
        int k = part;
        for (; k + 7 * PARTS < nsplit; k += 8 * PARTS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(k + u * PARTS) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; k < nsplit; k += PARTS) s += slabs[(size_t)k * n + i];
    }
    if (PARTS > 1) {
        red[part][e] = s;
        __syncthreads();
        if (part != 0) return;
#pragma unroll
        for (int p = 1; p < PARTS; ++p) s += red[p][e];
    }
    if (i >= n) return;
    const int ci = (int)(i % Cin);
    const size_t r = i / Cin;
    const int co = (int)(r % Cout), tap = (int)(r / Cout);
    float* o = dw + ((size_t)co * Cin + ci) * taps + tap;
    *o = accumulate ? *o + s : s;
}

#ifdef OSSID_WGRAD_F32
#define OSSID_WGRAD_SB 0
#else
#define OSSID_WGRAD_SB 1
#endif
struct WgradPlan {
    // variant: 0 <9,1,2,1,2,2,1> 128 co x 64 ci    1 <9,1,1,2,2,2,1> 64 x 128     2 <9,1,1,1,1,4,1> 32 x 128
    //          3 <1,1,2,2,2,2,1> 128 x 128 (1x1)   4 <1,1,1,2,2,2,1> 64 x 128 (1x1)
    //          5 <9,3,1,1,1,1,4> 32 x 32, all nine taps, k-steps shared out over the waves   6 <9,3,1,1,1,2,2> 32 x 64
    int variant, co_t, ci_t, kyb, wk, tiles_per_wave;
    int KT, chunks_per_row, nsplit, co_blocks, ci_blocks, ntiles, B, H, W;
    int sb;                 // 1: the split-bf16 kernel (wgrad_sb_body: KT a multiple of 16, one LDS buffer of bf16 images)
    long long n_chunks;
    size_t lds;
};

bool wgrad_plan(int B, int H, int W, int Cin, int Cout, int taps, WgradPlan& p) {
    if ((long long)B * H * W > 0x3fffffffLL) return false;
    p.kyb = 1, p.wk = 1;
    int ktmax = 48;
    if (taps == 9) {
        if (Cout <= 32 && Cin <= 32) p.variant = 5, p.co_t = 32, p.ci_t = 32, p.kyb = 3, p.wk = 4, p.tiles_per_wave = 9;
        else if (Cout <= 32 && Cin <= 64) p.variant = 6, p.co_t = 32, p.ci_t = 64, p.kyb = 3, p.wk = 2, p.tiles_per_wave = 9, ktmax = 40;
        else if (Cout <= 32) p.variant = 2, p.co_t = 32, p.ci_t = 128, p.tiles_per_wave = 3;
        else if (Cout <= 64 || Cout == 96) p.variant = 1, p.co_t = 64, p.ci_t = 128, p.tiles_per_wave = 6;
        else p.variant = 0, p.co_t = 128, p.ci_t = 64, p.tiles_per_wave = 6;
        p.B = B, p.H = H, p.W = W;
        p.sb = OSSID_WGRAD_SB;
        // split form: k-steps of 16 pixels, WK of them per chunk for the tilings whose waves share the k-steps
        if (p.sb && p.variant == 5) ktmax = 64;
        if (p.sb && p.variant == 6) ktmax = 32;
        p.chunks_per_row = (W + ktmax - 1) / ktmax;
        p.KT = p.sb ? ((W + p.chunks_per_row - 1) / p.chunks_per_row + 15) & ~15
                    : ((W + p.chunks_per_row - 1) / p.chunks_per_row + 1) & ~1;
        if (p.sb && p.KT < 16 * p.wk) p.KT = 16 * p.wk;
    } else if (taps == 1) {
        if (Cout <= 64) p.variant = 4, p.co_t = 64, p.ci_t = 128, p.tiles_per_wave = 2;
        else p.variant = 3, p.co_t = 128, p.ci_t = 128, p.tiles_per_wave = 4;
        p.B = 1, p.H = 1, p.W = B * H * W;          // no halo: the whole tensor is one long pixel row
        p.sb = OSSID_WGRAD_SB;
        p.KT = 32;
        p.chunks_per_row = (p.W + p.KT - 1) / p.KT;
    } else {
        return false;
    }
    p.n_chunks = (long long)p.B * p.H * p.chunks_per_row;
    p.co_blocks = (Cout + p.co_t - 1) / p.co_t;
    p.ci_blocks = (Cin + p.ci_t - 1) / p.ci_t;
    p.ntiles = p.co_blocks * p.ci_blocks * ((taps == 9 && p.kyb == 1) ? 3 : 1);
    const int halo = taps == 9 ? 2 : 0;
    p.lds = p.sb ? (size_t)2 * (p.KT * wgrad_sb_pitch(p.co_t) + p.kyb * (p.KT + halo) * wgrad_sb_pitch(p.ci_t))
                 : (size_t)2 * (p.KT * p.co_t + p.kyb * (p.KT + halo) * p.ci_t) * sizeof(float);
    // How many K-splits (multiples of 8: one per XCD)? More splits = more workgroups in flight but every split costs a
    // slab (written here, read by the reduction). Model, in microseconds: rounds of resident workgroups x (chunks per
    // workgroup x MFMA time of a chunk + a fixed ~4 us to fill the pipeline and store the tiles) + slab traffic at ~3 TB/s.
    const int per_cu = p.lds > 80 * 1024 ? 1 : (p.lds > 53 * 1024 ? 2 : 3);
    const double t_chunk = p.sb ? (double)(p.KT / 16) * p.tiles_per_wave / p.wk * 96.0 / 2400.0 + 0.5
                                : (double)(p.KT / 2) * p.tiles_per_wave / p.wk * 64.0 / 2400.0 + 0.35;
    const double dw_bytes = (double)taps * Cout * Cin * 4.0;
    double best = 1e30;
    int best_s = 1;
    for (int sp = 1; sp <= 1024; ++sp) {
        if (sp > 1 && sp > p.n_chunks) break;
        const double blocks = (double)p.ntiles * sp;
        const double rounds = ceil(blocks / (256.0 * per_cu));
        const double chunks_pb = ceil((double)p.n_chunks / sp);
        const int slabs = sp * p.wk;
        const double t_red = (slabs <= 16 ? slabs : slabs <= 128 ? slabs / 4.0 : slabs / 32.0) * 0.12;   // dependent L2 round trips
        const double t = rounds * (chunks_pb * t_chunk + 4.0) + slabs * dw_bytes * 2.0 / 3.0e6 + t_red;
        if (t < best) best = t, best_s = sp;
    }
    p.nsplit = best_s;
    return true;
}

template <bool SB, int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
int launch_wgrad_group_form(const WgradGroup& g, size_t lds, hipStream_t s) {
    auto kern = wgrad_group_kernel<SB, TAPS, KYB, TM, TN, WM, WN, WK>;
    OSSID_ENSURE_LDS(kern, lds);
    hipLaunchKernelGGL(kern, dim3((unsigned)g.first_block[g.n]), dim3(256), lds, s, g);
    return ossid_launch_status();
}
template <int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
int launch_wgrad_group(const WgradGroup& g, size_t lds, bool sb, hipStream_t s) {
    if (sb) return launch_wgrad_group_form<true, TAPS, KYB, TM, TN, WM, WN, WK>(g, lds, s);
    return launch_wgrad_group_form<false, TAPS, KYB, TM, TN, WM, WN, WK>(g, lds, s);
}

template <bool SB, int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
int launch_wgrad_form(const WgradArgs& a, const WgradPlan& p, hipStream_t s) {
    auto kern = wgrad_kernel<SB, TAPS, KYB, TM, TN, WM, WN, WK>;
    OSSID_ENSURE_LDS(kern, p.lds);
    const long nwg = (long)p.nsplit * p.ntiles;
    hipLaunchKernelGGL(kern, dim3((unsigned)nwg), dim3(256), p.lds, s, a);
    return ossid_launch_status();
}
template <int TAPS, int KYB, int TM, int TN, int WM, int WN, int WK>
int launch_wgrad(const WgradArgs& a, const WgradPlan& p, hipStream_t s) {
    if (p.sb) return launch_wgrad_form<true, TAPS, KYB, TM, TN, WM, WN, WK>(a, p, s);
    return launch_wgrad_form<false, TAPS, KYB, TM, TN, WM, WN, WK>(a, p, s);
}

template <int QX>
int launch_chan_op(const ChanOpArgs& a, int P, hipStream_t s) {
    const int C4 = (a.C + 3) / 4;
    hipLaunchKernelGGL(chan_op_kernel<QX>, dim3((C4 + QX - 1) / QX, P), dim3(256), 0, s, a);
    return ossid_launch_status();
}

// Segmentation head of the loss (models/dtoid/__init__.py:210-232): p = sigmoid(logit), BCELoss(p, mask) with torch's
// clamps (log terms >= -100; backward through input * (1 - input) clamped at 1e-12), and the foreground IoU of (p > 0.5)
// against (mask > 0) per image -- one pass instead of ~20 elementwise / reduction launches. The un-normalised gradient
// d(sum BCE)/d(logit) is stored in the forward pass; backward is a scale by (upstream gradient / N).
__global__ __launch_bounds__(256) void seg_bce_iou_kernel(const float* __restrict__ logit, const float* __restrict__ mask, long long hw,
                                                          int blocks_per_image, float* __restrict__ prob,
                                                          float* __restrict__ dlogit, double* __restrict__ partials) {
    const int b = blockIdx.y;
    const long long per = (hw + blocks_per_image - 1) / blocks_per_image;
    const long long i0 = (long long)blockIdx.x * per, i1 = i0 + per < hw ? i0 + per : hw;
    const float* x = logit + (size_t)b * hw;
    const float* y = mask + (size_t)b * hw;
    double loss = 0.0;
    float inter = 0.f, uni = 0.f;
    for (long long i = i0 + threadIdx.x; i < i1; i += 256) {
        const float xv = x[i], yv = y[i];
        const float p = 1.0f / (1.0f + expf(-xv));
        const float lp = fmaxf(logf(p), -100.0f), l1p = fmaxf(log1pf(-p), -100.0f);
        loss += (double)((yv - 1.0f) * l1p - yv * lp);
        const float pq = p * (1.0f - p);
        prob[(size_t)b * hw + i] = p;
        dlogit[(size_t)b * hw + i] = (p - yv) / fmaxf(pq, 1e-12f) * pq;
        const bool pr = p > 0.5f, gt = yv > 0.0f;
        inter += (pr && gt) ? 1.f : 0.f;
        uni += (pr || gt) ? 1.f : 0.f;
    }
    __shared__ double red[3][256];
    red[0][threadIdx.x] = loss, red[1][threadIdx.x] = inter, red[2][threadIdx.x] = uni;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st)
            for (int k = 0; k < 3; ++k) red[k][threadIdx.x] += red[k][threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double* o = partials + ((size_t)b * blocks_per_image + blockIdx.x) * 3;
        o[0] = red[0][0], o[1] = red[1][0], o[2] = red[2][0];
    }
}

// out[0] = mean BCE over all B * hw elements, out[1 + b] = IoU of image b (0 when the union is empty)
// one wave: lane l takes blocks l, l + 64, ... of an image, then a fixed-order butterfly in double (deterministic; a single
// thread walking all B x blocks_per_image partials took 58 us on the step's critical chain)
__global__ __launch_bounds__(64) void seg_bce_iou_finalize_kernel(const double* __restrict__ partials, int B, int blocks_per_image,
                                                                  long long hw, float* __restrict__ out) {
    if (blockIdx.x != 0) return;
    const int lane = threadIdx.x;
    double total = 0.0;
    for (int b = 0; b < B; ++b) {
        double l = 0.0, in = 0.0, un = 0.0;
        for (int k = lane; k < blocks_per_image; k += 64) {
            const double* o = partials + ((size_t)b * blocks_per_image + k) * 3;
            l += o[0], in += o[1], un += o[2];
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) l += __shfl_xor(l, m), in += __shfl_xor(in, m), un += __shfl_xor(un, m);
        total += l;
        if (lane == 0) out[1 + b] = un > 0.0 ? (float)(in / un) : 0.0f;
    }
    if (lane == 0) out[0] = (float)(total / ((double)B * (double)hw));
}

// =====================================================================================================================
// 3x3 / pad 1 convolution with ONE output channel (the decoder's seg_final 16 -> 1 at 480 x 640, network.py:362) in training:
// a per-pixel dot product of 9 x C inputs -- vector-ALU work bounded by the 157 MB of input, not a matrix-core shape (a 32-row
// MFMA tile would be 31/32 padding). x [B][H][W][C] channels-last, w [C][3][3] (= nn.Conv2d(C, 1, 3).weight), out / g [B][H][W].
__global__ __launch_bounds__(256) void conv3x3_c1_fwd_kernel(const float4* __restrict__ x, int H, int W, int C4, const float* __restrict__ w,
                                                             const float* __restrict__ bias, size_t total, float* __restrict__ out) {
    extern __shared__ float wl[];                  // [9][C]: tap-major, so that a tap's weights are float4-contiguous over channels
    const int C = C4 * 4;
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[i] = w[(i % C) * 9 + i / C];
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const size_t b = p / ((size_t)W * H);
    float acc = bias ? bias[0] : 0.0f;
    if (C4 == 4) {
        // the decoder's 16 -> 1 layer: all 36 loads of the pixel issued before the first use, from clamped addresses (a load
        // under a condition is a wait in front of the next one: the branchy form below ran at 0.75 TB/s of its 157 MB input);
        // a tap outside the image counts with factor 0, in the same tap order
        float4 v[9][4];
        float f[9];
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int y = yy + ky - 1, xc = xx + kx - 1;
                f[ky * 3 + kx] = (y >= 0 && y < H && xc >= 0 && xc < W) ? 1.0f : 0.0f;
                const float4* px = x + ((b * H + min(max(y, 0), H - 1)) * W + min(max(xc, 0), W - 1)) * 4;
#pragma unroll
                for (int c = 0; c < 4; ++c) v[ky * 3 + kx][c] = px[c];
            }
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            if (f[t] == 0.0f) continue;
            const float4* wt = (const float4*)(wl + t * C);
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 q = wt[c];
                acc += v[t][c].x * q.x + v[t][c].y * q.y + v[t][c].z * q.z + v[t][c].w * q.w;
            }
        }
        out[p] = acc;
        return;
    }
#pragma unroll
    for (int ky = 0; ky < 3; ++ky) {
        const int y = yy + ky - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int xc = xx + kx - 1;
            if (xc < 0 || xc >= W) continue;
            const float4* px = x + ((b * H + y) * W + xc) * C4;
            const float4* wt = (const float4*)(wl + (ky * 3 + kx) * C);
            for (int c = 0; c < C4; ++c) {
                const float4 v = px[c], q = wt[c];
                acc += v.x * q.x + v.y * q.y + v.z * q.z + v.w * q.w;
            }
        }
    }
    out[p] = acc;
}

// dx[b][y][x][c] = sum_taps g[b][y - ky + 1][x - kx + 1] * w[c][ky][kx]
__global__ __launch_bounds__(256) void conv3x3_c1_dgrad_kernel(const float* __restrict__ g, int H, int W, int C4, const float* __restrict__ w,
                                                               size_t total, float4* __restrict__ dx) {
    extern __shared__ float wl[];                  // [9][C]
    const int C = C4 * 4;
    for (int i = threadIdx.x; i < 9 * C; i += 256) wl[i] = w[(i % C) * 9 + i / C];
    __syncthreads();
    const size_t p = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (p >= total) return;
    const int xx = (int)(p % W), yy = (int)((p / W) % H);
    const size_t b = p / ((size_t)W * H);
    float gv[9];
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            const int y = yy - ky + 1, xc = xx - kx + 1;
            gv[ky * 3 + kx] = (y >= 0 && y < H && xc >= 0 && xc < W) ? g[(b * H + y) * W + xc] : 0.0f;
        }
    for (int c = 0; c < C4; ++c) {
        float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            const float4 q = *(const float4*)(wl + t * C + 4 * c);
            r.x += gv[t] * q.x, r.y += gv[t] * q.y, r.z += gv[t] * q.z, r.w += gv[t] * q.w;
        }
        dx[p * C4 + c] = r;
    }
}

// dw[c][tap] = sum_pixels g[pixel] * x[pixel + tap][c], db = sum g: every thread walks a strided set of pixels with 9 x 16
// accumulators (C <= 16 per pass), waves reduce by shuffles and store one partial row each; a second launch adds the rows in a
// fixed order (bit-reproducible).
__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_kernel(const float4* __restrict__ x, const float* __restrict__ g, int H, int W,
                                                               int C4, int c4_0, size_t total, float* __restrict__ partials, int row_len) {
    float acc[9][16];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[t][c] = 0.0f;
    float gs = 0.0f;
    const int nc4 = C4 - c4_0 < 4 ? C4 - c4_0 : 4;
    for (size_t p = (size_t)blockIdx.x * 256 + threadIdx.x; p < total; p += (size_t)gridDim.x * 256) {
        const float gp = g[p];
        gs += gp;
        const int xx = (int)(p % W), yy = (int)((p / W) % H);
        const size_t b = p / ((size_t)W * H);
        if (nc4 == 4) {
            // (all 36 loads of the pixel before the first use, clamped addresses, a tap outside the image with gradient 0: the
            // accumulators take the same terms in the same order as the branchy form below, plus exact zeros)
            float4 v[9][4];
            float gt[9];
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const int y = yy + ky - 1, xc = xx + kx - 1;
                    gt[ky * 3 + kx] = (y >= 0 && y < H && xc >= 0 && xc < W) ? gp : 0.0f;
                    const float4* px = x + ((b * H + min(max(y, 0), H - 1)) * W + min(max(xc, 0), W - 1)) * C4 + c4_0;
#pragma unroll
                    for (int c = 0; c < 4; ++c) v[ky * 3 + kx][c] = px[c];
                }
#pragma unroll
            for (int t = 0; t < 9; ++t)
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    acc[t][4 * c + 0] += gt[t] * v[t][c].x, acc[t][4 * c + 1] += gt[t] * v[t][c].y;
                    acc[t][4 * c + 2] += gt[t] * v[t][c].z, acc[t][4 * c + 3] += gt[t] * v[t][c].w;
                }
            continue;
        }
#pragma unroll
        for (int ky = 0; ky < 3; ++ky) {
            const int y = yy + ky - 1;
            if (y < 0 || y >= H) continue;
#pragma unroll
            for (int kx = 0; kx < 3; ++kx) {
                const int xc = xx + kx - 1;
                if (xc < 0 || xc >= W) continue;
                const float4* px = x + ((b * H + y) * W + xc) * C4 + c4_0;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    if (c >= nc4) break;
                    const float4 v = px[c];
                    acc[ky * 3 + kx][4 * c + 0] += gp * v.x, acc[ky * 3 + kx][4 * c + 1] += gp * v.y;
                    acc[ky * 3 + kx][4 * c + 2] += gp * v.z, acc[ky * 3 + kx][4 * c + 3] += gp * v.w;
                }
            }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            float v = acc[t][c];
            for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
            acc[t][c] = v;
        }
    for (int m = 32; m >= 1; m >>= 1) gs += __shfl_xor(gs, m);
    if (lane == 0) {
        float* o = partials + ((size_t)blockIdx.x * 4 + wave) * row_len;
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int c = 0; c < 16; ++c) o[c * 9 + t] = acc[t][c];           // [c][tap], as the parameter
        o[144] = gs;
    }
}

__global__ __launch_bounds__(256) void conv3x3_c1_wgrad_finalize_kernel(const float* __restrict__ partials, int rows, int row_len, int n_valid,
                                                                        float* __restrict__ dw, float* __restrict__ db, int write_db) {
    __shared__ double red[256];
    const int j = blockIdx.x;                        // one output value per block: 0..143 = dw (this 16-channel pass), 144 = db
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += 256) s += (double)partials[(size_t)r * row_len + j];
    red[threadIdx.x] = s;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if ((int)threadIdx.x < st) red[threadIdx.x] += red[threadIdx.x + st];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        if (j < 144) {
            if (j < n_valid) dw[j] = (float)red[0];
        } else if (write_db && db) {
            db[0] = (float)red[0];
        }
    }
}

// [cout][cin][k][k] convolution weights <-> the [cout][kpad] matrix whose columns follow ossid_im2col_stem's order
// ((ky * k + kx) * cin + c, zero-padded to kpad): the strided stems run as im2col + a 1x1 MFMA convolution.
__global__ void stem_weight_relayout_kernel(const float* __restrict__ src, float* __restrict__ dst, int cout, int cin, int k,
                                            int kpad, int inverse) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int kk = k * k;
    if (!inverse) {
        if (i >= cout * kpad) return;
        const int o = i / kpad, col = i % kpad;
        float v = 0.f;
        if (col < kk * cin) {
            const int tap = col / cin, c = col % cin;
            v = src[((size_t)o * cin + c) * kk + tap];
        }
        dst[i] = v;
    } else {
        if (i >= cout * cin * kk) return;
        const int o = i / (cin * kk), r = i % (cin * kk), c = r / kk, tap = r % kk;
        dst[i] = src[(size_t)o * kpad + tap * cin + c];
    }
}

}  // namespace

extern "C" {

int ossid_chan_op_partials(long long n_rows, int C) {
    const int C4 = (C + 3) / 4;
    const int QX = C4 <= 8 ? 8 : C4 <= 16 ? 16 : C4 <= 32 ? 32 : 64;
    const int gx = (C4 + QX - 1) / QX, RY = 256 / QX;
    long P = (1024 + gx - 1) / gx;
    if (P > 512) P = 512;          // (partials are combined by 8-16 threads per channel in the fold / finalize kernels)
    const long maxP = (long)((n_rows + 4 * RY - 1) / (4 * RY));          // at least 4 rows per thread
    if (P > maxP) P = maxP;
    if (P < 1) P = 1;
    return (int)P;
}

size_t ossid_seg_bce_iou_workspace_bytes(int B) { return (size_t)(B > 0 ? B : 1) * 64 * 3 * sizeof(double); }

int ossid_seg_bce_iou_fwd(const float* logit, const float* mask, int B, long long hw, float* prob, float* dlogit_sum,
                          float* out, void* workspace, size_t workspace_bytes, void* stream) {
    if (!logit || !mask || !prob || !dlogit_sum || !out || !workspace || B <= 0 || B > 65535 || hw <= 0) return OSSID_EINVAL;
    if (workspace_bytes < ossid_seg_bce_iou_workspace_bytes(B) || ((uintptr_t)workspace & 7)) return OSSID_EINVAL;
    const int bpi = 64;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(seg_bce_iou_kernel, dim3(bpi, B), dim3(256), 0, s, logit, mask, hw, bpi, prob, dlogit_sum, (double*)workspace);
    hipLaunchKernelGGL(seg_bce_iou_finalize_kernel, dim3(1), dim3(64), 0, s, (const double*)workspace, B, bpi, hw, out);
    return ossid_launch_status();
}

int ossid_conv3x3_c1_fwd(const float* x, int B, int H, int W, int C, const float* w, const float* bias, float* out, void* stream) {
    if (!x || !w || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || C > 1024 || ((uintptr_t)x & 15)) return OSSID_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(conv3x3_c1_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)9 * C * 4, (hipStream_t)stream,
                       (const float4*)x, H, W, C / 4, w, bias, total, out);
    return ossid_launch_status();
}

int ossid_conv3x3_c1_dgrad(const float* g, int B, int H, int W, int C, const float* w, float* dx, void* stream) {
    if (!g || !w || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || C > 1024 || ((uintptr_t)dx & 15)) return OSSID_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipLaunchKernelGGL(conv3x3_c1_dgrad_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), (size_t)9 * C * 4, (hipStream_t)stream,
                       g, H, W, C / 4, w, total, (float4*)dx);
    return ossid_launch_status();
}

static const int C1_WGRAD_BLOCKS = 1024, C1_WGRAD_ROW = 160;       // 4 waves x 1024 blocks partial rows of 145 (padded) floats
size_t ossid_conv3x3_c1_wgrad_workspace_bytes(void) { return (size_t)C1_WGRAD_BLOCKS * 4 * C1_WGRAD_ROW * sizeof(float); }

int ossid_conv3x3_c1_wgrad(const float* x, const float* g, int B, int H, int W, int C, void* workspace, size_t workspace_bytes,
                           float* dw, float* db, void* stream) {
    if (!x || !g || !dw || !workspace || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || C > 1024 || ((uintptr_t)x & 15))
        return OSSID_EINVAL;
    if (workspace_bytes < ossid_conv3x3_c1_wgrad_workspace_bytes()) return OSSID_EINVAL;
    const size_t total = (size_t)B * H * W;
    hipStream_t s = (hipStream_t)stream;
    for (int c4_0 = 0; c4_0 < C / 4; c4_0 += 4) {           // 16 channels per pass (the decoder's layer has 16)
        hipLaunchKernelGGL(conv3x3_c1_wgrad_kernel, dim3(C1_WGRAD_BLOCKS), dim3(256), 0, s, (const float4*)x, g, H, W, C / 4, c4_0, total,
                           (float*)workspace, C1_WGRAD_ROW);
        const int n_valid = (C / 4 - c4_0 < 4 ? C / 4 - c4_0 : 4) * 4 * 9;
        hipLaunchKernelGGL(conv3x3_c1_wgrad_finalize_kernel, dim3(145), dim3(256), 0, s, (const float*)workspace, C1_WGRAD_BLOCKS * 4,
                           C1_WGRAD_ROW, n_valid, dw + (size_t)c4_0 * 4 * 9, db, c4_0 == 0 ? 1 : 0);
    }
    return ossid_launch_status();
}

int ossid_stem_weight_relayout(const float* src, float* dst, int cout, int cin, int k, int kpad, int inverse, void* stream) {
    if (!src || !dst || cout <= 0 || cin <= 0 || k <= 0 || kpad < k * k * cin) return OSSID_EINVAL;
    const int n = inverse ? cout * cin * k * k : cout * kpad;
    hipLaunchKernelGGL(stem_weight_relayout_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, src, dst, cout, cin, k,
                       kpad, inverse ? 1 : 0);
    return ossid_launch_status();
}

int ossid_fill_zero(void* ptr, size_t bytes, void* stream) {
    if (!ptr && bytes) return OSSID_EINVAL;
    if (bytes == 0) return OSSID_OK;
    return hipMemsetAsync(ptr, 0, bytes, (hipStream_t)stream) == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
}

int ossid_chan_op(const ossid_chan_op_desc* d, void* stream) {
    if (!d) return OSSID_EINVAL;
    if (d->n_rows < 0 || d->channels <= 0 || d->channels % 4) return OSSID_EINVAL;
    if (d->n_rows == 0) return OSSID_OK;
    if (!d->g || (d->mask_mode != 0 && !d->x) || d->mask_mode < 0 || d->mask_mode > 3 || d->sum_mode < 0 || d->sum_mode > 3)
        return OSSID_EINVAL;
    if (d->sum_mode != 0 && (!d->partials || (!d->sums && !d->defer_finalize))) return OSSID_EINVAL;
    if (!d->out && d->sum_mode == 0) return OSSID_EINVAL;
    ChanOpArgs a;
    a.g = d->g, a.x = d->x, a.out = d->out, a.alpha = d->alpha, a.beta = d->beta, a.kappa = d->kappa;
    a.ms = d->mask_scale, a.mt = d->mask_shift, a.n_rows = d->n_rows, a.C = d->channels;
    a.pivot = d->pivot;
    if (d->sum_mode == 3 && !d->pivot) return OSSID_EINVAL;
    a.g_cs = d->g_stride > 0 ? d->g_stride : d->channels;
    a.x_cs = d->x_stride > 0 ? d->x_stride : d->channels;
    a.out_cs = d->out_stride > 0 ? d->out_stride : d->channels;
    if ((a.g_cs % 4) || (a.x_cs % 4) || (a.out_cs % 4)) return OSSID_EINVAL;
    a.mask_mode = d->mask_mode, a.accumulate = d->accumulate, a.sum_mode = d->sum_mode;
    a.partials = d->partials;
    const int P = ossid_chan_op_partials(d->n_rows, d->channels);
    a.rows_per_block = (int)((d->n_rows + P - 1) / P);
    const int C4 = d->channels / 4;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    if (C4 <= 8) rc = launch_chan_op<8>(a, P, s);
    else if (C4 <= 16) rc = launch_chan_op<16>(a, P, s);
    else if (C4 <= 32) rc = launch_chan_op<32>(a, P, s);
    else rc = launch_chan_op<64>(a, P, s);
    if (rc != OSSID_OK || d->sum_mode == 0 || d->defer_finalize) return rc;
    return launch_colsum_finalize((const float*)d->partials, P, d->channels, d->sums,
                                  d->sums_row_stride > 0 ? d->sums_row_stride : d->channels, d->sum_mode == 3 ? d->pivot : nullptr, s);
}

int ossid_bn_fold_fwd(const float* sums, int sums_row_stride, const float* partials, int n_partials, const float* pivot, int C,
                      double n, const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                      float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, void* stream) {
    if ((!sums && n_partials <= 0) || (n_partials > 0 && !partials) || C <= 0 || n <= 0 || !scale || !shift || !mean_out ||
        !rstd_out || (!running_mean != !running_var))
        return OSSID_EINVAL;
    if (n_partials > 128)
        hipLaunchKernelGGL(bn_fold_fwd_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, sums,
                           sums_row_stride > 0 ? sums_row_stride : C, partials, n_partials, pivot, C, n, gamma, beta, eps,
                           momentum, running_mean, running_var, scale, shift, mean_out, rstd_out);
    else if (n_partials > 16)
        hipLaunchKernelGGL(bn_fold_fwd_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, sums,
                           sums_row_stride > 0 ? sums_row_stride : C, partials, n_partials, pivot, C, n, gamma, beta, eps,
                           momentum, running_mean, running_var, scale, shift, mean_out, rstd_out);
    else
        hipLaunchKernelGGL(bn_fold_fwd_kernel<4>, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, sums,
                           sums_row_stride > 0 ? sums_row_stride : C, partials, n_partials, pivot, C, n, gamma, beta, eps,
                           momentum, running_mean, running_var, scale, shift, mean_out, rstd_out);
    return ossid_launch_status();
}

int ossid_bn_fold_fwd_tail(float* table, int row_stride, int tail_c0, const float* tail_partials, int n_partials,
                           const float* tail_pivot, int C, double n, const float* gamma, const float* beta, float eps, float momentum,
                           float* running_mean, float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out,
                           void* stream) {
    if (!table || !tail_partials || !tail_pivot || n_partials <= 0 || C <= 0 || tail_c0 <= 0 || tail_c0 >= C || (tail_c0 % 64) % 32 ||
        (tail_c0 % 32) || row_stride < C || n <= 0 || !scale || !shift || !mean_out || !rstd_out || (!running_mean != !running_var))
        return OSSID_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n_partials > 128)
        hipLaunchKernelGGL(bn_fold_fwd_tail_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, s, table, row_stride, tail_c0, tail_partials,
                           n_partials, tail_pivot, C, n, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean_out,
                           rstd_out);
    else
        hipLaunchKernelGGL(bn_fold_fwd_tail_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, s, table, row_stride, tail_c0, tail_partials,
                           n_partials, tail_pivot, C, n, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean_out,
                           rstd_out);
    return ossid_launch_status();
}

int ossid_bn_fold_bwd(const float* dscale, const float* dshift, const float* partials, int n_partials, const float* gamma,
                      const float* mean, const float* rstd, int C, double n, float* dgamma, float* dbeta, float* coef_x,
                      float* coef_1, int accumulate, float* zero_row, void* stream) {
    if (((!dscale || !dshift) && n_partials <= 0) || (n_partials > 0 && !partials) || !mean || !rstd || C <= 0 || n <= 0 ||
        !coef_x || !coef_1)
        return OSSID_EINVAL;
    if (n_partials > 128)
        hipLaunchKernelGGL(bn_fold_bwd_kernel<64>, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                           partials, n_partials, gamma, mean, rstd, C, n, dgamma, dbeta, coef_x, coef_1, accumulate, zero_row);
    else if (n_partials > 16)
        hipLaunchKernelGGL(bn_fold_bwd_kernel<16>, dim3((C + 15) / 16), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                           partials, n_partials, gamma, mean, rstd, C, n, dgamma, dbeta, coef_x, coef_1, accumulate, zero_row);
    else
        hipLaunchKernelGGL(bn_fold_bwd_kernel<4>, dim3((C + 63) / 64), dim3(256), 0, (hipStream_t)stream, dscale, dshift,
                           partials, n_partials, gamma, mean, rstd, C, n, dgamma, dbeta, coef_x, coef_1, accumulate, zero_row);
    return ossid_launch_status();

}

int ossid_colsum_finalize(const float* partials, int n_partials, int C, float* sums, int sums_row_stride, void* stream) {
    if (!partials || n_partials <= 0 || C <= 0 || !sums) return OSSID_EINVAL;
    return launch_colsum_finalize(partials, n_partials, C, sums, sums_row_stride > 0 ? sums_row_stride : C, nullptr,
                                  (hipStream_t)stream);
}

int ossid_conv_pack_weights_table(const ossid_pack_row* rows_device, int n_rows, long long total_blocks, void* stream) {
    if (!rows_device || n_rows <= 0 || total_blocks <= 0 || total_blocks > 0x7fffffffLL) return OSSID_EINVAL;
    static_assert(sizeof(PackRow) == sizeof(ossid_pack_row), "pack row layout");
    hipLaunchKernelGGL(pack_all_kernel, dim3((unsigned)total_blocks), dim3(256), 0, (hipStream_t)stream,
                       (const PackRow*)rows_device, n_rows);
    return ossid_launch_status();
}

int ossid_avgpool2_nhwc(const float* x, int B, int H, int W, int C, int stride, float* out, int backward, void* stream) {
    if (!x || !out || B <= 0 || H < 2 || W < 2 || C <= 0 || C % 4 || (stride != 1 && stride != 2)) return OSSID_EINVAL;
    const int Ho = (H - 2) / stride + 1, Wo = (W - 2) / stride + 1;
    hipStream_t s = (hipStream_t)stream;
    if (!backward) {
        const size_t total = (size_t)B * Ho * Wo * (C / 4);
        hipLaunchKernelGGL(avgpool2_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)x, H, W,
                           C / 4, stride, Ho, Wo, total, (float4*)out);
    } else {      // x = d out [B][Ho][Wo][C], out = d in [B][H][W][C]
        const size_t total = (size_t)B * H * W * (C / 4);
        hipLaunchKernelGGL(avgpool2_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, (const float4*)x, H, W,
                           C / 4, stride, Ho, Wo, total, (float4*)out);
    }
    return ossid_launch_status();
}

int ossid_upsample_nearest_bwd_nhwc(const float* dup, int B, int Hs, int Ws, int H, int W, int C, const int32_t* row_start,
                                    const int32_t* col_start, float* dsrc, void* stream) {
    if (!dup || !dsrc || !row_start || !col_start || B <= 0 || Hs <= 0 || Ws <= 0 || H < Hs || W < Ws || C <= 0 || C % 4)
        return OSSID_EINVAL;
    const size_t total = (size_t)B * Hs * Ws * (C / 4);
    hipLaunchKernelGGL(upsample_bwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dup, Hs, Ws, H, W, C / 4, row_start, col_start, total, (float4*)dsrc);
    return ossid_launch_status();
}

int ossid_conv_pack_weights_dgrad(const float* w, int Cout, int Cin, int taps, float* wpk, void* stream) {
    if (taps != 1 && taps != 9) return OSSID_EINVAL;
    return ossid_conv_pack_weights_form(w, Cout, Cin, taps, 1, 0, wpk, stream);
}

#ifndef OSSID_WGRAD_FEWCH
#define OSSID_WGRAD_FEWCH 1      // the decoder's few-channel 3x3 layers on csrc/wgrad_fc.hip (0: the general kernel, for A/B runs)
#endif
#ifndef OSSID_WGRAD_T9
#define OSSID_WGRAD_T9 1         // the dense blocks' 3x3 layers (128 -> 32) on csrc/wgrad_t9.hip (0: this file's grouped kernel)
#endif
#ifndef OSSID_WGRAD_T9_SINGLE
#define OSSID_WGRAD_T9_SINGLE 1  // ... and every other plain 3x3 layer with input channels in 128s (the head) through ossid_conv_wgrad
#endif

size_t ossid_conv_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout, int taps) {
    WgradPlan p;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || !wgrad_plan(B, H, W, Cin, Cout, taps, p)) return 0;
    size_t n = (size_t)p.nsplit * p.wk * taps * Cout * Cin * sizeof(float);
    if (OSSID_WGRAD_FEWCH && ossid_wgrad_fewch_takes(Cin, Cout, taps, Cin, Cout)) {
        const size_t m = ossid_wgrad_fewch_workspace_bytes(B, H, W, Cin, Cout);
        if (m > n) n = m;
    }
    if (OSSID_WGRAD_T9 && OSSID_WGRAD_T9_SINGLE) {          // (a shape-only probe: pointers and strides are checked at the call)
        ossid_wgrad_desc d = {};
        d.batch = B, d.height = H, d.width = W, d.cin = Cin, d.cout = Cout, d.taps = taps;
        if (Cin <= 256 && ossid_wgrad_t9_takes(&d)) {
            const size_t m = ossid_wgrad_t9_workspace_bytes(&d, 1);
            if (m > n) n = m;
        }
    }
    return n;
}

int ossid_conv_wgrad_split_bf16(void) { return OSSID_WGRAD_SB; }

int ossid_conv_wgrad(const ossid_wgrad_desc* d, void* stream) {
    if (!d) return OSSID_EINVAL;
    const int B = d->batch, H = d->height, W = d->width, Cin = d->cin, Cout = d->cout, taps = d->taps;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || (Cin % 4) || (Cout % 4)) return OSSID_EINVAL;
    if (!d->x || !d->dy || !d->dw || !d->workspace || (d->pre_scale && !d->pre_shift)) return OSSID_EINVAL;
    if (d->dy_add) return OSSID_EINVAL;                        // (only csrc/wgrad_t9.hip's 1x1 jobs, reached through the group entry)
    if (OSSID_WGRAD_FEWCH && ossid_wgrad_fewch_takes(Cin, Cout, taps, d->in_channel_stride > 0 ? d->in_channel_stride : Cin,
                                                     d->dy_channel_stride > 0 ? d->dy_channel_stride : Cout) &&
        d->workspace_bytes >= ossid_wgrad_fewch_workspace_bytes(B, H, W, Cin, Cout) && !((uintptr_t)d->x & 15) &&
        !((uintptr_t)d->dy & 15) && !((uintptr_t)d->workspace & 15))
        return ossid_wgrad_fewch(d, stream);                 // 2-D pixel tiles, every tap from one staged patch (csrc/wgrad_fc.hip)
    // (measured per layer at batch 8, 29 x 39, tools/train_layers_bench.py: 256 -> 256 / 96 / 48 take 0.091 / 0.054 / 0.034 ms there
    // against 0.104 / 0.081 / 0.054 here; 512 -> 256, 640 -> 256, 768 -> 512 are no faster there -- 0.152 / 0.196 / 0.409 against
    // 0.143 / 0.162 / 0.421: this file's 128 x 128 tiles re-use a staged element four times -- so only up to 256 input channels)
    if (OSSID_WGRAD_T9 && OSSID_WGRAD_T9_SINGLE && Cin <= 256 && ossid_wgrad_t9_takes(d) && !((uintptr_t)d->workspace & 15) &&
        d->workspace_bytes >= ossid_wgrad_t9_workspace_bytes(d, 1))
        return ossid_wgrad_t9_group(d, 1, d->workspace, d->workspace_bytes, stream);      // the same, split-bf16 (csrc/wgrad_t9.hip)
    WgradPlan p;
    if (!wgrad_plan(B, H, W, Cin, Cout, taps, p)) return OSSID_EINVAL;
    if (d->workspace_bytes < (size_t)p.nsplit * p.wk * taps * Cout * Cin * sizeof(float)) return OSSID_EINVAL;
    WgradArgs a;
    a.x = d->x, a.dy = d->dy, a.pre_scale = d->pre_scale, a.pre_shift = d->pre_shift, a.slabs = (float*)d->workspace;
    a.B = p.B, a.H = p.H, a.W = p.W, a.Cin = Cin, a.Cout = Cout;
    a.in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : Cin;
    a.dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : Cout;
    if ((a.in_cs % 4) || (a.dy_cs % 4) || a.in_cs < Cin || a.dy_cs < Cout) return OSSID_EINVAL;
    a.Hs = d->src_height > 0 ? d->src_height : p.H, a.Ws = d->src_width > 0 ? d->src_width : p.W;
    if ((a.Hs != p.H || a.Ws != p.W) && (taps != 9 || a.Hs > H || a.Ws > W)) return OSSID_EINVAL;
    a.scale_h = (float)a.Hs / (float)p.H, a.scale_w = (float)a.Ws / (float)p.W;
    a.pre_relu = d->pre_relu, a.KT = p.KT, a.chunks_per_row = p.chunks_per_row, a.nsplit = p.nsplit;
    a.co_blocks = p.co_blocks, a.ci_blocks = p.ci_blocks, a.ntiles = p.ntiles, a.n_chunks = p.n_chunks;
    hipStream_t s = (hipStream_t)stream;
    int rc;
    switch (p.variant) {
        case 0: rc = launch_wgrad<9, 1, 2, 1, 2, 2, 1>(a, p, s); break;
        case 1: rc = launch_wgrad<9, 1, 1, 2, 2, 2, 1>(a, p, s); break;
        case 2: rc = launch_wgrad<9, 1, 1, 1, 1, 4, 1>(a, p, s); break;
        case 3: rc = launch_wgrad<1, 1, 2, 2, 2, 2, 1>(a, p, s); break;
        case 4: rc = launch_wgrad<1, 1, 1, 2, 2, 2, 1>(a, p, s); break;
        case 5: rc = launch_wgrad<9, 3, 1, 1, 1, 1, 4>(a, p, s); break;
        default: rc = launch_wgrad<9, 3, 1, 1, 1, 2, 2>(a, p, s); break;
    }
    if (rc != OSSID_OK) return rc;
    const size_t n = (size_t)taps * Cout * Cin;
    const int slabs = p.nsplit * p.wk;
    if (slabs <= 16)
        hipLaunchKernelGGL(wgrad_reduce2_kernel<1>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, (const float*)d->workspace,
                           slabs, taps, Cout, Cin, d->dw, d->accumulate);
    else if (slabs <= 128)
        hipLaunchKernelGGL(wgrad_reduce2_kernel<4>, dim3((unsigned)((n + 63) / 64)), dim3(256), 0, s, (const float*)d->workspace,
                           slabs, taps, Cout, Cin, d->dw, d->accumulate);
    else
        hipLaunchKernelGGL(wgrad_reduce2_kernel<32>, dim3((unsigned)((n + 7) / 8)), dim3(256), 0, s, (const float*)d->workspace,
                           slabs, taps, Cout, Cin, d->dw, d->accumulate);
    return ossid_launch_status();
}

static int fill_wgrad_args(const ossid_wgrad_desc* d, const WgradPlan& p, WgradArgs& a) {
    const int H = d->height, W = d->width, Cin = d->cin, Cout = d->cout, taps = d->taps;
    a.x = d->x, a.dy = d->dy, a.pre_scale = d->pre_scale, a.pre_shift = d->pre_shift, a.slabs = (float*)d->workspace;
    a.B = p.B, a.H = p.H, a.W = p.W, a.Cin = Cin, a.Cout = Cout;
    a.in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : Cin;
    a.dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : Cout;
    if ((a.in_cs % 4) || (a.dy_cs % 4) || a.in_cs < Cin || a.dy_cs < Cout) return OSSID_EINVAL;
    a.Hs = d->src_height > 0 ? d->src_height : p.H, a.Ws = d->src_width > 0 ? d->src_width : p.W;
    if ((a.Hs != p.H || a.Ws != p.W) && (taps != 9 || a.Hs > H || a.Ws > W)) return OSSID_EINVAL;
    a.scale_h = (float)a.Hs / (float)p.H, a.scale_w = (float)a.Ws / (float)p.W;
    a.pre_relu = d->pre_relu, a.KT = p.KT, a.chunks_per_row = p.chunks_per_row, a.nsplit = p.nsplit;
    a.co_blocks = p.co_blocks, a.ci_blocks = p.ci_blocks, a.ntiles = p.ntiles, a.n_chunks = p.n_chunks;
    return OSSID_OK;
}

// How many K-splits each problem of a GROUP gets: the group shares the chip, so the target is just under two workgroups
// per CU over the whole group (ONE round of resident workgroups: 640 measured 1 ms slower per step than 500, the second
// round runs a quarter full), shared out in proportion to each problem's work (chunks x tiles).
static void group_splits(WgradPlan* plans, int n) {
    double total = 0.0;
    for (int i = 0; i < n; ++i) total += (double)plans[i].n_chunks * plans[i].ntiles;
    for (int i = 0; i < n; ++i) {
        const double share = (double)plans[i].n_chunks * plans[i].ntiles / total;
        static const double target = getenv("OSSID_WGRAD_GROUP_BLOCKS") ? atof(getenv("OSSID_WGRAD_GROUP_BLOCKS")) : 500.0;
        long sp = (long)(target * share / plans[i].ntiles + 0.5);
        if (sp < 1) sp = 1;
        if (sp > plans[i].n_chunks) sp = (long)plans[i].n_chunks;
        if (sp > 256) sp = 256;
        plans[i].nsplit = (int)sp;
    }
}

// The problems of a group that csrc/wgrad_t9.hip takes -- `t9`: 3x3, 128 -> 32 (the dense layers' second convolution), `t1`:
// 1x1, c -> 128 (their first) -- provided they share one geometry; the rest stay on the kernels of this file.
struct WgradSplit {
    ossid_wgrad_desc t9[2 * OSSID_WGRAD_GROUP_MAX], t1[2 * OSSID_WGRAD_GROUP_MAX], rest[2 * OSSID_WGRAD_GROUP_MAX];
    int n9, n1, nr;
};
static void split_tiled(const ossid_wgrad_desc* descs, int n, WgradSplit& S) {
    S.n9 = S.n1 = S.nr = 0;
    for (int i = 0; i < n; ++i) {
        const ossid_wgrad_desc& d = descs[i];
        auto same = [&](const ossid_wgrad_desc& o) { return d.batch == o.batch && d.height == o.height && d.width == o.width; };
        if (OSSID_WGRAD_T9 && ossid_wgrad_t9_takes(&d) &&
            (S.n9 == 0 || (same(S.t9[0]) && ossid_wgrad_t9_class(&d) == ossid_wgrad_t9_class(&S.t9[0]))))
            S.t9[S.n9++] = d;
        else if (OSSID_WGRAD_T9 && ossid_wgrad_t1_takes(&d) && (S.n1 == 0 || same(S.t1[0]))) S.t1[S.n1++] = d;
        else S.rest[S.nr++] = d;
    }
}
// the t1 list in calls of at most ossid_wgrad_t1_max_jobs() jobs: [first, first + count)
static int t1_chunk(const ossid_wgrad_desc* t1, int n1, int first) {
    int count = 0;
    while (first + count < n1 && ossid_wgrad_t1_job_count(t1 + first, count + 1) <= ossid_wgrad_t1_max_jobs()) ++count;
    return count;
}

size_t ossid_conv_wgrad_group_workspace_bytes(const ossid_wgrad_desc* descs, int n) {
    if (!descs || n <= 0 || n > 2 * OSSID_WGRAD_GROUP_MAX) return 0;
    {
        WgradSplit S;
        split_tiled(descs, n, S);
        if (S.n9 + S.n1 > 0) {
            size_t total = S.n9 ? (ossid_wgrad_t9_workspace_bytes(S.t9, S.n9) + 255) & ~(size_t)255 : 0;
            for (int f = 0; f < S.n1;) {
                const int cnt = t1_chunk(S.t1, S.n1, f);
                if (cnt <= 0) return 0;
                total += (ossid_wgrad_t1_workspace_bytes(S.t1 + f, cnt) + 255) & ~(size_t)255;
                f += cnt;
            }
            if (S.nr == 0) return total;
            const size_t b = ossid_conv_wgrad_group_workspace_bytes(S.rest, S.nr);     // (nothing eligible left: one level of recursion)
            return b ? total + b : 0;
        }
    }
    WgradPlan plans[4 * OSSID_WGRAD_GROUP_MAX];
    for (int i = 0; i < n; ++i)
        if (!wgrad_plan(descs[i].batch, descs[i].height, descs[i].width, descs[i].cin, descs[i].cout, descs[i].taps, plans[i]))
            return 0;
    // splits are chosen per variant bucket, as ossid_conv_wgrad_group does
    size_t total = 0;
    for (int v = 0; v < 7; ++v) {
        WgradPlan sub[4 * OSSID_WGRAD_GROUP_MAX];
        int idx[4 * OSSID_WGRAD_GROUP_MAX], m = 0;
        for (int i = 0; i < n; ++i)
            if (plans[i].variant == v) sub[m] = plans[i], idx[m++] = i;
        for (int c0 = 0; c0 < m; c0 += OSSID_WGRAD_GROUP_MAX) {
            const int cnt = m - c0 < OSSID_WGRAD_GROUP_MAX ? m - c0 : OSSID_WGRAD_GROUP_MAX;
            group_splits(sub + c0, cnt);
            for (int j = 0; j < cnt; ++j) {
                const ossid_wgrad_desc& d = descs[idx[c0 + j]];
                total += ((size_t)sub[c0 + j].nsplit * sub[c0 + j].wk * d.taps * d.cout * d.cin * sizeof(float) + 255) & ~(size_t)255;
            }
        }
    }
    return total;
}

int ossid_conv_wgrad_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (!descs || n <= 0 || n > 2 * OSSID_WGRAD_GROUP_MAX || !workspace) return OSSID_EINVAL;
    if (workspace_bytes < ossid_conv_wgrad_group_workspace_bytes(descs, n)) return OSSID_EINVAL;
    {
        WgradSplit S;
        split_tiled(descs, n, S);
        if (S.n9 + S.n1 > 0) {
            char* ws = (char*)workspace;
            if (S.n9) {
                const size_t a = (ossid_wgrad_t9_workspace_bytes(S.t9, S.n9) + 255) & ~(size_t)255;
                const int rc = ossid_wgrad_t9_group(S.t9, S.n9, ws, a, stream);
                if (rc != OSSID_OK) return rc;
                ws += a;
            }
            for (int f = 0; f < S.n1;) {
                const int cnt = t1_chunk(S.t1, S.n1, f);
                if (cnt <= 0) return OSSID_EINVAL;
                const size_t a = (ossid_wgrad_t1_workspace_bytes(S.t1 + f, cnt) + 255) & ~(size_t)255;
                const int rc = ossid_wgrad_t1_group(S.t1 + f, cnt, ws, a, stream);
                if (rc != OSSID_OK) return rc;
                ws += a, f += cnt;
            }
            if (S.nr == 0) return OSSID_OK;
            return ossid_conv_wgrad_group(S.rest, S.nr, ws, workspace_bytes - (size_t)(ws - (char*)workspace), stream);
        }
    }
    WgradPlan plans[4 * OSSID_WGRAD_GROUP_MAX];
    for (int i = 0; i < n; ++i) {
        const ossid_wgrad_desc& d = descs[i];
        if (d.batch <= 0 || d.height <= 0 || d.width <= 0 || d.cin <= 0 || d.cout <= 0 || (d.cin % 4) || (d.cout % 4) || !d.x ||
            !d.dy || !d.dw || (d.pre_scale && !d.pre_shift) || d.dy_add)
            return OSSID_EINVAL;
        if (!wgrad_plan(d.batch, d.height, d.width, d.cin, d.cout, d.taps, plans[i])) return OSSID_EINVAL;
    }
    hipStream_t s = (hipStream_t)stream;
    char* ws = (char*)workspace;
    ReduceGroup red;
    red.n = 0;
    int red_blocks = 0;
    for (int v = 0; v < 7; ++v) {
        int idx[4 * OSSID_WGRAD_GROUP_MAX], m = 0;
        for (int i = 0; i < n; ++i)
            if (plans[i].variant == v) idx[m++] = i;
        for (int c0 = 0; c0 < m; c0 += OSSID_WGRAD_GROUP_MAX) {
            const int cnt = m - c0 < OSSID_WGRAD_GROUP_MAX ? m - c0 : OSSID_WGRAD_GROUP_MAX;
            WgradPlan sub[OSSID_WGRAD_GROUP_MAX];
            for (int j = 0; j < cnt; ++j) sub[j] = plans[idx[c0 + j]];
            group_splits(sub, cnt);
            WgradGroup g;
            g.n = cnt;
            g.first_block[0] = 0;
            size_t lds = 0;
            for (int j = 0; j < cnt; ++j) {
                ossid_wgrad_desc d = descs[idx[c0 + j]];
                d.workspace = ws;
                const size_t bytes = (size_t)sub[j].nsplit * sub[j].wk * d.taps * d.cout * d.cin * sizeof(float);
                ws += (bytes + 255) & ~(size_t)255;
                const int rc = fill_wgrad_args(&d, sub[j], g.a[j]);
                if (rc != OSSID_OK) return rc;
                g.first_block[j + 1] = g.first_block[j] + sub[j].nsplit * sub[j].ntiles;
                if (sub[j].lds > lds) lds = sub[j].lds;
                ReduceRow& r = red.r[red.n++];
                r.slabs = (const float*)d.workspace, r.dw = d.dw, r.nslabs = sub[j].nsplit * sub[j].wk, r.taps = d.taps;
                r.Cout = d.cout, r.Cin = d.cin, r.accumulate = d.accumulate, r.first_block = red_blocks;
                red_blocks += (int)(((size_t)d.taps * d.cout * d.cin + 63) / 64);
            }
            int rc;
            switch (v) {
                case 0: rc = launch_wgrad_group<9, 1, 2, 1, 2, 2, 1>(g, lds, sub[0].sb != 0, s); break;
                case 1: rc = launch_wgrad_group<9, 1, 1, 2, 2, 2, 1>(g, lds, sub[0].sb != 0, s); break;
                case 2: rc = launch_wgrad_group<9, 1, 1, 1, 1, 4, 1>(g, lds, sub[0].sb != 0, s); break;
                case 3: rc = launch_wgrad_group<1, 1, 2, 2, 2, 2, 1>(g, lds, sub[0].sb != 0, s); break;
                case 4: rc = launch_wgrad_group<1, 1, 1, 2, 2, 2, 1>(g, lds, sub[0].sb != 0, s); break;
                case 5: rc = launch_wgrad_group<9, 3, 1, 1, 1, 1, 4>(g, lds, sub[0].sb != 0, s); break;
                default: rc = launch_wgrad_group<9, 3, 1, 1, 1, 2, 2>(g, lds, sub[0].sb != 0, s); break;
            }
            if (rc != OSSID_OK) return rc;
        }
    }
    if (red.n > 0) hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3((unsigned)red_blocks), dim3(256), 0, s, red);
    return ossid_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// D15  DetectionLoss (models/dtoid/loss.py:46-175): focal classification loss with IoU anchor assignment + smooth-L1 box
// regression, forward and the gradients with respect to both network outputs in one pass over the anchors.
//   per image b: valid annotations = rows with label != -1; IoU of every anchor with every valid annotation (calc_iou,
//   loss.py:10-37, union clamped at 1e-8); max / first argmax; positive: IoU >= 0.5, negative: IoU < 0.4, in between ignored;
//   p = clamp(prob, 1e-4, 1 - 1e-4); target 1 for the assigned annotation's class of a positive anchor, 0 for the other
//   classes of positives and for all classes of negatives: L = alpha (1-p)^gamma (-log p) | (1-alpha) p^gamma (-log(1-p));
//   sum / max(#positives, 1)  (an image without annotation: every anchor negative, sum not divided, loss.py:78-93);
//   smooth-L1 (beta 1/9) on the positives between reg and the encoded box ((dx, dy)/0.1, (log dw, log dh)/0.2, widths
//   clamped at 1), mean over #positives x 4; both losses are then averaged over the batch.
// Kernel 1 writes UN-normalised gradients and per-block partial sums; kernel 2 (one wave per image) adds the partials
// in a fixed order, writes the two losses and the per-image normalisers; the backward kernel scales the stored gradients.
namespace {

constexpr int LOSS_MAXG = 16;

__global__ __launch_bounds__(256) void det_loss_fwd_kernel(const float* __restrict__ cls, const float* __restrict__ reg,
                                                           const float4* __restrict__ anchors, const float* __restrict__ ann,
                                                           int A, int C, int G, float alpha, float gamma,
                                                           float* __restrict__ dcls_raw, float* __restrict__ dreg_raw,
                                                           float* __restrict__ partials) {
    __shared__ float sg[LOSS_MAXG][5];
    __shared__ float red[4][3];
    const int b = blockIdx.y, a = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < G * 5) sg[threadIdx.x / 5][threadIdx.x % 5] = ann[((size_t)b * G) * 5 + threadIdx.x];
    __syncthreads();
    float s_cls = 0.0f, s_reg = 0.0f, s_pos = 0.0f;
    if (a < A) {
        const float4 an = anchors[a];
        const float aw = an.z - an.x, ah = an.w - an.y, area_a = aw * ah;
        float best = -1.0f;
        int arg = -1;
        for (int g = 0; g < G; ++g) {
            if (sg[g][4] == -1.0f) continue;
            float iw = fminf(an.z, sg[g][2]) - fmaxf(an.x, sg[g][0]), ih = fminf(an.w, sg[g][3]) - fmaxf(an.y, sg[g][1]);
            iw = fmaxf(iw, 0.0f), ih = fmaxf(ih, 0.0f);
            const float inter = iw * ih;
            const float ua = fmaxf(area_a + (sg[g][2] - sg[g][0]) * (sg[g][3] - sg[g][1]) - inter, 1e-8f);
            const float iou = inter / ua;
            if (iou > best) best = iou, arg = g;
        }
        const bool has_gt = arg >= 0;
        const bool positive = has_gt && best >= 0.5f;
        const bool negative = !has_gt || best < 0.4f;
        const int label = positive ? (int)sg[arg][4] : -1;
        const float* pc = cls + ((size_t)b * A + a) * C;
        float* gc = dcls_raw + ((size_t)b * A + a) * C;
        for (int k = 0; k < C; ++k) {
            const float praw = pc[k];
            const float p = fminf(fmaxf(praw, 1e-4f), 1.0f - 1e-4f);
            const float pass = (praw >= 1e-4f && praw <= 1.0f - 1e-4f) ? 1.0f : 0.0f;     // clamp's gradient
            float L = 0.0f, dL = 0.0f;
            if (positive && k == label) {
                const float q = 1.0f - p, lg = logf(p), w = powf(q, gamma);
                L = -alpha * w * lg;
                dL = alpha * (gamma * powf(q, gamma - 1.0f) * lg - w / p);
            } else if (positive || negative) {
                const float lg = logf(1.0f - p), w = powf(p, gamma);
                L = -(1.0f - alpha) * w * lg;
                dL = (1.0f - alpha) * (-gamma * powf(p, gamma - 1.0f) * lg + w / (1.0f - p));
            }
            s_cls += L;
            gc[k] = dL * pass;
        }
        float4 gr = make_float4(0.f, 0.f, 0.f, 0.f);
        if (positive) {
            s_pos = 1.0f;
            const float acx = an.x + 0.5f * aw, acy = an.y + 0.5f * ah;
            float gw = sg[arg][2] - sg[arg][0], gh = sg[arg][3] - sg[arg][1];
            const float gcx = sg[arg][0] + 0.5f * gw, gcy = sg[arg][1] + 0.5f * gh;
            gw = fmaxf(gw, 1.0f), gh = fmaxf(gh, 1.0f);
            const float t[4] = {(gcx - acx) / aw / 0.1f, (gcy - acy) / ah / 0.1f, logf(gw / aw) / 0.2f, logf(gh / ah) / 0.2f};
            const float4 r4 = *(const float4*)(reg + ((size_t)b * A + a) * 4);
            const float r[4] = {r4.x, r4.y, r4.z, r4.w};
            float g4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float diff = t[k] - r[k], d = fabsf(diff);
                const bool quad = d <= 1.0f / 9.0f;
                s_reg += quad ? 0.5f * 9.0f * d * d : d - 0.5f / 9.0f;
                const float sgn = diff > 0.0f ? 1.0f : (diff < 0.0f ? -1.0f : 0.0f);
                g4[k] = -sgn * (quad ? 9.0f * d : 1.0f);
            }
            gr = make_float4(g4[0], g4[1], g4[2], g4[3]);
        }
        *(float4*)(dreg_raw + ((size_t)b * A + a) * 4) = gr;
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) {
        s_cls += __shfl_xor(s_cls, m);
        s_reg += __shfl_xor(s_reg, m);
        s_pos += __shfl_xor(s_pos, m);
    }
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][0] = s_cls, red[threadIdx.x >> 6][1] = s_reg, red[threadIdx.x >> 6][2] = s_pos;
    __syncthreads();
    if (threadIdx.x < 3)
        partials[((size_t)b * gridDim.x + blockIdx.x) * 3 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// one wave per image: partial sums in a fixed order (lane-strided, then a fixed butterfly); wave 0 lane 0 of the LAST
// step is replaced by a second launch-free trick: the batch means are formed by block 0 after a grid of B waves has
// written the per-image values -- here simply ONE block handles all images (B <= 1024 / 64 waves at a time, looped).
__global__ __launch_bounds__(64) void det_loss_finalize_kernel(const float* __restrict__ partials, const float* __restrict__ ann,
                                                               int B, int G, int nblk, float* __restrict__ losses,
                                                               float* __restrict__ scales) {
    double tot_cls = 0.0, tot_reg = 0.0;
    for (int b = 0; b < B; ++b) {
        double s[3] = {0.0, 0.0, 0.0};
        for (int i = threadIdx.x; i < nblk; i += 64)
#pragma unroll
            for (int k = 0; k < 3; ++k) s[k] += (double)partials[((size_t)b * nblk + i) * 3 + k];
#pragma unroll
        for (int k = 0; k < 3; ++k)
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) s[k] += __shfl_xor(s[k], m);
        bool has_gt = false;
        for (int g = 0; g < G; ++g) has_gt |= ann[((size_t)b * G + g) * 5 + 4] != -1.0f;
        const double ncls = has_gt ? fmax(s[2], 1.0) : 1.0;
        const double nreg = s[2] > 0.0 ? 4.0 * s[2] : 1.0;
        tot_cls += s[0] / ncls;
        tot_reg += s[2] > 0.0 ? s[1] / nreg : 0.0;
        if (threadIdx.x == 0) {
            scales[b] = (float)(1.0 / (ncls * B));
            scales[B + b] = (float)(s[2] > 0.0 ? 1.0 / (nreg * B) : 0.0);
        }
    }
    if (threadIdx.x == 0) losses[0] = (float)(tot_cls / B), losses[1] = (float)(tot_reg / B);
}

__global__ __launch_bounds__(256) void det_loss_bwd_kernel(const float* __restrict__ dcls_raw, const float* __restrict__ dreg_raw,
                                                           const float* __restrict__ scales, const float* __restrict__ gout,
                                                           int B, int A, int C, float* __restrict__ dcls,
                                                           float* __restrict__ dreg) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t n = (size_t)B * A;
    if (i >= n) return;
    const int b = (int)(i / A);
    const float sc = gout[0] * scales[b], sr = gout[1] * scales[B + b];
    for (int k = 0; k < C; ++k) dcls[i * C + k] = sc * dcls_raw[i * C + k];
    const float4 g = *(const float4*)(dreg_raw + i * 4);
    *(float4*)(dreg + i * 4) = make_float4(sr * g.x, sr * g.y, sr * g.z, sr * g.w);
}

}  // namespace

extern "C" {

size_t ossid_focal_smoothl1_loss_workspace_floats(int B, int A) {
    return B > 0 && A > 0 ? (size_t)B * ((A + 255) / 256) * 3 : 0;
}

int ossid_focal_smoothl1_loss_fwd(const float* cls, const float* reg, const float* anchors, const float* annotations, int B,
                                  int A, int C, int G, float alpha, float gamma, float* dcls_raw, float* dreg_raw,
                                  float* workspace, float* losses2, float* scales2B, void* stream) {
    if (!cls || !reg || !anchors || !annotations || !dcls_raw || !dreg_raw || !workspace || !losses2 || !scales2B) return OSSID_EINVAL;
    if (B <= 0 || B > 65535 || A <= 0 || C <= 0 || C > 64 || G <= 0 || G > LOSS_MAXG) return OSSID_EINVAL;
    const int nblk = (A + 255) / 256;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(det_loss_fwd_kernel, dim3(nblk, B), dim3(256), 0, s, cls, reg, (const float4*)anchors, annotations, A, C, G,
                       alpha, gamma, dcls_raw, dreg_raw, workspace);
    hipLaunchKernelGGL(det_loss_finalize_kernel, dim3(1), dim3(64), 0, s, (const float*)workspace, annotations, B, G, nblk,
                       losses2, scales2B);
    return ossid_launch_status();
}

int ossid_focal_smoothl1_loss_bwd(const float* dcls_raw, const float* dreg_raw, const float* scales2B, const float* grad_losses2,
                                  int B, int A, int C, float* dcls, float* dreg, void* stream) {
    if (!dcls_raw || !dreg_raw || !scales2B || !grad_losses2 || !dcls || !dreg || B <= 0 || A <= 0 || C <= 0) return OSSID_EINVAL;
    const size_t n = (size_t)B * A;
    hipLaunchKernelGGL(det_loss_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, dcls_raw,
                       dreg_raw, scales2B, grad_losses2, B, A, C, dcls, dreg);
    return ossid_launch_status();
}

}  // extern "C"
