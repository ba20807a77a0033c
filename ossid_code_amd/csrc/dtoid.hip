// DTOID device ops for gfx950 that are not plain dense convolutions:
//   - per-sample depthwise cross-correlation with a data-dependent 3x3 kernel (forward + both backward passes),
//     the op behind conv2d_dw_group (/root/reference/python/ossid/models/dtoid/network.py:186-192, :365-371),
//     which the reference runs as a groups=B*C grouped convolution;
//   - greedy IoU NMS, the op behind torchvision.ops.nms (network.py:563, models/dtoid/utils.py:33);
//   - anchor box decode + clip (BBoxTransform / ClipBoxes, network.py:42-70, :78-88);
//   - the AMSGrad-Adam update of online_learning.py:258-263 as ONE launch over a flat parameter buffer.
// All of them are HBM-bound elementwise / stencil / reduction kernels: coalesced 4-byte lanes over rows, LDS halo
// tiles for the stencil, wave reductions for the kernel-gradient sums.
#include "common.h"

namespace {

// ---- depthwise 3x3 cross-correlation, zero padding 1 -------------------------------------------------------------
// out[p][y][x] = sum_{i,j} in[p][y+i-1][x+j-1] * k[p][i][j]      (flip = 0; the forward pass)
// with flip = 1 the kernel is applied rotated by 180 degrees, which turns the same routine into the gradient w.r.t.
// the input: dx = dout (*) rot180(k).
constexpr int DW_TX = 64, DW_TY = 4;
__global__ __launch_bounds__(256) void dw_xcorr_kernel(const float* __restrict__ in, const float* __restrict__ k, int H,
                                                       int W, int flip, float* __restrict__ out, int in_planes,
                                                       int plane0) {
    __shared__ float tile[(DW_TY + 2) * (DW_TX + 2)];
    const int p = blockIdx.z;
    const int x0 = blockIdx.x * DW_TX - 1, y0 = blockIdx.y * DW_TY - 1;
    // in_planes < planes: the input is ONE image [C,H,W] shared by every template (test time, network.py:512)
    const float* src = in + (size_t)((plane0 + p) % in_planes) * H * W;
    for (int i = threadIdx.x; i < (DW_TY + 2) * (DW_TX + 2); i += 256) {
        int lx = i % (DW_TX + 2), ly = i / (DW_TX + 2);
        int gx = x0 + lx, gy = y0 + ly;
        tile[i] = (gx >= 0 && gx < W && gy >= 0 && gy < H) ? src[(size_t)gy * W + gx] : 0.0f;
    }
    float w[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) w[i] = k[(size_t)p * 9 + (flip ? 8 - i : i)];
    __syncthreads();
    const int tx = threadIdx.x % DW_TX, ty = threadIdx.x / DW_TX;
    const int x = blockIdx.x * DW_TX + tx, y = blockIdx.y * DW_TY + ty;
    if (x >= W || y >= H) return;
    float acc = 0.0f;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) acc = fmaf(tile[(ty + i) * (DW_TX + 2) + tx + j], w[3 * i + j], acc);
    out[(size_t)p * H * W + (size_t)y * W + x] = acc;
}

// dk[p][i][j] = sum_{y,x} dout[p][y][x] * in[p][y+i-1][x+j-1]: one workgroup per plane, nine running sums per
// lane, then a 64-lane butterfly and a 4-entry LDS combine (fixed order, so the result is reproducible).
__global__ __launch_bounds__(256) void dw_xcorr_bwd_k_kernel(const float* __restrict__ in,
                                                             const float* __restrict__ dout, int H, int W,
                                                             float* __restrict__ dk) {
    __shared__ float red[4][9];
    const int p = blockIdx.x;
    const float* a = in + (size_t)p * H * W;
    const float* g = dout + (size_t)p * H * W;
    float s[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) s[i] = 0.0f;
    for (int idx = threadIdx.x; idx < H * W; idx += 256) {
        const int y = idx / W, x = idx - y * W;
        const float gv = g[idx];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int yy = y + i - 1;
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                const int xx = x + j - 1;
                const float v = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? a[(size_t)yy * W + xx] : 0.0f;
                s[3 * i + j] = fmaf(gv, v, s[3 * i + j]);
            }
        }
    }
#pragma unroll
    for (int i = 0; i < 9; ++i)
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s[i] += __shfl_xor(s[i], m);
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int i = 0; i < 9; ++i) red[threadIdx.x >> 6][i] = s[i];
    __syncthreads();
    if (threadIdx.x < 9)
        dk[(size_t)p * 9 + threadIdx.x] =
            (red[0][threadIdx.x] + red[1][threadIdx.x]) + (red[2][threadIdx.x] + red[3][threadIdx.x]);
}

// ---- NMS ---------------------------------------------------------------------------------------------------------
// boxes [n][4] (x1,y1,x2,y2) sorted by descending score. Pass 1: 64x64 blocks of the upper triangle -> bitmask of
// "j is suppressed by i" (IoU > thr). Pass 2: one wave walks the boxes in order, OR-ing the masks of the kept ones.
__device__ __forceinline__ float iou(const float4 a, const float4 b) {
    float iw = fminf(a.z, b.z) - fmaxf(a.x, b.x), ih = fminf(a.w, b.w) - fmaxf(a.y, b.y);
    iw = fmaxf(iw, 0.0f);
    ih = fmaxf(ih, 0.0f);
    float inter = iw * ih;
    float ua = ((a.z - a.x) * (a.w - a.y) + (b.z - b.x) * (b.w - b.y)) - inter;
    return inter / ua;
}

// zero n 16-byte units (instead of hipMemsetAsync: these entry points also run inside captured graphs, beside ~190 kernel
// nodes, where a memset node aborted at replay)
__global__ __launch_bounds__(256) void zero16_kernel(float4* __restrict__ p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) p[i] = make_float4(0.f, 0.f, 0.f, 0.f);
}
static inline void zero_async(void* p, size_t bytes, hipStream_t s) {      // p 16-byte aligned, bytes a multiple of 16
    const size_t n = bytes / 16;
    unsigned blocks = (unsigned)((n + 255) / 256);
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(zero16_kernel, dim3(blocks), dim3(256), 0, s, (float4*)p, n);
}

__global__ __launch_bounds__(64) void nms_mask_kernel(const float4* __restrict__ boxes, int n, float thr,
                                                      unsigned long long* __restrict__ mask, int words) {
    const int rb = blockIdx.y, cb = blockIdx.x;
    if (cb < rb) return;
    __shared__ float4 cbox[64];
    const int ci = cb * 64 + threadIdx.x;
    if (ci < n) cbox[threadIdx.x] = boxes[ci];
    __syncthreads();
    const int i = rb * 64 + threadIdx.x;
    if (i >= n) return;
    const float4 bi = boxes[i];
    unsigned long long bits = 0;
    const int jn = min(64, n - cb * 64);
    for (int j = (rb == cb) ? threadIdx.x + 1 : 0; j < jn; ++j)
        if (iou(bi, cbox[j]) > thr) bits |= 1ull << j;
    mask[(size_t)i * words + cb] = bits;
}

__global__ __launch_bounds__(64) void nms_scan_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                                      int* __restrict__ keep, int* __restrict__ nkeep) {
    extern __shared__ unsigned long long remv[];
    for (int w = threadIdx.x; w < words; w += 64) remv[w] = 0;
    __syncthreads();
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const bool dead = (remv[i >> 6] >> (i & 63)) & 1ull;   // same word for every lane: wave-uniform
        if (!dead) {
            if (threadIdx.x == 0) keep[cnt] = i;
            ++cnt;
            for (int w = (i >> 6) + threadIdx.x; w < words; w += 64) remv[w] |= mask[(size_t)i * words + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) *nkeep = cnt;
}

// Up to 1 024 boxes (words <= 16: the test-time case, top-1000 candidates): the whole suppression matrix is staged in LDS
// (<= 128 KB) by 256 threads, then ONE wave runs the greedy scan 64 boxes at a time. Inside a block of 64 the decisions depend
// on each other only through the block's own 64 x 64 diagonal piece of the matrix: lane j holds row 64b + j of it, the scan over
// kept boxes runs on the scalar unit (find-first-set over what is neither removed nor taken, v_readlane of that row, an or --
// instead of an LDS round trip and a cross-lane shuffle per candidate). The rows of the boxes that were kept are then or-ed into the removed set of all later words
// (lane w owns word w) with independent LDS reads, four per trip. 1 000 candidates: 100 -> 41 us (300 kept) .. 75 us (740
// kept), tools/nms_bench.py under rocprofv3; same keep list, same order.
__global__ __launch_bounds__(256) void nms_scan_small_kernel(const unsigned long long* __restrict__ mask, int n, int words,
                                                             int* __restrict__ keep, int* __restrict__ nkeep) {
    extern __shared__ unsigned long long lmask[];
    const int total = n * words;
    for (int i = threadIdx.x; i < total; i += 256) lmask[i] = mask[i];
    __syncthreads();
    if (threadIdx.x >= 64) return;
    const int lane = threadIdx.x;
    unsigned long long remv = 0;                     // word `lane` of the removed set (lanes >= words stay 0)
    int cnt = 0;
    for (int b = 0; b < words; ++b) {
        const int row = 64 * b + lane;
        const unsigned long long diag = row < n ? lmask[(size_t)row * words + b] : 0ull;
        const unsigned lo = (unsigned)diag, hi = (unsigned)(diag >> 32);
        // removed bits of this block so far: word b lives in lane b (two v_readlane: a scalar pair)
        // (the builtin returns int: through unsigned, or bit 31 of the low word would sign-extend over the high one)
        unsigned long long cur = ((unsigned long long)(unsigned)__builtin_amdgcn_readlane((unsigned)(remv >> 32), b) << 32) |
                                 (unsigned)__builtin_amdgcn_readlane((unsigned)remv, b);
        const int nb = n - 64 * b < 64 ? n - 64 * b : 64;
        unsigned long long kept = 0;
        const unsigned long long valid = nb == 64 ? ~0ull : (1ull << nb) - 1ull;
        for (;;) {                                   // scalar loop (every operand is wave-uniform), one trip per KEPT box:
            const unsigned long long avail = ~cur & valid;       // the lowest box neither removed nor taken yet is the next kept one
            if (!avail) break;                                   // (a row only has bits above its own index)
            const int j = __ffsll((long long)avail) - 1;
            kept |= 1ull << j;
            cur |= (1ull << j) | ((unsigned long long)(unsigned)__builtin_amdgcn_readlane(hi, j) << 32) |
                   (unsigned)__builtin_amdgcn_readlane(lo, j);
        }
        // keep list: lane j of a kept box writes its index at the running count + the kept boxes before it in the block
        if ((kept >> lane) & 1ull) keep[cnt + __popcll(kept & ((1ull << lane) - 1ull))] = row;
        cnt += __popcll(kept);
        // the kept rows suppress later boxes in every word (independent loads, or-ed as they arrive)
        // (four rows per trip: the reads are issued together, a dependent chain of LDS latencies per kept box is what the
        // loop would otherwise be; a row index past the last kept one re-reads row 64b of the block -- harmless only if that
        // row is itself kept, so the fill value is the block's first kept row)
        if (lane < words && kept) {
            unsigned long long k2 = kept;
            const int first = __ffsll((long long)kept) - 1;
            const unsigned long long* base = lmask + (size_t)(64 * b) * words + lane;
            while (k2) {
                int j[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    j[u] = k2 ? __ffsll((long long)k2) - 1 : first;
                    k2 &= k2 - 1;
                }
                const unsigned long long r0 = base[(size_t)j[0] * words], r1 = base[(size_t)j[1] * words];
                const unsigned long long r2 = base[(size_t)j[2] * words], r3 = base[(size_t)j[3] * words];
                remv |= (r0 | r1) | (r2 | r3);
            }
        }
    }
    if (lane == 0) *nkeep = cnt;
}

// ---- depthwise cross-correlation, channels-last, ONE image against B kernels (the test-time correlation of
// network.py:365-371 with the image broadcast over the templates): out[b][y][x][c] = sum_taps x[y+dy][x+dx][c] * k[b][c][tap].
// A thread owns 4 consecutive channels of a strip of DWX consecutive output pixels of one row: its 36 kernel taps live in
// registers, the 3 x (DWX + 2) input window (L2-resident image) is loaded in one go -- 30 independent 16-byte reads for 8
// outputs instead of 72, and one round trip of latency per strip (round 3's one-pixel threads: 0.14 ms for a 61 MB result,
// latency- not bandwidth-bound) -- one 16-byte store per pixel: the result is born in the layout the next convolution stages from.
constexpr int DWX = 8;
__global__ __launch_bounds__(256) void dw_xcorr_nhwc_kernel(const float4* __restrict__ x, const float* __restrict__ k,
                                                            int C4, int H, int W, int strips, size_t total,
                                                            float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xa = (int)(r % strips) * DWX;
    r /= strips;
    const int yy = (int)(r % H), b = (int)(r / H);
    const float* kk = k + ((size_t)b * C4 + c4) * 36;          // [4 channels][9 taps]
    float w[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) w[t] = kk[t];
    float4 v[3][DWX + 2];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = yy + dy - 1;
        const bool rv = y >= 0 && y < H;
#pragma unroll
        for (int j = 0; j < DWX + 2; ++j) {
            const int xq = xa - 1 + j;
            v[dy][j] = (rv && xq >= 0 && xq < W) ? x[((size_t)y * W + xq) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
#pragma unroll
    for (int j = 0; j < DWX; ++j) {
        if (xa + j >= W) break;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float4 q = v[dy][j + dx];
                const int t = dy * 3 + dx;
                acc.x = fmaf(q.x, w[t], acc.x);
                acc.y = fmaf(q.y, w[9 + t], acc.y);
                acc.z = fmaf(q.z, w[18 + t], acc.z);
                acc.w = fmaf(q.w, w[27 + t], acc.w);
            }
        out[(((size_t)b * H + yy) * W + xa + j) * C4 + c4] = acc;
    }
}

// ---- conv(image - avg_t) without the convolution: conv is linear, so conv(x - a_t) = conv(x) - conv(a_t), the first term
// is the SAME for every template (one batch-1 convolution per frame instead of one per template) and the second is the
// response to a per-channel constant image: a per-template vector for each of the 9 border patterns (which taps fall
// inside the frame), one tiny GEMM per object (`csub`). This kernel finishes `norm(F.elu(conv(image_feat - avg)))` of
// network.py:346 for all templates: out[t][px][coff + o] = post(ELU(S[px][o] - csub[t][pattern(px)][o])), S with bias.
__global__ __launch_bounds__(256) void bcast_sub_epilogue_kernel(const float4* __restrict__ S, const float4* __restrict__ csub,
                                                                 int C4, int H, int W, size_t total,
                                                                 const float4* __restrict__ sc, const float4* __restrict__ sh,
                                                                 float* __restrict__ out, int out_cs, int out_coff) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int o4 = (int)(i % C4);
    size_t r = i / C4;
    const int px = (int)(r % ((size_t)H * W)), t = (int)(r / ((size_t)H * W));
    const int y = px / W, x = px - y * W;
    const int pat = (y == 0 ? 0 : (y == H - 1 ? 2 : 1)) * 3 + (x == 0 ? 0 : (x == W - 1 ? 2 : 1));
    const float4 s = S[(size_t)px * C4 + o4], c = csub[((size_t)t * 9 + pat) * C4 + o4];
    float v[4] = {s.x - c.x, s.y - c.y, s.z - c.z, s.w - c.w};
    const float4 a = sc ? sc[o4] : make_float4(1.f, 1.f, 1.f, 1.f), b = sh ? sh[o4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float av[4] = {a.x, a.y, a.z, a.w}, bv[4] = {b.x, b.y, b.z, b.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u = elu_fast(v[k]);
        v[k] = u * av[k] + bv[k];
    }
    *(float4*)(out + ((size_t)t * H * W + px) * out_cs + out_coff + 4 * o4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- conv(image * avg_t) with the channel contraction LAST: out_t[px][o] = sum_c avg_t[c] * G[c][px][o],
// G[c][px][o] = sum_taps W[o][c][tap] * x[px + tap][c] (zero outside the frame). G does not depend on the template, so for
// many templates a 640-deep GEMM per template (library GEMM on G viewed as [C][HW*Cout]) replaces a 5 760-deep
// convolution per template (network.py:345). This kernel builds G: one workgroup per (channel, 64-pixel run); a thread
// keeps the 9 taps of its four output channels in registers and walks the pixels; 16-byte stores, 1 KB per pixel row.
__global__ __launch_bounds__(256) void dot_expand_kernel(const float* __restrict__ x, const float4* __restrict__ wg,
                                                         int C, int Cout4, int H, int W, float4* __restrict__ G) {
    const int c = blockIdx.y, o4 = threadIdx.x % Cout4, pl = threadIdx.x / Cout4, np = 256 / Cout4;
    const int HW = H * W;
    float4 w[9];
#pragma unroll
    for (int t = 0; t < 9; ++t) w[t] = wg[((size_t)c * 9 + t) * Cout4 + o4];
    const int p0 = blockIdx.x * 64;
    for (int p = p0 + pl; p < min(p0 + 64, HW); p += np) {
        const int y = p / W, xx = p - y * W;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = y + dy - 1;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const int xq = xx + dx - 1;
                if (xq < 0 || xq >= W) continue;
                const float f = x[((size_t)yy * W + xq) * C + c];
                const float4 ww = w[dy * 3 + dx];
                acc.x = fmaf(ww.x, f, acc.x), acc.y = fmaf(ww.y, f, acc.y), acc.z = fmaf(ww.z, f, acc.z), acc.w = fmaf(ww.w, f, acc.w);
            }
        }
        G[((size_t)c * HW + p) * Cout4 + o4] = acc;
    }
}

// out[t][px][coff + o] = post(ELU(z[t][px][o] + bias[o])): the epilogue behind the GEMM above
__global__ __launch_bounds__(256) void bias_elu_affine_slice_kernel(const float4* __restrict__ z, const float4* __restrict__ bias,
                                                                    const float4* __restrict__ sc, const float4* __restrict__ sh,
                                                                    int C4, size_t rows, float* __restrict__ out, int out_cs,
                                                                    int out_coff) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * C4) return;
    const int o4 = (int)(i % C4);
    const size_t r = i / C4;
    const float4 v4 = z[i], b = bias ? bias[o4] : make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 a = sc ? sc[o4] : make_float4(1.f, 1.f, 1.f, 1.f), s = sh ? sh[o4] : make_float4(0.f, 0.f, 0.f, 0.f);
    float v[4] = {v4.x + b.x, v4.y + b.y, v4.z + b.z, v4.w + b.w};
    const float av[4] = {a.x, a.y, a.z, a.w}, sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const float u = elu_fast(v[k]);
        v[k] = u * av[k] + sv[k];
    }
    *(float4*)(out + r * out_cs + out_coff + 4 * o4) = make_float4(v[0], v[1], v[2], v[3]);
}

// ---- row gather (+ sigmoid): out[k][:] = f(src[idx[k]][:]) -- the per-detection segmentation maps picked out of the
// per-template ones (network.py:575-579) with the sigmoid of dtoid/__init__.py:147 applied on the way, one pass instead of
// a gather and an elementwise kernel over up to 500 x 480 x 640 floats
__global__ __launch_bounds__(256) void gather_rows_kernel(const float4* __restrict__ src, size_t row4,
                                                          const long long* __restrict__ idx, int apply_sigmoid,
                                                          float4* __restrict__ out) {
    const size_t k = blockIdx.y;
    const float4* s = src + (size_t)idx[k] * row4;
    float4* o = out + k * row4;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < row4; i += (size_t)gridDim.x * 256) {
        float4 v = s[i];
        if (apply_sigmoid) {
            v.x = 1.0f / (1.0f + expf(-v.x)), v.y = 1.0f / (1.0f + expf(-v.y));
            v.z = 1.0f / (1.0f + expf(-v.z)), v.w = 1.0f / (1.0f + expf(-v.w));
        }
        o[i] = v;
    }
}

// ---- anchor decode + clip ------------------------------------------------------------------------------------------
// anchors [A][4] shared by every batch row, deltas [R][A][4] -> boxes [R][A][4]; std (.1,.1,.2,.2), mean 0.
__device__ __forceinline__ float4 decode_clip_one(const float4 a, const float4 d, float img_w, float img_h);
__global__ __launch_bounds__(256) void decode_clip_kernel(const float4* __restrict__ anchors,
                                                          const float4* __restrict__ deltas, int A, size_t total,
                                                          float img_w, float img_h, float4* __restrict__ out) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    out[i] = decode_clip_one(anchors[i % A], deltas[i], img_w, img_h);
}
__device__ __forceinline__ float4 decode_clip_one(const float4 a, const float4 d, float img_w, float img_h) {
    const float w = a.z - a.x, h = a.w - a.y;
    const float cx = a.x + 0.5f * w, cy = a.y + 0.5f * h;
    const float dx = d.x * 0.1f, dy = d.y * 0.1f, dw = d.z * 0.2f, dh = d.w * 0.2f;
    const float pcx = cx + dx * w, pcy = cy + dy * h;
    const float pw = expf(dw) * w, ph = expf(dh) * h;
    float4 o;
    o.x = fmaxf(pcx - 0.5f * pw, 0.0f);
    o.y = fmaxf(pcy - 0.5f * ph, 0.0f);
    o.z = fminf(pcx + 0.5f * pw, img_w);
    o.w = fminf(pcy + 0.5f * ph, img_h);
    return o;
}

// ---- AMSGrad Adam over a flat buffer (torch.optim.Adam(amsgrad=True, weight_decay) semantics) --------------------
__global__ __launch_bounds__(256) void amsgrad_kernel(float4* __restrict__ p, const float4* __restrict__ g,
                                                      float4* __restrict__ m, float4* __restrict__ v,
                                                      float4* __restrict__ vmax, size_t n4, float lr, float b1,
                                                      float b2, float eps, float wd, float bc1, float bc2_sqrt) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    const float step = lr / bc1;
    for (; i < n4; i += stride) {
        float4 P = p[i], G = g[i], M = m[i], V = v[i], X = vmax[i];
        float* pp = (float*)&P;
        float* gg = (float*)&G;
        float* mm = (float*)&M;
        float* vv = (float*)&V;
        float* xx = (float*)&X;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const float gr = gg[c] + wd * pp[c];
            mm[c] = b1 * mm[c] + (1.0f - b1) * gr;
            vv[c] = b2 * vv[c] + (1.0f - b2) * gr * gr;
            xx[c] = fmaxf(xx[c], vv[c]);
            const float denom = sqrtf(xx[c]) / bc2_sqrt + eps;
            pp[c] = pp[c] - step * (mm[c] / denom);
        }
        p[i] = P;
        m[i] = M;
        v[i] = V;
        vmax[i] = X;
    }
}

}  // namespace

extern "C" {

int ossid_dw_xcorr_fwd(const float* x, int x_planes, const float* k, int planes, int H, int W, float* out,
                       void* stream) {
    if (planes < 0 || H <= 0 || W <= 0 || planes > 65535 * 16 || x_planes <= 0 || x_planes > planes ||
        (planes && planes % x_planes))
        return OSSID_EINVAL;
    if (planes == 0) return OSSID_OK;
    if (!x || !k || !out) return OSSID_EINVAL;
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        dim3 grid((W + DW_TX - 1) / DW_TX, (H + DW_TY - 1) / DW_TY, np);
        hipLaunchKernelGGL(dw_xcorr_kernel, grid, dim3(256), 0, (hipStream_t)stream, x, k + (size_t)p0 * 9, H, W, 0,
                           out + (size_t)p0 * H * W, x_planes, p0);
    }
    return ossid_launch_status();
}

int ossid_dw_xcorr_nhwc_bcast(const float* x, const float* k, int batch, int channels, int H, int W, float* out,
                              void* stream) {
    if (batch < 0 || channels <= 0 || (channels % 4) || H <= 0 || W <= 0) return OSSID_EINVAL;
    if (batch == 0) return OSSID_OK;
    if (!x || !k || !out) return OSSID_EINVAL;
    const int strips = (W + DWX - 1) / DWX;
    const size_t total = (size_t)batch * H * strips * (channels / 4);
    if ((total + 255) / 256 > 0x7fffffffull) return OSSID_EINVAL;
    hipLaunchKernelGGL(dw_xcorr_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, k, channels / 4, H, W, strips, total, (float4*)out);
    return ossid_launch_status();
}

int ossid_dw_xcorr_bwd_x(const float* dout, const float* k, int planes, int H, int W, float* dx, void* stream) {
    if (planes < 0 || H <= 0 || W <= 0 || planes > 65535 * 16) return OSSID_EINVAL;
    if (planes == 0) return OSSID_OK;
    if (!dout || !k || !dx) return OSSID_EINVAL;
    for (int p0 = 0; p0 < planes; p0 += 65535) {
        const int np = planes - p0 < 65535 ? planes - p0 : 65535;
        dim3 grid((W + DW_TX - 1) / DW_TX, (H + DW_TY - 1) / DW_TY, np);
        hipLaunchKernelGGL(dw_xcorr_kernel, grid, dim3(256), 0, (hipStream_t)stream, dout, k + (size_t)p0 * 9, H, W, 1,
                           dx + (size_t)p0 * H * W, planes, p0);
    }
    return ossid_launch_status();
}

int ossid_dw_xcorr_bwd_k(const float* x, const float* dout, int planes, int H, int W, float* dk, void* stream) {
    if (planes < 0 || H <= 0 || W <= 0) return OSSID_EINVAL;
    if (planes == 0) return OSSID_OK;
    if (!x || !dout || !dk) return OSSID_EINVAL;
    hipLaunchKernelGGL(dw_xcorr_bwd_k_kernel, dim3(planes), dim3(256), 0, (hipStream_t)stream, x, dout, H, W, dk);
    return ossid_launch_status();
}

size_t ossid_nms_workspace_bytes(int n) {
    const size_t words = (size_t)(n + 63) / 64;
    return (size_t)n * words * 8 + 256;
}

int ossid_nms(const float* boxes, int n, float iou_threshold, void* workspace, size_t workspace_bytes, int32_t* keep,
              int32_t* num_keep, void* stream) {
    if (n < 0 || n > 16384 || !num_keep) return OSSID_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    if (n == 0) return hipMemsetAsync(num_keep, 0, 4, s) == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
    if (!boxes || !keep || !workspace || workspace_bytes < ossid_nms_workspace_bytes(n)) return OSSID_EINVAL;
    const int words = (n + 63) / 64;
    unsigned long long* mask = (unsigned long long*)workspace;
    zero_async(mask, (((size_t)n * words * 8) + 15) / 16 * 16, s);      // (the workspace has 256 spare bytes)
    hipLaunchKernelGGL(nms_mask_kernel, dim3(words, words), dim3(64), 0, s, (const float4*)boxes, n, iou_threshold,
                       mask, words);
    if (words <= 16) {
        const int lds = n * words * 8;
        OSSID_ENSURE_LDS(nms_scan_small_kernel, (size_t)lds);
        hipLaunchKernelGGL(nms_scan_small_kernel, dim3(1), dim3(256), lds, s, mask, n, words, keep, num_keep);
    } else {
        hipLaunchKernelGGL(nms_scan_kernel, dim3(1), dim3(64), (size_t)words * 8, s, mask, n, words, keep, num_keep);
    }
    return ossid_launch_status();
}

int ossid_bcast_sub_epilogue(const float* S, const float* csub, int templates, int H, int W, int channels,
                             const float* post_scale, const float* post_shift, float* out, int out_channel_stride,
                             int out_channel_offset, void* stream) {
    if (templates < 0 || H < 2 || W < 2 || channels <= 0 || (channels % 4) || (out_channel_stride % 4) ||
        (out_channel_offset % 4) || out_channel_stride < out_channel_offset + channels)
        return OSSID_EINVAL;
    if (templates == 0) return OSSID_OK;
    if (!S || !csub || !out || (post_scale && !post_shift)) return OSSID_EINVAL;
    const size_t total = (size_t)templates * H * W * (channels / 4);
    if ((total + 255) / 256 > 0x7fffffffull) return OSSID_EINVAL;
    hipLaunchKernelGGL(bcast_sub_epilogue_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)S, (const float4*)csub, channels / 4, H, W, total, (const float4*)post_scale,
                       (const float4*)post_shift, out, out_channel_stride, out_channel_offset);
    return ossid_launch_status();
}

int ossid_dot_expand(const float* x, const float* w_cto, int channels, int cout, int H, int W, float* G, void* stream) {
    if (channels <= 0 || cout <= 0 || (cout % 4) || 256 % (cout / 4) || cout / 4 > 256 || H <= 0 || W <= 0 || channels > 65535)
        return OSSID_EINVAL;
    if (!x || !w_cto || !G) return OSSID_EINVAL;
    hipLaunchKernelGGL(dot_expand_kernel, dim3((H * W + 63) / 64, channels), dim3(256), 0, (hipStream_t)stream, x,
                       (const float4*)w_cto, channels, cout / 4, H, W, (float4*)G);
    return ossid_launch_status();
}

int ossid_bias_elu_affine_slice(const float* z, long long rows, int channels, const float* bias, const float* post_scale,
                                const float* post_shift, float* out, int out_channel_stride, int out_channel_offset,
                                void* stream) {
    if (rows < 0 || channels <= 0 || (channels % 4) || (out_channel_stride % 4) || (out_channel_offset % 4) ||
        out_channel_stride < out_channel_offset + channels || (post_scale && !post_shift))
        return OSSID_EINVAL;
    if (rows == 0) return OSSID_OK;
    if (!z || !out) return OSSID_EINVAL;
    const size_t total = (size_t)rows * (channels / 4);
    if ((total + 255) / 256 > 0x7fffffffull) return OSSID_EINVAL;
    hipLaunchKernelGGL(bias_elu_affine_slice_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)z, (const float4*)bias, (const float4*)post_scale, (const float4*)post_shift,
                       channels / 4, (size_t)rows, out, out_channel_stride, out_channel_offset);
    return ossid_launch_status();
}

int ossid_gather_rows(const float* src, int n_src_rows, long long row_floats, const long long* idx, int k,
                      int apply_sigmoid, float* out, void* stream) {
    if (k < 0 || n_src_rows < 0 || row_floats <= 0 || (row_floats % 4) || k > 65535) return OSSID_EINVAL;
    if (k == 0) return OSSID_OK;
    if (!src || !idx || !out || n_src_rows == 0) return OSSID_EINVAL;
    const size_t row4 = (size_t)row_floats / 4;
    unsigned gx = (unsigned)((row4 + 255) / 256);
    if (gx > 1024) gx = 1024;
    hipLaunchKernelGGL(gather_rows_kernel, dim3(gx, k), dim3(256), 0, (hipStream_t)stream, (const float4*)src, row4, idx,
                       apply_sigmoid, (float4*)out);
    return ossid_launch_status();
}

int ossid_decode_clip_boxes(const float* anchors, const float* deltas, int rows, int A, float img_w, float img_h,
                            float* boxes, void* stream) {
    if (rows < 0 || A <= 0) return OSSID_EINVAL;
    if (rows == 0) return OSSID_OK;
    if (!anchors || !deltas || !boxes) return OSSID_EINVAL;
    const size_t total = (size_t)rows * A;
    hipLaunchKernelGGL(decode_clip_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)anchors, (const float4*)deltas, A, total, img_w, img_h, (float4*)boxes);
    return ossid_launch_status();
}

int ossid_amsgrad_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, float* max_exp_avg_sq,
                       size_t n, float lr, float beta1, float beta2, float eps, float weight_decay, int step,
                       void* stream) {
    if (step < 1 || (n & 3)) return OSSID_EINVAL;   // flat buffers are padded to a multiple of 4 floats
    if (n == 0) return OSSID_OK;
    if (!param || !grad || !exp_avg || !exp_avg_sq || !max_exp_avg_sq) return OSSID_EINVAL;
    const double bc1d = 1.0 - pow((double)beta1, (double)step), bc2d = 1.0 - pow((double)beta2, (double)step);
    const float bc1 = (float)bc1d;
    const size_t n4 = n / 4;
    size_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(amsgrad_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (float4*)param,
                       (const float4*)grad, (float4*)exp_avg, (float4*)exp_avg_sq, (float4*)max_exp_avg_sq, n4, lr,
                       beta1, beta2, eps, weight_decay, bc1, (float)sqrt(bc2d));
    return ossid_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// D12  torch.topk(scores, k) over the ~570 k anchor scores of a frame (network.py:555) as a radix SELECT instead of a sort:
// three 11/11/10-bit histogram passes find the exact k-th largest value T, two ordered passes gather the elements > T and
// the lowest-index elements == T that fill up to k (positions by block-prefix sums: no atomics on the output, the result is
// deterministic), one workgroup sorts the k survivors by (value descending, index ascending). Six small launches over
// 2.3 MB instead of a full merge sort.
namespace {

__device__ __forceinline__ unsigned topk_key(float v) {
    const unsigned u = __float_as_uint(v);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);          // unsigned order == float order (NaNs sort as largest)
}

// which bin of `hist` (nbins entries, counts of the keys under the current prefix) holds the rem-th largest key:
// returns the bin, and through `rem` the rank that remains inside it. Every thread of the block returns the same values.
__device__ int topk_find_bin(const int* __restrict__ hist, int nbins, int& rem, int* lds /* 257 ints */) {
    const int t = threadIdx.x, per = nbins / 256;               // blockDim.x == 256, nbins a multiple of 256
    int seg = 0;
    for (int j = 0; j < per; ++j) seg += hist[nbins - 1 - (t * per + j)];      // thread t owns the t-th segment FROM THE TOP
    // inclusive scan of the 256 segment sums (round 3 walked them with one thread: 256 dependent LDS reads, ~8 us per call
    // and six calls per top-k); the one segment whose range (excl, incl] holds `rem` reports itself
    const int lane = t & 63, wave = t >> 6;
    int incl = seg;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) lds[wave] = incl;
    __syncthreads();
    for (int w = 0; w < wave; ++w) incl += lds[w];
    const int excl = incl - seg;
    __syncthreads();
    if (excl < rem && rem <= incl) lds[256] = t, lds[0] = excl;   // (keys above the segment)
    __syncthreads();
    const int segi = lds[256];
    int run = lds[0];
    __syncthreads();
    int bin = nbins - 1 - segi * per;
    for (int j = 0; j < per; ++j, --bin) {
        const int c = hist[bin];
        if (run + c >= rem) break;
        run += c;
    }
    rem -= run;
    return bin;
}

struct TopkState {
    unsigned prefix, mask;      // keys of interest: (key & mask) == prefix
    int rem;
};

__device__ TopkState topk_state(const int* __restrict__ hist, int level, int k, int* lds) {
    TopkState s;
    s.prefix = 0, s.mask = 0, s.rem = k;
    if (level >= 2) {
        const int b = topk_find_bin(hist, 2048, s.rem, lds);
        s.prefix = (unsigned)b << 21, s.mask = 0xFFE00000u;
    }
    if (level >= 3) {
        const int b = topk_find_bin(hist + 2048, 2048, s.rem, lds);
        s.prefix |= (unsigned)b << 10, s.mask = 0xFFFFFC00u;
    }
    if (level >= 4) {
        const int b = topk_find_bin(hist + 4096, 1024, s.rem, lds);
        s.prefix |= (unsigned)b, s.mask = 0xFFFFFFFFu;
    }
    return s;
}

__global__ __launch_bounds__(256) void topk_hist_kernel(const float* __restrict__ x, int xs, int n, int level, int k, int* __restrict__ hist) {
    __shared__ int lh[2048];
    __shared__ int scratch[257];
    const TopkState s = topk_state(hist, level, k, scratch);
    for (int i = threadIdx.x; i < 2048; i += 256) lh[i] = 0;
    __syncthreads();
    const int shift = level == 1 ? 21 : (level == 2 ? 10 : 0), bmask = level == 3 ? 1023 : 2047;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const unsigned key = topk_key(x[(size_t)i * xs]);
        if ((key & s.mask) == s.prefix) atomicAdd(&lh[(key >> shift) & bmask], 1);
    }
    __syncthreads();
    int* out = hist + (level - 1) * 2048;
    for (int i = threadIdx.x; i < 2048; i += 256)
        if (lh[i]) atomicAdd(&out[i], lh[i]);
}

// pass 1 (write == 0): per-block counts of (key > T, key == T) over the block's CONTIGUOUS chunk -> counts[block][2].
// pass 2 (write == 1): positions from the counts of the blocks before; greater elements first (k_gt of them), then ties.
__global__ __launch_bounds__(256) void topk_gather_kernel(const float* __restrict__ x, int xs, int n, int k, const int* __restrict__ hist,
                                                          int* __restrict__ counts, int write, float* __restrict__ sel_val,
                                                          int* __restrict__ sel_idx) {
    __shared__ int scratch[257];
    __shared__ int wsum[4][2];
    const TopkState s = topk_state(hist, 4, k, scratch);
    const unsigned T = s.prefix;
    const int need_ties = s.rem, k_gt = k - need_ties;
    const int chunk = (n + gridDim.x - 1) / gridDim.x, lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    int base_gt = 0, base_eq = 0;
    if (write) {
        for (int bq = threadIdx.x; bq < (int)blockIdx.x; bq += 256) base_gt += counts[2 * bq], base_eq += counts[2 * bq + 1];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) base_gt += __shfl_xor(base_gt, m), base_eq += __shfl_xor(base_eq, m);
        if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6][0] = base_gt, wsum[threadIdx.x >> 6][1] = base_eq;
        __syncthreads();
        base_gt = wsum[0][0] + wsum[1][0] + wsum[2][0] + wsum[3][0];
        base_eq = wsum[0][1] + wsum[1][1] + wsum[2][1] + wsum[3][1];
        __syncthreads();
    }
    int run_gt = 0, run_eq = 0;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i0 = lo; i0 < hi; i0 += 256) {
        const int i = i0 + threadIdx.x;
        const bool in = i < hi;
        const float v = in ? x[(size_t)i * xs] : 0.0f;
        const unsigned key = topk_key(v);
        const bool gt = in && key > T, eq = in && key == T;
        const unsigned long long mg = __ballot(gt), me = __ballot(eq);
        if (lane == 0) wsum[wave][0] = __popcll(mg), wsum[wave][1] = __popcll(me);
        __syncthreads();
        int pre_gt = 0, pre_eq = 0, tot_gt = 0, tot_eq = 0;
#pragma unroll
        for (int w = 0; w < 4; ++w) {
            if (w < wave) pre_gt += wsum[w][0], pre_eq += wsum[w][1];
            tot_gt += wsum[w][0], tot_eq += wsum[w][1];
        }
        if (write) {
            const unsigned long long below = lane ? (~0ull >> (64 - lane)) : 0ull;
            if (gt) {
                const int pos = base_gt + run_gt + pre_gt + __popcll(mg & below);
                sel_val[pos] = v, sel_idx[pos] = i;
            } else if (eq) {
                const int rank = base_eq + run_eq + pre_eq + __popcll(me & below);       // index order
                if (rank < need_ties) sel_val[k_gt + rank] = v, sel_idx[k_gt + rank] = i;
            }
        }
        run_gt += tot_gt, run_eq += tot_eq;
        __syncthreads();
    }
    if (!write && threadIdx.x == 0) counts[2 * blockIdx.x] = run_gt, counts[2 * blockIdx.x + 1] = run_eq;
}

// one workgroup: bitonic sort of the k (<= 2048) selected (value, index) pairs, value descending, index ascending
// boxes != NULL: also decode + clip the box of every survivor (anchors [A][4], deltas [rows][A][4], element i = row i / A):
// the arithmetic of decode_clip_kernel on k boxes instead of all of them.
__global__ __launch_bounds__(1024) void topk_sort_kernel(const float* __restrict__ sel_val, const int* __restrict__ sel_idx, int k,
                                                         float* __restrict__ out_val, long long* __restrict__ out_idx,
                                                         const float4* __restrict__ anchors, const float4* __restrict__ deltas, int A,
                                                         float img_w, float img_h, float4* __restrict__ boxes) {
    __shared__ unsigned long long keys[2048];
    for (int i = threadIdx.x; i < 2048; i += 1024)
        keys[i] = i < k ? (((unsigned long long)topk_key(sel_val[i]) << 32) | (unsigned)(~sel_idx[i])) : 0ull;
    __syncthreads();
    for (int size = 2; size <= 2048; size <<= 1)
        for (int stride = size >> 1; stride > 0; stride >>= 1) {
            const int t = threadIdx.x;                      // 1024 compare-exchange pairs per step
            const int i = ((t / stride) * stride * 2) + (t % stride), j = i + stride;
            const bool desc = ((i & size) == 0);
            const unsigned long long a = keys[i], b = keys[j];
            if ((a < b) == desc) keys[i] = b, keys[j] = a;
            __syncthreads();
        }
    for (int i = threadIdx.x; i < k; i += 1024) {
        const unsigned long long kk = keys[i];
        const unsigned hi = (unsigned)(kk >> 32);
        const unsigned u = (hi & 0x80000000u) ? (hi & 0x7FFFFFFFu) : ~hi;
        out_val[i] = __uint_as_float(u);
        const unsigned id = ~(unsigned)(kk & 0xFFFFFFFFu);
        out_idx[i] = (long long)id;
        if (boxes) boxes[i] = decode_clip_one(anchors[id % A], deltas[id], img_w, img_h);
    }
}

// The detection list of a frame (network.py:566-581) from the sorted candidates and the NMS keep list, ONE launch: row j of
// every output belongs to candidate keep[j] -- its score, its box, the index of the template that fired (candidate index / A),
// that template's segmentation map (optionally through the sigmoid of models/dtoid/__init__.py:147) and heat map.
__global__ __launch_bounds__(256) void detect_emit_kernel(const float* __restrict__ scores, const long long* __restrict__ indices,
                                                          const float4* __restrict__ boxes, const int* __restrict__ keep, int A,
                                                          const float* __restrict__ seg, long long seg_row, const float* __restrict__ heat,
                                                          long long heat_row, int seg_sigmoid, float* __restrict__ o_scores,
                                                          float4* __restrict__ o_boxes, float* __restrict__ o_obj, float* __restrict__ o_seg,
                                                          float* __restrict__ o_heat) {
    const int j = blockIdx.y, c = keep[j];
    const long long tmpl = indices[c] / A;
    if (blockIdx.x == 0 && threadIdx.x == 0) o_scores[j] = scores[c], o_boxes[j] = boxes[c], o_obj[j] = (float)tmpl;
    const size_t t0 = (size_t)blockIdx.x * 256 + threadIdx.x, step = (size_t)gridDim.x * 256;
    if (seg_row) {
        const float* s = seg + (size_t)tmpl * seg_row;
        float* o = o_seg + (size_t)j * seg_row;
        auto sg = [&](float v) { return seg_sigmoid ? 1.0f / (1.0f + expf(-v)) : v; };
        if ((seg_row & 3) == 0) {
            for (size_t i = t0; i < (size_t)seg_row / 4; i += step) {
                float4 v = ((const float4*)s)[i];
                v.x = sg(v.x), v.y = sg(v.y), v.z = sg(v.z), v.w = sg(v.w);
                ((float4*)o)[i] = v;
            }
        } else {
            for (size_t i = t0; i < (size_t)seg_row; i += step) o[i] = sg(s[i]);
        }
    }
    if (heat_row) {
        const float* s = heat + (size_t)tmpl * heat_row;
        float* o = o_heat + (size_t)j * heat_row;
        for (size_t i = t0; i < (size_t)heat_row; i += step) o[i] = s[i];
    }
}

}  // namespace

extern "C" {

size_t ossid_topk_workspace_bytes(int n, int k) {
    (void)n;
    // histograms (3 levels) + per-block counts (512 blocks x 2) + selected values / indices
    return (size_t)(3 * 2048 + 2 * 512) * sizeof(int) + (size_t)k * (sizeof(float) + sizeof(int)) + 64;
}

static int topk_launch(const float* scores, int stride, int n, int k, void* workspace, float* values, long long* indices,
                       const float* anchors, const float* deltas, int A, float img_w, float img_h, float* boxes, hipStream_t s) {
    int* hist = (int*)workspace;
    int* counts = hist + 3 * 2048;
    float* sel_val = (float*)(counts + 2 * 512);
    int* sel_idx = (int*)(sel_val + k);
    zero_async(hist, 3 * 2048 * sizeof(int), s);
    int blocks = (n + 2047) / 2048;
    if (blocks > 512) blocks = 512;
    for (int level = 1; level <= 3; ++level)
        hipLaunchKernelGGL(topk_hist_kernel, dim3(blocks), dim3(256), 0, s, scores, stride, n, level, k, hist);
    hipLaunchKernelGGL(topk_gather_kernel, dim3(blocks), dim3(256), 0, s, scores, stride, n, k, (const int*)hist, counts, 0, sel_val, sel_idx);
    hipLaunchKernelGGL(topk_gather_kernel, dim3(blocks), dim3(256), 0, s, scores, stride, n, k, (const int*)hist, counts, 1, sel_val, sel_idx);
    hipLaunchKernelGGL(topk_sort_kernel, dim3(1), dim3(1024), 0, s, (const float*)sel_val, (const int*)sel_idx, k, values, indices,
                       (const float4*)anchors, (const float4*)deltas, A, img_w, img_h, (float4*)boxes);
    return ossid_launch_status();
}

int ossid_topk(const float* scores, int n, int k, void* workspace, size_t workspace_bytes, float* values, long long* indices,
               void* stream) {
    if (!scores || !workspace || !values || !indices || n <= 0 || k <= 0 || k > n || k > 2048) return OSSID_EINVAL;
    if (workspace_bytes < ossid_topk_workspace_bytes(n, k)) return OSSID_EINVAL;
    return topk_launch(scores, 1, n, k, workspace, values, indices, nullptr, nullptr, 1, 0.f, 0.f, nullptr, (hipStream_t)stream);
}

size_t ossid_detect_post_workspace_bytes(int n, int k) {
    return ((ossid_topk_workspace_bytes(n, k) + 255) / 256) * 256 + ossid_nms_workspace_bytes(k);
}

int ossid_detect_post(const float* scores, int n, int score_stride, int k, const float* anchors, const float* deltas, int A,
                      float img_w, float img_h, float iou_threshold, void* workspace, size_t workspace_bytes, float* out_scores,
                      long long* out_indices, float* out_boxes, int32_t* keep, int32_t* num_keep, void* stream) {
    if (!scores || !anchors || !deltas || !workspace || !out_scores || !out_indices || !out_boxes || !keep || !num_keep) return OSSID_EINVAL;
    if (n <= 0 || k <= 0 || k > n || k > 2048 || A <= 0 || score_stride <= 0 || (n % A)) return OSSID_EINVAL;
    if (workspace_bytes < ossid_detect_post_workspace_bytes(n, k)) return OSSID_EINVAL;
    const int rc = topk_launch(scores, score_stride, n, k, workspace, out_scores, out_indices, anchors, deltas, A, img_w, img_h,
                               out_boxes, (hipStream_t)stream);
    if (rc != OSSID_OK) return rc;
    char* nms_ws = (char*)workspace + ((ossid_topk_workspace_bytes(n, k) + 255) / 256) * 256;
    return ossid_nms(out_boxes, k, iou_threshold, nms_ws, ossid_nms_workspace_bytes(k), keep, num_keep, stream);
}

int ossid_detect_emit(const float* scores, const long long* indices, const float* boxes, const int32_t* keep, int count, int A,
                      const float* seg, long long seg_row_floats, const float* heat, long long heat_row_floats, int seg_sigmoid,
                      float* out_scores, float* out_boxes, float* out_obj, float* out_seg, float* out_heat, void* stream) {
    if (count < 0 || count > 65535 || A <= 0 || seg_row_floats < 0 || heat_row_floats < 0) return OSSID_EINVAL;
    if (count == 0) return OSSID_OK;
    if (!scores || !indices || !boxes || !keep || !out_scores || !out_boxes || !out_obj) return OSSID_EINVAL;
    if ((seg_row_floats && (!seg || !out_seg)) || (heat_row_floats && (!heat || !out_heat))) return OSSID_EINVAL;
    const long long most = seg_row_floats / 4 > heat_row_floats ? seg_row_floats / 4 : heat_row_floats;
    unsigned gx = (unsigned)((most + 255) / 256);
    if (gx < 1) gx = 1;
    if (gx > 512) gx = 512;
    hipLaunchKernelGGL(detect_emit_kernel, dim3(gx, count), dim3(256), 0, (hipStream_t)stream, scores, indices, (const float4*)boxes, keep,
                       A, seg, seg_row_floats, heat, heat_row_floats, seg_sigmoid, out_scores, (float4*)out_boxes, out_obj, out_seg,
                       out_heat);
    return ossid_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// D1-D4  the strided stems of the two backbones and the pooling around them, channels-last.
//   im2col_stem   the 7x7 / stride 2 / pad 3 stem of DenseNet-121 (network.py:164-169, 3 -> 64) and the 3x3 / stride 2 /
//                 no-pad stem of SqueezeNet-1.1 (:203-208, 4 -> 64) have 3 / 4 input channels: too few for the MFMA
//                 convolution's channel tiling. Their receptive fields are gathered into rows [out pixel][k*k*Cin padded
//                 to a multiple of 16] -- with normalizeImageRange ((x - mean) / std, utils/__init__.py:33-39) applied to
//                 real pixels on the way, zero padding in NORMALISED space as the reference has it -- and the stem
//                 becomes a 1x1 convolution on csrc/conv.hip (weights re-laid to match by the host).
//   stem_tail     x0 + conv2d_dw_group(x0, k) (network.py:178-179, 186-192), eval BatchNorm affine, ReLU -- one pass.
//   maxpool_nhwc  nn.MaxPool2d(k, stride, padding, ceil_mode) (DenseNet pool0: 3/2/1; SqueezeNet: 3/2/0 ceil).
namespace {

__global__ __launch_bounds__(256) void im2col_stem_kernel(const float* __restrict__ img, int Cin, int H, int W, int k, int stride,
                                                          int pad, int Ho, int Wo, int Kpad, const float* __restrict__ mean,
                                                          const float* __restrict__ inv_std, size_t total,
                                                          float4* __restrict__ out) {
    // one thread = 4 consecutive columns of one output row; column j = (ky * k + kx) * Cin + ci
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int K4 = Kpad / 4;
    const int q = (int)(i % K4);
    size_t r = i / K4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const int j = 4 * q + e;
        float val = 0.0f;
        if (j < k * k * Cin) {
            const int ci = j % Cin, t = j / Cin, kx = t % k, ky = t / k;
            const int y = yo * stride - pad + ky, x = xo * stride - pad + kx;
            if (y >= 0 && y < H && x >= 0 && x < W) {
                val = img[(((size_t)b * Cin + ci) * H + y) * W + x];                 // NCHW input, as the caller has it
                if (mean) val = (val - mean[ci]) * inv_std[ci];
            }
        }
        v[e] = val;
    }
    out[i] = make_float4(v[0], v[1], v[2], v[3]);
}

__global__ __launch_bounds__(256) void stem_tail_kernel(const float4* __restrict__ x0, const float* __restrict__ kern,
                                                        int kern_bs, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, int H, int W, int C4, size_t total,
                                                        float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xx = (int)(r % W);
    r /= W;
    const int yy = (int)(r % H), b = (int)(r / H);
    const float* kk = kern + (size_t)b * kern_bs + (size_t)c4 * 36;      // [C][3][3]: 4 channels x 9 taps
    float4 acc = x0[i];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = yy + dy - 1;
        if (y < 0 || y >= H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int x = xx + dx - 1;
            if (x < 0 || x >= W) continue;
            const float4 v = x0[(((size_t)b * H + y) * W + x) * C4 + c4];
            const int t = dy * 3 + dx;
            acc.x = fmaf(v.x, kk[t], acc.x), acc.y = fmaf(v.y, kk[9 + t], acc.y);
            acc.z = fmaf(v.z, kk[18 + t], acc.z), acc.w = fmaf(v.w, kk[27 + t], acc.w);
        }
    }
    const float4 sc = *(const float4*)(scale + 4 * c4), sh = *(const float4*)(shift + 4 * c4);
    out[i] = make_float4(fmaxf(acc.x * sc.x + sh.x, 0.f), fmaxf(acc.y * sc.y + sh.y, 0.f), fmaxf(acc.z * sc.z + sh.z, 0.f),
                         fmaxf(acc.w * sc.w + sh.w, 0.f));
}

// Test time, one pass: relu(norm0(x0 + dw(x0, k))) and pool0 = MaxPool2d(3, 2, 1) behind it (network.py:170-179), without the
// full-resolution tensor in between (19.7 MB written and re-read at 480 x 640: stem_tail 48 us + max-pool 8 us + a copy into the
// dense block's buffer 6 us). A thread owns 4 channels of TWO neighbouring pooled pixels of a row: the 5 x 7 window of x0 under
// them is loaded in one go (35 independent 16-byte reads), the 3 x 5 modulated values are formed once and shared by the two
// pooling windows; the result goes straight into the first channels of the block buffer (out_cs floats per pixel).
__global__ __launch_bounds__(256) void stem_tail_pool_kernel(const float4* __restrict__ x0, const float* __restrict__ kern, int kern_bs,
                                                             const float* __restrict__ scale, const float* __restrict__ shift,
                                                             int H, int W, int C4, int Ho, int Wo, int strips, size_t total,
                                                             float* __restrict__ out, int out_cs) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xo0 = (int)(r % strips) * 2;
    r /= strips;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    const float* kk = kern + (size_t)b * kern_bs + (size_t)c4 * 36;          // 4 channels x 9 taps
    float w[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) w[t] = kk[t];
    const float4 sc = *(const float4*)(scale + 4 * c4), sh = *(const float4*)(shift + 4 * c4);
    const int ya = 2 * yo - 2, xa = 2 * xo0 - 2;                             // window origin in x0
    float4 v[5][7];
#pragma unroll
    for (int dy = 0; dy < 5; ++dy) {
        const int y = ya + dy;
        const bool rv = y >= 0 && y < H;
        const int yc = min(max(y, 0), H - 1);
#pragma unroll
        for (int dx = 0; dx < 7; ++dx) {
            const int x = xa + dx;
            const float4 q = x0[(((size_t)b * H + yc) * W + min(max(x, 0), W - 1)) * C4 + c4];
            const float f = (rv && x >= 0 && x < W) ? 1.0f : 0.0f;         // zero padding of the depthwise convolution
            v[dy][dx] = make_float4(f * q.x, f * q.y, f * q.z, f * q.w);
        }
    }
    float4 best[2];
    best[0] = best[1] = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
#pragma unroll
    for (int my = 0; my < 3; ++my) {                                         // modulated pixel (ya + 1 + my, xa + 1 + mx)
        const int y = ya + 1 + my;
        if (y < 0 || y >= H) continue;                                       // max-pool padding: not a candidate
#pragma unroll
        for (int mx = 0; mx < 5; ++mx) {
            const int x = xa + 1 + mx;
            if (x < 0 || x >= W) continue;
            float4 acc = v[my + 1][mx + 1];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const float4 q = v[my + dy][mx + dx];
                    const int t = dy * 3 + dx;
                    acc.x = fmaf(q.x, w[t], acc.x), acc.y = fmaf(q.y, w[9 + t], acc.y);
                    acc.z = fmaf(q.z, w[18 + t], acc.z), acc.w = fmaf(q.w, w[27 + t], acc.w);
                }
            const float4 a = make_float4(fmaxf(acc.x * sc.x + sh.x, 0.f), fmaxf(acc.y * sc.y + sh.y, 0.f),
                                         fmaxf(acc.z * sc.z + sh.z, 0.f), fmaxf(acc.w * sc.w + sh.w, 0.f));
#pragma unroll
            for (int o = 0; o < 2; ++o)
                if (mx >= 2 * o && mx <= 2 * o + 2)
                    best[o] = make_float4(fmaxf(best[o].x, a.x), fmaxf(best[o].y, a.y), fmaxf(best[o].z, a.z), fmaxf(best[o].w, a.w));
        }
    }
#pragma unroll
    for (int o = 0; o < 2; ++o)
        if (xo0 + o < Wo) *(float4*)(out + (((size_t)b * Ho + yo) * Wo + xo0 + o) * out_cs + 4 * c4) = best[o];
}

// relu(scale * x + shift) averaged over 2 x 2 windows (stride 1 or 2): the front of a DenseNet transition at test time with the
// pool moved IN FRONT of the 1x1 convolution (norm -> relu -> conv -> pool of network.py:164-184's densenet121; a bias-free 1x1
// convolution and an average commute), so that the convolution runs on a quarter of the pixels. x [B][H][W][in_cs] (first C
// channels), out [B][Ho][Wo][C].
__global__ __launch_bounds__(256) void bn_relu_avgpool2_kernel(const float* __restrict__ x, int in_cs, const float* __restrict__ scale,
                                                               const float* __restrict__ shift, int H, int W, int C4, int stride,
                                                               int Ho, int Wo, size_t total, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    const float4 sc = *(const float4*)(scale + 4 * c4), sh = *(const float4*)(shift + 4 * c4);
    float4 q[4];
#pragma unroll
    for (int e = 0; e < 4; ++e)
        q[e] = *(const float4*)(x + (((size_t)b * H + yo * stride + (e >> 1)) * W + xo * stride + (e & 1)) * in_cs + 4 * c4);
    float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        s.x += fmaxf(q[e].x * sc.x + sh.x, 0.f), s.y += fmaxf(q[e].y * sc.y + sh.y, 0.f);
        s.z += fmaxf(q[e].z * sc.z + sh.z, 0.f), s.w += fmaxf(q[e].w * sc.w + sh.w, 0.f);
    }
    out[i] = make_float4(0.25f * s.x, 0.25f * s.y, 0.25f * s.z, 0.25f * s.w);
}

__global__ __launch_bounds__(256) void maxpool_nhwc_kernel(const float4* __restrict__ x, int H, int W, int C4, int k, int stride,
                                                           int pad, int Ho, int Wo, size_t total, float4* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    float4 m = make_float4(-INFINITY, -INFINITY, -INFINITY, -INFINITY);
    for (int dy = 0; dy < k; ++dy) {
        const int y = yo * stride - pad + dy;
        if (y < 0 || y >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = xo * stride - pad + dx;
            if (xx < 0 || xx >= W) continue;
            const float4 v = x[(((size_t)b * H + y) * W + xx) * C4 + c4];
            m.x = fmaxf(m.x, v.x), m.y = fmaxf(m.y, v.y), m.z = fmaxf(m.z, v.z), m.w = fmaxf(m.w, v.w);
        }
    }
    out[i] = m;
}

}  // namespace

extern "C" {

int ossid_im2col_stem(const float* img_nchw, int B, int Cin, int H, int W, int k, int stride, int pad, int Kpad,
                      const float* mean, const float* inv_std, float* out, void* stream) {
    if (!img_nchw || !out || B <= 0 || Cin <= 0 || H <= 0 || W <= 0 || k <= 0 || stride <= 0 || pad < 0 || Kpad % 4 ||
        Kpad < k * k * Cin || (!mean != !inv_std))
        return OSSID_EINVAL;
    const int Ho = (H + 2 * pad - k) / stride + 1, Wo = (W + 2 * pad - k) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return OSSID_EINVAL;
    const size_t total = (size_t)B * Ho * Wo * (Kpad / 4);
    hipLaunchKernelGGL(im2col_stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, img_nchw, Cin,
                       H, W, k, stride, pad, Ho, Wo, Kpad, mean, inv_std, total, (float4*)out);
    return ossid_launch_status();
}

int ossid_stem_tail_nhwc(const float* x0, const float* kernels, int kernels_batch_stride, const float* scale, const float* shift,
                         int B, int H, int W, int C, float* out, void* stream) {
    if (!x0 || !kernels || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || kernels_batch_stride < 0)
        return OSSID_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(stem_tail_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x0, kernels, kernels_batch_stride, scale, shift, H, W, C / 4, total, (float4*)out);
    return ossid_launch_status();
}

int ossid_stem_tail_pool_nhwc(const float* x0, const float* kernels, int kernels_batch_stride, const float* scale, const float* shift,
                              int B, int H, int W, int C, float* out, int out_channel_stride, void* stream) {
    if (!x0 || !kernels || !scale || !shift || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || kernels_batch_stride < 0 ||
        out_channel_stride < C || out_channel_stride % 4)
        return OSSID_EINVAL;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1, strips = (Wo + 1) / 2;
    const size_t total = (size_t)B * Ho * strips * (C / 4);
    hipLaunchKernelGGL(stem_tail_pool_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x0, kernels, kernels_batch_stride, scale, shift, H, W, C / 4, Ho, Wo, strips, total, out,
                       out_channel_stride);
    return ossid_launch_status();
}

int ossid_bn_relu_avgpool2_nhwc(const float* x, int B, int H, int W, int C, int in_channel_stride, const float* scale,
                                const float* shift, int stride, float* out, void* stream) {
    if (!x || !scale || !shift || !out || B <= 0 || H < 2 || W < 2 || C <= 0 || C % 4 || in_channel_stride < C ||
        in_channel_stride % 4 || (stride != 1 && stride != 2))
        return OSSID_EINVAL;
    const int Ho = (H - 2) / stride + 1, Wo = (W - 2) / stride + 1;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(bn_relu_avgpool2_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x,
                       in_channel_stride, scale, shift, H, W, C / 4, stride, Ho, Wo, total, (float4*)out);
    return ossid_launch_status();
}

int ossid_maxpool_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, int ceil_mode, float* out,
                       void* stream) {
    if (!x || !out || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || k <= 0 || stride <= 0 || pad < 0 || 2 * pad > k)
        return OSSID_EINVAL;
    auto osz = [&](int n) {
        int o = ceil_mode ? (n + 2 * pad - k + stride - 1) / stride + 1 : (n + 2 * pad - k) / stride + 1;
        if (ceil_mode && (o - 1) * stride >= n + pad) --o;          // torch: the last window must start inside the input
        return o;
    };
    const int Ho = osz(H), Wo = osz(W);
    if (Ho <= 0 || Wo <= 0) return OSSID_EINVAL;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, H, W, C / 4, k, stride, pad, Ho, Wo, total, (float4*)out);
    return ossid_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// Training-side companions of the stem kernels above (the finetune step, channels-last):
//   (x + conv2d_dw_group(x, k), its two gradients and the fused norm0 / pool0 passes live in csrc/stem.hip)
//   maxpool with argmax index (uint8, position inside the window) and its backward as a gather over the <= ceil(k/s)^2
//   windows that contain an input pixel (no atomics)
namespace {

__global__ __launch_bounds__(256) void maxpool_idx_nhwc_kernel(const float4* __restrict__ x, int H, int W, int C4, int k, int stride,
                                                               int pad, int Ho, int Wo, size_t total, float4* __restrict__ out,
                                                               uchar4* __restrict__ idx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xo = (int)(r % Wo);
    r /= Wo;
    const int yo = (int)(r % Ho), b = (int)(r / Ho);
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned char a[4] = {255, 255, 255, 255};
    for (int dy = 0; dy < k; ++dy) {
        const int y = yo * stride - pad + dy;
        if (y < 0 || y >= H) continue;
        for (int dx = 0; dx < k; ++dx) {
            const int xx = xo * stride - pad + dx;
            if (xx < 0 || xx >= W) continue;
            const float4 v4 = x[(((size_t)b * H + y) * W + xx) * C4 + c4];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (v[e] > m[e] || a[e] == 255) m[e] = v[e], a[e] = (unsigned char)(dy * k + dx);   // first maximum wins
        }
    }
    out[i] = make_float4(m[0], m[1], m[2], m[3]);
    idx[i] = make_uchar4(a[0], a[1], a[2], a[3]);
}

__global__ __launch_bounds__(256) void maxpool_bwd_nhwc_kernel(const float4* __restrict__ dout, const uchar4* __restrict__ idx, int H,
                                                               int W, int C4, int k, int stride, int pad, int Ho, int Wo,
                                                               size_t total, float4* __restrict__ dx) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int X = (int)(r % W);
    r /= W;
    const int Y = (int)(r % H), b = (int)(r / H);
    float s[4] = {0.f, 0.f, 0.f, 0.f};
    // windows (yo, xo) that contain (Y, X): yo*stride - pad <= Y < yo*stride - pad + k
    const int yo_hi = min(Ho - 1, (Y + pad) / stride), xo_hi = min(Wo - 1, (X + pad) / stride);
    for (int yo = yo_hi; yo >= 0 && yo * stride - pad + k > Y; --yo)
        for (int xo = xo_hi; xo >= 0 && xo * stride - pad + k > X; --xo) {
            const unsigned char pos = (unsigned char)((Y - (yo * stride - pad)) * k + (X - (xo * stride - pad)));
            const size_t o = (((size_t)b * Ho + yo) * Wo + xo) * C4 + c4;
            const uchar4 a = idx[o];
            const float4 g = dout[o];
            if (a.x == pos) s[0] += g.x;
            if (a.y == pos) s[1] += g.y;
            if (a.z == pos) s[2] += g.z;
            if (a.w == pos) s[3] += g.w;
        }
    dx[i] = make_float4(s[0], s[1], s[2], s[3]);
}

// Separable linear resampling of a channels-last image by per-output-row / per-output-column TAP TABLES:
//   out[b][oy][ox][c] (+)= sum_i sum_j wy[oy][i] * wx[ox][j] * x[b][iy[oy][i]][ix[ox][j]][c]
// with T taps per output index (index -1 = unused). One kernel is bilinear resizing (T = 2: F.interpolate's neighbours and
// weights), its adjoint (the transposed tables), cropping (T = 1, weight 1: the valid 3x3 convolutions of the template
// encoders are padded convolutions whose interior is kept) and zero-padding back (the crop's adjoint).
__global__ __launch_bounds__(256) void resample_taps_kernel(const float4* __restrict__ x, int Hin, int Win, int C4, int x_cs4,
                                                            int Hout, int Wout, const int* __restrict__ ty_idx,
                                                            const float* __restrict__ ty_w, const int* __restrict__ tx_idx,
                                                            const float* __restrict__ tx_w, int T, size_t total,
                                                            float4* __restrict__ out, int out_cs4, int out_coff4) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i % C4);
    size_t r = i / C4;
    const int ox = (int)(r % Wout);
    r /= Wout;
    const int oy = (int)(r % Hout);
    const int b = (int)(r / Hout);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int a = 0; a < T; ++a) {
        const int iy = ty_idx[oy * T + a];
        if (iy < 0) continue;
        const float wy = ty_w[oy * T + a];
        for (int e = 0; e < T; ++e) {
            const int ix = tx_idx[ox * T + e];
            if (ix < 0) continue;
            const float w = wy * tx_w[ox * T + e];
            const float4 v = x[(((size_t)b * Hin + iy) * Win + ix) * x_cs4 + c4];
            acc[0] += w * v.x, acc[1] += w * v.y, acc[2] += w * v.z, acc[3] += w * v.w;
        }
    }
    out[(((size_t)b * Hout + oy) * Wout + ox) * out_cs4 + out_coff4 + c4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
}

}  // namespace

extern "C" {

int ossid_resample_taps_nhwc(const float* x, int B, int Hin, int Win, int C, int x_channel_stride, int Hout, int Wout,
                             const int32_t* taps_y_idx, const float* taps_y_w, const int32_t* taps_x_idx, const float* taps_x_w,
                             int T, float* out, int out_channel_stride, int out_channel_offset, void* stream) {
    if (!x || !out || !taps_y_idx || !taps_y_w || !taps_x_idx || !taps_x_w) return OSSID_EINVAL;
    if (B <= 0 || Hin <= 0 || Win <= 0 || Hout <= 0 || Wout <= 0 || C <= 0 || C % 4 || T <= 0 || T > 8) return OSSID_EINVAL;
    const int xcs = x_channel_stride > 0 ? x_channel_stride : C, ocs = out_channel_stride > 0 ? out_channel_stride : C;
    if (xcs % 4 || ocs % 4 || out_channel_offset % 4 || out_channel_offset < 0 || out_channel_offset + C > ocs || xcs < C)
        return OSSID_EINVAL;
    if (((uintptr_t)x & 15) || ((uintptr_t)out & 15)) return OSSID_EINVAL;
    const size_t total = (size_t)B * Hout * Wout * (C / 4);
    hipLaunchKernelGGL(resample_taps_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, Hin, Win, C / 4, xcs / 4, Hout, Wout, taps_y_idx, taps_y_w, taps_x_idx, taps_x_w, T, total,
                       (float4*)out, ocs / 4, out_channel_offset / 4);
    return ossid_launch_status();
}

int ossid_maxpool_idx_nhwc(const float* x, int B, int H, int W, int C, int k, int stride, int pad, int ceil_mode, float* out,
                           uint8_t* argmax, void* stream) {
    if (!x || !out || !argmax || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || k <= 0 || k > 15 || stride <= 0 || pad < 0 ||
        2 * pad > k)
        return OSSID_EINVAL;
    auto osz = [&](int n) {
        int o = ceil_mode ? (n + 2 * pad - k + stride - 1) / stride + 1 : (n + 2 * pad - k) / stride + 1;
        if (ceil_mode && (o - 1) * stride >= n + pad) --o;
        return o;
    };
    const int Ho = osz(H), Wo = osz(W);
    if (Ho <= 0 || Wo <= 0) return OSSID_EINVAL;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    hipLaunchKernelGGL(maxpool_idx_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)x, H, W, C / 4, k, stride, pad, Ho, Wo, total, (float4*)out, (uchar4*)argmax);
    return ossid_launch_status();
}

int ossid_maxpool_bwd_nhwc(const float* dout, const uint8_t* argmax, int B, int H, int W, int C, int k, int stride, int pad, int Ho,
                           int Wo, float* dx, void* stream) {
    if (!dout || !argmax || !dx || B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4 || k <= 0 || stride <= 0 || Ho <= 0 || Wo <= 0)
        return OSSID_EINVAL;
    const size_t total = (size_t)B * H * W * (C / 4);
    hipLaunchKernelGGL(maxpool_bwd_nhwc_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                       (const float4*)dout, (const uchar4*)argmax, H, W, C / 4, k, stride, pad, Ho, Wo, total, (float4*)dx);
    return ossid_launch_status();
}

}  // extern "C"

// =====================================================================================================================
// D6  the last library calls of the correlation head, as own kernels (deterministic: fixed-order sums):
//   conv1x1_c1     nn.Conv2d(C, 1, 1) on a channels-last x [rows][C] -- `corr_conv_heatmap` 512 -> 1 (network.py:334, :349) --
//                  with the sigmoid of `heat_map = torch.sigmoid(...)` optionally fused; backward: dx = g * w, and per-
//                  workgroup partial sums of (g * x per channel, g) for the weight / bias gradient
//   spatial_mean   F.avg_pool2d(template_feat, 7) on 7x7 template features (network.py:343) = the mean over the HW positions
//                  of [B][C][HW] (NCHW) or [B][HW][C] (channels-last) -> [B][C]; backward = broadcast of g / HW
//   small_matmul   out [M][N] = a [M][K] b [K][N] for a handful of rows (conv_sub's response to the per-template constant
//                  image: 21 x 640 x 2304) -- no library GEMM on the per-object path
namespace {

// one wave per row: lanes stride over the channels in float4
__global__ __launch_bounds__(256) void conv1x1_c1_fwd_kernel(const float4* __restrict__ x, const float* __restrict__ w,
                                                             const float* __restrict__ bias, long long rows, int C4, int sigmoid,
                                                             float* __restrict__ out) {
    const long long r = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const int lane = threadIdx.x & 63;
    float s = 0.0f;
    for (int c = lane; c < C4; c += 64) {
        const float4 a = x[(size_t)r * C4 + c];          // (w: a slice of the flat parameter buffer, any 4-byte alignment)
        s = fmaf(a.x, w[4 * c], s), s = fmaf(a.y, w[4 * c + 1], s), s = fmaf(a.z, w[4 * c + 2], s), s = fmaf(a.w, w[4 * c + 3], s);
    }
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    if (lane == 0) {
        s += bias ? bias[0] : 0.0f;
        out[r] = sigmoid ? 1.0f / (1.0f + expf(-s)) : s;
    }
}

// grid = row chunks; thread (c4, strip): dx[r][c] = g[r] * w[c]; partial sums over the chunk's rows of g[r] * x[r][c]
// -> partials[chunk][0..C), of g[r] -> partials[chunk][C]
__global__ __launch_bounds__(256) void conv1x1_c1_bwd_kernel(const float4* __restrict__ x, const float* __restrict__ g,
                                                             const float* __restrict__ w, long long rows, int C4,
                                                             int rows_per_block, float4* __restrict__ dx,
                                                             float* __restrict__ partials) {
    __shared__ float red[256][4];
    __shared__ float redg[256];
    const int c4 = threadIdx.x % C4, strip = threadIdx.x / C4, nstrips = 256 / C4;
    const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
    const float4 wv = make_float4(w[4 * c4], w[4 * c4 + 1], w[4 * c4 + 2], w[4 * c4 + 3]);
    float s[4] = {0.f, 0.f, 0.f, 0.f}, sg = 0.f;
    for (long long r = r0 + strip; r < r1; r += nstrips) {
        const float gv = g[r];
        const float4 xv = x[(size_t)r * C4 + c4];
        if (dx) dx[(size_t)r * C4 + c4] = make_float4(gv * wv.x, gv * wv.y, gv * wv.z, gv * wv.w);
        s[0] = fmaf(gv, xv.x, s[0]), s[1] = fmaf(gv, xv.y, s[1]), s[2] = fmaf(gv, xv.z, s[2]), s[3] = fmaf(gv, xv.w, s[3]);
        if (c4 == 0) sg += gv;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) red[strip * C4 + c4][i] = s[i];
    redg[threadIdx.x] = sg;
    __syncthreads();
    const int C = 4 * C4;
    if (strip == 0) {
        float t[4] = {red[c4][0], red[c4][1], red[c4][2], red[c4][3]};
        for (int y = 1; y < nstrips; ++y)
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] += red[y * C4 + c4][i];
        float* row = partials + (size_t)blockIdx.x * (C + 1);
        row[4 * c4] = t[0], row[4 * c4 + 1] = t[1], row[4 * c4 + 2] = t[2], row[4 * c4 + 3] = t[3];
        if (c4 == 0) {
            float tg = redg[0];
            for (int y = 1; y < nstrips; ++y) tg += redg[y * C4];
            row[C] = tg;
        }
    }
}

// out[i] = sum over chunks of partials[chunk][i], fixed order (32 outputs x 8 chunk ranges per workgroup)
__global__ __launch_bounds__(256) void rows_reduce_kernel(const float* __restrict__ partials, int nparts, int n, float* __restrict__ out) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    const int per = (nparts + 7) / 8, g0 = part * per, g1 = min(nparts, g0 + per);
    float s = 0.0f;
    if (i < n)
        for (int g = g0; g < g1; ++g) s += partials[(size_t)g * n + i];
    red[part][col] = s;
    __syncthreads();
    if (part == 0 && i < n) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        out[i] = t;
    }
}

__global__ __launch_bounds__(256) void spatial_mean_kernel(const float* __restrict__ x, int HW, int C, int channels_last, int backward,
                                                           size_t total, float* __restrict__ out) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;     // forward: i over [B][C]; backward: i over x's elements
    if (i >= total) return;
    const float inv = 1.0f / (float)HW;
    if (!backward) {
        const size_t b = i / C, c = i % C;
        const float* p = channels_last ? x + b * HW * C + c : x + i * HW;
        const size_t st = channels_last ? C : 1;
        float s = 0.0f;
        for (int k = 0; k < HW; ++k) s += p[k * st];
        out[i] = s * inv;
    } else {                                                      // x = g [B][C]; out = dx in the forward input's layout
        size_t b, c;
        if (channels_last) b = i / ((size_t)HW * C), c = i % C;
        else b = i / ((size_t)HW * C), c = (i / HW) % C;
        out[i] = x[b * C + c] * inv;
    }
}

__global__ __launch_bounds__(256) void small_matmul_kernel(const float* __restrict__ a, const float* __restrict__ b, int K, int N,
                                                           float* __restrict__ out) {
    const int n = blockIdx.x * 256 + threadIdx.x, m = blockIdx.y;
    if (n >= N) return;
    const float* ar = a + (size_t)m * K;
    float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    int k = 0;
    for (; k + 16 <= K; k += 16) {           // 16 independent loads in flight per trip: the loop is a chain of L2 round trips
        float bv[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) bv[u] = b[(size_t)(k + u) * N + n];
#pragma unroll
        for (int u = 0; u < 16; ++u) s[u & 7] = fmaf(ar[k + u], bv[u], s[u & 7]);
    }
    for (; k < K; ++k) s[0] = fmaf(ar[k], b[(size_t)k * N + n], s[0]);
    out[(size_t)m * N + n] = ((s[0] + s[1]) + (s[2] + s[3])) + ((s[4] + s[5]) + (s[6] + s[7]));
}

}  // namespace

extern "C" {

int ossid_conv1x1_c1_fwd(const float* x, long long rows, int C, const float* w, const float* bias, int sigmoid, float* out,
                         void* stream) {
    if (!x || !w || !out || rows < 0 || C <= 0 || C % 4 || ((uintptr_t)x & 15)) return OSSID_EINVAL;
    if (rows == 0) return OSSID_OK;
    hipLaunchKernelGGL(conv1x1_c1_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, (const float4*)x,
                       w, bias, rows, C / 4, sigmoid, out);
    return ossid_launch_status();
}

static int conv1x1_c1_rows_per_block(long long rows) { return rows > 65536 ? 256 : 64; }

size_t ossid_conv1x1_c1_bwd_workspace_floats(long long rows, int C) {
    if (rows <= 0 || C <= 0) return 0;
    const int rpb = conv1x1_c1_rows_per_block(rows);
    return (size_t)((rows + rpb - 1) / rpb) * (C + 1);
}

int ossid_conv1x1_c1_bwd(const float* x, const float* g, long long rows, int C, const float* w, float* workspace, float* dx,
                         float* dw_db, void* stream) {
    const int C4 = C / 4;
    if (!x || !g || !w || !workspace || !dw_db || rows <= 0 || C <= 0 || C % 4 || C4 > 256 || (C4 & (C4 - 1)) ||
        ((uintptr_t)x & 15) || ((uintptr_t)dx & 15))
        return OSSID_EINVAL;
    const int rpb = conv1x1_c1_rows_per_block(rows);
    const int chunks = (int)((rows + rpb - 1) / rpb);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(conv1x1_c1_bwd_kernel, dim3(chunks), dim3(256), 0, s, (const float4*)x, g, w, rows, C4, rpb,
                       (float4*)dx, workspace);
    hipLaunchKernelGGL(rows_reduce_kernel, dim3((C + 1 + 31) / 32), dim3(256), 0, s, (const float*)workspace, chunks, C + 1, dw_db);
    return ossid_launch_status();
}

int ossid_spatial_mean(const float* x, int B, int HW, int C, int channels_last, int backward, float* out, void* stream) {
    if (!x || !out || B <= 0 || HW <= 0 || C <= 0) return OSSID_EINVAL;
    const size_t total = backward ? (size_t)B * HW * C : (size_t)B * C;
    hipLaunchKernelGGL(spatial_mean_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, x, HW, C,
                       channels_last, backward, total, out);
    return ossid_launch_status();
}

int ossid_small_matmul(const float* a, const float* b, int M, int K, int N, float* out, void* stream) {
    if (!a || !b || !out || M <= 0 || M > 65535 || K <= 0 || N <= 0) return OSSID_EINVAL;
    hipLaunchKernelGGL(small_matmul_kernel, dim3((N + 255) / 256, M), dim3(256), 0, (hipStream_t)stream, a, b, K, N, out);
    return ossid_launch_status();
}

}  // extern "C"
