// D16  weight gradients of the DenseNet blocks' 3x3 convolutions (128 -> 32, torchvision _DenseLayer.conv2 inside
// ImageFeatExtract, /root/reference/python/ossid/models/dtoid/network.py:164-184; run backward by scripts/online_learning.py:668)
// from 2-D pixel tiles with ALL NINE TAPS read from one staged patch, on the split-bf16 matrix cores.
//
//   dW[co][ci][ky][kx] = sum_px dY[px][co] * relu(s[ci] X[px + (ky-1, kx-1)][ci] + t[ci])      (zero outside the image)
//
// csrc/train.hip's general kernel gives a workgroup ONE image row (<= 48 pixels between two barriers) and ONE kernel row: every
// input row is fetched, prologue'd and split into bf16 pieces three times, and at 128 -> 32 a wave has 27 MFMAs per staged
// chunk. The 58 layers of this shape are 1.65 ms of the finetune step's weight-gradient stream at 57-79 TFLOP/s; their operands
// (100 MB per layer at 120 x 160 x 8) would pass in 20 us, and the 3 x 36 tile products in 15 us of matrix pipe.
// Here a workgroup owns a 4 x 16 pixel tile: the 6 x 18 input pixels under it are staged ONCE (prologue, split into hi / lo
// bf16 images [pixel][channel], the layout of train.hip's kernel: operands by ds_read_b64_tr_b16 with the pixels on the MFMA's
// K axis), wave w owns input channels 32w .. 32w+31 and walks the nine taps of the tile's four 16-pixel rows: 108 MFMAs per
// tile and wave between two barriers (four times the old ratio), nine accumulator tiles (144 registers) live across all the
// tiles of the workgroup. Workgroups are persistent (two per CU), each belongs to ONE problem of the group (the L layers of a
// block are one launch), prefetches its next tile's loads under the current tile's MFMAs, and writes one partial slab
// [tap][co][ci]; a second kernel adds a problem's slabs in a fixed order: bit-reproducible, no float atomics.
// Arithmetic: dy_lo * x_hi + dy_hi * x_lo + dy_hi * x_hi per product, f32 accumulation, as train.hip's split form. A
// -DOSSID_WGRAD_F32 build does not use this file (ossid_wgrad_t9_takes returns false).
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));

constexpr int T9_CIN = 128, T9_COUT = 32;
constexpr int T9_TH = 4, T9_TW = 16, T9_PH = T9_TH + 2, T9_PW = T9_TW + 2;
constexpr int T9_PX = 320, T9_PD = 64;                 // bytes per pixel of the x / dy images (pitch mod 256 = 64: see train.hip)
constexpr int T9_XIMG = T9_PH * T9_PW * T9_PX, T9_DIMG = T9_TH * T9_TW * T9_PD;      // bytes per part
constexpr int T9_LDS = 2 * (T9_XIMG + T9_DIMG);        // 77 312 B: two workgroups per CU
constexpr int T9_MAX = 24;                             // problems per launch (a DenseNet-121 block has at most 24 layers)
constexpr int T9_SLAB = 9 * T9_COUT * T9_CIN;          // floats

struct T9Problem {
    const float *x, *dy, *pre_scale, *pre_shift;
    float* slabs;                                      // [nwg][9][32][128]
    float* dw;
    int in_cs, dy_cs, pre_relu, accumulate;
};
struct T9Args {
    T9Problem p[T9_MAX];
    int n, nwg;                                        // problems, workgroups per problem
    int B, H, W, tiles_y, tiles_x, ntiles;
};

__global__ __launch_bounds__(256, 1) void wgrad_t9_kernel(const T9Args A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xl = lds;                                    // [2 parts][PH][PW][PX]
    char* dyl = lds + 2 * T9_XIMG;                     // [2 parts][TH * TW][PD]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int pi = blockIdx.x / A.nwg, wg = blockIdx.x - pi * A.nwg;
    if (pi >= A.n) return;
    const T9Problem P = A.p[pi];
    const int H = A.H, W = A.W;

    v16f acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    // staging maps: x patch = 108 pixels x 32 float4 (13.5 per thread), dy tile = 64 pixels x 8 float4 (2 per thread)
    constexpr int NX = (T9_PH * T9_PW * (T9_CIN / 4) + 255) / 256, ND = (T9_TH * T9_TW * (T9_COUT / 4)) / 256;
    const int xq = tid & 31, dq = tid & 7;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (P.pre_scale) {
        ps = *(const float4*)(P.pre_scale + 4 * xq);
        pt = *(const float4*)(P.pre_shift + 4 * xq);
    }
    float4 sx[NX], sd[ND];
    // the loads only: nothing here may wait for them (they fly under the current tile's MFMAs); prologue, zero padding and the
    // bf16 split happen in commit(), when the tile's matrix work has been issued
    auto fetch = [&](int tile) {
        const int tx = tile % A.tiles_x, r1 = tile / A.tiles_x;
        const int b = r1 / A.tiles_y, oy0 = (r1 % A.tiles_y) * T9_TH, ox0 = tx * T9_TW;
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const int idx = (tid >> 5) + 8 * e;                                // patch pixel
            const int py = idx / T9_PW, px = idx - py * T9_PW;
            const int yc = min(max(oy0 - 1 + py, 0), H - 1), xc = min(max(ox0 - 1 + px, 0), W - 1);     // (clamped: unconditional loads)
            sx[e] = *(const float4*)(P.x + ((size_t)(b * H + yc) * W + xc) * P.in_cs + 4 * xq);
        }
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int idx = (tid >> 3) + 32 * e;                               // tile pixel
            const int py = idx / T9_TW, px = idx - py * T9_TW;
            sd[e] = *(const float4*)(P.dy + ((size_t)(b * H + min(oy0 + py, H - 1)) * W + min(ox0 + px, W - 1)) * P.dy_cs + 4 * dq);
        }
    };
    auto split_store = [&](const float4& fv, char* hi_at, int part_stride) {
        const float v[4] = {fv.x, fv.y, fv.z, fv.w};
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
    auto commit = [&](int tile) {
        const int tx = tile % A.tiles_x, r1 = tile / A.tiles_x;
        const int oy0 = (r1 % A.tiles_y) * T9_TH, ox0 = tx * T9_TW;
#pragma unroll
        for (int e = 0; e < NX; ++e) {
            const int idx = (tid >> 5) + 8 * e;
            if (idx >= T9_PH * T9_PW) continue;
            const int py = idx / T9_PW, px = idx - py * T9_PW;
            const int yy = oy0 - 1 + py, xx = ox0 - 1 + px;
            float4 v = sx[e];
            if (P.pre_scale) {
                v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                if (P.pre_relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            }
            const float f = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.0f : 0.0f;      // zero padding (finite clamped values)
            split_store(make_float4(f * v.x, f * v.y, f * v.z, f * v.w), xl + (size_t)idx * T9_PX + 8 * xq, T9_XIMG);
        }
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int idx = (tid >> 3) + 32 * e;
            const int py = idx / T9_TW, px = idx - py * T9_TW;
            const float f = (oy0 + py < H && ox0 + px < W) ? 1.0f : 0.0f;
            const float4 v = sd[e];
            split_store(make_float4(f * v.x, f * v.y, f * v.z, f * v.w), dyl + (size_t)idx * T9_PD + 8 * dq, T9_DIMG);
        }
    };
    // transposed-read addresses (as csrc/train.hip): within its group of 16 lanes, lane 4q+p supplies the address of block row q
    // (a pixel), channels 4p..4p+3 of the block's 16; groups 0 / 1 take channels 0-15 / 16-31 of a 32-channel tile, the wave's
    // halves the pixels 8h..8h+7 of a 16-pixel row (two blocks of 4 pixels each)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const char* a_base = dyl + (size_t)(8 * h + tq) * T9_PD + (16 * tg + 4 * tp) * 2;
    const char* b_base = xl + (size_t)(8 * h + tq) * T9_PX + (wave * 32 + 16 * tg + 4 * tp) * 2;
    auto tr8 = [&](const char* at, int pitch) {
        const v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)at);
        const v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(at + 4 * pitch));
        typedef short v8i16 __attribute__((ext_vector_type(8)));
        const v8i16 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(v8bf16, v);
    };

    if (wg < A.ntiles) fetch(wg);
    for (int tile = wg; tile < A.ntiles; tile += A.nwg) {
        __syncthreads();                                   // the previous tile's readers are done
        commit(tile);
        __syncthreads();
        if (tile + A.nwg < A.ntiles) fetch(tile + A.nwg);  // in flight under this tile's MFMAs
#pragma unroll 1
        for (int r = 0; r < T9_TH; ++r) {
            const v8bf16 a_hi = tr8(a_base + (size_t)(r * T9_TW) * T9_PD, T9_PD);
            const v8bf16 a_lo = tr8(a_base + T9_DIMG + (size_t)(r * T9_TW) * T9_PD, T9_PD);
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const char* at = b_base + (size_t)((r + ky) * T9_PW + kx) * T9_PX;
                    const v8bf16 b_hi = tr8(at, T9_PX), b_lo = tr8(at + T9_XIMG, T9_PX);
                    const int t = ky * 3 + kx;
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[t], 0, 0, 0);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[t], 0, 0, 0);
                }
        }
    }
    float* slab = P.slabs + (size_t)wg * T9_SLAB;
    const int ci = wave * 32 + c;
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[((size_t)t * T9_COUT + co) * T9_CIN + ci] = acc[t][r];
        }
}

// dw[co][ci][tap] (+)= sum over a problem's slabs [tap][co][ci], fixed order; 8 slab ranges x 32 outputs per workgroup
__global__ __launch_bounds__(256) void wgrad_t9_reduce_kernel(const T9Args A) {
    __shared__ float red[8][32];
    constexpr int per_problem = (T9_SLAB + 31) / 32;
    const int pi = blockIdx.x / per_problem, blk = blockIdx.x - pi * per_problem;
    const T9Problem P = A.p[pi];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blk * 32 + col;                          // slab element: (tap, co, ci)
    const int per = (A.nwg + 7) / 8, g0 = part * per, g1 = min(A.nwg, g0 + per);
    float s = 0.0f;
    if (i < T9_SLAB) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = P.slabs[(size_t)(g + u) * T9_SLAB + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < g1; ++g) s += P.slabs[(size_t)g * T9_SLAB + i];
    }
    red[part][col] = s;
    __syncthreads();
    if (part == 0 && i < T9_SLAB) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        const int tap = i / (T9_COUT * T9_CIN), rem = i - tap * (T9_COUT * T9_CIN);
        const int co = rem / T9_CIN, ci = rem - co * T9_CIN;
        float* o = P.dw + ((size_t)co * T9_CIN + ci) * 9 + tap;
        *o = P.accumulate ? *o + t : t;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// The dense layers' FIRST convolution (1x1, c -> 128, torchvision _DenseLayer.conv1): dW[co][ci] = sum_px dZ[px][co] *
// relu(s[ci] X[px][ci] + t[ci]) with X the first c channels of the block's resident buffer. Same operand scheme; a job =
// (layer, block of up to 256 input channels): 64-pixel stages (dZ 64 x 128 and X 64 x 256 split into bf16 images), wave w owns
// output channels 32w .. 32w+31 against up to eight 32-channel column tiles: 96 MFMAs per stage and wave between two barriers
// (train.hip's tiling for this shape: 36 per 48-pixel chunk, and the f32 -> 2 x bf16 split of a staged element feeds four tile
// products there, eight here). Persistent workgroups, one slab [128][256] each, fixed-order reduction.
constexpr int T1_COUT = 128, T1_CIB = 256, T1_PXS = 64;
constexpr int T1_PD = 320, T1_PXB = 576;               // bytes per pixel of the dZ / X images (pitch mod 256 = 64)
constexpr int T1_DIMG = T1_PXS * T1_PD, T1_XIMG = T1_PXS * T1_PXB;
constexpr int T1_LDS = 2 * (T1_DIMG + T1_XIMG);        // 114 688 B
constexpr int T1_MAX = 48;                             // jobs per launch
constexpr int T1_SLAB = T1_COUT * T1_CIB;

struct T1Job {
    const float *x, *dy, *pre_scale, *pre_shift;       // x / pre_* already offset to the job's first input channel
    float* dw;                                         // dw + first input channel; rows of cin_total floats
    int cw, cin_total, in_cs, dy_cs, pre_relu, accumulate;    // cw = input channels of this block (<= 256, a multiple of 32)
};
struct T1Args {
    T1Job j[T1_MAX];
    float* slabs;                                      // [n][nwg][128][256]
    long long npx;
    int n, nwg, nstages;
};

__global__ __launch_bounds__(256, 1) void wgrad_t1_kernel(const T1Args A) {
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* dyl = lds;                                   // [2 parts][64][PD]
    char* xl = lds + 2 * T1_DIMG;                      // [2 parts][64][PXB]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ji = blockIdx.x / A.nwg, wg = blockIdx.x - ji * A.nwg;
    if (ji >= A.n) return;
    const T1Job J = A.j[ji];
    const int ntn = J.cw / 32;                         // column tiles in use (uniform)

    v16f acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int xq = tid & 63, dq = tid & 31;
    const bool xq_ok = 4 * xq < J.cw;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (J.pre_scale && xq_ok) {
        ps = *(const float4*)(J.pre_scale + 4 * xq);
        pt = *(const float4*)(J.pre_shift + 4 * xq);
    }
    const bool has_pre = J.pre_scale != nullptr, relu = J.pre_relu != 0;
    float4 sx[16], sd[8];
    auto fetch = [&](int stage) {                       // loads only (see wgrad_t9_kernel)
        const long long p0 = (long long)stage * T1_PXS;
        if (xq_ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long p = min(p0 + (tid >> 6) + 4 * e, A.npx - 1);
                sx[e] = *(const float4*)(J.x + (size_t)p * J.in_cs + 4 * xq);
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long p = min(p0 + (tid >> 5) + 8 * e, A.npx - 1);
            sd[e] = *(const float4*)(J.dy + (size_t)p * J.dy_cs + 4 * dq);
        }
    };
    auto split_store = [&](const float4& fv, char* hi_at, int part_stride) {
        const float v[4] = {fv.x, fv.y, fv.z, fv.w};
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
    auto commit = [&](int stage) {
        const long long p0 = (long long)stage * T1_PXS;
        if (xq_ok) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int px = (tid >> 6) + 4 * e;
                float4 v = sx[e];
                if (has_pre) {
                    v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                    if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
                split_store(v, xl + (size_t)px * T1_PXB + 8 * xq, T1_XIMG);      // (pixels past the end pair with zero dZ rows)
            }
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int px = (tid >> 5) + 8 * e;
            const float f = (p0 + px < A.npx) ? 1.0f : 0.0f;
            const float4 v = sd[e];
            split_store(make_float4(f * v.x, f * v.y, f * v.z, f * v.w), dyl + (size_t)px * T1_PD + 8 * dq, T1_DIMG);
        }
    };
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const char* a_base = dyl + (size_t)(8 * h + tq) * T1_PD + (wave * 32 + 16 * tg + 4 * tp) * 2;
    const char* b_base = xl + (size_t)(8 * h + tq) * T1_PXB + (16 * tg + 4 * tp) * 2;
    auto tr8 = [&](const char* at, int pitch) {
        const v4i16 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)at);
        const v4i16 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4i16*)(at + 4 * pitch));
        typedef short v8i16 __attribute__((ext_vector_type(8)));
        const v8i16 v = __builtin_shufflevector(lo4, hi4, 0, 1, 2, 3, 4, 5, 6, 7);
        return __builtin_bit_cast(v8bf16, v);
    };

    if (wg < A.nstages) fetch(wg);
    for (int stage = wg; stage < A.nstages; stage += A.nwg) {
        __syncthreads();
        commit(stage);
        __syncthreads();
        if (stage + A.nwg < A.nstages) fetch(stage + A.nwg);
#pragma unroll 1
        for (int k = 0; k < T1_PXS / 16; ++k) {
            const v8bf16 a_hi = tr8(a_base + (size_t)(16 * k) * T1_PD, T1_PD);
            const v8bf16 a_lo = tr8(a_base + T1_DIMG + (size_t)(16 * k) * T1_PD, T1_PD);
#pragma unroll
            for (int n = 0; n < 8; ++n) {
                if (n < ntn) {
                    const char* at = b_base + (size_t)(16 * k) * T1_PXB + n * 64;
                    const v8bf16 b_hi = tr8(at, T1_PXB), b_lo = tr8(at + T1_XIMG, T1_PXB);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo, b_hi, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_lo, acc[n], 0, 0, 0);
                    acc[n] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi, b_hi, acc[n], 0, 0, 0);
                }
            }
        }
    }
    float* slab = A.slabs + ((size_t)ji * A.nwg + wg) * T1_SLAB;
#pragma unroll
    for (int n = 0; n < 8; ++n) {
        if (n >= ntn) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int co = wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            slab[(size_t)co * T1_CIB + n * 32 + c] = acc[n][r];
        }
    }
}

// dw[co][ci] (+)= sum over a job's slabs [co][256], fixed order
__global__ __launch_bounds__(256) void wgrad_t1_reduce_kernel(const T1Args A) {
    __shared__ float red[8][32];
    constexpr int per_job = T1_SLAB / 32;
    const int ji = blockIdx.x / per_job, blk = blockIdx.x - ji * per_job;
    const T1Job J = A.j[ji];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blk * 32 + col;                          // slab element (co, ci)
    const int co = i / T1_CIB, ci = i - co * T1_CIB;
    if (blk * 32 % T1_CIB >= J.cw) return;                 // (whole block outside the job's channels: uniform)
    const float* base = A.slabs + (size_t)ji * A.nwg * T1_SLAB + i;
    const int per = (A.nwg + 7) / 8, g0 = part * per, g1 = min(A.nwg, g0 + per);
    float s = 0.0f;
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(g + u) * T1_SLAB];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < g1; ++g) s += base[(size_t)g * T1_SLAB];
    red[part][col] = s;
    __syncthreads();
    if (part == 0 && ci < J.cw) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        float* o = J.dw + (size_t)co * J.cin_total + ci;
        *o = J.accumulate ? *o + t : t;
    }
}

int g_t1_grid = 0;
int t1_grid() {
    if (!g_t1_grid) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_t1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T1_LDS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_t1_kernel, 256, T1_LDS) != hipSuccess || per_cu <= 0)
            return 256;
        g_t1_grid = per_cu * p.multiProcessorCount;
    }
    return g_t1_grid;
}

int g_t9_grid = 0;
int t9_grid() {
    if (!g_t9_grid) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_t9_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, T9_LDS) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_t9_kernel, 256, T9_LDS) != hipSuccess || per_cu <= 0)
            return 512;
        g_t9_grid = per_cu * p.multiProcessorCount;
    }
    return g_t9_grid;
}

}  // namespace

// (not part of the public ABI: csrc/train.hip's ossid_conv_wgrad_group / ..._workspace_bytes route eligible problems here)
bool ossid_wgrad_t9_takes(const ossid_wgrad_desc* d) {
#ifdef OSSID_WGRAD_F32
    (void)d;
    return false;
#else
    const int in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : d->cin;
    const int dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : d->cout;
    return d->taps == 9 && d->cin == T9_CIN && d->cout == T9_COUT && (d->src_height <= 0 || d->src_height == d->height) &&
           (d->src_width <= 0 || d->src_width == d->width) && in_cs % 4 == 0 && dy_cs % 4 == 0 && !((uintptr_t)d->x & 15) &&
           !((uintptr_t)d->dy & 15) && d->batch > 0 && d->height > 0 && d->width > 0;
#endif
}

// workgroups per problem for a group of n same-shaped problems
static int t9_nwg(int n, long long ntiles) {
    int nwg = t9_grid() / n;
    if (nwg < 1) nwg = 1;
    if (nwg > ntiles) nwg = (int)ntiles;
    return nwg;
}

size_t ossid_wgrad_t9_workspace_bytes(const ossid_wgrad_desc* descs, int n) {
    if (n <= 0 || n > T9_MAX) return 0;
    const long long nt = (long long)descs[0].batch * ((descs[0].height + T9_TH - 1) / T9_TH) * ((descs[0].width + T9_TW - 1) / T9_TW);
    return (size_t)n * t9_nwg(n, nt) * T9_SLAB * sizeof(float);
}

// n <= 24 problems of ONE geometry (batch, height, width), all accepted by ossid_wgrad_t9_takes
int ossid_wgrad_t9_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0 || n > T9_MAX || !workspace || ((uintptr_t)workspace & 15)) return OSSID_EINVAL;
    T9Args a;
    a.n = n, a.B = descs[0].batch, a.H = descs[0].height, a.W = descs[0].width;
    a.tiles_y = (a.H + T9_TH - 1) / T9_TH, a.tiles_x = (a.W + T9_TW - 1) / T9_TW;
    const long long nt = (long long)a.B * a.tiles_y * a.tiles_x;
    if (nt > 0x7fffffff) return OSSID_EINVAL;
    a.ntiles = (int)nt;
    a.nwg = t9_nwg(n, nt);
    if (workspace_bytes < (size_t)n * a.nwg * T9_SLAB * sizeof(float)) return OSSID_EINVAL;
    for (int i = 0; i < n; ++i) {
        const ossid_wgrad_desc& d = descs[i];
        if (!ossid_wgrad_t9_takes(&d) || d.batch != a.B || d.height != a.H || d.width != a.W || !d.dw || (d.pre_scale && !d.pre_shift))
            return OSSID_EINVAL;
        T9Problem& p = a.p[i];
        p.x = d.x, p.dy = d.dy, p.pre_scale = d.pre_scale, p.pre_shift = d.pre_shift, p.dw = d.dw;
        p.slabs = (float*)workspace + (size_t)i * a.nwg * T9_SLAB;
        p.in_cs = d.in_channel_stride > 0 ? d.in_channel_stride : d.cin;
        p.dy_cs = d.dy_channel_stride > 0 ? d.dy_channel_stride : d.cout;
        p.pre_relu = d.pre_relu, p.accumulate = d.accumulate;
    }
    hipStream_t s = (hipStream_t)stream;
    t9_grid();                                             // (sets the dynamic-LDS attribute once)
    hipLaunchKernelGGL(wgrad_t9_kernel, dim3((unsigned)(n * a.nwg)), dim3(256), T9_LDS, s, a);
    hipLaunchKernelGGL(wgrad_t9_reduce_kernel, dim3((unsigned)(n * ((T9_SLAB + 31) / 32))), dim3(256), 0, s, a);
    return ossid_launch_status();
}

// ---- 1x1, c -> 128 (the dense layers' first convolution) -------------------------------------------------------------------
bool ossid_wgrad_t1_takes(const ossid_wgrad_desc* d) {
#ifdef OSSID_WGRAD_F32
    (void)d;
    return false;
#else
    const int in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : d->cin;
    const int dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : d->cout;
    return d->taps == 1 && d->cout == T1_COUT && d->cin >= 32 && d->cin % 32 == 0 && in_cs % 4 == 0 && dy_cs % 4 == 0 &&
           !((uintptr_t)d->x & 15) && !((uintptr_t)d->dy & 15) && (!d->pre_scale || !((uintptr_t)d->pre_scale & 15)) &&
           (!d->pre_shift || !((uintptr_t)d->pre_shift & 15)) && d->batch > 0 && d->height > 0 && d->width > 0;
#endif
}

static int t1_jobs(const ossid_wgrad_desc* descs, int n) {
    int jobs = 0;
    for (int i = 0; i < n; ++i) jobs += (descs[i].cin + T1_CIB - 1) / T1_CIB;
    return jobs;
}
static int t1_nwg(int jobs, long long nstages) {
    int nwg = t1_grid() / jobs;
    if (nwg < 1) nwg = 1;
    if (nwg > nstages) nwg = (int)nstages;
    return nwg;
}

// n problems of ONE pixel count (batch * height * width), all accepted by ossid_wgrad_t1_takes, at most T1_MAX jobs
int ossid_wgrad_t1_max_jobs(void) { return T1_MAX; }
int ossid_wgrad_t1_job_count(const ossid_wgrad_desc* descs, int n) { return t1_jobs(descs, n); }

size_t ossid_wgrad_t1_workspace_bytes(const ossid_wgrad_desc* descs, int n) {
    const int jobs = t1_jobs(descs, n);
    if (n <= 0 || jobs > T1_MAX) return 0;
    const long long npx = (long long)descs[0].batch * descs[0].height * descs[0].width;
    return (size_t)jobs * t1_nwg(jobs, (npx + T1_PXS - 1) / T1_PXS) * T1_SLAB * sizeof(float);
}

int ossid_wgrad_t1_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream) {
    const int jobs = t1_jobs(descs, n);
    if (n <= 0 || jobs > T1_MAX || !workspace || ((uintptr_t)workspace & 15)) return OSSID_EINVAL;
    T1Args a;
    a.npx = (long long)descs[0].batch * descs[0].height * descs[0].width;
    const long long ns = (a.npx + T1_PXS - 1) / T1_PXS;
    if (ns > 0x7fffffff) return OSSID_EINVAL;
    a.nstages = (int)ns, a.n = jobs, a.nwg = t1_nwg(jobs, ns), a.slabs = (float*)workspace;
    if (workspace_bytes < (size_t)jobs * a.nwg * T1_SLAB * sizeof(float)) return OSSID_EINVAL;
    int ji = 0;
    for (int i = 0; i < n; ++i) {
        const ossid_wgrad_desc& d = descs[i];
        if (!ossid_wgrad_t1_takes(&d) || (long long)d.batch * d.height * d.width != a.npx || !d.dw || (d.pre_scale && !d.pre_shift))
            return OSSID_EINVAL;
        for (int c0 = 0; c0 < d.cin; c0 += T1_CIB) {
            T1Job& j = a.j[ji++];
            j.x = d.x + c0, j.dy = d.dy, j.dw = d.dw + c0;
            j.pre_scale = d.pre_scale ? d.pre_scale + c0 : nullptr, j.pre_shift = d.pre_shift ? d.pre_shift + c0 : nullptr;
            j.cw = d.cin - c0 < T1_CIB ? d.cin - c0 : T1_CIB, j.cin_total = d.cin;
            j.in_cs = d.in_channel_stride > 0 ? d.in_channel_stride : d.cin;
            j.dy_cs = d.dy_channel_stride > 0 ? d.dy_channel_stride : d.cout;
            j.pre_relu = d.pre_relu, j.accumulate = d.accumulate;
        }
    }
    hipStream_t s = (hipStream_t)stream;
    t1_grid();
    hipLaunchKernelGGL(wgrad_t1_kernel, dim3((unsigned)(jobs * a.nwg)), dim3(256), T1_LDS, s, a);
    hipLaunchKernelGGL(wgrad_t1_reduce_kernel, dim3((unsigned)(jobs * (T1_SLAB / 32))), dim3(256), 0, s, a);
    return ossid_launch_status();
}
