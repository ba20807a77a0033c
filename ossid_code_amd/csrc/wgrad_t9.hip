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
// tiles of the workgroup. Workgroups are persistent (one per CU: 330 registers), each belongs to ONE job of the launch (the L
// layers of a block are one launch), prefetches its next tile's loads under the current tile's MFMAs, and writes one partial slab
// [tap][co][ci]; a second kernel adds a problem's slabs in a fixed order: bit-reproducible, no float atomics.
// Arithmetic: dy_lo * x_hi + dy_hi * x_lo + dy_hi * x_hi per product, f32 accumulation, as train.hip's split form. A
// -DOSSID_WGRAD_F32 build does not use this file (ossid_wgrad_t9_takes returns false).
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf16 __attribute__((ext_vector_type(8)));
typedef short v4i16 __attribute__((ext_vector_type(4)));

constexpr int T9_CIN = 128;                            // input channels of a job (a block of the layer's)
constexpr int T9_TH = 4, T9_TW = 16, T9_PH = T9_TH + 2, T9_PW = T9_TW + 2;
constexpr int T9_PX = 320;                             // bytes per pixel of the x images (pitch mod 256 = 64: see train.hip)
constexpr int T9_XIMG = T9_PH * T9_PW * T9_PX;         // bytes per part
constexpr int t9_pd(int tm) { return tm == 1 ? 64 : 192; }                 // bytes per pixel of the dY images (32 / 64 channels)
constexpr int t9_lds(int tm) { return 2 * (T9_XIMG + T9_TH * T9_TW * t9_pd(tm)); }   // 77 312 / 93 696 B
constexpr int T9_MAX = 48;                             // jobs per launch
constexpr int t9_slab(int tm) { return 9 * tm * 32 * T9_CIN; }             // floats

// A job = (layer, tile of 32 TM output channels, block of 128 input channels): pointers already offset to the job's first
// channels. The dense blocks' layers (128 -> 32) are one job each (TM = 1); the head's 3x3 layers are cut into jobs of 64 output
// channels (TM = 2: every staged input pixel then feeds two tile products per tap).
struct T9Job {
    const float *x, *dy, *pre_scale, *pre_shift;
    float* dw;                                         // dw + (co0 * cin_total + ci0) * 9
    int cin_total, cout_w, in_cs, dy_cs, pre_relu, accumulate;      // cout_w: output channels of this job that exist (<= 32 TM)
};
struct T9Args {
    T9Job j[T9_MAX];
    float* slabs;                                      // [n][nwg][9][32 TM][128]
    int n, nwg;                                        // jobs, workgroups per job
    int B, H, W, tiles_y, tiles_x, ntiles;
};

template <int TM>
__global__ __launch_bounds__(256, 1) void wgrad_t9_kernel(const T9Args A) {
    constexpr int PD = t9_pd(TM), DIMG = T9_TH * T9_TW * PD, D4 = TM * 8;      // float4 of dY per pixel
    extern __shared__ __attribute__((aligned(16))) char lds[];
    char* xl = lds;                                    // [2 parts][PH][PW][PX]
    char* dyl = lds + 2 * T9_XIMG;                     // [2 parts][TH * TW][PD]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ji = blockIdx.x / A.nwg, wg = blockIdx.x - ji * A.nwg;
    if (ji >= A.n) return;
    const T9Job P = A.j[ji];
    const int H = A.H, W = A.W;

    v16f acc[TM][9];
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[m][t][r] = 0.0f;

    // staging maps: x patch = 108 pixels x 32 float4 (13.5 per thread), dY tile = 64 pixels x D4 float4 (2 TM per thread)
    constexpr int NX = (T9_PH * T9_PW * (T9_CIN / 4) + 255) / 256, ND = (T9_TH * T9_TW * D4) / 256;
    const int xq = tid & 31, dq = tid % D4;
    const bool dq_ok = 4 * dq < P.cout_w;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool has_pre = P.pre_scale != nullptr, relu = P.pre_relu != 0;
    if (has_pre) {
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
            const int idx = tid / D4 + (256 / D4) * e;                         // tile pixel
            const int py = idx / T9_TW, px = idx - py * T9_TW;
            sd[e] = *(const float4*)(P.dy + ((size_t)(b * H + min(oy0 + py, H - 1)) * W + min(ox0 + px, W - 1)) * P.dy_cs +
                                     (dq_ok ? 4 * dq : 0));
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
            if (has_pre) {
                v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                if (relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
            }
            const float f = (yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.0f : 0.0f;      // zero padding (finite clamped values)
            split_store(make_float4(f * v.x, f * v.y, f * v.z, f * v.w), xl + (size_t)idx * T9_PX + 8 * xq, T9_XIMG);
        }
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int idx = tid / D4 + (256 / D4) * e;
            const int py = idx / T9_TW, px = idx - py * T9_TW;
            const float f = (dq_ok && oy0 + py < H && ox0 + px < W) ? 1.0f : 0.0f;
            const float4 v = sd[e];
            split_store(make_float4(f * v.x, f * v.y, f * v.z, f * v.w), dyl + (size_t)idx * PD + 8 * dq, DIMG);
        }
    };
    // transposed-read addresses (as csrc/train.hip): within its group of 16 lanes, lane 4q+p supplies the address of block row q
    // (a pixel), channels 4p..4p+3 of the block's 16; groups 0 / 1 take channels 0-15 / 16-31 of a 32-channel tile, the wave's
    // halves the pixels 8h..8h+7 of a 16-pixel row (two blocks of 4 pixels each)
    const int tq = (lane >> 2) & 3, tp = lane & 3, tg = (lane >> 4) & 1;
    const char* a_base = dyl + (size_t)(8 * h + tq) * PD + (16 * tg + 4 * tp) * 2;
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
            v8bf16 a_hi[TM], a_lo[TM];
#pragma unroll
            for (int m = 0; m < TM; ++m) {
                a_hi[m] = tr8(a_base + (size_t)(r * T9_TW) * PD + m * 64, PD);
                a_lo[m] = tr8(a_base + DIMG + (size_t)(r * T9_TW) * PD + m * 64, PD);
            }
#pragma unroll
            for (int ky = 0; ky < 3; ++ky)
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    const char* at = b_base + (size_t)((r + ky) * T9_PW + kx) * T9_PX;
                    const v8bf16 b_hi = tr8(at, T9_PX), b_lo = tr8(at + T9_XIMG, T9_PX);
                    const int t = ky * 3 + kx;
#pragma unroll
                    for (int m = 0; m < TM; ++m) {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_lo[m], b_hi, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[m], b_lo, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_hi[m], b_hi, acc[m][t], 0, 0, 0);
                    }
                }
        }
    }
    float* slab = A.slabs + ((size_t)ji * A.nwg + wg) * t9_slab(TM);
    const int ci = wave * 32 + c;
#pragma unroll
    for (int m = 0; m < TM; ++m)
#pragma unroll
        for (int t = 0; t < 9; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int co = m * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                slab[((size_t)t * (TM * 32) + co) * T9_CIN + ci] = acc[m][t][r];
            }
}

// dw[co][ci][tap] (+)= sum over a job's slabs [tap][co][ci], fixed order; 8 slab ranges x 32 outputs per workgroup
template <int TM>
__global__ __launch_bounds__(256) void wgrad_t9_reduce_kernel(const T9Args A) {
    __shared__ float red[8][32];
    constexpr int SLAB = t9_slab(TM), per_job = SLAB / 32;
    const int ji = blockIdx.x / per_job, blk = blockIdx.x - ji * per_job;
    const T9Job P = A.j[ji];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blk * 32 + col;                          // slab element: (tap, co, ci); a block's 32 share (tap, co)
    const int tap = i / (TM * 32 * T9_CIN), rem = i - tap * (TM * 32 * T9_CIN);
    const int co = rem / T9_CIN, ci = rem - co * T9_CIN;
    if (co >= P.cout_w) return;                            // (uniform per block)
    const float* base = A.slabs + (size_t)ji * A.nwg * SLAB + i;
    const int per = (A.nwg + 7) / 8, g0 = part * per, g1 = min(A.nwg, g0 + per);
    float s = 0.0f;
    int g = g0;
    for (; g + 8 <= g1; g += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = base[(size_t)(g + u) * SLAB];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += v[u];
    }
    for (; g < g1; ++g) s += base[(size_t)g * SLAB];
    red[part][col] = s;
    __syncthreads();
    if (part == 0) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        float* o = P.dw + ((size_t)co * P.cin_total + ci) * 9 + tap;
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
    const float *dy_add, *dy_b, *dy_k;                 // optional: dy = dy + dy_b[co] * dy_add + dy_k[co] while staging
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
    float4 sx[16], sd[8], sa[8];
    const bool has_add = J.dy_add != nullptr;           // (uniform)
    float4 db4 = make_float4(0.f, 0.f, 0.f, 0.f), dk4 = db4;
    if (has_add) db4 = *(const float4*)(J.dy_b + 4 * dq), dk4 = *(const float4*)(J.dy_k + 4 * dq);
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
            if (has_add) sa[e] = *(const float4*)(J.dy_add + (size_t)p * J.dy_cs + 4 * dq);
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
            float4 v = sd[e];
            if (has_add)                                 // (g + c_x y) + c_1: the order of the generic pass this replaces
                v = make_float4((v.x + db4.x * sa[e].x) + dk4.x, (v.y + db4.y * sa[e].y) + dk4.y, (v.z + db4.z * sa[e].z) + dk4.z,
                                (v.w + db4.w * sa[e].w) + dk4.w);
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

int g_t9_grid[3] = {0, 0, 0};
template <int TM>
int t9_grid() {
    if (!g_t9_grid[TM]) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipFuncSetAttribute((const void*)wgrad_t9_kernel<TM>, hipFuncAttributeMaxDynamicSharedMemorySize, t9_lds(TM)) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, wgrad_t9_kernel<TM>, 256, t9_lds(TM)) != hipSuccess || per_cu <= 0)
            return 256;
        g_t9_grid[TM] = per_cu * p.multiProcessorCount;
    }
    return g_t9_grid[TM];
}

}  // namespace

// (not part of the public ABI: csrc/train.hip's ossid_conv_wgrad / ossid_conv_wgrad_group and their workspace queries route
// eligible problems here)
bool ossid_wgrad_t9_takes(const ossid_wgrad_desc* d) {
#ifdef OSSID_WGRAD_F32
    (void)d;
    return false;
#else
    const int in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : d->cin;
    const int dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : d->cout;
    return !d->dy_add && d->taps == 9 && d->cin >= T9_CIN && d->cin % T9_CIN == 0 && d->cout >= 32 && d->cout % 4 == 0 &&
           (d->src_height <= 0 || d->src_height == d->height) && (d->src_width <= 0 || d->src_width == d->width) &&
           in_cs % 4 == 0 && dy_cs % 4 == 0 && !((uintptr_t)d->x & 15) && !((uintptr_t)d->dy & 15) &&
           (!d->pre_scale || !((uintptr_t)d->pre_scale & 15)) && (!d->pre_shift || !((uintptr_t)d->pre_shift & 15)) &&
           d->batch > 0 && d->height > 0 && d->width > 0;
#endif
}

// output channels per job: 64 once the layer has that many
static int t9_tm(const ossid_wgrad_desc& d) { return d.cout >= 64 ? 2 : 1; }
static int t9_jobs_of(const ossid_wgrad_desc& d) { return ((d.cout + 32 * t9_tm(d) - 1) / (32 * t9_tm(d))) * (d.cin / T9_CIN); }
static long long t9_tiles(const ossid_wgrad_desc& d) {
    return (long long)d.batch * ((d.height + T9_TH - 1) / T9_TH) * ((d.width + T9_TW - 1) / T9_TW);
}
static int t9_nwg(int tm, int jobs, long long ntiles) {
    int nwg = (tm == 2 ? t9_grid<2>() : t9_grid<1>()) / jobs;
    if (nwg < 1) nwg = 1;
    if (nwg > ntiles) nwg = (int)ntiles;
    return nwg;
}
int ossid_wgrad_t9_max_jobs(void) { return T9_MAX; }
int ossid_wgrad_t9_class(const ossid_wgrad_desc* d) { return t9_tm(*d); }          // problems of one call share it
int ossid_wgrad_t9_job_count(const ossid_wgrad_desc* descs, int n) {
    int jobs = 0;
    for (int i = 0; i < n; ++i) jobs += t9_jobs_of(descs[i]);
    return jobs;
}

// n problems of ONE geometry (batch, height, width) and one class, all accepted by ossid_wgrad_t9_takes. A problem with more
// jobs than a launch takes is cut over several launches here; the list as a whole may be of any length.
size_t ossid_wgrad_t9_workspace_bytes(const ossid_wgrad_desc* descs, int n) {
    if (n <= 0) return 0;
    const int tm = t9_tm(descs[0]);
    const int jobs = ossid_wgrad_t9_job_count(descs, n);
    const int per_launch = jobs < T9_MAX ? jobs : T9_MAX;
    // (every launch of the call has at most per_launch jobs, hence at least this many workgroups per job)
    return (size_t)jobs * t9_nwg(tm, per_launch, t9_tiles(descs[0])) * t9_slab(tm) * sizeof(float);
}

int ossid_wgrad_t9_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream) {
    if (n <= 0 || !workspace || ((uintptr_t)workspace & 15)) return OSSID_EINVAL;
    const int tm = t9_tm(descs[0]);
    const int jobs_all = ossid_wgrad_t9_job_count(descs, n);
    const int per_launch = jobs_all < T9_MAX ? jobs_all : T9_MAX;
    T9Args a;
    a.B = descs[0].batch, a.H = descs[0].height, a.W = descs[0].width;
    a.tiles_y = (a.H + T9_TH - 1) / T9_TH, a.tiles_x = (a.W + T9_TW - 1) / T9_TW;
    const long long nt = t9_tiles(descs[0]);
    if (nt > 0x7fffffff) return OSSID_EINVAL;
    a.ntiles = (int)nt;
    a.nwg = t9_nwg(tm, per_launch, nt);
    if (workspace_bytes < (size_t)jobs_all * a.nwg * t9_slab(tm) * sizeof(float)) return OSSID_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    float* ws = (float*)workspace;
    auto flush = [&]() {
        if (a.n == 0) return;
        a.slabs = ws;
        if (tm == 2) {
            hipLaunchKernelGGL(wgrad_t9_kernel<2>, dim3((unsigned)(a.n * a.nwg)), dim3(256), t9_lds(2), s, a);
            hipLaunchKernelGGL(wgrad_t9_reduce_kernel<2>, dim3((unsigned)(a.n * (t9_slab(2) / 32))), dim3(256), 0, s, a);
        } else {
            hipLaunchKernelGGL(wgrad_t9_kernel<1>, dim3((unsigned)(a.n * a.nwg)), dim3(256), t9_lds(1), s, a);
            hipLaunchKernelGGL(wgrad_t9_reduce_kernel<1>, dim3((unsigned)(a.n * (t9_slab(1) / 32))), dim3(256), 0, s, a);
        }
        ws += (size_t)a.n * a.nwg * t9_slab(tm);
        a.n = 0;
    };
    a.n = 0;
    for (int i = 0; i < n; ++i) {
        const ossid_wgrad_desc& d = descs[i];
        if (!ossid_wgrad_t9_takes(&d) || t9_tm(d) != tm || d.batch != a.B || d.height != a.H || d.width != a.W || !d.dw ||
            (d.pre_scale && !d.pre_shift))
            return OSSID_EINVAL;
        const int in_cs = d.in_channel_stride > 0 ? d.in_channel_stride : d.cin;
        const int dy_cs = d.dy_channel_stride > 0 ? d.dy_channel_stride : d.cout;
        for (int co0 = 0; co0 < d.cout; co0 += 32 * tm)
            for (int ci0 = 0; ci0 < d.cin; ci0 += T9_CIN) {
                T9Job& j = a.j[a.n++];
                j.x = d.x + ci0, j.dy = d.dy + co0, j.dw = d.dw + ((size_t)co0 * d.cin + ci0) * 9;
                j.pre_scale = d.pre_scale ? d.pre_scale + ci0 : nullptr, j.pre_shift = d.pre_shift ? d.pre_shift + ci0 : nullptr;
                j.cin_total = d.cin, j.cout_w = d.cout - co0 < 32 * tm ? d.cout - co0 : 32 * tm;
                j.in_cs = in_cs, j.dy_cs = dy_cs, j.pre_relu = d.pre_relu, j.accumulate = d.accumulate;
                if (a.n == T9_MAX) flush();
            }
    }
    flush();
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
    if (d->dy_add && (!d->dy_add_scale || !d->dy_add_shift || ((uintptr_t)d->dy_add & 15) || ((uintptr_t)d->dy_add_scale & 15) ||
                      ((uintptr_t)d->dy_add_shift & 15)))
        return false;
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
            j.dy_add = d.dy_add, j.dy_b = d.dy_add_scale, j.dy_k = d.dy_add_shift;
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
