// The tail of DTOID's segmentation decoder in one launch (gfx950):
//   nearest up-sample [Hs][Ws] -> [H][W]  ->  conv3x3 32->16 + bias -> ELU -> BatchNorm(eval)  ->  conv3x3 16->1 + bias
// (/root/reference/python/ossid/models/dtoid/network.py:357-362: `x = F.interpolate(x, size=img_size)`,
//  `x = self.ns5(F.elu(self.s5(x)))`, `seg = self.seg_final(x)`.)
//
// At 480x640 with 21 templates these two layers were 1.9 ms of a 14.7 ms frame as separate launches: the 32->16 layer
// wasted half of every 32x32 matrix-core tile on channel padding and re-read each source row three times (2.5 GB past
// L2 for a 0.2 GB source), and the 16->1 layer is a 413 MB read for 1.9 GFLOP. Fused, the 16-channel full-resolution
// tensor (21 x 480 x 640 x 16 floats) never exists: a workgroup owns a 14 x 30 pixel output tile, stages the SOURCE
// pixels under its (18 x 34)-pixel up-sampled footprint once, runs the first conv for the 16 x 32 pixels the second conv
// needs on v_mfma_f32_16x16x4_f32 (16 output channels = one tile, no padding waste) into LDS, and finishes with the
// 144-tap second conv on the vector ALUs (weights broadcast from LDS).
//
// Arithmetic of the first conv (default build): split-bf16, as csrc/conv.hip -- every f32 product is three
// v_mfma_f32_16x16x32_bf16 (w_lo*x_hi + w_hi*x_lo + w_hi*x_hi, f32 accumulation; K = 32 = all input channels of a tap in
// ONE instruction): 48 pipe cycles per tap and 16-pixel tile instead of 256. The source patch is split when it is staged
// ([pixel][hi 32 ch | lo 32 ch] bf16, same 144-byte pixel stride), the weights -- including the row-merged ones, merged in
// f32 first -- when they are packed. -DOSSID_SEGTAIL_F32 keeps the exact-f32 form described next.
//
// MFMA operand layout (16x16x4, one block): A = weights, lane l holds W[co = l%16][k = l/16]; B = activations, lane l
// holds X[k = l/16][px = l%16]; D: lane l holds rows co = 4*(l/16)+r (r = 0..3) of column px = l%16. A lane's B quad is
// ONE ds_read_b128 of four consecutive channels of its pixel (k-group g = l/16 <-> channels 16*cb + 4g + i for the i-th
// MFMA of the quad); the 18 weight quads (9 taps x 2 channel blocks) live in registers for the whole workgroup.
#include "common.h"

namespace {

typedef float v4f __attribute__((ext_vector_type(4)));

constexpr int ST_TH = 14, ST_TW = 30;            // output tile
constexpr int ST_MH = ST_TH + 2, ST_MW = ST_TW + 2;   // first-conv ("mid") region: 16 x 32 = 32 MFMA column tiles
constexpr int ST_CIN = 32, ST_CMID = 16;
constexpr int ST_PSTRIDE = 36;                   // floats per staged source pixel (32 + 4: conflict-free ds_read_b128)
constexpr int ST_MSTRIDE = 20;                   // floats per mid pixel (16 + 4)
constexpr int ST_PR = 12, ST_PC = 20;            // source patch capacity (rows, cols) -- checked on the host
constexpr int ST_PATCH_FLOATS = (ST_PR * ST_PC + 1) * ST_PSTRIDE;    // + one all-zero pixel for the padding taps
constexpr int ST_MID_FLOATS = ST_MH * ST_MW * ST_MSTRIDE;
constexpr int ST_LDS_FLOATS = ST_PATCH_FLOATS + ST_MID_FLOATS + 144;

struct SegTailArgs {
    const float* x;          // [B][Hs][Ws][in_cs]
    const float4* w1p;       // packed [9 taps][2 blocks][64 lanes] float4
    const float *b1, *bn_scale, *bn_shift;   // [16]
    const float* w2;         // [16][3][3] (torch layout of a [1][16][3][3] weight)
    const float* b2;         // [1] or NULL
    float* out;              // [B][H][W]
    int H, W, Hs, Ws, in_cs, tiles_x, tiles_y;
    float scale_h, scale_w;
};

// (elu_fast: csrc/common.h)
__device__ __forceinline__ int src_index(int dst, float scale, int n_src) {
    return min((int)floorf((float)dst * scale), n_src - 1);   // F.interpolate(mode="nearest")
}

// ---- second conv (16 -> 1) on the vector ALUs: one output pixel per thread, two passes; weights broadcast from LDS
// (measured alternatives: weights through scalar loads, two rows per thread -- both slower)
__device__ __forceinline__ void seg_tail_second_conv(const SegTailArgs& A, const float* mid, const float* w2s, const int y0,
                                                     const int x0, const int b) {
    const int tid = threadIdx.x, H = A.H, W = A.W;
#ifdef ST_ABL_NOCONV2
    for (int p = tid; p < 1; p += 256) {
#else
    for (int p = tid; p < ST_TH * ST_TW; p += 256) {
#endif
        const int oy = p / ST_TW, ox = p - oy * ST_TW;
        const int y = y0 + oy, x = x0 + ox;
        if (y >= H || x >= W) continue;
        float s = A.b2 ? A.b2[0] : 0.0f;
#pragma unroll
        for (int dy = 0; dy < 3; ++dy)
#pragma unroll
            for (int dx = 0; dx < 3; ++dx) {
                const float* m = mid + (size_t)((oy + dy) * ST_MW + ox + dx) * ST_MSTRIDE;
                const float* w = w2s + (dy * 3 + dx) * 16;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float4 v = *(const float4*)(m + 4 * q), k = *(const float4*)(w + 4 * q);
                    s = fmaf(v.x, k.x, s);
                    s = fmaf(v.y, k.y, s);
                    s = fmaf(v.z, k.z, s);
                    s = fmaf(v.w, k.w, s);
                }
            }
        A.out[((size_t)b * H + y) * W + x] = s;
    }
}

__global__ __launch_bounds__(256, 2) void seg_tail_kernel(const SegTailArgs A, const float* __restrict__ w2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;
    float* mid = lds + ST_PATCH_FLOATS;
    float* w2s = mid + ST_MID_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int y0 = ty * ST_TH, x0 = tx * ST_TW;              // first output pixel of the tile
    const int H = A.H, W = A.W;

    // ---- source patch: the source pixels under up-sampled rows y0-2 .. y0+TH+1, cols x0-2 .. x0+TW+1 (clamped) --------
    const int sr0 = src_index(max(y0 - 2, 0), A.scale_h, A.Hs), sr1 = src_index(min(y0 + ST_TH + 1, H - 1), A.scale_h, A.Hs);
    const int sc0 = src_index(max(x0 - 2, 0), A.scale_w, A.Ws), sc1 = src_index(min(x0 + ST_TW + 1, W - 1), A.scale_w, A.Ws);
    const int nr = sr1 - sr0 + 1, nc = sc1 - sc0 + 1;        // <= ST_PR, ST_PC (host-checked)
    {
        const int nf4 = nr * nc * (ST_CIN / 4);
        for (int i = tid; i < nf4; i += 256) {
            const int p = i >> 3, j = i & 7;
            const int r = p / nc, cc = p - r * nc;
            const float4 v = *(const float4*)(A.x + ((size_t)(b * A.Hs + sr0 + r) * A.Ws + sc0 + cc) * A.in_cs + 4 * j);
            *(float4*)(patch + (size_t)(r * ST_PC + cc) * ST_PSTRIDE + 4 * j) = v;
        }
        if (tid < 144) w2s[tid] = w2[(tid & 15) * 9 + (tid >> 4)];       // second-conv weights as [tap][16]
        if (tid < ST_PSTRIDE / 4) *(float4*)(patch + (size_t)ST_PR * ST_PC * ST_PSTRIDE + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    // weight quads of the first conv, resident in registers
    float4 wq[18];
#pragma unroll
    for (int q = 0; q < 18; ++q) wq[q] = A.w1p[q * 64 + lane];
    float bias4[4], sc4[4], sh4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = A.b1[4 * g + r], sc4[r] = A.bn_scale[4 * g + r], sh4[r] = A.bn_shift[4 * g + r];
    __syncthreads();

    // ---- first conv on the matrix cores: 32 column tiles of 16 pixels (one mid row half each), 8 per wave, two at a time
    // (two independent accumulator chains keep the matrix pipe fed past the 8-pass latency of a dependent MFMA)
    const int zero_off = ST_PR * ST_PC * ST_PSTRIDE;
    // per-lane column offsets of the 3 taps for the two possible mid columns of this lane (left / right half row):
    // they do not depend on the tile, so they are worked out once (floats per pixel folded in; < 0 = zero padding)
    int cofs[2][3];
    bool cin[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int X = x0 - 1 + hx * 16 + c;
        cin[hx] = X >= 0 && X < W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = X - 1 + d;
            cofs[hx][d] = (xx >= 0 && xx < W) ? (src_index(xx, A.scale_w, A.Ws) - sc0) * ST_PSTRIDE + 4 * g : -1;
        }
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    // With an up-sampling ratio >= 2 the three input rows Y-1, Y, Y+1 of an output row come from at most TWO source rows, so
    // the kernel rows that read the same source row are merged (their weights added): 2 x 3 taps instead of 3 x 3 --
    // a third of the first conv's MFMAs gone. (Columns are not merged: the pattern would differ from lane to lane, and an
    // MFMA's weight operand is shared by its 16 pixels.) Rows touching the frame border keep the full 3 x 3 form.
    const bool can_merge = 2.0f * A.scale_h <= 1.0f;
#pragma unroll 1
#ifdef ST_ABL_NOMFMA
    for (int t0 = 2 * wave_u; t0 < 0; t0 += 8) {
#else
    for (int t0 = 2 * wave_u; t0 < ST_MH * 2; t0 += 8) {    // the pair (t0, t0+1) = the two halves of ONE mid row
#endif
        const int my = t0 >> 1;
        const int Y = y0 - 1 + my;                           // up-sampled row of this pair's mid pixels (wave-uniform)
        int rofs[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = Y - 1 + dy;
            rofs[dy] = (yy >= 0 && yy < H) ? (src_index(yy, A.scale_h, A.Hs) - sr0) * (ST_PC * ST_PSTRIDE) : -1;
        }
        bool inside[2];
        int mpos[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            inside[u] = Y >= 0 && Y < H && cin[u];
            mpos[u] = (my * ST_MW + u * 16 + c) * ST_MSTRIDE + 4 * g;
        }
        v4f acc0 = {bias4[0], bias4[1], bias4[2], bias4[3]}, acc1 = acc0;
        auto mfma8 = [&](const float4& a, const float4& p0, const float4& p1) {
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, p0.x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, p1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, p0.y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, p1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, p0.z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, p1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, p0.w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, p1.w, acc1, 0, 0, 0);
        };
        auto add4 = [](const float4& u, const float4& v) { return make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w); };
        const bool interior = rofs[0] >= 0 && rofs[2] >= 0;
        const bool va = rofs[1] == rofs[2], vb = rofs[0] == rofs[1];
        if (can_merge && interior && (va || vb)) {
            // variant A: rows 1,2 share a source row -> (w0 | w1+w2); variant B: rows 0,1 do -> (w0+w1 | w2)
            const int r0 = rofs[0], r1 = va ? rofs[1] : rofs[2];
            int off[2][6];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cf = cofs[u][dx];
                    off[u][dx] = cf >= 0 ? r0 + cf : zero_off + 4 * g;
                    off[u][3 + dx] = cf >= 0 ? r1 + cf : zero_off + 4 * g;
                }
            float4 p0 = *(const float4*)(patch + off[0][0]), p1 = *(const float4*)(patch + off[1][0]);
#pragma unroll
            for (int st = 0; st < 12; ++st) {
                const int tp = st >> 1, cb = st & 1, tn = (st + 1) >> 1, cbn = (st + 1) & 1;
                float4 n0 = p0, n1 = p1;
                if (st + 1 < 12) {
                    n0 = *(const float4*)(patch + off[0][tn] + 16 * cbn);
                    n1 = *(const float4*)(patch + off[1][tn] + 16 * cbn);
                }
                const int dx = tp % 3;
                // merged weight quad of (row group tp / 3, column dx, channel block cb)
                const float4 w0 = wq[(0 * 3 + dx) * 2 + cb], w1 = wq[(1 * 3 + dx) * 2 + cb], w2 = wq[(2 * 3 + dx) * 2 + cb];
                const float4 a = tp < 3 ? (va ? w0 : add4(w0, w1)) : (va ? add4(w1, w2) : w2);
                __builtin_amdgcn_sched_barrier(0);
                mfma8(a, p0, p1);
                __builtin_amdgcn_sched_barrier(0);
                p0 = n0, p1 = n1;
            }
        } else {
            int off[2][9];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int cf = cofs[u][dx];
                        off[u][dy * 3 + dx] = (rofs[dy] >= 0 && cf >= 0) ? rofs[dy] + cf : zero_off + 4 * g;
                    }
            // the B quads of step s+1 are read from LDS while step s's eight MFMAs run (pinned: left alone the compiler
            // issues each read right in front of its first use and every step waits out the LDS latency)
            float4 p0 = *(const float4*)(patch + off[0][0]), p1 = *(const float4*)(patch + off[1][0]);
#pragma unroll
            for (int st = 0; st < 18; ++st) {
                const int tn = (st + 1) >> 1, cbn = (st + 1) & 1;
                float4 n0 = p0, n1 = p1;
                if (st + 1 < 18) {
                    n0 = *(const float4*)(patch + off[0][tn] + 16 * cbn);
                    n1 = *(const float4*)(patch + off[1][tn] + 16 * cbn);
                }
                __builtin_amdgcn_sched_barrier(0);
                mfma8(wq[st], p0, p1);
                __builtin_amdgcn_sched_barrier(0);
                p0 = n0, p1 = n1;
            }
        }
        // ELU -> BatchNorm; a mid pixel outside the image is the second conv's zero padding
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const v4f acc = u ? acc1 : acc0;
            float4 o;
            float* op = &o.x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[r];
                v = elu_fast(v);
                op[r] = inside[u] ? v * sc4[r] + sh4[r] : 0.0f;
            }
            *(float4*)(mid + mpos[u]) = o;
        }
    }
    __syncthreads();

    seg_tail_second_conv(A, mid, w2s, y0, x0, b);
}

#ifndef OSSID_SEGTAIL_F32
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

// Packed weights of the split form: [15 operands][hi, lo][64 lanes] 16 bytes. Operand o = combo * 3 + dx with combo 0 / 1 / 2
// = kernel row ky, 3 = rows 0 + 1 merged, 4 = rows 1 + 2 merged (the row merging of the f32 kernel, done once here);
// lane (co = l % 16, g = l / 16) holds channels 8g .. 8g + 7 of W_o[co] as bf16.
constexpr int ST_SB_OPS = 15;
__global__ __launch_bounds__(256) void seg_tail_pack_sb_kernel(const float* __restrict__ w1, float4* __restrict__ w1p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= ST_SB_OPS * 2 * 64) return;
    const int lane = i & 63, part = (i >> 6) & 1, o = i >> 7, combo = o / 3, dx = o - combo * 3, g = lane >> 4, co = lane & 15;
    union {
        __bf16 b[8];
        float4 f;
    } u;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float* w = w1 + ((size_t)co * ST_CIN + 8 * g + e) * 9 + dx;
        const float v = combo < 3 ? w[combo * 3] : (combo == 3 ? w[0] + w[3] : w[3] + w[6]);
        const __bf16 hi = (__bf16)v;
        u.b[e] = part == 0 ? hi : (__bf16)(v - (float)hi);
    }
    w1p[i] = u.f;
}

__global__ __launch_bounds__(256, 2) void seg_tail_sb_kernel(const SegTailArgs A, const float* __restrict__ w2) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* patch = lds;                     // per source pixel ST_PSTRIDE floats: [hi: 32 x bf16][lo: 32 x bf16][16 bytes pad]
    float* mid = lds + ST_PATCH_FLOATS;
    float* w2s = mid + ST_MID_FLOATS;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, c = lane & 15;
    const int tile = blockIdx.x, b = blockIdx.y;
    const int ty = tile / A.tiles_x, tx = tile - ty * A.tiles_x;
    const int y0 = ty * ST_TH, x0 = tx * ST_TW;
    const int H = A.H, W = A.W;

    const int sr0 = src_index(max(y0 - 2, 0), A.scale_h, A.Hs), sr1 = src_index(min(y0 + ST_TH + 1, H - 1), A.scale_h, A.Hs);
    const int sc0 = src_index(max(x0 - 2, 0), A.scale_w, A.Ws), sc1 = src_index(min(x0 + ST_TW + 1, W - 1), A.scale_w, A.Ws);
    const int nr = sr1 - sr0 + 1, nc = sc1 - sc0 + 1;
    {
        const int nf4 = nr * nc * (ST_CIN / 4);
        for (int i = tid; i < nf4; i += 256) {
            const int p = i >> 3, j = i & 7;
            const int r = p / nc, cc = p - r * nc;
            const float4 v = *(const float4*)(A.x + ((size_t)(b * A.Hs + sr0 + r) * A.Ws + sc0 + cc) * A.in_cs + 4 * j);
            const float f[4] = {v.x, v.y, v.z, v.w};
            union {
                __bf16 h[4];
                uint2 u;
            } hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                hi.h[e] = (__bf16)f[e];
                lo.h[e] = (__bf16)(f[e] - (float)hi.h[e]);
            }
            char* px = (char*)(patch + (size_t)(r * ST_PC + cc) * ST_PSTRIDE);
            *(uint2*)(px + 8 * j) = hi.u;
            *(uint2*)(px + 64 + 8 * j) = lo.u;
        }
        if (tid < 144) w2s[tid] = w2[(tid & 15) * 9 + (tid >> 4)];
        if (tid < ST_PSTRIDE / 4) *(float4*)(patch + (size_t)ST_PR * ST_PC * ST_PSTRIDE + 4 * tid) = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    float4 wsb[ST_SB_OPS][2];               // every operand, hi and lo, resident in registers
#pragma unroll
    for (int o = 0; o < ST_SB_OPS; ++o)
#pragma unroll
        for (int part = 0; part < 2; ++part) wsb[o][part] = A.w1p[(o * 2 + part) * 64 + lane];
    float bias4[4], sc4[4], sh4[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) bias4[r] = A.b1[4 * g + r], sc4[r] = A.bn_scale[4 * g + r], sh4[r] = A.bn_shift[4 * g + r];
    __syncthreads();

    const int zero_off = ST_PR * ST_PC * ST_PSTRIDE;
    int cofs[2][3];
    bool cin[2];
#pragma unroll
    for (int hx = 0; hx < 2; ++hx) {
        const int X = x0 - 1 + hx * 16 + c;
        cin[hx] = X >= 0 && X < W;
#pragma unroll
        for (int d = 0; d < 3; ++d) {
            const int xx = X - 1 + d;
            cofs[hx][d] = (xx >= 0 && xx < W) ? (src_index(xx, A.scale_w, A.Ws) - sc0) * ST_PSTRIDE + 4 * g : -1;
        }
    }
    const int wave_u = __builtin_amdgcn_readfirstlane(wave);
    const bool can_merge = 2.0f * A.scale_h <= 1.0f;
#pragma unroll 1
    for (int t0 = 2 * wave_u; t0 < ST_MH * 2; t0 += 8) {
        const int my = t0 >> 1;
        const int Y = y0 - 1 + my;
        int rofs[3];
#pragma unroll
        for (int dy = 0; dy < 3; ++dy) {
            const int yy = Y - 1 + dy;
            rofs[dy] = (yy >= 0 && yy < H) ? (src_index(yy, A.scale_h, A.Hs) - sr0) * (ST_PC * ST_PSTRIDE) : -1;
        }
        bool inside[2];
        int mpos[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            inside[u] = Y >= 0 && Y < H && cin[u];
            mpos[u] = (my * ST_MW + u * 16 + c) * ST_MSTRIDE + 4 * g;
        }
        v4f acc0 = {bias4[0], bias4[1], bias4[2], bias4[3]}, acc1 = acc0;
        // one tap of both tiles: the operands of the NEXT tap are read by the caller while these six MFMAs run
        auto tap3 = [&](const float4& whi, const float4& wlo, const float4& h0, const float4& l0, const float4& h1, const float4& l1) {
            const v8bf ah = __builtin_bit_cast(v8bf, whi), al = __builtin_bit_cast(v8bf, wlo);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(v8bf, h0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, __builtin_bit_cast(v8bf, h1), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(v8bf, l0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(v8bf, l1), acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(v8bf, h0), acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, __builtin_bit_cast(v8bf, h1), acc1, 0, 0, 0);
        };
        const bool interior = rofs[0] >= 0 && rofs[2] >= 0;
        const bool va = rofs[1] == rofs[2], vb = rofs[0] == rofs[1];
        if (can_merge && interior && (va || vb)) {
            // variant A: rows 1,2 share a source row -> operands (row 0 | rows 1+2); B: rows 0,1 do -> (rows 0+1 | row 2)
            const int r0 = rofs[0], r1 = va ? rofs[1] : rofs[2];
            int off[2][6];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    const int cf = cofs[u][dx];
                    off[u][dx] = cf >= 0 ? r0 + cf : zero_off + 4 * g;
                    off[u][3 + dx] = cf >= 0 ? r1 + cf : zero_off + 4 * g;
                }
            float4 h0 = *(const float4*)(patch + off[0][0]), l0 = *(const float4*)(patch + off[0][0] + 16);
            float4 h1 = *(const float4*)(patch + off[1][0]), l1 = *(const float4*)(patch + off[1][0] + 16);
#pragma unroll
            for (int st = 0; st < 6; ++st) {
                const int sn = st + 1 < 6 ? st + 1 : st, dx = st % 3;
                const float4 nh0 = *(const float4*)(patch + off[0][sn]), nl0 = *(const float4*)(patch + off[0][sn] + 16);
                const float4 nh1 = *(const float4*)(patch + off[1][sn]), nl1 = *(const float4*)(patch + off[1][sn] + 16);
                // (wave-uniform choice between two resident operands: selects, no branch)
                const int oa = (st < 3 ? 0 : 4) * 3 + dx, ob = (st < 3 ? 3 : 2) * 3 + dx;
                float4 whi, wlo;
                whi.x = va ? wsb[oa][0].x : wsb[ob][0].x, whi.y = va ? wsb[oa][0].y : wsb[ob][0].y;
                whi.z = va ? wsb[oa][0].z : wsb[ob][0].z, whi.w = va ? wsb[oa][0].w : wsb[ob][0].w;
                wlo.x = va ? wsb[oa][1].x : wsb[ob][1].x, wlo.y = va ? wsb[oa][1].y : wsb[ob][1].y;
                wlo.z = va ? wsb[oa][1].z : wsb[ob][1].z, wlo.w = va ? wsb[oa][1].w : wsb[ob][1].w;
                __builtin_amdgcn_sched_barrier(0);
                tap3(whi, wlo, h0, l0, h1, l1);
                __builtin_amdgcn_sched_barrier(0);
                h0 = nh0, l0 = nl0, h1 = nh1, l1 = nl1;
            }
        } else {
            int off[2][9];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const int cf = cofs[u][dx];
                        off[u][dy * 3 + dx] = (rofs[dy] >= 0 && cf >= 0) ? rofs[dy] + cf : zero_off + 4 * g;
                    }
            float4 h0 = *(const float4*)(patch + off[0][0]), l0 = *(const float4*)(patch + off[0][0] + 16);
            float4 h1 = *(const float4*)(patch + off[1][0]), l1 = *(const float4*)(patch + off[1][0] + 16);
#pragma unroll
            for (int st = 0; st < 9; ++st) {
                const int sn = st + 1 < 9 ? st + 1 : st;
                const float4 nh0 = *(const float4*)(patch + off[0][sn]), nl0 = *(const float4*)(patch + off[0][sn] + 16);
                const float4 nh1 = *(const float4*)(patch + off[1][sn]), nl1 = *(const float4*)(patch + off[1][sn] + 16);
                __builtin_amdgcn_sched_barrier(0);
                tap3(wsb[st][0], wsb[st][1], h0, l0, h1, l1);          // operand (ky, dx) = combo ky, column dx = index st
                __builtin_amdgcn_sched_barrier(0);
                h0 = nh0, l0 = nl0, h1 = nh1, l1 = nl1;
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const v4f acc = u ? acc1 : acc0;
            float4 o;
            float* op = &o.x;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v = acc[r];
                v = elu_fast(v);
                op[r] = inside[u] ? v * sc4[r] + sh4[r] : 0.0f;
            }
            *(float4*)(mid + mpos[u]) = o;
        }
    }
    __syncthreads();
    seg_tail_second_conv(A, mid, w2s, y0, x0, b);
}
#endif

// w1 [16][32][3][3] (torch) -> [tap][cb][lane = g*16 + co] float4 of channels 16cb + 4g + 0..3
__global__ __launch_bounds__(256) void seg_tail_pack_kernel(const float* __restrict__ w1, float4* __restrict__ w1p) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= 18 * 64) return;
    const int lane = i & 63, q = i >> 6, cb = q & 1, tap = q >> 1, g = lane >> 4, co = lane & 15;
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = w1[((size_t)co * ST_CIN + 16 * cb + 4 * g + e) * 9 + tap];
    w1p[i] = make_float4(v[0], v[1], v[2], v[3]);
}

}  // namespace

extern "C" {

int ossid_seg_tail_split_bf16(void) {
#ifndef OSSID_SEGTAIL_F32
    return 1;
#else
    return 0;
#endif
}

#ifndef OSSID_SEGTAIL_F32
size_t ossid_seg_tail_packed_floats(void) { return ST_SB_OPS * 2 * 64 * 4; }

int ossid_seg_tail_pack_weights(const float* w1, float* w1p, void* stream) {
    if (!w1 || !w1p) return OSSID_EINVAL;
    hipLaunchKernelGGL(seg_tail_pack_sb_kernel, dim3((ST_SB_OPS * 2 * 64 + 255) / 256), dim3(256), 0, (hipStream_t)stream, w1,
                       (float4*)w1p);
    return ossid_launch_status();
}
#else
size_t ossid_seg_tail_packed_floats(void) { return 18 * 64 * 4; }

int ossid_seg_tail_pack_weights(const float* w1, float* w1p, void* stream) {
    if (!w1 || !w1p) return OSSID_EINVAL;
    hipLaunchKernelGGL(seg_tail_pack_kernel, dim3(5), dim3(256), 0, (hipStream_t)stream, w1, (float4*)w1p);
    return ossid_launch_status();
}
#endif

int ossid_seg_tail_fwd(const float* x, int batch, int src_height, int src_width, int in_channel_stride, int height,
                       int width, const float* w1p, const float* b1, const float* post_scale, const float* post_shift,
                       const float* w2, const float* b2, float* out, void* stream) {
    if (batch < 0 || height <= 0 || width <= 0 || src_height <= 0 || src_width <= 0 || src_height > height ||
        src_width > width || in_channel_stride < ST_CIN || (in_channel_stride % 4) || batch > 65535)
        return OSSID_EINVAL;
    if (batch == 0) return OSSID_OK;
    if (!x || !w1p || !b1 || !post_scale || !post_shift || !w2 || !out) return OSSID_EINVAL;
    SegTailArgs a;
    a.x = x, a.w1p = (const float4*)w1p, a.b1 = b1, a.bn_scale = post_scale, a.bn_shift = post_shift, a.w2 = w2, a.b2 = b2;
    a.out = out, a.H = height, a.W = width, a.Hs = src_height, a.Ws = src_width, a.in_cs = in_channel_stride;
    a.scale_h = (float)src_height / (float)height, a.scale_w = (float)src_width / (float)width;
    // the source footprint of an (TH+4) x (TW+4) up-sampled window must fit the staged patch
    const int need_r = (int)((ST_TH + 3) * a.scale_h) + 2, need_c = (int)((ST_TW + 3) * a.scale_w) + 2;
    if (need_r > ST_PR || need_c > ST_PC) return OSSID_EINVAL;
    a.tiles_x = (width + ST_TW - 1) / ST_TW, a.tiles_y = (height + ST_TH - 1) / ST_TH;
    const int lds = ST_LDS_FLOATS * 4;
#ifndef OSSID_SEGTAIL_F32
    OSSID_ENSURE_LDS(seg_tail_sb_kernel, (size_t)lds);
    hipLaunchKernelGGL(seg_tail_sb_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(256), lds, (hipStream_t)stream, a, w2);
#else
    OSSID_ENSURE_LDS(seg_tail_kernel, (size_t)lds);
    hipLaunchKernelGGL(seg_tail_kernel, dim3(a.tiles_x * a.tiles_y, batch), dim3(256), lds, (hipStream_t)stream, a, w2);
#endif
    return ossid_launch_status();
}

}  // extern "C"
