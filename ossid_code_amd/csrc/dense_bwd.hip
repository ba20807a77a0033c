// D16  The data gradient of a dense layer's 1x1 convolution with the block's gradient accumulation fused in (finetune step).
//
// torchvision's _DenseLayer (ImageFeatExtract, /root/reference/python/ossid/models/dtoid/network.py:164-184) reads the
// concatenation of everything before it: backward, the gradient dz [N][128] of its bottleneck goes through conv1 (c -> 128)
// to ALL c channels of the concatenation, through relu(norm1(.)) -- a mask and a per-channel scale -- and is ADDED to the
// gradient those channels have collected from the later layers; norm1's (d shift, d scale) are column sums of the masked
// gradient. Round 3 did that in two launches per layer: the 1x1 data gradient (csrc/conv.hip) wrote da [N][c], a generic pass
// (csrc/train.hip, chan_op) read da, the activations and the gradient buffer and wrote the gradient buffer: 5 c N floats of
// traffic per layer -- summed over a block's layers the O(L^2) term of the step, 8.3 GB and ~2 ms of kernel time at batch 8.
// Here da never exists: G[px][ch] += alpha[ch] * m * (dz[px][:] . W1[:, ch]),  m = (ms[ch] x[px][ch] + mt[ch] > 0), in the
// epilogue of the product, 3 c N floats of traffic, one launch.
//   product   TRANSPOSED, as csrc/dense.hip's shares: pixels on the MFMA's M axis (A = dz, staged once per 64-pixel stage into
//             LDS as split-bf16, rows [pixel][unit][hi 16 | lo 16]: an operand = one ds_read_b128), channels on N (B = conv1's
//             data-gradient layout from ossid_conv_pack_weights_dgrad, used as it is: the operand layouts of
//             v_mfma_f32_32x32x16_bf16 are symmetric), three bf16 products per f32 product, f32 accumulation
//   epilogue  a lane holds ONE channel and 16 pixels: activations / old gradient / new gradient are dword accesses, 128 bytes
//             per pixel row across the lanes (coalesced); the column sums (sum g m, sum g m x) are per-lane register sums --
//             no cross-lane reduction, which is what kept them out of csrc/conv.hip's epilogue (channels on M there)
//   grid      persistent workgroups walk the stages; wave w owns channel tiles w, w + 4, ...; one partial row [2][c] of the
//             column sums per workgroup, summed in a fixed order by ossid_bn_fold_bwd (bit-reproducible, no float atomics)
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

constexpr int DB_MID = 128, DB_PXS = 64;
constexpr int DB_PSTR = DB_MID / 16 * 4 + 1;            // float4 per pixel in LDS: 8 units x (hi, lo) x 2 halves + 1 of padding
constexpr int DB_MAXT = 8;                              // channel tiles per wave at most (c <= 1024)

struct DenseBwdArgs {
    const float* dz;          // [N][128]
    const float4* wpk;        // conv1 in the data-gradient layout: [c/32][8 units][2 parts][64 lanes] x 16 B
    const float* x;           // activations [N][cs] (the block's buffer)
    float* G;                 // gradient buffer [N][cs]: first c channels accumulated into
    const float *alpha, *ms, *mt;      // [c]
    float* partials;          // [gridDim.x][2][c]
    const float *za, *zb, *zk;         // optional: dz = dz + zb[k] * za + zk[k] while staging (za [N][128])
    long long N;
    int c, cs, nstages;
};

__global__ __launch_bounds__(256, 2) void dense_dgrad1_acc_kernel(const DenseBwdArgs A) {
    __shared__ __attribute__((aligned(16))) float4 zl[DB_PXS * DB_PSTR];     // 33 792 B (+ 16 KB of column sums below)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int ntiles = A.c / 32;

    // running column sums of this lane's channels, one slot per (wave, channel-tile turn, lane): only this lane touches its
    // slots (kept in LDS so that the loop over the channel tiles stays a LOOP: unrolled eight times it spilled 363 registers)
    __shared__ float sums[2][4 * DB_MAXT * 64];
#pragma unroll
    for (int k = 0; k < DB_MAXT; ++k) sums[0][(wave * DB_MAXT + k) * 64 + lane] = sums[1][(wave * DB_MAXT + k) * 64 + lane] = 0.0f;
    // staging map: 64 pixels x 32 float4 of dz, 8 per thread; channel quad j of a pixel -> unit j / 4, half-quad in the unit
    const int j = tid & 31;
    float4 st[8], sa[8];
    const bool has_add = A.za != nullptr;                                        // (uniform)
    float4 zb = make_float4(0.f, 0.f, 0.f, 0.f), zk = zb;
    if (has_add) zb = *(const float4*)(A.zb + 4 * j), zk = *(const float4*)(A.zk + 4 * j);
    auto fetch = [&](int stage) {
        const long long p0 = (long long)stage * DB_PXS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long p = min(p0 + (tid >> 5) + 8 * e, A.N - 1);           // (clamped: rows past the end are never stored)
            st[e] = *(const float4*)(A.dz + (size_t)p * DB_MID + 4 * j);
            if (has_add) sa[e] = *(const float4*)(A.za + (size_t)p * DB_MID + 4 * j);
        }
    };
    auto commit = [&]() {
        uint2* p2 = (uint2*)zl;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int px = (tid >> 5) + 8 * e;
            float v[4] = {st[e].x, st[e].y, st[e].z, st[e].w};
            if (has_add) {                                   // (g + c_x y) + c_1: the order of the generic pass this replaces
                v[0] = (v[0] + zb.x * sa[e].x) + zk.x, v[1] = (v[1] + zb.y * sa[e].y) + zk.y;
                v[2] = (v[2] + zb.z * sa[e].z) + zk.z, v[3] = (v[3] + zb.w * sa[e].w) + zk.w;
            }
            union {
                __bf16 b4[4];
                uint2 u2;
            } ph, pl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ph.b4[i] = (__bf16)v[i];
                pl.b4[i] = (__bf16)(v[i] - (float)ph.b4[i]);
            }
            p2[(px * DB_PSTR + u * 4) * 2 + jj] = ph.u2;
            p2[(px * DB_PSTR + u * 4) * 2 + 4 + jj] = pl.u2;
        }
    };
    if ((int)blockIdx.x < A.nstages) fetch(blockIdx.x);
    for (int stage = blockIdx.x; stage < A.nstages; stage += gridDim.x) {
        __syncthreads();                                   // the previous stage's readers are done
        commit();
        __syncthreads();
        if (stage + (int)gridDim.x < A.nstages) fetch(stage + gridDim.x);          // in flight under this stage's work
        const long long p0 = (long long)stage * DB_PXS;
#pragma unroll 1
        for (int k = 0; k < DB_MAXT; ++k) {
            const int ct = wave + 4 * k;
            if (ct >= ntiles) break;                       // (uniform per wave)
            const int ch = ct * 32 + n;
            const float al = A.alpha[ch], ms = A.ms[ch], mt = A.mt[ch];
            float s1 = 0.0f, s2 = 0.0f;
            float4 w[8][2];
#pragma unroll
            for (int u = 0; u < 8; ++u)
#pragma unroll
                for (int part = 0; part < 2; ++part) w[u][part] = A.wpk[(((size_t)ct * 8 + u) * 2 + part) * 64 + lane];
#pragma unroll 1
            for (int pt = 0; pt < 2; ++pt) {
                // old gradient and activations of this lane's channel for the tile's 16 pixel rows of its half: requested
                // before the product, used behind it (32-bit element offsets: rows x channel stride < 2^32, host-checked)
                float xv[16], gv[16];
                const unsigned rbase = (unsigned)p0 + pt * 32 + 4 * h;      // row of register r: rbase + 8 (r >> 2) + (r & 3)
                const unsigned nlast = (unsigned)(A.N - 1);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned off = min(rbase + 8 * (r >> 2) + (r & 3), nlast) * (unsigned)A.cs + ch;
                    xv[r] = A.x[off];
                    gv[r] = A.G[off];
                }
                v16f acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
                const float4* zp = zl + (size_t)(pt * 32 + n) * DB_PSTR + h;
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const v8bf ah = __builtin_bit_cast(v8bf, zp[u * 4]), alo = __builtin_bit_cast(v8bf, zp[u * 4 + 2]);
                    const v8bf bh = __builtin_bit_cast(v8bf, w[u][0]), bl = __builtin_bit_cast(v8bf, w[u][1]);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bh, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned row = rbase + 8 * (r >> 2) + (r & 3);
                    if (row > nlast) continue;
                    const float m = (ms * xv[r] + mt > 0.0f) ? 1.0f : 0.0f;
                    const float gm = acc[r] * m;
                    s1 += gm, s2 += gm * xv[r];
                    A.G[row * (unsigned)A.cs + ch] = al * acc[r] * m + gv[r];
                }
            }
            sums[0][(wave * DB_MAXT + k) * 64 + lane] += s1;
            sums[1][(wave * DB_MAXT + k) * 64 + lane] += s2;
        }
    }
    // column sums: the two halves of a wave hold the same channels (different pixel rows)
    float* prow = A.partials + (size_t)blockIdx.x * 2 * A.c;
#pragma unroll 1
    for (int k = 0; k < DB_MAXT; ++k) {
        const int ct = wave + 4 * k;
        if (ct >= ntiles) break;
        const float a1 = sums[0][(wave * DB_MAXT + k) * 64 + lane], a2 = sums[1][(wave * DB_MAXT + k) * 64 + lane];
        const float t1 = a1 + __shfl_xor(a1, 32), t2 = a2 + __shfl_xor(a2, 32);
        if (h == 0) prow[ct * 32 + n] = t1, prow[A.c + ct * 32 + n] = t2;
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// FORWARD of the same convolution with norm2's batch statistics in its epilogue: y1[px][co] = W1[co][:] . relu(s1 x[px][:] + t1)
// (c -> 128) and, per output channel, the sums of (y1 - pivot) and (y1 - pivot)^2 as partial rows for ossid_bn_fold_fwd --
// round 3 ran the convolution (csrc/conv.hip) and then a generic pass that read y1 again just for the two sums (58 launches
// and 0.9 GB per step). Same transposed scheme: pixels on M (A = the staged activations), output channels on N (B = conv1's
// FORWARD layout for the three-way split, ossid_conv_pack_weights_form(exact = 2)), so a lane holds one output channel and its
// sums are register sums. Arithmetic: the three-way split of csrc/conv.hip's FORM 2 (x = p0 + p1 + p2 in bf16, six products:
// f32-level accuracy -- a ReLU decides on these outputs in training), f32 accumulation.
// Workgroup = 64 pixels x 128 output channels (wave w = channels 32w ..), the reduction in chunks of 64 input channels through
// double-buffered LDS (rows [pixel][unit][p0 | p1 | p2 x 2 halves]); persistent over the 64-pixel stages, one partial row each.
constexpr int DF_KCH = 64;
constexpr int DF_PSTR = DF_KCH / 16 * 6 + 1;            // float4 per pixel and chunk: 4 units x 3 pieces x 2 halves + 1

struct DenseFwdArgs {
    const float* x;           // [N][cs], first c channels
    const float4* wpk;        // conv1, forward layout, three pieces: [4 tiles][c/16][3][64 lanes] x 16 B
    const float *ps, *pt;     // norm1 as scale / shift [c] (ReLU behind it)
    float* y1;                // [N][128]
    float* partials;          // [gridDim.x][3][128]: sums of (y1 - p), (y1 - p)^2, and p = the workgroup's own pivot
    float* counts;            // [gridDim.x]: pixels the workgroup summed over
    long long N;
    int c, cs, nstages;
};

// PT = 32-pixel tiles per stage: 2 where the pixels make more than ~1.5 stages of 64 per CU, else 1 (more, shorter workgroups)
template <int PT>
__global__ __launch_bounds__(256, 2) void dense_fwd1_stats_kernel(const DenseFwdArgs A) {
    constexpr int PXS = 32 * PT;
    __shared__ __attribute__((aligned(16))) float4 xl[2][PXS * DF_PSTR];     // 2 x 25 600 B at PT = 2
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int units = A.c / 16, nchunks = (A.c + DF_KCH - 1) / DF_KCH;
    const int co = wave * 32 + n;
    // The sums are taken about a PIVOT (no E[y^2] - E[y]^2 cancellation), and it has to be a value near the channel's mean that
    // is known before the first sum: each workgroup takes its own first output of the channel (a sample: within a few standard
    // deviations of the mean); ossid_bn_fold_fwd_rows moves every row's sums to one common pivot, in double. (A fixed pivot --
    // 0, or BatchNorm's running mean -- was measured: a fresh network's bottleneck outputs have |mean| >> std in some channels and
    // the statistics came out 1e-3 off.)
    float pv = 0.0f;
    bool have_pv = false;
    float s1 = 0.0f, s2 = 0.0f, npx = 0.0f;
    // staging map: 32 PT pixels x 16 float4 per chunk, 2 PT per thread
    const int j = tid & 15;
    float4 st[2 * PT];
    auto fetch = [&](long long p0, int ch0) {
        const bool ok = ch0 + 4 * j < A.c;                                       // (a ragged last chunk: c is a multiple of 32)
#pragma unroll
        for (int e = 0; e < 2 * PT; ++e) {
            const long long p = min(p0 + (tid >> 4) + 16 * e, A.N - 1);
            st[e] = ok ? *(const float4*)(A.x + (size_t)p * A.cs + ch0 + 4 * j) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto commit = [&](int buf, int ch0) {
        const bool ok = ch0 + 4 * j < A.c;
        float4 ps = make_float4(0.f, 0.f, 0.f, 0.f), pt = ps;
        if (ok) ps = *(const float4*)(A.ps + ch0 + 4 * j), pt = *(const float4*)(A.pt + ch0 + 4 * j);
        uint2* p2 = (uint2*)xl[buf];
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < 2 * PT; ++e) {
            const int px = (tid >> 4) + 16 * e;
            float v[4] = {fmaxf(st[e].x * ps.x + pt.x, 0.f), fmaxf(st[e].y * ps.y + pt.y, 0.f), fmaxf(st[e].z * ps.z + pt.z, 0.f),
                          fmaxf(st[e].w * ps.w + pt.w, 0.f)};
            union {
                __bf16 b4[4];
                uint2 u2;
            } pc[3];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float r = v[i];
#pragma unroll
                for (int k = 0; k < 3; ++k) {
                    pc[k].b4[i] = (__bf16)r;
                    r -= (float)pc[k].b4[i];
                }
            }
#pragma unroll
            for (int k = 0; k < 3; ++k) p2[(px * DF_PSTR + u * 6 + k * 2) * 2 + jj] = pc[k].u2;
        }
    };
    const float4* W4 = A.wpk + (size_t)wave * units * 3 * 64 + lane;
    for (int stage = blockIdx.x; stage < A.nstages; stage += gridDim.x) {
        const long long p0 = (long long)stage * PXS;
        v16f acc[PT];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[pt][r] = 0.0f;
        fetch(p0, 0);
        // this wave's weights ride one chunk ahead of their MFMAs in registers (4 units x 3 pieces; units past the last one
        // re-read the last: never multiplied)
        float4 wc[4][3], wn[4][3];
        auto wload = [&](int ch, float4 (&w)[4][3]) {
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const float4* wq = W4 + (size_t)min(ch * 4 + u, units - 1) * 3 * 64;
#pragma unroll
                for (int k = 0; k < 3; ++k) w[u][k] = wq[k * 64];
            }
        };
        wload(0, wc);
        __syncthreads();                                   // (the previous stage's last chunk has been read)
        commit(0, 0);
#pragma unroll 1
        for (int ch = 0; ch < nchunks; ++ch) {
            __syncthreads();
            if (ch + 1 < nchunks) {
                fetch(p0, (ch + 1) * DF_KCH);              // in flight under this chunk's MFMAs
                wload(ch + 1, wn);
            }
            const int nu = min(4, units - ch * 4);
            const float4* zb = xl[ch & 1] + (size_t)n * DF_PSTR + h;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (u >= nu) break;
                const v8bf b0 = __builtin_bit_cast(v8bf, wc[u][0]), b1 = __builtin_bit_cast(v8bf, wc[u][1]), b2 = __builtin_bit_cast(v8bf, wc[u][2]);
#pragma unroll
                for (int pt = 0; pt < PT; ++pt) {
                    const float4* zp = zb + (size_t)pt * 32 * DF_PSTR + u * 6;
                    const v8bf a0 = __builtin_bit_cast(v8bf, zp[0]), a1 = __builtin_bit_cast(v8bf, zp[2]), a2 = __builtin_bit_cast(v8bf, zp[4]);
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc[pt], 0, 0, 0);      // smallest terms first
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[pt], 0, 0, 0);
                    acc[pt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[pt], 0, 0, 0);
                }
            }
            if (ch + 1 < nchunks) {
                commit((ch + 1) & 1, (ch + 1) * DF_KCH);
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int k = 0; k < 3; ++k) wc[u][k] = wn[u][k];
            }
        }
        if (!have_pv) {                                    // (uniform: the workgroup's first stage; its row p0 exists)
            pv = __shfl(acc[0][0], n);                     // lane n (h = 0) holds pixel row p0 of channel co: both halves take it
            have_pv = true;
        }
        npx += (float)min((long long)PXS, A.N - p0);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const long long row = p0 + pt * 32 + 8 * (r >> 2) + 4 * h + (r & 3);
                if (row >= A.N) continue;
                const float v = acc[pt][r];
                A.y1[(size_t)row * DB_MID + co] = v;
                const float d = v - pv;
                s1 += d, s2 += d * d;
            }
    }
    const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
    if (h == 0) {
        float* prow = A.partials + (size_t)blockIdx.x * 3 * DB_MID;
        prow[co] = t1, prow[DB_MID + co] = t2, prow[2 * DB_MID + co] = pv;
    }
    if (tid == 0) A.counts[blockIdx.x] = npx;
}

// Training BatchNorm folded to (scale, shift) from partial rows that each carry their OWN pivot: rows [P][3][C] = (sum (x - p_w),
// sum (x - p_w)^2, p_w), counts [P] = elements per row. Every row is moved to the common pivot q = p_0 in double --
// S1' = S1 + n_w d, S2' = S2 + 2 d S1 + n_w d^2 with d = p_w - q -- and the rest is ossid_bn_fold_fwd's arithmetic.
__global__ __launch_bounds__(256) void bn_fold_fwd_rows_kernel(const float* __restrict__ partials, const float* __restrict__ counts, int P,
                                                               int C, double n, const float* __restrict__ gamma,
                                                               const float* __restrict__ beta, float eps, float momentum,
                                                               float* __restrict__ running_mean, float* __restrict__ running_var,
                                                               float* __restrict__ scale, float* __restrict__ shift,
                                                               float* __restrict__ mean_out, float* __restrict__ rstd_out) {
    // 4 channels x 64 row ranges per block: a thread's <= 8 rows are loaded together (this launch sits on the step's critical
    // chain 58 times: its time is the number of dependent rounds of loads)
    __shared__ double red[64][4][2];
    const int col = threadIdx.x & 3, part = threadIdx.x >> 2;
    const int c = blockIdx.x * 4 + col;
    double a1 = 0.0, a2 = 0.0, q = 0.0;
    if (c < C) {
        q = (double)partials[2 * (size_t)C + c];
        for (int w0 = part; w0 < P; w0 += 64 * 8) {
            float s1[8], s2[8], pw[8], nw[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int w = w0 + 64 * k;
                const bool ok = w < P;
                const float* row = partials + (size_t)(ok ? w : 0) * 3 * C + c;
                s1[k] = row[0], s2[k] = row[C], pw[k] = row[2 * (size_t)C], nw[k] = ok ? counts[w] : -1.0f;
            }
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                if (nw[k] < 0.0f) continue;
                const double d = (double)pw[k] - q;
                a1 += (double)s1[k] + (double)nw[k] * d;
                a2 += (double)s2[k] + 2.0 * d * (double)s1[k] + (double)nw[k] * d * d;
            }
        }
    }
    red[part][col][0] = a1, red[part][col][1] = a2;
    __syncthreads();
    if (part != 0 || c >= C) return;
    for (int k = 1; k < 64; ++k) a1 += red[k][col][0], a2 += red[k][col][1];
    const double dm = a1 / n;
    const double mean = q + dm;
    double var = a2 / n - dm * dm;
    if (var < 0.0) var = 0.0;
    const double rstd = 1.0 / sqrt(var + (double)eps);
    const float g = gamma ? gamma[c] : 1.0f, b = beta ? beta[c] : 0.0f;
    const float sc = (float)((double)g * rstd);
    scale[c] = sc;
    shift[c] = (float)((double)b - mean * (double)sc);
    mean_out[c] = (float)mean;
    rstd_out[c] = (float)rstd;
    if (running_mean) {   // torch: running = (1 - momentum) * running + momentum * batch (unbiased variance)
        const double unb = n > 1.0 ? var * n / (n - 1.0) : var;
        running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * mean);
        running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unb);
    }
}

int g_df_grid = 0;
int df_grid() {
    if (!g_df_grid) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_fwd1_stats_kernel<2>, 256, 0) != hipSuccess || per_cu <= 0)
            return 512;
        if (per_cu > 2) per_cu = 2;
        g_df_grid = per_cu * p.multiProcessorCount;
    }
    return g_df_grid;
}
// pixels per stage for n_rows: 64, or 32 where 64-pixel stages would leave the chip under ~1.5 workgroups per CU
int df_pxs(long long n_rows) { return (n_rows + 63) / 64 >= 384 ? 64 : 32; }

// ---------------------------------------------------------------------------------------------------------------------
// The 3x3 data gradient of a dense layer (32 -> 128: the gradient of the layer's own 32 channels back to its bottleneck) with
// norm2 / ReLU's backward in its epilogue: db[px][ch] = alpha[ch] m (g * W2')[px][ch], m = (ms[ch] y1[px][ch] + mt[ch] > 0), and
// the column sums (sum g m, sum g m y1) as partial rows for ossid_bn_fold_bwd. Round 3: the convolution (Winograd or direct),
// then a generic pass reading its output and y1 and writing it again. Transposed product once more: pixels on M (A = the staged
// gradient patch of a 4 x 8 tile + halo, 32 channels = 2 units, every tap read from it), output channels on N (B = conv2's
// data-gradient layout, ossid_conv_pack_weights_dgrad: 36 operand quads per wave, loaded ONCE and kept in registers by the
// persistent workgroup); wave w owns output channels 32w .. 32w+31 of the tile.
constexpr int D3_TR = 4, D3_TC = 8, D3_PR = D3_TR + 2, D3_PC = D3_TC + 2, D3_NPOS = D3_PR * D3_PC;
constexpr int D3_PSTR = 2 * 4 + 1;                      // float4 per patch position: 2 units x (hi, lo) x 2 halves + 1

struct DenseBwd3Args {
    const float* g;           // gradient of the layer's 32 channels: [N][gcs] (a channel slice of the block's gradient buffer)
    const float4* wpk;        // conv2, data-gradient layout: [4 tiles][2 units][9 taps][2 parts][64 lanes] x 16 B
    const float* y1;          // [N][128] bottleneck pre-activations
    float* db;                // [N][128] out
    const float *alpha, *ms, *mt;      // [128]
    float* partials;          // [gridDim.x][2][128]
    int B, H, W, gcs, tiles_x, tiles_y, ntiles;
};

__global__ __launch_bounds__(256, 2) void dense_dgrad3_mask_kernel(const DenseBwd3Args A) {
    __shared__ __attribute__((aligned(16))) float4 patch[64 * D3_PSTR];       // 60 positions (+ 4 spare rows for the staging map)
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int H = A.H, W = A.W;
    const int ch = wave * 32 + n;
    const float al = A.alpha[ch], ms = A.ms[ch], mt = A.mt[ch];
    float s1 = 0.0f, s2 = 0.0f;
    float4 w[2][9][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int part = 0; part < 2; ++part) w[u][tap][part] = A.wpk[((((size_t)wave * 2 + u) * 9 + tap) * 2 + part) * 64 + lane];
    // staging map: 60 positions x 8 float4 = 480 quads, two per thread (the map's last 32 slots fall on the spare rows)
    const int j = tid & 7;
    float4 st[2];
    auto geom = [&](int tile, int& b, int& y0, int& x0) {
        const int per = A.tiles_x * A.tiles_y;
        b = tile / per;
        const int r = tile - b * per;
        y0 = (r / A.tiles_x) * D3_TR, x0 = (r % A.tiles_x) * D3_TC;
    };
    auto fetch = [&](int tile) {
        int b, y0, x0;
        geom(tile, b, y0, x0);
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int pos = min((tid >> 3) + 32 * e, D3_NPOS - 1);
            const int pr = pos / D3_PC, pc = pos - pr * D3_PC;
            const int yc = min(max(y0 - 1 + pr, 0), H - 1), xc = min(max(x0 - 1 + pc, 0), W - 1);
            st[e] = *(const float4*)(A.g + ((size_t)(b * H + yc) * W + xc) * A.gcs + 4 * j);
        }
    };
    auto commit = [&](int tile) {
        int b, y0, x0;
        geom(tile, b, y0, x0);
        uint2* p2 = (uint2*)patch;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < 2; ++e) {
            const int pos = (tid >> 3) + 32 * e;
            const int pr = pos / D3_PC, pc = pos - pr * D3_PC;
            const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
            const float f = (pos < D3_NPOS && yy >= 0 && yy < H && xx >= 0 && xx < W) ? 1.0f : 0.0f;      // zero padding
            const float v[4] = {f * st[e].x, f * st[e].y, f * st[e].z, f * st[e].w};
            union {
                __bf16 b4[4];
                uint2 u2;
            } ph, pl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ph.b4[i] = (__bf16)v[i];
                pl.b4[i] = (__bf16)(v[i] - (float)ph.b4[i]);
            }
            p2[(pos * D3_PSTR + u * 4) * 2 + jj] = ph.u2;
            p2[(pos * D3_PSTR + u * 4) * 2 + 4 + jj] = pl.u2;
        }
    };
    const int p0 = (n >> 3) * D3_PC + (n & 7);           // patch position of tap (0, 0) for this lane's pixel (as the A operand's row)
    if ((int)blockIdx.x < A.ntiles) fetch(blockIdx.x);
    for (int tile = blockIdx.x; tile < A.ntiles; tile += gridDim.x) {
        __syncthreads();
        commit(tile);
        __syncthreads();
        if (tile + (int)gridDim.x < A.ntiles) fetch(tile + gridDim.x);
        int b, y0, x0;
        geom(tile, b, y0, x0);
        // this lane's channel of the tile's 16 pixel rows of its half: y1 requested before the product
        float yv[16];
        unsigned valid = 0;
        int rowi[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = 8 * (r >> 2) + 4 * h + (r & 3);                      // pixel of the tile: row m >> 3, column m & 7
            const int yy = y0 + (m >> 3), xx = x0 + (m & 7);
            valid |= (yy < H && xx < W) ? 1u << r : 0u;
            rowi[r] = (b * H + min(yy, H - 1)) * W + min(xx, W - 1);
            yv[r] = A.y1[(size_t)rowi[r] * DB_MID + ch];
        }
        v16f acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
        const float4* pb = patch + (size_t)p0 * D3_PSTR + h;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const float4* px4 = pb + ((tap / 3) * D3_PC + tap % 3) * D3_PSTR + u * 4;
                const v8bf ah = __builtin_bit_cast(v8bf, px4[0]), alo = __builtin_bit_cast(v8bf, px4[2]);
                const v8bf bh = __builtin_bit_cast(v8bf, w[u][tap][0]), bl = __builtin_bit_cast(v8bf, w[u][tap][1]);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, bh, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
                acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
            }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            if (!(valid >> r & 1)) continue;
            const float m = (ms * yv[r] + mt > 0.0f) ? 1.0f : 0.0f;
            const float gm = acc[r] * m;
            s1 += gm, s2 += gm * yv[r];
            A.db[(size_t)rowi[r] * DB_MID + ch] = al * acc[r] * m;
        }
    }
    const float t1 = s1 + __shfl_xor(s1, 32), t2 = s2 + __shfl_xor(s2, 32);
    if (h == 0) {
        float* prow = A.partials + (size_t)blockIdx.x * 2 * DB_MID;
        prow[ch] = t1, prow[DB_MID + ch] = t2;
    }
}

int g_d3_grid = 0;
int d3_grid() {
    if (!g_d3_grid) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_dgrad3_mask_kernel, 256, 0) != hipSuccess || per_cu <= 0)
            return 512;
        if (per_cu > 2) per_cu = 2;
        g_d3_grid = per_cu * p.multiProcessorCount;
    }
    return g_d3_grid;
}

int g_db_grid = 0;
int db_grid() {
    if (!g_db_grid) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, dense_dgrad1_acc_kernel, 256, 0) != hipSuccess || per_cu <= 0)
            return 512;
        if (per_cu > 2) per_cu = 2;
        g_db_grid = per_cu * p.multiProcessorCount;
    }
    return g_db_grid;
}

}  // namespace

extern "C" {

// partial rows [P][2][c] the launch leaves for ossid_bn_fold_bwd (row 0 = sum g m, row 1 = sum g m x)
int ossid_dense_dgrad1_acc_partials(long long n_rows) {
    const long long stages = (n_rows + DB_PXS - 1) / DB_PXS;
    const int g = db_grid();
    return (int)(stages < g ? stages : g);
}

int ossid_dense_dgrad1_acc(const float* dz, const float* wpk_dgrad, const float* x, float* G, long long n_rows, int c,
                           int channel_stride, const float* alpha, const float* mask_scale, const float* mask_shift, float* partials,
                           const float* dz_add, const float* dz_add_scale, const float* dz_add_shift, void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!dz || !wpk_dgrad || !x || !G || !alpha || !mask_scale || !mask_shift || !partials || n_rows <= 0 || c < 32 || (c % 32) ||
        c > 32 * 4 * DB_MAXT || channel_stride < c || (unsigned long long)n_rows * (unsigned long long)channel_stride >= (1ull << 32) || ((uintptr_t)dz & 15) || ((uintptr_t)wpk_dgrad & 15))
        return OSSID_EINVAL;
    DenseBwdArgs a;
    a.dz = dz, a.wpk = (const float4*)wpk_dgrad, a.x = x, a.G = G, a.alpha = alpha, a.ms = mask_scale, a.mt = mask_shift;
    if (dz_add && (!dz_add_scale || !dz_add_shift || ((uintptr_t)dz_add & 15) || ((uintptr_t)dz_add_scale & 15) ||
                   ((uintptr_t)dz_add_shift & 15)))
        return OSSID_EINVAL;
    a.za = dz_add, a.zb = dz_add_scale, a.zk = dz_add_shift;
    a.partials = partials, a.N = n_rows, a.c = c, a.cs = channel_stride;
    const long long stages = (n_rows + DB_PXS - 1) / DB_PXS;
    if (stages > 0x7fffffff) return OSSID_EINVAL;
    a.nstages = (int)stages;
    hipLaunchKernelGGL(dense_dgrad1_acc_kernel, dim3((unsigned)ossid_dense_dgrad1_acc_partials(n_rows)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ossid_launch_status();
}

int ossid_dense_fwd1_stats_partials(long long n_rows) {
    const int pxs = df_pxs(n_rows);
    const long long stages = (n_rows + pxs - 1) / pxs;
    const int g = df_grid();
    return (int)(stages < g ? stages : g);
}

int ossid_bn_fold_fwd_rows(const float* partials, const float* counts, int n_partials, int C, double n, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var, float* scale,
                           float* shift, float* mean_out, float* rstd_out, void* stream) {
    if (!partials || !counts || n_partials <= 0 || C <= 0 || n <= 0 || !scale || !shift || !mean_out || !rstd_out ||
        (!running_mean != !running_var))
        return OSSID_EINVAL;
    hipLaunchKernelGGL(bn_fold_fwd_rows_kernel, dim3((C + 3) / 4), dim3(256), 0, (hipStream_t)stream, partials, counts, n_partials, C,
                       n, gamma, beta, eps, momentum, running_mean, running_var, scale, shift, mean_out, rstd_out);
    return ossid_launch_status();
}

int ossid_dense_fwd1_stats(const float* x, int channel_stride, int c, const float* pre_scale, const float* pre_shift,
                           const float* wpk_x6, long long n_rows, float* y1, float* partials, float* counts, void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!x || !pre_scale || !pre_shift || !wpk_x6 || !y1 || !partials || !counts || n_rows <= 0 || c < 32 || (c % 32) || channel_stride < c ||
        (channel_stride % 4) || ((uintptr_t)x & 15) || ((uintptr_t)wpk_x6 & 15) || ((uintptr_t)pre_scale & 15) || ((uintptr_t)pre_shift & 15))
        return OSSID_EINVAL;
    DenseFwdArgs a;
    a.x = x, a.wpk = (const float4*)wpk_x6, a.ps = pre_scale, a.pt = pre_shift, a.y1 = y1, a.partials = partials, a.counts = counts;
    a.N = n_rows, a.c = c, a.cs = channel_stride;
    const int pxs = df_pxs(n_rows);
    const long long stages = (n_rows + pxs - 1) / pxs;
    if (stages > 0x7fffffff) return OSSID_EINVAL;
    a.nstages = (int)stages;
    const unsigned grid = (unsigned)ossid_dense_fwd1_stats_partials(n_rows);
    if (pxs == 64) hipLaunchKernelGGL(dense_fwd1_stats_kernel<2>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL(dense_fwd1_stats_kernel<1>, dim3(grid), dim3(256), 0, (hipStream_t)stream, a);
    return ossid_launch_status();
}

static long long d3_tiles(int B, int H, int W) { return (long long)B * ((H + D3_TR - 1) / D3_TR) * ((W + D3_TC - 1) / D3_TC); }

int ossid_dense_dgrad3_mask_partials(int B, int H, int W) {
    const long long nt = d3_tiles(B, H, W);
    const int g = d3_grid();
    return (int)(nt < g ? nt : g);
}

int ossid_dense_dgrad3_mask(const float* g, int g_channel_stride, const float* wpk_dgrad, const float* y1, float* db, int B, int H,
                            int W, const float* alpha, const float* mask_scale, const float* mask_shift, float* partials,
                            void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!g || !wpk_dgrad || !y1 || !db || !alpha || !mask_scale || !mask_shift || !partials || B <= 0 || H <= 0 || W <= 0 ||
        g_channel_stride < 32 || (g_channel_stride % 4) || ((uintptr_t)g & 15) || ((uintptr_t)wpk_dgrad & 15))
        return OSSID_EINVAL;
    const long long nt = d3_tiles(B, H, W);
    if (nt > 0x7fffffff || (long long)B * H * W * DB_MID >= (1LL << 31)) return OSSID_EINVAL;
    DenseBwd3Args a;
    a.g = g, a.wpk = (const float4*)wpk_dgrad, a.y1 = y1, a.db = db, a.alpha = alpha, a.ms = mask_scale, a.mt = mask_shift;
    a.partials = partials, a.B = B, a.H = H, a.W = W, a.gcs = g_channel_stride;
    a.tiles_x = (W + D3_TC - 1) / D3_TC, a.tiles_y = (H + D3_TR - 1) / D3_TR, a.ntiles = (int)nt;
    hipLaunchKernelGGL(dense_dgrad3_mask_kernel, dim3((unsigned)ossid_dense_dgrad3_mask_partials(B, H, W)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ossid_launch_status();
}

}  // extern "C"
