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
    float4 st[8];
    auto fetch = [&](int stage) {
        const long long p0 = (long long)stage * DB_PXS;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const long long p = min(p0 + (tid >> 5) + 8 * e, A.N - 1);           // (clamped: rows past the end are never stored)
            st[e] = *(const float4*)(A.dz + (size_t)p * DB_MID + 4 * j);
        }
    };
    auto commit = [&]() {
        uint2* p2 = (uint2*)zl;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int px = (tid >> 5) + 8 * e;
            const float v[4] = {st[e].x, st[e].y, st[e].z, st[e].w};
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
                           void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!dz || !wpk_dgrad || !x || !G || !alpha || !mask_scale || !mask_shift || !partials || n_rows <= 0 || c < 32 || (c % 32) ||
        c > 32 * 4 * DB_MAXT || channel_stride < c || (unsigned long long)n_rows * (unsigned long long)channel_stride >= (1ull << 32) || ((uintptr_t)dz & 15) || ((uintptr_t)wpk_dgrad & 15))
        return OSSID_EINVAL;
    DenseBwdArgs a;
    a.dz = dz, a.wpk = (const float4*)wpk_dgrad, a.x = x, a.G = G, a.alpha = alpha, a.ms = mask_scale, a.mt = mask_shift;
    a.partials = partials, a.N = n_rows, a.c = c, a.cs = channel_stride;
    const long long stages = (n_rows + DB_PXS - 1) / DB_PXS;
    if (stages > 0x7fffffff) return OSSID_EINVAL;
    a.nstages = (int)stages;
    hipLaunchKernelGGL(dense_dgrad1_acc_kernel, dim3((unsigned)ossid_dense_dgrad1_acc_partials(n_rows)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ossid_launch_status();
}

}  // extern "C"
