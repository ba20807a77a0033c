// D16  weight gradient of the decoder's two few-channel 3x3 layers in the finetune step -- s4: 64 -> 32 on 232 x 312, s5: 32 -> 16
// on 480 x 640, both behind a nearest-neighbour up-sampling and a folded BatchNorm (reference models/dtoid/network.py:354-360,
// run backward by scripts/online_learning.py:668) -- on the exact-f32 matrix cores, from 2-D pixel tiles.
//
//   dW[co][ci][ky][kx] = sum_px dY[px][co] * P(up(X))[px + (ky-1, kx-1)][ci],   P = per-channel affine (+ReLU), zero outside
//
// Why a kernel of their own: with 16 / 32 output channels the general weight-gradient kernel (csrc/train.hip) is bound by
// staging, not arithmetic -- a workgroup there owns ONE image row and ONE kernel row, so every input row is fetched, prologue'd
// and split three times (0.44 + 0.69 ms at batch 8, 33-49 TFLOP/s). Here a workgroup owns a 4 x 32 pixel tile: the 6 x 34 input
// pixels under it are staged ONCE (through the up-sampling's index map, with the prologue), all nine taps read that patch from
// LDS, and the pixels sit on the MFMA's K axis (pairs / quads of neighbouring pixels per instruction):
//   Cout 32: v_mfma_f32_32x32x2_f32, A = dY [pixel pair][32 co], B = patch [pixel pair + tap][32 ci]: 9 taps x Cin/32 column tiles
//   Cout 16: v_mfma_f32_16x16x4_f32, A = dY [pixel quad][16 co], B = patch [pixel quad + tap][16 ci]: 9 taps x Cin/16 column tiles
// The column tiles are dealt to the four waves (each wave walks ALL pixels of the tile for its tiles: no cross-wave sums);
// every operand is one conflict-free ds_read_b32. Persistent workgroups keep their accumulators over many tiles (the next
// tile's loads in flight under this tile's MFMAs) and write ONE partial [Cout][Cin][9] slab each; a second kernel adds the
// slabs in a fixed order (bit-reproducible). Exact f32: 22.6 / 21.3 GFLOP = 0.15 ms each of matrix pipe; measured 0.29 / 0.27 ms.
// (Measured and NOT adopted: the same kernel at 128 -> 32 for the dense blocks' 3x3 layers, grouped per block -- 2.27 ms over the
// four blocks against 1.64 ms for the split-bf16 grouped kernel of csrc/train.hip, tools/wgrad_group_bench.py: with 128 input
// channels that kernel's staging is amortised over four column tiles per tap and the 3-product arithmetic wins.)
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

struct FcArgs {
    const float *x, *dy, *pre_scale, *pre_shift;
    float* slabs;
    int B, H, W, Hs, Ws, in_cs, dy_cs, pre_relu, tiles_y, tiles_x, ntiles;
    float scale_h, scale_w;
};

constexpr int FC_TW = 32, FC_PW = FC_TW + 2;
constexpr int fc_th(int cin) { return cin >= 128 ? 2 : 4; }      // rows per tile: the patch must leave room for two workgroups per CU

// workgroup `wg` of `nwg` that share the problem: tiles wg, wg + nwg, ...; one partial slab at index wg
template <int COUT, int CIN>
__device__ __forceinline__ void fewch_body(const FcArgs& a, const int wg, const int nwg) {
    constexpr int FC_TH = fc_th(CIN), FC_PH = FC_TH + 2;
    constexpr bool SMALL = COUT == 16;                   // 16x16x4 tiles; else 32x32x2
    constexpr int TN = SMALL ? 16 : 32;                  // columns (input channels) per tile
    constexpr int NTILES = 9 * (CIN / TN);               // (tap, channel block) column tiles
    constexpr int NTW = (NTILES + 3) / 4;                // per wave
    constexpr int X4 = CIN / 4, D4 = COUT / 4;
    constexpr int NX = (FC_PH * FC_PW * X4 + 255) / 256, ND = (FC_TH * FC_TW * D4 + 255) / 256;
    __shared__ float4 xs4[FC_PH * FC_PW * X4];           // patch [py][px][ci]
    __shared__ float4 dys4[FC_TH * FC_TW * D4];          // dY tile [pixel][co]
    const float* xs = (const float*)xs4;
    const float* dys = (const float*)dys4;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n = SMALL ? (lane & 15) : (lane & 31), kk = SMALL ? (lane >> 4) : (lane >> 5);
    constexpr int KP = SMALL ? 4 : 2;                    // pixels per MFMA
    // this wave's column tiles: t = wave + 4 i -> tap = t / (CIN / TN), channel block = t % (CIN / TN)
    int boff[NTW];                                       // patch offset (floats) of the tile's tap and channel, lane part included
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int t = min(wave + 4 * i, NTILES - 1);
        const int tap = t / (CIN / TN), cb = t % (CIN / TN);
        boff[i] = ((tap / 3) * FC_PW + (tap % 3) + kk) * CIN + cb * TN + n;
    }
    v16f acc32[SMALL ? 1 : NTW];
    v4f acc16[SMALL ? NTW : 1];
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        if constexpr (SMALL) acc16[i] = v4f{0.f, 0.f, 0.f, 0.f};
        else
#pragma unroll
            for (int r = 0; r < 16; ++r) acc32[i][r] = 0.0f;
    }
    const int xq = tid % X4, dq = tid % D4;
    float4 ps = make_float4(1.f, 1.f, 1.f, 1.f), pt = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.pre_scale) {
        ps = make_float4(a.pre_scale[4 * xq], a.pre_scale[4 * xq + 1], a.pre_scale[4 * xq + 2], a.pre_scale[4 * xq + 3]);
        pt = make_float4(a.pre_shift[4 * xq], a.pre_shift[4 * xq + 1], a.pre_shift[4 * xq + 2], a.pre_shift[4 * xq + 3]);
    }
    // With 128 input channels the 17 staging registers of a prefetched patch do not fit beside the 144 accumulators: that
    // shape stages at the top of its tile, in batches of NB loads (two workgroups per CU cover each other's loads instead).
    constexpr bool PF = CIN < 128;
    constexpr int NB = PF ? NX : 6;
    float4 sx[NB], sd[ND];
    auto fetch = [&](int tile, int e0) {                     // patch elements e0 .. e0 + NB - 1 (and, at e0 == 0, the dY tile)
        const int tx = tile % a.tiles_x, r1 = tile / a.tiles_x;
        const int b = r1 / a.tiles_y, oy0 = (r1 % a.tiles_y) * FC_TH, ox0 = tx * FC_TW;
#pragma unroll
        for (int ee = 0; ee < NB; ++ee) {
            const int e = e0 + ee;
            const int idx = (tid + e * 256) / X4;            // patch pixel
            const int py = idx / FC_PW, px = idx - py * FC_PW;
            const int yy = oy0 - 1 + py, xx = ox0 - 1 + px;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (e < NX && idx < FC_PH * FC_PW && yy >= 0 && yy < a.H && xx >= 0 && xx < a.W) {
                const int sy = (a.Hs == a.H) ? yy : min((int)floorf((float)yy * a.scale_h), a.Hs - 1);
                const int sxx = (a.Ws == a.W) ? xx : min((int)floorf((float)xx * a.scale_w), a.Ws - 1);
                v = *(const float4*)(a.x + ((size_t)(b * a.Hs + sy) * a.Ws + sxx) * a.in_cs + 4 * xq);
                if (a.pre_scale) {
                    v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                    if (a.pre_relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                }
            }
            sx[ee] = v;
        }
        if (e0 != 0) return;
#pragma unroll
        for (int e = 0; e < ND; ++e) {
            const int idx = (tid + e * 256) / D4;            // tile pixel
            const int py = idx / FC_TW, px = idx - py * FC_TW;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (idx < FC_TH * FC_TW && oy0 + py < a.H && ox0 + px < a.W)
                v = *(const float4*)(a.dy + ((size_t)(b * a.H + oy0 + py) * a.W + ox0 + px) * a.dy_cs + 4 * dq);
            sd[e] = v;
        }
    };
    auto commit = [&](int e0) {
#pragma unroll
        for (int ee = 0; ee < NB; ++ee)
            if (e0 + ee < NX && tid + (e0 + ee) * 256 < FC_PH * FC_PW * X4) xs4[tid + (e0 + ee) * 256] = sx[ee];
        if (e0 != 0) return;
#pragma unroll
        for (int e = 0; e < ND; ++e)
            if (tid + e * 256 < FC_TH * FC_TW * D4) dys4[tid + e * 256] = sd[e];
    };
    if (PF && wg < a.ntiles) fetch(wg, 0);
    for (int tile = wg; tile < a.ntiles; tile += nwg) {
        __syncthreads();                                     // the previous tile's readers are done
        if (PF) {
            commit(0);
        } else {
#pragma unroll 1
            for (int e0 = 0; e0 < NX; e0 += NB) {
                fetch(tile, e0);
                commit(e0);
            }
        }
        __syncthreads();
        if (PF && tile + nwg < a.ntiles) fetch(tile + nwg, 0);              // in flight under this tile's MFMAs
#pragma unroll 1
        for (int r = 0; r < FC_TH; ++r) {
            const float* arow = dys + (r * FC_TW + kk) * COUT + n;          // A: dY[pixel KP j + kk][co n]
            const float* brow = xs + r * FC_PW * CIN;                       // B: patch row r (+ tap rows inside boff)
#pragma unroll 4
            for (int j = 0; j < FC_TW / KP; ++j) {
                const float av = arow[KP * j * COUT];
#pragma unroll
                for (int i = 0; i < NTW; ++i) {
                    const float bv = brow[boff[i] + KP * j * CIN];
                    if constexpr (SMALL) acc16[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc16[i], 0, 0, 0);
                    else acc32[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc32[i], 0, 0, 0);
                }
            }
        }
    }
    // one partial slab per workgroup, [co][ci][tap] = the parameter's layout
    float* slab = a.slabs + (size_t)wg * (COUT * CIN * 9);
#pragma unroll
    for (int i = 0; i < NTW; ++i) {
        const int t = wave + 4 * i;
        if (t >= NTILES) continue;
        const int tap = t / (CIN / TN), cb = t % (CIN / TN);
        const int ci = cb * TN + n;
        if constexpr (SMALL) {
#pragma unroll
            for (int r = 0; r < 4; ++r) slab[((size_t)(4 * kk + r) * CIN + ci) * 9 + tap] = acc16[i][r];
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) slab[((size_t)((r & 3) + 8 * (r >> 2) + 4 * kk) * CIN + ci) * 9 + tap] = acc32[i][r];
        }
    }
}

template <int COUT, int CIN>
__global__ __launch_bounds__(256, 2) void wgrad_fewch_kernel(FcArgs a) {
    fewch_body<COUT, CIN>(a, blockIdx.x, gridDim.x);
}

// dw[i] (+)= sum over slabs, fixed order: 32 outputs x 8 slab ranges per workgroup, 8 independent loads in flight per thread
__global__ __launch_bounds__(256) void fc_slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int n, float* __restrict__ out,
                                                             int accumulate) {
    __shared__ float red[8][32];
    const int col = threadIdx.x & 31, part = threadIdx.x >> 5;
    const int i = blockIdx.x * 32 + col;
    const int per = (nslabs + 7) / 8, g0 = part * per, g1 = min(nslabs, g0 + per);
    float s = 0.0f;
    if (i < n) {
        int g = g0;
        for (; g + 8 <= g1; g += 8) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = slabs[(size_t)(g + u) * n + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < g1; ++g) s += slabs[(size_t)g * n + i];
    }
    red[part][col] = s;
    __syncthreads();
    if (part == 0 && i < n) {
        float t = red[0][col];
#pragma unroll
        for (int u = 1; u < 8; ++u) t += red[u][col];
        out[i] = accumulate ? out[i] + t : t;
    }
}

template <typename K>
int fc_grid(K kern, int ntiles, int* cache) {
    if (!*cache) {
        int dev = 0, per_cu = 0;
        hipDeviceProp_t p;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&p, dev) != hipSuccess ||
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, kern, 256, 0) != hipSuccess || per_cu <= 0)
            return ntiles < 512 ? ntiles : 512;
        *cache = per_cu * p.multiProcessorCount;
    }
    return ntiles < *cache ? ntiles : *cache;
}
int g_grid_16_32 = 0, g_grid_32_64 = 0;

}  // namespace

// (not part of the public ABI: csrc/train.hip's ossid_conv_wgrad / ossid_conv_wgrad_workspace_bytes route here)
bool ossid_wgrad_fewch_takes(int Cin, int Cout, int taps, int in_cs, int dy_cs) {
    return taps == 9 && ((Cout == 16 && Cin == 32) || (Cout == 32 && Cin == 64)) && in_cs % 4 == 0 && dy_cs % 4 == 0;
}

size_t ossid_wgrad_fewch_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    const int th = fc_th(Cin);
    const long long nt = (long long)B * ((H + th - 1) / th) * ((W + FC_TW - 1) / FC_TW);
    if (nt <= 0 || nt > 0x7fffffff) return 0;
    const int grid = Cout == 16 ? fc_grid(wgrad_fewch_kernel<16, 32>, (int)nt, &g_grid_16_32)
                                : fc_grid(wgrad_fewch_kernel<32, 64>, (int)nt, &g_grid_32_64);
    return (size_t)grid * Cout * Cin * 9 * sizeof(float);
}

int ossid_wgrad_fewch(const ossid_wgrad_desc* d, void* stream) {
    const int B = d->batch, H = d->height, W = d->width, Cin = d->cin, Cout = d->cout;
    FcArgs a{};
    a.x = d->x, a.dy = d->dy, a.pre_scale = d->pre_scale, a.pre_shift = d->pre_shift, a.slabs = (float*)d->workspace;
    a.B = B, a.H = H, a.W = W;
    a.Hs = d->src_height > 0 ? d->src_height : H, a.Ws = d->src_width > 0 ? d->src_width : W;
    if (a.Hs > H || a.Ws > W) return OSSID_EINVAL;
    a.in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : Cin;
    a.dy_cs = d->dy_channel_stride > 0 ? d->dy_channel_stride : Cout;
    if (a.in_cs < Cin || a.dy_cs < Cout || ((uintptr_t)d->x & 15) || ((uintptr_t)d->dy & 15) || ((uintptr_t)d->workspace & 15))
        return OSSID_EINVAL;
    a.pre_relu = d->pre_relu;
    a.scale_h = (float)a.Hs / (float)H, a.scale_w = (float)a.Ws / (float)W;
    a.tiles_y = (H + fc_th(Cin) - 1) / fc_th(Cin), a.tiles_x = (W + FC_TW - 1) / FC_TW;
    const long long nt = (long long)B * a.tiles_y * a.tiles_x;
    if (nt > 0x7fffffff) return OSSID_EINVAL;
    a.ntiles = (int)nt;
    const size_t need = ossid_wgrad_fewch_workspace_bytes(B, H, W, Cin, Cout);
    if (!need || d->workspace_bytes < need) return OSSID_EINVAL;
    hipStream_t s = (hipStream_t)stream;
    int grid;
    if (Cout == 16) {
        grid = fc_grid(wgrad_fewch_kernel<16, 32>, a.ntiles, &g_grid_16_32);
        hipLaunchKernelGGL((wgrad_fewch_kernel<16, 32>), dim3(grid), dim3(256), 0, s, a);
    } else {
        grid = fc_grid(wgrad_fewch_kernel<32, 64>, a.ntiles, &g_grid_32_64);
        hipLaunchKernelGGL((wgrad_fewch_kernel<32, 64>), dim3(grid), dim3(256), 0, s, a);
    }
    const int nw = Cout * Cin * 9;
    hipLaunchKernelGGL(fc_slab_reduce_kernel, dim3((nw + 31) / 32), dim3(256), 0, s, (const float*)d->workspace, grid, nw, d->dw,
                       d->accumulate);
    return ossid_launch_status();
}
