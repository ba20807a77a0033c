// D4  the image backbone's stem in TRAINING and at test time: DenseNet-121 conv0 = nn.Conv2d(3, 64, 7, stride 2, padding 3,
// bias=False) (reference models/dtoid/network.py:164-170 via torchvision densenet121; run by ImageFeatExtract.forward :175-184)
// as an IMPLICIT-im2col convolution on the f32 matrix cores -- the [pixel][147] column matrix (393 MB at the finetune batch)
// is never written -- and its weight gradient.
//
// GEMM view (forward):  D[co][px] = sum_{ky,ci,kx} W[co][ci][ky][kx] * X[b][ci][2 oy + ky - 3][2 ox + kx - 3]
//   v_mfma_f32_32x32x2_f32, output channels on M (two 32-channel tiles), 32 consecutive output pixels of one row on N, the 147
//   taps on K.  A workgroup (4 waves) owns an 8-row x 32-column output tile: the 21 x 69 x 3 input patch under it is staged
//   once in LDS as rows [y][ci][x] (normalizeImageRange applied to real pixels on the way, zero padding in normalised space
//   as the reference has it), so the 21 (ky, ci) rows an OUTPUT row reads are 21 consecutive LDS rows starting at 6 * row;
//   K runs over (kx, pairs of such rows): the two k-halves of a lane pair read addresses 72 floats apart, every B operand is
//   one ds_read_b32 at a per-lane base plus an immediate offset. The weights (77 steps x 2 channel tiles x 64 lanes, 39 KB)
//   live in LDS in operand order for the workgroup's whole (persistent) life, gathered straight from the parameter's own
//   layout: no packing launch. A wave owns one 32-channel tile x four output rows (4 accumulator tiles). The next tile's
//   patch is fetched into registers in front of the MFMA loop and lands under it. Epilogue: accumulators -> a wave-private
//   LDS tile [32 px][32 co] -> whole 128-byte lines of the channels-last output.
// Exact f32 (the MFMA is an fmaf chain): this layer sits in front of a BatchNorm + ReLU + max-pool whose decisions the
// training step must take as the reference's f32 arithmetic does. 11.6 GFLOP at batch 8 = 74 us of matrix
// pipe against 157 MB of output.
//
// Weight gradient:  dW[co][c] = sum_px dY[px][co] * col_c(px), c = (ci, ky, kx) in the parameter's order, pixels on K (pairs
//   of neighbouring output pixels per MFMA), 5 accumulator tiles per wave (one 32-channel tile x 147 -> 160 columns); a
//   workgroup stages a 4-row x 32-column tile of dY and the 13 x 69 x 3 patch under it (the next tile's loads in flight under
//   this tile's MFMAs), each wave takes a channel tile x two rows; the row-pair sums meet in LDS in a fixed order,
//   workgroups write partial [64][147] slabs that a second kernel adds in a fixed order (bit-reproducible, no float atomics).
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

constexpr int KS = 7, ST = 2, CIN = 3, PADS = 3, CO = 64;
constexpr int NTAP = CIN * KS * KS;                 // 147
constexpr int PXW = (32 - 1) * ST + KS;             // 69 input columns under 32 output columns
constexpr int PXS = 72;                             // LDS row stride (floats)
constexpr int NR = KS * CIN;                        // 21 (ky, ci) rows per output row
constexpr int NPAIR = (NR + 1) / 2;                 // 11 pairs of them (the 22nd row has zero weights)
constexpr int NSTEP = KS * NPAIR;                   // 77 K-steps of 2

struct StemArgs {
    const float* img;        // [B][3][H][W]
    const float* w;          // [64][3][7][7]
    const float* bias;       // [64] or null
    const float* mean;       // [3] or null
    const float* inv_std;    // [3] or null
    float* out;              // forward: [B][Ho][Wo][64]; weight gradient: partial slabs [grid][64*147]
    const float* dy;         // weight gradient: [B][Ho][Wo][64]
    int B, H, W, Ho, Wo, tiles_y, tiles_x, ntiles;
};

// The input patch under a tile: ROWS_Y input rows x 3 channels x 72 columns starting at image row y0, column x0 (may be
// negative / past the edge: zero), LDS rows [y][ci], columns >= 69 zero. Two halves so that a tile's global loads are ALL in
// flight at once and, issued for the NEXT tile in front of this tile's MFMA loop, land underneath it: fetch() into
// registers (element e = i * 256 + thread), commit() into LDS after the barrier that retires the previous patch.
template <int ROWS_Y>
struct Patch {
    static constexpr int TOTAL = ROWS_Y * CIN * PXS;
    static constexpr int N = (TOTAL + 255) / 256;
    float v[N];
    __device__ __forceinline__ void fetch(const StemArgs& a, int b, int y0, int x0) {
        const float* img = a.img + (size_t)b * CIN * a.H * a.W;
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            const int row = e / PXS, x = e - row * PXS;
            const int y = row / CIN, ci = row - y * CIN;
            const int gy = y0 + y, gx = x0 + x;
            float t = 0.0f;
            if (e < TOTAL && x < PXW && gy >= 0 && gy < a.H && gx >= 0 && gx < a.W) {
                t = img[((size_t)ci * a.H + gy) * a.W + gx];
                if (a.mean) t = (t - a.mean[ci]) * a.inv_std[ci];
            }
            v[i] = t;
        }
    }
    __device__ __forceinline__ void commit(float* patch) const {
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int e = i * 256 + (int)threadIdx.x;
            if (e < TOTAL) patch[e] = v[i];
        }
    }
};

__device__ __forceinline__ void tile_origin(const StemArgs& a, int tile, int th, int& b, int& oy0, int& ox0) {
    const int tx = tile % a.tiles_x, r1 = tile / a.tiles_x;
    b = r1 / a.tiles_y;
    oy0 = (r1 % a.tiles_y) * th, ox0 = tx * 32;
}

constexpr int F_TH = 8;                                   // forward: output rows per workgroup tile
constexpr int F_PY = (F_TH - 1) * ST + KS;                // 21 input rows
constexpr int F_ROWS = F_PY * CIN + 1;                    // 63 LDS rows + one all-zero row
constexpr int F_EPS = 32 + 4;                             // row stride of a wave's epilogue tile (floats): conflict-free 16-byte stores

// wave w: channel tile w & 1, output rows 4 * (w >> 1) .. + 3 of the tile (4 accumulator tiles); the weights of the whole
// persistent workgroup sit in LDS in operand order [step][channel tile][lane] (one conflict-free ds_read_b32 per step)
__global__ __launch_bounds__(256, 2) void stem_conv_fwd_kernel(StemArgs a) {
    __shared__ float patch[F_ROWS * PXS];                 // 18.4 KB
    __shared__ float ep[4][32 * F_EPS];                   // 18.4 KB
    __shared__ float wl[NSTEP * 2 * 64];                  // 39.4 KB
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, k = lane >> 5;
    const int mt = wave & 1, rg = wave >> 1;
    // weights: step s = kx * 11 + jp multiplies rows R = 2 jp + k (ky = R / 3, ci = R % 3) at column offset kx
#pragma unroll 13
    for (int i = 0; i < (NSTEP * 128 + 255) / 256; ++i) {   // (unrolled: 39 dependent L2 round trips otherwise)
        const int e = i * 256 + (int)threadIdx.x;
        if (e < NSTEP * 128) {
            const int l = e & 63, m2 = (e >> 6) & 1, st = e >> 7;
            const int kx = st / NPAIR, jp = st - kx * NPAIR, R = 2 * jp + (l >> 5);
            wl[e] = R < NR ? a.w[(size_t)(32 * m2 + (l & 31)) * NTAP + (R % CIN) * KS * KS + (R / CIN) * KS + kx] : 0.0f;
        }
    }
    const float* wsrc = wl + mt * 64 + lane;
    for (int e = threadIdx.x; e < PXS; e += 256) patch[(F_ROWS - 1) * PXS + e] = 0.0f;
    const float* bsrc = patch + (ST * CIN * 4 * rg + k) * PXS + 2 * n;
    float* myep = ep[wave];
    Patch<F_PY> pf;
    int b, oy0, ox0;
    if ((int)blockIdx.x < a.ntiles) {
        tile_origin(a, blockIdx.x, F_TH, b, oy0, ox0);
        pf.fetch(a, b, oy0 * ST - PADS, ox0 * ST - PADS);
    }
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        tile_origin(a, tile, F_TH, b, oy0, ox0);
        __syncthreads();                                   // the previous tile's readers are done with the patch
        pf.commit(patch);
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) {            // the next tile's loads fly under this tile's MFMAs
            int nb, noy, nox;
            tile_origin(a, tile + gridDim.x, F_TH, nb, noy, nox);
            pf.fetch(a, nb, noy * ST - PADS, nox * ST - PADS);
        }
        v16f acc[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[r][i] = 0.0f;
#pragma unroll 1
        for (int kx = 0; kx < KS; ++kx) {                  // (rolled: fully unrolled, the 385 LDS reads get hoisted into spills)
            const float* wk = wsrc + kx * NPAIR * 128;
            const float* bk = bsrc + kx;
#pragma unroll
            for (int jp = 0; jp < NPAIR; ++jp) {
                const float wv = wk[jp * 128];
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[r] = mfma(wv, bk[(ST * CIN * r + 2 * jp) * PXS], acc[r]);
            }
        }
        // epilogue: [32 px][32 co] per output row through the wave's own LDS tile, then whole 128-byte lines per pixel
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int oy = oy0 + 4 * rg + r;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int co = 8 * q + 4 * k;
                float4 v = make_float4(acc[r][4 * q], acc[r][4 * q + 1], acc[r][4 * q + 2], acc[r][4 * q + 3]);
                if (a.bias) {
                    const float* bb = a.bias + 32 * mt + co;          // (a slice of the flat parameter buffer: any alignment)
                    v.x += bb[0], v.y += bb[1], v.z += bb[2], v.w += bb[3];
                }
                *(float4*)(myep + n * F_EPS + co) = v;
            }
            __builtin_amdgcn_wave_barrier();
            if (oy < a.Ho) {
                float* orow = a.out + (((size_t)b * a.Ho + oy) * a.Wo + ox0) * CO + 32 * mt;
#pragma unroll
                for (int it = 0; it < 4; ++it) {
                    const int px = it * 8 + (lane >> 3), c4 = lane & 7;
                    const float4 v = *(const float4*)(myep + px * F_EPS + 4 * c4);
                    if (ox0 + px < a.Wo) *(float4*)(orow + (size_t)px * CO + 4 * c4) = v;
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
}

constexpr int G_TH = 4;                                   // weight gradient: output rows per tile (1 per wave)
constexpr int G_PY = (G_TH - 1) * ST + KS;                // 13 input rows
constexpr int G_NT = 5;                                   // 147 columns -> five 32-column tiles
constexpr int G_RED_STRIDE = 148;
constexpr int G_PATCH = G_PY * CIN * PXS;                 // 2 808 floats
constexpr int G_DY = G_TH * 32 * CO;                      // 8 192 floats
static_assert(CO * G_RED_STRIDE <= G_PATCH + G_DY, "the cross-wave reduction reuses the tile's LDS");

__global__ __launch_bounds__(256, 2) void stem_conv_wgrad_kernel(StemArgs a) {
    __shared__ float4 lds4[(G_PATCH + G_DY) / 4];          // 44 KB: dY tile [128 px][64 co], then the patch
    float* dyt = (float*)lds4;
    float* patch = dyt + G_DY;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int n = lane & 31, k = lane >> 5;
    const int mt = wave & 1, rg = wave >> 1;              // this wave: channel tile mt, output rows 2 rg and 2 rg + 1 of the tile
    // column c = 32 nt + n of the [147] tap axis in the parameter's order (ci, ky, kx): where its input sits in the patch,
    // relative to the output pixel's origin ((2 r) * 3 rows, 2 * ox columns)
    int cbase[G_NT];
#pragma unroll
    for (int nt = 0; nt < G_NT; ++nt) {
        const int c = min(32 * nt + n, NTAP - 1);
        const int ci = c / (KS * KS), ky = (c / KS) % KS, kx = c % KS;
        cbase[nt] = (ky * CIN + ci) * PXS + kx + ST * k;
    }
    v16f acc[G_NT];
#pragma unroll
    for (int nt = 0; nt < G_NT; ++nt)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[nt][i] = 0.0f;
    Patch<G_PY> pf;
    float4 dyv[G_DY / 4 / 256];                                         // 8 float4 of the dY tile per thread
    auto fetch_tile = [&](int tile) {
        int b, oy0, ox0;
        tile_origin(a, tile, G_TH, b, oy0, ox0);
        pf.fetch(a, b, oy0 * ST - PADS, ox0 * ST - PADS);
#pragma unroll
        for (int i = 0; i < G_DY / 4 / 256; ++i) {                      // dY tile, zero outside the image
            const int e = i * 256 + (int)threadIdx.x;
            const int c4 = e & 15, px = (e >> 4) & 31, r = e >> 9;
            dyv[i] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (oy0 + r < a.Ho && ox0 + px < a.Wo)
                dyv[i] = *(const float4*)(a.dy + (((size_t)b * a.Ho + oy0 + r) * a.Wo + ox0 + px) * CO + 4 * c4);
        }
    };
    if ((int)blockIdx.x < a.ntiles) fetch_tile(blockIdx.x);
    for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
        __syncthreads();
        pf.commit(patch);
#pragma unroll
        for (int i = 0; i < G_DY / 4 / 256; ++i) lds4[i * 256 + threadIdx.x] = dyv[i];
        __syncthreads();
        if (tile + (int)gridDim.x < a.ntiles) fetch_tile(tile + gridDim.x);
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int row = 2 * rg + r;                                   // wave-uniform
            const float* arow = dyt + (row * 32 + k) * CO + 32 * mt + n;  // A: dY[pixel 2 j + k][32 mt + n]
            const float* brow = patch + ST * CIN * row * PXS;            // B: the patch rows under output row `row`
#pragma unroll 4
            for (int j = 0; j < 16; ++j) {
                const float av = arow[2 * j * CO];
#pragma unroll
                for (int nt = 0; nt < G_NT; ++nt) acc[nt] = mfma(av, brow[cbase[nt] + 2 * ST * j], acc[nt]);
            }
        }
    }
    // the two row-pair waves of a channel tile, added in wave order through LDS (fixed order: bit-reproducible)
    float* red = (float*)lds4;
    for (int w = 0; w < 2; ++w) {
        __syncthreads();
        if (rg == w) {
#pragma unroll
            for (int nt = 0; nt < G_NT; ++nt)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int co = 32 * mt + (i & 3) + 8 * (i >> 2) + 4 * k, c = 32 * nt + n;
                    if (c < NTAP) {
                        float* p = red + co * G_RED_STRIDE + c;
                        *p = w == 0 ? acc[nt][i] : *p + acc[nt][i];
                    }
                }
        }
    }
    __syncthreads();
    float* slab = a.out + (size_t)blockIdx.x * (CO * NTAP);
    for (int e = threadIdx.x; e < CO * NTAP; e += 256) slab[e] = red[(e / NTAP) * G_RED_STRIDE + e % NTAP];
}

// out[i] (+)= sum over slabs g of slabs[g][i], fixed order: 32 outputs x 8 slab ranges per workgroup, each range summed
// with 8 independent loads in flight (one thread walking hundreds of slabs is a chain of L2 round trips), ranges then added
// in order.
__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ slabs, int nslabs, int n, float* __restrict__ out,
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

// Persistent grids: as many workgroups as are resident at once (asked of the runtime once per kernel), each looping over tiles.
template <typename K>
int stem_grid(K kern, int ntiles, int* cache) {
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
int g_fwd_grid = 0, g_wgrad_grid = 0;

bool stem_shape_ok(int B, int Cin, int H, int W, int Cout, int k, int stride, int pad) {
    return B > 0 && H > 0 && W > 0 && Cin == CIN && Cout == CO && k == KS && stride == ST && pad == PADS &&
           (H + 2 * PADS - KS) / ST + 1 > 0 && (W + 2 * PADS - KS) / ST + 1 > 0;
}


// =====================================================================================================================
// The rest of the training stem, channels-last, C / 4 a power of two <= 64 (a thread owns 4 channels; 256 / (C / 4) pixel
// strips per workgroup):
//   dw3x3_add      y = x + conv2d_dw_group(x, k_b) (network.py:178-179, 186-192; flip = taps rotated by 180 degrees = its
//                  data gradient), a thread walks a strip of one row with a rolling 3x3 window (3 new loads per output, the
//                  36 tap weights in registers), optionally with the column sums of y about y's first pixel as pivot (the
//                  batch statistics of norm0) left as per-workgroup partials [P][2][C] for ossid_bn_fold_fwd
//   dw3x3_bwd_k    dk[b][c][tap] = sum_px g[b][px][c] x[b][px + tap][c]: per-workgroup partials, fixed-order slab_reduce
//   stem_pool_fwd  max-pool 3/2/1 of relu(scale[c] * m + shift[c]) with the window position of the maximum (uint8): the
//                  normalised tensor is never written
//   stem_pool_bwd  the mirror image in two passes over m: pass 1 (dm == NULL) forms gm = [scale m + shift > 0] * (sum of the
//                  pooled gradients whose maximum sat here) on the fly and leaves the column sums (gm, gm * m) as partials
//                  for ossid_bn_fold_bwd; pass 2 forms gm again and writes dm = scale gm + coef_x m + coef_1 -- the gradient
//                  through BatchNorm's output AND its batch statistics. Neither the un-pooled gradient nor gm is ever stored.
struct StemElArgs {
    const float4* x;          // dw3x3: input; pool: m
    const float4* g;          // bwd_k: upstream gradient; pool bwd: pooled gradient dp
    const float* kern;        // [B or 1][C][3][3]
    const uint8_t* idx_in;
    float4* out;
    uint8_t* idx_out;
    float* partials;
    float* pivot_out;
    const float *scale, *shift, *coef_x, *coef_1;
    int kern_bs, B, H, W, C4, Ho, Wo, flip, rows_per_block;
};

// per-channel (s1, s2) of the workgroup's threads -> one partial row pair [2][C], strips added in a fixed order
__device__ __forceinline__ void block_colsum_store(const float (&s1)[4], const float (&s2)[4], int c4, int strip, int nstrips, int C4,
                                                   float* row) {
    __shared__ float red[256][8];
    float* r = red[strip * C4 + c4];
#pragma unroll
    for (int i = 0; i < 4; ++i) r[i] = s1[i], r[4 + i] = s2[i];
    __syncthreads();
    if (strip == 0) {
        float t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) t[i] = red[c4][i];
        for (int y = 1; y < nstrips; ++y)
#pragma unroll
            for (int i = 0; i < 8; ++i) t[i] += red[y * C4 + c4][i];
        *(float4*)(row + 4 * c4) = make_float4(t[0], t[1], t[2], t[3]);
        *(float4*)(row + 4 * C4 + 4 * c4) = make_float4(t[4], t[5], t[6], t[7]);
    }
}

constexpr int DW_XB = 4;       // consecutive pixels per thread: their 3 x 6 input window is loaded in one go

// rows y-1 .. y+1, columns xa-1 .. xa+DW_XB of image b (zero outside): 18 independent loads
__device__ __forceinline__ void load_window(const float4* __restrict__ x, int b, int y, int xa, int H, int W, int C4, int c4,
                                            float4 (&v)[3][DW_XB + 2]) {
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int yy = y + dy - 1;
        const bool rv = yy >= 0 && yy < H;
        const float4* row = x + ((size_t)b * H + (rv ? yy : y)) * W * C4 + c4;
#pragma unroll
        for (int j = 0; j < DW_XB + 2; ++j) {
            const int xx = xa - 1 + j;
            v[dy][j] = (rv && xx >= 0 && xx < W) ? row[(size_t)xx * C4] : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
}

// grid (ceil(W / (4 strips)), ceil(H / rows_per_block), B)
template <bool STATS>
__global__ __launch_bounds__(256) void dw3x3_add_kernel(StemElArgs a) {
    const int C4 = a.C4, c4 = threadIdx.x % C4, strip = threadIdx.x / C4, nstrips = 256 / C4;
    const int b = blockIdx.z, H = a.H, W = a.W;
    float w[4][9];
    {
        const float* kk = a.kern + (size_t)b * a.kern_bs + (size_t)c4 * 36;
#pragma unroll
        for (int ch = 0; ch < 4; ++ch)
#pragma unroll
            for (int t = 0; t < 9; ++t) w[ch][t] = kk[ch * 9 + (a.flip ? 8 - t : t)];
    }
    float pv[4] = {0.f, 0.f, 0.f, 0.f};
    if (STATS) {     // pivot = y at (image 0, row 0, column 0), recomputed by every thread for its channels (4 taps inside)
        const float* k0 = a.kern + (size_t)c4 * 36;
        const float4 v00 = a.x[c4];
        float acc[4] = {v00.x, v00.y, v00.z, v00.w};
#pragma unroll
        for (int dy = 1; dy < 3; ++dy)
#pragma unroll
            for (int dx = 1; dx < 3; ++dx) {
                if (dy - 1 >= H || dx - 1 >= W) continue;
                const float4 v = a.x[((size_t)(dy - 1) * W + (dx - 1)) * C4 + c4];
                const float vv[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int ch = 0; ch < 4; ++ch) acc[ch] = fmaf(vv[ch], k0[ch * 9 + dy * 3 + dx], acc[ch]);
            }
#pragma unroll
        for (int ch = 0; ch < 4; ++ch) pv[ch] = acc[ch];
        if (blockIdx.x == 0 && blockIdx.y == 0 && blockIdx.z == 0 && strip == 0)
            *(float4*)(a.pivot_out + 4 * c4) = make_float4(pv[0], pv[1], pv[2], pv[3]);
    }
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int xa = (blockIdx.x * nstrips + strip) * DW_XB;
    const int ya = blockIdx.y * a.rows_per_block, ye = min(H, ya + a.rows_per_block);
    if (xa < W)
        for (int y = ya; y < ye; ++y) {
            float4 v[3][DW_XB + 2];
            load_window(a.x, b, y, xa, H, W, C4, c4, v);
#pragma unroll
            for (int j = 0; j < DW_XB; ++j) {
                if (xa + j >= W) break;
                float acc[4] = {v[1][j + 1].x, v[1][j + 1].y, v[1][j + 1].z, v[1][j + 1].w};
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 q = v[dy][j + dx];
                        const float qq[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                        for (int ch = 0; ch < 4; ++ch) acc[ch] = fmaf(qq[ch], w[ch][dy * 3 + dx], acc[ch]);
                    }
                a.out[(((size_t)b * H + y) * W + xa + j) * C4 + c4] = make_float4(acc[0], acc[1], acc[2], acc[3]);
                if (STATS) {
#pragma unroll
                    for (int ch = 0; ch < 4; ++ch) {
                        const float d = acc[ch] - pv[ch];
                        s1[ch] += d, s2[ch] += d * d;
                    }
                }
            }
        }
    if (STATS) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        block_colsum_store(s1, s2, c4, strip, nstrips, C4, a.partials + blk * 8 * C4);
    }
}

// grid (ceil(W / (4 strips)), row chunks, B); a thread sums its 36 (channel, tap) products over 4 pixels per row of the chunk
__global__ __launch_bounds__(256) void dw3x3_bwd_k_kernel(StemElArgs a) {
    __shared__ float red[256 * 36];
    const int C4 = a.C4, c4 = threadIdx.x % C4, strip = threadIdx.x / C4, nstrips = 256 / C4;
    const int b = blockIdx.z, H = a.H, W = a.W;
    const int xa = (blockIdx.x * nstrips + strip) * DW_XB;
    const int ya = blockIdx.y * a.rows_per_block, ye = min(H, ya + a.rows_per_block);
    float s[36];
#pragma unroll
    for (int t = 0; t < 36; ++t) s[t] = 0.0f;
    if (xa < W)
        for (int y = ya; y < ye; ++y) {
            float4 v[3][DW_XB + 2], gq[DW_XB];
            load_window(a.x, b, y, xa, H, W, C4, c4, v);
#pragma unroll
            for (int j = 0; j < DW_XB; ++j)
                gq[j] = xa + j < W ? a.g[(((size_t)b * H + y) * W + xa + j) * C4 + c4] : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int j = 0; j < DW_XB; ++j)
#pragma unroll
                for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                    for (int dx = 0; dx < 3; ++dx) {
                        const float4 q = v[dy][j + dx];
                        const int t = dy * 3 + dx;
                        s[t] = fmaf(gq[j].x, q.x, s[t]), s[9 + t] = fmaf(gq[j].y, q.y, s[9 + t]);
                        s[18 + t] = fmaf(gq[j].z, q.z, s[18 + t]), s[27 + t] = fmaf(gq[j].w, q.w, s[27 + t]);
                    }
        }
#pragma unroll
    for (int t = 0; t < 36; ++t) red[(strip * C4 + c4) * 36 + t] = s[t];
    __syncthreads();
    const size_t chunk = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
    for (int e = threadIdx.x; e < C4 * 36; e += 256) {        // e = c4 * 36 + (channel * 9 + tap)
        float t = red[e];
        for (int y = 1; y < nstrips; ++y) t += red[(size_t)y * C4 * 36 + e];
        a.partials[(chunk * gridDim.z + b) * C4 * 36 + e] = t;
    }
}

__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(StemElArgs a, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int C4 = a.C4, c4 = (int)(i % C4);
    size_t r = i / C4;
    const int xo = (int)(r % a.Wo);
    r /= a.Wo;
    const int yo = (int)(r % a.Ho), b = (int)(r / a.Ho);
    const float4 sc4 = *(const float4*)(a.scale + 4 * c4), sh4 = *(const float4*)(a.shift + 4 * c4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    float m[4] = {-INFINITY, -INFINITY, -INFINITY, -INFINITY};
    unsigned char am[4] = {255, 255, 255, 255};
#pragma unroll
    for (int dy = 0; dy < 3; ++dy) {
        const int y = 2 * yo - 1 + dy;
        if (y < 0 || y >= a.H) continue;
#pragma unroll
        for (int dx = 0; dx < 3; ++dx) {
            const int x = 2 * xo - 1 + dx;
            if (x < 0 || x >= a.W) continue;
            const float4 v4 = a.x[(((size_t)b * a.H + y) * a.W + x) * C4 + c4];
            const float v[4] = {v4.x, v4.y, v4.z, v4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float yv = fmaxf(sc[e] * v[e] + sh[e], 0.0f);
                if (yv > m[e] || am[e] == 255) m[e] = yv, am[e] = (unsigned char)(dy * 3 + dx);      // first maximum wins
            }
        }
    }
    a.out[i] = make_float4(m[0], m[1], m[2], m[3]);
    ((uchar4*)a.idx_out)[i] = make_uchar4(am[0], am[1], am[2], am[3]);
}

// A thread owns the 2x2 input block (rows 2a, 2a+1; columns 2c, 2c+1) of its 4 channels: the four pool windows (a or a+1,
// c or c+1) are the only ones that contain any of its pixels, so 4 argmax + 4 gradient + 4 m loads -- all independent --
// serve 4 pixels. grid (ceil(ceil(W/2) / strips), ceil(ceil(H/2) / rows_per_block), B).
// APPLY = false: column sums (gm, gm * m) -> partials; true: dm = scale gm + coef_x m + coef_1
template <bool APPLY>
__global__ __launch_bounds__(256) void stem_pool_bwd_kernel(StemElArgs a) {
    const int C4 = a.C4, c4 = threadIdx.x % C4, strip = threadIdx.x / C4, nstrips = 256 / C4;
    const int b = blockIdx.z, H = a.H, W = a.W, Ho = a.Ho, Wo = a.Wo;
    const float4 sc4 = *(const float4*)(a.scale + 4 * c4), sh4 = *(const float4*)(a.shift + 4 * c4);
    const float sc[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, sh[4] = {sh4.x, sh4.y, sh4.z, sh4.w};
    float cx[4] = {0.f, 0.f, 0.f, 0.f}, c1[4] = {0.f, 0.f, 0.f, 0.f};
    if (APPLY) {
        const float4 a4 = *(const float4*)(a.coef_x + 4 * c4), b4 = *(const float4*)(a.coef_1 + 4 * c4);
        cx[0] = a4.x, cx[1] = a4.y, cx[2] = a4.z, cx[3] = a4.w, c1[0] = b4.x, c1[1] = b4.y, c1[2] = b4.z, c1[3] = b4.w;
    }
    const float4* mb = a.x + (size_t)b * H * W * C4 + c4;
    const float4* gb = a.g + (size_t)b * Ho * Wo * C4 + c4;
    const uchar4* ib = (const uchar4*)a.idx_in + (size_t)b * Ho * Wo * C4 + c4;
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    const int c = blockIdx.x * nstrips + strip;                     // column pair
    const int a0 = blockIdx.y * a.rows_per_block, a1 = min((H + 1) / 2, a0 + a.rows_per_block);
    const float4 zero = make_float4(0.f, 0.f, 0.f, 0.f);
    if (2 * c < W)
        for (int ra = a0; ra < a1; ++ra) {
            // the four windows (ra + wy, c + wx) and the four pixels (2 ra + py, 2 c + px)
            float4 gw[2][2], mv[2][2];
            uchar4 am[2][2];
#pragma unroll
            for (int wy = 0; wy < 2; ++wy)
#pragma unroll
                for (int wx = 0; wx < 2; ++wx) {
                    const bool ok = ra + wy < Ho && c + wx < Wo;
                    const size_t o = ((size_t)(ok ? ra + wy : 0) * Wo + (ok ? c + wx : 0)) * C4;
                    gw[wy][wx] = ok ? gb[o] : zero;
                    am[wy][wx] = ok ? ib[o] : make_uchar4(255, 255, 255, 255);
                    const bool pk = 2 * ra + wy < H && 2 * c + wx < W;
                    mv[wy][wx] = pk ? mb[((size_t)(2 * ra + wy) * W + 2 * c + wx) * C4] : zero;
                }
#pragma unroll
            for (int py = 0; py < 2; ++py)
#pragma unroll
                for (int px = 0; px < 2; ++px) {
                    if (2 * ra + py >= H || 2 * c + px >= W) continue;
                    float g[4] = {0.f, 0.f, 0.f, 0.f};
                    // pixel (2 ra + py, 2 c + px) inside window (ra + wy, c + wx): row offset 2 ra + py - (2 (ra + wy) - 1) =
                    // py + 1 - 2 wy in 0..2, i.e. wy <= py (an even row lies in one window, an odd row in two); columns alike
#pragma unroll
                    for (int wy = 0; wy <= py; ++wy)
#pragma unroll
                        for (int wx = 0; wx <= px; ++wx) {
                            const unsigned char pos = (unsigned char)((py + 1 - 2 * wy) * 3 + (px + 1 - 2 * wx));
                            const uchar4 q = am[wy][wx];
                            const float4 gv = gw[wy][wx];
                            if (q.x == pos) g[0] += gv.x;
                            if (q.y == pos) g[1] += gv.y;
                            if (q.z == pos) g[2] += gv.z;
                            if (q.w == pos) g[3] += gv.w;
                        }
                    const float m4[4] = {mv[py][px].x, mv[py][px].y, mv[py][px].z, mv[py][px].w};
                    float o4[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float gm = (sc[e] * m4[e] + sh[e] > 0.0f) ? g[e] : 0.0f;
                        if (APPLY) o4[e] = sc[e] * gm + cx[e] * m4[e] + c1[e];
                        else s1[e] += gm, s2[e] += gm * m4[e];
                    }
                    if (APPLY)
                        a.out[(((size_t)b * H + 2 * ra + py) * W + 2 * c + px) * C4 + c4] = make_float4(o4[0], o4[1], o4[2], o4[3]);
                }
        }
    if (!APPLY) {
        const size_t blk = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
        block_colsum_store(s1, s2, c4, strip, nstrips, C4, a.partials + blk * 8 * C4);
    }
}

bool el_shape_ok(int B, int H, int W, int C) {
    const int C4 = C / 4;
    return B > 0 && B <= 65535 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C4 <= 64 && (C4 & (C4 - 1)) == 0;
}
constexpr int DW_ROWS = 2, POOL_ROWS = 4, BWDK_ROWS = 4;     // image rows (pool: row pairs) per workgroup
int dw_gx(int W, int C4) { return (W + (256 / C4) * DW_XB - 1) / ((256 / C4) * DW_XB); }
int pool_gx(int W, int C4) { return ((W + 1) / 2 + 256 / C4 - 1) / (256 / C4); }

}  // namespace

extern "C" {

int ossid_stem_conv_fwd(const float* img_nchw, int B, int Cin, int H, int W, const float* weight, int Cout, int k, int stride,
                        int pad, const float* bias, const float* mean, const float* inv_std, float* out, void* stream) {
    if (!img_nchw || !weight || !out || !stem_shape_ok(B, Cin, H, W, Cout, k, stride, pad) || (!mean != !inv_std) ||
        ((uintptr_t)out & 15))
        return OSSID_EINVAL;
    StemArgs a{};
    a.img = img_nchw, a.w = weight, a.bias = bias, a.mean = mean, a.inv_std = inv_std, a.out = out;
    a.B = B, a.H = H, a.W = W, a.Ho = (H + 2 * PADS - KS) / ST + 1, a.Wo = (W + 2 * PADS - KS) / ST + 1;
    a.tiles_y = (a.Ho + F_TH - 1) / F_TH, a.tiles_x = (a.Wo + 31) / 32;
    const long long nt = (long long)B * a.tiles_y * a.tiles_x;
    if (nt > 0x7fffffff) return OSSID_EINVAL;
    a.ntiles = (int)nt;
    hipLaunchKernelGGL(stem_conv_fwd_kernel, dim3(stem_grid(stem_conv_fwd_kernel, a.ntiles, &g_fwd_grid)), dim3(256), 0,
                       (hipStream_t)stream, a);
    return ossid_launch_status();
}

size_t ossid_stem_conv_wgrad_workspace_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const int Ho = (H + 2 * PADS - KS) / ST + 1, Wo = (W + 2 * PADS - KS) / ST + 1;
    if (Ho <= 0 || Wo <= 0) return 0;
    const long long nt = (long long)B * ((Ho + G_TH - 1) / G_TH) * ((Wo + 31) / 32);
    if (nt > 0x7fffffff) return 0;
    return (size_t)stem_grid(stem_conv_wgrad_kernel, (int)nt, &g_wgrad_grid) * CO * NTAP * sizeof(float);
}

int ossid_stem_conv_wgrad(const float* img_nchw, const float* dy, int B, int Cin, int H, int W, int Cout, int k, int stride,
                          int pad, const float* mean, const float* inv_std, void* workspace, size_t workspace_bytes,
                          float* dweight, int accumulate, void* stream) {
    if (!img_nchw || !dy || !dweight || !workspace || !stem_shape_ok(B, Cin, H, W, Cout, k, stride, pad) || (!mean != !inv_std) ||
        ((uintptr_t)dy & 15) || ((uintptr_t)workspace & 15))
        return OSSID_EINVAL;
    const size_t need = ossid_stem_conv_wgrad_workspace_bytes(B, H, W);
    if (!need || workspace_bytes < need) return OSSID_EINVAL;
    StemArgs a{};
    a.img = img_nchw, a.dy = dy, a.mean = mean, a.inv_std = inv_std, a.out = (float*)workspace;
    a.B = B, a.H = H, a.W = W, a.Ho = (H + 2 * PADS - KS) / ST + 1, a.Wo = (W + 2 * PADS - KS) / ST + 1;
    a.tiles_y = (a.Ho + G_TH - 1) / G_TH, a.tiles_x = (a.Wo + 31) / 32;
    a.ntiles = B * a.tiles_y * a.tiles_x;
    const int grid = stem_grid(stem_conv_wgrad_kernel, a.ntiles, &g_wgrad_grid);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(stem_conv_wgrad_kernel, dim3(grid), dim3(256), 0, s, a);
    const int nw = CO * NTAP;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((nw + 31) / 32), dim3(256), 0, s, (const float*)workspace, grid, nw, dweight,
                       accumulate);
    return ossid_launch_status();
}

int ossid_dw_add_stats_partials(int B, int H, int W, int C) {
    if (!el_shape_ok(B, H, W, C)) return 0;
    return dw_gx(W, C / 4) * ((H + DW_ROWS - 1) / DW_ROWS) * B;
}

int ossid_dw_add_stats_nhwc(const float* x, const float* kernels, int kernels_batch_stride, int B, int H, int W, int C, int flip,
                            float* out, float* partials, float* pivot_out, void* stream) {
    if (!x || !kernels || !out || !el_shape_ok(B, H, W, C) || kernels_batch_stride < 0 || (!partials != !pivot_out) ||
        ((uintptr_t)x & 15) || ((uintptr_t)out & 15))
        return OSSID_EINVAL;
    StemElArgs a{};
    a.x = (const float4*)x, a.kern = kernels, a.kern_bs = kernels_batch_stride, a.out = (float4*)out, a.partials = partials;
    a.pivot_out = pivot_out, a.B = B, a.H = H, a.W = W, a.C4 = C / 4, a.flip = flip, a.rows_per_block = DW_ROWS;
    const dim3 grid(dw_gx(W, a.C4), (H + DW_ROWS - 1) / DW_ROWS, B);
    if (partials)
        hipLaunchKernelGGL(dw3x3_add_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(dw3x3_add_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return ossid_launch_status();
}

int ossid_dw_add_nhwc(const float* x, const float* kernels, int kernels_batch_stride, int B, int H, int W, int C, int flip,
                      float* out, void* stream) {
    return ossid_dw_add_stats_nhwc(x, kernels, kernels_batch_stride, B, H, W, C, flip, out, nullptr, nullptr, stream);
}

size_t ossid_dw_bwd_k_workspace_floats(int B, int H, int W, int C) {
    if (!el_shape_ok(B, H, W, C)) return 0;
    return (size_t)((H + BWDK_ROWS - 1) / BWDK_ROWS) * dw_gx(W, C / 4) * B * C * 9;
}

int ossid_dw_bwd_k_nhwc(const float* x, const float* g, int B, int H, int W, int C, float* workspace, float* dk, void* stream) {
    if (!x || !g || !workspace || !dk || !el_shape_ok(B, H, W, C) || ((uintptr_t)x & 15) || ((uintptr_t)g & 15)) return OSSID_EINVAL;
    StemElArgs a{};
    a.x = (const float4*)x, a.g = (const float4*)g, a.partials = workspace, a.B = B, a.H = H, a.W = W, a.C4 = C / 4;
    a.rows_per_block = BWDK_ROWS;
    const int gx = dw_gx(W, a.C4), gy = (H + BWDK_ROWS - 1) / BWDK_ROWS, chunks = gx * gy;
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(dw3x3_bwd_k_kernel, dim3(gx, gy, B), dim3(256), 0, s, a);
    const int n = B * C * 9;
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((n + 31) / 32), dim3(256), 0, s, (const float*)workspace, chunks, n, dk, 0);
    return ossid_launch_status();
}

int ossid_stem_pool_fwd(const float* m, const float* scale, const float* shift, int B, int H, int W, int C, float* out,
                        uint8_t* argmax, void* stream) {
    if (!m || !scale || !shift || !out || !argmax || !el_shape_ok(B, H, W, C) || ((uintptr_t)m & 15) || ((uintptr_t)out & 15) ||
        ((uintptr_t)argmax & 3))
        return OSSID_EINVAL;
    StemElArgs a{};
    a.x = (const float4*)m, a.scale = scale, a.shift = shift, a.out = (float4*)out, a.idx_out = argmax;
    a.B = B, a.H = H, a.W = W, a.C4 = C / 4, a.Ho = (H - 1) / 2 + 1, a.Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * a.Ho * a.Wo * a.C4;
    hipLaunchKernelGGL(stem_pool_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, total);
    return ossid_launch_status();
}

int ossid_stem_pool_bwd_partials(int B, int H, int W, int C) {
    return el_shape_ok(B, H, W, C) ? B * (((H + 1) / 2 + POOL_ROWS - 1) / POOL_ROWS) * pool_gx(W, C / 4) : 0;
}

int ossid_stem_pool_bwd(const float* m, const uint8_t* argmax, const float* dpooled, const float* scale, const float* shift,
                        const float* coef_x, const float* coef_1, int B, int H, int W, int C, float* partials, float* dm,
                        void* stream) {
    if (!m || !argmax || !dpooled || !scale || !shift || !el_shape_ok(B, H, W, C) || (!dm == !partials) ||
        (dm && (!coef_x || !coef_1)) || ((uintptr_t)m & 15) || ((uintptr_t)dpooled & 15) || ((uintptr_t)argmax & 3) ||
        ((uintptr_t)dm & 15))
        return OSSID_EINVAL;
    StemElArgs a{};
    a.x = (const float4*)m, a.idx_in = argmax, a.g = (const float4*)dpooled, a.scale = scale, a.shift = shift, a.coef_x = coef_x;
    a.coef_1 = coef_1, a.partials = partials, a.out = (float4*)dm;
    a.B = B, a.H = H, a.W = W, a.C4 = C / 4, a.Ho = (H - 1) / 2 + 1, a.Wo = (W - 1) / 2 + 1, a.rows_per_block = POOL_ROWS;
    const dim3 grid(pool_gx(W, a.C4), ((H + 1) / 2 + POOL_ROWS - 1) / POOL_ROWS, B);
    if (dm)
        hipLaunchKernelGGL(stem_pool_bwd_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
    else
        hipLaunchKernelGGL(stem_pool_bwd_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
    return ossid_launch_status();
}

}  // extern "C"
