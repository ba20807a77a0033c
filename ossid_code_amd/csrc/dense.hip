// A DenseNet block at test time with ONE launch per layer (gfx950): the bottleneck sums of all later layers are kept up to
// date incrementally instead of being recomputed from the whole concatenation.
//
// Stands behind torchvision's _DenseLayer inside ImageFeatExtract (/root/reference/python/ossid/models/dtoid/network.py:164-184:
// densenet121 features; each layer is norm1 -> ReLU -> conv1 1x1 (c -> 128) -> norm2 -> ReLU -> conv2 3x3 (128 -> 32) on the
// concatenation of everything before it) in eval mode, where both BatchNorms are per-channel affines.
//
// csrc/conv.hip runs a layer as two launches (1x1 over the c-channel prefix, 3x3), and at batch 1 -- 1 200 pixels in blocks 3
// and 4 -- each of the 116 launches is a chain of memory round trips of ~9 us that nothing overlaps. But the 1x1 is LINEAR in
// its input channels and the concatenation only ever GROWS:
//     y1_M = W1_M . relu(bn1_M(x[:, :c_M])) = sum over the slabs j < M of  W1_M[:, slab j] . relu(bn1_M(slab j))
// so the share of slab j can be added to y1_M the moment slab j exists. The layer kernel therefore does, per 4 x 8 pixel tile:
//   (1) stage relu(bn2_L(y1_L)) of the tile + halo into LDS as split-bf16 (y1_L is complete: every earlier slab has added
//       its share), 3x3 convolution 128 -> 32 with the reduction split over the four waves, partial tiles meet in LDS
//   (2) write the 32 new channels into the block's buffer, keep them in LDS
//   (3) for every LATER layer M: y1_M[tile] += W1_M[:, this slab] . relu(bn1_M(slab)) -- a 1x1 needs no halo, so the tile's
//       own slab suffices; the old sums are loaded straight into the MFMA accumulator and stored back
// One launch per layer instead of two, and its critical path is the 3x3 (K = 1152) plus a K = 32 product instead of the
// 3x3 plus a K = c <= 992 product. Step (3) is shared out over G workgroups per tile (each repeats the cheap steps (1)-(2),
// only group 0 stores the slab): with G = ceil(later / 4) every wave has ONE later layer, whose old sums, weights and affine
// are requested before step (1) starts -- the whole launch is then ~two memory round trips deep.
// The block-entry kernel computes the share of the block's INPUT channels for every layer at once (K = C0).
//
// Arithmetic: f32 tensors and accumulation, every product as three bf16 matrix-core products exactly like csrc/conv.hip's
// default form (same packed weights: ossid_conv_pack_weights of conv1 / conv2); results differ from the two-launch path by the
// ORDER of the f32 summation only. A -DOSSID_CONV_F32 build has no split form: the entry points return OSSID_EINVAL and the
// caller keeps the two-launch path.
// Traffic: y1 of all layers lives in one [L][P][128] buffer (14.7 MB for block 3 at 1 200 pixels); a tile is always handled
// by workgroups with the same id modulo 8, i.e. on the same XCD, so its read-modify-write mostly stays in that L2. For many
// pixels (a batch of images) the O(L^2) read-modify-write outweighs the launches it saves: the caller picks per block.
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v16f mfma3(const float4& whi, const float4& wlo, const float4& xhi, const float4& xlo, v16f c) {
    const v8bf ah = __builtin_bit_cast(v8bf, whi), al = __builtin_bit_cast(v8bf, wlo);
    const v8bf bh = __builtin_bit_cast(v8bf, xhi), bl = __builtin_bit_cast(v8bf, xlo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

// Pointers read from the device table are generic to the compiler (flat_load: slower, and counted on lgkmcnt as well); they
// are global addresses by contract.
typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ldg(const void* p) {
    const v4f v = *(const v4f __attribute__((address_space(1)))*)(unsigned long long)p;
    return make_float4(v.x, v.y, v.z, v.w);
}

// eight f32 -> (hi, lo) bf16 octets: one MFMA operand each
__device__ __forceinline__ void split8(const float (&v)[8], float4& hi, float4& lo) {
    union {
        __bf16 b[8];
        float4 f;
    } ph, pl;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        ph.b[e] = (__bf16)v[e];
        pl.b[e] = (__bf16)(v[e] - (float)ph.b[e]);
    }
    hi = ph.f, lo = pl.f;
}

// per layer of the block (device table, built once by the host): conv1's packed weights and norm1's affine
struct DenseTarget {
    const float4* w1pk;   // ossid_conv_pack_weights(conv1.weight [128][c_M][1]): [4 tiles][c_M/16][2 parts][64] x 16 B
    const float* s1;      // norm1 as scale / shift [c_M]
    const float* t1;
    long long units;      // c_M / 16
};

struct DenseArgs {
    float* y;             // [L][P][128] bottleneck sums
    float* buf;           // [P][ctot] the block's resident buffer
    const float4* w2pk;   // this layer's conv2, packed: [1][8 units][9 taps][2][64] x 16 B
    const float* s2;      // norm2 affine [128]
    const float* t2;
    const DenseTarget* tab;
    long long P;
    int B, H, W, ctot, coff, layer, nlayers, G, tiles_x, tiles_y, ntiles, ntiles_pad, c0;
};

constexpr int MID = 128, GROWTH = 32;
constexpr int TR = 4, TC = 8;                       // pixel tile
constexpr int PR = TR + 2, PC = TC + 2, NPOS = PR * PC;
constexpr int PSTR = MID / 16 * 4 + 1;              // float4 per patch position: 8 units x (hi, lo) x 2 halves + 1 of padding
constexpr int SSTR = GROWTH + 4;                    // floats per pixel of the slab tile in LDS

// registers of one later layer's share: requested early, used after the slab exists
struct TargetRegs {
    float4 yold[4][4];    // [channel tile][register quad]: the 16 accumulator values of this lane
    float4 w[4][2][2];    // [channel tile][unit][hi / lo]
    float4 s[4], t[4];    // affine of this lane's 16 slab channels: [unit][half-quad]
};

// pixc: an in-image pixel for every lane (a lane outside the image reads a neighbour's row and stores nothing): the loads are
// unconditional, so nothing waits between them
__device__ __forceinline__ void target_load(const DenseArgs& A, int m, int lane, long long pixc, TargetRegs& R) {
    const DenseTarget T = A.tab[m];
    const int h = lane >> 5;
    const int u0 = A.coff / 16;
    const float* yrow = A.y + ((size_t)m * A.P + pixc) * MID + 4 * h;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int q = 0; q < 4; ++q) R.yold[tt][q] = *(const float4*)(yrow + 32 * tt + 8 * q);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int k = 0; k < 2; ++k) R.w[tt][u][k] = ldg(T.w1pk + (((size_t)tt * T.units + u0 + u) * 2 + k) * 64 + lane);
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int ch = A.coff + 16 * u + 8 * h + 4 * i;
            R.s[2 * u + i] = ldg(T.s1 + ch);
            R.t[2 * u + i] = ldg(T.t1 + ch);
        }
}

// y1_m[tile] += W1_m[:, slab] . relu(bn1_m(slab)); v = this lane's 16 slab channels (unit u: v[8u .. 8u+7])
__device__ __forceinline__ void target_apply(const DenseArgs& A, int m, int lane, long long pix, bool valid, const TargetRegs& R,
                                             const float (&v)[16]) {
    const int h = lane >> 5;
    float4 xh[2], xl[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        float a[8];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 s = R.s[2 * u + i], t = R.t[2 * u + i];
            a[4 * i + 0] = fmaxf(v[8 * u + 4 * i + 0] * s.x + t.x, 0.f);
            a[4 * i + 1] = fmaxf(v[8 * u + 4 * i + 1] * s.y + t.y, 0.f);
            a[4 * i + 2] = fmaxf(v[8 * u + 4 * i + 2] * s.z + t.z, 0.f);
            a[4 * i + 3] = fmaxf(v[8 * u + 4 * i + 3] * s.w + t.w, 0.f);
        }
        split8(a, xh[u], xl[u]);
    }
    float* yrow = A.y + ((size_t)m * A.P + (valid ? pix : 0)) * MID + 4 * h;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt) {
        v16f acc;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc[4 * q + 0] = R.yold[tt][q].x, acc[4 * q + 1] = R.yold[tt][q].y;
            acc[4 * q + 2] = R.yold[tt][q].z, acc[4 * q + 3] = R.yold[tt][q].w;
        }
        acc = mfma3(R.w[tt][0][0], R.w[tt][0][1], xh[0], xl[0], acc);
        acc = mfma3(R.w[tt][1][0], R.w[tt][1][1], xh[1], xl[1], acc);
        if (valid) {
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(yrow + 32 * tt + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
        }
    }
}

__global__ __launch_bounds__(256) void dense_layer_kernel(const DenseArgs A) {
    __shared__ __attribute__((aligned(16))) float4 patch[NPOS * PSTR];          // 31 680 B; the partial tiles reuse it
    __shared__ __attribute__((aligned(16))) float slab[32 * SSTR];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x / A.ntiles_pad, t = blockIdx.x - g * A.ntiles_pad;
    if (t >= A.ntiles) return;
    const int per_img = A.tiles_x * A.tiles_y;
    const int b = t / per_img, r = t - b * per_img;
    const int y0 = (r / A.tiles_x) * TR, x0 = (r % A.tiles_x) * TC;
    const int H = A.H, W = A.W;

    // ---- (1a) the patch loads: relu(bn2(y1_L)) of the tile + halo -------------------------------------------------------
    constexpr int NLD = (NPOS * (MID / 4) + 255) / 256;                      // 8 float4 per thread
    const float* yl = A.y + (size_t)A.layer * A.P * MID;
    const int j = tid & 31;                                                    // this thread's channel quad (256 % 32 == 0)
    float4 st[NLD];
    bool ok[NLD];
#pragma unroll
    for (int e = 0; e < NLD; ++e) {                                          // (unconditional loads from clamped addresses)
        const int pos = (tid >> 5) + 8 * e;
        const int pr = pos / PC, pc = pos - pr * PC;
        const int yy = y0 - 1 + pr, xx = x0 - 1 + pc;
        ok[e] = pos < NPOS && yy >= 0 && yy < H && xx >= 0 && xx < W;
        const int yc = min(max(yy, 0), H - 1), xc = min(max(xx, 0), W - 1);
        st[e] = *(const float4*)(yl + ((size_t)(b * H + yc) * W + xc) * MID + 4 * j);
    }
    const float4 s2 = *(const float4*)(A.s2 + 4 * j), t2 = *(const float4*)(A.t2 + 4 * j);
    // ---- (1b) this wave's 3x3 weights: channel units 2 wave, 2 wave + 1, all nine taps ----------------------------------
    float4 w2[2][9][2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int k = 0; k < 2; ++k) w2[uu][tap][k] = A.w2pk[((size_t)((2 * wave + uu) * 9 + tap) * 2 + k) * 64 + lane];
    // ---- (3a) this wave's first later layer: everything it needs that does not depend on the slab -------------------------
    const int py = y0 + (n >> 3), px = x0 + (n & 7);
    const bool valid = py < H && px < W;
    const long long pix = (long long)(b * H + py) * W + px;
    const long long pixc = (long long)(b * H + min(py, H - 1)) * W + min(px, W - 1);
    const int nlater = A.nlayers - 1 - A.layer;
    const int slot = g * 4 + wave, stride = 4 * A.G;
    TargetRegs R;
    if (slot < nlater) target_load(A, A.layer + 1 + slot, lane, pixc, R);

    // ---- (1c) patch -> LDS as (hi, lo) bf16 -----------------------------------------------------------------------------
    {
        uint2* p2 = (uint2*)patch;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int pos = (tid >> 5) + 8 * e;
            if (pos >= NPOS) continue;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            if (ok[e]) {
                v[0] = fmaxf(st[e].x * s2.x + t2.x, 0.f), v[1] = fmaxf(st[e].y * s2.y + t2.y, 0.f);
                v[2] = fmaxf(st[e].z * s2.z + t2.z, 0.f), v[3] = fmaxf(st[e].w * s2.w + t2.w, 0.f);
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
            p2[(pos * PSTR + u * 4) * 2 + jj] = ph.u2;
            p2[(pos * PSTR + u * 4) * 2 + 4 + jj] = pl.u2;
        }
    }
    __syncthreads();

    // ---- (1d) 3x3, this wave's 32 of the 128 reduction channels ---------------------------------------------------------
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    {
        const int p0 = (n >> 3) * PC + (n & 7);
#pragma unroll
        for (int uu = 0; uu < 2; ++uu)
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const float4* px4 = patch + (size_t)(p0 + (tap / 3) * PC + tap % 3) * PSTR + (2 * wave + uu) * 4 + h;
                acc = mfma3(w2[uu][tap][0], w2[uu][tap][1], px4[0], px4[2], acc);
            }
    }
    __syncthreads();                                                           // the patch is dead: partial tiles go there
    float* red = (float*)patch;                                                // [4 waves][16][64]
#pragma unroll
    for (int i = 0; i < 16; ++i) red[(wave * 16 + i) * 64 + lane] = acc[i];
    __syncthreads();
    // ---- (2) wave w finishes register quad w: channels 8 w + 4 h + 0..3 of pixel n ----------------------------------------
    {
        float v[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float sum = red[(0 * 16 + 4 * wave + i) * 64 + lane];
#pragma unroll
            for (int k = 1; k < 4; ++k) sum += red[(k * 16 + 4 * wave + i) * 64 + lane];
            v[i] = sum;
        }
        const float4 o = make_float4(v[0], v[1], v[2], v[3]);
        if (g == 0 && valid) *(float4*)(A.buf + (size_t)pix * A.ctot + A.coff + 8 * wave + 4 * h) = o;
        *(float4*)(slab + n * SSTR + 8 * wave + 4 * h) = o;
    }
    __syncthreads();
    if (slot >= nlater) return;                                                // (after the last barrier)
    // ---- (3b) shares of the later layers -----------------------------------------------------------------------------------
    float v[16];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const float4 q = *(const float4*)(slab + n * SSTR + 16 * u + 8 * h + 4 * i);
            v[8 * u + 4 * i + 0] = q.x, v[8 * u + 4 * i + 1] = q.y, v[8 * u + 4 * i + 2] = q.z, v[8 * u + 4 * i + 3] = q.w;
        }
    target_apply(A, A.layer + 1 + slot, lane, pix, valid, R, v);
#pragma unroll 1
    for (int s = slot + stride; s < nlater; s += stride) {
        target_load(A, A.layer + 1 + s, lane, pixc, R);
        target_apply(A, A.layer + 1 + s, lane, pix, valid, R, v);
    }
}

// Block entry: y1_M = W1_M[:, :C0] . relu(bn1_M(x[:, :C0])) for every layer M of the block (plain stores: this initialises
// the sums). Workgroup = (32 flat pixels, layer M), wave = channel tile; the pixel run is staged once with M's affine.
struct EntryArgs {
    float* y;
    const float* buf;
    const DenseTarget* tab;
    long long P;
    int ctot, c0, nlayers, ntiles, ntiles_pad;
};

template <int C0>
__global__ __launch_bounds__(256) void dense_entry_kernel(const EntryArgs A) {
    constexpr int F4 = C0 / 4, UN = C0 / 16, ESTR = UN * 4 + 1, NLD = 32 * F4 / 256;
    static_assert(256 % F4 == 0 && (32 * F4) % 256 == 0, "C0 in 64, 128, 256, 512");
    extern __shared__ __attribute__((aligned(16))) float4 xs[];              // [32][ESTR]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int m = blockIdx.x / A.ntiles_pad, t = blockIdx.x - m * A.ntiles_pad;
    if (t >= A.ntiles) return;
    const DenseTarget T = A.tab[m];
    const long long p0 = (long long)t * 32;
    const int j = tid % F4;
    float4 st[NLD];
#pragma unroll
    for (int e = 0; e < NLD; ++e) {
        const long long p = p0 + (tid + 256 * e) / F4;
        st[e] = *(const float4*)(A.buf + (size_t)(p < A.P ? p : A.P - 1) * A.ctot + 4 * j);      // (rows past the end are never stored)
    }
    const float4 s1 = ldg(T.s1 + 4 * j), t1 = ldg(T.t1 + 4 * j);
    constexpr int GU = 4;                                                      // units per weight prefetch group
    float4 wq[2][GU][2];
    const float4* W4 = T.w1pk + (size_t)wave * T.units * 2 * 64 + lane;
#pragma unroll
    for (int i = 0; i < GU; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) wq[0][i][k] = ldg(W4 + ((size_t)i * 2 + k) * 64);
    {
        uint2* p2 = (uint2*)xs;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int pos = (tid + 256 * e) / F4;
            float v[4];
            v[0] = fmaxf(st[e].x * s1.x + t1.x, 0.f), v[1] = fmaxf(st[e].y * s1.y + t1.y, 0.f);
            v[2] = fmaxf(st[e].z * s1.z + t1.z, 0.f), v[3] = fmaxf(st[e].w * s1.w + t1.w, 0.f);
            union {
                __bf16 b4[4];
                uint2 u2;
            } ph, pl;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ph.b4[i] = (__bf16)v[i];
                pl.b4[i] = (__bf16)(v[i] - (float)ph.b4[i]);
            }
            p2[(pos * ESTR + u * 4) * 2 + jj] = ph.u2;
            p2[(pos * ESTR + u * 4) * 2 + 4 + jj] = pl.u2;
        }
    }
    __syncthreads();
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    const float4* xb = xs + (size_t)n * ESTR + h;
#pragma unroll
    for (int gq = 0; gq < UN / GU; ++gq) {
        if (gq + 1 < UN / GU) {
#pragma unroll
            for (int i = 0; i < GU; ++i)
#pragma unroll
                for (int k = 0; k < 2; ++k) wq[(gq + 1) & 1][i][k] = ldg(W4 + ((size_t)((gq + 1) * GU + i) * 2 + k) * 64);
        }
#pragma unroll
        for (int i = 0; i < GU; ++i) {
            const float4* x4 = xb + (gq * GU + i) * 4;
            acc = mfma3(wq[gq & 1][i][0], wq[gq & 1][i][1], x4[0], x4[2], acc);
        }
    }
    const long long p = p0 + n;
    if (p < A.P) {
        float* yrow = A.y + ((size_t)m * A.P + p) * MID + 32 * wave + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) *(float4*)(yrow + 8 * q) = make_float4(acc[4 * q], acc[4 * q + 1], acc[4 * q + 2], acc[4 * q + 3]);
    }
}

template <int C0>
int launch_entry(const EntryArgs& a, hipStream_t s) {
    constexpr size_t lds = (size_t)32 * (C0 / 16 * 4 + 1) * 16;
    auto k = dense_entry_kernel<C0>;
    OSSID_ENSURE_LDS(k, lds);
    hipLaunchKernelGGL(k, dim3((unsigned)(a.ntiles_pad * a.nlayers)), dim3(256), lds, s, a);
    return ossid_launch_status();
}

}  // namespace

extern "C" {

int ossid_dense_fused_available(void) { return OSSID_CONV_SB; }

size_t ossid_dense_table_bytes(int nlayers) { return (size_t)(nlayers > 0 ? nlayers : 0) * sizeof(DenseTarget); }

int ossid_dense_entry(const float* buf, int ctot, int c0, long long pixels, int nlayers, const void* table, float* y, void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!buf || !table || !y || pixels <= 0 || nlayers <= 0 || c0 > ctot || (ctot % 4)) return OSSID_EINVAL;
    EntryArgs a;
    a.y = y, a.buf = buf, a.tab = (const DenseTarget*)table, a.P = pixels, a.ctot = ctot, a.c0 = c0, a.nlayers = nlayers;
    const long long nt = (pixels + 31) / 32;
    if (nt * nlayers > 0x3fffffffLL) return OSSID_EINVAL;
    a.ntiles = (int)nt, a.ntiles_pad = (int)((nt + 7) / 8 * 8);
    hipStream_t s = (hipStream_t)stream;
    switch (c0) {
        case 64: return launch_entry<64>(a, s);
        case 128: return launch_entry<128>(a, s);
        case 256: return launch_entry<256>(a, s);
        case 512: return launch_entry<512>(a, s);
        default: return OSSID_EINVAL;
    }
}

int ossid_dense_layer(float* y, float* buf, int B, int H, int W, int ctot, int c0, int layer, int nlayers, const float* w2pk,
                      const float* s2, const float* t2, const void* table, void* stream) {
    if (!OSSID_CONV_SB) return OSSID_EINVAL;
    if (!y || !buf || !w2pk || !s2 || !t2 || !table || B <= 0 || H <= 0 || W <= 0 || layer < 0 || layer >= nlayers) return OSSID_EINVAL;
    const int coff = c0 + GROWTH * layer;
    if ((ctot % 4) || (c0 % 16) || coff + GROWTH > ctot) return OSSID_EINVAL;
    DenseArgs a;
    a.y = y, a.buf = buf, a.w2pk = (const float4*)w2pk, a.s2 = s2, a.t2 = t2, a.tab = (const DenseTarget*)table;
    a.B = B, a.H = H, a.W = W, a.ctot = ctot, a.coff = coff, a.layer = layer, a.nlayers = nlayers, a.c0 = c0;
    a.P = (long long)B * H * W;
    a.tiles_x = (W + TC - 1) / TC, a.tiles_y = (H + TR - 1) / TR;
    const long long nt = (long long)B * a.tiles_x * a.tiles_y;
    const int nlater = nlayers - 1 - layer;
    int G = (nlater + 3) / 4;
    if (G < 1) G = 1;
    // enough groups for one later layer per wave while the launch stays within ~2 workgroups per CU
    while (G > 1 && nt * G > 512) --G;
    if (nt * G > 0x3fffffffLL) return OSSID_EINVAL;
    a.G = G, a.ntiles = (int)nt, a.ntiles_pad = (int)((nt + 7) / 8 * 8);
    hipLaunchKernelGGL(dense_layer_kernel, dim3((unsigned)(a.ntiles_pad * G)), dim3(256), 0, (hipStream_t)stream, a);
    return ossid_launch_status();
}

}  // extern "C"
