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
#include <type_traits>

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

// compile-time loop: the body sees its index as a constant BEFORE the optimiser's first scalar-replacement pass (a `#pragma
// unroll` loop is unrolled after it: register arrays indexed by the loop variable then live in scratch memory)
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
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
    float yold[4][16];    // [channel tile][accumulator register]
    float4 w[4][2][2];    // [channel tile][unit][hi / lo]
    float4 s[4], t[4];    // affine of this lane's 16 slab channels: [unit][half-quad]
};

// The shares are computed TRANSPOSED -- pixels on the MFMA's M axis, output channels on N (the operand layouts of
// v_mfma_f32_32x32x16_bf16 are symmetric, so the packed weights serve as the B operand as they are) -- which puts one
// CHANNEL in a lane and the tile's pixels in its 16 accumulator registers: register 4q + i of lane (n, h) is pixel
// (row q, column 4h + i) of the tile, channel 32 tt + n. A load / store of one register then covers two whole 128-byte rows
// of y (coalesced) instead of 16 bytes in each of 32 rows. rowoff[r]: that pixel's row in y (clamped into the image: lanes
// outside it read a neighbour and store nothing), all loads unconditional so nothing waits between them.
__device__ __forceinline__ void target_load(const DenseArgs& A, int m, int lane, const int (&rowoff)[16], TargetRegs& R) {
    const DenseTarget T = A.tab[m];
    const int h = lane >> 5, n = lane & 31;
    const int u0 = A.coff / 16;
    const float* yb = A.y + (size_t)m * A.P * MID + n;
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) R.yold[tt][r] = yb[(size_t)rowoff[r] * MID + 32 * tt];
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

// The same loads in slices. A wave issues in order and a CU's load path serves one 1 KB wave-instruction per ~16 cycles for
// all four waves: a block of loads in front of the 3x3's MFMA loop holds the loop back until the path has taken them all.
// So the 24 weight / affine loads are woven into the loop's last 12 (unit, tap) steps, two each -- by then the 3x3's own
// weight registers are being released, which keeps the kernel under 256 registers (beyond that the allocator parks loaded
// values in accumulator registers through a copy that waits for the load: measured 6.5 us for the loop) -- and the old sums
// are requested right after the loop, to land while the partial tiles meet in LDS.
template <int IDX>
__device__ __forceinline__ void target_load_step(const DenseArgs& A, const DenseTarget& T, int lane, TargetRegs& R) {
    const int h = lane >> 5;
    const int u0 = A.coff / 16;
    if constexpr (IDX < 16) {
        constexpr int tt = IDX >> 2, u = (IDX >> 1) & 1, k = IDX & 1;
        R.w[tt][u][k] = ldg(T.w1pk + (((size_t)tt * T.units + u0 + u) * 2 + k) * 64 + lane);
    } else {
        constexpr int q = (IDX - 16) & 3;                                     // [unit][half-quad]
        const int ch = A.coff + 16 * (q >> 1) + 8 * h + 4 * (q & 1);
        if constexpr (IDX < 20) R.s[q] = ldg(T.s1 + ch);
        else R.t[q] = ldg(T.t1 + ch);
    }
}
__device__ __forceinline__ void target_load_yold(const DenseArgs& A, int m, int lane, const int (&rowoff)[16], TargetRegs& R) {
    const float* yb = A.y + (size_t)m * A.P * MID + (lane & 31);
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) R.yold[tt][r] = yb[(size_t)rowoff[r] * MID + 32 * tt];
}

// y1_m[tile] += W1_m[:, slab] . relu(bn1_m(slab)); v = the 16 slab channels 16u + 8h + 0..7 (u = 0, 1) of pixel n
__device__ __forceinline__ void target_apply(const DenseArgs& A, int m, int lane, const int (&rowoff)[16], unsigned rowvalid,
                                             const TargetRegs& R, const float (&v)[16]) {
    const int n = lane & 31;
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
    float* yb = A.y + (size_t)m * A.P * MID + n;
    // four independent accumulator chains, interleaved product by product (one tile after the other would leave the matrix
    // pipe waiting for each chain's dependent results)
    v16f acc[4];
#pragma unroll
    for (int tt = 0; tt < 4; ++tt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[tt][r] = R.yold[tt][r];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const v8bf ah = __builtin_bit_cast(v8bf, xh[u]), al = __builtin_bit_cast(v8bf, xl[u]);      // (activations as A: the transposed product)
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, __builtin_bit_cast(v8bf, R.w[tt][u][0]), acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(v8bf, R.w[tt][u][1]), acc[tt], 0, 0, 0);
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
            acc[tt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, __builtin_bit_cast(v8bf, R.w[tt][u][0]), acc[tt], 0, 0, 0);
    }
    if (__builtin_amdgcn_ballot_w64(rowvalid != 0xffffu) == 0) {             // the whole tile inside the image: plain stores
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r) yb[(size_t)rowoff[r] * MID + 32 * tt] = acc[tt][r];
    } else {
#pragma unroll
        for (int tt = 0; tt < 4; ++tt)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (rowvalid >> r & 1) yb[(size_t)rowoff[r] * MID + 32 * tt] = acc[tt][r];
    }
}

#ifdef OSSID_DENSE_TIMING   // diagnostic build only (tools/dense_timeline.py): s_memrealtime stamps (100 MHz) of every wave of the LAST launch
__device__ unsigned long long dense_stamps[8192 * 8];
#define DSTAMP(i)                                                                                        \
    do {                                                                                                 \
        unsigned long long t_;                                                                           \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_) : : "memory");               \
        if (lane == 0 && blockIdx.x * 4 + wave < 8192) dense_stamps[(blockIdx.x * 4 + wave) * 8 + (i)] = t_; \
    } while (0)
#else
#define DSTAMP(i)
#endif

__global__ __launch_bounds__(256) void dense_layer_kernel(const DenseArgs A) {
    // 60 positions + 4 that only take the writes of the last staging round's spare threads (every load and every LDS write
    // unconditional: a load whose value is used under a condition gets sunk into it, with a wait for ALL loads behind it)
    __shared__ __attribute__((aligned(16))) float4 patch[64 * PSTR];            // 33 792 B; the partial tiles reuse it
    __shared__ __attribute__((aligned(16))) float slab[32 * SSTR];
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, n = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = blockIdx.x / A.ntiles_pad, t = blockIdx.x - g * A.ntiles_pad;
    if (t >= A.ntiles) return;
    DSTAMP(0);
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
    // ---- (1b) this wave's 3x3 weights: channel units 2 wave, 2 wave + 1, all nine taps, as a ring WRING (unit, tap) steps
    // deep: only the first steps' weights are requested in front of the staging (a wave issues in order and the CU's load
    // path takes ~16 cycles per wave-instruction: with all 36 in front, the staging started 2.1 us into the launch), the
    // rest inside the MFMA loop, WRING steps ahead of their use
    constexpr int WRING = 6;
    float4 w2r[WRING][2];
    auto w2_load = [&](auto S) {
        constexpr int step = decltype(S)::value, tap = step / 2, uu = step % 2;
#pragma unroll
        for (int k = 0; k < 2; ++k) w2r[step % WRING][k] = A.w2pk[((size_t)((2 * wave + uu) * 9 + tap) * 2 + k) * 64 + lane];
    };
    static_for<0, WRING>(w2_load);
    const int py = y0 + (n >> 3), px = x0 + (n & 7);
    const bool valid = py < H && px < W;
    const long long pix = (long long)(b * H + py) * W + px;
    const int nlater = A.nlayers - 1 - A.layer;
    const int slot = wave * A.G + g, stride = 4 * A.G;        // later layer s -> group s % G, wave s / G: few per workgroup
    int rowoff[16];                                                            // accumulator register 4q + i <-> pixel (q, 4h + i)
    unsigned rowvalid = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int yy = y0 + q, xx = x0 + 4 * h + i;
            rowoff[4 * q + i] = (b * H + min(yy, H - 1)) * W + min(xx, W - 1);
            rowvalid |= (yy < H && xx < W) ? 1u << (4 * q + i) : 0u;
        }
    DSTAMP(1);

    // ---- (1c) patch -> LDS as (hi, lo) bf16 -----------------------------------------------------------------------------
    {
        uint2* p2 = (uint2*)patch;
        const int u = j >> 2, jj = j & 3;
#pragma unroll
        for (int e = 0; e < NLD; ++e) {
            const int pos = (tid >> 5) + 8 * e;
            // (a 0 / 1 factor, not a condition: a load whose value is used under one is sunk into it, and a wait for ALL loads
            // with it; the clamped address holds a finite value)
            const float okf = ok[e] ? 1.0f : 0.0f;
            float v[4];
            v[0] = okf * fmaxf(st[e].x * s2.x + t2.x, 0.f), v[1] = okf * fmaxf(st[e].y * s2.y + t2.y, 0.f);
            v[2] = okf * fmaxf(st[e].z * s2.z + t2.z, 0.f), v[3] = okf * fmaxf(st[e].w * s2.w + t2.w, 0.f);
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
    DSTAMP(2);
    // ---- (1d) 3x3, this wave's 32 of the 128 reduction channels (one accumulator chain per unit), with (3a) woven in: the
    // loads of this wave's first later layer -- old sums, weights, affine: nothing that depends on the slab. Not in front of
    // the staging: a CU's load path takes one 1 KB wave-instruction per ~16 cycles, the 3x3 weights alone keep it busy for
    // ~1.4 us, and a wave cannot go on before all its loads are issued (tools/dense_timeline.py: the staging started 3.6 us
    // into the launch with these loads in front of it); and not as one block in front of the MFMAs, which would wait likewise.
    TargetRegs R;
    const bool has = slot < nlater;
    const int m0 = A.layer + 1 + (has ? slot : 0);
    const DenseTarget T0 = A.tab[has ? m0 : A.layer];
    v16f acc2[2];
#pragma unroll
    for (int uu = 0; uu < 2; ++uu)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc2[uu][i] = 0.f;
    // (two copies of the loop, with and without the woven loads: a branch INSIDE it makes every step wait for all
    // outstanding loads -- the wait-count pass gives up at the joins; measured 6.5 us for this phase)
    const int p0 = (n >> 3) * PC + (n & 7);
    const float4* pbase = patch + (size_t)p0 * PSTR + 2 * wave * 4 + h;
    if (has) {
        static_for<0, 18>([&](auto S) {
            constexpr int step = decltype(S)::value, tap = step / 2, uu = step % 2;
            const float4* px4 = pbase + ((tap / 3) * PC + tap % 3) * PSTR + uu * 4;
            acc2[uu] = mfma3(w2r[step % WRING][0], w2r[step % WRING][1], px4[0], px4[2], acc2[uu]);
            if constexpr (step + WRING < 18) w2_load(std::integral_constant<int, step + WRING>{});
            if constexpr (step >= 6) {
                target_load_step<2 * (step - 6)>(A, T0, lane, R);
                target_load_step<2 * (step - 6) + 1>(A, T0, lane, R);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        target_load_yold(A, m0, lane, rowoff, R);
    } else {
        static_for<0, 18>([&](auto S) {
            constexpr int step = decltype(S)::value, tap = step / 2, uu = step % 2;
            const float4* px4 = pbase + ((tap / 3) * PC + tap % 3) * PSTR + uu * 4;
            acc2[uu] = mfma3(w2r[step % WRING][0], w2r[step % WRING][1], px4[0], px4[2], acc2[uu]);
            if constexpr (step + WRING < 18) w2_load(std::integral_constant<int, step + WRING>{});
            __builtin_amdgcn_sched_barrier(0);
        });
    }
    v16f acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = acc2[0][i] + acc2[1][i];
    DSTAMP(3);
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
    DSTAMP(4);
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
    target_apply(A, A.layer + 1 + slot, lane, rowoff, rowvalid, R, v);
    DSTAMP(5);
#ifdef OSSID_DENSE_TIMING
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    DSTAMP(6);
#endif
#pragma unroll 1
    for (int s = slot + stride; s < nlater; s += stride) {
        target_load(A, A.layer + 1 + s, lane, rowoff, R);
        target_apply(A, A.layer + 1 + s, lane, rowoff, rowvalid, R, v);
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

#ifdef OSSID_DENSE_TIMING
int ossid_dense_debug_stamps(void* host_out, size_t bytes) {
    return hipMemcpyFromSymbol(host_out, HIP_SYMBOL(dense_stamps), bytes < sizeof(dense_stamps) ? bytes : sizeof(dense_stamps)) == hipSuccess
               ? OSSID_OK : OSSID_ELAUNCH;
}
#endif

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
    // The later layers' shares are vector-memory work (old sums in, weights in, new sums out: ~70 wave-instructions each), and a
    // CU's waves queue at its one texture-address unit: as many groups as later layers (one share per workgroup) while the
    // launch stays within one workgroup per CU; beyond that, one share per wave, at most ~2 workgroups per CU.
    int G = nlater < 1 ? 1 : nlater;
    while (G > 1 && nt * G > 256 && G > (nlater + 3) / 4) --G;
    while (G > 1 && nt * G > 512) --G;
    if (nt * G > 0x3fffffffLL) return OSSID_EINVAL;
    a.G = G, a.ntiles = (int)nt, a.ntiles_pad = (int)((nt + 7) / 8 * 8);
    hipLaunchKernelGGL(dense_layer_kernel, dim3((unsigned)(a.ntiles_pad * G)), dim3(256), 0, (hipStream_t)stream, a);
    return ossid_launch_status();
}

}  // extern "C"
