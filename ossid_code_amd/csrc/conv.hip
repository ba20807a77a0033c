// 3x3 (stride 1, pad 1) and 1x1 convolutions as implicit GEMMs on the f32 matrix cores (gfx950), channels-last.
//
// Stand behind the dense nn.Conv2d layers of DTOID at test time
// (/root/reference/python/ossid/models/dtoid/network.py:102-110 classification trunk, :135-143 regression trunk,
// :288-326 correlation / fusion / segmentation-decoder convolutions, :164-184 the DenseNet-121 blocks of the image
// backbone), which the reference runs through cuDNN, with the elementwise layers around them folded in:
//   prologue  per-input-channel affine (+ReLU) applied while the input is staged: DenseNet's BN->ReLU->Conv order
//             (eval-mode BatchNorm = affine); nearest-neighbour up-sampling of the input (decoder, network.py:354-357)
//   epilogue  bias -> ELU -> per-output-channel affine: the head's `norm(F.elu(conv(x)))` (network.py:330-357)
//   in/out    channel strides and an output channel offset, so a dense block reads the first c channels of ONE resident
//             [B][H][W][C_total] buffer and appends its 32 new channels in place (no torch.cat, no re-reads)
//
// GEMM view: D[co][px] = sum_{ci,tap} W[co][ci][tap] * X[px + tap][ci].  v_mfma_f32_32x32x2_f32 with output channels
// on M (accumulator registers), pixels on N (lane&31), exact f32 arithmetic (fmaf chains; no Winograd, no reduced
// precision).
//   - weights wpk [ceil(Cout/32)][Cin/8][TAPS][64 lanes][4]: lane (c,h) holds W[32mt+c][8kb+4h+0..3][tap] -- the A
//     operands of four chained MFMAs, streamed from L2 with a two-deep register pipeline (as csrc/pn2.hip); every quad
//     feeds NT pixel tiles
//   - the input patch of a workgroup's pixels (+ halo, zero padded) is staged channels-innermost through double-buffered
//     LDS in 16-channel chunks: a B operand quad is ONE ds_read_b128, the next chunk's global loads fly under the
//     current chunk's MFMAs (issue-early / write-late), one barrier per chunk
//   - output: each lane owns 4 consecutive channels per register quad -> 16-byte stores
// Workgroup = 4 waves as WM (channel tiles) x WN (pixel groups); pixels are a flat run of the image (FLAT) or a
// segment of one row (ROWSEG = 1), or a 2-D tile of WN*NT rows x 32 columns (ROWSEG = 2: wide images of the decoder --
// a third of the halo of a row segment, no ragged last segment worth speaking of).
#include <stdlib.h>

#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// Split-bf16 arithmetic (kernel template argument FORM = 1; chosen per launch, see ossid_conv_desc::exact): every f32 operand is a pair
// of bf16 values, x = hi + lo with hi = bf16(x), lo = bf16(x - hi) (16 significant bits together), and a 16-channel slice
// of the reduction is three v_mfma_f32_32x32x16_bf16 -- w_lo*x_hi + w_hi*x_lo + w_hi*x_hi, accumulated in f32 -- instead of
// eight v_mfma_f32_32x32x2_f32: 96 pipe cycles instead of 512. The dropped w_lo*x_lo term is ~2^-16 of a product; measured
// against float64 the results sit at ~5e-6 of the output scale (exact form: ~1e-6), tests hold 2e-5. Weights are split when
// they are packed (common.h, ossid_conv_pack_quad), activations when they are staged into LDS ([position][hi of the chunk's
// channels | lo ...] bf16: an MFMA operand is one ds_read_b128 of 8 channels).
// FORM = 2, the three-way split: x = p0 + p1 + p2 with p0 = bf16(x), p1 = bf16(x - p0), p2 = bf16(x - p0 - p1) -- 24
// significant bits, i.e. the f32 value itself up to its last bit -- and six products per slice (all pairs (i, j) with
// i + j <= 2; the dropped ones are <= 2^-24 of a product, the size of f32's own rounding): f32-level accuracy (measured
// like the exact form: ~1e-6 of the output scale) at 192 pipe cycles per 16-channel slice instead of 512. For the layers
// whose output a ReLU / max-pool decides on in training (ossid_conv_desc::exact = 2).
__device__ __forceinline__ v16f mfma6(const float4 (&w)[3], const float4& x0, const float4& x1, const float4& x2, v16f c) {
    const v8bf a0 = __builtin_bit_cast(v8bf, w[0]), a1 = __builtin_bit_cast(v8bf, w[1]), a2 = __builtin_bit_cast(v8bf, w[2]);
    const v8bf b0 = __builtin_bit_cast(v8bf, x0), b1 = __builtin_bit_cast(v8bf, x1), b2 = __builtin_bit_cast(v8bf, x2);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, c, 0, 0, 0);          // smallest terms first
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, c, 0, 0, 0);
}
__device__ __forceinline__ v16f mfma3(const float4& whi, const float4& wlo, const float4& xhi, const float4& xlo, v16f c) {
    const v8bf ah = __builtin_bit_cast(v8bf, whi), al = __builtin_bit_cast(v8bf, wlo);
    const v8bf bh = __builtin_bit_cast(v8bf, xhi), bl = __builtin_bit_cast(v8bf, xlo);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, c, 0, 0, 0);
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
}

#ifndef OSSID_WPF
#define OSSID_WPF 1
#endif
// One float4 of padding per patch position: a wave's ds_read_b128 takes the same 16 bytes of 32 consecutive positions, a
// lane stride of KCH*4 bytes -- 64 / 128 / 256 / 512 bytes, i.e. every lane of a 16-lane group on the same banks. On the f32
// instruction this cost nothing measurable (round 2: the MFMAs hid it); on the split forms SQ_LDS_BANK_CONFLICT was 72-91 %
// of all LDS cycles (tools/pmc_lds_conflicts.py, profiles/r03_lds_conflicts.txt) and the padding is worth 20-40 % of a layer.
// Likewise OSSID_WPF (weight prefetch distance in groups): 1, 2 and 3 time the same; the main loop's MFMA pipe is
// 74-89 % busy at the clock the part actually holds under this load (1.95-2.03 GHz, tools/conv_timeline.py).
#ifndef OSSID_LDS_PAD
#define OSSID_LDS_PAD 1
#endif
#ifndef OSSID_MEDIUM_WGS
#define OSSID_MEDIUM_WGS 600
#endif

struct ConvArgs {
    const float* x;
    const float4* wpk;
    const float *bias, *bn_scale, *bn_shift, *pre_scale, *pre_shift;
    float* out;
    int H, W, Cin, Cout, n_cotiles, act, buf_pos, Hs, Ws, in_cs, out_cs, out_coff, pre_relu;
    long long in_bs;          // floats between consecutive images of x (0: one image shared by the whole batch)
    int pre_bs;               // floats between consecutive images' pre_scale / pre_shift rows (0: one row for all)
    int gx, gy, gz;   // logical grid: pixel blocks x channel-tile groups x images (the launch itself is 1-D)
    float scale_h, scale_w;
    float* e_partials;        // -DOSSID_TIMING builds: per-wave time stamps (desc->scratch)
    int split;                // pieces per operand - 1: 1 = split-bf16 (three products), 2 = three-way split (six), 0 = exact f32; wpk in that layout
};

// Workgroup = 4 waves = WM (channel tiles) x WK (split of the reduction) x WN (pixel groups), each wave NT pixel tiles.
// WK > 1 is for small problems (batch-1 backbone layers: a few dozen workgroups in all): the waves of a workgroup
// share ONE output tile and each walks 1/WK of every channel chunk, the partial accumulators meet in LDS.
// KCH = channels per LDS chunk (16, or 64 with split-K); NLD = float4 staged per thread per chunk; TAPS = 9 or 1, or
// 4 = one PHASE of a 3x3 convolution on a 2x nearest-neighbour up-sampled input: output pixel (2i+a, 2j+b) only ever sees
// source pixels (i+a-1, i+a) x (j+b-1, j+b), so each of the four phases (a,b) is a 2x2 convolution of the SOURCE with
// row/column-merged weights -- 4/9 of the multiply-adds of convolving the up-sampled image (network.py:354-356). Geometry
// (H, W, pixel tiles, patch) is the source's; a block's group index carries the phase; outputs go to [2H][2W].
// NPH (TAPS == 4 only) = phases per workgroup: 1 (the phase rides on the group index: four workgroups stage the same source patch)
// or 4 (ONE workgroup stages the patch once per chunk and runs all four phases off it, 4 NT accumulator tiles per wave: the
// decoder's few-channel layers are staging / set-up bound -- K = 4 taps x 64 or 128 channels -- not matrix-pipe bound).
template <int FORM, int WM, int WK, int NT, int ROWSEG, int NLD, int TAPS, int KCH, int NPH = 1>
__device__ __forceinline__ void conv_nhwc_body(const ConvArgs& A) {
    constexpr bool SB = FORM != 0;              // operands as bf16 pieces (FORM = number of pieces - 1), 16-channel units
    constexpr int WN = 4 / (WM * WK);
    constexpr int BPX = WN * NT * 32;
    constexpr int F4 = KCH / 4;                 // float4 STAGED per patch position (f32 from global memory)
    constexpr int WPQ = FORM + 1;               // pieces: float4 per weight unit and lane, 16-byte slots per (unit, lane half) in LDS
    constexpr int F4P = (FORM == 2 ? KCH / 16 * 6 : F4) + OSSID_LDS_PAD;     // float4 per patch position in LDS
    constexpr int UNIT = SB ? 16 : 8;           // reduction channels per weight unit
    constexpr int NKB = KCH / UNIT / WK;        // units of a chunk handled by one wave
    constexpr int KY = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : 1);   // prefetch groups per channel block (one kernel row each)
    constexpr int GQ = TAPS == 9 ? 3 : (TAPS == 4 ? 2 : (NKB >= 2 ? 2 : 1));   // weight quads per prefetch group
    constexpr int GPC = TAPS == 9 ? NKB * 3 : (TAPS == 4 ? NKB * 2 : NKB / GQ);   // groups per chunk per wave
    static_assert(WM * WK * WN == 4 && KCH % (UNIT * WK) == 0 && 256 % F4 == 0, "bad tiling");
    static_assert(NPH == 1 || (NPH == 4 && TAPS == 4 && WK == 1), "all-phases form: phase convolutions without split-K only");
    extern __shared__ __attribute__((aligned(16))) float4 patch[];   // [2][buf_pos][F4P]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    // the wave id is uniform, and the compiler must KNOW it: the weight-quad index (channel tile, chunk, tap) is then scalar
    // arithmetic and a weight load costs one vector instruction (base + lane) instead of six
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave % WM, wk = (wave / WM) % WK, wn = wave / (WM * WK);
#ifdef OSSID_TIMING   // diagnostic build only (tools/conv_timeline.py): per-wave s_memrealtime stamps (100 MHz, one clock for the whole chip) + HW_ID go to e_partials
    auto tnow = []() {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    };
    auto cnow = []() {                                   // shader-clock counter (per XCD origin: differences only)
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    };
    unsigned long long tstamp[4] = {tnow(), 0, 0, 0}, cstamp[2] = {0, 0};
    auto tdump = [&]() {
        if (lane == 0 && A.e_partials) {
            unsigned long long* o = (unsigned long long*)A.e_partials + ((size_t)blockIdx.x * 4 + wave) * 6;
            o[0] = tstamp[0], o[1] = tstamp[1], o[2] = tstamp[2], o[3] = tstamp[3];
            o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);      // HW_REG_HW_ID
            o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((cstamp[1] - cstamp[0]) << 8);   // XCC_ID | loop cycles
        }
    };
#endif
    // ---- logical block (bx, by, b) from the 1-D launch id, XCD-aware ------------------------------------------------
    // Workgroup ids go round-robin over the 8 XCDs, each with a private 4 MB L2. The packed weights of ONE channel-tile
    // group (WM x 32 output channels x Cin x taps: 2.9 MB for 640->256, 3.5 MB for 768->512) fit there, those of the
    // whole layer do not, so every XCD is given pixel blocks of a single group: id%8 picks (group, slot), id/8 walks
    // the pixel blocks. (A performance choice only -- any placement computes the same values.)
    int bx, by, b;
    {
        const int L = blockIdx.x, P = A.gx * A.gz;
        int pt;
        if (A.gy <= 8 && (8 % A.gy) == 0) {
            // the R = 8/gy XCDs of a group each take a CONTIGUOUS run of pixel blocks: neighbouring blocks share halo
            // rows (and, with the fused up-sampling, whole source rows), which then hit in that XCD's L2
            const int k = L & 7, R = 8 / A.gy, per = (P + R - 1) / R;
            by = k % A.gy;
            pt = (L >> 3) < per ? (k / A.gy) * per + (L >> 3) : P;
        } else if ((A.gy & 7) == 0) {
            const int j = L >> 3;
            by = (L & 7) + 8 * (j / P);
            pt = j % P;
        } else {
            by = (L / A.gx) % A.gy;
            pt = (L % A.gx) + A.gx * (L / (A.gx * A.gy));
        }
        if (pt >= P) return;
        bx = pt % A.gx;
        b = pt / A.gx;
    }
    int ph_a = 0, ph_b = 0, phase = 0;              // TAPS == 4: the phase (a, b) rides on the group index
    if (TAPS == 4 && NPH == 1) {
        const int groups = A.gy >> 2;
        phase = by / groups;
        by -= phase * groups;
        ph_a = phase >> 1, ph_b = phase & 1;
    }
    const int H = A.H, W = A.W, HW = H * W;

    // ---- geometry of this workgroup's pixels and of its patch ----------------------------------------------------
    int y_first = 0, x_first = 0, PW, PR;
    if (TAPS == 1) {
        PW = BPX;
        PR = 1;
    } else if (ROWSEG == 2) {
        const int segs = (W + 31) / 32;
        y_first = (bx / segs) * (BPX / 32);
        x_first = (bx % segs) * 32;
        PW = 34;
        PR = BPX / 32 + 2;
    } else if (ROWSEG) {
        const int segs = (W + BPX - 1) / BPX;
        y_first = bx / segs;
        x_first = (bx % segs) * BPX;
        PW = BPX + 2;
        PR = 3;
    } else {
        const int px0 = bx * BPX;
        y_first = px0 / W;
        const int y_last = min(px0 + BPX - 1, HW - 1) / W;
        PW = W + 2;
        PR = y_last - y_first + 3;
    }
    const int npos = PR * PW;

    // per-thread staging map: element e -> global float offset of its float4 (or -1: zero padding)
    int goff[NLD], lidx[NLD];
#pragma unroll
    for (int e = 0; e < NLD; ++e) {
        const int idx = tid + e * 256;
        const int pos = idx / F4, j = idx % F4;
        if (TAPS == 1) {
            const int px = bx * BPX + pos;
            const bool ok = pos < npos && px < HW;
            goff[e] = ok ? (px * A.in_cs + 4 * j) : -1;
        } else {
            const int pr = pos / PW, pc = pos - pr * PW;
            const int yy = y_first - 1 + pr, xx = x_first - 1 + pc;
            const bool ok = pos < npos && yy >= 0 && yy < H && xx >= 0 && xx < W;
            // fused nearest-neighbour upsample (F.interpolate(mode="nearest") in front of the conv): the conv reads
            // its [H][W] input straight from the [Hs][Ws] source, index = min(floor(dst * in/out), in-1)
            const int sy = (A.Hs == H) ? yy : min((int)floorf((float)yy * A.scale_h), A.Hs - 1);
            const int sx = (A.Ws == W) ? xx : min((int)floorf((float)xx * A.scale_w), A.Ws - 1);
            goff[e] = ok ? ((sy * A.Ws + sx) * A.in_cs + 4 * j) : -1;
        }
        // LDS slot of the staged quad. f32: float4 j of the position. Split form: per position and 16-channel unit 64
        // bytes [hi ch 0-7][hi ch 8-15][lo ch 0-7][lo ch 8-15] (an MFMA operand = one ds_read_b128); this thread's four
        // channels are half of one of those pieces: index in 8-byte units of its hi half, the lo half sits 4 further
        // (pieces of a position: [p0 of all its channels][p1 ...]([p2 ...]), KCH * 2 bytes each -- the threads of a position
        // write 8 consecutive bytes each, piece k sits F4 eight-byte units further)
        lidx[e] = pos >= npos ? -1 : (SB ? pos * F4P * 2 + j : pos * F4P + j);
    }

    // this lane's pixel in each of its NT tiles: patch position of tap (0,0), output pixel index (or -1)
    int pos0[NT], opx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = (wn * NT + t) * 32 + c;
        if (TAPS == 1) {
            const int px = bx * BPX + j;
            pos0[t] = j;
            opx[t] = (px < HW) ? (b * HW + px) : -1;
        } else if (ROWSEG == 2) {
            const int row = wn * NT + t, yy = y_first + row, xx = x_first + c;
            pos0[t] = row * PW + c;
            opx[t] = (yy < H && xx < W) ? (TAPS == 4 ? b * 4 * HW + (2 * yy + ph_a) * 2 * W + 2 * xx + ph_b : b * HW + yy * W + xx) : -1;
        } else if (ROWSEG) {
            const int xx = x_first + j;
            pos0[t] = j;
            opx[t] = (xx < W) ? (b * HW + y_first * W + xx) : -1;
        } else {
            const int px = bx * BPX + j;
            const int pxc = min(px, HW - 1);
            const int y = pxc / W, xx = pxc - y * W;
            pos0[t] = (y - y_first) * PW + xx;
            opx[t] = (px < HW) ? (TAPS == 4 ? b * 4 * HW + (2 * y + ph_a) * 2 * W + 2 * xx + ph_b : b * HW + px) : -1;
        }
    }

    const int co_tile = by * WM + wm;
    const bool active = co_tile < A.n_cotiles;
    const int nq = (A.Cin / UNIT) * TAPS;                       // weight units per channel tile
    const float4* W4 = A.wpk + ((size_t)phase * A.n_cotiles + (active ? co_tile : 0)) * nq * WPQ * 64;     // (+ lane at the load)
    const int ph_units = A.n_cotiles * nq;                       // weight units between two phases (NPH == 4)

    v16f acc[NPH * NT];
    {
        const int cb = (active ? co_tile : 0) * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = cb + 8 * q;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bv = (A.bias && wk == 0 && co + i < A.Cout) ? A.bias[co + i] : 0.0f;
#pragma unroll
                for (int t = 0; t < NPH * NT; ++t) acc[t][4 * q + i] = bv;
            }
        }
    }

    float4 st[NLD];
    const float* xb = A.x + (size_t)b * A.in_bs;          // this image (in_bs = 0: every batch entry reads the same one)
    const int jch = 4 * (tid % F4);   // this thread always stages channels ci0 + jch .. +3 (256 % F4 == 0)
    // the loads only: nothing here may wait for them (they fly under the chunk's MFMAs). The input affine (+ReLU) is
    // applied in stage_write, when the chunk's matrix work has been issued -- done right behind the loads it would put
    // an s_waitcnt in FRONT of the MFMA section and expose the whole memory latency once per chunk
    int st_ci0 = 0;
    auto stage_load = [&](int ci0) {
        // a ragged LAST chunk (Cin no multiple of KCH) stages zeros for the channels past Cin; their weight quads are
        // clamped to the last real one (quad_of), and 0 x w adds nothing
        const bool chan_ok = ci0 + jch < A.Cin;
        st_ci0 = ci0;
#pragma unroll
        for (int e = 0; e < NLD; ++e)
            st[e] = (goff[e] >= 0 && chan_ok) ? *(const float4*)(xb + (size_t)goff[e] + ci0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage_write = [&](int buf) {
        if (A.pre_scale && st_ci0 + jch < A.Cin) {   // BN(eval) (+ReLU) of the INPUT, on real pixels only: the zero halo stays zero
            const float4 ps = *(const float4*)(A.pre_scale + (size_t)b * A.pre_bs + st_ci0 + jch),
                         pt = *(const float4*)(A.pre_shift + (size_t)b * A.pre_bs + st_ci0 + jch);
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                if (goff[e] < 0) continue;
                float4 v = st[e];
                v.x = v.x * ps.x + pt.x, v.y = v.y * ps.y + pt.y, v.z = v.z * ps.z + pt.z, v.w = v.w * ps.w + pt.w;
                if (A.pre_relu) v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                st[e] = v;
            }
        }
        if constexpr (SB) {
            uint2* p2 = (uint2*)(patch + (size_t)buf * A.buf_pos * F4P);
#pragma unroll
            for (int e = 0; e < NLD; ++e) {
                if (lidx[e] < 0) continue;
                const float v[4] = {st[e].x, st[e].y, st[e].z, st[e].w};
                union {
                    __bf16 b[4];
                    uint2 u;
                } pc[3];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float r = v[i];
#pragma unroll
                    for (int k = 0; k < WPQ; ++k) {
                        pc[k].b[i] = (__bf16)r;
                        r -= (float)pc[k].b[i];          // exact: the remainder is representable
                    }
                }
#pragma unroll
                for (int k = 0; k < WPQ; ++k) p2[lidx[e] + F4 * k] = pc[k].u;
            }
        } else {
#pragma unroll
            for (int e = 0; e < NLD; ++e)
                if (lidx[e] >= 0) patch[(size_t)buf * A.buf_pos * F4P + lidx[e]] = st[e];
        }
    };

    // weight quads of prefetch group gi (a per-wave linear counter over (chunk, channel block, kernel row))
    auto quad_of = [&](int gi, int i) {
        int q;
        if (TAPS == 9) {
            const int ky = gi % 3, kbl = (gi / 3) % NKB, ch = gi / (3 * NKB);
            q = ((ch * (KCH / UNIT) + wk * NKB + kbl) * 9) + ky * 3 + i;
        } else if (TAPS == 4) {
            const int gl = NPH == 4 ? gi % GPC + GPC * (gi / (GPC * NPH)) : gi;      // (chunk, group) without the phase
            const int r = gl % 2, kbl = (gl / 2) % NKB, ch = gl / (2 * NKB);
            q = ((ch * (KCH / UNIT) + wk * NKB + kbl) * 4) + r * 2 + i;
            if (NPH == 4) return (q < nq ? q : nq - 1) + ((gi / GPC) % NPH) * ph_units;
        } else {
            const int g = gi % GPC, ch = gi / GPC;
            q = ch * (KCH / UNIT) + wk * NKB + g * GQ + i;
        }
        return q < nq ? q : nq - 1;
    };

    const int nchunks = (A.Cin + KCH - 1) / KCH;
    stage_load(0);
    stage_write(0);
    // weight quads ride PF prefetch groups ahead of their MFMAs in a register ring (loads return in issue order, so
    // a wait on a weight quad also waits for every staging load issued before it: PF groups of MFMAs cover both)
    // (split form: a group's MFMAs take 96 cycles per unit and pixel tile instead of 512, so the ring is deeper where a
    // group is short -- the distance has to cover an L2 round trip either way)
    constexpr int GROUP_CYCLES = GQ * NT * (FORM == 2 ? 192 : (SB ? 96 : 512));
    constexpr int PF = !SB ? OSSID_WPF : (GROUP_CYCLES >= 768 ? 1 : (GROUP_CYCLES >= 384 ? 2 : 3));
    float4 wq[PF + 1][GQ][WPQ];
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int i = 0; i < GQ; ++i)
#pragma unroll
            for (int k = 0; k < WPQ; ++k) wq[d][i][k] = W4[((size_t)quad_of(d, i) * WPQ + k) * 64 + lane];
    __syncthreads();
#ifdef OSSID_TIMING
    tstamp[1] = tnow();
    cstamp[0] = cnow();
#endif

    int gi = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) stage_load((ch + 1) * KCH);     // in flight under this chunk's MFMAs
        const float4* pb = patch + (size_t)(ch & 1) * A.buf_pos * F4P + h;
#pragma unroll
        for (int g2 = 0; g2 < GPC * NPH; ++g2) {
            const int g = g2 % GPC, ph = g2 / GPC;                // (compile-time: the loop is unrolled)
            const int pa = NPH == 4 ? (ph >> 1) : ph_a, pb2 = NPH == 4 ? (ph & 1) : ph_b;
#pragma unroll
            for (int i = 0; i < GQ; ++i)
#pragma unroll
                for (int k = 0; k < WPQ; ++k) wq[PF][i][k] = W4[((size_t)quad_of(gi + PF, i) * WPQ + k) * 64 + lane];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < GQ; ++i) {
                // (a workgroup's spare waves -- channel tiles past the last -- run the same MFMAs on tile 0's weights and
                // store nothing: a per-wave branch here would put the accumulators through VGPR<->AGPR copies and a
                // matrix-pipe drain around every group)
                const int kb = wk * NKB + (TAPS == 9 ? g / 3 : (TAPS == 4 ? g / 2 : g * GQ + i));   // unit inside the chunk
                const int toff = TAPS == 9 ? (g % 3) * PW + i : (TAPS == 4 ? (pa + g % 2) * PW + pb2 + i : 0);
                if constexpr (SB) {
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4* px = pb + (size_t)(pos0[t] + toff) * F4P + 2 * kb;      // (+ h in pb: 16 bytes = 8 channels)
                        if constexpr (FORM == 2) acc[ph * NT + t] = mfma6(wq[0][i], px[0], px[F4 / 2], px[F4], acc[ph * NT + t]);
                        else acc[ph * NT + t] = mfma3(wq[0][i][0], wq[0][i][WPQ - 1], px[0], px[F4 / 2], acc[ph * NT + t]);
                    }
                } else {
                    const float4 a = wq[0][i][0];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 bq = pb[(size_t)(pos0[t] + toff) * F4P + 2 * kb];
                        acc[ph * NT + t] = mfma(a.x, bq.x, acc[ph * NT + t]);
                        acc[ph * NT + t] = mfma(a.y, bq.y, acc[ph * NT + t]);
                        acc[ph * NT + t] = mfma(a.z, bq.z, acc[ph * NT + t]);
                        acc[ph * NT + t] = mfma(a.w, bq.w, acc[ph * NT + t]);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int d = 0; d < PF; ++d)
#pragma unroll
                for (int i = 0; i < GQ; ++i)
#pragma unroll
                    for (int k = 0; k < WPQ; ++k) wq[d][i][k] = wq[d + 1][i][k];
            ++gi;
        }
        if (ch + 1 < nchunks) stage_write((ch + 1) & 1);
        __syncthreads();
    }
    (void)KY;
#ifdef OSSID_TIMING
    cstamp[1] = cnow();
    tstamp[2] = tnow();
#endif

    // ---- split-K: the WK partial tiles meet in LDS (fixed summation order), each wave then finishes 4/WK register quads
    if (WK > 1) {
        float* red = (float*)patch;                               // [WK][WM*WN][NT][16][64]; the patch is dead now
        const int slot = wm + WM * wn;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(((size_t)(wk * (WM * WN) + slot) * NT + t) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float sum = red[(((size_t)(0 * (WM * WN) + slot) * NT + t) * 16 + r) * 64 + lane];
#pragma unroll
                for (int k = 1; k < WK; ++k) sum += red[(((size_t)(k * (WM * WN) + slot) * NT + t) * 16 + r) * 64 + lane];
                acc[t][r] = sum;
            }
    }

#ifdef OSSID_TIMING
    if (!active) { tstamp[3] = tnow(); tdump(); return; }
#endif
    if (!active) return;
    // ---- epilogue: (ELU) -> (per-channel affine) -> 16-byte stores ----------------------------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (WK > 1 && (q * WK) / 4 != wk) continue;               // with split-K the quads are shared out over the waves
        const int co = co_tile * 32 + 8 * q + 4 * h;
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = co + i < A.Cout;
            sc[i] = (A.bn_scale && in) ? A.bn_scale[co + i] : 1.0f;
            sh[i] = (A.bn_shift && in) ? A.bn_shift[co + i] : 0.0f;
        }
#pragma unroll
        for (int t2 = 0; t2 < NPH * NT; ++t2) {
            const int t = t2 % NT, ph = t2 / NT;
            if (opx[t] < 0) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float u = acc[t2][4 * q + i];
                if (A.act == 1) u = elu_fast(u);
                else if (A.act == 2) u = fmaxf(u, 0.0f);
                v[i] = u * sc[i] + sh[i];
            }
            // (NPH == 4: opx is phase (0, 0)'s pixel (2y, 2x) of the [2H][2W] output; phase (a, b) sits a rows and b columns further)
            float* o = A.out + ((size_t)opx[t] + (NPH == 4 ? (ph >> 1) * 2 * W + (ph & 1) : 0)) * A.out_cs + A.out_coff + co;
            const bool full = co + 3 < A.Cout;
            if (full) {
                *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i < A.Cout) o[i] = v[i];
            }
        }
    }
#ifdef OSSID_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tstamp[3] = tnow();
    tdump();
#endif
}

template <int FORM, int WM, int WK, int NT, int ROWSEG, int NLD, int TAPS, int KCH, int NPH = 1>
__global__ __launch_bounds__(256) void conv_nhwc_kernel(const ConvArgs A) {
    conv_nhwc_body<FORM, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>(A);
}
// ... and with two waves per SIMD asked of the register allocator (the all-phases form: 4 NT accumulator tiles per wave would
// otherwise leave one wave per SIMD to hide the staging latency behind)
template <int FORM, int WM, int WK, int NT, int ROWSEG, int NLD, int TAPS, int KCH, int NPH = 1>
__global__ __launch_bounds__(256, 2) void conv_nhwc_kernel_w2(const ConvArgs A) {
    conv_nhwc_body<FORM, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>(A);
}

// weight repack on the device: w [Cout][Cin][taps] (torch layout, taps = kh*kw) -> wpk (see the file header)
__global__ __launch_bounds__(256) void pack_conv_kernel(const float* __restrict__ w, int Cout, int Cin, int taps, int dgrad,
                                                        int exact, float4* __restrict__ wpk, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    wpk[i] = ossid_conv_pack_quad(w, Cout, Cin, taps, dgrad, exact, i);
}

template <int FORM, int WM, int WK, int NT, int ROWSEG, int NLD, int TAPS, int KCH, int NPH = 1>
int launch_conv_form(ConvArgs a, int B, hipStream_t s) {
    constexpr int WN = 4 / (WM * WK), BPX = WN * NT * 32, F4 = KCH / 4;
    // a ragged last chunk (Cin no multiple of KCH) stages zeros past Cin and clamps the weight quads: any variant, Cin % 8 == 0
    if (a.Cin % KCH && (a.Cin % 8 != 0 || KCH < 32)) return OSSID_EINVAL;
    int rows, PW, nblk;
    if (TAPS == 1) {
        rows = 1;
        PW = BPX;
        nblk = (a.H * a.W + BPX - 1) / BPX;
    } else if (ROWSEG == 2) {
        rows = BPX / 32 + 2;
        PW = 34;
        nblk = ((a.H + BPX / 32 - 1) / (BPX / 32)) * ((a.W + 31) / 32);
    } else if (ROWSEG) {
        rows = 3;
        PW = BPX + 2;
        nblk = a.H * ((a.W + BPX - 1) / BPX);
    } else {
        rows = (BPX + a.W - 2) / a.W + 1 + 2;
        if (rows > a.H + 2) rows = a.H + 2;
        PW = a.W + 2;
        nblk = (a.H * a.W + BPX - 1) / BPX;
    }
    a.buf_pos = rows * PW;
    if (a.buf_pos * F4 > NLD * 256) return OSSID_EINVAL;
    size_t lds = (size_t)2 * a.buf_pos * ((FORM == 2 ? KCH / 16 * 6 : F4) + OSSID_LDS_PAD) * 16;
    const size_t red = WK > 1 ? (size_t)WK * (WM * WN) * NT * 16 * 64 * 4 : 0;
    if (red > lds) lds = red;
    auto kern0 = [] {        // (if constexpr: only the wrapper a tiling uses is instantiated)
        if constexpr (NPH == 4) return &conv_nhwc_kernel_w2<FORM, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>;
        else return &conv_nhwc_kernel<FORM, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>;
    }();
    OSSID_ENSURE_LDS(kern0, lds);
    a.gx = nblk, a.gy = (a.n_cotiles + WM - 1) / WM * ((TAPS == 4 && NPH == 1) ? 4 : 1), a.gz = B;
    const long P = (long)a.gx * a.gz;
    long nwg;
    if (a.gy <= 8 && 8 % a.gy == 0)
        nwg = 8 * ((P + 8 / a.gy - 1) / (8 / a.gy));
    else
        nwg = P * a.gy;          // gy a multiple of 8 (exact) or the plain mapping
    if (nwg > 0x7fffffffL) return OSSID_EINVAL;
    hipLaunchKernelGGL(kern0, dim3((unsigned)nwg), dim3(256), lds, s, a);
    return ossid_launch_status();
}

// a.split picks the arithmetic; every tiling exists in both forms except where a wave's share of a chunk would be half a
// 16-channel unit (KCH / WK == 8: the caller names the split form's tiling separately)
template <int WM, int WK, int NT, int ROWSEG, int NLD, int TAPS, int KCH, int NPH = 1>
int launch_conv(const ConvArgs& a, int B, hipStream_t s) {
    if constexpr (KCH % (16 * WK) == 0) {
        if (a.split == 1) return launch_conv_form<1, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>(a, B, s);
        if (a.split == 2) return launch_conv_form<2, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>(a, B, s);
    }
    return launch_conv_form<0, WM, WK, NT, ROWSEG, NLD, TAPS, KCH, NPH>(a, B, s);
}

}  // namespace

extern "C" {

int ossid_conv_split_bf16(void) { return OSSID_CONV_SB; }

size_t ossid_conv_packed_floats(int Cout, int Cin, int taps) {
    return (size_t)((Cout + 31) / 32) * (Cin / 8) * taps * 64 * 4;
}

size_t ossid_conv_packed_floats_form(int Cout, int Cin, int taps, int exact) {
    const size_t n = ossid_conv_packed_floats(Cout, Cin, taps);
    return (exact == 2 && OSSID_CONV_SB) ? n / 2 * 3 : n;       // three pieces per value instead of two
}

int ossid_conv_pack_weights_form(const float* w, int Cout, int Cin, int taps, int dgrad, int exact, float* wpk, void* stream) {
    if (!w || !wpk || Cout <= 0 || Cin <= 0 || (dgrad ? Cout : Cin) % 16 || (taps != 1 && taps != 9 && taps != 4)) return OSSID_EINVAL;
    if (exact < 0 || exact > 2) return OSSID_EINVAL;
    const size_t total = (dgrad ? ossid_conv_packed_floats_form(Cin, Cout, taps, exact) : ossid_conv_packed_floats_form(Cout, Cin, taps, exact)) / 4;
    hipLaunchKernelGGL(pack_conv_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       Cout, Cin, taps, dgrad ? 1 : 0, exact, (float4*)wpk, total);
    return ossid_launch_status();
}

int ossid_conv_pack_weights(const float* w, int Cout, int Cin, int taps, float* wpk, void* stream) {
    return ossid_conv_pack_weights_form(w, Cout, Cin, taps, 0, 0, wpk, stream);
}

int ossid_conv_nhwc_fwd(const ossid_conv_desc* d, void* stream) {
    if (!d) return OSSID_EINVAL;
    const int B = d->batch, H = d->height, W = d->width, Cin = d->cin, Cout = d->cout;
    if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || (Cout % 4) || B > 65535) return OSSID_EINVAL;
    if (d->taps != 1 && d->taps != 9 && d->taps != 4) return OSSID_EINVAL;
    if (B == 0) return OSSID_OK;
    if (!d->x || !d->wpk || !d->out || d->act < 0 || d->act > 2) return OSSID_EINVAL;
    ConvArgs a;
    a.x = d->x, a.wpk = (const float4*)d->wpk, a.bias = d->bias, a.bn_scale = d->post_scale, a.bn_shift = d->post_shift;
    a.pre_scale = d->pre_scale, a.pre_shift = d->pre_shift, a.pre_relu = d->pre_relu, a.out = d->out;
    a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.n_cotiles = (Cout + 31) / 32, a.act = d->act, a.buf_pos = 0;
    a.Hs = d->src_height > 0 ? d->src_height : H, a.Ws = d->src_width > 0 ? d->src_width : W;
    a.in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : Cin;
    a.in_bs = d->in_batch_stride >= 0 ? d->in_batch_stride : (long long)a.Hs * a.Ws * a.in_cs;
    a.pre_bs = d->pre_batch_stride;
    a.out_cs = d->out_channel_stride > 0 ? d->out_channel_stride : Cout;
    a.out_coff = d->out_channel_offset;
    if (a.Hs > H || a.Ws > W || a.in_cs < Cin || a.out_cs < a.out_coff + Cout || (a.in_cs % 4) || (a.out_cs % 4) ||
        (a.out_coff % 4) || (a.pre_scale && !a.pre_shift) || (a.in_bs % 4) || a.pre_bs < 0 || (a.pre_bs % 4) || (d->taps == 1 && (a.Hs != H || a.Ws != W)))
        return OSSID_EINVAL;
    a.scale_h = (float)a.Hs / (float)H, a.scale_w = (float)a.Ws / (float)W;
    a.e_partials = (float*)d->scratch;       // (-DOSSID_TIMING builds only: per-wave time stamps)
    if (d->exact < 0 || d->exact > 2) return OSSID_EINVAL;
    a.split = !OSSID_CONV_SB ? 0 : (d->exact == 0 ? 1 : (d->exact == 2 ? 2 : 0));
    hipStream_t s = (hipStream_t)stream;
    const int tiles = a.n_cotiles;
    const long px = (long)B * H * W;
    // How many workgroups would the plain tiling (128 output channels x 64 pixels) give? Under ~0.6 per CU the problem is
    // "small" (batch-1 backbone layers, batch-8 head layers): one channel tile per workgroup, the four waves split
    // the reduction instead (64-channel chunks), which multiplies the workgroup count by up to 4 and cuts every wave's
    // MFMA chain -- the critical path of such a launch -- by 4.
    const long plain_wgs = (px + 63) / 64 * ((tiles + 3) / 4);
    if (d->taps == 1) {   // no halo: the patch is just the pixel run
        // small: under one workgroup per CU, so LDS is no constraint and the launch is a chain of memory latencies, one
        // per chunk: 256-channel chunks (ragged last one: DenseNet's widths are multiples of 32), 64 channels per wave
        // four channel tiles x 32 pixels with 64-channel chunks (a ragged last one for DenseNet's widths): with 16-channel
        // chunks a 1x1 layer has only 8 MFMAs per wave between barriers. Measured on the batch-8 DenseNet layers
        // (64..1024 -> 128..640 at 9 k..154 k pixels): forward 45 -> 56, data gradient 55 -> 59 TFLOP/s over the set
        // (measured against: two channel tiles x split-K, one tile x split-K in 256-channel chunks, two pixel tiles per wave,
        // 128- / 256-channel chunks -- DESIGN.md 5c / 5e; none of them is compiled in any more)
        if (tiles >= 4 && px >= 4096) return launch_conv<4, 1, 1, false, 2, 1, 64>(a, B, s);
        if (plain_wgs < 160) return launch_conv<1, 4, 1, false, 8, 1, 256>(a, B, s);
        if (tiles >= 4)
            return px >= 128L * 512 ? launch_conv<4, 1, 4, false, 2, 1, 16>(a, B, s)
                                    : launch_conv<4, 1, 1, false, 1, 1, 16>(a, B, s);
        if (tiles >= 2) return launch_conv<2, 1, 2, false, 2, 1, 16>(a, B, s);
        return launch_conv<1, 1, 1, false, 2, 1, 16>(a, B, s);
    }
    if (d->taps == 4) {       // the four phases of a 3x3 conv on a 2x up-sampled input; H, W = SOURCE size, out = [B][2H][2W]
        if (a.Hs != H || a.Ws != W || a.pre_scale) return OSSID_EINVAL;
        const bool wide = W > 100;
        const long per128 = wide ? (long)((H + 3) / 4) * ((W + 31) / 32) : ((long)H * W + 127) / 128;
        if (tiles >= 4) {
            if (wide) return launch_conv<4, 1, 2, 2, 8, 4, 16>(a, B, s);
            const long groups = (tiles + 3) / 4 * 4;      // the phases multiply the workgroup count
            int best = 2;
            double best_cost = 1e30;
            for (int nt = 1; nt <= 4; ++nt) {
                const long nwg = (((long)H * W + nt * 32 - 1) / (nt * 32)) * B * groups;
                const double cost = (double)((nwg + 511) / 512) * (nt + 0.25);
                if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nt > best)) best = nt, best_cost = cost;
            }
            switch (best) {
                case 1: return launch_conv<4, 1, 1, 0, 8, 4, 16>(a, B, s);
                case 2: return launch_conv<4, 1, 2, 0, 8, 4, 16>(a, B, s);
                case 3: return launch_conv<4, 1, 3, 0, 8, 4, 16>(a, B, s);
                default: return launch_conv<4, 1, 4, 0, 8, 4, 16>(a, B, s);
            }
        }
        (void)per128;
#ifndef OSSID_PHASE_SPLIT      // (-DOSSID_PHASE_SPLIT: one phase per workgroup everywhere, the A/B build)
        // the decoder's few-channel layers on wide images (128 -> 64 at 116 x 156, 64 -> 32 at 232 x 312): all four phases
        // per workgroup off ONE staged patch
        if (a.split == 1) {
            // (one 32-pixel tile per wave and phase: 3.601 vs 3.624 ms with two)
            if (wide) return tiles >= 2 ? launch_conv<2, 1, 1, 2, 8, 4, 16, 4>(a, B, s) : launch_conv<1, 1, 1, 2, 8, 4, 16, 4>(a, B, s);
            // (... and a tie with one tile per wave: 3.611 vs 3.608)
            // (the narrow layers -- 128 -> 64 at 58 x 78, 256 -> 128 at 29 x 39 -- measured slower in this form: graph 3.656 vs 3.596 ms)
        }
#endif
        if (tiles >= 2) return wide ? launch_conv<2, 1, 2, 2, 8, 4, 16>(a, B, s) : launch_conv<2, 1, 2, 0, 8, 4, 16>(a, B, s);
        return wide ? launch_conv<1, 1, 2, 2, 8, 4, 16>(a, B, s) : launch_conv<1, 1, 1, 0, 8, 4, 16>(a, B, s);
    }
    const bool rowseg = W > 100;
    // workgroups of the variant the rules below would pick on a wide image (2-D tiles of 8 / 4 / NT rows x 32 columns)
    const long wide_wgs = !rowseg ? (1L << 40)
                                  : (long)B * ((W + 31) / 32) * (tiles >= 4 ? (H + 1) / 2 * ((tiles + 3) / 4)
                                                                            : tiles >= 2 ? (H + 3) / 4 : (H + 7) / 8);
    // under one workgroup per CU either way: row segments of 32 pixels, one channel tile, the reduction split over the
    // four waves in 64-channel chunks
    // (only for Cin >= 128: with fewer input channels three of the four waves would multiply the zero padding of the
    // chunk -- the data gradient of a dense layer's 128->32 conv is such a 32->128 layer)
    // (64-channel chunks: with 128 the first chunk's loads -- the launch's start-up latency -- are twice as long and the
    // three-way split's patch would take 156 KB of LDS; measured on 128 -> 32 at 30 x 40 x 8: 23 -> 18 us, 17 in the three-way form)
    if ((plain_wgs < 160 || wide_wgs < 256) && Cin >= 128) return launch_conv<1, 4, 1, true, 7, 9, 64>(a, B, s);
    // medium problems on narrow images (the 29x39 head at n_t ~ 10 or batch 8: ~1 plain workgroup per CU, i.e. one or
    // two waves per SIMD and a ragged tail): one channel tile x 32 flat pixels per workgroup, reduction split over the
    // four waves in 32-channel chunks -> 8x as many, 4x shorter work items
    const int rows32 = (32 + W - 2) / W + 3;                          // patch rows of a 32-pixel flat run
    if (plain_wgs < OSSID_MEDIUM_WGS && (Cin % 32) == 0 && (rows32 < H + 2 ? rows32 : H + 2) * (W + 2) * 8 <= 6 * 256) {
        // two channel tiles x 32 pixels, the reduction split over two waves: measured 5-8 % faster than one tile split
        // over four on the batch-8 head layers (256..640 -> 256/512 at 29x39), fewer partial tiles through LDS
        if (tiles >= 2) return launch_conv<2, 2, 1, false, 6, 9, 32>(a, B, s);
        if (a.split) return launch_conv<1, 4, 1, false, 12, 9, 64>(a, B, s);   // a wave's share of a chunk must be a whole 16-channel unit
        return launch_conv<1, 4, 1, false, 6, 9, 32>(a, B, s);
    }
#define OSSID_CONV(WM_, NT_)                                                                                         \
    (rowseg ? launch_conv<WM_, 1, NT_, 2, 8, 9, 16>(a, B, s) : launch_conv<WM_, 1, NT_, 0, 8, 9, 16>(a, B, s))
    // waves go to channel tiles while there are at least that many; the rest of the workgroup takes more pixels.
    // 128 pixels per workgroup unless that leaves the 256 CUs with under ~3 workgroups each. (256-pixel tiles with 13
    // staged float4 per thread were measured slower on the decoder's few-channel layers: their K is only 288-576, so
    // the per-workgroup set-up, not the weight stream, is what counts.)
    if (tiles >= 4) {
        // Pixel tiles per wave (NT x 32 pixels per workgroup): two workgroups fit a CU, and the tail of a launch packs
        // them two to a CU again, so time goes in whole "rounds" of 512 workgroups. Pick the NT with the least
        // rounds x tile work (a quarter tile of fixed cost per workgroup favours the larger tile on a draw).
        const long groups = (tiles + 3) / 4;
        int best = 2;
        double best_cost = 1e30;
        for (int nt = 1; nt <= 4; ++nt) {
            const long per = rowseg ? (long)((H + nt - 1) / nt) * ((W + 31) / 32) : ((long)H * W + nt * 32 - 1) / (nt * 32);
            const long nwg = per * B * groups;
            const double cost = (double)((nwg + 511) / 512) * (nt + 0.25);
            if (cost < best_cost - 1e-9 || (cost < best_cost + 1e-9 && nt > best)) best = nt, best_cost = cost;
        }
        switch (best) {
            case 1: return OSSID_CONV(4, 1);
            case 2: return OSSID_CONV(4, 2);
            case 3: return OSSID_CONV(4, 3);
            default: return OSSID_CONV(4, 4);
        }
    }
    if (tiles >= 2) return OSSID_CONV(2, 2);
    return rowseg ? OSSID_CONV(1, 2) : OSSID_CONV(1, 1);   // wide image: 8 rows x 32 columns, each weight quad feeds two rows
#undef OSSID_CONV
}

}  // extern "C"

// =====================================================================================================================
// Round-1 entry points of the 3x3 weight gradient, kept for ABI stability: thin wrappers around ossid_conv_wgrad
// (csrc/train.hip: LDS-staged MFMA kernel, deterministic split-K).
extern "C" {

int ossid_conv3x3_wgrad_splits(int B, int H, int W, int Cin, int Cout) {
    const size_t per = (size_t)Cout * Cin * 9 * sizeof(float);
    const size_t bytes = ossid_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 9);
    return per ? (int)(bytes / per) : 0;
}

size_t ossid_conv3x3_wgrad_workspace_bytes(int B, int H, int W, int Cin, int Cout) {
    return ossid_conv_wgrad_workspace_bytes(B, H, W, Cin, Cout, 9);
}

int ossid_conv3x3_wgrad(const float* x, const float* dy, int B, int H, int W, int Cin, int Cout, int in_channel_stride,
                        int dy_channel_stride, void* workspace, size_t workspace_bytes, float* dw, int accumulate,
                        void* stream) {
    ossid_wgrad_desc d = {};
    d.x = x, d.dy = dy, d.dw = dw, d.workspace = workspace, d.workspace_bytes = workspace_bytes;
    d.batch = B, d.height = H, d.width = W, d.cin = Cin, d.cout = Cout, d.taps = 9, d.accumulate = accumulate;
    d.in_channel_stride = in_channel_stride, d.dy_channel_stride = dy_channel_stride;
    return ossid_conv_wgrad(&d, stream);
}

}  // extern "C"
