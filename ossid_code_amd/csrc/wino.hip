// 3x3 (stride 1, pad 1) convolutions as Winograd F(2x2, 3x3) on the f32 matrix cores (gfx950), channels-last.
//
// Same layers and the same fused prologue / epilogue as csrc/conv.hip (the dense 3x3 nn.Conv2d layers of DTOID's head,
// /root/reference/python/ossid/models/dtoid/network.py:102-110, :135-143, :288-326, and their data gradients in the
// finetune step, scripts/online_learning.py:650-679), at 16 multiplies per 2x2 output tile and channel pair instead of
// 36: the direct kernel's main loop runs the MFMA pipe 74-89 % busy at the 1.95-2.03 GHz the part holds under this load
// (tools/conv_timeline.py), so the remaining lever on these layers is the multiply count, not the schedule.
//
//   Y = A^T [ sum_ci (G g G^T) o (B^T d B) ] A      d = 4x4 input patch, g = 3x3 filter, Y = 2x2 outputs
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// GEMM view: for each of the 16 transform positions xi = (i, j):  M_xi[co][tile] = sum_ci U_xi[co][ci] * V_xi[ci][tile],
// v_mfma_f32_32x32x2_f32 with output channels on M, 32 tiles on N (lane & 31). All arithmetic f32 (the transforms only
// add, subtract and halve); results differ from the direct kernel by rounding only (tests: <= 2e-5 of the output scale).
//   - U = G g G^T is computed when the weights are packed (ossid_conv_pack_weights_wino; layout
//     [ceil(Cout/32)][Cin/8][16 xi][64 lanes][4]: lane (c,h) holds U_xi[32mt+c][8kb+4h+0..3]), streamed from L2 with a
//     two-deep register pipeline as in conv.hip
//   - V = B^T d B: a workgroup owns 32 tiles (a run of the flattened (image, tile row, tile column) index) x 64 output
//     channels. Each thread loads the three patch rows its half of the transform needs for one (tile, channel quad) of a
//     16-channel chunk (12 x 16-byte loads, fused input affine (+ReLU) applied to real pixels), transforms in registers
//     and writes 8 float4 to LDS [16 xi][32 tiles][4 quads], double buffered; next chunk's loads fly under the MFMAs
//   - waves: 2 channel tiles x 2 halves of xi (rows i in {0,1} / {2,3}): 8 accumulators of 16 registers per wave. The
//     halves of A^T M A meet in LDS after the loop: each wave keeps one output row of its tiles, gives the other away
//   - epilogue as conv.hip: bias -> ELU / ReLU -> per-channel affine -> 16-byte stores
// Cin must be a multiple of 16; odd H / W: the last tile row / column computes outputs that are not stored.
#include <stddef.h>
#include <stdlib.h>

#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));
typedef int v4i32 __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));      // (vector arithmetic lowers to v_pk_add_f32 with neg modifiers)
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));
typedef __bf16 v4bf __attribute__((ext_vector_type(4)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ float4 f4add(float4 a, float4 b) { return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w); }
__device__ __forceinline__ float4 f4sub(float4 a, float4 b) { return make_float4(a.x - b.x, a.y - b.y, a.z - b.z, a.w - b.w); }

struct WinoArgs {
    const float* x;
    const float4* wpk;
    const float *bias, *bn_scale, *bn_shift, *pre_scale, *pre_shift;
    float* out;
    int H, W, Cin, Cout, n_cotiles, act, in_cs, out_cs, out_coff, pre_relu, pre_bs;
    long long in_bs;
    int TH, TW, T;     // tiles per image (rows, columns), tiles in the batch
    int gx, gy;        // groups of 32 tiles, groups of ct channel tiles
    int ct;            // channel tiles per workgroup: 2 (256 threads) or 4 (512 threads; the input transform, which every
                       // channel group of a tile group repeats, is then shared by 128 output channels instead of 64)
    unsigned x_bytes;  // extent of x in bytes (the staging loads are bounds-checked buffer loads)
    float* dbg;        // -DOSSID_TIMING builds: per-wave time stamps (tools/conv_timeline.py --wino)
};

constexpr int KCH = 16, F4 = 4, VBUF = 16 * 32 * F4;      // float4 per LDS buffer

// `block` is the VIRTUAL block id of the launch geometry below. slice / ks / partial: the tail of the grid (its last,
// partial round of resident workgroups) is cut along the reduction -- workgroup (block, slice) walks channel chunks
// [slice * n / ks, (slice + 1) * n / ks) and stores its share of the OUTPUT-TRANSFORMED sums raw to `partial`; a finishing
// launch adds the ks shares in a fixed order and runs the epilogue (wino_finish_kernel). ks = 1, partial = nullptr: the
// whole reduction and the epilogue here.
struct WinoTail {
    int lcut, ks;          // virtual blocks [0, lcut) run whole; block lcut + t / ks, slice t % ks for t = blockIdx - lcut
    float* partial;        // [tail blocks][ks][2 ct waves][2][16][64] floats
};
constexpr int wino_partial_floats(int ct) { return 2 * ct * 2 * 16 * 64; }

__device__ __forceinline__ bool wino_block_map(const WinoArgs& A, const int L, int& bx, int& by) {
    const int P = A.gx;
    int pt;
    if (A.gy <= 8 && (8 % A.gy) == 0) {
        const int k = L & 7, R = 8 / A.gy, per = (P + R - 1) / R;
        by = k % A.gy;
        pt = (L >> 3) < per ? (k / A.gy) * per + (L >> 3) : P;
    } else if ((A.gy & 7) == 0) {
        const int j = L >> 3;
        by = (L & 7) + 8 * (j / P);
        pt = j % P;
    } else {
        by = L % A.gy;
        pt = L / A.gy;
    }
    bx = pt;
    return pt < P;
}

// bias -> (ELU / ReLU) -> per-channel affine -> 16-byte stores of output row 2 ty + wx of the block's 32 tiles
__device__ __forceinline__ void wino_epilogue(const WinoArgs& A, const int bx, const int co_tile, const int wx, const int c,
                                              const int h, const v16f (&keep)[2], const int q0 = 0, const int q1 = 4) {
    const int H = A.H, W = A.W, TPI = A.TH * A.TW;
    const int gt = bx * 32 + c;
    const bool tile_ok = gt < A.T;
    const int b = tile_ok ? gt / TPI : 0, rem = tile_ok ? gt - b * TPI : 0, ty = rem / A.TW, tx = rem - ty * A.TW;
    const int oy = 2 * ty + wx;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!tile_ok || oy >= H) break;
        if (q < q0 || q >= q1) continue;
        const int co = co_tile * 32 + 8 * q + 4 * h;
        float bi[4], sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool in = co + i < A.Cout;
            bi[i] = (A.bias && in) ? A.bias[co + i] : 0.0f;
            sc[i] = (A.bn_scale && in) ? A.bn_scale[co + i] : 1.0f;
            sh[i] = (A.bn_shift && in) ? A.bn_shift[co + i] : 0.0f;
        }
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2) {
            const int ox = 2 * tx + b2;
            if (ox >= W) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float u = keep[b2][4 * q + i] + bi[i];
                if (A.act == 1) u = elu_fast(u);
                else if (A.act == 2) u = fmaxf(u, 0.0f);
                v[i] = u * sc[i] + sh[i];
            }
            float* o = A.out + ((size_t)b * H * W + (size_t)oy * W + ox) * A.out_cs + A.out_coff + co;
            if (co + 3 < A.Cout) {
                *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i < A.Cout) o[i] = v[i];
            }
        }
    }
}

template <int CT>
__device__ __forceinline__ void wino_conv_body(const WinoArgs& A, const int block, const int slice = 0, const int ks = 1,
                                               float* __restrict__ partial = nullptr) {
    extern __shared__ __attribute__((aligned(16))) float4 vb[];   // [2][16 xi][32 tiles][4 quads]
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // uniform TO THE COMPILER: weight addresses become scalar
    const int wm = wave % CT, wx = wave / CT;
    // the input transform is the work of the first 256 threads (waves 0-3) whatever the workgroup size: 32 tiles x 4 channel
    // quads x 2 transform halves; with CT = 4 the other four waves only multiply
    const bool stager = wave < 4;
#ifdef OSSID_TIMING   // diagnostic build only: per-wave s_memrealtime stamps (100 MHz), shader cycles of the main loop, HW_ID
    auto tnow = []() {
        unsigned long long t;
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    };
    auto cnow = []() {
        unsigned long long t;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) : : "memory");
        return t;
    };
    unsigned long long tstamp[4] = {tnow(), 0, 0, 0}, cstamp[2] = {0, 0};
    auto tdump = [&]() {
        if (lane == 0 && A.dbg) {
            unsigned long long* o = (unsigned long long*)A.dbg + ((size_t)block * 2 * CT + wave) * 6;
            o[0] = tstamp[0], o[1] = tstamp[1], o[2] = tstamp[2], o[3] = tstamp[3];
            o[4] = __builtin_amdgcn_s_getreg((31 << 11) | 4);
            o[5] = (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20) | ((cstamp[1] - cstamp[0]) << 8);
        }
    };
#endif
    // ---- logical block (bx = tile group, by = channel group) from the 1-D launch id, XCD-aware as in conv.hip: every
    // XCD streams the transformed weights of ONE channel group (64 x Cin x 16 floats: 2.6 MB at Cin = 640) through its L2
    int bx, by;
    if (!wino_block_map(A, block, bx, by)) return;
    const int H = A.H, W = A.W, TPI = A.TH * A.TW;

    // ---- staging role: (tile tl, channel quad j, transform half ih) -> 3 patch rows x 4 columns ------------------
    const int j = tid & 3, tl = (tid >> 2) & 31, ih = __builtin_amdgcn_readfirstlane((tid >> 7) & 1);   // (wave-uniform)
    // Staging loads are bounds-checked buffer loads: an element outside the image (or of a tile past the last) gets the
    // byte offset 0xffffffff and the hardware returns zeros -- no per-lane branch, no select of the four loaded words.
    // Which lanes hold a real pixel for patch element k is wave-uniform DATA (a 64-bit lane mask in scalar registers,
    // built once): one v_cndmask per load picks the offset.
    unsigned gbase = 0;         // BYTE offset of patch element (row ih, column 0) incl. the image, possibly "negative"
    unsigned gmask = 0;         // bit r*4+cc: that element is a real pixel
    const float *psb = A.pre_scale, *ptb = A.pre_shift;
    {
        const int gt = bx * 32 + tl;
        const bool tv = gt < A.T;
        const int b = tv ? gt / TPI : 0, rem = tv ? gt - b * TPI : 0, ty = rem / A.TW, tx = rem - ty * A.TW;
        if (psb) psb += (size_t)b * A.pre_bs, ptb += (size_t)b * A.pre_bs;
        gbase = (unsigned)(((long long)b * A.in_bs + ((2 * ty - 1 + ih) * W + 2 * tx - 1) * A.in_cs + 4 * j) * 4);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                const int yy = 2 * ty - 1 + ih + r, xx = 2 * tx - 1 + cc;
                if (tv && yy >= 0 && yy < H && xx >= 0 && xx < W) gmask |= 1u << (r * 4 + cc);
            }
    }
    unsigned long long lanes[12];
#pragma unroll
    for (int k = 0; k < 12; ++k) lanes[k] = __ballot((gmask >> k) & 1);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc((void*)A.x, 0, A.x_bytes, 0x00020000);
    const int rstride = W * A.in_cs * 4, cstride = A.in_cs * 4;          // bytes
    v4f st[3][4];
    auto stage_load = [&](int ci0) {
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) {
                unsigned off = gbase + (unsigned)(r * rstride + cc * cstride + ci0 * 4);
                asm("v_cndmask_b32 %0, -1, %1, %2" : "=v"(off) : "v"(off), "s"(lanes[r * 4 + cc]));
                const v4i32 v = __builtin_amdgcn_raw_buffer_load_b128(xrs, (int)off, 0, 0);
                st[r][cc] = __builtin_bit_cast(v4f, v);
            }
    };
    // (the loads ONLY above: the input's affine is applied where the patch is consumed -- applied right behind the loads it was a
    // wait for them in front of the chunk's MFMAs, the latency the prefetch is there to hide)
    auto stage_affine = [&](int ci0) {
        if (psb) {      // affine (+ReLU) of the INPUT on real pixels only: the zero padding stays zero (x validity, 0 or 1)
            const v4f ps = *(const v4f*)(psb + ci0 + 4 * j), pt = *(const v4f*)(ptb + ci0 + 4 * j);
            const float lo1 = A.pre_relu ? 0.0f : -__builtin_inff();        // ReLU or nothing as ONE max (no per-word select)
            const v4f lo = {lo1, lo1, lo1, lo1};
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int cc = 0; cc < 4; ++cc) {
                    float ok;
                    asm("v_cndmask_b32 %0, 0, 1.0, %1" : "=v"(ok) : "s"(lanes[r * 4 + cc]));
                    st[r][cc] = __builtin_elementwise_max(st[r][cc] * ps + pt, lo) * ok;
                }
        }
    };
    // rows of B^T d for this half (loaded patch rows p, q, s = rows ih, ih+1, ih+2 of d), then the column transform
    auto transform_write = [&](int buf, int ci0) {
        stage_affine(ci0);
        v4f* o = (v4f*)(vb + (size_t)buf * VBUF + (size_t)tl * F4 + j);
        // a - b as fma(b, -1, a): exact, and one v_pk_fma_f32 per two floats where a plain subtraction is scalarised
        const v4f m1 = {-1.f, -1.f, -1.f, -1.f};
        auto sub = [&](v4f x, v4f y) { return __builtin_elementwise_fma(y, m1, x); };
#ifndef OSSID_WINO_F32
        // split-bf16: V = vh + vl with vh = bf16(V), vl = bf16(V - vh). LDS unit (xi, tile) = 4 x 16 bytes: slot part * 2 + khalf
        // holds channels 8 khalf .. 8 khalf + 7 of part (hi / lo) as bf16 -- one ds_read_b128 per MFMA operand. This thread
        // owns channels 4j .. 4j + 3: 8 bytes at (j & 1) * 8 of slots (0, j >> 1) and (1, j >> 1).
        // Slots are XOR-swizzled with (tile >> 2) & 3: an operand read takes one 16-byte slot of 32 consecutive tiles, 64 bytes
        // apart -- tiles t and t + 4 on the same 16 banks; unswizzled, SQ_LDS_BANK_CONFLICT was 70 % of this kernel's LDS cycles.
        const int sw = (tl >> 2) & 3;
        char* ub = (char*)(vb + (size_t)buf * VBUF + (size_t)tl * F4) + (j & 1) * 8;
        char* ob = ub + (((j >> 1) ^ sw) * 16);                  // hi piece; the lo piece: slot (2 + (j >> 1)) ^ sw
        const int lo_delta = ((((j >> 1) + 2) ^ sw) - ((j >> 1) ^ sw)) * 16;
        // (written out on 32-bit words: two v_cvt_pk_bf16_f32 give hi, a shift / a mask turn each packed half back into the
        // f32 it stands for, two packed subtractions and two more packed conversions give lo -- 10 vector instructions per
        // position; the vector-typed form of the same arithmetic compiled to 7 % more instructions in this phase)
        typedef float v2f __attribute__((ext_vector_type(2)));
        typedef __bf16 v2bf __attribute__((ext_vector_type(2)));
        auto pk = [](float x, float y) {
            const v2f t = {x, y};
            return __builtin_bit_cast(unsigned, __builtin_convertvector(t, v2bf));
        };
        auto put = [&](int xi, v4f v) {
            const unsigned p01 = pk(v[0], v[1]), p23 = pk(v[2], v[3]);
            const float l0 = v[0] - __builtin_bit_cast(float, p01 << 16), l1 = v[1] - __builtin_bit_cast(float, p01 & 0xffff0000u);
            const float l2 = v[2] - __builtin_bit_cast(float, p23 << 16), l3 = v[3] - __builtin_bit_cast(float, p23 & 0xffff0000u);
            char* p = ob + (size_t)xi * 32 * F4 * 16;
            *(uint2*)p = make_uint2(p01, p23);
            *(uint2*)(p + lo_delta) = make_uint2(pk(l0, l1), pk(l2, l3));
        };
        auto cols = [&](int i, const v4f (&R)[4]) {
            put(i * 4 + 0, sub(R[0], R[2]));
            put(i * 4 + 1, R[1] + R[2]);
            put(i * 4 + 2, sub(R[2], R[1]));
            put(i * 4 + 3, sub(R[1], R[3]));
        };
#else
        auto cols = [&](int i, const v4f (&R)[4]) {
            o[(size_t)(i * 4 + 0) * 32 * F4] = sub(R[0], R[2]);
            o[(size_t)(i * 4 + 1) * 32 * F4] = R[1] + R[2];
            o[(size_t)(i * 4 + 2) * 32 * F4] = sub(R[2], R[1]);
            o[(size_t)(i * 4 + 3) * 32 * F4] = sub(R[1], R[3]);
        };
#endif
#ifdef OSSID_WINO_ABL_RAWTF      // ablation (wrong results, timing only): the staged patch stored as it is -- the loads and their
        {                        // waits stay, the transform's vector-ALU work goes: what does the LATENCY of the patch cost?
            for (int k = 0; k < 8; ++k) {
                const v4f v = st[k % 3][k & 3];
                *(uint2*)(ob + (size_t)((2 * ih) * 4 + k) * 32 * F4 * 16) = make_uint2(__builtin_bit_cast(unsigned, v[0]), __builtin_bit_cast(unsigned, v[1]));
                *(uint2*)(ob + (size_t)((2 * ih) * 4 + k) * 32 * F4 * 16 + lo_delta) = make_uint2(__builtin_bit_cast(unsigned, v[2]), __builtin_bit_cast(unsigned, v[3]));
            }
            return;
        }
#endif
        v4f Ra[4], Rb[4];
        if (ih == 0) {            // (two code paths, not selects: ih is wave-uniform)
            asm volatile("" ::: "memory");
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) Ra[cc] = sub(st[0][cc], st[2][cc]), Rb[cc] = st[1][cc] + st[2][cc];   // d0 - d2, d1 + d2
            cols(0, Ra);
            cols(1, Rb);
        } else {
            asm volatile("" ::: "memory");
#pragma unroll
            for (int cc = 0; cc < 4; ++cc) Ra[cc] = sub(st[1][cc], st[0][cc]), Rb[cc] = sub(st[0][cc], st[2][cc]);   // d2 - d1, d1 - d3
            cols(2, Ra);
            cols(3, Rb);
        }
    };

    // ---- MFMA role: channel tile wm of the group, transform rows i in {2wx, 2wx+1}, the group's 32 tiles -----------
    const int co_tile = by * CT + wm;
    const bool active = co_tile < A.n_cotiles;
    const int nq = (A.Cin / 8) * 16;                              // 16-byte weight units per channel tile (either layout)
    const float4* W4 = A.wpk + (size_t)(active ? co_tile : 0) * nq * 64;      // (scalar; + lane at the load)
    v16f acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[e][r] = 0.0f;

    const int nchunks = A.Cin / KCH;
    const int ch0 = (int)((long long)slice * nchunks / ks), ch1 = (int)((long long)(slice + 1) * nchunks / ks);
    // CT = 4 (split-bf16 build): the two groups of four waves take turns at the input transform. The group whose turn it is
    // transforms chunk ch + 1 at the START of iteration ch -- vector-ALU work that runs beside the OTHER group's MFMAs -- and
    // multiplies afterwards; the other group requests the patch of chunk ch + 2 (one more iteration of cover for those loads)
    // and multiplies at once. With waves 0-3 transforming behind everybody's MFMAs, the matrix pipe stood still for that
    // phase: 47 % of the kernel (profiles/r03_wino_ct4.txt). CT = 2: four waves, all of them stage (two workgroups per CU run
    // out of phase by themselves).
#if !defined(OSSID_WINO_F32) && !defined(OSSID_WINO_NO_TURNS)      // (-DOSSID_WINO_NO_TURNS: the A/B switch, round 3's schedule)
    constexpr bool TURNS = CT == 4;
#else
    constexpr bool TURNS = false;
#endif
    const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);                  // 0: waves 0-3, 1: waves 4-7 (CT = 4 only)
    if (TURNS) {
        if (grp == (ch0 & 1)) {
            stage_load(ch0 * KCH);
            transform_write(ch0 & 1, ch0 * KCH);
            if (ch0 + 2 < ch1) stage_load((ch0 + 2) * KCH);        // (its registers are free again)
        } else if (ch0 + 1 < ch1) {
            stage_load((ch0 + 1) * KCH);
        }
    } else if (stager) {
        stage_load(ch0 * KCH);
        transform_write(ch0 & 1, ch0 * KCH);
    }
#ifndef OSSID_WINO_F32
    // ---- split-bf16 core: per 16-channel chunk and transform position ONE v_mfma_f32_32x32x16_bf16 triple
    // (lo*vh + hi*vl + hi*vh, small terms first) instead of eight v_mfma_f32_32x32x2_f32. Weight units of position xi of
    // chunk ch: ((ch * 16 + xi) * 2 + part) * 64 + lane. ONE register set of two positions (hi and lo of each: four 16-byte
    // loads), refilled in place: the pair of the position two steps ahead is requested right behind the three MFMAs that
    // consumed this one (the accumulators, the staged patch and the operands leave room for no more: 256 registers).
    auto unit_of = [&](int p, int part) {       // p = running position index (8 per chunk)
        const int ch = p >> 3, xi = 8 * wx + (p & 7);
        const int u = (ch * 16 + xi) * 2 + part;
        return u < nq ? u : nq - 1;
    };
    float4 wq[2][2];                            // [position parity][hi, lo]
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int k = 0; k < 2; ++k) wq[i][k] = W4[(size_t)unit_of(8 * ch0 + i, k) * 64 + lane];
    __syncthreads();
#ifdef OSSID_TIMING
    tstamp[1] = tnow();
    cstamp[0] = cnow();
#endif
    int p0 = 8 * ch0;
#pragma unroll 1
    for (int ch = ch0; ch < ch1; ++ch) {
        if (TURNS) {
            if (grp == ((ch + 1) & 1)) {
                if (ch + 1 < ch1) transform_write((ch + 1) & 1, (ch + 1) * KCH);    // beside the other group's MFMAs of this chunk
#ifndef OSSID_WINO_LATE_STAGE
                if (ch + 3 < ch1) stage_load((ch + 3) * KCH);                   // this group's next chunk: two iterations of cover
#endif
            }
        } else if (stager && ch + 1 < ch1) {
            stage_load((ch + 1) * KCH);   // in flight under this chunk's MFMAs
        }
        const int sh = h ^ ((c >> 2) & 3), sl = sh ^ 2;          // swizzled slots of this lane's hi / lo operand (transform_write)
        const float4* pb = vb + (size_t)(ch & 1) * VBUF + (size_t)(8 * wx * 32 + c) * F4;
        float4 bh = pb[sh], bl = pb[sl];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const v8bf ahi = __builtin_bit_cast(v8bf, wq[e & 1][0]), alo = __builtin_bit_cast(v8bf, wq[e & 1][1]);
            const v8bf vh = __builtin_bit_cast(v8bf, bh), vl = __builtin_bit_cast(v8bf, bl);
            const int en = e < 7 ? e + 1 : e;               // next position's operands are read under this one's MFMAs
            bh = pb[(size_t)en * 32 * F4 + sh], bl = pb[(size_t)en * 32 * F4 + sl];
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(alo, vh, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, vl, acc[e], 0, 0, 0);
            acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ahi, vh, acc[e], 0, 0, 0);
#ifndef OSSID_WINO_ABL_NOW          // (ablations, wrong results, timing only: what the loop costs without its weight stream ...
            wq[e & 1][0] = W4[(size_t)unit_of(p0 + e + 2, 0) * 64 + lane];
            wq[e & 1][1] = W4[(size_t)unit_of(p0 + e + 2, 1) * 64 + lane];
#endif
            __builtin_amdgcn_sched_barrier(0);      // keep this order: the compiler would sink the loads to their use
        }
        p0 += 8;
#ifndef OSSID_WINO_ABL_NOTF         // ... and without the input transform of the next chunk)
        if (!TURNS && stager && ch + 1 < ch1) transform_write((ch + 1) & 1, (ch + 1) * KCH);
#endif
#ifdef OSSID_WINO_LATE_STAGE     // (A/B: the patch requested BEHIND the chunk's MFMAs -- loads return in order, and a wait for a weight quad
        if (TURNS && grp == ((ch + 1) & 1) && ch + 3 < ch1) stage_load((ch + 3) * KCH);     // issued behind the patch waits for the patch too)
#endif
        __syncthreads();
    }
#else
    // prefetch group gi = (chunk, 8-channel block kb, half xh of the wave's 8 positions): 4 quads, 16 MFMAs
    auto quad_of = [&](int gi, int i) {
        const int q = ((gi >> 1) * 16) + 8 * wx + 4 * (gi & 1) + i;
        return q < nq ? q : nq - 1;
    };
    // weight quads: ONE register set, refilled in place -- the load of the quad four steps ahead (the next group's) is
    // issued right behind the four MFMAs that consumed this one, so every quad still has 16 MFMAs of cover
    float4 cur[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) cur[i] = W4[(size_t)quad_of(4 * ch0, i) * 64 + lane];
    __syncthreads();
#ifdef OSSID_TIMING
    tstamp[1] = tnow();
    cstamp[0] = cnow();
#endif

    int gi = 4 * ch0;
#pragma unroll 1
    for (int ch = ch0; ch < ch1; ++ch) {
        if (stager && ch + 1 < ch1) stage_load((ch + 1) * KCH);   // in flight under this chunk's MFMAs
        const float4* pb = vb + (size_t)(ch & 1) * VBUF + (size_t)(8 * wx * 32 + c) * F4 + h;
        float4 bq = pb[0];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int kb = g >> 1, xh = g & 1;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int e = 4 * xh + i;
                const float4 a = cur[i], b4 = bq;
                // the operand quad of the NEXT step is read while this step's MFMAs run (the last step of a chunk reads
                // its own position again: harmless)
                const int en = (i < 3) ? e + 1 : (g < 3 ? 4 * ((g + 1) & 1) : e), kbn = (i < 3) ? kb : (g < 3 ? (g + 1) >> 1 : kb);
                bq = pb[(size_t)en * 32 * F4 + 2 * kbn];
                acc[e] = mfma(a.x, b4.x, acc[e]);
                acc[e] = mfma(a.y, b4.y, acc[e]);
                acc[e] = mfma(a.z, b4.z, acc[e]);
                acc[e] = mfma(a.w, b4.w, acc[e]);
                cur[i] = W4[(size_t)quad_of(gi + 1, i) * 64 + lane];
                __builtin_amdgcn_sched_barrier(0);      // keep this order: the compiler would sink the load to its use
            }
            ++gi;
        }
        if (stager && ch + 1 < ch1) transform_write((ch + 1) & 1, (ch + 1) * KCH);
        __syncthreads();
    }
#endif

#ifdef OSSID_TIMING
    cstamp[1] = cnow();
    tstamp[2] = tnow();
#endif
    // ---- output transform. M[i][j] = acc[(i - 2wx) * 4 + j]. S = this half's share of A^T M (rows a = 0, 1), then
    // P[a][b] = (S A)[a][b]; the wave keeps row a = wx and hands row 1 - wx to its partner through LDS (V is dead).
    v16f keep[2], give[2];
    {
        v16f S0[4], S1[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            if (wx == 0) S0[jj] = acc[jj] + acc[4 + jj], S1[jj] = acc[4 + jj];           // rows 0, 1: (M0 + M1, M1)
            else S0[jj] = acc[jj], S1[jj] = -acc[jj] - acc[4 + jj];                     // rows 2, 3: (M2, -M2 - M3)
        }
        const v16f P00 = S0[0] + S0[1] + S0[2], P01 = S0[1] - S0[2] - S0[3];
        const v16f P10 = S1[0] + S1[1] + S1[2], P11 = S1[1] - S1[2] - S1[3];
        if (wx == 0) keep[0] = P00, keep[1] = P01, give[0] = P10, give[1] = P11;
        else keep[0] = P10, keep[1] = P11, give[0] = P00, give[1] = P01;
    }
    float* ex = (float*)vb;                                        // [wm][wx][b][16][64]
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 16; ++r) ex[((size_t)((wm * 2 + wx) * 2 + b2) * 16 + r) * 64 + lane] = give[b2][r];      // (CT x 2 x 2 x 4 KB: the two V buffers)
    __syncthreads();
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[b2][r] += ex[((size_t)((wm * 2 + (1 - wx)) * 2 + b2) * 16 + r) * 64 + lane];

#ifdef OSSID_TIMING
    if (!active) { tstamp[3] = tnow(); tdump(); return; }
#endif
    if (!active) return;
    if (partial) {            // a slice of the reduction: raw sums for the finishing launch
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int r = 0; r < 16; ++r) partial[((size_t)(wave * 2 + b2) * 16 + r) * 64 + lane] = keep[b2][r];
        return;
    }
    wino_epilogue(A, bx, co_tile, wx, c, h, keep);
#ifdef OSSID_TIMING
    __builtin_amdgcn_s_waitcnt(0);
    tstamp[3] = tnow();
    tdump();
#endif
}

template <int CT>
__global__ __launch_bounds__(128 * CT, 2) void wino_conv_kernel(const WinoArgs A, const WinoTail T) {
    const int b = blockIdx.x;
    if (T.ks <= 1 || b < T.lcut) {
        wino_conv_body<CT>(A, b);
    } else {
        const int t = b - T.lcut, j = t / T.ks, sl = t - j * T.ks;
        wino_conv_body<CT>(A, T.lcut + j, sl, T.ks, T.partial + ((size_t)j * T.ks + sl) * wino_partial_floats(CT));
    }
}

// The tail's second half: one workgroup per tail block adds the ks shares (fixed order: bit-reproducible) and runs the epilogue.
// (grid: tail blocks x 4 -- blockIdx.y takes one channel quad-row of the tile, so that the handful of tail blocks becomes a
// few hundred workgroups whose 8 * ks loads per thread are all independent)
template <int CT>
__device__ __forceinline__ void wino_finish_body(const WinoArgs& A, const int L, const float* __restrict__ part, const int ks,
                                                 const int q) {
    constexpr int WINO_PARTIAL_FLOATS = wino_partial_floats(CT);
    int bx, by;
    if (!wino_block_map(A, L, bx, by)) return;
    const int tid = threadIdx.x, lane = tid & 63, h = lane >> 5, c = lane & 31, wave = tid >> 6;
    const int wm = wave % CT, wx = wave / CT;
    const int co_tile = by * CT + wm;
    if (co_tile >= A.n_cotiles) return;
    v16f keep[2];
    float s[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    for (int k = 0; k < ks; ++k) {
        const float* src = part + (size_t)k * WINO_PARTIAL_FLOATS + ((size_t)(wave * 2) * 16 + 4 * q) * 64 + lane;
        float v[2][4];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int i = 0; i < 4; ++i) v[b2][i] = src[((size_t)b2 * 16 + i) * 64];
#pragma unroll
        for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
            for (int i = 0; i < 4; ++i) s[b2][i] += v[b2][i];
    }
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep[b2][r] = 0.0f;
#pragma unroll
    for (int b2 = 0; b2 < 2; ++b2)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (q == 0) keep[b2][i] = s[b2][i];
            else if (q == 1) keep[b2][4 + i] = s[b2][i];
            else if (q == 2) keep[b2][8 + i] = s[b2][i];
            else keep[b2][12 + i] = s[b2][i];
        }
    wino_epilogue(A, bx, co_tile, wx, c, h, keep, q, q + 1);
}
template <int CT>
__global__ __launch_bounds__(128 * CT) void wino_finish_kernel(const WinoArgs A, const WinoTail T) {
    wino_finish_body<CT>(A, T.lcut + blockIdx.x, T.partial + (size_t)blockIdx.x * T.ks * wino_partial_floats(CT), T.ks, blockIdx.y);
}

// Two independent layers in ONE grid (the classification and the regression trunk's i-th convolution, network.py:113-121 /
// :146-154: same shapes, different inputs and weights): 2 x 788 workgroups fill the chip's 512 slots in 3.08 rounds
// where two launches of 788 take 2 x 2.
struct WinoPair {
    WinoArgs a, b;
    int n0;
    WinoTail t;           // over the COMBINED virtual ids [0, n0 + n1)
};
template <int CT>
__global__ __launch_bounds__(128 * CT, 2) void wino_conv_pair_kernel(const WinoPair G) {
    int v = blockIdx.x, sl = 0, ks = 1;
    float* part = nullptr;
    if (G.t.ks > 1 && v >= G.t.lcut) {
        const int t = v - G.t.lcut, j = t / G.t.ks;
        sl = t - j * G.t.ks, ks = G.t.ks, v = G.t.lcut + j;
        part = G.t.partial + ((size_t)j * ks + sl) * wino_partial_floats(CT);
    }
    if (v < G.n0) wino_conv_body<CT>(G.a, v, sl, ks, part);
    else wino_conv_body<CT>(G.b, v - G.n0, sl, ks, part);
}
template <int CT>
__global__ __launch_bounds__(128 * CT) void wino_finish_pair_kernel(const WinoPair G) {
    const int v = G.t.lcut + blockIdx.x;
    const float* part = G.t.partial + (size_t)blockIdx.x * G.t.ks * wino_partial_floats(CT);
    if (v < G.n0) wino_finish_body<CT>(G.a, v, part, G.t.ks, blockIdx.y);
    else wino_finish_body<CT>(G.b, v - G.n0, part, G.t.ks, blockIdx.y);
}

// U = G g G^T packed as the kernel streams it. dgrad != 0: the weights of the data gradient (the transposed layer:
// output channels = the forward's inputs, filter rotated by 180 degrees), w stays the forward [Cout][Cin][3][3].
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, int Cout, int Cin, int dgrad,
                                                        float4* __restrict__ wpk, size_t total) {
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    wpk[i] = ossid_wino_pack_quad(w, Cout, Cin, dgrad, i);
}

}  // namespace

extern "C" {

int ossid_conv_wino_split_bf16(void) {
#ifdef OSSID_WINO_F32
    return 0;
#else
    return 1;
#endif
}

size_t ossid_conv_wino_packed_floats(int Cout, int Cin) {
    return (size_t)((Cout + 31) / 32) * (Cin / 8) * 16 * 64 * 4;
}

int ossid_conv_pack_weights_wino(const float* w, int Cout, int Cin, int dgrad, float* wpk, void* stream) {
    if (!w || !wpk || Cout <= 0 || Cin <= 0 || (dgrad ? Cout : Cin) % 16) return OSSID_EINVAL;
    const size_t total = (dgrad ? ossid_conv_wino_packed_floats(Cin, Cout) : ossid_conv_wino_packed_floats(Cout, Cin)) / 4;
    hipLaunchKernelGGL(wino_pack_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w, Cout,
                       Cin, dgrad, (float4*)wpk, total);
    return ossid_launch_status();
}

static int wino_args(const ossid_conv_desc* d, WinoArgs& a, long& nwg) {
    if (!d || !d->x || !d->wpk || !d->out) return OSSID_EINVAL;
    const int B = d->batch, H = d->height, W = d->width, Cin = d->cin, Cout = d->cout;
    if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || d->taps != 9 || d->act < 0 || d->act > 2)
        return OSSID_EINVAL;
    if ((d->src_height > 0 && d->src_height != H) || (d->src_width > 0 && d->src_width != W)) return OSSID_EINVAL;   // no fused up-sampling
    a.x = d->x, a.wpk = (const float4*)d->wpk, a.bias = d->bias, a.bn_scale = d->post_scale, a.bn_shift = d->post_shift;
    a.pre_scale = d->pre_scale, a.pre_shift = d->pre_shift, a.out = d->out, a.pre_relu = d->pre_relu;
    a.H = H, a.W = W, a.Cin = Cin, a.Cout = Cout, a.n_cotiles = (Cout + 31) / 32, a.act = d->act;
    a.in_cs = d->in_channel_stride > 0 ? d->in_channel_stride : Cin;
    a.in_bs = d->in_batch_stride >= 0 ? d->in_batch_stride : (long long)H * W * a.in_cs;
    a.pre_bs = d->pre_batch_stride;
    a.out_cs = d->out_channel_stride > 0 ? d->out_channel_stride : Cout;
    a.out_coff = d->out_channel_offset;
    a.dbg = (float*)d->scratch;
    {
        const long long xb = (a.in_bs == 0 ? (long long)H * W * a.in_cs : (long long)B * a.in_bs) * 4;
        if (xb > 0x7fffffffLL) return OSSID_EINVAL;       // 32-bit byte offsets in the staging loads
        a.x_bytes = (unsigned)xb;
    }
    if (a.in_cs < Cin || a.out_cs < a.out_coff + Cout || (a.in_cs % 4) || (a.out_cs % 4) || (a.out_coff % 4) ||
        (a.pre_scale && !a.pre_shift) || (a.in_bs % 4) || a.pre_bs < 0 || (a.pre_bs % 4) || (long long)H * W * a.in_cs > 0x7fffffffLL)
        return OSSID_EINVAL;
    a.TH = (H + 1) / 2, a.TW = (W + 1) / 2;
    const long long T = (long long)B * a.TH * a.TW;
    if (T > 0x7fffffffLL) return OSSID_EINVAL;
    a.T = (int)T;
    // 128 output channels per workgroup where the layer has them (the transform of a tile group is then done half as often);
    // OSSID_WINO_CT=2 / 4 forces one form (A/B runs)
    static const int ct_env = getenv("OSSID_WINO_CT") ? atoi(getenv("OSSID_WINO_CT")) : 0;
    // (measured, profiles/r03_wino_ct4.txt: 8-13 % on the 256 / 512-channel layers at 21 templates; at the finetune batch --
    // 75 tile groups -- the 512-thread workgroups leave the side streams' kernels less room and the step loses 0.4 ms)
    a.gx = (int)((T + 31) / 32);
    a.ct = ct_env == 2 || ct_env == 4 ? ct_env : ((a.n_cotiles >= 4 && a.gx >= 128) ? 4 : 2);
    a.gy = (a.n_cotiles + a.ct - 1) / a.ct;
    const long P = a.gx;
    if (a.gy <= 8 && 8 % a.gy == 0)
        nwg = 8 * ((P + 8 / a.gy - 1) / (8 / a.gy));
    else
        nwg = P * a.gy;
    return nwg > 0x3fffffffL ? OSSID_EINVAL : OSSID_OK;
}

// How to cut the tail of a grid of `nwg` equal workgroups on the chip's 512 slots (two per CU). Time goes in whole rounds of
// resident workgroups: 1 576 workgroups (768 -> 512 at 21 templates) cost four rounds for 3.08 rounds of work. The last,
// partial round is therefore cut along the reduction into ks slices per workgroup (one more, shorter round of `tail * ks`
// workgroups + a finishing launch) whenever that fits the slots and the reduction is long enough to split.
static void wino_plan_tail(long nwg, int nchunks, int ct, WinoTail& t) {
    const long S = ct == 4 ? 256 : 512;       // resident workgroups: two of 256 threads or one of 512 per CU
    t.lcut = (int)((nwg / S) * S), t.ks = 1, t.partial = nullptr;
    const long tail = nwg - t.lcut;
    if (tail == 0) return;
    // modelled time in units of one whole workgroup: full rounds + rounds of tail slices, each 1/ks of the reduction plus
    // a fixed share (staging ramp, output transform, raw store) + the finishing launch; calibrated on 768 -> 512 at 21
    // templates (1 576 workgroups: 0.924 ms whole = 4 rounds, 0.743 ms with the 40-workgroup tail in 12 slices)
    double best = (double)(t.lcut / S) + 1.0;
    const int cand[] = {2, 3, 4, 6, 8, 12};
    for (int ks : cand) {
        if (ks > nchunks / 4) break;                           // at least four 16-channel chunks per slice
        const double rounds = (double)((tail * ks + S - 1) / S);
        const double tm = (double)(t.lcut / S) + rounds * (1.0 / ks + 0.08) + 0.05;
        if (tm < best - 0.03) best = tm, t.ks = ks;
    }
}

size_t ossid_conv3x3_wino_workspace_bytes(const ossid_conv_desc* d) {
    WinoArgs a;
    long nwg = 0;
    if (wino_args(d, a, nwg) != OSSID_OK) return 0;
    WinoTail t;
    wino_plan_tail(nwg, a.Cin / KCH, a.ct, t);
    return t.ks > 1 ? (size_t)(nwg - t.lcut) * t.ks * wino_partial_floats(a.ct) * sizeof(float) : 0;
}

size_t ossid_conv3x3_wino_pair_workspace_bytes(const ossid_conv_desc* d0, const ossid_conv_desc* d1) {
    WinoArgs a, b;
    long n0 = 0, n1 = 0;
    if (wino_args(d0, a, n0) != OSSID_OK || wino_args(d1, b, n1) != OSSID_OK || a.Cin != b.Cin || a.ct != b.ct) return 0;
    WinoTail t;
    wino_plan_tail(n0 + n1, a.Cin / KCH, a.ct, t);
    return t.ks > 1 ? (size_t)(n0 + n1 - t.lcut) * t.ks * wino_partial_floats(a.ct) * sizeof(float) : 0;
}

int ossid_conv3x3_wino_fwd(const ossid_conv_desc* d, void* stream) {
    WinoArgs a;
    long nwg = 0;
    const int rc = wino_args(d, a, nwg);
    if (rc != OSSID_OK) return rc;
    WinoTail t;
    wino_plan_tail(nwg, a.Cin / KCH, a.ct, t);
#ifndef OSSID_TIMING
    // the tail split needs scratch for the slices' raw sums (desc->scratch, scratch_bytes);
    // without it the launch simply runs whole workgroups
    const size_t need = t.ks > 1 ? (size_t)(nwg - t.lcut) * t.ks * wino_partial_floats(a.ct) * sizeof(float) : 0;
    if (need && d->scratch && (size_t)d->scratch_bytes >= need) t.partial = (float*)d->scratch;
    else t.ks = 1;
#else
    t.ks = 1;
#endif
    const long grid = t.ks > 1 ? t.lcut + (nwg - t.lcut) * t.ks : nwg;
    const size_t lds = (size_t)2 * VBUF * 16;
    if (a.ct == 4) {
        OSSID_ENSURE_LDS(wino_conv_kernel<4>, lds);
        hipLaunchKernelGGL(wino_conv_kernel<4>, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, a, t);
        if (t.ks > 1)
            hipLaunchKernelGGL(wino_finish_kernel<4>, dim3((unsigned)(nwg - t.lcut), 4), dim3(512), 0, (hipStream_t)stream, a, t);
    } else {
        OSSID_ENSURE_LDS(wino_conv_kernel<2>, lds);
        hipLaunchKernelGGL(wino_conv_kernel<2>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, a, t);
        if (t.ks > 1)
            hipLaunchKernelGGL(wino_finish_kernel<2>, dim3((unsigned)(nwg - t.lcut), 4), dim3(256), 0, (hipStream_t)stream, a, t);
    }
    return ossid_launch_status();
}

int ossid_conv3x3_wino_fwd_pair(const ossid_conv_desc* d0, const ossid_conv_desc* d1, void* stream) {
    WinoPair g;
    long n0 = 0, n1 = 0;
    int rc = wino_args(d0, g.a, n0);
    if (rc != OSSID_OK) return rc;
    rc = wino_args(d1, g.b, n1);
    if (rc != OSSID_OK) return rc;
    g.n0 = (int)n0;
    if (g.a.ct != g.b.ct) return OSSID_EINVAL;                  // (the two layers of a pair have the same shape)
    wino_plan_tail(n0 + n1, g.a.Cin / KCH, g.a.ct, g.t);
#ifndef OSSID_TIMING
    // scratch for the tail slices: d0->scratch / scratch_bytes, see ossid_conv3x3_wino_pair_workspace_bytes
    const size_t need = g.t.ks > 1 ? (size_t)(n0 + n1 - g.t.lcut) * g.t.ks * wino_partial_floats(g.a.ct) * sizeof(float) : 0;
    if (need && g.a.Cin == g.b.Cin && d0->scratch && (size_t)d0->scratch_bytes >= need) g.t.partial = (float*)d0->scratch;
    else g.t.ks = 1;
#else
    g.t.ks = 1;
#endif
    const long tail = n0 + n1 - g.t.lcut;
    const long grid = g.t.ks > 1 ? g.t.lcut + tail * g.t.ks : n0 + n1;
    const size_t lds = (size_t)2 * VBUF * 16;
    if (g.a.ct == 4) {
        OSSID_ENSURE_LDS(wino_conv_pair_kernel<4>, lds);
        hipLaunchKernelGGL(wino_conv_pair_kernel<4>, dim3((unsigned)grid), dim3(512), lds, (hipStream_t)stream, g);
        if (g.t.ks > 1) hipLaunchKernelGGL(wino_finish_pair_kernel<4>, dim3((unsigned)tail, 4), dim3(512), 0, (hipStream_t)stream, g);
    } else {
        OSSID_ENSURE_LDS(wino_conv_pair_kernel<2>, lds);
        hipLaunchKernelGGL(wino_conv_pair_kernel<2>, dim3((unsigned)grid), dim3(256), lds, (hipStream_t)stream, g);
        if (g.t.ks > 1) hipLaunchKernelGGL(wino_finish_pair_kernel<2>, dim3((unsigned)tail, 4), dim3(256), 0, (hipStream_t)stream, g);
    }
    return ossid_launch_status();
}

}  // extern "C"
