// 3x3 / stride 1 / pad 1 convolution as an implicit GEMM on the f32 matrix cores (gfx950), channels-last.
//
// Stands behind the dense 3x3 nn.Conv2d layers of the DTOID head at test time
// (/root/reference/python/ossid/models/dtoid/network.py:102-110 classification trunk, :135-143 regression trunk,
// :288-326 correlation / fusion / segmentation-decoder convolutions), which the reference runs through cuDNN.
// Optional fused epilogue: bias -> ELU -> BatchNorm(eval) affine, i.e. the reference's `norm(F.elu(conv(x)))` pattern
// (network.py:330-357) in one pass over the output.
//
// GEMM view: D[co][px] = sum_{ci,tap} W[co][ci][tap] * X[px + tap][ci].  v_mfma_f32_32x32x2_f32 with output channels
// on M (accumulator registers), pixels on N (lane&31), exact f32 arithmetic (fmaf chains; no Winograd, no reduced
// precision).
//   - input  x   [B][H][W][Cin]   (NHWC; Cin % 16 == 0)
//   - weight wpk [ceil(Cout/32)][Cin/8][9][64 lanes][4]: lane (c,h) holds W[32mt+c][8kb+4h+0..3][tap] -- the A operands
//     of four chained MFMAs, streamed from L2 with a two-deep register pipeline (as csrc/pn2.hip); every quad feeds
//     NT pixel tiles, so weight traffic is 1/(4 NT) dword per MFMA
//   - the input patch of a workgroup's pixels (+ halo, zero padded) is staged channels-innermost through double-buffered
//     LDS in 16-channel chunks: a B operand quad is ONE ds_read_b128, the next chunk's global loads fly under the
//     current chunk's MFMAs (issue-early / write-late), one barrier per chunk
//   - output out [B][H][W][Cout]: each lane owns 4 consecutive channels per register quad -> 16-byte stores
// Workgroup = 4 waves as WM (channel tiles) x WN (pixel groups); pixels are a flat run of the image (FLAT, narrow
// images such as the 29x39 feature map) or a segment of one row (ROWSEG, wide images of the decoder).
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

constexpr int KC = 16;          // channels per LDS chunk
constexpr int GQ = 3;           // weight quads per prefetch group (one kernel row)

// NLD = float4 staged per thread per chunk (the patch has at most NLD*64 positions)
template <int WM, int NT, bool ROWSEG, int NLD>
__global__ __launch_bounds__(256) void conv3x3_nhwc_kernel(const float* __restrict__ x, const float4* __restrict__ wpk,
                                                           const float* __restrict__ bias,
                                                           const float* __restrict__ bn_scale,
                                                           const float* __restrict__ bn_shift, float* __restrict__ out,
                                                           int H, int W, int Cin, int Cout, int n_cotiles, int act,
                                                           int buf_pos, int Hs, int Ws, float scale_h, float scale_w) {
    constexpr int WN = 4 / WM;
    constexpr int BPX = WN * NT * 32;
    extern __shared__ __attribute__((aligned(16))) float4 patch[];   // [2][buf_pos][KC/4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int wm = wave % WM, wn = wave / WM;
    const int b = blockIdx.z;
    const int HW = H * W;

    // ---- geometry of this workgroup's pixels and of its patch ----------------------------------------------------
    int y_first, x_first, PW, PR;
    if (ROWSEG) {
        const int segs = (W + BPX - 1) / BPX;
        y_first = blockIdx.x / segs;
        x_first = (blockIdx.x % segs) * BPX;
        PW = BPX + 2;
        PR = 3;
    } else {
        const int px0 = blockIdx.x * BPX;
        y_first = px0 / W;
        x_first = 0;
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
        const int pos = idx >> 2, j = idx & 3;
        const int pr = pos / PW, pc = pos - pr * PW;
        const int yy = y_first - 1 + pr, xx = x_first - 1 + pc;
        const bool ok = pos < npos && yy >= 0 && yy < H && xx >= 0 && xx < W;
        // fused nearest-neighbour upsample (F.interpolate(mode="nearest") in front of the conv, network.py:354-357):
        // the conv reads its [H][W] input straight from the [Hs][Ws] source, index = min(floor(dst * in/out), in-1)
        const int sy = (Hs == H) ? yy : min((int)floorf((float)yy * scale_h), Hs - 1);
        const int sx = (Ws == W) ? xx : min((int)floorf((float)xx * scale_w), Ws - 1);
        goff[e] = ok ? (((b * Hs + sy) * Ws + sx) * Cin + 4 * j) : -1;
        lidx[e] = pos < npos ? idx : -1;
    }

    // this lane's pixel in each of its NT tiles: patch position of tap (0,0), validity, output offset
    int pos0[NT], opx[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int j = (wn * NT + t) * 32 + c;
        if (ROWSEG) {
            const int xx = x_first + j;
            pos0[t] = min(j, BPX - 1);
            opx[t] = (xx < W) ? (b * HW + y_first * W + xx) : -1;
        } else {
            const int px = blockIdx.x * BPX + j;
            const int pxc = min(px, HW - 1);
            const int y = pxc / W, xx = pxc - y * W;
            pos0[t] = (y - y_first) * PW + xx;
            opx[t] = (px < HW) ? (b * HW + px) : -1;
        }
    }

    const int co_tile = blockIdx.y * WM + wm;
    const bool active = co_tile < n_cotiles;
    const int nq = (Cin / 8) * 9;                               // weight quads per channel tile
    const float4* W4 = wpk + (size_t)(active ? co_tile : 0) * nq * 64 + lane;

    v16f acc[NT];
    {
        const int cb = (active ? co_tile : 0) * 32 + 4 * h;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int co = cb + 8 * q;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float bv = (bias && co + i < Cout) ? bias[co + i] : 0.0f;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t][4 * q + i] = bv;
            }
        }
    }

    float4 st[NLD];
    auto stage_load = [&](int ci0) {
#pragma unroll
        for (int e = 0; e < NLD; ++e)
            st[e] = goff[e] >= 0 ? *(const float4*)(x + (size_t)goff[e] + ci0) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto stage_write = [&](int buf) {
#pragma unroll
        for (int e = 0; e < NLD; ++e)
            if (lidx[e] >= 0) patch[(size_t)buf * buf_pos * 4 + lidx[e]] = st[e];
    };

    const int nchunks = Cin / KC;
    stage_load(0);
    stage_write(0);
    float4 cur[GQ], nxt[GQ];
#pragma unroll
    for (int i = 0; i < GQ; ++i) cur[i] = W4[(size_t)i * 64];
    __syncthreads();

    int qbase = 0;
#pragma unroll 1
    for (int ch = 0; ch < nchunks; ++ch) {
        if (ch + 1 < nchunks) stage_load((ch + 1) * KC);      // in flight under this chunk's MFMAs
        const float4* pb = patch + (size_t)(ch & 1) * buf_pos * 4 + h;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
            for (int ky = 0; ky < 3; ++ky) {
#pragma unroll
                for (int i = 0; i < GQ; ++i) {
                    int n = qbase + GQ + i;
                    n = n < nq ? n : nq - 1;
                    nxt[i] = W4[(size_t)n * 64];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int kx = 0; kx < 3; ++kx) {
                    if (!active) break;    // a workgroup's spare waves only help with staging
                    const float4 a = cur[kx];
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const float4 bq = pb[(size_t)(pos0[t] + ky * PW + kx) * 4 + 2 * kb];
                        acc[t] = mfma(a.x, bq.x, acc[t]);
                        acc[t] = mfma(a.y, bq.y, acc[t]);
                        acc[t] = mfma(a.z, bq.z, acc[t]);
                        acc[t] = mfma(a.w, bq.w, acc[t]);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < GQ; ++i) cur[i] = nxt[i];
                qbase += GQ;
            }
        }
        if (ch + 1 < nchunks) stage_write((ch + 1) & 1);
        __syncthreads();
    }

    if (!active) return;
    // ---- epilogue: (ELU) -> (BN affine) -> 16-byte stores ---------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int co = co_tile * 32 + 8 * q + 4 * h;
        float sc[4], sh[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            sc[i] = (bn_scale && co + i < Cout) ? bn_scale[co + i] : 1.0f;
            sh[i] = (bn_shift && co + i < Cout) ? bn_shift[co + i] : 0.0f;
        }
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            if (opx[t] < 0) continue;
            float v[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float u = acc[t][4 * q + i];
                if (act == 1) u = u > 0.0f ? u : expm1f(u);
                v[i] = u * sc[i] + sh[i];
            }
            float* o = out + (size_t)opx[t] * Cout + co;
            if (co + 3 < Cout) {
                *(float4*)o = make_float4(v[0], v[1], v[2], v[3]);
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (co + i < Cout) o[i] = v[i];
            }
        }
    }
}

// weight repack on the device: w [Cout][Cin][3][3] (torch layout) -> wpk (see the file header)
__global__ __launch_bounds__(256) void pack_conv3x3_kernel(const float* __restrict__ w, int Cout, int Cin,
                                                           float4* __restrict__ wpk, size_t total) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int lane = i & 63;
    size_t r = i >> 6;
    const int tap = r % 9;
    r /= 9;
    const int kb = r % (Cin / 8);
    const int mt = r / (Cin / 8);
    const int co = mt * 32 + (lane & 31), ci = kb * 8 + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = (co < Cout) ? w[((size_t)co * Cin + ci + e) * 9 + tap] : 0.0f;
    wpk[i] = make_float4(v[0], v[1], v[2], v[3]);
}

template <int WM, int NT, bool ROWSEG, int NLD>
int launch_conv(const float* x, const float4* wpk, const float* bias, const float* sc, const float* sh, float* out, int B,
                int H, int W, int Cin, int Cout, int act, int Hs, int Ws, hipStream_t s) {
    constexpr int WN = 4 / WM, BPX = WN * NT * 32;
    const int n_cotiles = (Cout + 31) / 32;
    int rows, PW, nblk;
    if (ROWSEG) {
        rows = 3;
        PW = BPX + 2;
        nblk = H * ((W + BPX - 1) / BPX);
    } else {
        rows = (BPX + W - 2) / W + 1 + 2;
        if (rows > H + 2) rows = H + 2;
        PW = W + 2;
        nblk = (H * W + BPX - 1) / BPX;
    }
    const int buf_pos = rows * PW;
    if (buf_pos * 4 > NLD * 256) return OSSID_EINVAL;
    const size_t lds = (size_t)2 * buf_pos * KC * 4;
    auto kern = conv3x3_nhwc_kernel<WM, NT, ROWSEG, NLD>;
    if (lds > 48 * 1024 &&
        hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return OSSID_ELAUNCH;
    dim3 grid(nblk, (n_cotiles + WM - 1) / WM, B);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, x, wpk, bias, sc, sh, out, H, W, Cin, Cout, n_cotiles, act,
                       buf_pos, Hs, Ws, (float)Hs / (float)H, (float)Ws / (float)W);
    return ossid_launch_status();
}

}  // namespace

extern "C" {

size_t ossid_conv3x3_packed_floats(int Cout, int Cin) { return (size_t)((Cout + 31) / 32) * (Cin / 8) * 9 * 64 * 4; }

int ossid_conv3x3_pack_weights(const float* w, int Cout, int Cin, float* wpk, void* stream) {
    if (!w || !wpk || Cout <= 0 || Cin <= 0 || Cin % 16) return OSSID_EINVAL;
    const size_t total = ossid_conv3x3_packed_floats(Cout, Cin) / 4;
    hipLaunchKernelGGL(pack_conv3x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, w,
                       Cout, Cin, (float4*)wpk, total);
    return ossid_launch_status();
}

int ossid_conv3x3_nhwc_fwd(const float* x, const float* wpk, const float* bias, const float* bn_scale,
                           const float* bn_shift, float* out, int B, int H, int W, int Cin, int Cout, int act,
                           int src_h, int src_w, void* stream) {
    if (B < 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || Cin % 16 || (Cout % 4) || B > 65535) return OSSID_EINVAL;
    if (B == 0) return OSSID_OK;
    if (!x || !wpk || !out || (act != 0 && act != 1)) return OSSID_EINVAL;
    const int Hs = src_h > 0 ? src_h : H, Ws = src_w > 0 ? src_w : W;
    if (Hs > H || Ws > W) return OSSID_EINVAL;   // only up-sampling is fused
    hipStream_t s = (hipStream_t)stream;
    const float4* w4 = (const float4*)wpk;
    const bool rowseg = W > 100;
    const int tiles = (Cout + 31) / 32;
#define OSSID_CONV(WM_, NT_, NLDF_, NLDR_)                                                                           \
    (rowseg ? launch_conv<WM_, NT_, true, NLDR_>(x, w4, bias, bn_scale, bn_shift, out, B, H, W, Cin, Cout, act, Hs,  \
                                                 Ws, s)                                                              \
            : launch_conv<WM_, NT_, false, NLDF_>(x, w4, bias, bn_scale, bn_shift, out, B, H, W, Cin, Cout, act, Hs, \
                                                  Ws, s))
    // waves go to channel tiles while there are at least that many; the rest of the workgroup takes more pixels.
    // 128 pixels per workgroup unless that leaves the 256 CUs with under ~3 workgroups each; layers with one or two
    // channel tiles take 256 pixels so that each streamed weight quad still feeds two pixel tiles.
    const long px = (long)B * H * W;
    if (tiles >= 4) {
        const long blocks128 = (px + 127) / 128 * ((tiles + 3) / 4);
        return blocks128 >= 768 ? OSSID_CONV(4, 4, 8, 8) : OSSID_CONV(4, 2, 8, 8);
    }
    // (256-pixel tiles with 13 staged float4 per thread were measured slower on the decoder's few-channel layers:
    // their K is only 288-576, so the per-workgroup set-up, not the weight stream, is what counts)
    if (tiles >= 2) return OSSID_CONV(2, 2, 8, 8);
    return OSSID_CONV(1, 1, 8, 8);
#undef OSSID_CONV
}

}  // extern "C"
