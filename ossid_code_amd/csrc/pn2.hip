// PointNet++ (SSG) hypothesis scorer for gfx950: furthest-point sampling, ball query, and the three
// set-abstraction MLPs + FC head on the f32 matrix cores (v_mfma_f32_32x32x2_f32).
//
// Stands behind zephyr.models.pointnet2.PointNet2SSG.forward({"point_x": ...})
// (ctor /root/reference/python/ossid/scripts/online_learning.py:212-227, call
// utils/zephyr_utils.py:34); the algorithm is pointnet2_ops v3.0.0's, restated in SPEC.md 4 and
// oracle/zephyr_oracle.c. Results are bit-identical to the oracle: each output is ONE fmaf chain
// started from the folded bias, walking input channels in the canonical order (8-blocks
// ascending, offsets 0,4,1,5,2,6,3,7) -- which is exactly what a chain of 32x32x2 f32 MFMAs
// produces when lane half h supplies channel 8b+4h+i at step i: D = fma(a_k1,b_k1,fma(a_k0,b_k0,C)).
//
// MLP design (sa1/sa2/sa3/p2 kernels): samples sit on the MFMA N axis (lane&31), channels on M.
// A layer's 32x32 accumulator tile IS the next layer's B operand (register r of lane half h holds
// channel (r&3)+8(r>>2)+4h), so activations never leave the register file between layers: no LDS,
// no barriers, 64-wide waves each carrying 32 samples. Weights are pre-packed on the host so one
// coalesced 16-B load per lane feeds four MFMAs. The max-pool over a group's samples is a
// butterfly reduce-scatter across the 32 lanes (16 shuffles per 32 channels).
#include "common.h"

namespace {

typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mfma(float a, float b, v16f c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
}

// accumulator tile initialised with the folded bias: register 4q+e of lane half h <- b[32mt+8q+4h+e]
__device__ __forceinline__ v16f bias_tile(const float* __restrict__ b, int mt, int h) {
    v16f acc;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        float4 t = *(const float4*)(b + mt * 32 + 8 * q + 4 * h);
        acc[4 * q + 0] = t.x;
        acc[4 * q + 1] = t.y;
        acc[4 * q + 2] = t.z;
        acc[4 * q + 3] = t.w;
    }
    return acc;
}

__device__ __forceinline__ v16f relu16(v16f a) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fmaxf(a[i], 0.0f);
    return a;
}

__device__ __forceinline__ v16f max16(v16f a, v16f b) {
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = fmaxf(a[i], b[i]);
    return a;
}

// Launder the weight pointers once per loop iteration: without it LICM hoists every weight load of a fully
// unrolled layer out of the enclosing tile/centre loop -- the whole weight set -- and spills it. The laundering is an
// opaque ZERO added to the pointer (an SGPR the compiler cannot see through), so the pointer keeps its global
// address space (laundering the pointer itself degrades every load to flat_load).
__device__ __forceinline__ int opaque_zero() {
    int z = 0;
    asm volatile("" : "+s"(z));
    return z;
}
template <class T>
__device__ __forceinline__ const T* opaque(const T* p) {
    return p + opaque_zero();
}

// four chained MFMAs: one packed weight quad against registers 4q..4q+3 of an activation tile
__device__ __forceinline__ v16f mfma4(float4 a, const v16f& x, int q4, v16f acc) {
    acc = mfma(a.x, x[q4 + 0], acc);
    acc = mfma(a.y, x[q4 + 1], acc);
    acc = mfma(a.z, x[q4 + 2], acc);
    acc = mfma(a.w, x[q4 + 3], acc);
    return acc;
}

// ---- weight streaming --------------------------------------------------------------------------------
// A layer's packed weights are one linear sequence of "quads" (a float4 per lane = the A operands of four
// chained MFMAs), ordered [m-tile][k-block]. hipcc left alone hoists every load of a fully unrolled
// layer to the top and spills hundreds of registers, so the stream is cut into groups of G quads with
// an explicit two-deep register pipeline: group g+1 is loaded while group g feeds the matrix core, and a
// sched_barrier between groups keeps the compiler from re-merging them (G quads = 4G MFMAs = 256G cycles
// of matrix work cover the L2 latency of the next group's loads).

// Fully unrolled form for layers whose output tiles stay in registers (Y[t][mt] statically indexed).
template <int KT, int MT, int NT, int G, class Epi>
__device__ __forceinline__ void stream_layer(const float4* __restrict__ W4, const float* __restrict__ bias,
                                             const v16f (&X)[NT][KT], int h, Epi&& epi) {
    constexpr int QPM = KT * 4, TOTAL = MT * QPM, NG = TOTAL / G;
    static_assert(TOTAL % G == 0, "group size must divide the quad count");
    float4 cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = W4[(size_t)i * 64];
    v16f acc[NT];
#pragma unroll
    for (int g = 0; g < NG; ++g) {
        if (g + 1 < NG) {
#pragma unroll
            for (int i = 0; i < G; ++i) nxt[i] = W4[(size_t)((g + 1) * G + i) * 64];
        }
        __builtin_amdgcn_sched_barrier(0);   // the prefetch stays AHEAD of this group's MFMAs
#pragma unroll
        for (int i = 0; i < G; ++i) {
            const int quad = g * G + i, mt = quad / QPM, kq = quad % QPM, kt = kq / 4, q = kq % 4;
            if (kq == 0) {
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = bias_tile(bias, mt, h);
            }
#pragma unroll
            for (int t = 0; t < NT; ++t) acc[t] = mfma4(cur[i], X[t][kt], 4 * q, acc[t]);
            if (kq == QPM - 1) epi(mt, acc);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < G; ++i) cur[i] = nxt[i];
    }
}

// Rolled-over-m-tiles form for a group's LAST layer: each 32-channel output tile is consumed (max-pooled)
// as soon as it is complete, so only one accumulator tile per column tile is live and mt may be dynamic.
template <int KT, int NT, int G, class Epi>
__device__ __forceinline__ void stream_layer_rolled(const float4* __restrict__ W4, const float* __restrict__ bias,
                                                    const v16f (&X)[NT][KT], int h, int MT, Epi&& epi) {
    constexpr int QPM = KT * 4, NG = QPM / G;
    static_assert(QPM % G == 0, "group size must divide the quads per m-tile");
    const int last = MT * QPM - 1;
    float4 cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = W4[(size_t)i * 64];
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
        v16f acc[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = bias_tile(bias, mt, h);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                int nq = mt * QPM + (g + 1) * G + i;  // next group, possibly the next m-tile's first
                nq = nq > last ? last : nq;
                nxt[i] = W4[(size_t)nq * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int kq = g * G + i, kt = kq / 4, q = kq % 4;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma4(cur[i], X[t][kt], 4 * q, acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) cur[i] = nxt[i];
        }
        epi(mt, acc);
    }
}

// ---- a module's LAST layer, operands swapped ---------------------------------------------------------------
// D = A.B is symmetric in how lanes index the two non-k dimensions, so feeding the activation register as the A
// operand and the packed weight as the B operand yields the TRANSPOSED tile: rows (registers) = the 32 samples,
// columns (lane&31) = 32 output channels -- from the same packed weights and with the same per-element fmaf chain
// (fma(a,b,c) == fma(b,a,c)), hence bit-identical values. In this orientation the max-pool over samples is an
// in-register max over the 16 accumulator registers plus ONE exchange between the two lane halves, and each lane
// ends up owning one channel: a coalesced 128-byte store instead of a 16-shuffle butterfly per 32 channels.
__device__ __forceinline__ v16f mfma4_swapped(const v16f& x, int q4, float4 w, v16f acc) {
    acc = mfma(x[q4 + 0], w.x, acc);
    acc = mfma(x[q4 + 1], w.y, acc);
    acc = mfma(x[q4 + 2], w.z, acc);
    acc = mfma(x[q4 + 3], w.w, acc);
    return acc;
}

__device__ __forceinline__ v16f splat16(float v) {
    v16f a;
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = v;
    return a;
}

// max over the 32 samples of a swapped tile (16 registers x 2 lane halves), ReLU folded in (it commutes with max);
// every lane returns the pooled value of channel lane&31
__device__ __forceinline__ float pool_swapped(const v16f& a) {
    float m0 = fmaxf(fmaxf(a[0], a[1]), fmaxf(a[2], a[3]));
    float m1 = fmaxf(fmaxf(a[4], a[5]), fmaxf(a[6], a[7]));
    float m2 = fmaxf(fmaxf(a[8], a[9]), fmaxf(a[10], a[11]));
    float m3 = fmaxf(fmaxf(a[12], a[13]), fmaxf(a[14], a[15]));
    float m = fmaxf(fmaxf(m0, m1), fmaxf(m2, m3));
    // the other lane half's value: v_permlane32_swap (gfx950) is a plain VALU op; __shfl_xor(m, 32) goes through the LDS
    // crossbar (ds_bpermute: address VGPR, lgkmcnt wait) in the middle of an MFMA stream
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
    m = fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
    return fmaxf(m, 0.0f);
}

template <int KT, int NT, int G, class Epi>
__device__ __forceinline__ void stream_last_layer(const float4* __restrict__ W4, const float* __restrict__ bias,
                                                  const v16f (&X)[NT][KT], int c, int MT, Epi&& epi) {
    constexpr int QPM = KT * 4, NG = QPM / G;
    static_assert(QPM % G == 0, "group size must divide the quads per m-tile");
    const int last = MT * QPM - 1;
    float4 cur[G], nxt[G];
#pragma unroll
    for (int i = 0; i < G; ++i) cur[i] = W4[(size_t)i * 64];
#pragma unroll 1
    for (int mt = 0; mt < MT; ++mt) {
        v16f acc[NT];
        const float bv = bias[mt * 32 + c];
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = splat16(bv);
#pragma unroll
        for (int g = 0; g < NG; ++g) {
#pragma unroll
            for (int i = 0; i < G; ++i) {
                int nq = mt * QPM + (g + 1) * G + i;
                nq = nq > last ? last : nq;
                nxt[i] = W4[(size_t)nq * 64];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                const int kq = g * G + i, kt = kq / 4, q = kq % 4;
#pragma unroll
                for (int t = 0; t < NT; ++t) acc[t] = mfma4_swapped(X[t][kt], 4 * q, cur[i], acc[t]);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) cur[i] = nxt[i];
        }
        epi(mt, acc);
    }
}

// Max over the 32 columns (lanes of one half) of each of the 16 rows of an accumulator tile.
// Returns, in lane i = lane&31, the maximum of register r(i) = 8*b4 + 4*b3 + 2*b2 + b1 (bits of i);
// lanes i and i^1 hold the same value. 16 shuffles instead of 80.
__device__ __forceinline__ float reduce_scatter_max(const v16f& v, int lane) {
    float w[8], x[4], y[2], z;
    const bool b4 = lane & 16, b3 = lane & 8, b2 = lane & 4, b1 = lane & 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float send = b4 ? v[j] : v[j + 8];
        float keep = b4 ? v[j + 8] : v[j];
        w[j] = fmaxf(keep, __shfl_xor(send, 16));
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float send = b3 ? w[j] : w[j + 4];
        float keep = b3 ? w[j + 4] : w[j];
        x[j] = fmaxf(keep, __shfl_xor(send, 8));
    }
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        float send = b2 ? x[j] : x[j + 2];
        float keep = b2 ? x[j + 2] : x[j];
        y[j] = fmaxf(keep, __shfl_xor(send, 4));
    }
    {
        float send = b1 ? y[0] : y[1];
        float keep = b1 ? y[1] : y[0];
        z = fmaxf(keep, __shfl_xor(send, 2));
    }
    z = fmaxf(z, __shfl_xor(z, 1));
    return z;
}

// channel (within a 32-row tile) that reduce_scatter_max leaves in this lane
__device__ __forceinline__ int scatter_row(int lane) {
    int i = lane & 31, h = lane >> 5;
    int r = ((i >> 4) & 1) * 8 + ((i >> 3) & 1) * 4 + ((i >> 2) & 1) * 2 + ((i >> 1) & 1);
    return (r & 3) + 8 * (r >> 2) + 4 * h;
}

// ---- furthest point sampling: one workgroup per point set -----------------------------------------
// LDS: pts[n] = (x,y,z,|p|^2), tmp[n] running min distance. One barrier per pick: the four waves'
// (best, index) pairs go through a double-buffered 4-entry LDS slot.
__global__ __launch_bounds__(256) void fps_kernel(const float* __restrict__ xyz, int stride, int n, int npoint,
                                                  int* __restrict__ idx_out, float* __restrict__ new_xyz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* pts = (float4*)smem;
    float* tmp = (float*)(pts + n);
    __shared__ float rbest[2][4];
    __shared__ int rbesti[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = xyz + (size_t)blockIdx.x * n * stride;
    for (int k = tid; k < n; k += 256) {
        float x = base[(size_t)k * stride], y = base[(size_t)k * stride + 1], z = base[(size_t)k * stride + 2];
        pts[k] = make_float4(x, y, z, (x * x + y * y) + z * z);
        tmp[k] = 1e10f;
    }
    __syncthreads();
    int old = 0;
    int* io = idx_out + (size_t)blockIdx.x * npoint;
    float* xo = new_xyz + (size_t)blockIdx.x * npoint * 3;
    if (tid == 0) {
        io[0] = 0;
        xo[0] = pts[0].x;
        xo[1] = pts[0].y;
        xo[2] = pts[0].z;
    }
    for (int j = 1; j < npoint; ++j) {
        const float4 po = pts[old];
        float best = -1.0f;
        int besti = 0;
        for (int k = tid; k < n; k += 256) {
            float4 p = pts[k];
            if (p.w <= 1e-3f) continue;
            float dx = p.x - po.x, dy = p.y - po.y, dz = p.z - po.z;
            float d = (dx * dx + dy * dy) + dz * dz;
            float d2 = fminf(d, tmp[k]);
            tmp[k] = d2;
            if (d2 > best) {
                best = d2;
                besti = k;
            }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            float ob = __shfl_xor(best, m);
            int oi = __shfl_xor(besti, m);
            if (ob > best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
        const int buf = j & 1;
        if (lane == 0) {
            rbest[buf][wave] = best;
            rbesti[buf][wave] = besti;
        }
        __syncthreads();
        best = rbest[buf][0];
        besti = rbesti[buf][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            float ob = rbest[buf][w];
            int oi = rbesti[buf][w];
            if (ob > best || (ob == best && oi < besti)) {
                best = ob;
                besti = oi;
            }
        }
        old = besti;
        if (tid == 0) {
            io[j] = old;
            float4 p = pts[old];
            xo[3 * j] = p.x;
            xo[3 * j + 1] = p.y;
            xo[3 * j + 2] = p.z;
        }
    }
}

// ---- furthest point sampling, register-resident form -------------------------------------------------
// Thread t owns the CONTIGUOUS indices [t*P, t*P+P): coordinates and running distances live in registers, so a
// pick costs P distance updates per lane (VALU), one DPP max-scan across the wave, one ballot and one LDS hand-off
// between the four waves. Because lower lanes (and lower waves) own lower indices, "first maximum in index order"
// -- pointnet2's tie rule -- is simply the lowest lane holding the maximum: no (value, index) pair reduction.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max(float v) {
    // lanes with no source lane keep `old` = -2 (below every candidate: distances are >= 0, the sentinel is -1)
    int src = __float_as_int(v);
    int moved = __builtin_amdgcn_update_dpp(__float_as_int(-2.0f), src, CTRL, ROW_MASK, 0xf, false);
    return fmaxf(v, __int_as_float(moved));
}

// inclusive max-scan of a wave64 (gfx9 DPP: row_shr 1,2,4,8 then row_bcast 15 / 31); lane 63 ends with the maximum
__device__ __forceinline__ float wave_max_dpp(float v) {
    v = dpp_max<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max<0x118, 0xf>(v);   // row_shr:8
    v = dpp_max<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_max<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

template <int P>
__global__ __launch_bounds__(256) void fps_reg_kernel(const float* __restrict__ xyz, int stride, int n, int npoint,
                                                      int* __restrict__ idx_out, float* __restrict__ new_xyz) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* pts = (float4*)smem;   // read-only copy for the "current point" broadcast
    __shared__ float rbest[2][4];
    __shared__ int rbesti[2][4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* base = xyz + (size_t)blockIdx.x * n * stride;
    float px[P], py[P], pz[P], tmp[P];
    int anyz = 0;
#pragma unroll
    for (int j = 0; j < P; ++j) {
        const int k = tid * P + j;
        float x = 0.f, y = 0.f, z = 0.f;
        if (k < n) {
            x = base[(size_t)k * stride];
            y = base[(size_t)k * stride + 1];
            z = base[(size_t)k * stride + 2];
            pts[k] = make_float4(x, y, z, 0.f);
        }
        px[j] = x, py[j] = y, pz[j] = z;
        anyz |= (z != 0.0f);
        // a point that may never be picked (|p|^2 <= 1e-3, or padding) gets running distance -1: min(d, -1) stays -1
        // and -1 never beats the initial best of -1, which is exactly pointnet2's `continue`
        tmp[j] = ((k < n) && (((x * x + y * y) + z * z) > 1e-3f)) ? 1e10f : -1.0f;
    }
    // the scorer's point sets are planar (SPEC 3.4: channel 2 is 0): with every z == 0 the dz terms are exact zeros and
    // can be skipped without changing a single bit; any non-zero z in the set selects the general loop
    const bool planar = !__syncthreads_or(anyz);
    int old = 0;
    int* io = idx_out + (size_t)blockIdx.x * npoint;
    float* xo = new_xyz + (size_t)blockIdx.x * npoint * 3;
    if (tid == 0) {
        io[0] = 0;
        xo[0] = pts[0].x;
        xo[1] = pts[0].y;
        xo[2] = pts[0].z;
    }
    for (int jj = 1; jj < npoint; ++jj) {
        const float4 po = pts[old];
        float best = -1.0f;
        int besti = 0;
        if (planar) {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const float dx = px[j] - po.x, dy = py[j] - po.y;
                const float d2 = fminf(dx * dx + dy * dy, tmp[j]);
                tmp[j] = d2;
                if (d2 > best) {
                    best = d2;
                    besti = tid * P + j;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < P; ++j) {
                const float dx = px[j] - po.x, dy = py[j] - po.y, dz = pz[j] - po.z;
                const float d2 = fminf((dx * dx + dy * dy) + dz * dz, tmp[j]);
                tmp[j] = d2;
                if (d2 > best) {
                    best = d2;
                    besti = tid * P + j;
                }
            }
        }
        const float wmax = wave_max_dpp(best);
        const unsigned long long who = __ballot(best == wmax);
        const int src = __builtin_amdgcn_readfirstlane(__ffsll((long long)who) - 1);
        const int widx = __builtin_amdgcn_readlane(besti, src);
        const int buf = jj & 1;
        if (lane == 0) {
            rbest[buf][wave] = wmax;
            rbesti[buf][wave] = widx;
        }
        __syncthreads();
        float b = rbest[buf][0];
        int bi = rbesti[buf][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) {
            const float ob = rbest[buf][w];
            if (ob > b) {   // strict: an equal maximum in a later wave has a higher index
                b = ob;
                bi = rbesti[buf][w];
            }
        }
        old = (b > -1.0f) ? bi : 0;   // nothing selectable (all points at the origin): pointnet2 keeps index 0
        if (tid == 0) {
            io[jj] = old;
            const float4 p = pts[old];
            xo[3 * jj] = p.x;
            xo[3 * jj + 1] = p.y;
            xo[3 * jj + 2] = p.z;
        }
    }
}

// ---- ball query: one wave per centre, 64 candidate points per ballot -------------------------------
// A workgroup of 16 waves stages the point set ONCE in LDS and covers up to 512 centres (all of SA1's): with 64 centres
// per 4-wave workgroup the same 2048 points were staged eight times per hypothesis (and the strided 12-of-32-byte row
// reads cost twice their bytes), which was most of the kernel.
constexpr int BQ_CPB = 512;       // centres per workgroup
constexpr int BQ_THREADS = 1024;
__global__ __launch_bounds__(BQ_THREADS) void ball_query_kernel(const float* __restrict__ xyz, int stride, int n,
                                                                const float* __restrict__ new_xyz, int npoint, float r2,
                                                                int* __restrict__ idx) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float4* pts = (float4*)smem;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.x;
    const float* base = xyz + (size_t)b * n * stride;
    for (int k = tid; k < n; k += BQ_THREADS)
        pts[k] = make_float4(base[(size_t)k * stride], base[(size_t)k * stride + 1], base[(size_t)k * stride + 2], 0.f);
    __syncthreads();
    const int jend = min((int)(blockIdx.y + 1) * BQ_CPB, npoint);
    for (int j = blockIdx.y * BQ_CPB + wave; j < jend; j += BQ_THREADS / 64) {
        const float* c = new_xyz + ((size_t)b * npoint + j) * 3;
        const float cx = c[0], cy = c[1], cz = c[2];
        int* o = idx + ((size_t)b * npoint + j) * 64;
        int cnt = 0, first = -1;
        // four 64-point chunks per trip (most centres need ~25 chunks to collect 64 neighbours: the trip's latency
        // chain -- LDS read, compare, ballot, scalar bookkeeping -- is what bounds the kernel, so give it 4x the work);
        // slots are still handed out in index order, hits past the 64th are simply not stored
        for (int k0 = 0; k0 < n && cnt < 64; k0 += 256) {
            bool hit[4];
            unsigned long long bal[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = k0 + 64 * u + lane;
                hit[u] = false;
                if (k < n) {
                    float4 p = pts[k];
                    float dx = cx - p.x, dy = cy - p.y, dz = cz - p.z;
                    float d2 = (dx * dx + dy * dy) + dz * dz;
                    hit[u] = d2 < r2;
                }
                bal[u] = __ballot(hit[u]);
            }
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                if (bal[u]) {
                    if (first < 0) first = k0 + 64 * u + __ffsll((long long)bal[u]) - 1;
                    int slot = cnt + __popcll(bal[u] & ((1ull << lane) - 1ull));
                    if (hit[u] && slot < 64) o[slot] = k0 + 64 * u + lane;
                    cnt += __popcll(bal[u]);
                }
            }
        }
        if (first < 0) first = 0;
        if (cnt > 64) cnt = 64;
        if (lane >= cnt) o[lane] = first;
    }
}

// ---- SA1: gather (K=8) -> 64 -> 64 -> 128 -> max over the group's 64 samples -------------------------
// One wave per centre; both 32-sample column tiles ride together so every weight quad feeds 8 MFMAs.
constexpr int SA1_CPW = 8;  // centres per wave
// LDS image of the three layers (floats): W1p 512 | W2p 4096 | W3p 8192 | b1 64 | b2 64 | b3 128  = 52 224 B.
// The probe in tools/probes/mfma_probe.hip shows why: a chained f32 MFMA stream fed with A operands from L2 tops out
// at 62 % of the matrix peak at two waves per SIMD (the vector-memory path, not the matrix core, sets the pace),
// the same stream fed from LDS reaches 90-94 %.
constexpr int SA1_W1 = 0, SA1_W2 = 512, SA1_W3 = 512 + 4096, SA1_B1 = SA1_W3 + 8192, SA1_B2 = SA1_B1 + 64,
              SA1_B3 = SA1_B2 + 64, SA1_LDS_FLOATS = SA1_B3 + 128;

__device__ __forceinline__ void stage_lds(float* dst, const float* __restrict__ src, int nfloats) {
    for (int i = threadIdx.x * 4; i < nfloats; i += blockDim.x * 4) *(float4*)(dst + i) = *(const float4*)(src + i);
}

__global__ __launch_bounds__(256, 2) void sa1_kernel(const float* __restrict__ point_x, int M,
                                                     const int* __restrict__ ball, const float* __restrict__ cxyz,
                                                     int np, int total, const float* __restrict__ W1p,
                                                     const float* __restrict__ b1, const float* __restrict__ W2p,
                                                     const float* __restrict__ b2, const float* __restrict__ W3p,
                                                     const float* __restrict__ b3, float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    stage_lds(wl + SA1_W1, W1p, 512);
    stage_lds(wl + SA1_W2, W2p, 4096);
    stage_lds(wl + SA1_W3, W3p, 8192);
    stage_lds(wl + SA1_B1, b1, 64);
    stage_lds(wl + SA1_B2, b2, 64);
    stage_lds(wl + SA1_B3, b3, 128);
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    const int row = scatter_row(lane);
    // The gather of a centre's 64 samples is two dependent global loads (ball index -> feature row): it is issued one
    // centre AHEAD -- the indices under the current centre's first layers, the rows under its last layer -- so that a
    // wave never sits on ~2 memory latencies between centres.
    const int centre0 = (blockIdx.x * 4 + wave) * SA1_CPW;
    if (centre0 >= total) return;
    int i_n[2];
    float4 r_n[2];
    float cn[3];
    auto fetch_index = [&](int ce) {
        i_n[0] = ball[(size_t)ce * 64 + c];
        i_n[1] = ball[(size_t)ce * 64 + 32 + c];
        cn[0] = cxyz[(size_t)ce * 3], cn[1] = cxyz[(size_t)ce * 3 + 1], cn[2] = cxyz[(size_t)ce * 3 + 2];
    };
    auto fetch_rows = [&](int ce) {
        const size_t base = (size_t)(ce / np) * M;
        r_n[0] = *(const float4*)(point_x + (base + i_n[0]) * 8 + 4 * h);
        r_n[1] = *(const float4*)(point_x + (base + i_n[1]) * 8 + 4 * h);
    };
    fetch_index(centre0);
    fetch_rows(centre0);
#pragma unroll 1
    for (int cw = 0; cw < SA1_CPW; ++cw) {
        const int centre = centre0 + cw;
        if (centre >= total) break;
        const int next = min(centre + 1, total - 1);          // (the last centre prefetches itself: harmless)
        const float* w = wl + opaque_zero();   // per-iteration laundering (see opaque_zero)
        float4 x[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#ifdef OSSID_ABL_NOGATHER
            float4 r = make_float4(lane * 1e-3f + t, 0.1f * cw, 0.2f, 0.3f);
#else
            float4 r = r_n[t];
#endif
            if (h == 0) {
                r.x = r.x - cn[0];
                r.y = r.y - cn[1];
                r.z = r.z - cn[2];
            }
            x[t] = r;
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch_index(next);
        __builtin_amdgcn_sched_barrier(0);
        v16f Y1[2][2], Y2[2][2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt) {
            float4 a = ((const float4*)(w + SA1_W1))[mt * 64 + lane];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                v16f acc = bias_tile(w + SA1_B1, mt, h);
                acc = mfma(a.x, x[t].x, acc);
                acc = mfma(a.y, x[t].y, acc);
                acc = mfma(a.z, x[t].z, acc);
                acc = mfma(a.w, x[t].w, acc);
                Y1[t][mt] = relu16(acc);
            }
        }
        stream_layer<2, 2, 2, 4>((const float4*)(w + SA1_W2) + lane, w + SA1_B2, Y1, h, [&](int mt, v16f(&acc)[2]) {
            Y2[0][mt] = relu16(acc[0]);
            Y2[1][mt] = relu16(acc[1]);
        });
        float* out = feat + (size_t)centre * 128 + c;
        __builtin_amdgcn_sched_barrier(0);
        fetch_rows(next);
        __builtin_amdgcn_sched_barrier(0);
        stream_last_layer<2, 2, 4>((const float4*)(w + SA1_W3) + lane, w + SA1_B3, Y2, c, 4,
                                   [&](int mt, v16f(&acc)[2]) {
#ifdef OSSID_ABL_NOEPI
                                       asm volatile("" ::"v"(acc[0]), "v"(acc[1]));
                                       if (mt == 77) out[0] = acc[0][0];
#else
                                       const float r = pool_swapped(max16(acc[0], acc[1]));
                                       if (h == 0) out[mt * 32] = r;
#endif
                                   });
    }
}

// ---- P2: per-point part of SA2's first layer: p[pt][o] = chain(bias, W[:, :128] . feat1[pt]) ----------
// (the grouped-xyz columns are applied per sample in sa2_kernel, continuing the same chain)
__global__ __launch_bounds__(256, 2) void p2_kernel(const float* __restrict__ feat, int total_tiles,
                                                    const float* __restrict__ Wp, const float* __restrict__ bias,
                                                    float* __restrict__ P) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    const int tile0 = (blockIdx.x * 4 + wave) * 2;
    if (tile0 >= total_tiles) return;
    const bool two = tile0 + 1 < total_tiles;
    v16f X[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int pt = (tile0 + ((t == 1 && two) ? 1 : 0)) * 32 + c;
        const float* r = feat + (size_t)pt * 128 + 4 * h;
#pragma unroll
        for (int kt = 0; kt < 4; ++kt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float4 v = *(const float4*)(r + kt * 32 + 8 * q);
                X[t][kt][4 * q + 0] = v.x;
                X[t][kt][4 * q + 1] = v.y;
                X[t][kt][4 * q + 2] = v.z;
                X[t][kt][4 * q + 3] = v.w;
            }
    }
    float* o0 = P + (size_t)(tile0 * 32 + c) * 128 + 4 * h;
    stream_layer_rolled<4, 2, 8>((const float4*)Wp + lane, bias, X, h, 4, [&](int mt, v16f(&acc)[2]) {
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
            float* o = o0 + (size_t)t * 32 * 128 + mt * 32;
#pragma unroll
            for (int q = 0; q < 4; ++q)
                *(float4*)(o + 8 * q) =
                    make_float4(acc[t][4 * q], acc[t][4 * q + 1], acc[t][4 * q + 2], acc[t][4 * q + 3]);
        }
    });
}

// ---- SA2: (P gather + xyz columns) -> relu -> 128 -> 256 -> max over 64 samples ----------------------
// One wave per centre, its two 32-sample column tiles one after the other (a tile holds 64 + 64
// activation registers); the first tile's pooled maxima wait in the output row and are merged by the
// same lane when the second tile is done.
// Weights: the last layer's 128 KB stay resident in LDS; the middle layer's 64 KB do not fit beside them, so they go
// round a two-slot LDS RING of 8 KB chunks (8 quads = half an output tile) filled by LDS-DMA (global_load_lds_dwordx4:
// the packed quad image is lane-linear, one 1 KB wave-instruction per quad, no VGPRs): the 8 waves walk the layer in
// step, each issues one quad of chunk k+1 right after the barrier that publishes chunk k, and the 2 x 32 MFMAs per
// SIMD of a chunk cover the L2 latency of the next. (Streaming this layer through registers from L2 instead ran the
// matrix core at ~62 % for a third of the kernel's work.)
constexpr int SA2_CPW = 8;
constexpr int SA2_THREADS = 512;                       // 8 waves = the whole CU at two waves per SIMD
constexpr int SA2_CHUNK = 8 * 256;                     // floats per ring slot: 8 quads x 64 lanes x 4
constexpr int SA2_W3 = 0, SA2_B3 = 32768, SA2_B2 = SA2_B3 + 256, SA2_WX = SA2_B2 + 128, SA2_RING = SA2_WX + 384,
              SA2_KEEP = SA2_RING + 2 * SA2_CHUNK,
              SA2_LDS_FLOATS = SA2_KEEP + 8 * 256;   // W3p 128 KB | b3 | b2 | wxyz | ring 16 KB | keep 8 KB = 158 720 B


__global__ __launch_bounds__(SA2_THREADS, 2) void sa2_kernel(const float* __restrict__ P, const float* __restrict__ xyz1,
                                                             int np1, const int* __restrict__ ball,
                                                             const float* __restrict__ cxyz, int np2, int total,
                                                             const float* __restrict__ wxyz,
                                                             const float* __restrict__ W2p, const float* __restrict__ b2,
                                                             const float* __restrict__ W3p, const float* __restrict__ b3,
                                                             float* __restrict__ feat) {
    extern __shared__ __attribute__((aligned(16))) float wl[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, h = lane >> 5, c = lane & 31;
    // chunk g of the middle layer = quads 8g..8g+7; this wave moves quad 8g+wave into slot g&1
    const float4* W2q = (const float4*)W2p + (size_t)wave * 64 + lane;
    float* ring_mine = wl + SA2_RING + wave * 256;
    lds_dma_quad(W2q, ring_mine);
    stage_lds(wl + SA2_W3, W3p, 32768);
    stage_lds(wl + SA2_B3, b3, 256);
    stage_lds(wl + SA2_B2, b2, 128);
    stage_lds(wl + SA2_WX, wxyz, 384);
    float* keep = wl + SA2_KEEP + wave * 256 + c;          // the first tile's pooled maxima of this wave's centre

    // The 8 waves keep step through the ring's barriers, so a tile's gather must not sit between them: the sample
    // indices of tile it+1 are fetched under tile it's first chunk and its 16 P quads + xyz under tile it's last layer.
    constexpr int NTILE = SA2_CPW * 2;
    const int centre0 = (blockIdx.x * (SA2_THREADS / 64) + wave) * SA2_CPW;
    auto centre_of = [&](int it) { return min(centre0 + (it >> 1), total - 1); };
    float4 pvn[16];
    float nx, ny, nz, ncx, ncy, ncz;
    int inext;
    auto fetch_index = [&](int it) {
        const int ce = centre_of(it);
        inext = ball[(size_t)ce * 64 + (it & 1) * 32 + c];
        ncx = cxyz[(size_t)ce * 3], ncy = cxyz[(size_t)ce * 3 + 1], ncz = cxyz[(size_t)ce * 3 + 2];
    };
    auto fetch_rows = [&](int it) {
        const size_t pt = (size_t)(centre_of(it) / np2) * np1 + inext;
        nx = xyz1[pt * 3], ny = xyz1[pt * 3 + 1], nz = xyz1[pt * 3 + 2];
#pragma unroll
        for (int j = 0; j < 16; ++j) pvn[j] = *(const float4*)(P + pt * 128 + (j >> 2) * 32 + 8 * (j & 3) + 4 * h);
    };
    fetch_index(0);
    fetch_rows(0);
#pragma unroll 1
    for (int it = 0; it < NTILE; ++it) {
        const float* w = wl + opaque_zero();
        const int t = it & 1;
        const bool live = centre0 + (it >> 1) < total;       // a spare wave keeps step with the ring, stores nothing
        float* out = feat + (size_t)centre_of(it) * 256 + c;
        const float dx = nx - ncx, dy = ny - ncy, dz = nz - ncz;
        v16f X1[1][4], Y2[1][4];
        v16f acc;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            lds_dma_wait_all();       // this wave's piece of chunk g (issued a chunk ago) has landed ...
            __syncthreads();          // ... everyone's has, and nobody reads slot (g+1)&1 any more
            if (!(it == NTILE - 1 && g == 7))
                lds_dma_quad(W2q + (size_t)(((g + 1) & 7) * 8) * 64, ring_mine + ((g + 1) & 1) * SA2_CHUNK);
            if (g == 0) {
#pragma unroll
                for (int kt = 0; kt < 4; ++kt)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const int off = kt * 32 + 8 * q + 4 * h;
                        const float4 p = pvn[kt * 4 + q];
                        float4 wx = *(const float4*)(w + SA2_WX + off), wy = *(const float4*)(w + SA2_WX + 128 + off),
                               wz = *(const float4*)(w + SA2_WX + 256 + off);
                        X1[0][kt][4 * q + 0] = fmaxf(fmaf(wz.x, dz, fmaf(wy.x, dy, fmaf(wx.x, dx, p.x))), 0.0f);
                        X1[0][kt][4 * q + 1] = fmaxf(fmaf(wz.y, dz, fmaf(wy.y, dy, fmaf(wx.y, dx, p.y))), 0.0f);
                        X1[0][kt][4 * q + 2] = fmaxf(fmaf(wz.z, dz, fmaf(wy.z, dy, fmaf(wx.z, dx, p.z))), 0.0f);
                        X1[0][kt][4 * q + 3] = fmaxf(fmaf(wz.w, dz, fmaf(wy.w, dy, fmaf(wx.w, dx, p.w))), 0.0f);
                        if (q == 3) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later rows from piling up
                    }
                fetch_index(min(it + 1, NTILE - 1));          // retired by the next barrier's wait, under this chunk
            }
            const float4* rb = (const float4*)(w + SA2_RING + (g & 1) * SA2_CHUNK) + lane;
            float4 a[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) a[j] = rb[j * 64];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int quad = g * 8 + j, mt = quad / 16, kq = quad % 16, kt = kq / 4, q = kq % 4;
                if (kq == 0) acc = bias_tile(w + SA2_B2, mt, h);
                acc = mfma4(a[j], X1[0][kt], 4 * q, acc);
                if (kq == 15) Y2[0][mt] = relu16(acc);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        fetch_rows(min(it + 1, NTILE - 1));                    // in flight under the last layer
        __builtin_amdgcn_sched_barrier(0);
        stream_last_layer<4, 1, 4>((const float4*)(w + SA2_W3) + lane, w + SA2_B3, Y2, c, 8,
                                   [&](int mt, v16f(&acc3)[1]) {
                                       float r = pool_swapped(acc3[0]);
                                       if (h == 0) {
                                           if (t == 0) {
                                               keep[mt * 32] = r;
                                           } else if (live) {
                                               out[mt * 32] = fmaxf(r, keep[mt * 32]);
                                           }
                                       }
                                   });
    }
}

// ---- SA3 (GroupAll): (feat2, xyz2) K=264 -> 256 -> 512 -> 1024 -> max over all np2 points -----------
// One workgroup per hypothesis, one 32-point column tile per wave at a time, one wave per SIMD
// (the 256->512 layer keeps 128 + 256 accumulator registers live).
__global__ __launch_bounds__(256, 1) void sa3_kernel(const float* __restrict__ feat2, const float* __restrict__ xyz2,
                                                     int np2, const float* __restrict__ W1p,
                                                     const float* __restrict__ b1, const float* __restrict__ W2p,
                                                     const float* __restrict__ b2, const float* __restrict__ W3p,
                                                     const float* __restrict__ b3, float* __restrict__ feat3) {
    __shared__ float red[4][1024];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5, c = lane & 31;
    const int row = scatter_row(lane);
    const int n = blockIdx.x;
    for (int i = tid; i < 4096; i += 256) (&red[0][0])[i] = -INFINITY;
    __syncthreads();
    const int ntiles = np2 / 32;
#pragma unroll 1
    for (int t = wave; t < ntiles; t += 4) {
        W1p = opaque(W1p), W2p = opaque(W2p), W3p = opaque(W3p);
        b1 = opaque(b1), b2 = opaque(b2), b3 = opaque(b3);
        const size_t pt = (size_t)n * np2 + t * 32 + c;
        const float* r = feat2 + pt * 256 + 4 * h;
        v16f Y1[1][8], Y2[1][16];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) Y1[0][mt] = bias_tile(b1, mt, h);
        // first layer, k-outer: one activation quad (this lane's 4 channels of block kb) against the
        // eight output tiles; weights [8][33][64][4]. Two-deep pipeline over kb like stream_layer.
        const float4* W4 = (const float4*)W1p + lane;
        float4 bc = *(const float4*)(r), bn;
        float4 ac[8], an[8];
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) ac[mt] = W4[(size_t)(mt * 33) * 64];
#pragma unroll 1
        for (int kb = 0; kb < 33; ++kb) {
            const int kn = kb + 1 < 33 ? kb + 1 : 32;
            if (kn < 32) {
                bn = *(const float4*)(r + 8 * kn);
            } else {  // the (x, y, z, 0 | 0, 0, 0, 0) block
                bn = make_float4(0.f, 0.f, 0.f, 0.f);
                if (h == 0) {
                    bn.x = xyz2[pt * 3];
                    bn.y = xyz2[pt * 3 + 1];
                    bn.z = xyz2[pt * 3 + 2];
                }
            }
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) an[mt] = W4[(size_t)(mt * 33 + kn) * 64];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) {
                Y1[0][mt] = mfma(ac[mt].x, bc.x, Y1[0][mt]);
                Y1[0][mt] = mfma(ac[mt].y, bc.y, Y1[0][mt]);
                Y1[0][mt] = mfma(ac[mt].z, bc.z, Y1[0][mt]);
                Y1[0][mt] = mfma(ac[mt].w, bc.w, Y1[0][mt]);
            }
            __builtin_amdgcn_sched_barrier(0);
            bc = bn;
#pragma unroll
            for (int mt = 0; mt < 8; ++mt) ac[mt] = an[mt];
        }
#pragma unroll
        for (int mt = 0; mt < 8; ++mt) Y1[0][mt] = relu16(Y1[0][mt]);
        stream_layer<8, 16, 1, 4>((const float4*)W2p + lane, b2, Y1, h,
                                  [&](int mt, v16f(&acc)[1]) { Y2[0][mt] = relu16(acc[0]); });
        stream_last_layer<16, 1, 8>((const float4*)W3p + lane, b3, Y2, c, 32, [&](int mt, v16f(&acc)[1]) {
            const float v = pool_swapped(acc[0]);
            if (h == 0) {
                float* sl = &red[wave][mt * 32 + c];
                *sl = fmaxf(*sl, v);
            }
        });
    }
    __syncthreads();
    for (int o = tid; o < 1024; o += 256)
        feat3[(size_t)n * 1024 + o] = fmaxf(fmaxf(red[0][o], red[1][o]), fmaxf(red[2][o], red[3][o]));
}

// ---- FC head 1024 -> 512 -> 256 -> 1, eight hypotheses per workgroup ---------------------------------
// Weights are stored transposed [k][cout] so a wave's loads are contiguous; activations sit in LDS.
constexpr int FC_HB = 8;
__device__ __forceinline__ int canon(int e) { return (e >> 1) + 4 * (e & 1); }  // 0,4,1,5,2,6,3,7

__global__ __launch_bounds__(256) void fc_head_kernel(const float* __restrict__ feat3, int B,
                                                      const float* __restrict__ Wt1, const float* __restrict__ bb1,
                                                      const float* __restrict__ Wt2, const float* __restrict__ bb2,
                                                      const float* __restrict__ W3, const float* __restrict__ bb3,
                                                      float* __restrict__ scores) {
    __shared__ float x[FC_HB][1024];
    __shared__ float h1[FC_HB][512];
    __shared__ float h2[FC_HB][256];
    const int tid = threadIdx.x;
    const int b0 = blockIdx.x * FC_HB;
    for (int i = tid; i < FC_HB * 1024; i += 256) {
        int hb = i >> 10, k = i & 1023;
        x[hb][k] = (b0 + hb < B) ? feat3[(size_t)(b0 + hb) * 1024 + k] : 0.0f;
    }
    __syncthreads();
    {
        float acc[2][FC_HB];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hb = 0; hb < FC_HB; ++hb) acc[j][hb] = bb1[tid + 256 * j];
        for (int kb = 0; kb < 1024; kb += 8)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb + canon(e);
                const float w0 = Wt1[(size_t)k * 512 + tid], w1 = Wt1[(size_t)k * 512 + tid + 256];
#pragma unroll
                for (int hb = 0; hb < FC_HB; ++hb) {
                    const float xv = x[hb][k];
                    acc[0][hb] = fmaf(w0, xv, acc[0][hb]);
                    acc[1][hb] = fmaf(w1, xv, acc[1][hb]);
                }
            }
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int hb = 0; hb < FC_HB; ++hb) h1[hb][tid + 256 * j] = fmaxf(acc[j][hb], 0.0f);
    }
    __syncthreads();
    {
        float acc[FC_HB];
#pragma unroll
        for (int hb = 0; hb < FC_HB; ++hb) acc[hb] = bb2[tid];
        for (int kb = 0; kb < 512; kb += 8)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb + canon(e);
                const float w = Wt2[(size_t)k * 256 + tid];
#pragma unroll
                for (int hb = 0; hb < FC_HB; ++hb) acc[hb] = fmaf(w, h1[hb][k], acc[hb]);
            }
#pragma unroll
        for (int hb = 0; hb < FC_HB; ++hb) h2[hb][tid] = fmaxf(acc[hb], 0.0f);
    }
    __syncthreads();
    if (tid < FC_HB && b0 + tid < B) {
        float acc = bb3[0];
        for (int kb = 0; kb < 256; kb += 8)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int k = kb + canon(e);
                acc = fmaf(W3[k], h2[tid][k], acc);
            }
        scores[b0 + tid] = acc;
    }
}

inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

struct Workspace {
    int* fps1;
    float* xyz1;
    int* ball1;
    float* feat1;
    float* p2;
    int* fps2;
    float* xyz2;
    int* ball2;
    float* feat2;
    float* feat3;
    size_t bytes;
};

Workspace carve(char* base, int B, int M, int np1, int np2) {
    (void)M;
    Workspace w;
    size_t off = 0;
    auto take = [&](size_t n) {
        char* p = base ? base + off : nullptr;
        off += align256(n);
        return p;
    };
    w.fps1 = (int*)take((size_t)B * np1 * 4);
    w.xyz1 = (float*)take((size_t)B * np1 * 12);
    w.ball1 = (int*)take((size_t)B * np1 * 64 * 4);
    w.feat1 = (float*)take((size_t)B * np1 * 128 * 4);
    w.p2 = (float*)take((size_t)B * np1 * 128 * 4);
    w.fps2 = (int*)take((size_t)B * np2 * 4);
    w.xyz2 = (float*)take((size_t)B * np2 * 12);
    w.ball2 = (int*)take((size_t)B * np2 * 64 * 4);
    w.feat2 = (float*)take((size_t)B * np2 * 256 * 4);
    w.feat3 = (float*)take((size_t)B * 1024 * 4);
    w.bytes = off;
    return w;
}

template <int P>
int launch_fps_reg(const float* xyz, int stride, int B, int n, int npoint, int* idx, float* new_xyz, hipStream_t s) {
    hipLaunchKernelGGL(fps_reg_kernel<P>, dim3(B), dim3(256), (size_t)n * 16, s, xyz, stride, n, npoint, idx, new_xyz);
    return ossid_launch_status();
}

int launch_fps(const float* xyz, int stride, int B, int n, int npoint, int* idx, float* new_xyz, hipStream_t s) {
    if (n <= 256 * 2) return launch_fps_reg<2>(xyz, stride, B, n, npoint, idx, new_xyz, s);
    if (n <= 256 * 4) return launch_fps_reg<4>(xyz, stride, B, n, npoint, idx, new_xyz, s);
    if (n <= 256 * 8) return launch_fps_reg<8>(xyz, stride, B, n, npoint, idx, new_xyz, s);
    if (n <= 256 * 12) return launch_fps_reg<12>(xyz, stride, B, n, npoint, idx, new_xyz, s);
    // larger sets: points and running distances in LDS
    size_t lds = (size_t)n * 20;
    if (lds > 160 * 1024 - 1024) return OSSID_EINVAL;
    OSSID_ENSURE_LDS(fps_kernel, lds);
    hipLaunchKernelGGL(fps_kernel, dim3(B), dim3(256), lds, s, xyz, stride, n, npoint, idx, new_xyz);
    return ossid_launch_status();
}

int launch_ball(const float* xyz, int stride, int B, int n, const float* new_xyz, int npoint, float radius, int* idx,
                hipStream_t s) {
    size_t lds = (size_t)n * 16;
    if (lds > 160 * 1024 - 1024) return OSSID_EINVAL;
    OSSID_ENSURE_LDS(ball_query_kernel, lds);
    const float r2 = radius * radius;
    hipLaunchKernelGGL(ball_query_kernel, dim3(B, (npoint + BQ_CPB - 1) / BQ_CPB), dim3(BQ_THREADS), lds, s, xyz, stride,
                       n, new_xyz, npoint, r2, idx);
    return ossid_launch_status();
}

}  // namespace

extern "C" {

int ossid_pn2_fps(const float* xyz, int stride, int B, int n, int npoint, int32_t* idx, float* new_xyz,
                  void* stream) {
    if (B < 0 || n <= 0 || npoint <= 0 || stride < 3) return OSSID_EINVAL;
    if (B == 0) return OSSID_OK;
    if (!xyz || !idx || !new_xyz) return OSSID_EINVAL;
    return launch_fps(xyz, stride, B, n, npoint, idx, new_xyz, (hipStream_t)stream);
}

int ossid_pn2_ball_query(const float* xyz, int stride, int B, int n, const float* new_xyz, int npoint, float radius,
                         int nsample, int32_t* idx, void* stream) {
    if (B < 0 || n <= 0 || npoint <= 0 || stride < 3 || nsample != 64) return OSSID_EINVAL;
    if (B == 0) return OSSID_OK;
    if (!xyz || !idx || !new_xyz) return OSSID_EINVAL;
    return launch_ball(xyz, stride, B, n, new_xyz, npoint, radius, idx, (hipStream_t)stream);
}

size_t ossid_pn2_workspace_bytes(int B, int M, int npoint1, int npoint2) {
    if (B <= 0) return 256;
    return carve(nullptr, B, M, npoint1, npoint2).bytes;
}

const char* ossid_pn2_stage_names(void) { return "fps1,ball1,sa1,p2,fps2,ball2,sa2,sa3,fc"; }

int ossid_event_create(void** event_out_host) {
    if (!event_out_host) return OSSID_EINVAL;
    hipEvent_t e;
    if (hipEventCreate(&e) != hipSuccess) return OSSID_ELAUNCH;
    *event_out_host = (void*)e;
    return OSSID_OK;
}
int ossid_event_destroy(void* event) { return hipEventDestroy((hipEvent_t)event) == hipSuccess ? OSSID_OK : OSSID_ELAUNCH; }
int ossid_event_record(void* event, void* stream) {
    return hipEventRecord((hipEvent_t)event, (hipStream_t)stream) == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
}
int ossid_event_elapsed_ms(void* start, void* stop, float* ms_out_host) {
    if (!ms_out_host) return OSSID_EINVAL;
    if (hipEventSynchronize((hipEvent_t)stop) != hipSuccess) return OSSID_ELAUNCH;
    return hipEventElapsedTime(ms_out_host, (hipEvent_t)start, (hipEvent_t)stop) == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
}

const char* ossid_pn2_kernel_names(void) {
    return "fps_reg_kernel,fps_kernel,ball_query_kernel,sa1_kernel,p2_kernel,sa2_kernel,sa3_kernel,fc_head_kernel";
}

// The nine stages of ossid_pn2_score, one launcher each.
}  // extern "C"
namespace {
struct Pn2Call {
    const float* point_x;
    int B, M, np1, np2;
    const ossid_pn2_weights* w;
    Workspace ws;
    hipStream_t s;
};

int pn2_check(const float* point_x, int B, int M, const ossid_pn2_weights* w, void* workspace, size_t workspace_bytes,
              void* stream, Pn2Call* c) {
    if (B < 0 || !w) return OSSID_EINVAL;
    if (B == 0) return 1;      // nothing to do
    const int np1 = w->npoint1, np2 = w->npoint2;
    if (!point_x || !workspace || !w->blob) return OSSID_EINVAL;
    if (np1 <= 0 || np2 <= 0 || np1 % 32 || np2 % 32 || M < np1 || np1 < np2) return OSSID_EINVAL;
    if (((uintptr_t)workspace & 255) || ((uintptr_t)point_x & 15)) return OSSID_EINVAL;
    c->ws = carve((char*)workspace, B, M, np1, np2);
    if (c->ws.bytes > workspace_bytes) return OSSID_EINVAL;
    c->point_x = point_x; c->B = B; c->M = M; c->np1 = np1; c->np2 = np2; c->w = w; c->s = (hipStream_t)stream;
    return OSSID_OK;
}

int pn2_fps1(const Pn2Call& c) { return launch_fps(c.point_x, 8, c.B, c.M, c.np1, c.ws.fps1, c.ws.xyz1, c.s); }
int pn2_ball1(const Pn2Call& c) { return launch_ball(c.point_x, 8, c.B, c.M, c.ws.xyz1, c.np1, c.w->radius1, c.ws.ball1, c.s); }
int pn2_fps2(const Pn2Call& c) { return launch_fps(c.ws.xyz1, 3, c.B, c.np1, c.np2, c.ws.fps2, c.ws.xyz2, c.s); }
int pn2_ball2(const Pn2Call& c) { return launch_ball(c.ws.xyz1, 3, c.B, c.np1, c.ws.xyz2, c.np2, c.w->radius2, c.ws.ball2, c.s); }

int pn2_sa1(const Pn2Call& c) {
    const float* blob = c.w->blob;
    const int total = c.B * c.np1;
    const int grid = (total + 4 * SA1_CPW - 1) / (4 * SA1_CPW);
    OSSID_ENSURE_LDS(sa1_kernel, (size_t)SA1_LDS_FLOATS * 4);
    hipLaunchKernelGGL(sa1_kernel, dim3(grid), dim3(256), SA1_LDS_FLOATS * 4, c.s, c.point_x, c.M, c.ws.ball1, c.ws.xyz1, c.np1, total,
                       blob + c.w->w_off[0], blob + c.w->b_off[0], blob + c.w->w_off[1], blob + c.w->b_off[1],
                       blob + c.w->w_off[2], blob + c.w->b_off[2], c.ws.feat1);
    return ossid_launch_status();
}
int pn2_p2(const Pn2Call& c) {
    const float* blob = c.w->blob;
    const int tiles = c.B * c.np1 / 32;
    hipLaunchKernelGGL(p2_kernel, dim3((tiles + 7) / 8), dim3(256), 0, c.s, c.ws.feat1, tiles, blob + c.w->w_off[3],
                       blob + c.w->b_off[3], c.ws.p2);
    return ossid_launch_status();
}
int pn2_sa2(const Pn2Call& c) {
    const float* blob = c.w->blob;
    const int total = c.B * c.np2;
    const int per_wg = (SA2_THREADS / 64) * SA2_CPW;
    const int grid = (total + per_wg - 1) / per_wg;
    OSSID_ENSURE_LDS(sa2_kernel, (size_t)SA2_LDS_FLOATS * 4);
    hipLaunchKernelGGL(sa2_kernel, dim3(grid), dim3(SA2_THREADS), SA2_LDS_FLOATS * 4, c.s, c.ws.p2, c.ws.xyz1, c.np1, c.ws.ball2,
                       c.ws.xyz2, c.np2, total, blob + c.w->wxyz2_off, blob + c.w->w_off[4], blob + c.w->b_off[4],
                       blob + c.w->w_off[5], blob + c.w->b_off[5], c.ws.feat2);
    return ossid_launch_status();
}
int pn2_sa3(const Pn2Call& c) {
    const float* blob = c.w->blob;
    hipLaunchKernelGGL(sa3_kernel, dim3(c.B), dim3(256), 0, c.s, c.ws.feat2, c.ws.xyz2, c.np2, blob + c.w->w_off[6],
                       blob + c.w->b_off[6], blob + c.w->w_off[7], blob + c.w->b_off[7], blob + c.w->w_off[8],
                       blob + c.w->b_off[8], c.ws.feat3);
    return ossid_launch_status();
}
int pn2_fc(const Pn2Call& c, float* scores) {
    const float* blob = c.w->blob;
    hipLaunchKernelGGL(fc_head_kernel, dim3((c.B + FC_HB - 1) / FC_HB), dim3(256), 0, c.s, c.ws.feat3, c.B, blob + c.w->w_off[9],
                       blob + c.w->b_off[9], blob + c.w->w_off[10], blob + c.w->b_off[10], blob + c.w->w_off[11],
                       blob + c.w->b_off[11], scores);
    return ossid_launch_status();
}
}  // namespace

extern "C" {

int ossid_pn2_score(const float* point_x, int B, int M, const ossid_pn2_weights* w, void* workspace,
                    size_t workspace_bytes, float* scores, int32_t* dbg_fps1, int32_t* dbg_ball1, float* dbg_feat1,
                    int32_t* dbg_fps2, int32_t* dbg_ball2, float* dbg_feat2, float* dbg_feat3,
                    void* const* stage_events_host, void* stream) {
    Pn2Call c;
    int rc = pn2_check(point_x, B, M, w, workspace, workspace_bytes, stream, &c);
    if (rc) return rc < 0 ? rc : OSSID_OK;
    if (!scores) return OSSID_EINVAL;
    const Workspace& ws = c.ws;
    const int np1 = c.np1, np2 = c.np2;
    hipStream_t s = c.s;
    int stage = 0;
    auto mark = [&]() {
        if (stage_events_host && hipEventRecord((hipEvent_t)stage_events_host[stage], s) != hipSuccess) rc = OSSID_ELAUNCH;
        ++stage;
    };
    rc = OSSID_OK;
    mark();
    if ((rc = pn2_fps1(c))) return rc;
    mark();
    if ((rc = pn2_ball1(c))) return rc;
    mark();
    if ((rc = pn2_sa1(c))) return rc;
    mark();
    if ((rc = pn2_p2(c))) return rc;
    mark();
    if ((rc = pn2_fps2(c))) return rc;
    mark();
    if ((rc = pn2_ball2(c))) return rc;
    mark();
    if ((rc = pn2_sa2(c))) return rc;
    mark();
    if ((rc = pn2_sa3(c))) return rc;
    mark();
    if ((rc = pn2_fc(c, scores))) return rc;
    mark();

    auto cp = [&](void* dst, const void* src, size_t n) {
        if (dst && hipMemcpyAsync(dst, src, n, hipMemcpyDeviceToDevice, s) != hipSuccess) rc = OSSID_ELAUNCH;
    };
    cp(dbg_fps1, ws.fps1, (size_t)B * np1 * 4);
    cp(dbg_ball1, ws.ball1, (size_t)B * np1 * 64 * 4);
    cp(dbg_feat1, ws.feat1, (size_t)B * np1 * 128 * 4);
    cp(dbg_fps2, ws.fps2, (size_t)B * np2 * 4);
    cp(dbg_ball2, ws.ball2, (size_t)B * np2 * 64 * 4);
    cp(dbg_feat2, ws.feat2, (size_t)B * np2 * 256 * 4);
    cp(dbg_feat3, ws.feat3, (size_t)B * 1024 * 4);
    return rc;
}

}  // extern "C"
