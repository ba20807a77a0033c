// Shared device/host helpers for libossid_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ossid_hip.h"


static inline int ossid_launch_status() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? OSSID_OK : OSSID_ELAUNCH;
}

// ELU(alpha = 1) without libm's expm1f (~40 vector instructions per value -- 160 M of them per test-time frame in the head's
// convolution epilogues: ~0.13 ms of vector-ALU time): a degree-7 Taylor polynomial near zero, where exp(x) - 1 would cancel,
// and the hardware exponential elsewhere; relative error < 1e-6 (tests/test_dtoid_gpu.py holds it against torch's ELU).
#ifdef __HIPCC__
__device__ __forceinline__ float elu_fast(float x) {
#ifdef OSSID_ELU_LIBM            // (A/B and reference build: libm's expm1f)
    return x > 0.0f ? x : expm1f(x);
#endif
    const float xm = fminf(x, 0.0f);
    float p = 1.0f / 5040.0f;
    p = fmaf(p, xm, 1.0f / 720.0f);
    p = fmaf(p, xm, 1.0f / 120.0f);
    p = fmaf(p, xm, 1.0f / 24.0f);
    p = fmaf(p, xm, 1.0f / 6.0f);
    p = fmaf(p, xm, 0.5f);
    p = fmaf(p, xm, 1.0f);
    const float near0 = p * xm, far = __expf(xm) - 1.0f;
    const float neg = xm > -0.35f ? near0 : far;
    return x > 0.0f ? x : neg;
}
#endif

// csrc/wgrad_fc.hip: the decoder's few-channel 3x3 weight gradients from 2-D pixel tiles (internal: reached through
// ossid_conv_wgrad / ossid_conv_wgrad_workspace_bytes of csrc/train.hip)
bool ossid_wgrad_fewch_takes(int Cin, int Cout, int taps, int in_cs, int dy_cs);
size_t ossid_wgrad_fewch_workspace_bytes(int B, int H, int W, int Cin, int Cout);
int ossid_wgrad_fewch(const ossid_wgrad_desc* d, void* stream);

// csrc/wgrad_t9.hip: plain 3x3 weight gradients with input channels in 128s (the dense blocks' 128 -> 32, the head's layers) from
// 2-D pixel tiles, all taps from one staged patch (internal: reached through ossid_conv_wgrad(_group) and their workspace queries)
bool ossid_wgrad_t9_takes(const ossid_wgrad_desc* d);
int ossid_wgrad_t9_class(const ossid_wgrad_desc* d);      // problems of one ossid_wgrad_t9_group call share it (and the geometry)
size_t ossid_wgrad_t9_workspace_bytes(const ossid_wgrad_desc* descs, int n);
int ossid_wgrad_t9_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream);
// ... and the dense layers' 1x1 convolution (c -> 128) in blocks of 256 input channels ("jobs": at most
// ossid_wgrad_t1_max_jobs() per call, ossid_wgrad_t1_job_count says how many a list makes)
bool ossid_wgrad_t1_takes(const ossid_wgrad_desc* d);
int ossid_wgrad_t1_max_jobs(void);
int ossid_wgrad_t1_job_count(const ossid_wgrad_desc* descs, int n);
size_t ossid_wgrad_t1_workspace_bytes(const ossid_wgrad_desc* descs, int n);
int ossid_wgrad_t1_group(const ossid_wgrad_desc* descs, int n, void* workspace, size_t workspace_bytes, void* stream);

// Kernels that declare more dynamic LDS than the default limit need hipFuncAttributeMaxDynamicSharedMemorySize. It is a
// property of the FUNCTION, not of a launch: raise it to the hardware maximum ONCE per (function, device), the first
// time the function is launched (always a warm-up pass, never inside a stream capture), instead of before every launch.
// Round 1 set it per launch -- also from inside torch.cuda.graph captures, where a rocprofv3-traced run segfaulted in the
// launch path ("capture under the profiler").
struct OssidLdsAttr {
    int done[16];
};
static inline int ossid_ensure_dyn_lds(const void* fn, size_t bytes, OssidLdsAttr& st) {
    if (bytes <= 48 * 1024) return OSSID_OK;
    if (bytes > 160 * 1024) return OSSID_EINVAL;
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return OSSID_ELAUNCH;
    dev &= 15;
    if (__atomic_load_n(&st.done[dev], __ATOMIC_ACQUIRE)) return OSSID_OK;
    if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) return OSSID_ELAUNCH;
    __atomic_store_n(&st.done[dev], 1, __ATOMIC_RELEASE);
    return OSSID_OK;
}
#define OSSID_ENSURE_LDS(kern, bytes)                                             \
    do {                                                                          \
        static OssidLdsAttr ossid_lds_attr_;                                      \
        const int rc_ = ossid_ensure_dyn_lds((const void*)(kern), (bytes), ossid_lds_attr_); \
        if (rc_ != OSSID_OK) return rc_;                                          \
    } while (0)

// Winograd F(2x2, 3x3) filter transform U = G g G^T (G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]) in the packed layout of
// csrc/wino.hip. w is the forward weight [Cout][Cin][3][3]; dgrad != 0 packs the data gradient's layer (M = Cin output
// channels, K = Cout reduction channels, filter rotated by 180 degrees). Element i is one 16-byte unit.
//   split-bf16 form (default):  [ceil(M/32)][K/16][16 xi][2 parts][64 lanes][8 bf16] -- lane (c,h) of (mt, chunk, xi, part)
//       holds U_xi[32mt+c][16chunk+8h+0..7] as bf16: part 0 = hi = bf16(U), part 1 = lo = bf16(U - hi); the kernel forms
//       U*V ~ hi*vh + hi*vl + lo*vh on v_mfma_f32_32x32x16_bf16 (the dropped lo*vl term is ~2^-16 of the product)
//   exact-f32 form (-DOSSID_WINO_F32): [ceil(M/32)][K/8][16 xi][64 lanes][4 floats] -- lane (c,h) of (mt, kb, xi) holds
//       U_xi[32mt+c][8kb+4h+0..3]
// Both have the same size (ossid_conv_wino_packed_floats).
__device__ __forceinline__ float ossid_wino_u(const float* __restrict__ w, int Cout, int Cin, int dgrad, int m, int k, int xi) {
    const int ti = xi >> 2, tj = xi & 3;
    const float* g = dgrad ? w + ((size_t)k * Cin + m) * 9 : w + ((size_t)m * Cin + k) * 9;
    float t[3];
#pragma unroll
    for (int b = 0; b < 3; ++b) {
        const float g0 = dgrad ? g[8 - b] : g[b], g1 = dgrad ? g[5 - b] : g[3 + b], g2 = dgrad ? g[2 - b] : g[6 + b];
        t[b] = ti == 0 ? g0 : (ti == 1 ? 0.5f * (g0 + g1 + g2) : (ti == 2 ? 0.5f * (g0 - g1 + g2) : g2));
    }
    return tj == 0 ? t[0] : (tj == 1 ? 0.5f * (t[0] + t[1] + t[2]) : (tj == 2 ? 0.5f * (t[0] - t[1] + t[2]) : t[2]));
}

__device__ __forceinline__ float4 ossid_wino_pack_quad(const float* __restrict__ w, int Cout, int Cin, int dgrad, size_t i) {
    const int lane = (int)(i & 63);
    size_t r = i >> 6;
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;
#ifndef OSSID_WINO_F32
    const int part = (int)(r & 1);
    r >>= 1;
    const int xi = (int)(r & 15);
    r >>= 4;
    const int KC = K / 16;
    const int ch = (int)(r % KC), mt = (int)(r / KC);
    const int m = mt * 32 + (lane & 31), k0 = ch * 16 + 8 * (lane >> 5);
    union {
        __bf16 hv[8];
        float4 f;
    } o;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float u = m < M ? ossid_wino_u(w, Cout, Cin, dgrad, m, k0 + e, xi) : 0.0f;
        const __bf16 hi = (__bf16)u;
        o.hv[e] = part == 0 ? hi : (__bf16)(u - (float)hi);
    }
    return o.f;
#else
    const int xi = (int)(r & 15);
    r >>= 4;
    const int KB = K / 8;
    const int kb = (int)(r % KB), mt = (int)(r / KB);
    const int m = mt * 32 + (lane & 31), k0 = kb * 8 + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = m < M ? ossid_wino_u(w, Cout, Cin, dgrad, m, k0 + e, xi) : 0.0f;
    return make_float4(v[0], v[1], v[2], v[3]);
#endif
}

// Direct-convolution weights (csrc/conv.hip): element i = one 16-byte unit of the packed layout of a layer with M output and
// K reduction channels. Forward (dgrad == 0): M = Cout, K = Cin, value w[m][k][tap]; data gradient: M = Cin, K = Cout,
// value w[k][m][taps-1-tap] (transposed, rotated by 180 degrees). w is [Cout][Cin][taps].
//   split-bf16 form (default):  [ceil(M/32)][K/16][taps][2 parts][64 lanes][8 bf16] -- lane (c,h) of (mt, u, tap, part) holds
//       W[32mt+c][16u+8h+0..7][tap]: part 0 = hi = bf16(W), part 1 = lo = bf16(W - hi)
//   exact-f32 form (exact == 1; every layer of a -DOSSID_CONV_F32 build): [ceil(M/32)][K/8][taps][64 lanes][4 floats] --
//       lane (c,h) holds W[32mt+c][8kb+4h+0..3][tap]
//   three-way split (exact == 2): as the split form with [3 parts] -- p0 = bf16(W), p1 = bf16(W - p0), p2 = bf16(W - p0 - p1)
// The first two have the same size (ossid_conv_packed_floats), the third 1.5 x that (ossid_conv_packed_floats_form).
#ifdef OSSID_CONV_F32
#define OSSID_CONV_SB 0
#else
#define OSSID_CONV_SB 1
#endif
__device__ __forceinline__ float4 ossid_conv_pack_quad(const float* __restrict__ w, int Cout, int Cin, int taps, int dgrad, int exact,
                                                       size_t i) {
    const int lane = (int)(i & 63);
    size_t r = i >> 6;
    const int K = dgrad ? Cout : Cin, M = dgrad ? Cin : Cout;
    auto at = [&](int m, int k, int tap) {
        return dgrad ? w[((size_t)k * Cin + m) * taps + (taps - 1 - tap)] : w[((size_t)m * Cin + k) * taps + tap];
    };
    if (OSSID_CONV_SB && exact != 1) {
        const int parts = exact == 2 ? 3 : 2;
        const int part = (int)(r % parts);
        r /= parts;
        const int tap = (int)(r % taps);
        r /= taps;
        const int KU = K / 16;
        const int u = (int)(r % KU), mt = (int)(r / KU);
        const int m = mt * 32 + (lane & 31), k0 = u * 16 + 8 * (lane >> 5);
        union {
            __bf16 hv[8];
            float4 f;
        } o;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float v = m < M ? at(m, k0 + e, tap) : 0.0f;
            __bf16 pc = (__bf16)v;
            for (int k = 0; k < part; ++k) {
                v -= (float)pc;
                pc = (__bf16)v;
            }
            o.hv[e] = pc;
        }
        return o.f;
    }
    const int tap = (int)(r % taps);
    r /= taps;
    const int KB = K / 8;
    const int kb = (int)(r % KB), mt = (int)(r / KB);
    const int m = mt * 32 + (lane & 31), k0 = kb * 8 + 4 * (lane >> 5);
    float v[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) v[e] = m < M ? at(m, k0 + e, tap) : 0.0f;
    return make_float4(v[0], v[1], v[2], v[3]);
}

// 64-lane wave reductions (xor butterfly; every lane ends with the result)
__device__ __forceinline__ int wave_sum_i32(int v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m);
    return v;
}
__device__ __forceinline__ float wave_max_f32(float v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v = fmaxf(v, __shfl_xor(v, m));
    return v;
}

// LDS-DMA of one 1 KB quad image, issued from INLINE ASM: through the builtin hipcc knows an LDS write is in flight and
// puts s_waitcnt vmcnt(0) in front of the next ds_read of ANY LDS address -- draining the chunk that was just issued
// and exposing its whole L2 latency every step. The asm form is invisible to that bookkeeping (and to its vmcnt counts,
// which stay conservative); completion is waited for by hand (lds_dma_wait_all) before the barrier that publishes it.
__device__ __forceinline__ void lds_dma_quad(const void* g, float* lds_wave_base) {
    unsigned dst = (unsigned)(size_t)(__attribute__((address_space(3))) void*)lds_wave_base;
    dst = __builtin_amdgcn_readfirstlane(dst);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(g), "s"(dst)
                 : "memory");
}
__device__ __forceinline__ void lds_dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
