// Micro-probe: what limits a chained v_mfma_f32_32x32x2_f32 stream on gfx950?
//   mode 0: one dependent chain per wave, operands in registers
//   mode 1: two independent chains per wave
//   mode 2: one chain, A operand = float4 loaded from global every 4 MFMAs (pipeline as pn2.hip stream_layer)
//   mode 3: like 2 with two column tiles sharing the A quad (NT=2)
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run: ./mfma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
__device__ __forceinline__ v16f mfma(float a, float b, v16f c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(const float4* __restrict__ W, float* out, int iters, int nquads) {
    const int lane = threadIdx.x & 63;
    v16f acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float b0 = (float)lane * 1e-3f, b1 = b0 + 1.f;
    const float4* W4 = W + lane;
    if (MODE <= 1) {
        float a = 1e-3f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                acc0 = mfma(a, b0, acc0);
                if (MODE == 1) acc1 = mfma(a, b1, acc1); else acc0 = mfma(a, b1, acc0);
            }
        }
    } else {
        constexpr int G = 8;
        float4 cur[G], nxt[G];
        for (int i = 0; i < G; ++i) cur[i] = W4[(size_t)i * 64];
        int pos = G;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < G; ++i) { nxt[i] = W4[(size_t)((pos + i) % nquads) * 64]; }
            pos += G;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) {
                acc0 = mfma(cur[i].x, b0, acc0); acc0 = mfma(cur[i].y, b0, acc0);
                acc0 = mfma(cur[i].z, b0, acc0); acc0 = mfma(cur[i].w, b0, acc0);
                if (MODE == 3) {
                    acc1 = mfma(cur[i].x, b1, acc1); acc1 = mfma(cur[i].y, b1, acc1);
                    acc1 = mfma(cur[i].z, b1, acc1); acc1 = mfma(cur[i].w, b1, acc1);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < G; ++i) cur[i] = nxt[i];
        }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// mode 4/5: A quads come from LDS (staged once per workgroup), one / two column tiles per wave
template <int NT>
__global__ __launch_bounds__(256, 2) void probe_lds(const float4* __restrict__ W, float* out, int iters, int nquads) {
    extern __shared__ float4 wl[];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < nquads * 64; i += 256) wl[i] = W[i];
    __syncthreads();
    v16f acc0, acc1;
    for (int i = 0; i < 16; ++i) { acc0[i] = 0.f; acc1[i] = 0.f; }
    float b0 = (float)lane * 1e-3f, b1 = b0 + 1.f;
    const float4* W4 = wl + lane;
    int pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            float4 a = W4[(size_t)((pos + i) % nquads) * 64];
            acc0 = mfma(a.x, b0, acc0); acc0 = mfma(a.y, b0, acc0); acc0 = mfma(a.z, b0, acc0); acc0 = mfma(a.w, b0, acc0);
            if (NT == 2) { acc1 = mfma(a.x, b1, acc1); acc1 = mfma(a.y, b1, acc1); acc1 = mfma(a.z, b1, acc1); acc1 = mfma(a.w, b1, acc1); }
        }
        pos += 8;
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += acc0[i] + acc1[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NT>
void run_lds(const char* name, int blocks_per_cu) {
    const int nquads = 48, iters = 2000;    // 48 KB of weights in LDS
    float4* W; float* out;
    hipMalloc(&W, (size_t)nquads * 64 * sizeof(float4));
    hipMemset(W, 0, (size_t)nquads * 64 * sizeof(float4));
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t lds = (size_t)nquads * 1024;
    hipLaunchKernelGGL(probe_lds<NT>, dim3(blocks), dim3(256), lds, 0, W, out, 10, nquads);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_lds<NT>, dim3(blocks), dim3(256), lds, 0, W, out, iters, nquads);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * 4 * iters * 32 * NT;
    double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-44s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, tf, tf / 157.3 * 100);
    hipFree(W); hipFree(out);
}

template <int MODE>
void run(const char* name, int blocks_per_cu, int mfma_per_iter) {
    const int nquads = 4096, iters = 2000;
    float4* W; float* out;
    hipMalloc(&W, (size_t)nquads * 64 * sizeof(float4));
    hipMemset(W, 0, (size_t)nquads * 64 * sizeof(float4));
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, W, out, 10, nquads);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, W, out, iters, nquads);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mf = (double)blocks * 4 * iters * mfma_per_iter;     // MFMAs issued
    double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-44s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, tf, tf / 157.3 * 100);
    hipFree(W); hipFree(out);
}

// modes 6-9: the operand delivery of csrc/conv.hip in isolation. B quads (activations) always come from LDS by ds_read_b128,
// one per 4 MFMAs per pixel tile; A quads (weights): 6 = from global/L2, each feeding NT=3 pixel tiles (the current kernel);
// 7 = from LDS, NT=3; 8 = 2 weight quads x 2 pixel tiles from LDS (16 MFMAs per 4 LDS reads); 9 = like 6 plus a
// workgroup barrier every 216 MFMAs (the per-chunk barrier).
template <int MODE>
__global__ __launch_bounds__(256, 2) void probe_conv(const float4* __restrict__ W, float* out, int iters, int nquads) {
    extern __shared__ float4 wl[];                       // [0, 24*64): weight quads; [24*64, ...): "patch" quads
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 48 * 64; i += 256) wl[i] = W[i % (nquads * 64)];
    __syncthreads();
    v16f acc[4];
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const float4* A4 = wl + lane;
    const float4* B4 = wl + 24 * 64 + lane;
    const float4* G4 = W + lane;
    int pos = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 6; ++i) {
            if (MODE == 8) {
                const float4 a0 = A4[(size_t)((pos + 2 * i) % 24) * 64], a1 = A4[(size_t)((pos + 2 * i + 1) % 24) * 64];
                const float4 b0 = B4[(size_t)((pos + i) % 24) * 64], b1 = B4[(size_t)((pos + i + 7) % 24) * 64];
                acc[0] = mfma(a0.x, b0.x, acc[0]); acc[0] = mfma(a0.y, b0.y, acc[0]); acc[0] = mfma(a0.z, b0.z, acc[0]); acc[0] = mfma(a0.w, b0.w, acc[0]);
                acc[1] = mfma(a0.x, b1.x, acc[1]); acc[1] = mfma(a0.y, b1.y, acc[1]); acc[1] = mfma(a0.z, b1.z, acc[1]); acc[1] = mfma(a0.w, b1.w, acc[1]);
                acc[2] = mfma(a1.x, b0.x, acc[2]); acc[2] = mfma(a1.y, b0.y, acc[2]); acc[2] = mfma(a1.z, b0.z, acc[2]); acc[2] = mfma(a1.w, b0.w, acc[2]);
                acc[3] = mfma(a1.x, b1.x, acc[3]); acc[3] = mfma(a1.y, b1.y, acc[3]); acc[3] = mfma(a1.z, b1.z, acc[3]); acc[3] = mfma(a1.w, b1.w, acc[3]);
            } else {
                const float4 a = (MODE == 7) ? A4[(size_t)((pos + i) % 24) * 64] : G4[(size_t)((pos + i) % nquads) * 64];
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const float4 b = B4[(size_t)((pos + i + 5 * t) % 24) * 64];
                    acc[t] = mfma(a.x, b.x, acc[t]); acc[t] = mfma(a.y, b.y, acc[t]); acc[t] = mfma(a.z, b.z, acc[t]); acc[t] = mfma(a.w, b.w, acc[t]);
                }
            }
        }
        pos += 6;
        if (MODE == 9 && (it % 3) == 2) __syncthreads();
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int i = 0; i < 16; ++i) s += acc[t][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run_conv(const char* name, int blocks_per_cu) {
    const int nquads = 4096, iters = 2000;
    float4* W; float* out;
    hipMalloc(&W, (size_t)nquads * 64 * sizeof(float4));
    hipMemset(W, 0, (size_t)nquads * 64 * sizeof(float4));
    int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    size_t lds = (size_t)48 * 1024;
    hipFuncSetAttribute((const void*)probe_conv<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(probe_conv<MODE>, dim3(blocks), dim3(256), lds, 0, W, out, 10, nquads);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(probe_conv<MODE>, dim3(blocks), dim3(256), lds, 0, W, out, iters, nquads);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const int per_it = MODE == 8 ? 6 * 16 : 6 * 12;
    double mf = (double)blocks * 4 * iters * per_it;
    double tf = mf * 4096.0 / (ms * 1e-3) / 1e12;
    printf("%-60s blocks/CU=%d  %.3f ms  %.1f TFLOP/s  (%.1f%% of 157.3)\n", name, blocks_per_cu, ms, tf, tf / 157.3 * 100);
    hipFree(W); hipFree(out);
}

int main() {
    for (int bpc : {1, 2}) {
        run<0>("0: one dependent chain, regs", bpc, 32);
        run<1>("1: two independent chains, regs", bpc, 32);
        run<2>("2: one chain, A quad from global (G=8)", bpc, 32);
        run<3>("3: two tiles share A quad (NT=2)", bpc, 64);
        run_lds<1>("4: one chain, A quad from LDS", bpc);
        run_lds<2>("5: two tiles, A quad from LDS", bpc);
    }
    for (int bpc : {1, 2}) {
        run_conv<6>("6: conv-like, A from L2 (NT=3), B from LDS", bpc);
        run_conv<7>("7: conv-like, A from LDS (NT=3), B from LDS", bpc);
        run_conv<8>("8: conv-like, 2x2 blocking, A and B from LDS", bpc);
        run_conv<9>("9: like 6 + barrier every 216 MFMAs", bpc);
    }
    return 0;
}
