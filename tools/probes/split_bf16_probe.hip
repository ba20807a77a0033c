// Micro-probe for the "split-bf16" lever of DESIGN.md 8: an f32 product a*b as three bf16 MFMA products
//   a = ah + al, b = bh + bl (ah = bf16(a), al = bf16(a - ah)):  a*b ~ ah*bh + ah*bl + al*bh   (al*bl ~ 2^-16 |ab| dropped)
// (1) rate: chained v_mfma_f32_32x32x2_f32 (8 per 16 channels) against v_mfma_f32_32x32x16_bf16 (3 per 16 channels), operands in
//     registers, two waves per SIMD, with the shader clock measured (s_memtime / s_memrealtime) -- the compute ceiling of the idea
// (2) error: one 32x32x16 block on random data, three-product bf16 result vs the f32 MFMA and vs float64
// build: hipcc --offload-arch=gfx950 -O3 -o split_bf16_probe split_bf16_probe.hip ; run: ./split_bf16_probe
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>
typedef float v16f __attribute__((ext_vector_type(16)));
typedef __bf16 v8bf __attribute__((ext_vector_type(8)));

__device__ __forceinline__ unsigned long long rt() { unsigned long long t; asm volatile("s_memrealtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }
__device__ __forceinline__ unsigned long long ct() { unsigned long long t; asm volatile("s_memtime %0\n s_waitcnt lgkmcnt(0)" : "=s"(t)); return t; }

template <int MODE>
__global__ __launch_bounds__(256, 2) void rate(float* out, unsigned long long* stamps, int iters) {
    const int lane = threadIdx.x & 63;
    v16f acc[4];
    for (int e = 0; e < 4; ++e) for (int i = 0; i < 16; ++i) acc[e][i] = 0.f;
    const float a = 1e-3f * lane, b = 1.0f + 1e-3f * lane;
    v8bf ah, bh;
    for (int i = 0; i < 8; ++i) ah[i] = (__bf16)(a + i), bh[i] = (__bf16)(b - i);
    const unsigned long long r0 = rt(), c0 = ct();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (MODE == 0) {
#pragma unroll
                for (int k = 0; k < 8; ++k) acc[e] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[e], 0, 0, 0);
            } else {
#pragma unroll
                for (int k = 0; k < 3; ++k) acc[e] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc[e], 0, 0, 0);
            }
        }
    }
    const unsigned long long c1 = ct(), r1 = rt();
    float s = 0.f;
    for (int e = 0; e < 4; ++e) for (int i = 0; i < 16; ++i) s += acc[e][i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0) stamps[2 * blockIdx.x] = r1 - r0, stamps[2 * blockIdx.x + 1] = c1 - c0;
}

// one 32x32x16 block: A [32][16], B [16][32] row-major floats -> D [32][32] by f32 MFMA and by the three bf16 products
__global__ void accuracy(const float* A, const float* B, float* Df, float* Ds) {
    const int lane = threadIdx.x, c = lane & 31, h = lane >> 5;
    v16f f, s;
    for (int i = 0; i < 16; ++i) f[i] = 0.f, s[i] = 0.f;
    for (int k = 0; k < 8; ++k) f = __builtin_amdgcn_mfma_f32_32x32x2f32(A[c * 16 + 2 * k + h], B[(2 * k + h) * 32 + c], f, 0, 0, 0);
    v8bf ah, al, bh, bl;
    for (int i = 0; i < 8; ++i) {          // lane (c, h) holds k = 8h .. 8h+7 of row / column c
        const float a = A[c * 16 + 8 * h + i], b = B[(8 * h + i) * 32 + c];
        ah[i] = (__bf16)a, al[i] = (__bf16)(a - (float)ah[i]);
        bh[i] = (__bf16)b, bl[i] = (__bf16)(b - (float)bh[i]);
    }
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, s, 0, 0, 0);      // small terms first
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, s, 0, 0, 0);
    s = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, s, 0, 0, 0);
    for (int i = 0; i < 16; ++i) {
        const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
        Df[row * 32 + c] = f[i], Ds[row * 32 + c] = s[i];
    }
}

int main() {
    const int blocks = 512, iters = 4000;
    float* out; unsigned long long* st;
    hipMalloc(&out, blocks * 256 * 4); hipMalloc(&st, blocks * 16);
    std::vector<unsigned long long> h(2 * blocks);
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 2; ++rep) {
            if (mode == 0) hipLaunchKernelGGL(rate<0>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
            else hipLaunchKernelGGL(rate<1>, dim3(blocks), dim3(256), 0, 0, out, st, iters);
            hipDeviceSynchronize();
        }
        hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
        double r = 0, c = 0;
        for (int b = 0; b < blocks; ++b) r += h[2 * b], c += h[2 * b + 1];
        r /= blocks, c /= blocks;
        const double us = r / 100.0, ghz = c / us / 1e3;
        // per wave: iters * 4 tiles * (32*32*16*2 flop per 16-channel block); 4 waves x 512 blocks, 2 blocks per CU at a time
        const double flop = (double)iters * 4 * 32768.0 * 4 * blocks;
        const double t = us * 1e-6 * (blocks / 512.0);          // one round of 512 resident workgroups
        printf("%s: %.1f us per workgroup, clock %.3f GHz, %.1f f32-equivalent TFLOP/s\n",
               mode == 0 ? "f32 MFMA 32x32x2 (8 per 16 channels)  " : "bf16 MFMA 32x32x16 (3 per 16 channels)", us, ghz, flop / t / 1e12);
    }
    std::vector<float> A(512), B(512), Df(1024), Ds(1024);
    unsigned seed = 12345;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return ((seed >> 8) & 0xffff) / 32768.0f - 1.0f; };
    for (auto& v : A) v = rnd();
    for (auto& v : B) v = rnd();
    float *dA, *dB, *dF, *dS;
    hipMalloc(&dA, 2048); hipMalloc(&dB, 2048); hipMalloc(&dF, 4096); hipMalloc(&dS, 4096);
    hipMemcpy(dA, A.data(), 2048, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 2048, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(accuracy, dim3(1), dim3(64), 0, 0, dA, dB, dF, dS);
    hipMemcpy(Df.data(), dF, 4096, hipMemcpyDeviceToHost); hipMemcpy(Ds.data(), dS, 4096, hipMemcpyDeviceToHost);
    double ef = 0, es = 0, mx = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0; for (int k = 0; k < 16; ++k) ref += (double)A[i * 16 + k] * (double)B[k * 32 + j];
        ef = fmax(ef, fabs(Df[i * 32 + j] - ref)); es = fmax(es, fabs(Ds[i * 32 + j] - ref)); mx = fmax(mx, fabs(ref));
    }
    printf("one 32x32x16 block, max |error| / max |value|: f32 MFMA %.2e, three bf16 products %.2e\n", ef / mx, es / mx);
    return 0;
}
