// MFMA issue-rate lab: fp32-input MFMA forms, register-only loops, to find the real per-SIMD issue interval of
//   v_mfma_f32_32x32x2_f32 vs v_mfma_f32_16x16x4_f32 with 1..4 accumulators, 1..2 waves per SIMD, and with LDS reads
//   interleaved the way the GEMM main loop has them.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_lab.hip -o /tmp/mfma_lab && /tmp/mfma_lab
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1);} } while (0)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) acc[i][q] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 16; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k16(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 4; ++q) acc[i][q] = 0.f;
    float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u)
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
    }
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int q = 0; q < 4; ++q) s += acc[i][q];
    if (s == 12345.f) out[0] = s;
}
// 2x2 tile of 32x32 accumulators, operands re-read from LDS every 4 k-steps (ds_read_b128), like the GEMM loop
__global__ __launch_bounds__(256) void k32_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 128 * 36];
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) sm[i] = (float)(i % 17) * 0.01f;
    __syncthreads();
    f32x16 acc[2][2];
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 31, lh = lane >> 5;
    const float* a = sm + ((wave >> 1) * 64 + li) * 36 + 4 * lh;
    const float* b = sm + 128 * 36 + ((wave & 1) * 64 + li) * 36 + 4 * lh;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            f32x4 fa[2], fb[2];
            fa[0] = *(const f32x4*)(a + 8 * j); fa[1] = *(const f32x4*)(a + 32 * 36 + 8 * j);
            fb[0] = *(const f32x4*)(b + 8 * j); fb[1] = *(const f32x4*)(b + 32 * 36 + 8 * j);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int jn = 0; jn < 2; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[i][t], fb[jn][t], acc[i][jn], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 2; ++i) for (int j = 0; j < 2; ++j) for (int q = 0; q < 16; ++q) s += acc[i][j][q];
    if (s == 12345.f) out[0] = s;
}
// same work with 16x16x4: a 64x64 wave tile = 4x4 accumulators of 16x16
__global__ __launch_bounds__(256) void k16_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[2 * 128 * 36];
    for (int i = threadIdx.x; i < 2 * 128 * 36; i += 256) sm[i] = (float)(i % 17) * 0.01f;
    __syncthreads();
    f32x4 acc[4][4];
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) acc[i][j][q] = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, li = lane & 15, lq = lane >> 4;
    // lane (i = l & 15, k = l >> 4): one ds_read_b128 at k-offset 4*lq covers 4 MFMA steps {4 lq' + t}
    const float* a = sm + ((wave >> 1) * 64 + li) * 36 + 4 * lq;
    const float* b = sm + 128 * 36 + ((wave & 1) * 64 + li) * 36 + 4 * lq;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {    // 16 k per j (4 lane quarters x 4 elements)
            f32x4 fa[4], fb[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { fa[i] = *(const f32x4*)(a + i * 16 * 36 + 16 * j); fb[i] = *(const f32x4*)(b + i * 16 * 36 + 16 * j); }
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int jn = 0; jn < 4; ++jn)
                        acc[i][jn] = __builtin_amdgcn_mfma_f32_16x16x4f32(fa[i][t], fb[jn][t], acc[i][jn], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int j = 0; j < 4; ++j) for (int q = 0; q < 4; ++q) s += acc[i][j][q];
    if (s == 12345.f) out[0] = s;
}

template <class F> float timeit(F f) {
    f();
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipEventRecord(a)); for (int i = 0; i < 5; ++i) f(); CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b)); CK(hipGetLastError());
    return ms * 1e3f / 5;
}

int main() {
    float* d; CK(hipMalloc(&d, 4));
    const int iters = 2000;
    for (int bpc : {1, 2}) {    // workgroups (of 4 waves) per CU -> waves per SIMD
        const int grid = 256 * bpc;
#define RUN(KERN, LABEL, NMFMA_PER_IT, FLOP_PER_MFMA, ...)                                                                  \
    {                                                                                                                       \
        float us = timeit([&] { hipLaunchKernelGGL(KERN, dim3(grid), dim3(256), 0, 0, __VA_ARGS__); });                     \
        const double mf = (double)grid * 4 * iters * (NMFMA_PER_IT);                                                        \
        printf("  %-44s %d wave/SIMD  %9.1f us  %7.1f TF/s  %6.1f cycles/MFMA/SIMD @2.4GHz\n", LABEL, bpc, us,              \
               mf * (FLOP_PER_MFMA) / us / 1e6, us * 1e-6 * 2.4e9 / ((double)bpc * iters * (NMFMA_PER_IT)));                \
    }
        RUN(k32<1>, "32x32x2 1 acc (dependent chain)", 16, 4096.0, d, iters, 1.f, 2.f)
        RUN(k32<2>, "32x32x2 2 acc", 32, 4096.0, d, iters, 1.f, 2.f)
        RUN(k32<4>, "32x32x2 4 acc", 64, 4096.0, d, iters, 1.f, 2.f)
        RUN(k16<1>, "16x16x4 1 acc (dependent chain)", 16, 2048.0, d, iters, 1.f, 2.f)
        RUN(k16<2>, "16x16x4 2 acc", 32, 2048.0, d, iters, 1.f, 2.f)
        RUN(k16<4>, "16x16x4 4 acc", 64, 2048.0, d, iters, 1.f, 2.f)
        RUN(k16<8>, "16x16x4 8 acc", 128, 2048.0, d, iters, 1.f, 2.f)
        RUN(k32_lds, "32x32x2 2x2 acc + ds_read_b128 fragments", 64, 4096.0, d, iters)
        RUN(k16_lds, "16x16x4 4x4 acc + ds_read_b128 fragments", 128, 2048.0, d, iters)
    }
    return 0;
}
