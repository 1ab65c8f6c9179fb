// Lab: what does one k-step of the split-product GEMM loop cost on a SIMD?  24 v_mfma_f32_32x32x16_bf16 (2 x 2 accumulators x 6 partial
// products) + the 3-way bf16 split of 4 fragments (8 fp32 per lane each), in several instruction mixes / orders, 1 or 2 waves per SIMD.
// No memory traffic inside the loop: registers only.   hipcc --offload-arch=gfx950 -O3 tools/x3_lab.hip -o /tmp/x3_lab && /tmp/x3_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct Planes { u32x4 h, m, l; };
__device__ __forceinline__ float bf(unsigned x) { return __builtin_bit_cast(float, x); }
__device__ __forceinline__ unsigned pk_rne(float a, float b) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2){a, b}, bf16x2));
}
__device__ __forceinline__ float sub_plain(float a, float b) {      // a - b, never SLP-packed
    float r;
    asm volatile("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
// SPLIT: 0 = RNE planes, compiler's subtraction (v_pk_add_f32 where it packs); 1 = RNE planes, plain v_sub_f32;
// 2 = truncated hi / mid planes (v_perm_b32 packs the top halves, v_and clears the low ones), plain v_sub_f32, RNE lo
template <int SPLIT>
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    if (SPLIT == 0) {
        h = pk_rne(x0, x1);
        const float r0 = x0 - bf(h << 16), r1 = x1 - bf(h & 0xFFFF0000u);
        m = pk_rne(r0, r1);
        const float s0 = r0 - bf(m << 16), s1 = r1 - bf(m & 0xFFFF0000u);
        l = pk_rne(s0, s1);
    } else if (SPLIT == 1) {
        h = pk_rne(x0, x1);
        const float r0 = sub_plain(x0, bf(h << 16)), r1 = sub_plain(x1, bf(h & 0xFFFF0000u));
        m = pk_rne(r0, r1);
        const float s0 = sub_plain(r0, bf(m << 16)), s1 = sub_plain(r1, bf(m & 0xFFFF0000u));
        l = pk_rne(s0, s1);
    } else {
        const unsigned u0 = __builtin_bit_cast(unsigned, x0), u1 = __builtin_bit_cast(unsigned, x1);
        h = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        const float r0 = sub_plain(x0, bf(u0 & 0xFFFF0000u)), r1 = sub_plain(x1, bf(u1 & 0xFFFF0000u));
        const unsigned v0 = __builtin_bit_cast(unsigned, r0), v1 = __builtin_bit_cast(unsigned, r1);
        m = __builtin_amdgcn_perm(v1, v0, 0x07060302u);
        const float s0 = sub_plain(r0, bf(v0 & 0xFFFF0000u)), s1 = sub_plain(r1, bf(v1 & 0xFFFF0000u));
        l = __builtin_amdgcn_perm(__builtin_bit_cast(unsigned, s1), __builtin_bit_cast(unsigned, s0), 0x07060302u);
    }
}
template <int SPLIT>
__device__ __forceinline__ void split8(const float (&x)[8], Planes& o) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        unsigned h, m, l;
        split_pair<SPLIT>(x[2 * j], x[2 * j + 1], h, m, l);
        o.h[j] = h; o.m[j] = m; o.l[j] = l;
    }
}
__device__ __forceinline__ f32x16 mfma(const u32x4 a, const u32x4 b, const f32x16 c) {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ void touch(float (&x)[8]) {
#pragma unroll
    for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(x[j]));
}
__device__ __forceinline__ void touch(Planes& p) {
#pragma unroll
    for (int j = 0; j < 4; ++j) { asm volatile("" : "+v"(p.h[j])); asm volatile("" : "+v"(p.m[j])); asm volatile("" : "+v"(p.l[j])); }
}
// MODE: 0 = MFMAs only; 1 = split only; 2 = split, then MFMAs (serial); 3 = interleaved (one fragment's split behind every 6 MFMAs,
// pinned 1 MFMA : 6 VALU); 4 = interleaved, scheduler's own order (no pinning)
// MODE 5 / 6 (second lab): the same k-step WITH its LDS traffic.  5 = the shipped structure: 8 ds_read_b128 refill the fp32 fragments,
// 144 split instructions, 4 ds_write_b128 stage the next tile;  6 = split once per workgroup at staging time: 72 split instructions on
// the thread's 16 staged elements, 12 ds_write_b64 of planes, 12 ds_read_b128 of plane fragments, no split in the fragment path.
template <int MODE, int SPLIT>
__global__ __launch_bounds__(256, 2) void lab(const float* in, float* out, int iters, long long* cyc) {
    __shared__ __attribute__((aligned(16))) float lds[16384];
    float raw[4][8];
#pragma unroll
    for (int f = 0; f < 4; ++f)
#pragma unroll
        for (int j = 0; j < 8; ++j) raw[f][j] = in[(threadIdx.x * 4 + f) * 8 + j];
    Planes p[2][4];
#pragma unroll
    for (int f = 0; f < 4; ++f) { split8<SPLIT>(raw[f], p[0][f]); p[1][f] = p[0][f]; }
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[i][j][q] = 0.f;
    auto mf = [&](int g, const Planes (&c)[4]) {
        const int pr = g / 4, r = g % 4, i = r / 2, jn = r % 2;
        const u32x4 a = pr == 0 ? c[i].l : (pr == 2 || pr == 3 ? c[i].m : c[i].h);
        const u32x4 b = pr == 1 ? c[2 + jn].l : (pr == 2 || pr == 4 ? c[2 + jn].m : c[2 + jn].h);
        acc[i][jn] = mfma(a, b, acc[i][jn]);
    };
    const long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            const Planes (&c)[4] = p[half];
            Planes (&n)[4] = p[half ^ 1];
            if (MODE == 0) {
#pragma unroll
                for (int g = 0; g < 24; ++g) mf(g, c);
            } else if (MODE == 1) {
#pragma unroll
                for (int f = 0; f < 4; ++f) { touch(raw[f]); split8<SPLIT>(raw[f], n[f]); touch(n[f]); }
            } else if (MODE == 5) {
                float* base = lds + (threadIdx.x & 255) * 4 + half * 4096;
#pragma unroll
                for (int f = 0; f < 4; ++f) {
#pragma unroll
                    for (int g = 6 * f; g < 6 * f + 6; ++g) mf(g, c);
                    split8<SPLIT>(raw[f], n[f]);
                    const float4 a = *reinterpret_cast<const float4*>(base + f * 1024), b = *reinterpret_cast<const float4*>(base + f * 1024 + 2048);
                    raw[f][0] = a.x; raw[f][1] = a.y; raw[f][2] = a.z; raw[f][3] = a.w;
                    raw[f][4] = b.x; raw[f][5] = b.y; raw[f][6] = b.z; raw[f][7] = b.w;
                    *reinterpret_cast<float4*>(base + f * 1024 + 1024 * (half ^ 1)) = float4{raw[f][0], raw[f][1], raw[f][2], raw[f][3]};
                }
#pragma unroll
                for (int g = 0; g < 24; ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                }
            } else if (MODE == 6) {
                float* base = lds + (threadIdx.x & 255) * 4 + half * 4096;
                // staging: this thread's 16 elements of the next stage -> planes -> LDS (12 ds_write_b64)
                Planes st[2];
#pragma unroll
                for (int f = 0; f < 2; ++f) {
#pragma unroll
                    for (int g = 12 * f; g < 12 * f + 12; ++g) mf(g, c);
                    touch(raw[f]);
                    split8<SPLIT>(raw[f], st[f]);
                    unsigned long long* w = reinterpret_cast<unsigned long long*>(base + f * 1024);
                    w[0] = ((unsigned long long)st[f].h[1] << 32) | st[f].h[0];   w[256] = ((unsigned long long)st[f].h[3] << 32) | st[f].h[2];
                    w[512] = ((unsigned long long)st[f].m[1] << 32) | st[f].m[0]; w[768] = ((unsigned long long)st[f].m[3] << 32) | st[f].m[2];
                    w[1024] = ((unsigned long long)st[f].l[1] << 32) | st[f].l[0]; w[1280] = ((unsigned long long)st[f].l[3] << 32) | st[f].l[2];
                }
                // fragments of the next k-step: 4 fragments x 3 planes, one ds_read_b128 each
#pragma unroll
                for (int f = 0; f < 4; ++f) {
                    const u32x4 h = *reinterpret_cast<const u32x4*>(base + f * 768), m = *reinterpret_cast<const u32x4*>(base + f * 768 + 256),
                                l = *reinterpret_cast<const u32x4*>(base + f * 768 + 512);
                    n[f].h = h; n[f].m = m; n[f].l = l;
                }
            } else if (MODE == 2) {
#pragma unroll
                for (int f = 0; f < 4; ++f) { touch(raw[f]); split8<SPLIT>(raw[f], n[f]); }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int g = 0; g < 24; ++g) mf(g, n);
            } else {
#pragma unroll
                for (int f = 0; f < 4; ++f) {
#pragma unroll
                    for (int g = 6 * f; g < 6 * f + 6; ++g) mf(g, c);
                    touch(raw[f]);
                    split8<SPLIT>(raw[f], n[f]);
                }
                if (MODE == 3) {
#pragma unroll
                    for (int g = 0; g < 24; ++g) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    }
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    const long long t1 = clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int q = 0; q < 16; ++q) s += acc[i][j][q];
#pragma unroll
    for (int f = 0; f < 4; ++f) s += bf(p[0][f].h[0]) + bf(p[1][f].l[3]);
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}
template <int MODE, int SPLIT>
void run(const char* name, const float* in, float* out, long long* cyc, int wgs_per_cu) {
    const int iters = 2000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((lab<MODE, SPLIT>), dim3(grid), dim3(256), 0, 0, in, out, 10, cyc);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((lab<MODE, SPLIT>), dim3(grid), dim3(256), 0, 0, in, out, iters, cyc);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
    // one loop trip = 2 k-steps; a SIMD holds wgs_per_cu waves
    printf("%-44s waves/SIMD %d: %8.1f ns per k-step per SIMD (wall)   %8.1f clock64 ticks per k-step per wave\n", name, wgs_per_cu,
           ms * 1e6 / (iters * 2.0), (double)c / (iters * 2.0));
}
int main() {
    float *in, *out; long long* cyc;
    hipMalloc(&in, 256 * 32 * 4); hipMalloc(&out, 512 * 256 * 4); hipMalloc(&cyc, 8);
    std::vector<float> h(256 * 32);
    for (size_t i = 0; i < h.size(); ++i) h[i] = 0.001f * (float)((i * 2654435761u) % 1999) - 1.f;
    hipMemcpy(in, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 2; ++w) {
        run<0, 0>("24 MFMAs", in, out, cyc, w);
        run<1, 0>("split: RNE, compiler subtraction", in, out, cyc, w);
        run<1, 1>("split: RNE, plain v_sub", in, out, cyc, w);
        run<1, 2>("split: truncated hi/mid, plain v_sub", in, out, cyc, w);
        run<2, 0>("serial: RNE/compiler", in, out, cyc, w);
        run<2, 1>("serial: RNE/plain", in, out, cyc, w);
        run<3, 0>("interleaved pinned: RNE/compiler", in, out, cyc, w);
        run<3, 1>("interleaved pinned: RNE/plain", in, out, cyc, w);
        run<3, 2>("interleaved pinned: truncated/plain", in, out, cyc, w);
        run<4, 0>("interleaved free: RNE/compiler", in, out, cyc, w);
        run<4, 1>("interleaved free: RNE/plain", in, out, cyc, w);
        run<4, 2>("interleaved free: truncated/plain", in, out, cyc, w);
        run<5, 0>("k-step + LDS, shipped structure", in, out, cyc, w);
        run<6, 0>("k-step + LDS, split at staging", in, out, cyc, w);
    }
    return 0;
}
