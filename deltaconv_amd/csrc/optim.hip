// SGD with momentum and weight decay for ALL parameters of a model in one launch -- the optimizer of the reference's
// classification / part-segmentation scripts (/root/reference/experiments/train_modelnet.py:67, train_scanobjectnn.py:77,
// train_shapenet.py:95: torch.optim.SGD(lr, momentum = 0.9, weight_decay = 1e-4), dampening 0, no Nesterov):
//     g' = g + wd * p;   buf = momentum * buf + g';   p -= lr * buf          (buf starts at zero: the first step gives buf = g')
// torch's own fused form (multi_tensor_apply) hands every workgroup 64 K elements: ~150 workgroups for the 2 M parameters of
// the ModelNet40 net, 31 us of a 3 ms step.  Here: pointer table by value in the kernel arguments (no device table to keep in
// sync with autograd's fresh gradient tensors), 4096 elements per workgroup, float4 where the three pointers allow it.
// `lr` is read from DEVICE memory: a scheduler changes it between replays of a captured step without a re-capture.
// HBM-bound: 20 bytes per parameter.
#include "common.h"

namespace {

constexpr int SGD_MAX_TENSORS = 96;       // 96 * (3 * 8 + 4) + scalars < 4 KiB of kernel arguments
constexpr int SGD_CHUNK = 4096;           // elements per workgroup
constexpr int SGD_THREADS = 256;

struct SgdTable {
    float* p[SGD_MAX_TENSORS];
    const float* g[SGD_MAX_TENSORS];
    float* buf[SGD_MAX_TENSORS];
    int first_chunk[SGD_MAX_TENSORS + 1];  // prefix sums of the tensors' chunk counts
    int numel[SGD_MAX_TENSORS];
    int count;
};

__global__ __launch_bounds__(SGD_THREADS) void sgd_kernel(SgdTable t, const float* __restrict__ lr_dev, float momentum,
                                                          float weight_decay) {
    // which tensor does this chunk belong to: binary search over <= 96 prefix sums (kernel arguments: scalar loads)
    int lo = 0, hi = t.count;
    const int chunk = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first_chunk[mid] <= chunk) lo = mid;
        else hi = mid;
    }
    const int n = t.numel[lo];
    const long base = (long)(chunk - t.first_chunk[lo]) * SGD_CHUNK;
    float* __restrict__ p = t.p[lo];
    const float* __restrict__ g = t.g[lo];
    float* __restrict__ b = t.buf[lo];
    const float lr = *lr_dev;
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(b)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < SGD_CHUNK / (SGD_THREADS * 4); ++it) {
        const long e = base + ((long)it * SGD_THREADS + threadIdx.x) * 4;
        if (e >= n) break;
        if (vec && e + 3 < n) {
            const float4 pv = *reinterpret_cast<const float4*>(p + e), gv = *reinterpret_cast<const float4*>(g + e);
            float4 bv = *reinterpret_cast<const float4*>(b + e), po;
            bv.x = fmaf(momentum, bv.x, fmaf(weight_decay, pv.x, gv.x)); po.x = fmaf(-lr, bv.x, pv.x);
            bv.y = fmaf(momentum, bv.y, fmaf(weight_decay, pv.y, gv.y)); po.y = fmaf(-lr, bv.y, pv.y);
            bv.z = fmaf(momentum, bv.z, fmaf(weight_decay, pv.z, gv.z)); po.z = fmaf(-lr, bv.z, pv.z);
            bv.w = fmaf(momentum, bv.w, fmaf(weight_decay, pv.w, gv.w)); po.w = fmaf(-lr, bv.w, pv.w);
            *reinterpret_cast<float4*>(b + e) = bv;
            *reinterpret_cast<float4*>(p + e) = po;
        } else {
            for (long q = e; q < e + 4 && q < n; ++q) {
                const float bq = fmaf(momentum, b[q], fmaf(weight_decay, p[q], g[q]));
                b[q] = bq;
                p[q] = fmaf(-lr, bq, p[q]);
            }
        }
    }
}

}  // namespace

// params / grads / bufs: HOST arrays of `count` device pointers (fp32 tensors of numel[i] elements, contiguous); lr: device
// scalar.  Tensors beyond 96 go into further launches.  Stream-ordered, capturable.
DC_EXPORT int dc_sgd_step(const int64_t* params, const int64_t* grads, const int64_t* bufs, const int64_t* numel, int32_t count,
                          const float* lr, float momentum, float weight_decay, void* stream) {
    DC_REQUIRE(count >= 0 && (count == 0 || (params && grads && bufs && numel && lr)), "dc_sgd_step: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < count; t0 += SGD_MAX_TENSORS) {
        SgdTable t;
        t.count = 0;
        int chunks = 0;
        for (int i = t0; i < count && t.count < SGD_MAX_TENSORS; ++i) {
            DC_REQUIRE(numel[i] >= 0 && numel[i] < 2147483647L, "dc_sgd_step: tensor too large");
            if (numel[i] == 0) continue;
            DC_REQUIRE(params[i] && grads[i] && bufs[i], "dc_sgd_step: null tensor");
            const int c = t.count++;
            t.p[c] = reinterpret_cast<float*>(params[i]);
            t.g[c] = reinterpret_cast<const float*>(grads[i]);
            t.buf[c] = reinterpret_cast<float*>(bufs[i]);
            t.numel[c] = (int)numel[i];
            t.first_chunk[c] = chunks;
            chunks += dc_cdiv(numel[i], SGD_CHUNK);
        }
        t.first_chunk[t.count] = chunks;
        if (chunks) hipLaunchKernelGGL(sgd_kernel, dim3(chunks), dim3(SGD_THREADS), 0, s, t, lr, momentum, weight_decay);
    }
    DC_CHECK_LAUNCH("dc_sgd_step");
    return DC_OK;
}

// ---- several small (strided) copies in one launch ---------------------------------------------------------------------------
// Step glue around a replayed step: the batch load (`data.to(device)` of the reference's loops, experiments/train_modelnet.py:99 --
// here device-to-device into the captured inputs: pos, normals, labels) and the first layer's operand blocks (`torch.cat([x, ...])`
// of deltaconv/nn/deltaconv.py:57,65: x and v of layer 0 arrive from outside and are copied into the left columns of the layer's
// operand buffers) were one ~4.7 us launch per tensor.  Units are 4-byte words (fp32 / int32 / halves of int64); entry i copies
// rows_i x cols_i words, row strides ld_src_i / ld_dst_i.  All of these are <= 0.5 MB: one word per thread access, coalesced along
// the row, is enough.
namespace {
constexpr int CPY_MAX = 16;
struct CopyTable {
    const unsigned* src[CPY_MAX];
    unsigned* dst[CPY_MAX];
    long lds[CPY_MAX], ldd[CPY_MAX], words[CPY_MAX];
    int cols[CPY_MAX];
    int first_block[CPY_MAX + 1];
    int count;
};
__global__ __launch_bounds__(256) void copy_many_kernel(CopyTable t) {
    int lo = 0, hi = t.count;
    const int blk = blockIdx.x;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (t.first_block[mid] <= blk) lo = mid;
        else hi = mid;
    }
    const unsigned* __restrict__ src = t.src[lo];
    unsigned* __restrict__ dst = t.dst[lo];
    const long lds = t.lds[lo], ldd = t.ldd[lo], words = t.words[lo];
    const int cols = t.cols[lo];
    const long base = (long)(blk - t.first_block[lo]) * 1024 + threadIdx.x;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const long w = base + it * 256;
        if (w < words) {
            const long r = w / cols, c = w % cols;
            dst[r * ldd + c] = src[r * lds + c];
        }
    }
}
}  // namespace

// srcs / dsts: HOST arrays of `count` device addresses (4-byte aligned); ld_src / ld_dst / rows / cols in 4-byte words.
DC_EXPORT int dc_copy_many(const int64_t* srcs, const int64_t* dsts, const int64_t* ld_src, const int64_t* ld_dst,
                           const int32_t* rows, const int32_t* cols, int32_t count, void* stream) {
    DC_REQUIRE(count >= 0 && (count == 0 || (srcs && dsts && ld_src && ld_dst && rows && cols)), "dc_copy_many: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    for (int t0 = 0; t0 < count; t0 += CPY_MAX) {
        CopyTable t;
        t.count = 0;
        int blocks = 0;
        for (int i = t0; i < count && i < t0 + CPY_MAX; ++i) {
            DC_REQUIRE(rows[i] >= 0 && cols[i] >= 0 && ld_src[i] >= cols[i] && ld_dst[i] >= cols[i], "dc_copy_many: bad entry");
            if (rows[i] == 0 || cols[i] == 0) continue;
            DC_REQUIRE(srcs[i] && dsts[i] && srcs[i] % 4 == 0 && dsts[i] % 4 == 0, "dc_copy_many: null or misaligned tensor");
            const int c = t.count++;
            t.src[c] = reinterpret_cast<const unsigned*>(srcs[i]);
            t.dst[c] = reinterpret_cast<unsigned*>(dsts[i]);
            t.lds[c] = (long)ld_src[i];
            t.ldd[c] = (long)ld_dst[i];
            t.words[c] = (long)rows[i] * cols[i];
            t.cols[c] = cols[i];
            t.first_block[c] = blocks;
            blocks += dc_cdiv(t.words[c], 1024L);
        }
        if (!t.count) continue;
        t.first_block[t.count] = blocks;
        hipLaunchKernelGGL(copy_many_kernel, dim3(blocks), dim3(256), 0, s, t);
    }
    DC_CHECK_LAUNCH("dc_copy_many");
    return DC_OK;
}
