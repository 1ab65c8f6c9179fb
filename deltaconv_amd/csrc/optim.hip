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

// ---- Adam, all parameters in one launch ----------------------------------------------------------------------------------------
// The optimizer of the reference's shape-segmentation script (/root/reference/experiments/train_shapeseg.py:82: torch.optim.Adam(lr = 5e-3),
// betas (0.9, 0.999), eps 1e-8, no weight decay, no amsgrad), in torch's op order (torch/optim/adam.py, single-tensor form):
//     t = step + 1;  g' = g + wd p;  m += (1 - b1)(g' - m);  v = b2 v + (1 - b2) g' g';
//     p += -(lr / (1 - b1^t)) * (m / (sqrt(v) / sqrt(1 - b2^t) + eps))
// ATen's fused form takes three multi_tensor_apply launches of ~43 us for the 8 x 128-channel segmentation net; at one cloud per
// rank (8-GPU strong scaling) that is 4 % of the step.  `step` is ONE device scalar for all parameters: every workgroup reads it,
// the LAST workgroup to finish (a ticket counter, reset for the next launch) writes step + 1 -- no second launch, and no workgroup
// can read the new value.  lr from device memory as in dc_sgd_step.  HBM-bound: 28 bytes per parameter.
namespace {
constexpr int ADAM_MAX_TENSORS = 80;      // 80 * (4 * 8 + 4) + scalars < 4 KiB of kernel arguments

struct AdamTable {
    float* p[ADAM_MAX_TENSORS];
    const float* g[ADAM_MAX_TENSORS];
    float* m[ADAM_MAX_TENSORS];
    float* v[ADAM_MAX_TENSORS];
    int first_chunk[ADAM_MAX_TENSORS + 1];
    int numel[ADAM_MAX_TENSORS];
    int count;
};

__global__ __launch_bounds__(SGD_THREADS) void adam_kernel(AdamTable t, const float* __restrict__ lr_dev, float* step_dev,
                                                           int* ticket, int bump, double beta1d, double beta2d, float eps,
                                                           float weight_decay) {
    __shared__ float coef[2];             // step size lr / (1 - b1^t), sqrt(1 - b2^t)
    if (threadIdx.x == 0) {
        const double tt = (double)*step_dev + 1.0;
        coef[0] = (float)((double)*lr_dev / (1.0 - pow(beta1d, tt)));
        coef[1] = (float)sqrt(1.0 - pow(beta2d, tt));
    }
    __syncthreads();
    // (torch hands `beta2`, `1 - beta1`, `1 - beta2` to its kernels as doubles rounded to fp32: 1.f - (float)beta2 is 1.3e-5 off)
    const float step_size = coef[0], bc2s = coef[1], beta2 = (float)beta2d, w1 = (float)(1.0 - beta1d), w2 = (float)(1.0 - beta2d);
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
    float* __restrict__ m = t.m[lo];
    float* __restrict__ v = t.v[lo];
    auto one = [&](float& pq, float gq, float& mq, float& vq) {
        if (weight_decay != 0.f) gq = gq + weight_decay * pq;
        mq = mq + w1 * (gq - mq);
        vq = vq * beta2 + w2 * gq * gq;
        const float denom = sqrtf(vq) / bc2s + eps;
        pq = pq + (-step_size) * (mq / denom);
    };
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15) == 0;
#pragma unroll
    for (int it = 0; it < SGD_CHUNK / (SGD_THREADS * 4); ++it) {
        const long e = base + ((long)it * SGD_THREADS + threadIdx.x) * 4;
        if (e >= n) break;
        if (vec && e + 3 < n) {
            float4 pv = *reinterpret_cast<const float4*>(p + e), mv = *reinterpret_cast<const float4*>(m + e),
                   vv = *reinterpret_cast<const float4*>(v + e);
            const float4 gv = *reinterpret_cast<const float4*>(g + e);
            one(pv.x, gv.x, mv.x, vv.x); one(pv.y, gv.y, mv.y, vv.y); one(pv.z, gv.z, mv.z, vv.z); one(pv.w, gv.w, mv.w, vv.w);
            *reinterpret_cast<float4*>(m + e) = mv;
            *reinterpret_cast<float4*>(v + e) = vv;
            *reinterpret_cast<float4*>(p + e) = pv;
        } else {
            for (long q = e; q < e + 4 && q < n; ++q) one(p[q], g[q], m[q], v[q]);
        }
    }
    if (bump) {
        // every workgroup read *step_dev above, before it arrives here: the one that takes the last ticket is alone
        __syncthreads();
        if (threadIdx.x == 0 && atomicAdd(ticket, 1) == (int)gridDim.x - 1) {
            atomicExch(ticket, 0);
            *step_dev = *step_dev + 1.f;
        }
    }
}
}  // namespace

// params / grads / exp_avgs / exp_avg_sqs / numel: HOST arrays of `count` device addresses / element counts (fp32, contiguous);
// lr: device scalar; step: device scalar (fp32, torch's `state["step"]`), incremented by one; ticket: device int32, zero (left zero).
DC_EXPORT int dc_adam_step(const int64_t* params, const int64_t* grads, const int64_t* exp_avgs, const int64_t* exp_avg_sqs,
                           const int64_t* numel, int32_t count, const float* lr, float* step, int32_t* ticket, double beta1,
                           double beta2, float eps, float weight_decay, void* stream) {
    DC_REQUIRE(count >= 0 && lr && step && ticket && (count == 0 || (params && grads && exp_avgs && exp_avg_sqs && numel)),
               "dc_adam_step: null pointer");
    hipStream_t s = static_cast<hipStream_t>(stream);
    int live = 0;
    for (int i = 0; i < count; ++i) live += numel[i] > 0;
    if (!live) return DC_OK;
    int seen = 0;
    for (int t0 = 0; t0 < count;) {
        AdamTable t;
        t.count = 0;
        int chunks = 0, i = t0;
        for (; i < count && t.count < ADAM_MAX_TENSORS; ++i) {
            DC_REQUIRE(numel[i] >= 0 && numel[i] < 2147483647L, "dc_adam_step: tensor too large");
            if (numel[i] == 0) continue;
            DC_REQUIRE(params[i] && grads[i] && exp_avgs[i] && exp_avg_sqs[i], "dc_adam_step: null tensor");
            const int c = t.count++;
            t.p[c] = reinterpret_cast<float*>(params[i]);
            t.g[c] = reinterpret_cast<const float*>(grads[i]);
            t.m[c] = reinterpret_cast<float*>(exp_avgs[i]);
            t.v[c] = reinterpret_cast<float*>(exp_avg_sqs[i]);
            t.numel[c] = (int)numel[i];
            t.first_chunk[c] = chunks;
            chunks += dc_cdiv(numel[i], SGD_CHUNK);
        }
        t0 = i;
        t.first_chunk[t.count] = chunks;
        seen += t.count;
        if (chunks)          // the launch that holds the last live tensor advances `step`
            hipLaunchKernelGGL(adam_kernel, dim3(chunks), dim3(SGD_THREADS), 0, s, t, lr, step, ticket, (int)(seen == live), beta1,
                               beta2, eps, weight_decay);
    }
    DC_CHECK_LAUNCH("dc_adam_step");
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
