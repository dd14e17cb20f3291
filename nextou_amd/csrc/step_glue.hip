// Step glue (ABI v13): the gradient clip and the SGD update of a whole parameter list as three launches.
//
// The training step the plug-in surface prescribes (nnU-Net's train_step, mirrored by nextou_amd/harness.py) ends with
// clip_grad_norm_(parameters, 12) and torch.optim.SGD(momentum 0.99, nesterov, weight decay).step().  PyTorch runs both as
// multi-tensor-apply kernels whose launch arguments hold at most 36-110 tensors and 320 chunks: for the 358 trainable tensors
// of a cfg-2 network that is a few dozen launches, most of them a few dozen small workgroups on 256 CUs (575 us replayed).  Here the tensors are named by a
// TABLE IN DEVICE MEMORY (one row per tensor: parameter, gradient, momentum buffer, element count) and a list of (tensor, chunk)
// pairs, so one launch covers every tensor with one workgroup per 64-KB chunk:
//     multi_sumsq_kernel     per-chunk sum of squares of the gradients (float64, fixed order)
//     clip_coef_kernel       total norm (float32, as torch reports it) and the clip factor min(1, max_norm / (norm + 1e-6))
//     clip_sgd_kernel        g <- g * factor (stored, like clip_grad_norm_);  d = g + wd * p;  buf = momentum * buf + d;
//                            d = nesterov ? d + momentum * buf : buf;  p = p - lr * d
// HBM-bound elementwise work: 4 B read per element for the norm, 12 B read + 12 B written for the update.
#include "common.h"

namespace nextou {
namespace {

constexpr int kGlueThreads = 256;

struct SgdRow {          // one row of the device table (4 x int64)
    float* param;
    float* grad;
    float* momentum;     // NULL: no momentum buffer (momentum == 0)
    long long numel;
};
static_assert(sizeof(SgdRow) == 32, "the table is an int64 [T][4] array on the Python side");

__device__ __forceinline__ double block_sum(double v) {        // fixed tree: wave shuffles, then the waves' sums in index order
    __shared__ double wave_sum[kGlueThreads / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if ((threadIdx.x & 63) == 0) wave_sum[threadIdx.x >> 6] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kGlueThreads / 64; ++w) t += wave_sum[w];
    }
    return t;           // valid in thread 0
}

__global__ __launch_bounds__(kGlueThreads) void multi_sumsq_kernel(const SgdRow* __restrict__ table, const int2* __restrict__ chunks,
                                                                   int chunk_elems, double* __restrict__ partial) {
    const int2 ck = chunks[blockIdx.x];
    const SgdRow row = table[ck.x];
    const long long base = (long long)ck.y * chunk_elems;
    long long left = row.numel - base;
    const int n = left > chunk_elems ? chunk_elems : (int)left;
    const float* g = row.grad + base;
    double acc = 0.0;
    if ((reinterpret_cast<uintptr_t>(g) & 15u) == 0) {
        const int n4 = n >> 2;
        const float4* g4 = reinterpret_cast<const float4*>(g);
#pragma unroll 4
        for (int i = threadIdx.x; i < n4; i += kGlueThreads) {
            const float4 v = g4[i];
            acc = fma((double)v.x, (double)v.x, acc);
            acc = fma((double)v.y, (double)v.y, acc);
            acc = fma((double)v.z, (double)v.z, acc);
            acc = fma((double)v.w, (double)v.w, acc);
        }
        for (int i = (n4 << 2) + threadIdx.x; i < n; i += kGlueThreads) acc = fma((double)g[i], (double)g[i], acc);
    } else {
        for (int i = threadIdx.x; i < n; i += kGlueThreads) acc = fma((double)g[i], (double)g[i], acc);
    }
    const double t = block_sum(acc);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

// out[0] = total gradient norm, out[1] = clip factor — torch.nn.utils.clip_grad_norm_'s float32 arithmetic on the norm
__global__ __launch_bounds__(kGlueThreads) void clip_coef_kernel(const double* __restrict__ partial, int n, float max_norm,
                                                                 float* __restrict__ out) {
    double acc = 0.0;
    for (int i = threadIdx.x; i < n; i += kGlueThreads) acc += partial[i];
    const double t = block_sum(acc);
    if (threadIdx.x == 0) {
        const float norm = (float)sqrt(t);
        float coef = max_norm / (norm + 1e-6f);
        if (coef > 1.0f) coef = 1.0f;          // (a NaN norm gives a NaN factor, as in torch: clamp keeps NaN)
        out[0] = norm;
        out[1] = coef;
    }
}

struct SgdArgs {
    float lr, momentum, weight_decay;
    int nesterov;
};

__device__ __forceinline__ void sgd_element(float& p, float& g, float& m, bool has_m, float coef, bool clip, const SgdArgs& a) {
    if (clip) g = g * coef;                              // what clip_grad_norm_ leaves in .grad
    float d = g;
    if (a.weight_decay != 0.f) d = fmaf(a.weight_decay, p, d);
    if (has_m) {
        m = m * a.momentum;
        m = m + d;
        d = a.nesterov ? fmaf(a.momentum, m, d) : m;
    }
    p = fmaf(-a.lr, d, p);
}

__global__ __launch_bounds__(kGlueThreads) void clip_sgd_kernel(const SgdRow* __restrict__ table, const int2* __restrict__ chunks,
                                                                int chunk_elems, const float* __restrict__ norm_coef,
                                                                const float* __restrict__ lr_dev, SgdArgs a) {
    const int2 ck = chunks[blockIdx.x];
    const SgdRow row = table[ck.x];
    const long long base = (long long)ck.y * chunk_elems;
    long long left = row.numel - base;
    const int n = left > chunk_elems ? chunk_elems : (int)left;
    float* p = row.param + base;
    float* g = row.grad + base;
    float* m = row.momentum ? row.momentum + base : nullptr;
    const bool has_m = m != nullptr && a.momentum != 0.f;
    const bool clip = norm_coef != nullptr;
    const float coef = clip ? norm_coef[1] : 1.f;
    if (lr_dev) a.lr = *lr_dev;
    const uintptr_t bits = reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m);
    int done = 0;
    if ((bits & 15u) == 0) {
        const int n4 = n >> 2;
        float4* p4 = reinterpret_cast<float4*>(p);
        float4* g4 = reinterpret_cast<float4*>(g);
        float4* m4 = reinterpret_cast<float4*>(m);
#pragma unroll 2
        for (int i = threadIdx.x; i < n4; i += kGlueThreads) {
            float4 pv = p4[i], gv = g4[i], mv = has_m ? m4[i] : make_float4(0.f, 0.f, 0.f, 0.f);
            sgd_element(pv.x, gv.x, mv.x, has_m, coef, clip, a);
            sgd_element(pv.y, gv.y, mv.y, has_m, coef, clip, a);
            sgd_element(pv.z, gv.z, mv.z, has_m, coef, clip, a);
            sgd_element(pv.w, gv.w, mv.w, has_m, coef, clip, a);
            p4[i] = pv;
            if (has_m) m4[i] = mv;
            if (clip) g4[i] = gv;
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < n; i += kGlueThreads) {
        float pv = p[i], gv = g[i], mv = has_m ? m[i] : 0.f;
        sgd_element(pv, gv, mv, has_m, coef, clip, a);
        p[i] = pv;
        if (has_m) m[i] = mv;
        if (clip) g[i] = gv;
    }
}

// nextou_device_write_i64: host values -> device memory as KERNEL ARGUMENTS (3.5 KB per launch).  A table rebuilt while a
// hipGraph is being captured cannot come through a host-to-device copy (pageable sources are not capturable, and a pinned
// source is re-read at every replay — it would have to stay alive and unchanged); launch arguments are copied into the graph
// node, so the replayed step rewrites exactly the table it was captured with.
constexpr int kPokeWords = 448;
struct PokeArgs {
    long long v[kPokeWords];
};
__global__ __launch_bounds__(512) void poke_kernel(long long* __restrict__ dst, PokeArgs a, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = a.v[threadIdx.x];
}

int check_lists(const char* who, const void* table, const void* chunks, int n_tensors, int n_chunks, int chunk_elems) {
    NEXTOU_REQUIRE(table && chunks, "%s: null pointer", who);
    NEXTOU_REQUIRE(n_tensors > 0 && n_chunks >= n_tensors && chunk_elems >= 1024 && chunk_elems % 4 == 0 && chunk_elems <= (1 << 24),
                   "%s: bad sizes (tensors %d, chunks %d, chunk %d elements)", who, n_tensors, n_chunks, chunk_elems);
    return 0;
}

}  // namespace
}  // namespace nextou

using namespace nextou;

extern "C" int nextou_grad_norm_clip_coef(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                          int64_t total_elems, double* partial, float max_norm, float* norm_coef,
                                          nextou_stream_t stream) {
    if (int e = check_lists("grad_norm_clip_coef", table, chunks, n_tensors, n_chunks, chunk_elems)) return e;
    NEXTOU_REQUIRE(partial && norm_coef && max_norm > 0.f && total_elems > 0, "grad_norm_clip_coef: null pointer or max_norm <= 0");
    hipStream_t s = (hipStream_t)stream;
    {
        ProfScope prof(s, kBoundHbm, 4.0 * (double)total_elems, "multi_sumsq_kernel[T%d chunks %d]", n_tensors, n_chunks);
        hipLaunchKernelGGL(multi_sumsq_kernel, dim3(n_chunks), dim3(kGlueThreads), 0, s, reinterpret_cast<const SgdRow*>(table),
                           reinterpret_cast<const int2*>(chunks), chunk_elems, partial);
    }
    hipLaunchKernelGGL(clip_coef_kernel, dim3(1), dim3(kGlueThreads), 0, s, partial, n_chunks, max_norm, norm_coef);
    return check_launch("grad_norm_clip_coef");
}

extern "C" int nextou_clip_sgd_update(const int64_t* table, int n_tensors, const int32_t* chunks, int n_chunks, int chunk_elems,
                                      int64_t total_elems, const float* norm_coef, float lr, const float* lr_dev, float momentum,
                                      float weight_decay, int nesterov, nextou_stream_t stream) {
    if (int e = check_lists("clip_sgd_update", table, chunks, n_tensors, n_chunks, chunk_elems)) return e;
    NEXTOU_REQUIRE(momentum >= 0.f && weight_decay >= 0.f && (!nesterov || momentum > 0.f) && total_elems > 0,
                   "clip_sgd_update: momentum %g, weight_decay %g, nesterov %d", momentum, weight_decay, nesterov);
    hipStream_t s = (hipStream_t)stream;
    SgdArgs a{lr, momentum, weight_decay, nesterov};
    const double per = 8.0 + (momentum != 0.f ? 8.0 : 0.0) + (norm_coef ? 8.0 : 4.0);   // p rw, buf rw, g r (+ w when clipped)
    ProfScope prof(s, kBoundHbm, per * (double)total_elems, "clip_sgd_kernel<%s>[T%d chunks %d]", norm_coef ? "clip" : "plain", n_tensors,
                   n_chunks);
    hipLaunchKernelGGL(clip_sgd_kernel, dim3(n_chunks), dim3(kGlueThreads), 0, s, reinterpret_cast<const SgdRow*>(table),
                       reinterpret_cast<const int2*>(chunks), chunk_elems, norm_coef, lr_dev, a);
    return check_launch("clip_sgd_kernel");
}

extern "C" int nextou_device_write_i64(int64_t* dst, const int64_t* host_values, int64_t n, nextou_stream_t stream) {
    NEXTOU_REQUIRE(dst && host_values && n > 0 && n <= (1ll << 24), "device_write_i64: null pointer or n=%lld", (long long)n);
    hipStream_t s = (hipStream_t)stream;
    for (int64_t off = 0; off < n; off += kPokeWords) {
        PokeArgs a;
        const int cnt = (int)((n - off) < kPokeWords ? (n - off) : kPokeWords);
        for (int i = 0; i < cnt; ++i) a.v[i] = host_values[off + i];
        for (int i = cnt; i < kPokeWords; ++i) a.v[i] = 0;
        hipLaunchKernelGGL(poke_kernel, dim3(1), dim3(512), 0, s, reinterpret_cast<long long*>(dst) + off, a, cnt);
    }
    return check_launch("device_write_i64");
}
