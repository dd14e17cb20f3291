// K6  Batch / instance normalisation fused with LeakyReLU, training and inference, forward and
// backward — the (norm -> nonlin) tail of every conv block on the NexToU path:
//   reference torch_nn.py:84-90 (BasicConv: conv -> norm -> act), NexToU_Encoder_Decoder.py:384-390
//   (FFN: conv+BN -> act -> conv+BN), :710-720 / :833-842 (fc1 / fc2: conv+BN) and the
//   ConvDropoutNormReLU blocks of the plain conv stages (:125-136, :281-298).
// PyTorch-ROCm runs these as MIOpenBatchNorm{Fwd,Bwd}Spatial + leaky_relu{,_backward}: 38.8 ms + 5.6 ms of
// the 327 ms cfg-2 step, 5-7x above what the bytes cost (profiles/r01_cfg2_step_kernel_trace_final.md).
//
// Bound: HBM.  Layout (B, C, S) with S contiguous (NCDHW).  A channel's B*S values are cut into tiles
// of rows (samples) x column ranges; a workgroup owns one tile of one channel, so scale/shift are
// wave-uniform scalars and every access is a 16-byte load/store of consecutive addresses.
//   forward : stats kernel (1 read; per-tile sum / sum-of-squares in float64) -> apply kernel
//             (finalises the channel from the tile partials in a fixed order, 1 read + 1 write).
//   backward: reduce kernel (reads x, gy: sum dz, sum dz*xhat in float64) -> apply kernel
//             (reads x, gy, writes gx).  dz = gy * (z > 0 ? 1 : slope) with z recomputed from x, so the
//             activation's input is never stored and the LeakyReLU passes disappear.
// The second kernel of each pair walks the channels in reverse launch order: what the first kernel
// read last is what it finds in the 256 MB Infinity Cache first.
// Sums are float64 and combined in a fixed order: results are bit-reproducible run to run.
#include "common.h"
#include <cstdlib>
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>

namespace nextou {

constexpr int kThreads = 256;

struct TilePlan {
    int vec;         // elements per 16-byte access (1 = scalar path)
    long long cols;  // S / vec
    int row_len, row_tiles;
    long long col_len;
    int col_tiles;
    int tw_log2;     // threads along the columns = 1 << tw_log2
    int tiles;       // per channel
};

// ~4096 workgroups over the whole tensor, at least 4 vectors per thread.
static TilePlan plan_tiles(int B, int C, long long S, int vec_full, bool aligned) {
    TilePlan p;
    p.vec = (aligned && S % vec_full == 0) ? vec_full : 1;
    p.cols = S / p.vec;
    const long long total = (long long)B * p.cols;
    long long per_channel = cdiv(4096, C);
    if (per_channel > 1024) per_channel = 1024;
    long long work = cdiv64(total, per_channel);
    if (work < 1024) work = 1024;
    if (p.cols >= work) {
        p.col_len = cdiv64(work, kThreads) * kThreads;
        p.col_tiles = (int)cdiv64(p.cols, p.col_len);
        p.row_len = 1;
        p.row_tiles = B;
    } else {
        p.col_len = p.cols;
        p.col_tiles = 1;
        p.row_len = (int)(work / p.cols);
        if (p.row_len < 1) p.row_len = 1;
        p.row_tiles = cdiv(B, p.row_len);
    }
    while ((long long)p.row_tiles * p.col_tiles > 1024) {  // the finalising wave reads <= 1024 partials
        if (p.col_tiles > 1) { p.col_len *= 2; p.col_tiles = (int)cdiv64(p.cols, p.col_len); }
        else { p.row_len *= 2; p.row_tiles = cdiv(B, p.row_len); }
    }
    const long long span = p.col_len < p.cols ? p.col_len : p.cols;
    p.tw_log2 = 0;
    while (p.tw_log2 < 8 && (1ll << p.tw_log2) < span) ++p.tw_log2;
    p.tiles = p.row_tiles * p.col_tiles;
    return p;
}

template <typename T, int VEC> struct Pack;
// NEXTOU_K6_NT (A/B: tools/k6_nt_ab.sh, profiles/r02_k6_nontemporal_ab.md — faster stand-alone, SLOWER inside the step, so off): 1 = non-temporal stores,
// 2 = non-temporal loads, 3 = both, for the fp32
// 16-byte accesses — every K6 tensor is streamed once per kernel, nothing is reused from L2.
#ifndef NEXTOU_K6_NT
#define NEXTOU_K6_NT 0
#endif
#ifndef NEXTOU_K6_NT_REDUCE
#define NEXTOU_K6_NT_REDUCE 0
#endif
using f32x4_t = __attribute__((ext_vector_type(4))) float;
template <> struct Pack<float, 4> {
    float v[4];
    __device__ void load(const float* p) {
#if NEXTOU_K6_NT & 2
        const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
#else
        const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
#endif
    }
    // the reductions (statistics, backward sums) read every element exactly once and write nothing: NEXTOU_K6_NT_REDUCE=1
    // makes those loads non-temporal (A/B: profiles/r02_k6_nontemporal_ab.md)
    __device__ void load_stream(const float* p) {
#if NEXTOU_K6_NT_REDUCE
        const f32x4_t t = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
        v[0] = t[0]; v[1] = t[1]; v[2] = t[2]; v[3] = t[3];
#else
        load(p);
#endif
    }
    __device__ void store(float* p) const {
#if NEXTOU_K6_NT & 1
        f32x4_t t; t[0] = v[0]; t[1] = v[1]; t[2] = v[2]; t[3] = v[3];
        __builtin_nontemporal_store(t, reinterpret_cast<f32x4_t*>(p));
#else
        *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
#endif
    }
};
template <> struct Pack<float, 1> {
    float v[1];
    __device__ void load_stream(const float* p) { load(p); }
    __device__ void load(const float* p) { v[0] = *p; }
    __device__ void store(float* p) const { *p = v[0]; }
};
__device__ inline float bf16_to_f32(unsigned short h) { return __uint_as_float((unsigned)h << 16); }
__device__ inline unsigned short f32_to_bf16(float f) {  // round to nearest even, NaN kept quiet
    unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
template <> struct Pack<__hip_bfloat16, 8> {
    float v[8];
    __device__ void load_stream(const __hip_bfloat16* p) { load(p); }
    __device__ void load(const __hip_bfloat16* p) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { v[2 * i] = __uint_as_float(w[i] << 16); v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u); }
    }
    __device__ void store(__hip_bfloat16* p) const {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = (unsigned)f32_to_bf16(v[2 * i]) | ((unsigned)f32_to_bf16(v[2 * i + 1]) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Pack<__hip_bfloat16, 1> {
    float v[1];
    __device__ void load_stream(const __hip_bfloat16* p) { load(p); }
    __device__ void load(const __hip_bfloat16* p) { v[0] = bf16_to_f32(*reinterpret_cast<const unsigned short*>(p)); }
    __device__ void store(__hip_bfloat16* p) const { *reinterpret_cast<unsigned short*>(p) = f32_to_bf16(v[0]); }
};

// fp16 (nnU-Net's default autocast dtype): v_cvt_f32_f16 / v_cvt_f16_f32 (round to nearest even)
template <> struct Pack<__half, 8> {
    float v[8];
    __device__ void load_stream(const __half* p) { load(p); }
    __device__ void load(const __half* p) {
        const uint4 t = *reinterpret_cast<const uint4*>(p);
        const unsigned w[4] = {t.x, t.y, t.z, t.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            v[2 * i] = __half2float(__ushort_as_half((unsigned short)(w[i] & 0xffffu)));
            v[2 * i + 1] = __half2float(__ushort_as_half((unsigned short)(w[i] >> 16)));
        }
    }
    __device__ void store(__half* p) const {
        unsigned w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            w[i] = (unsigned)__half_as_ushort(__float2half_rn(v[2 * i])) | ((unsigned)__half_as_ushort(__float2half_rn(v[2 * i + 1])) << 16);
        *reinterpret_cast<uint4*>(p) = make_uint4(w[0], w[1], w[2], w[3]);
    }
};
template <> struct Pack<__half, 1> {
    float v[1];
    __device__ void load_stream(const __half* p) { load(p); }
    __device__ void load(const __half* p) { v[0] = __half2float(*p); }
    __device__ void store(__half* p) const { *p = __float2half_rn(v[0]); }
};

struct Tile {
    int r0, r1, ty, trows;
    long long c0, c1, tx, tcols;
};
__device__ inline Tile decode_tile(int B, long long cols, int row_len, long long col_len, int col_tiles, int tw_log2) {
    Tile t;
    const int rt = blockIdx.x / col_tiles, ct = blockIdx.x - rt * col_tiles;
    t.r0 = rt * row_len;
    t.r1 = min(B, t.r0 + row_len);
    t.c0 = (long long)ct * col_len;
    t.c1 = min(cols, t.c0 + col_len);
    t.tcols = 1ll << tw_log2;
    t.trows = kThreads >> tw_log2;
    t.tx = threadIdx.x & (t.tcols - 1);
    t.ty = threadIdx.x >> tw_log2;
    return t;
}

__device__ inline double norm_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Workgroup sum of two doubles; the result is valid in thread 0.
__device__ inline void block_sum2(double& a, double& b) {
    __shared__ double red[2][kThreads / 64];
    a = norm_wave_sum(a);
    b = norm_wave_sum(b);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) { red[0][w] = a; red[1][w] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        a = red[0][0]; b = red[1][0];
        for (int i = 1; i < kThreads / 64; ++i) { a += red[0][i]; b += red[1][i]; }
    }
}

// Every wave adds the channel's tile partials in the same fixed order (lane-strided, then a butterfly).
__device__ inline void channel_sums(const double2* partial, int tiles, double& a, double& b) {
    const int lane = threadIdx.x & 63;
    a = 0.0; b = 0.0;
    for (int i = lane; i < tiles; i += 64) { const double2 p = partial[i]; a += p.x; b += p.y; }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
}

// Workgroup version for the per-channel finalize launches (kFinThreads threads, one workgroup per channel): thread t adds
// partials t, t + kFinThreads, ... with four loads in flight, then a fixed-shape combine (butterfly per wave, waves in
// index order).  The single-wave loop above walks `tiles / 64` dependent L2 round trips per lane — 128 of them at 8192
// workgroups, which cost more than the finer grid gained (profiles/r01_kernel_bench_norm_cl_blocks.txt).
constexpr int kFinThreads = 256;
__device__ inline void channel_sums_block(const double2* partial, int tiles, double& a, double& b) {
    __shared__ double fin_red[2][kFinThreads / 64];
    double s0 = 0.0, q0 = 0.0, s1 = 0.0, q1 = 0.0, s2 = 0.0, q2 = 0.0, s3 = 0.0, q3 = 0.0;
    int i = threadIdx.x;
    for (; i + 3 * kFinThreads < tiles; i += 4 * kFinThreads) {
        const double2 p0 = partial[i], p1 = partial[i + kFinThreads], p2 = partial[i + 2 * kFinThreads],
                      p3 = partial[i + 3 * kFinThreads];
        s0 += p0.x; q0 += p0.y; s1 += p1.x; q1 += p1.y; s2 += p2.x; q2 += p2.y; s3 += p3.x; q3 += p3.y;
    }
    for (; i < tiles; i += kFinThreads) { const double2 p = partial[i]; s0 += p.x; q0 += p.y; }
    a = (s0 + s1) + (s2 + s3);
    b = (q0 + q1) + (q2 + q3);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { a += __shfl_xor(a, off, 64); b += __shfl_xor(b, off, 64); }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { fin_red[0][w] = a; fin_red[1][w] = b; }
    __syncthreads();
    a = fin_red[0][0]; b = fin_red[1][0];
#pragma unroll
    for (int k = 1; k < kFinThreads / 64; ++k) { a += fin_red[0][k]; b += fin_red[1][k]; }
}

__device__ inline float leaky(float z, float slope) { return z > 0.f ? z : z * slope; }

// ------------------------------------------------------------------------------------------------------
template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_stats_kernel(const T* __restrict__ x, double2* __restrict__ partial,
                                                            int B, int C, long long cols, int row_len,
                                                            long long col_len, int col_tiles, int tw_log2) {
    const int c = blockIdx.y;
    const Tile t = decode_tile(B, cols, row_len, col_len, col_tiles, tw_log2);
    double s = 0.0, q = 0.0;
    for (int r = t.r0 + t.ty; r < t.r1; r += t.trows) {
        const T* row = x + ((long long)r * C + c) * cols * VEC;
        long long col = t.c0 + t.tx;
        for (; col + 3 * t.tcols < t.c1; col += 4 * t.tcols) {  // four independent 16-byte loads in flight
            Pack<T, VEC> p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u].load(row + (col + u * t.tcols) * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) { const double v = (double)p[u].v[i]; s += v; q = fma(v, v, q); }
        }
        for (; col < t.c1; col += t.tcols) {
            Pack<T, VEC> p;
            p.load(row + col * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) { const double v = (double)p.v[i]; s += v; q = fma(v, v, q); }
        }
    }
    block_sum2(s, q);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.x + blockIdx.x] = make_double2(s, q);
}

struct ChannelAffine { float scale, shift, mean, invstd; };

// training: batch statistics from the tile partials; inference: the running statistics.
// pre_bias: the bias of the convolution that produced x, folded in here instead of being added to x
// (x + b has the same batch-normalised value as x; only the running mean and the inference shift see b).
__device__ inline ChannelAffine channel_affine(const double2* partial, int tiles, int c, int wmod, double count,
                                               const float* weight, const float* bias, const float* pre_bias,
                                               const float* running_mean, const float* running_var, int training,
                                               float eps, double* var_out) {
    ChannelAffine a;
    const int pc = wmod > 0 ? c % wmod : c;
    if (training) {
        double s, q;
        channel_sums(partial + (size_t)c * tiles, tiles, s, q);
        const double mean = s / count;
        double var = q / count - mean * mean;
        if (var < 0.0) var = 0.0;
        a.mean = (float)mean;
        a.invstd = (float)(1.0 / sqrt(var + (double)eps));
        *var_out = var;
    } else {
        a.mean = running_mean[c] - (pre_bias ? pre_bias[pc] : 0.f);
        a.invstd = 1.0f / sqrtf(running_var[c] + eps);
        *var_out = 0.0;
    }
    const float w = weight ? weight[pc] : 1.f, b = bias ? bias[pc] : 0.f;
    a.scale = w * a.invstd;
    a.shift = fmaf(-a.mean, a.scale, b);
    return a;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                            const double2* __restrict__ partial,
                                                            const float* __restrict__ weight, const float* __restrict__ bias,
                                                            const float* __restrict__ pre_bias,
                                                            float* running_mean, float* running_var,
                                                            float* __restrict__ save_mean, float* __restrict__ save_invstd,
                                                            int B, int C, long long cols, int row_len, long long col_len,
                                                            int col_tiles, int tw_log2, int wmod, double count,
                                                            int training, float momentum, float eps, float slope, int n_partial) {
    const int c = C - 1 - blockIdx.y;
    double var;
    // n_partial > 0: the statistics partials came from another kernel (mr_grp_cm_kernel's epilogue), n_partial per channel
    const ChannelAffine a = channel_affine(partial, n_partial > 0 ? n_partial : (int)gridDim.x, c, wmod, count, weight, bias, pre_bias, running_mean,
                                           running_var, training, eps, &var);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (save_mean) save_mean[c] = a.mean;
        if (save_invstd) save_invstd[c] = a.invstd;
        if (training && running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            const double batch_mean = (double)a.mean + (pre_bias ? (double)pre_bias[wmod > 0 ? c % wmod : c] : 0.0);
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * batch_mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    }
    const Tile t = decode_tile(B, cols, row_len, col_len, col_tiles, tw_log2);
    for (int r = t.r0 + t.ty; r < t.r1; r += t.trows) {
        const size_t base = ((size_t)r * C + c) * cols * VEC;
        const T* row = x + base;
        T* out = y + base;
        long long col = t.c0 + t.tx;
        for (; col + 3 * t.tcols < t.c1; col += 4 * t.tcols) {
            Pack<T, VEC> p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u].load(row + (col + u * t.tcols) * VEC);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) p[u].v[i] = leaky(fmaf(p[u].v[i], a.scale, a.shift), slope);
                p[u].store(out + (col + u * t.tcols) * VEC);
            }
        }
        for (; col < t.c1; col += t.tcols) {
            Pack<T, VEC> p;
            p.load(row + col * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) p.v[i] = leaky(fmaf(p.v[i], a.scale, a.shift), slope);
            p.store(out + col * VEC);
        }
    }
}

// ------------------------------------------------------------------------------------------------------
struct BwdAffine { float scale, shift, mean, invstd; };
__device__ inline BwdAffine bwd_affine(int c, int wmod, const float* weight, const float* bias, const float* save_mean,
                                       const float* save_invstd) {
    BwdAffine a;
    a.mean = save_mean[c];
    a.invstd = save_invstd[c];
    const int pc = wmod > 0 ? c % wmod : c;
    const float w = weight ? weight[pc] : 1.f, b = bias ? bias[pc] : 0.f;
    a.scale = w * a.invstd;
    a.shift = fmaf(-a.mean, a.scale, b);  // identical to the forward's expression: same sign of z
    return a;
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                 double2* __restrict__ partial,
                                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                                 const float* __restrict__ save_mean,
                                                                 const float* __restrict__ save_invstd, int B, int C,
                                                                 long long cols, int row_len, long long col_len,
                                                                 int col_tiles, int tw_log2, int wmod, float slope) {
    const int c = blockIdx.y;
    const BwdAffine a = bwd_affine(c, wmod, weight, bias, save_mean, save_invstd);
    const Tile t = decode_tile(B, cols, row_len, col_len, col_tiles, tw_log2);
    double s1 = 0.0, s2 = 0.0;
    for (int r = t.r0 + t.ty; r < t.r1; r += t.trows) {
        const size_t base = ((size_t)r * C + c) * cols * VEC;
        const T* row = x + base;
        const T* grow = gy + base;
        long long col = t.c0 + t.tx;
        for (; col + t.tcols < t.c1; col += 2 * t.tcols) {
            Pack<T, VEC> p[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { p[u].load(row + (col + u * t.tcols) * VEC); g[u].load(grow + (col + u * t.tcols) * VEC); }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float z = fmaf(p[u].v[i], a.scale, a.shift);
                    const float dz = z > 0.f ? g[u].v[i] : g[u].v[i] * slope;
                    const float xh = (p[u].v[i] - a.mean) * a.invstd;
                    s1 += (double)dz;
                    s2 = fma((double)dz, (double)xh, s2);
                }
        }
        for (; col < t.c1; col += t.tcols) {
            Pack<T, VEC> p, g;
            p.load(row + col * VEC);
            g.load(grow + col * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = fmaf(p.v[i], a.scale, a.shift);
                const float dz = z > 0.f ? g.v[i] : g.v[i] * slope;
                const float xh = (p.v[i] - a.mean) * a.invstd;
                s1 += (double)dz;
                s2 = fma((double)dz, (double)xh, s2);
            }
        }
    }
    block_sum2(s1, s2);
    if (threadIdx.x == 0) partial[(size_t)c * gridDim.x + blockIdx.x] = make_double2(s1, s2);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                T* __restrict__ gx, const double2* __restrict__ partial,
                                                                const float* __restrict__ weight, const float* __restrict__ bias,
                                                                const float* __restrict__ save_mean,
                                                                const float* __restrict__ save_invstd,
                                                                float* __restrict__ gweight, float* __restrict__ gbias, int B,
                                                                int C, long long cols, int row_len, long long col_len,
                                                                int col_tiles, int tw_log2, int wmod, double count,
                                                                int training, float slope) {
    const int c = C - 1 - blockIdx.y;
    const BwdAffine a = bwd_affine(c, wmod, weight, bias, save_mean, save_invstd);
    double s1, s2;
    channel_sums(partial + (size_t)c * gridDim.x, gridDim.x, s1, s2);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (gweight) gweight[c] = (float)s2;
        if (gbias) gbias[c] = (float)s1;
    }
    // training: gx = scale * (dz - mean(dz) - xhat * mean(dz * xhat));  inference: gx = scale * dz
    const float k1 = training ? (float)(s1 / count) : 0.f;
    const float k2 = training ? (float)(s2 / count) : 0.f;
    const Tile t = decode_tile(B, cols, row_len, col_len, col_tiles, tw_log2);
    for (int r = t.r0 + t.ty; r < t.r1; r += t.trows) {
        const size_t base = ((size_t)r * C + c) * cols * VEC;
        const T* row = x + base;
        const T* grow = gy + base;
        T* out = gx + base;
        long long col = t.c0 + t.tx;
        for (; col + t.tcols < t.c1; col += 2 * t.tcols) {
            Pack<T, VEC> p[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { p[u].load(row + (col + u * t.tcols) * VEC); g[u].load(grow + (col + u * t.tcols) * VEC); }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) {
                    const float z = fmaf(p[u].v[i], a.scale, a.shift);
                    const float dz = z > 0.f ? g[u].v[i] : g[u].v[i] * slope;
                    const float xh = (p[u].v[i] - a.mean) * a.invstd;
                    g[u].v[i] = a.scale * ((dz - k1) - xh * k2);
                }
                g[u].store(out + (col + u * t.tcols) * VEC);
            }
        }
        for (; col < t.c1; col += t.tcols) {
            Pack<T, VEC> p, g;
            p.load(row + col * VEC);
            g.load(grow + col * VEC);
#pragma unroll
            for (int i = 0; i < VEC; ++i) {
                const float z = fmaf(p.v[i], a.scale, a.shift);
                const float dz = z > 0.f ? g.v[i] : g.v[i] * slope;
                const float xh = (p.v[i] - a.mean) * a.invstd;
                g.v[i] = a.scale * ((dz - k1) - xh * k2);
            }
            g.store(out + col * VEC);
        }
    }
}


// ======================================================================================================
// Channels-last (N,D,H,W,C) variants — the layout MIOpen's CK convolutions run in natively, used for the
// full-resolution stages where NCDHW costs a transpose of every activation around every convolution
// (tools/conv_probe.py: 33 -> 33 channels at 64x224x192: 6.05 ms forward in NCDHW, 3.33 ms in NDHWC).
// x is R = B*S rows of C contiguous channels.  q = 256 / C rows are covered by the q*C active threads of a workgroup
// with one 16-byte access each, so a thread owns VEC fixed channels ((t*VEC + j) mod C: every block span and the
// per-iteration stride are multiples of C) and keeps their sums in registers; the q*VEC partials of each channel meet
// in LDS in a fixed order.  Per-channel finalisation runs in its own C-workgroup launch between the two passes.
// ======================================================================================================
// Workgroups per launch: as many as give each one >= kClItersPerBlock passes over its span, at most kClMaxBlocks.
// Measured on MI355X with wall-clock timing of forward + backward (profiles/r01_kernel_bench_norm_cl_blocks.txt): the
// 727 MB stage-0 tensor wants the finest grid (1137 us at 2048 workgroups, 1090 us at 8192 — 2048 x 4 waves is exactly one
// residency of the chip and leaves a drain tail), the 91 MB stage-2 tensor the coarsest (143 us at 2048, 164 us at 8192:
// five passes per workgroup are too few).  ~20 passes per workgroup reproduces the best point of all three shapes.
// That only pays since the per-channel finalisation is parallel over the partials (channel_sums_block); with the
// single-wave finalize the finer grid was a net loss.  NEXTOU_CL_BLOCKS pins the count for experiments.
constexpr int kClMaxBlocks = 8192;
constexpr int kClItersPerBlock = 20;

static int cl_pinned_blocks() {      // 0 = not pinned
    static const int v = [] {
        const char* e = getenv("NEXTOU_CL_BLOCKS");
        int n = e ? atoi(e) : 0;
        if (n < 0) n = 0;
        if (n > kClMaxBlocks) n = kClMaxBlocks;
        return n;
    }();
    return v;
}

struct ClPlan {
    int vec, tact, blocks;
    long long span;  // elements per workgroup (multiple of tact * vec)
};

static ClPlan plan_cl(long long total, int C, int vec_full, bool aligned) {
    ClPlan p;
    p.vec = (aligned && total % vec_full == 0) ? vec_full : 1;
    p.tact = (kThreads / C) * C;
    const long long per_iter = (long long)p.tact * p.vec;
    long long iters;
    if (cl_pinned_blocks() > 0) {
        iters = cdiv64(total, (long long)cl_pinned_blocks() * per_iter);
        if (iters < 4) iters = 4;
    } else {
        iters = kClItersPerBlock;
        const long long floor_iters = cdiv64(total, (long long)kClMaxBlocks * per_iter);   // never more than kClMaxBlocks
        if (iters < floor_iters) iters = floor_iters;
    }
    p.span = iters * per_iter;
    p.blocks = (int)cdiv64(total, p.span);
    return p;
}

template <int VEC>
__device__ inline void cl_channels(int C, int (&ch)[VEC]) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) ch[j] = (threadIdx.x * VEC + j) % C;
}

// LDS meeting point: thread c < C adds the q*VEC partials of channel c in index order.
template <int VEC>
__device__ inline void cl_block_sums(const double (&a)[VEC], const double (&b)[VEC], int C, int tact, double2* partial,
                                     int nblocks) {
    extern __shared__ double2 cl_red[];
    if ((int)threadIdx.x < tact) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) cl_red[threadIdx.x * VEC + j] = make_double2(a[j], b[j]);
    }
    __syncthreads();
    if ((int)threadIdx.x < C) {
        double s = 0.0, q = 0.0;
        for (int i = threadIdx.x; i < tact * VEC; i += C) { s += cl_red[i].x; q += cl_red[i].y; }
        partial[(size_t)threadIdx.x * nblocks + blockIdx.x] = make_double2(s, q);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_cl_stats_kernel(const T* __restrict__ x, double2* __restrict__ partial,
                                                               long long total, int C, int tact, long long span) {
    double s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.0;
    if ((int)threadIdx.x < tact) {
        const long long base = (long long)blockIdx.x * span;
        const long long end = min(total, base + span);
        const long long stride = (long long)tact * VEC;
        long long e = base + (long long)threadIdx.x * VEC;
        for (; e + 3 * stride < end; e += 4 * stride) {
            Pack<T, VEC> p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u].load_stream(x + e + u * stride);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const double v = (double)p[u].v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
        }
        for (; e < end; e += stride) {
            Pack<T, VEC> p;
            p.load_stream(x + e);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const double v = (double)p.v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
        }
    }
    cl_block_sums<VEC>(s, q, C, tact, partial, gridDim.x);
}

// bn_cl_stats_kernel's sums over a channel RANGE of wider channels-last rows, the range written out densely on the way: the
// backward of the decoder's cat((up-convolution output + bias, skip), 1) hands the first C of the gradient's C + C2 channels to the
// transposed convolution — which needs a dense tensor — and their per-channel sums to the bias.  ATen's narrow().contiguous()
// followed by nextou_channel_sum read the range twice; this reads it once.  Work split, accumulation order and partial layout are
// bn_cl_stats_kernel's on the dense (P, C) tensor, so the sums are bit-identical to nextou_channel_sum of the copy.
// C % 4 == 0, ld % 4 == 0, c_off % 4 == 0; span (and hence every workgroup's base) is a multiple of tact * 4, tact a multiple of C:
// a lane's column never changes and its row advances by (tact * 4) / C per iteration.
// SHUF: the dense copy is written "un-shuffled" instead — row p of the (B, D2, H2, W2) output volume of a transposed convolution whose
// kernel equals its stride (sd, sh, sw) belongs to input point p_in = (b, d2 / sd, h2 / sh, w2 / sw) and tap t = ((d2 % sd) sh + h2 % sh) sw +
// w2 % sw, and goes to dst[(p_in * T + t) * C + c]: the (P_in, T * C) matrix whose product with the filter is the data gradient.  Reads and
// sums are unchanged (same order), only the store address differs.
template <bool SHUF>
__global__ __launch_bounds__(kThreads) void narrow_copy_stats_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                     double2* __restrict__ partial, long long total, int C, int tact,
                                                                     long long span, long long ld, int c_off, UpShuffle u) {
    double s[4], q[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) s[j] = q[j] = 0.0;
    if ((int)threadIdx.x < tact) {
        const long long base = (long long)blockIdx.x * span;
        const long long end = min(total, base + span);
        const long long stride = (long long)tact * 4;
        const long long rows_per_iter = stride / C;
        long long e = base + (long long)threadIdx.x * 4;
        long long row = e / C;
        const int col = (int)(e % C);
        const float* sp = src + row * ld + c_off + col;
        const long long sstep = rows_per_iter * ld;
        for (; e + 3 * stride < end; e += 4 * stride, sp += 4 * sstep, row += 4 * rows_per_iter) {
            Pack<float, 4> p[4];
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) p[u4].load_stream(sp + u4 * sstep);
#pragma unroll
            for (int u4 = 0; u4 < 4; ++u4) {
                if (SHUF) p[u4].store(dst + upconv_row(row + u4 * rows_per_iter, u) * C + col);
                else p[u4].store(dst + e + u4 * stride);
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double v = (double)p[u4].v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
            }
        }
        for (; e < end; e += stride, sp += sstep, row += rows_per_iter) {
            Pack<float, 4> p;
            p.load_stream(sp);
            if (SHUF) p.store(dst + upconv_row(row, u) * C + col);
            else p.store(dst + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) { const double v = (double)p.v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
        }
    }
    cl_block_sums<4>(s, q, C, tact, partial, gridDim.x);
}

// One workgroup per channel: statistics -> save_mean / save_invstd (+ running statistics), shared by both layouts' callers.
__global__ __launch_bounds__(kFinThreads) void bn_finalize_kernel(const double2* __restrict__ partial, int tiles, double count,
                                                         const float* __restrict__ pre_bias, float* running_mean,
                                                         float* running_var, float* __restrict__ save_mean,
                                                         float* __restrict__ save_invstd, int training, float momentum,
                                                         float eps, const float* __restrict__ weight = nullptr,
                                                         const float* __restrict__ bias = nullptr, float* __restrict__ scale_out = nullptr,
                                                         float* __restrict__ shift_out = nullptr) {
    const int c = blockIdx.x;
    float mean, invstd;
    double var = 0.0;
    if (training) {
        double s, q;
        channel_sums_block(partial + (size_t)c * tiles, tiles, s, q);
        const double m = s / count;
        var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        mean = (float)m;
        invstd = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        mean = running_mean[c] - (pre_bias ? pre_bias[c] : 0.f);
        invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    if (threadIdx.x == 0) {
        save_mean[c] = mean;
        save_invstd[c] = invstd;
        if (scale_out) {            // the affine K6's apply kernels compute per thread, for consumers that normalise on operand load (K7)
            const float sc = (weight ? weight[c] : 1.f) * invstd;
            scale_out[c] = sc;
            shift_out[c] = fmaf(-mean, sc, bias ? bias[c] : 0.f);
        }
        if (training && running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            const double batch_mean = (double)mean + (pre_bias ? (double)pre_bias[c] : 0.0);
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * batch_mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_cl_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                               const float* __restrict__ weight, const float* __restrict__ bias,
                                                               const float* __restrict__ save_mean,
                                                               const float* __restrict__ save_invstd, long long total, int C,
                                                               int tact, long long span, float slope,
                                                               const T* __restrict__ residual = nullptr) {
    if ((int)threadIdx.x >= tact) return;
    int ch[VEC];
    cl_channels<VEC>(C, ch);
    float scale[VEC], shift[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float w = weight ? weight[ch[j]] : 1.f, b = bias ? bias[ch[j]] : 0.f;
        scale[j] = w * save_invstd[ch[j]];
        shift[j] = fmaf(-save_mean[ch[j]], scale[j], b);
    }
    const long long base = (long long)blockIdx.x * span;
    const long long end = min(total, base + span);
    const long long stride = (long long)tact * VEC;
    long long e = base + (long long)threadIdx.x * VEC;
    if (residual) {        // y = leaky(norm(x)) + residual: the block's shortcut add rides on the pass that writes y anyway
        for (; e < end; e += stride) {
            Pack<T, VEC> p, r;
            p.load(x + e);
            r.load(residual + e);
#pragma unroll
            for (int j = 0; j < VEC; ++j) p.v[j] = leaky(fmaf(p.v[j], scale[j], shift[j]), slope) + r.v[j];
            p.store(y + e);
        }
        return;
    }
    for (; e + 3 * stride < end; e += 4 * stride) {
        Pack<T, VEC> p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u].load(x + e + u * stride);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) p[u].v[j] = leaky(fmaf(p[u].v[j], scale[j], shift[j]), slope);
            p[u].store(y + e + u * stride);
        }
    }
    for (; e < end; e += stride) {
        Pack<T, VEC> p;
        p.load(x + e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) p.v[j] = leaky(fmaf(p.v[j], scale[j], shift[j]), slope);
        p.store(y + e);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_cl_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                    double2* __restrict__ partial,
                                                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                                                    const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd, long long total,
                                                                    int C, int tact, long long span, float slope) {
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0;
    if ((int)threadIdx.x < tact) {
        int ch[VEC];
        cl_channels<VEC>(C, ch);
        float scale[VEC], shift[VEC], mean[VEC], invstd[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float w = weight ? weight[ch[j]] : 1.f, b = bias ? bias[ch[j]] : 0.f;
            mean[j] = save_mean[ch[j]];
            invstd[j] = save_invstd[ch[j]];
            scale[j] = w * invstd[j];
            shift[j] = fmaf(-mean[j], scale[j], b);
        }
        const long long base = (long long)blockIdx.x * span;
        const long long end = min(total, base + span);
        const long long stride = (long long)tact * VEC;
        long long e = base + (long long)threadIdx.x * VEC;
        for (; e + stride < end; e += 2 * stride) {
            Pack<T, VEC> p[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { p[u].load_stream(x + e + u * stride); g[u].load_stream(gy + e + u * stride); }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float z = fmaf(p[u].v[j], scale[j], shift[j]);
                    const float dz = z > 0.f ? g[u].v[j] : g[u].v[j] * slope;
                    const float xh = (p[u].v[j] - mean[j]) * invstd[j];
                    s1[j] += (double)dz;
                    s2[j] = fma((double)dz, (double)xh, s2[j]);
                }
        }
        for (; e < end; e += stride) {
            Pack<T, VEC> p, g;
            p.load_stream(x + e);
            g.load_stream(gy + e);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float z = fmaf(p.v[j], scale[j], shift[j]);
                const float dz = z > 0.f ? g.v[j] : g.v[j] * slope;
                const float xh = (p.v[j] - mean[j]) * invstd[j];
                s1[j] += (double)dz;
                s2[j] = fma((double)dz, (double)xh, s2[j]);
            }
        }
    }
    cl_block_sums<VEC>(s1, s2, C, tact, partial, gridDim.x);
}

// One workgroup per channel: (sum dz, sum dz*xhat) -> the two projection coefficients + the parameter gradients.
__global__ __launch_bounds__(kFinThreads) void bn_bwd_finalize_kernel(const double2* __restrict__ partial, int tiles, double count,
                                                             float2* __restrict__ coeff, float* __restrict__ gweight,
                                                             float* __restrict__ gbias, int training) {
    const int c = blockIdx.x;
    double s1, s2;
    channel_sums_block(partial + (size_t)c * tiles, tiles, s1, s2);
    if (threadIdx.x == 0) {
        coeff[c] = training ? make_float2((float)(s1 / count), (float)(s2 / count)) : make_float2(0.f, 0.f);
        if (gweight) gweight[c] = (float)s2;
        if (gbias) gbias[c] = (float)s1;
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_cl_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                   T* __restrict__ gx, const float2* __restrict__ coeff,
                                                                   const float* __restrict__ weight, const float* __restrict__ bias,
                                                                   const float* __restrict__ save_mean,
                                                                   const float* __restrict__ save_invstd, long long total,
                                                                   int C, int tact, long long span, float slope) {
    if ((int)threadIdx.x >= tact) return;
    int ch[VEC];
    cl_channels<VEC>(C, ch);
    float scale[VEC], shift[VEC], mean[VEC], invstd[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float w = weight ? weight[ch[j]] : 1.f, b = bias ? bias[ch[j]] : 0.f;
        mean[j] = save_mean[ch[j]];
        invstd[j] = save_invstd[ch[j]];
        scale[j] = w * invstd[j];
        shift[j] = fmaf(-mean[j], scale[j], b);
        const float2 k = coeff[ch[j]];
        k1[j] = k.x;
        k2[j] = k.y;
    }
    const long long base = (long long)blockIdx.x * span;
    const long long end = min(total, base + span);
    const long long stride = (long long)tact * VEC;
    long long e = base + (long long)threadIdx.x * VEC;
    for (; e + stride < end; e += 2 * stride) {
        Pack<T, VEC> p[2], g[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { p[u].load(x + e + u * stride); g[u].load(gy + e + u * stride); }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float z = fmaf(p[u].v[j], scale[j], shift[j]);
                const float dz = z > 0.f ? g[u].v[j] : g[u].v[j] * slope;
                const float xh = (p[u].v[j] - mean[j]) * invstd[j];
                g[u].v[j] = scale[j] * ((dz - k1[j]) - xh * k2[j]);
            }
            g[u].store(gx + e + u * stride);
        }
    }
    for (; e < end; e += stride) {
        Pack<T, VEC> p, g;
        p.load(x + e);
        g.load(gy + e);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float z = fmaf(p.v[j], scale[j], shift[j]);
            const float dz = z > 0.f ? g.v[j] : g.v[j] * slope;
            const float xh = (p.v[j] - mean[j]) * invstd[j];
            g.v[j] = scale[j] * ((dz - k1[j]) - xh * k2[j]);
        }
        g.store(gx + e);
    }
}

// Per-channel sum as float (the bias gradient of a convolution that is not followed by a norm).
__global__ __launch_bounds__(kFinThreads) void channel_sum_finalize_kernel(const double2* __restrict__ partial, int tiles,
                                                                  float* __restrict__ out) {
    double s, q;
    channel_sums_block(partial + (size_t)blockIdx.x * tiles, tiles, s, q);
    if (threadIdx.x == 0) out[blockIdx.x] = (float)s;
}

static bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

struct NormArgs {
    const void *x, *gy;
    void *y, *gx;
    const float *weight, *bias, *pre_bias;
    float *running_mean, *running_var, *save_mean, *save_invstd, *gweight, *gbias;
    double2* partial;
    int B, C;
    long long S;
    int wmod, training;
    float momentum, eps, slope;
    int n_partial;      // > 0: `partial` holds that many ready-made partials per channel (NCDHW forward only): no statistics launch
};

template <typename T, int VEC>
void launch_fwd(const NormArgs& a, const TilePlan& p, hipStream_t s, const char* tname) {
    const dim3 grid(p.tiles, a.C);
    const double bytes = (double)a.B * a.C * (double)a.S * sizeof(T);
    const double count = (double)a.B * (double)a.S;
    if (a.training && a.n_partial == 0) {
        ProfScope prof(s, kBoundHbm, bytes, "bn_stats_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, a.partial, a.B, a.C, p.cols,
                           p.row_len, p.col_len, p.col_tiles, p.tw_log2);
    }
    ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (T*)a.y, a.partial, a.weight,
                       a.bias, a.pre_bias, a.running_mean, a.running_var, a.save_mean, a.save_invstd, a.B, a.C, p.cols, p.row_len,
                       p.col_len, p.col_tiles, p.tw_log2, a.wmod, count, a.training, a.momentum, a.eps, a.slope, a.n_partial);
}

template <typename T, int VEC>
void launch_bwd(const NormArgs& a, const TilePlan& p, hipStream_t s, const char* tname) {
    const dim3 grid(p.tiles, a.C);
    const double bytes = (double)a.B * a.C * (double)a.S * sizeof(T);
    const double count = (double)a.B * (double)a.S;
    {
        ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_bwd_reduce_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_bwd_reduce_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (const T*)a.gy,
                           a.partial, a.weight, a.bias, a.save_mean, a.save_invstd, a.B, a.C, p.cols, p.row_len, p.col_len,
                           p.col_tiles, p.tw_log2, a.wmod, a.slope);
    }
    ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_bwd_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_bwd_apply_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (const T*)a.gy, (T*)a.gx,
                       a.partial, a.weight, a.bias, a.save_mean, a.save_invstd, a.gweight, a.gbias, a.B, a.C, p.cols,
                       p.row_len, p.col_len, p.col_tiles, p.tw_log2, a.wmod, count, a.training, a.slope);
}

template <bool FWD>
static void norm_dispatch(const NormArgs& a, const TilePlan& p, int dtype, hipStream_t s) {
    if (dtype == NEXTOU_DTYPE_F32) {
        if (p.vec == 4) FWD ? launch_fwd<float, 4>(a, p, s, "f32") : launch_bwd<float, 4>(a, p, s, "f32");
        else FWD ? launch_fwd<float, 1>(a, p, s, "f32,scalar") : launch_bwd<float, 1>(a, p, s, "f32,scalar");
    } else if (dtype == NEXTOU_DTYPE_F16) {
        if (p.vec == 8) FWD ? launch_fwd<__half, 8>(a, p, s, "f16") : launch_bwd<__half, 8>(a, p, s, "f16");
        else FWD ? launch_fwd<__half, 1>(a, p, s, "f16,scalar") : launch_bwd<__half, 1>(a, p, s, "f16,scalar");
    } else {
        if (p.vec == 8) FWD ? launch_fwd<__hip_bfloat16, 8>(a, p, s, "bf16") : launch_bwd<__hip_bfloat16, 8>(a, p, s, "bf16");
        else FWD ? launch_fwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar") : launch_bwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar");
    }
}

static int check_common(const char* what, int B, int C, int64_t S, int param_period, int dtype, int channels_last) {
    NEXTOU_REQUIRE(B > 0 && C > 0 && C <= 65535 && S > 0, "%s: bad size B=%d C=%d S=%lld", what, B, C, (long long)S);
    NEXTOU_REQUIRE(dtype == NEXTOU_DTYPE_F32 || dtype == NEXTOU_DTYPE_BF16 || dtype == NEXTOU_DTYPE_F16,
                   "%s: dtype %d not in {f32, bf16, f16}", what, dtype);
    NEXTOU_REQUIRE(param_period >= 0, "%s: param_period=%d", what, param_period);
    if (channels_last) {
        if (param_period) return fail(NEXTOU_ENOTSUP, "%s: channels-last layout has no instance-norm mode", what);
    }
    return 0;
}

// Workspace layout: [C x slots] double2 partial sums, then C float2 backward coefficients.  `slots` covers both plans of a
// call with these sizes: <= 1024 tiles per channel (NCDHW), plan_cl().blocks workgroups (channels-last, C <= 256).
static size_t partial_slots(int B, int C, long long S) {
    size_t slots = 1024;
    if (C <= kThreads) {
        const size_t cl = (size_t)plan_cl((long long)B * C * S, C, 1, false).blocks;   // the scalar plan has the most blocks
        if (cl > slots) slots = cl;
    }
    return slots;
}
static size_t kCoeffOffset(int B, int C, long long S) { return (size_t)C * partial_slots(B, C, S) * sizeof(double2); }

template <typename T, int VEC>
void launch_cl_fwd(const NormArgs& a, const ClPlan& p, hipStream_t s, const char* tname) {
    const long long total = (long long)a.B * a.C * a.S;
    const double bytes = (double)total * sizeof(T);
    const double count = (double)a.B * (double)a.S;
    const size_t lds = (size_t)p.tact * VEC * sizeof(double2);
    if (a.training) {
        ProfScope prof(s, kBoundHbm, bytes, "bn_cl_stats_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_cl_stats_kernel<T, VEC>), dim3(p.blocks), dim3(kThreads), lds, s, (const T*)a.x, a.partial,
                           total, a.C, p.tact, p.span);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(a.C), dim3(kFinThreads), 0, s, a.partial, p.blocks, count, a.pre_bias,
                       a.running_mean, a.running_var, a.save_mean, a.save_invstd, a.training, a.momentum, a.eps);
    ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_cl_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_cl_apply_kernel<T, VEC>), dim3(p.blocks), dim3(kThreads), 0, s, (const T*)a.x, (T*)a.y, a.weight,
                       a.bias, a.save_mean, a.save_invstd, total, a.C, p.tact, p.span, a.slope);
}

template <typename T, int VEC>
void launch_cl_bwd(const NormArgs& a, const ClPlan& p, hipStream_t s, const char* tname) {
    const long long total = (long long)a.B * a.C * a.S;
    const double bytes = (double)total * sizeof(T);
    const double count = (double)a.B * (double)a.S;
    const size_t lds = (size_t)p.tact * VEC * sizeof(double2);
    float2* coeff = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.partial) + kCoeffOffset(a.B, a.C, a.S));
    {
        ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_cl_bwd_reduce_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_cl_bwd_reduce_kernel<T, VEC>), dim3(p.blocks), dim3(kThreads), lds, s, (const T*)a.x,
                           (const T*)a.gy, a.partial, a.weight, a.bias, a.save_mean, a.save_invstd, total, a.C, p.tact,
                           p.span, a.slope);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a.C), dim3(kFinThreads), 0, s, a.partial, p.blocks, count, coeff, a.gweight,
                       a.gbias, a.training);
    ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_cl_bwd_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_cl_bwd_apply_kernel<T, VEC>), dim3(p.blocks), dim3(kThreads), 0, s, (const T*)a.x, (const T*)a.gy,
                       (T*)a.gx, coeff, a.weight, a.bias, a.save_mean, a.save_invstd, total, a.C, p.tact, p.span, a.slope);
}


// ------------------------------------------------------------------------------------------------------------
// K6's channels-last backward with TWO incoming gradients, summed on load (round 6): the tensor a plain encoder stage hands both to the
// next stage and — as the skip connection — to the decoder's concatenation (reference NexToU_Encoder_Decoder.py:143-150 `skips.append`,
// :311-337 `torch.cat((x, skip), 1)`) receives one gradient from each.  Autograd adds them in a pass of its own (aten::add: 564 us for the
// 881-MB stage-0 tensor of cfg 2, 2 x 209 us at stage 1) and K6's reduce and apply passes then read the sum; here the two passes read both
// gradients instead (dz = g1 + g2, the same fp32 add).  gy2 is the concatenation's gradient where it lies: rows of C floats at a row stride
// ld2 >= C (a channel range of the wider concatenated rows).  fp32, 16-byte aligned pieces (C, ld2, the channel offset multiples of 4).
// ------------------------------------------------------------------------------------------------------------
struct TwoGrad {
    const float* p;          // gy2 + row * ld2 + col of the thread's first element
    long long step;          // elements per loop stride: (tact * 4 / C) rows * ld2
};
__device__ inline TwoGrad two_grad_at(const float* gy2, long long ld2, long long e, int C, long long stride) {
    const long long row = e / C;
    const int col = (int)(e - row * C);
    return TwoGrad{gy2 + row * ld2 + col, (stride / C) * ld2};
}

__global__ __launch_bounds__(kThreads) void bn_cl_bwd_reduce2_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                     const float* __restrict__ gy2, long long ld2, double2* __restrict__ partial,
                                                                     const float* __restrict__ weight, const float* __restrict__ bias,
                                                                     const float* __restrict__ save_mean, const float* __restrict__ save_invstd,
                                                                     long long total, int C, int tact, long long span, float slope) {
    constexpr int VEC = 4;
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0;
    if ((int)threadIdx.x < tact) {
        int ch[VEC];
        cl_channels<VEC>(C, ch);
        float scale[VEC], shift[VEC], mean[VEC], invstd[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float w = weight ? weight[ch[j]] : 1.f, b = bias ? bias[ch[j]] : 0.f;
            mean[j] = save_mean[ch[j]];
            invstd[j] = save_invstd[ch[j]];
            scale[j] = w * invstd[j];
            shift[j] = fmaf(-mean[j], scale[j], b);
        }
        const long long base = (long long)blockIdx.x * span;
        const long long end = min(total, base + span);
        const long long stride = (long long)tact * VEC;
        long long e = base + (long long)threadIdx.x * VEC;
        TwoGrad t = two_grad_at(gy2, ld2, e, C, stride);
        for (; e + stride < end; e += 2 * stride) {
            Pack<float, VEC> p[2], g[2], h[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) { p[u].load_stream(x + e + u * stride); g[u].load_stream(gy + e + u * stride); h[u].load_stream(t.p + u * t.step); }
            t.p += 2 * t.step;
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float gsum = g[u].v[j] + h[u].v[j];
                    const float z = fmaf(p[u].v[j], scale[j], shift[j]);
                    const float dz = z > 0.f ? gsum : gsum * slope;
                    const float xh = (p[u].v[j] - mean[j]) * invstd[j];
                    s1[j] += (double)dz;
                    s2[j] = fma((double)dz, (double)xh, s2[j]);
                }
        }
        for (; e < end; e += stride) {
            Pack<float, VEC> p, g, h;
            p.load_stream(x + e);
            g.load_stream(gy + e);
            h.load_stream(t.p);
            t.p += t.step;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float gsum = g.v[j] + h.v[j];
                const float z = fmaf(p.v[j], scale[j], shift[j]);
                const float dz = z > 0.f ? gsum : gsum * slope;
                const float xh = (p.v[j] - mean[j]) * invstd[j];
                s1[j] += (double)dz;
                s2[j] = fma((double)dz, (double)xh, s2[j]);
            }
        }
    }
    cl_block_sums<VEC>(s1, s2, C, tact, partial, gridDim.x);
}

__global__ __launch_bounds__(kThreads) void bn_cl_bwd_apply2_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                    const float* __restrict__ gy2, long long ld2, float* __restrict__ gx,
                                                                    const float2* __restrict__ coeff, const float* __restrict__ weight,
                                                                    const float* __restrict__ bias, const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd, long long total, int C, int tact,
                                                                    long long span, float slope) {
    constexpr int VEC = 4;
    if ((int)threadIdx.x >= tact) return;
    int ch[VEC];
    cl_channels<VEC>(C, ch);
    float scale[VEC], shift[VEC], mean[VEC], invstd[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const float w = weight ? weight[ch[j]] : 1.f, b = bias ? bias[ch[j]] : 0.f;
        mean[j] = save_mean[ch[j]];
        invstd[j] = save_invstd[ch[j]];
        scale[j] = w * invstd[j];
        shift[j] = fmaf(-mean[j], scale[j], b);
        const float2 k = coeff[ch[j]];
        k1[j] = k.x;
        k2[j] = k.y;
    }
    const long long base = (long long)blockIdx.x * span;
    const long long end = min(total, base + span);
    const long long stride = (long long)tact * VEC;
    long long e = base + (long long)threadIdx.x * VEC;
    TwoGrad t = two_grad_at(gy2, ld2, e, C, stride);
    for (; e + stride < end; e += 2 * stride) {
        Pack<float, VEC> p[2], g[2], h[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) { p[u].load(x + e + u * stride); g[u].load(gy + e + u * stride); h[u].load(t.p + u * t.step); }
        t.p += 2 * t.step;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float gsum = g[u].v[j] + h[u].v[j];
                const float z = fmaf(p[u].v[j], scale[j], shift[j]);
                const float dz = z > 0.f ? gsum : gsum * slope;
                const float xh = (p[u].v[j] - mean[j]) * invstd[j];
                g[u].v[j] = scale[j] * ((dz - k1[j]) - xh * k2[j]);
            }
            g[u].store(gx + e + u * stride);
        }
    }
    for (; e < end; e += stride) {
        Pack<float, VEC> p, g, h;
        p.load(x + e);
        g.load(gy + e);
        h.load(t.p);
        t.p += t.step;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float gsum = g.v[j] + h.v[j];
            const float z = fmaf(p.v[j], scale[j], shift[j]);
            const float dz = z > 0.f ? gsum : gsum * slope;
            const float xh = (p.v[j] - mean[j]) * invstd[j];
            g.v[j] = scale[j] * ((dz - k1[j]) - xh * k2[j]);
        }
        g.store(gx + e);
    }
}


// ======================================================================================================
// Channels-last, WIDE rows (C > 256: the graph stages' 264 / 324 / 528 / 648 / 1056 / 1296-channel tensors, and any C
// when NEXTOU_CLW=1).  x is R = B*S rows of C contiguous channels, cut into column blocks of CW = cx * VEC channels and
// row blocks; a workgroup is cx lanes across the channels x 256 / cx lanes down the rows, so a thread owns VEC fixed
// channels (scale / shift in registers), every access is a 16-byte piece of a >= 128-byte run, and the partial sums of a
// channel meet in LDS in a fixed order (bit-reproducible).  cx is picked per C for lane utilisation: C = 264 -> 8 lanes
// (9 blocks of 32 channels, 92 %), C = 1056 -> 32 lanes (9 blocks of 128, 92 %).  Same finalize kernels as above.
// ======================================================================================================
struct ClwPlan {
    int vec, cx_log2, ncb, nrb;
    long long rows_per_block;
};

static ClwPlan plan_clw(long long rows, int C, int vec_full, bool aligned) {
    ClwPlan p;
    p.vec = (aligned && C % vec_full == 0) ? vec_full : 1;
    // lanes across the channels: 8 .. 64.  Score = lane utilisation x (contiguous bytes per row and workgroup, saturating at 512 B).
    // Round 2 maximised the utilisation alone and took 8-lane (128-byte) runs for C = 132: 37-42 % of the HBM peak.  Measured sweep
    // (profiles/r03_k6_clw_lane_split.md): C = 132 is fastest with 64 lanes (one 528-byte run per row, half the lanes idle):
    // apply 53 -> 33 us (70 %), backward apply 82 -> 51 us (67 %); C = 264 / 528 with 32 lanes (512-byte runs).  Ties -> wider.
    double best = -1.0;
    p.cx_log2 = 3;
    for (int lg = 3; lg <= 6; ++lg) {
        const int cw = (1 << lg) * p.vec;
        const double util = (double)C / ((double)cdiv(C, cw) * cw);
        const double bytes = (1 << lg) * (p.vec > 1 ? 16.0 : 4.0);          // a lane moves 16 bytes on the vector paths of every dtype
        const double run = bytes >= 512.0 ? 1.0 : bytes / 512.0;
        const double score = util * run;
        if (score >= best - 1e-9) { best = score; p.cx_log2 = lg; }
    }
    if (const char* e = getenv("NEXTOU_CLW_LG")) {          // experiment: force the lanes-across-channels split (3 .. 6)
        const int lg = atoi(e);
        if (lg >= 3 && lg <= 6) p.cx_log2 = lg;
    }
    p.ncb = cdiv(C, (1 << p.cx_log2) * p.vec);
    const int ry = kThreads >> p.cx_log2;
    long long nrb = cdiv64(4096, p.ncb);                       // ~4096 workgroups ...
    const long long max_by_rows = cdiv64(rows, (long long)ry * 8);   // ... of at least 8 rows per thread
    if (nrb > max_by_rows) nrb = max_by_rows;
    if (nrb > 1024) nrb = 1024;                                // partial_slots()
    if (nrb < 1) nrb = 1;
    p.rows_per_block = cdiv64(rows, nrb);
    p.nrb = (int)cdiv64(rows, p.rows_per_block);
    return p;
}

struct ClwThread {
    int cx, ry, nry, ch0;        // lane across channels, lane down the rows, rows lanes, first owned channel
    long long r0, r1;
    bool active;
};
template <int VEC>
__device__ inline ClwThread clw_decode(long long rows, int C, int cx_log2, long long rows_per_block) {
    ClwThread t;
    t.cx = threadIdx.x & ((1 << cx_log2) - 1);
    t.ry = threadIdx.x >> cx_log2;
    t.nry = kThreads >> cx_log2;
    t.ch0 = (blockIdx.y << cx_log2) * VEC + t.cx * VEC;
    t.r0 = (long long)blockIdx.x * rows_per_block;
    t.r1 = min(rows, t.r0 + rows_per_block);
    t.active = t.ch0 < C;        // C % VEC == 0 on the vector path; VEC == 1 otherwise
    return t;
}

// combine the row-lanes' partials of every owned channel in ry order; lanes with ry == 0 write them out
template <int VEC>
__device__ inline void clw_block_sums(const double (&a)[VEC], const double (&b)[VEC], const ClwThread& t, int cx_log2, int C,
                                      double2* partial, int nrb) {
    extern __shared__ double2 clw_red[];       // [nry][cx * VEC]
    const int cw = (1 << cx_log2) * VEC;
#pragma unroll
    for (int j = 0; j < VEC; ++j) clw_red[t.ry * cw + t.cx * VEC + j] = make_double2(a[j], b[j]);
    __syncthreads();
    if (t.ry == 0 && t.active) {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            double s = 0.0, q = 0.0;
            for (int r = 0; r < t.nry; ++r) { const double2 v = clw_red[r * cw + t.cx * VEC + j]; s += v.x; q += v.y; }
            partial[(size_t)(t.ch0 + j) * nrb + blockIdx.x] = make_double2(s, q);
        }
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_clw_stats_kernel(const T* __restrict__ x, double2* __restrict__ partial,
                                                                long long rows, int C, int cx_log2, long long rows_per_block) {
    const ClwThread t = clw_decode<VEC>(rows, C, cx_log2, rows_per_block);
    double s[VEC], q[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s[j] = q[j] = 0.0;
    if (t.active) {
        const T* base = x + t.ch0;
        long long r = t.r0 + t.ry;
        for (; r + 3ll * t.nry < t.r1; r += 4ll * t.nry) {
            Pack<T, VEC> p[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) p[u].load(base + (r + (long long)u * t.nry) * C);
#pragma unroll
            for (int u = 0; u < 4; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) { const double v = (double)p[u].v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
        }
        for (; r < t.r1; r += t.nry) {
            Pack<T, VEC> p;
            p.load(base + r * C);
#pragma unroll
            for (int j = 0; j < VEC; ++j) { const double v = (double)p.v[j]; s[j] += v; q[j] = fma(v, v, q[j]); }
        }
    }
    clw_block_sums<VEC>(s, q, t, cx_log2, C, partial, gridDim.x);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_clw_apply_kernel(const T* __restrict__ x, T* __restrict__ y,
                                                                const float* __restrict__ weight, const float* __restrict__ bias,
                                                                const float* __restrict__ save_mean,
                                                                const float* __restrict__ save_invstd, long long rows, int C,
                                                                int cx_log2, long long rows_per_block, float slope,
                                                                const T* __restrict__ residual = nullptr) {
    const ClwThread t = clw_decode<VEC>(rows, C, cx_log2, rows_per_block);
    if (!t.active) return;
    float scale[VEC], shift[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = t.ch0 + j;
        const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
        scale[j] = w * save_invstd[c];
        shift[j] = fmaf(-save_mean[c], scale[j], b);
    }
    long long r = t.r0 + t.ry;
    if (residual) {        // see bn_cl_apply_kernel
        for (; r + t.nry < t.r1; r += 2ll * t.nry) {
            Pack<T, VEC> p[2], q[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                p[u].load(x + (r + (long long)u * t.nry) * C + t.ch0);
                q[u].load(residual + (r + (long long)u * t.nry) * C + t.ch0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) {
#pragma unroll
                for (int j = 0; j < VEC; ++j) p[u].v[j] = leaky(fmaf(p[u].v[j], scale[j], shift[j]), slope) + q[u].v[j];
                p[u].store(y + (r + (long long)u * t.nry) * C + t.ch0);
            }
        }
        for (; r < t.r1; r += t.nry) {
            Pack<T, VEC> p, q;
            p.load(x + r * C + t.ch0);
            q.load(residual + r * C + t.ch0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) p.v[j] = leaky(fmaf(p.v[j], scale[j], shift[j]), slope) + q.v[j];
            p.store(y + r * C + t.ch0);
        }
        return;
    }
    for (; r + 3ll * t.nry < t.r1; r += 4ll * t.nry) {
        Pack<T, VEC> p[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) p[u].load(x + (r + (long long)u * t.nry) * C + t.ch0);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) p[u].v[j] = leaky(fmaf(p[u].v[j], scale[j], shift[j]), slope);
            p[u].store(y + (r + (long long)u * t.nry) * C + t.ch0);
        }
    }
    for (; r < t.r1; r += t.nry) {
        Pack<T, VEC> p;
        p.load(x + r * C + t.ch0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) p.v[j] = leaky(fmaf(p.v[j], scale[j], shift[j]), slope);
        p.store(y + r * C + t.ch0);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_clw_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                     double2* __restrict__ partial,
                                                                     const float* __restrict__ weight, const float* __restrict__ bias,
                                                                     const float* __restrict__ save_mean,
                                                                     const float* __restrict__ save_invstd, long long rows,
                                                                     int C, int cx_log2, long long rows_per_block, float slope) {
    const ClwThread t = clw_decode<VEC>(rows, C, cx_log2, rows_per_block);
    double s1[VEC], s2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) s1[j] = s2[j] = 0.0;
    if (t.active) {
        float scale[VEC], shift[VEC], mean[VEC], invstd[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const int c = t.ch0 + j;
            const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
            mean[j] = save_mean[c];
            invstd[j] = save_invstd[c];
            scale[j] = w * invstd[j];
            shift[j] = fmaf(-mean[j], scale[j], b);
        }
        long long r = t.r0 + t.ry;
        for (; r + t.nry < t.r1; r += 2ll * t.nry) {
            Pack<T, VEC> p[2], g[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                p[u].load(x + (r + (long long)u * t.nry) * C + t.ch0);
                g[u].load(gy + (r + (long long)u * t.nry) * C + t.ch0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int j = 0; j < VEC; ++j) {
                    const float z = fmaf(p[u].v[j], scale[j], shift[j]);
                    const float dz = z > 0.f ? g[u].v[j] : g[u].v[j] * slope;
                    const float xh = (p[u].v[j] - mean[j]) * invstd[j];
                    s1[j] += (double)dz;
                    s2[j] = fma((double)dz, (double)xh, s2[j]);
                }
        }
        for (; r < t.r1; r += t.nry) {
            Pack<T, VEC> p, g;
            p.load(x + r * C + t.ch0);
            g.load(gy + r * C + t.ch0);
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float z = fmaf(p.v[j], scale[j], shift[j]);
                const float dz = z > 0.f ? g.v[j] : g.v[j] * slope;
                const float xh = (p.v[j] - mean[j]) * invstd[j];
                s1[j] += (double)dz;
                s2[j] = fma((double)dz, (double)xh, s2[j]);
            }
        }
    }
    clw_block_sums<VEC>(s1, s2, t, cx_log2, C, partial, gridDim.x);
}

template <typename T, int VEC>
__global__ __launch_bounds__(kThreads) void bn_clw_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ gy,
                                                                    T* __restrict__ gx, const float2* __restrict__ coeff,
                                                                    const float* __restrict__ weight, const float* __restrict__ bias,
                                                                    const float* __restrict__ save_mean,
                                                                    const float* __restrict__ save_invstd, long long rows,
                                                                    int C, int cx_log2, long long rows_per_block, float slope) {
    const ClwThread t = clw_decode<VEC>(rows, C, cx_log2, rows_per_block);
    if (!t.active) return;
    float scale[VEC], shift[VEC], mean[VEC], invstd[VEC], k1[VEC], k2[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
        const int c = t.ch0 + j;
        const float w = weight ? weight[c] : 1.f, b = bias ? bias[c] : 0.f;
        mean[j] = save_mean[c];
        invstd[j] = save_invstd[c];
        scale[j] = w * invstd[j];
        shift[j] = fmaf(-mean[j], scale[j], b);
        k1[j] = coeff[c].x;
        k2[j] = coeff[c].y;
    }
    for (long long r = t.r0 + t.ry; r < t.r1; r += t.nry) {
        Pack<T, VEC> p, g;
        p.load(x + r * C + t.ch0);
        g.load(gy + r * C + t.ch0);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float z = fmaf(p.v[j], scale[j], shift[j]);
            const float dz = z > 0.f ? g.v[j] : g.v[j] * slope;
            const float xh = (p.v[j] - mean[j]) * invstd[j];
            g.v[j] = scale[j] * ((dz - k1[j]) - xh * k2[j]);
        }
        g.store(gx + r * C + t.ch0);
    }
}

template <typename T, int VEC>
void launch_clw_fwd(const NormArgs& a, const ClwPlan& p, hipStream_t s, const char* tname) {
    const long long rows = (long long)a.B * a.S;
    const double bytes = (double)rows * a.C * sizeof(T);
    const double count = (double)rows;
    const dim3 grid(p.nrb, p.ncb);
    const size_t lds = (size_t)kThreads * VEC * sizeof(double2);
    if (a.training) {
        ProfScope prof(s, kBoundHbm, bytes, "bn_clw_stats_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_clw_stats_kernel<T, VEC>), grid, dim3(kThreads), lds, s, (const T*)a.x, a.partial, rows, a.C,
                           p.cx_log2, p.rows_per_block);
    }
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(a.C), dim3(kFinThreads), 0, s, a.partial, p.nrb, count, a.pre_bias,
                       a.running_mean, a.running_var, a.save_mean, a.save_invstd, a.training, a.momentum, a.eps);
    ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_clw_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_clw_apply_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (T*)a.y, a.weight, a.bias,
                       a.save_mean, a.save_invstd, rows, a.C, p.cx_log2, p.rows_per_block, a.slope);
}

template <typename T, int VEC>
void launch_clw_bwd(const NormArgs& a, const ClwPlan& p, hipStream_t s, const char* tname) {
    const long long rows = (long long)a.B * a.S;
    const double bytes = (double)rows * a.C * sizeof(T);
    const double count = (double)rows;
    const dim3 grid(p.nrb, p.ncb);
    const size_t lds = (size_t)kThreads * VEC * sizeof(double2);
    float2* coeff = reinterpret_cast<float2*>(reinterpret_cast<char*>(a.partial) + kCoeffOffset(a.B, a.C, a.S));
    {
        ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_clw_bwd_reduce_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_clw_bwd_reduce_kernel<T, VEC>), grid, dim3(kThreads), lds, s, (const T*)a.x, (const T*)a.gy,
                           a.partial, a.weight, a.bias, a.save_mean, a.save_invstd, rows, a.C, p.cx_log2, p.rows_per_block,
                           a.slope);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(a.C), dim3(kFinThreads), 0, s, a.partial, p.nrb, count, coeff, a.gweight,
                       a.gbias, a.training);
    ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_clw_bwd_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_clw_bwd_apply_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (const T*)a.gy, (T*)a.gx,
                       coeff, a.weight, a.bias, a.save_mean, a.save_invstd, rows, a.C, p.cx_log2, p.rows_per_block, a.slope);
}

template <bool FWD>
static void norm_dispatch_clw(const NormArgs& a, const ClwPlan& p, int dtype, hipStream_t s) {
    if (dtype == NEXTOU_DTYPE_F32) {
        if (p.vec == 4) FWD ? launch_clw_fwd<float, 4>(a, p, s, "f32") : launch_clw_bwd<float, 4>(a, p, s, "f32");
        else FWD ? launch_clw_fwd<float, 1>(a, p, s, "f32,scalar") : launch_clw_bwd<float, 1>(a, p, s, "f32,scalar");
    } else if (dtype == NEXTOU_DTYPE_F16) {
        if (p.vec == 8) FWD ? launch_clw_fwd<__half, 8>(a, p, s, "f16") : launch_clw_bwd<__half, 8>(a, p, s, "f16");
        else FWD ? launch_clw_fwd<__half, 1>(a, p, s, "f16,scalar") : launch_clw_bwd<__half, 1>(a, p, s, "f16,scalar");
    } else {
        if (p.vec == 8) FWD ? launch_clw_fwd<__hip_bfloat16, 8>(a, p, s, "bf16") : launch_clw_bwd<__hip_bfloat16, 8>(a, p, s, "bf16");
        else FWD ? launch_clw_fwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar") : launch_clw_bwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar");
    }
}

// which channels-last scheme a call takes: the row-packing kernels need C <= 256 and keep only C of their 256 lanes busy
// once a row no longer fits twice (C > 128: 52 % at C = 132, where the column-blocked scheme reaches 82 %).
// NEXTOU_CLW=1 forces the wide scheme for every C, NEXTOU_CLW=2 restricts it to C > 256 (A/B).
// ======================================================================================================
// K6 for SMALL tensors in ONE launch each way (round 6, SURVEY.md 8(f)-1 at stages 4 / 5).  A graph-stage block of cfg 2 runs five norms
// over tensors of 0.4 - 14 MB; as statistics -> finalize -> apply (and reduce -> finalize -> apply backwards) each of them is six launches of
// 4 - 9 us whose run time is latency, not bytes (profiles/r06_gnn45_kernels.txt).  Statistics are per channel, so a workgroup that owns ALL
// elements of its channels needs nobody else: statistics, finalisation and apply in one kernel, no workspace, no cross-workgroup step.
//   bn_one_cl_kernel    channels-last rows (R, C): a workgroup owns a column block of 4 NQ channels (NQ = 1 | 2 float4 per row) and walks all
//                       rows twice (second pass L2 / L1-hot); C / (4 NQ) workgroups.  The lines of a row are shared by 8 / NQ column blocks, which
//                       the L2 absorbs at these sizes — plan_one() bounds the tensor (<= 2 MB: stage 5 of cfg 2; measured, see there).
//   bn_one_rows_kernel  channel-major (B, C, S): a wave owns a channel (batch statistics over its B rows) or — param_period > 0, the caller's
//                       (1, B' C, S) view — one row (instance statistics); C / 4 workgroups.
// Same arithmetic as the multi-launch kernels (float64 sums in a fixed order, K6's finalize and apply expressions term for term), so results
// differ from them only in the order of the float64 partial sums.  fp32, 16-byte aligned, element counts as plan_one() says; everything else
// (and NEXTOU_K6_ONE=0) takes the multi-launch path.
// ======================================================================================================
constexpr int kOneThreads = 256;

__device__ __forceinline__ double wave_sum_from(double v, int lowest) {     // xor tree over lane offsets 32 .. lowest: lanes equal mod `lowest` meet
    for (int o = 32; o >= lowest; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

struct OneAffine { float mean, invstd, scale, shift; };

// what bn_finalize_kernel + the apply kernels' prologue compute, for one channel, from its (sum, sum of squares) or the running statistics
__device__ inline OneAffine one_finalize(double s, double q, double count, int c, int pc, const float* weight, const float* bias,
                                         const float* pre_bias, float* running_mean, float* running_var, float* save_mean, float* save_invstd,
                                         int training, float momentum, float eps, bool writer) {
    OneAffine a;
    double var = 0.0;
    if (training) {
        const double m = s / count;
        var = q / count - m * m;
        if (var < 0.0) var = 0.0;
        a.mean = (float)m;
        a.invstd = (float)(1.0 / sqrt(var + (double)eps));
    } else {
        a.mean = running_mean[c] - (pre_bias ? pre_bias[pc] : 0.f);
        a.invstd = 1.0f / sqrtf(running_var[c] + eps);
    }
    a.scale = (weight ? weight[pc] : 1.f) * a.invstd;
    a.shift = fmaf(-a.mean, a.scale, bias ? bias[pc] : 0.f);
    if (writer) {
        if (save_mean) save_mean[c] = a.mean;
        if (save_invstd) save_invstd[c] = a.invstd;
        if (training && running_mean) {
            const double unbiased = count > 1.0 ? var * (count / (count - 1.0)) : var;
            const double batch_mean = (double)a.mean + (pre_bias ? (double)pre_bias[pc] : 0.0);
            running_mean[c] = (float)((1.0 - (double)momentum) * (double)running_mean[c] + (double)momentum * batch_mean);
            running_var[c] = (float)((1.0 - (double)momentum) * (double)running_var[c] + (double)momentum * unbiased);
        }
    }
    return a;
}

template <int NQ, bool BWD>
__global__ __launch_bounds__(kOneThreads) void bn_one_cl_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ out,
                                                                 const float* __restrict__ weight, const float* __restrict__ bias,
                                                                 const float* __restrict__ pre_bias, float* running_mean, float* running_var,
                                                                 float* save_mean, float* save_invstd, float* __restrict__ gweight,
                                                                 float* __restrict__ gbias, long long R, int C, int training, float momentum,
                                                                 float eps, float slope) {
    constexpr int CW = 4 * NQ, RL = kOneThreads / NQ;
    __shared__ double2 red[kOneThreads / 64][CW];
    __shared__ float4 aff[CW];              // forward: (scale, shift, -, -); backward: (k1, k2, -, -)
    const int rq = threadIdx.x % NQ, rl = threadIdx.x / NQ;
    const int c0 = blockIdx.x * CW + 4 * rq;
    const bool live = c0 < C;                // (C % 4 == 0: a float4 is all inside or all outside)
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const double count = (double)R;
    const float* xp = x + (live ? c0 : 0);
    float sc[4], sh[4], mn[4], is[4];
    if (BWD) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int c = live ? c0 + j : 0;
            mn[j] = save_mean[c];
            is[j] = save_invstd[c];
            sc[j] = (weight ? weight[c] : 1.f) * is[j];
            sh[j] = fmaf(-mn[j], sc[j], bias ? bias[c] : 0.f);
        }
    }
    // ---- pass 1: (sum, sum of squares) | (sum dz, sum dz * xhat)
    double a[4] = {0.0, 0.0, 0.0, 0.0}, b[4] = {0.0, 0.0, 0.0, 0.0};
    if (!BWD ? training != 0 : true) {
        constexpr int U = 4;
        long long r = rl;
        for (; r + (U - 1) * RL < R; r += U * RL) {
            float4 v[U], g[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                v[u] = *reinterpret_cast<const float4*>(xp + (r + u * RL) * C);
                if (BWD) g[u] = *reinterpret_cast<const float4*>(gy + (live ? c0 : 0) + (r + u * RL) * C);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                const float gv[4] = {BWD ? g[u].x : 0.f, BWD ? g[u].y : 0.f, BWD ? g[u].z : 0.f, BWD ? g[u].w : 0.f};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BWD) {
                        const float z = fmaf(xv[j], sc[j], sh[j]);
                        const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                        const float xh = (xv[j] - mn[j]) * is[j];
                        a[j] += (double)dz;
                        b[j] = fma((double)dz, (double)xh, b[j]);
                    } else {
                        const double d = (double)xv[j];
                        a[j] += d;
                        b[j] = fma(d, d, b[j]);
                    }
                }
            }
        }
        for (; r < R; r += RL) {
            const float4 v = *reinterpret_cast<const float4*>(xp + r * C);
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BWD) g = *reinterpret_cast<const float4*>(gy + (live ? c0 : 0) + r * C);
            const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BWD) {
                    const float z = fmaf(xv[j], sc[j], sh[j]);
                    const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                    const float xh = (xv[j] - mn[j]) * is[j];
                    a[j] += (double)dz;
                    b[j] = fma((double)dz, (double)xh, b[j]);
                } else {
                    const double d = (double)xv[j];
                    a[j] += d;
                    b[j] = fma(d, d, b[j]);
                }
            }
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            a[j] = wave_sum_from(a[j], NQ);
            b[j] = wave_sum_from(b[j], NQ);
            if (lane < NQ) red[wave][4 * lane + j] = make_double2(a[j], b[j]);     // lane == rq for the lowest lanes of a wave (64 % NQ == 0)
        }
    }
    __syncthreads();
    if (threadIdx.x < CW) {
        const int c = blockIdx.x * CW + threadIdx.x;
        if (c < C) {
            double s = 0.0, q = 0.0;
#pragma unroll
            for (int w = 0; w < kOneThreads / 64; ++w) { s += red[w][threadIdx.x].x; q += red[w][threadIdx.x].y; }
            if (BWD) {
                aff[threadIdx.x] = training ? make_float4((float)(s / count), (float)(q / count), 0.f, 0.f) : make_float4(0.f, 0.f, 0.f, 0.f);
                if (gweight) gweight[c] = (float)q;
                if (gbias) gbias[c] = (float)s;
            } else {
                const OneAffine f = one_finalize(s, q, count, c, c, weight, bias, pre_bias, running_mean, running_var, save_mean, save_invstd,
                                                 training, momentum, eps, true);
                aff[threadIdx.x] = make_float4(f.scale, f.shift, 0.f, 0.f);
            }
        }
    }
    __syncthreads();
    if (!live) return;
    float p0[4], p1[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { p0[j] = aff[4 * rq + j].x; p1[j] = aff[4 * rq + j].y; }
    // ---- pass 2: y | gx
    float* op = out + c0;
    const float* gp = BWD ? gy + c0 : nullptr;
    constexpr int U2 = 4;
    long long r = rl;
    for (; r + (U2 - 1) * RL < R; r += U2 * RL) {
        float4 v[U2], g[U2];
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            v[u] = *reinterpret_cast<const float4*>(xp + (r + u * RL) * C);
            if (BWD) g[u] = *reinterpret_cast<const float4*>(gp + (r + u * RL) * C);
        }
#pragma unroll
        for (int u = 0; u < U2; ++u) {
            const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
            const float gv[4] = {BWD ? g[u].x : 0.f, BWD ? g[u].y : 0.f, BWD ? g[u].z : 0.f, BWD ? g[u].w : 0.f};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BWD) {
                    const float z = fmaf(xv[j], sc[j], sh[j]);
                    const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                    const float xh = (xv[j] - mn[j]) * is[j];
                    o[j] = sc[j] * ((dz - p0[j]) - xh * p1[j]);
                } else {
                    o[j] = leaky(fmaf(xv[j], p0[j], p1[j]), slope);
                }
            }
            *reinterpret_cast<float4*>(op + (r + u * RL) * C) = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
    for (; r < R; r += RL) {
        const float4 v = *reinterpret_cast<const float4*>(xp + r * C);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        if (BWD) g = *reinterpret_cast<const float4*>(gp + r * C);
        const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
        float o[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (BWD) {
                const float z = fmaf(xv[j], sc[j], sh[j]);
                const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                const float xh = (xv[j] - mn[j]) * is[j];
                o[j] = sc[j] * ((dz - p0[j]) - xh * p1[j]);
            } else {
                o[j] = leaky(fmaf(xv[j], p0[j], p1[j]), slope);
            }
        }
        *reinterpret_cast<float4*>(op + r * C) = make_float4(o[0], o[1], o[2], o[3]);
    }
}

template <bool BWD>
__global__ __launch_bounds__(kOneThreads) void bn_one_rows_kernel(const float* __restrict__ x, const float* __restrict__ gy, float* __restrict__ out,
                                                                   const float* __restrict__ weight, const float* __restrict__ bias,
                                                                   const float* __restrict__ pre_bias, float* running_mean, float* running_var,
                                                                   float* save_mean, float* save_invstd, float* __restrict__ gweight,
                                                                   float* __restrict__ gbias, int B, int C, long long S, int wmod, int training,
                                                                   float momentum, float eps, float slope) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int c = blockIdx.x * (kOneThreads / 64) + wave;
    if (c >= C) return;                      // (whole waves leave: no barrier below)
    const int pc = wmod > 0 ? c % wmod : c;
    const double count = (double)B * (double)S;
    const long long S4 = S >> 2;            // S % 4 == 0 (plan_one)
    float sc = 0.f, sh = 0.f, mn = 0.f, is = 0.f;
    if (BWD) {
        mn = save_mean[c];
        is = save_invstd[c];
        sc = (weight ? weight[pc] : 1.f) * is;
        sh = fmaf(-mn, sc, bias ? bias[pc] : 0.f);
    }
    double a = 0.0, b = 0.0;
    if (BWD || training) {
        for (int bb = 0; bb < B; ++bb) {
            const float4* xr = reinterpret_cast<const float4*>(x + ((long long)bb * C + c) * S);
            const float4* gr = BWD ? reinterpret_cast<const float4*>(gy + ((long long)bb * C + c) * S) : nullptr;
            long long i = lane;
            for (; i + 192 < S4; i += 256) {
                float4 v[4], g[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) { v[u] = xr[i + 64 * u]; if (BWD) g[u] = gr[i + 64 * u]; }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const float xv[4] = {v[u].x, v[u].y, v[u].z, v[u].w};
                    const float gv[4] = {BWD ? g[u].x : 0.f, BWD ? g[u].y : 0.f, BWD ? g[u].z : 0.f, BWD ? g[u].w : 0.f};
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        if (BWD) {
                            const float z = fmaf(xv[j], sc, sh);
                            const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                            a += (double)dz;
                            b = fma((double)dz, (double)((xv[j] - mn) * is), b);
                        } else {
                            const double d = (double)xv[j];
                            a += d;
                            b = fma(d, d, b);
                        }
                    }
                }
            }
            for (; i < S4; i += 64) {
                const float4 v = xr[i];
                float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
                if (BWD) g = gr[i];
                const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    if (BWD) {
                        const float z = fmaf(xv[j], sc, sh);
                        const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                        a += (double)dz;
                        b = fma((double)dz, (double)((xv[j] - mn) * is), b);
                    } else {
                        const double d = (double)xv[j];
                        a += d;
                        b = fma(d, d, b);
                    }
                }
            }
        }
        a = wave_sum_from(a, 1);
        b = wave_sum_from(b, 1);
    }
    float p0, p1;
    if (BWD) {
        p0 = training ? (float)(a / count) : 0.f;
        p1 = training ? (float)(b / count) : 0.f;
        if (lane == 0) {
            if (gweight) gweight[c] = (float)b;
            if (gbias) gbias[c] = (float)a;
        }
    } else {
        const OneAffine f = one_finalize(a, b, count, c, pc, weight, bias, pre_bias, wmod > 0 ? nullptr : running_mean,
                                         wmod > 0 ? nullptr : running_var, save_mean, save_invstd, training, momentum, eps, lane == 0);
        p0 = f.scale;
        p1 = f.shift;
    }
    for (int bb = 0; bb < B; ++bb) {
        const float4* xr = reinterpret_cast<const float4*>(x + ((long long)bb * C + c) * S);
        const float4* gr = BWD ? reinterpret_cast<const float4*>(gy + ((long long)bb * C + c) * S) : nullptr;
        float4* orow = reinterpret_cast<float4*>(out + ((long long)bb * C + c) * S);
        for (long long i = lane; i < S4; i += 64) {
            const float4 v = xr[i];
            float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
            if (BWD) g = gr[i];
            const float xv[4] = {v.x, v.y, v.z, v.w}, gv[4] = {g.x, g.y, g.z, g.w};
            float o[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                if (BWD) {
                    const float z = fmaf(xv[j], sc, sh);
                    const float dz = z > 0.f ? gv[j] : gv[j] * slope;
                    o[j] = sc * ((dz - p0) - ((xv[j] - mn) * is) * p1);
                } else {
                    o[j] = leaky(fmaf(xv[j], p0, p1), slope);
                }
            }
            orow[i] = make_float4(o[0], o[1], o[2], o[3]);
        }
    }
}

// 0: not taken; 1 | 2: float4 columns per workgroup (channels-last); 3: the channel-major kernel
static int plan_one(int B, int C, long long S, int channels_last, int dtype, int param_period, bool aligned, int n_partial) {
    static const bool enabled = [] { const char* e = getenv("NEXTOU_K6_ONE"); return !(e && e[0] == '0'); }();
    if (!enabled || dtype != NEXTOU_DTYPE_F32 || !aligned || n_partial > 0) return 0;
    const long long total = (long long)B * C * S;
    // measured as replayed blocks (profiles/r06_k6_one.md): stage 5 of cfg 2 (336 rows: FFN 131 -> 106 us, graphers 254 -> 218 / 240 -> 203 us
    // forward + backward) wins, stage 4 (2 688 rows: FFN 282 -> 318 us) loses — a workgroup walking 2 688 rows of 16-byte pieces is slower than
    // three launches that spread them over the chip — so: tensors up to 2 MB
    long long max_total = 1LL << 19;
    if (const char* e = getenv("NEXTOU_K6_ONE_MAX")) max_total = atoll(e);       // experiments
    if (total > max_total) return 0;
    if (channels_last) {
        if (param_period || C % 4 != 0) return 0;
        return C >= 512 ? 2 : 1;
    }
    if (S % 4 != 0 || (long long)B * S > (1LL << 16)) return 0;                  // a wave per channel: at most 64 K elements of statistics each
    return 3;
}

static bool use_clw(int C) {
    static const int mode = [] { const char* e = getenv("NEXTOU_CLW"); return e ? atoi(e) : 0; }();
    if (mode == 1) return true;
    if (mode == 2) return C > kThreads;
    return C > kThreads / 2;
}

template <bool FWD>
static void norm_dispatch_cl(const NormArgs& a, const ClPlan& p, int dtype, hipStream_t s) {
    if (dtype == NEXTOU_DTYPE_F32) {
        if (p.vec == 4) FWD ? launch_cl_fwd<float, 4>(a, p, s, "f32") : launch_cl_bwd<float, 4>(a, p, s, "f32");
        else FWD ? launch_cl_fwd<float, 1>(a, p, s, "f32,scalar") : launch_cl_bwd<float, 1>(a, p, s, "f32,scalar");
    } else if (dtype == NEXTOU_DTYPE_F16) {
        if (p.vec == 8) FWD ? launch_cl_fwd<__half, 8>(a, p, s, "f16") : launch_cl_bwd<__half, 8>(a, p, s, "f16");
        else FWD ? launch_cl_fwd<__half, 1>(a, p, s, "f16,scalar") : launch_cl_bwd<__half, 1>(a, p, s, "f16,scalar");
    } else {
        if (p.vec == 8) FWD ? launch_cl_fwd<__hip_bfloat16, 8>(a, p, s, "bf16") : launch_cl_bwd<__hip_bfloat16, 8>(a, p, s, "bf16");
        else FWD ? launch_cl_fwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar") : launch_cl_bwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar");
    }
}

}  // namespace nextou

using namespace nextou;

extern "C" size_t nextou_norm_act_workspace_bytes(int B, int C, int64_t S, int dtype) {
    (void)dtype;
    if (B <= 0 || C <= 0 || S <= 0) return 0;
    // tile partials (NCDHW: <= 1024 tiles per channel, channels-last: <= 2048 workgroups) + the backward's coefficients
    return kCoeffOffset(B, C, S) + (size_t)C * sizeof(float2) + 256;
}

extern "C" int nextou_norm_act_fwd(const void* x, const float* weight, const float* bias, const float* pre_bias,
                                   float* running_mean, float* running_var, void* y, float* save_mean, float* save_invstd, void* ws,
                                   size_t ws_bytes, int B, int C, int64_t S, int param_period, int dtype, int channels_last,
                                   int training, float momentum, float eps, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && y, "norm_act_fwd: null pointer");
    if (int rc = check_common("norm_act_fwd", B, C, S, param_period, dtype, channels_last)) return rc;
    NEXTOU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "norm_act_fwd: running_mean / running_var must come together");
    NEXTOU_REQUIRE(training || running_mean, "norm_act_fwd: inference needs the running statistics");
    const int esz = dtype == NEXTOU_DTYPE_F32 ? 4 : 2;
    NormArgs a{};
    a.x = x; a.y = y; a.weight = weight; a.bias = bias; a.pre_bias = pre_bias; a.running_mean = running_mean; a.running_var = running_var;
    a.save_mean = save_mean; a.save_invstd = save_invstd; a.partial = (double2*)ws; a.B = B; a.C = C; a.S = S;
    a.wmod = param_period; a.training = training; a.momentum = momentum; a.eps = eps; a.slope = slope;
    if (const int one = plan_one(B, C, S, channels_last, dtype, param_period, aligned16(x) && aligned16(y), 0)) {
        hipStream_t s = (hipStream_t)stream;
        const float* xf = (const float*)x;
        float* yf = (float*)y;
        const double bytes = 4.0 * B * (double)C * (double)S;
        if (one == 3) {
            ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_one_rows_kernel<fwd>[B%d C%d S%lld]", B, C, (long long)S);
            hipLaunchKernelGGL((bn_one_rows_kernel<false>), dim3(cdiv(C, kOneThreads / 64)), dim3(kOneThreads), 0, s, xf, nullptr, yf, weight, bias, pre_bias,
                               running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, B, C, (long long)S, param_period, training, momentum,
                               eps, slope);
        } else {
            ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_one_cl_kernel<fwd,%d>[R%lld C%d]", one, (long long)B * S, C);
            if (one == 2)
                hipLaunchKernelGGL((bn_one_cl_kernel<2, false>), dim3(cdiv(C, 8)), dim3(kOneThreads), 0, s, xf, nullptr, yf, weight, bias, pre_bias,
                                   running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, (long long)B * S, C, training, momentum, eps, slope);
            else
                hipLaunchKernelGGL((bn_one_cl_kernel<1, false>), dim3(cdiv(C, 4)), dim3(kOneThreads), 0, s, xf, nullptr, yf, weight, bias, pre_bias,
                                   running_mean, running_var, save_mean, save_invstd, nullptr, nullptr, (long long)B * S, C, training, momentum, eps, slope);
        }
        return check_launch("bn_one_kernel<fwd>");
    }
    if (channels_last) {
        NEXTOU_REQUIRE(save_mean && save_invstd, "norm_act_fwd: the channels-last path needs save_mean / save_invstd");
        NEXTOU_REQUIRE(!training || (ws && ws_bytes >= nextou_norm_act_workspace_bytes(B, C, S, dtype)),
                       "norm_act_fwd: workspace %zu < %zu bytes", ws_bytes, nextou_norm_act_workspace_bytes(B, C, S, dtype));
        if (use_clw(C)) {
            norm_dispatch_clw<true>(a, plan_clw((long long)B * S, C, 16 / esz, aligned16(x) && aligned16(y)), dtype, (hipStream_t)stream);
            return check_launch("bn_clw_apply_kernel");
        }
        const ClPlan p = plan_cl((long long)B * C * S, C, 16 / esz, aligned16(x) && aligned16(y));
        norm_dispatch_cl<true>(a, p, dtype, (hipStream_t)stream);
        return check_launch("bn_cl_apply_kernel");
    }
    const TilePlan p = plan_tiles(B, C, S, 16 / esz, aligned16(x) && aligned16(y));
    if (training) {
        NEXTOU_REQUIRE(ws, "norm_act_fwd: null workspace");
        if (ws_bytes < (size_t)C * p.tiles * sizeof(double2))
            return fail(NEXTOU_ENOSPACE, "norm_act_fwd: workspace %zu < %zu bytes", ws_bytes, (size_t)C * p.tiles * sizeof(double2));
    }
    norm_dispatch<true>(a, p, dtype, (hipStream_t)stream);
    return check_launch("bn_apply_kernel");
}

extern "C" int nextou_norm_act_bwd(const void* x, const void* gy, const float* weight, const float* bias,
                                   const float* save_mean, const float* save_invstd, void* gx, float* gweight,
                                   float* gbias, void* ws, size_t ws_bytes, int B, int C, int64_t S, int param_period,
                                   int dtype, int channels_last, int training, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && gy && gx && save_mean && save_invstd && ws, "norm_act_bwd: null pointer");
    if (int rc = check_common("norm_act_bwd", B, C, S, param_period, dtype, channels_last)) return rc;
    const int esz = dtype == NEXTOU_DTYPE_F32 ? 4 : 2;
    NormArgs a{};
    a.x = x; a.gy = gy; a.gx = gx; a.weight = weight; a.bias = bias;
    a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
    a.gweight = gweight; a.gbias = gbias; a.partial = (double2*)ws; a.B = B; a.C = C; a.S = S; a.wmod = param_period;
    a.training = training; a.slope = slope;
    if (const int one = plan_one(B, C, S, channels_last, dtype, param_period, aligned16(x) && aligned16(gy) && aligned16(gx), 0)) {
        hipStream_t s = (hipStream_t)stream;
        const float *xf = (const float*)x, *gf = (const float*)gy;
        float* of = (float*)gx;
        const double bytes = 4.0 * B * (double)C * (double)S;
        if (one == 3) {
            ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_one_rows_kernel<bwd>[B%d C%d S%lld]", B, C, (long long)S);
            hipLaunchKernelGGL((bn_one_rows_kernel<true>), dim3(cdiv(C, kOneThreads / 64)), dim3(kOneThreads), 0, s, xf, gf, of, weight, bias, nullptr, nullptr,
                               nullptr, const_cast<float*>(save_mean), const_cast<float*>(save_invstd), gweight, gbias, B, C, (long long)S, param_period,
                               training, 0.f, 0.f, slope);
        } else {
            ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_one_cl_kernel<bwd,%d>[R%lld C%d]", one, (long long)B * S, C);
            if (one == 2)
                hipLaunchKernelGGL((bn_one_cl_kernel<2, true>), dim3(cdiv(C, 8)), dim3(kOneThreads), 0, s, xf, gf, of, weight, bias, nullptr, nullptr, nullptr,
                                   const_cast<float*>(save_mean), const_cast<float*>(save_invstd), gweight, gbias, (long long)B * S, C, training, 0.f, 0.f,
                                   slope);
            else
                hipLaunchKernelGGL((bn_one_cl_kernel<1, true>), dim3(cdiv(C, 4)), dim3(kOneThreads), 0, s, xf, gf, of, weight, bias, nullptr, nullptr, nullptr,
                                   const_cast<float*>(save_mean), const_cast<float*>(save_invstd), gweight, gbias, (long long)B * S, C, training, 0.f, 0.f,
                                   slope);
        }
        return check_launch("bn_one_kernel<bwd>");
    }
    if (channels_last) {
        if (ws_bytes < nextou_norm_act_workspace_bytes(B, C, S, dtype))
            return fail(NEXTOU_ENOSPACE, "norm_act_bwd: workspace %zu < %zu bytes", ws_bytes, nextou_norm_act_workspace_bytes(B, C, S, dtype));
        if (use_clw(C)) {
            norm_dispatch_clw<false>(a, plan_clw((long long)B * S, C, 16 / esz, aligned16(x) && aligned16(gy) && aligned16(gx)), dtype,
                                     (hipStream_t)stream);
            return check_launch("bn_clw_bwd_apply_kernel");
        }
        const ClPlan p = plan_cl((long long)B * C * S, C, 16 / esz, aligned16(x) && aligned16(gy) && aligned16(gx));
        norm_dispatch_cl<false>(a, p, dtype, (hipStream_t)stream);
        return check_launch("bn_cl_bwd_apply_kernel");
    }
    const TilePlan p = plan_tiles(B, C, S, 16 / esz, aligned16(x) && aligned16(gy) && aligned16(gx));
    if (ws_bytes < (size_t)C * p.tiles * sizeof(double2))
        return fail(NEXTOU_ENOSPACE, "norm_act_bwd: workspace %zu < %zu bytes", ws_bytes, (size_t)C * p.tiles * sizeof(double2));
    norm_dispatch<false>(a, p, dtype, (hipStream_t)stream);
    return check_launch("bn_bwd_apply_kernel");
}

// K6's backward for a channels-last fp32 tensor (C <= 128: the bn_cl kernels) whose gradient arrives as TWO tensors — see bn_cl_bwd_reduce2_kernel.
extern "C" int nextou_norm_act_bwd_two(const float* x, const float* gy, const float* gy2, int64_t ld2, const float* weight, const float* bias,
                                       const float* save_mean, const float* save_invstd, float* gx, float* gweight, float* gbias, void* ws,
                                       size_t ws_bytes, int B, int C, int64_t S, int training, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && gy && gy2 && gx && save_mean && save_invstd && ws, "norm_act_bwd_two: null pointer");
    if (int rc = check_common("norm_act_bwd_two", B, C, S, 0, NEXTOU_DTYPE_F32, 1)) return rc;
    if (use_clw(C) || C % 4 != 0 || ld2 < C || ld2 % 4 != 0 || !(aligned16(x) && aligned16(gy) && aligned16(gy2) && aligned16(gx)))
        return fail(NEXTOU_ENOTSUP, "norm_act_bwd_two: C = %d (<= 128, a multiple of 4), ld2 = %lld (>= C, a multiple of 4) and 16-byte aligned operands only",
                    C, (long long)ld2);
    if (ws_bytes < nextou_norm_act_workspace_bytes(B, C, S, NEXTOU_DTYPE_F32))
        return fail(NEXTOU_ENOSPACE, "norm_act_bwd_two: workspace %zu < %zu bytes", ws_bytes, nextou_norm_act_workspace_bytes(B, C, S, NEXTOU_DTYPE_F32));
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)B * C * S;
    const ClPlan p = plan_cl(total, C, 4, true);
    if (p.vec != 4) return fail(NEXTOU_ENOTSUP, "norm_act_bwd_two: the element count must be a multiple of 4");
    const double bytes = (double)total * sizeof(float);
    const double count = (double)B * (double)S;
    double2* partial = (double2*)ws;
    float2* coeff = reinterpret_cast<float2*>(reinterpret_cast<char*>(ws) + kCoeffOffset(B, C, S));
    const size_t lds = (size_t)p.tact * 4 * sizeof(double2);
    {
        ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_cl_bwd_reduce2_kernel<f32>[B%d C%d S%lld]", B, C, (long long)S);
        hipLaunchKernelGGL(bn_cl_bwd_reduce2_kernel, dim3(p.blocks), dim3(kThreads), lds, s, x, gy, gy2, (long long)ld2, partial, weight, bias, save_mean,
                           save_invstd, total, C, p.tact, p.span, slope);
    }
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(kFinThreads), 0, s, partial, p.blocks, count, coeff, gweight, gbias, training);
    ProfScope prof(s, kBoundHbm, 4.0 * bytes, "bn_cl_bwd_apply2_kernel<f32>[B%d C%d S%lld]", B, C, (long long)S);
    hipLaunchKernelGGL(bn_cl_bwd_apply2_kernel, dim3(p.blocks), dim3(kThreads), 0, s, x, gy, gy2, (long long)ld2, gx, coeff, weight, bias, save_mean,
                       save_invstd, total, C, p.tact, p.span, slope);
    return check_launch("bn_cl_bwd_apply2_kernel");
}

// K6's forward for a channel-major (B, C, S) fp32 tensor whose (sum, sum of squares) partials another kernel already wrote —
// mr_grp_cm_kernel's epilogue (csrc/mr_aggregate.hip): finalize + normalise + LeakyReLU in the one apply launch, no statistics pass.
// Batch statistics over (B, S) per channel (param_period = 0) or, with param_period = C_real > 0 and B = 1, instance statistics
// per row of the (1, B' C_real, S) view; training mode semantics (the running statistics, if given, are updated).
extern "C" int nextou_norm_act_fwd_partials(const float* x, const float* weight, const float* bias, float* running_mean,
                                            float* running_var, float* y, float* save_mean, float* save_invstd, const double* partial,
                                            int n_partial, int B, int C, int64_t S, int param_period, float momentum, float eps,
                                            float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && y && partial && n_partial > 0, "norm_act_fwd_partials: null pointer or no partials");
    if (int rc = check_common("norm_act_fwd_partials", B, C, S, param_period, NEXTOU_DTYPE_F32, 0)) return rc;
    NEXTOU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "norm_act_fwd_partials: running_mean / running_var must come together");
    NormArgs a{};
    a.x = x; a.y = y; a.weight = weight; a.bias = bias; a.running_mean = running_mean; a.running_var = running_var;
    a.save_mean = save_mean; a.save_invstd = save_invstd; a.partial = reinterpret_cast<double2*>(const_cast<double*>(partial));
    a.B = B; a.C = C; a.S = S; a.wmod = param_period; a.training = 1; a.momentum = momentum; a.eps = eps; a.slope = slope;
    a.n_partial = n_partial;
    const TilePlan p = plan_tiles(B, C, S, 4, aligned16(x) && aligned16(y));
    norm_dispatch<true>(a, p, NEXTOU_DTYPE_F32, (hipStream_t)stream);
    return check_launch("bn_apply_kernel");
}

// ------------------------------------------------------------------------------------------------------------
// K6 in pieces (round 3): for callers whose producer already delivered the per-tile partial sums — K7's
// statistics epilogues (csrc/pw_gemm.hip) — or whose consumer normalises on operand load.  Channels-last fp32 rows.
// ------------------------------------------------------------------------------------------------------------
extern "C" int nextou_norm_finalize(const double* partial, int tiles, double count, const float* pre_bias, float* running_mean,
                                    float* running_var, float* save_mean, float* save_invstd, const float* weight, const float* bias,
                                    float* scale, float* shift, int C, int training, float momentum, float eps, nextou_stream_t stream) {
    NEXTOU_REQUIRE(save_mean && save_invstd && C > 0 && count > 0.0, "norm_finalize: null pointer or empty problem");
    NEXTOU_REQUIRE(!training || (partial && tiles > 0), "norm_finalize: training needs the partial sums");
    NEXTOU_REQUIRE(training || (running_mean && running_var), "norm_finalize: inference needs the running statistics");
    NEXTOU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "norm_finalize: running_mean / running_var must come together");
    NEXTOU_REQUIRE((scale == nullptr) == (shift == nullptr), "norm_finalize: scale / shift must come together");
    hipLaunchKernelGGL(bn_finalize_kernel, dim3(C), dim3(kFinThreads), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(partial), tiles,
                       count, pre_bias, running_mean, running_var, save_mean, save_invstd, training, momentum, eps, weight, bias, scale, shift);
    return check_launch("bn_finalize_kernel");
}

extern "C" int nextou_norm_apply_rows(const float* x, const float* residual, float* y, const float* weight, const float* bias,
                                      const float* save_mean, const float* save_invstd, int64_t rows, int C, float slope,
                                      nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && y && save_mean && save_invstd && rows > 0 && C > 0, "norm_apply_rows: null pointer or empty problem");
    hipStream_t s = (hipStream_t)stream;
    const bool al = aligned16(x) && aligned16(y) && (!residual || aligned16(residual));
    const double bytes = (double)rows * C * sizeof(float);
    if (use_clw(C)) {
        const ClwPlan p = plan_clw(rows, C, 4, al);
        const dim3 grid(p.nrb, p.ncb);
        ProfScope prof(s, kBoundHbm, (residual ? 3.0 : 2.0) * bytes, "bn_clw_apply_kernel<f32%s>[R%lld C%d]", residual ? ",+res" : "", (long long)rows, C);
        if (p.vec == 4)
            hipLaunchKernelGGL((bn_clw_apply_kernel<float, 4>), grid, dim3(kThreads), 0, s, x, y, weight, bias, save_mean, save_invstd, (long long)rows,
                               C, p.cx_log2, p.rows_per_block, slope, residual);
        else
            hipLaunchKernelGGL((bn_clw_apply_kernel<float, 1>), grid, dim3(kThreads), 0, s, x, y, weight, bias, save_mean, save_invstd, (long long)rows,
                               C, p.cx_log2, p.rows_per_block, slope, residual);
        return check_launch("bn_clw_apply_kernel");
    }
    const long long total = (long long)rows * C;
    const ClPlan p = plan_cl(total, C, 4, al);
    ProfScope prof(s, kBoundHbm, (residual ? 3.0 : 2.0) * bytes, "bn_cl_apply_kernel<f32%s>[R%lld C%d]", residual ? ",+res" : "", (long long)rows, C);
    if (p.vec == 4)
        hipLaunchKernelGGL((bn_cl_apply_kernel<float, 4>), dim3(p.blocks), dim3(kThreads), 0, s, x, y, weight, bias, save_mean, save_invstd, total, C,
                           p.tact, p.span, slope, residual);
    else
        hipLaunchKernelGGL((bn_cl_apply_kernel<float, 1>), dim3(p.blocks), dim3(kThreads), 0, s, x, y, weight, bias, save_mean, save_invstd, total, C,
                           p.tact, p.span, slope, residual);
    return check_launch("bn_cl_apply_kernel");
}

extern "C" int nextou_norm_bwd_finalize(const double* partial, int tiles, double count, float* coeff, float* gweight, float* gbias, int C,
                                        int training, nextou_stream_t stream) {
    NEXTOU_REQUIRE(partial && coeff && tiles > 0 && C > 0 && count > 0.0, "norm_bwd_finalize: null pointer or empty problem");
    hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(kFinThreads), 0, (hipStream_t)stream, reinterpret_cast<const double2*>(partial), tiles,
                       count, reinterpret_cast<float2*>(coeff), gweight, gbias, training);
    return check_launch("bn_bwd_finalize_kernel");
}

extern "C" int nextou_norm_bwd_apply_rows(const float* x, const float* gy, float* gx, const float* coeff, const float* weight,
                                          const float* bias, const float* save_mean, const float* save_invstd, int64_t rows, int C,
                                          float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && gy && gx && coeff && save_mean && save_invstd && rows > 0 && C > 0, "norm_bwd_apply_rows: null pointer or empty problem");
    hipStream_t s = (hipStream_t)stream;
    const bool al = aligned16(x) && aligned16(gy) && aligned16(gx);
    const double bytes = (double)rows * C * sizeof(float);
    const float2* k = reinterpret_cast<const float2*>(coeff);
    if (use_clw(C)) {
        const ClwPlan p = plan_clw(rows, C, 4, al);
        const dim3 grid(p.nrb, p.ncb);
        ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_clw_bwd_apply_kernel<f32>[R%lld C%d]", (long long)rows, C);
        if (p.vec == 4)
            hipLaunchKernelGGL((bn_clw_bwd_apply_kernel<float, 4>), grid, dim3(kThreads), 0, s, x, gy, gx, k, weight, bias, save_mean, save_invstd,
                               (long long)rows, C, p.cx_log2, p.rows_per_block, slope);
        else
            hipLaunchKernelGGL((bn_clw_bwd_apply_kernel<float, 1>), grid, dim3(kThreads), 0, s, x, gy, gx, k, weight, bias, save_mean, save_invstd,
                               (long long)rows, C, p.cx_log2, p.rows_per_block, slope);
        return check_launch("bn_clw_bwd_apply_kernel");
    }
    const long long total = (long long)rows * C;
    const ClPlan p = plan_cl(total, C, 4, al);
    ProfScope prof(s, kBoundHbm, 3.0 * bytes, "bn_cl_bwd_apply_kernel<f32>[R%lld C%d]", (long long)rows, C);
    if (p.vec == 4)
        hipLaunchKernelGGL((bn_cl_bwd_apply_kernel<float, 4>), dim3(p.blocks), dim3(kThreads), 0, s, x, gy, gx, k, weight, bias, save_mean, save_invstd,
                           total, C, p.tact, p.span, slope);
    else
        hipLaunchKernelGGL((bn_cl_bwd_apply_kernel<float, 1>), dim3(p.blocks), dim3(kThreads), 0, s, x, gy, gx, k, weight, bias, save_mean, save_invstd,
                           total, C, p.tact, p.span, slope);
    return check_launch("bn_cl_bwd_apply_kernel");
}

extern "C" int nextou_channel_sum(const void* x, float* out, void* ws, size_t ws_bytes, int B, int C, int64_t S, int dtype,
                                  int channels_last, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && out && ws, "channel_sum: null pointer");
    if (int rc = check_common("channel_sum", B, C, S, 0, dtype, channels_last)) return rc;
    if (ws_bytes < nextou_norm_act_workspace_bytes(B, C, S, dtype))
        return fail(NEXTOU_ENOSPACE, "channel_sum: workspace %zu < %zu bytes", ws_bytes, nextou_norm_act_workspace_bytes(B, C, S, dtype));
    hipStream_t s = (hipStream_t)stream;
    const int esz = dtype == NEXTOU_DTYPE_F32 ? 4 : 2;
    const double bytes = (double)B * C * (double)S * esz;
    double2* partial = (double2*)ws;
    int tiles;
    ProfScope prof(s, kBoundHbm, bytes, "channel_sum<%s,%s>[B%d C%d S%lld]", dtype == NEXTOU_DTYPE_F32 ? "f32" : (dtype == NEXTOU_DTYPE_F16 ? "f16" : "bf16"),
                   channels_last ? "ndhwc" : "ncdhw", B, C, (long long)S);
    if (channels_last && use_clw(C)) {
        const long long rows = (long long)B * S;
        const ClwPlan p = plan_clw(rows, C, 16 / esz, aligned16(x));
        tiles = p.nrb;
        const dim3 grid(p.nrb, p.ncb);
        const size_t lds = (size_t)kThreads * p.vec * sizeof(double2);
#define NEXTOU_CLW_SUM(T, V) hipLaunchKernelGGL((bn_clw_stats_kernel<T, V>), grid, dim3(kThreads), lds, s, (const T*)x, partial, rows, C, p.cx_log2, p.rows_per_block)
        if (esz == 4) { if (p.vec == 4) NEXTOU_CLW_SUM(float, 4); else NEXTOU_CLW_SUM(float, 1); }
        else if (dtype == NEXTOU_DTYPE_F16) { if (p.vec == 8) NEXTOU_CLW_SUM(__half, 8); else NEXTOU_CLW_SUM(__half, 1); }
        else { if (p.vec == 8) NEXTOU_CLW_SUM(__hip_bfloat16, 8); else NEXTOU_CLW_SUM(__hip_bfloat16, 1); }
#undef NEXTOU_CLW_SUM
    } else if (channels_last) {
        const long long total = (long long)B * C * S;
        const ClPlan p = plan_cl(total, C, 16 / esz, aligned16(x));
        tiles = p.blocks;
        const size_t lds = (size_t)p.tact * p.vec * sizeof(double2);
        if (esz == 4) {
            if (p.vec == 4) hipLaunchKernelGGL((bn_cl_stats_kernel<float, 4>), dim3(p.blocks), dim3(kThreads), lds, s, (const float*)x, partial, total, C, p.tact, p.span);
            else hipLaunchKernelGGL((bn_cl_stats_kernel<float, 1>), dim3(p.blocks), dim3(kThreads), lds, s, (const float*)x, partial, total, C, p.tact, p.span);
        } else if (dtype == NEXTOU_DTYPE_F16) {
            if (p.vec == 8) hipLaunchKernelGGL((bn_cl_stats_kernel<__half, 8>), dim3(p.blocks), dim3(kThreads), lds, s, (const __half*)x, partial, total, C, p.tact, p.span);
            else hipLaunchKernelGGL((bn_cl_stats_kernel<__half, 1>), dim3(p.blocks), dim3(kThreads), lds, s, (const __half*)x, partial, total, C, p.tact, p.span);
        } else {
            if (p.vec == 8) hipLaunchKernelGGL((bn_cl_stats_kernel<__hip_bfloat16, 8>), dim3(p.blocks), dim3(kThreads), lds, s, (const __hip_bfloat16*)x, partial, total, C, p.tact, p.span);
            else hipLaunchKernelGGL((bn_cl_stats_kernel<__hip_bfloat16, 1>), dim3(p.blocks), dim3(kThreads), lds, s, (const __hip_bfloat16*)x, partial, total, C, p.tact, p.span);
        }
    } else {
        const TilePlan p = plan_tiles(B, C, S, 16 / esz, aligned16(x));
        tiles = p.tiles;
        const dim3 grid(p.tiles, C);
        if (esz == 4) {
            if (p.vec == 4) hipLaunchKernelGGL((bn_stats_kernel<float, 4>), grid, dim3(kThreads), 0, s, (const float*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
            else hipLaunchKernelGGL((bn_stats_kernel<float, 1>), grid, dim3(kThreads), 0, s, (const float*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
        } else if (dtype == NEXTOU_DTYPE_F16) {
            if (p.vec == 8) hipLaunchKernelGGL((bn_stats_kernel<__half, 8>), grid, dim3(kThreads), 0, s, (const __half*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
            else hipLaunchKernelGGL((bn_stats_kernel<__half, 1>), grid, dim3(kThreads), 0, s, (const __half*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
        } else {
            if (p.vec == 8) hipLaunchKernelGGL((bn_stats_kernel<__hip_bfloat16, 8>), grid, dim3(kThreads), 0, s, (const __hip_bfloat16*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
            else hipLaunchKernelGGL((bn_stats_kernel<__hip_bfloat16, 1>), grid, dim3(kThreads), 0, s, (const __hip_bfloat16*)x, partial, B, C, p.cols, p.row_len, p.col_len, p.col_tiles, p.tw_log2);
        }
    }
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C), dim3(kFinThreads), 0, s, partial, tiles, out);
    return check_launch("channel_sum");
}

static int narrow_copy_sum_impl(const char* who, const float* src, float* dst, float* sum_out, void* ws, size_t ws_bytes, int64_t P, int C,
                                int64_t ld, int c_off, const UpShuffle* shuffle, hipStream_t s) {
    NEXTOU_REQUIRE(src && dst && sum_out && ws, "%s: null pointer", who);
    NEXTOU_REQUIRE(P > 0 && P <= (1ll << 40) && C > 0 && ld >= C && c_off >= 0 && c_off + (int64_t)C <= ld,
                   "%s: bad size P=%lld C=%d ld=%lld c_off=%d", who, (long long)P, C, (long long)ld, c_off);
    if (C % 4 != 0 || ld % 4 != 0 || c_off % 4 != 0 || use_clw(C) || !aligned16(src) || !aligned16(dst))
        return fail(NEXTOU_ENOTSUP, "%s: takes 16-byte aligned rows of at most %d channels, counts and offsets multiples of 4 "
                    "(C=%d ld=%lld c_off=%d)", who, kThreads / 2, C, (long long)ld, c_off);
    const size_t need = nextou_norm_act_workspace_bytes(1, C, P, NEXTOU_DTYPE_F32);
    if (ws_bytes < need) return fail(NEXTOU_ENOSPACE, "%s: workspace %zu < %zu bytes", who, ws_bytes, need);
    const long long total = (long long)P * C;
    const ClPlan p = plan_cl(total, C, 4, true);        // total % 4 == 0 since C % 4 == 0
    double2* partial = (double2*)ws;
    {
        ProfScope prof(s, kBoundHbm, 8.0 * (double)total, "narrow_copy_stats_kernel<%s>[P%lld C%d of %lld]", shuffle ? "unshuffle" : "copy",
                       (long long)P, C, (long long)ld);
        const size_t lds = (size_t)p.tact * 4 * sizeof(double2);
        if (shuffle)
            hipLaunchKernelGGL(narrow_copy_stats_kernel<true>, dim3(p.blocks), dim3(kThreads), lds, s, src, dst, partial, total, C, p.tact,
                               p.span, (long long)ld, c_off, *shuffle);
        else
            hipLaunchKernelGGL(narrow_copy_stats_kernel<false>, dim3(p.blocks), dim3(kThreads), lds, s, src, dst, partial, total, C, p.tact,
                               p.span, (long long)ld, c_off, UpShuffle{});
    }
    hipLaunchKernelGGL(channel_sum_finalize_kernel, dim3(C), dim3(kFinThreads), 0, s, partial, p.blocks, sum_out);
    return check_launch(who);
}

extern "C" int nextou_narrow_copy_sum(const float* src, float* dst, float* sum_out, void* ws, size_t ws_bytes, int64_t P, int C,
                                      int64_t ld, int c_off, nextou_stream_t stream) {
    return narrow_copy_sum_impl("narrow_copy_sum", src, dst, sum_out, ws, ws_bytes, P, C, ld, c_off, nullptr, (hipStream_t)stream);
}

extern "C" int nextou_upconv_cat_rows_bwd(const float* g, float* gy2, float* gbias, void* ws, size_t ws_bytes, int B, int D, int H, int W,
                                          int sd, int sh, int sw, int C1, int C2, nextou_stream_t stream) {
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && sd >= 1 && sh >= 1 && sw >= 1 && sd <= 4 && sh <= 4 && sw <= 4 && C1 > 0 && C2 >= 0,
                   "upconv_cat_rows_bwd: bad size B=%d (%d,%d,%d) stride (%d,%d,%d) C %d+%d", B, D, H, W, sd, sh, sw, C1, C2);
    const UpShuffle u{D * sd, H * sh, W * sw, sd, sh, sw};
    const int64_t P = (int64_t)B * u.D2 * u.H2 * u.W2;
    return narrow_copy_sum_impl("upconv_cat_rows_bwd", g, gy2, gbias, ws, ws_bytes, P, C1, (int64_t)C1 + C2, 0, &u, (hipStream_t)stream);
}
