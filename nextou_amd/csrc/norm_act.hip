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
#include <hip/hip_bf16.h>

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
template <> struct Pack<float, 4> {
    float v[4];
    __device__ void load(const float* p) { const float4 t = *reinterpret_cast<const float4*>(p); v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; }
    __device__ void store(float* p) const { *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]); }
};
template <> struct Pack<float, 1> {
    float v[1];
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
    __device__ void load(const __hip_bfloat16* p) { v[0] = bf16_to_f32(*reinterpret_cast<const unsigned short*>(p)); }
    __device__ void store(__hip_bfloat16* p) const { *reinterpret_cast<unsigned short*>(p) = f32_to_bf16(v[0]); }
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
                                                            int training, float momentum, float eps, float slope) {
    const int c = C - 1 - blockIdx.y;
    double var;
    const ChannelAffine a = channel_affine(partial, gridDim.x, c, wmod, count, weight, bias, pre_bias, running_mean,
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
};

template <typename T, int VEC>
void launch_fwd(const NormArgs& a, const TilePlan& p, hipStream_t s, const char* tname) {
    const dim3 grid(p.tiles, a.C);
    const double bytes = (double)a.B * a.C * (double)a.S * sizeof(T);
    const double count = (double)a.B * (double)a.S;
    if (a.training) {
        ProfScope prof(s, kBoundHbm, bytes, "bn_stats_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
        hipLaunchKernelGGL((bn_stats_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, a.partial, a.B, a.C, p.cols,
                           p.row_len, p.col_len, p.col_tiles, p.tw_log2);
    }
    ProfScope prof(s, kBoundHbm, 2.0 * bytes, "bn_apply_kernel<%s>[B%d C%d S%lld]", tname, a.B, a.C, a.S);
    hipLaunchKernelGGL((bn_apply_kernel<T, VEC>), grid, dim3(kThreads), 0, s, (const T*)a.x, (T*)a.y, a.partial, a.weight,
                       a.bias, a.pre_bias, a.running_mean, a.running_var, a.save_mean, a.save_invstd, a.B, a.C, p.cols, p.row_len,
                       p.col_len, p.col_tiles, p.tw_log2, a.wmod, count, a.training, a.momentum, a.eps, a.slope);
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
    } else {
        if (p.vec == 8) FWD ? launch_fwd<__hip_bfloat16, 8>(a, p, s, "bf16") : launch_bwd<__hip_bfloat16, 8>(a, p, s, "bf16");
        else FWD ? launch_fwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar") : launch_bwd<__hip_bfloat16, 1>(a, p, s, "bf16,scalar");
    }
}

static int check_common(const char* what, int B, int C, int64_t S, int param_period, int dtype) {
    NEXTOU_REQUIRE(B > 0 && C > 0 && C <= 65535 && S > 0, "%s: bad size B=%d C=%d S=%lld", what, B, C, (long long)S);
    NEXTOU_REQUIRE(dtype == NEXTOU_DTYPE_F32 || dtype == NEXTOU_DTYPE_BF16, "%s: dtype %d not in {f32, bf16}", what, dtype);
    NEXTOU_REQUIRE(param_period >= 0, "%s: param_period=%d", what, param_period);
    return 0;
}

}  // namespace nextou

using namespace nextou;

extern "C" size_t nextou_norm_act_workspace_bytes(int B, int C, int64_t S, int dtype) {
    (void)B; (void)S; (void)dtype;
    if (C <= 0) return 0;
    return (size_t)C * 1024 * sizeof(double2);  // plan_tiles never cuts a channel into more than 1024 tiles
}

extern "C" int nextou_norm_act_fwd(const void* x, const float* weight, const float* bias, const float* pre_bias,
                                   float* running_mean, float* running_var, void* y, float* save_mean, float* save_invstd, void* ws,
                                   size_t ws_bytes, int B, int C, int64_t S, int param_period, int dtype, int training,
                                   float momentum, float eps, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && y, "norm_act_fwd: null pointer");
    if (int rc = check_common("norm_act_fwd", B, C, S, param_period, dtype)) return rc;
    NEXTOU_REQUIRE((running_mean == nullptr) == (running_var == nullptr), "norm_act_fwd: running_mean / running_var must come together");
    NEXTOU_REQUIRE(training || running_mean, "norm_act_fwd: inference needs the running statistics");
    const int esz = dtype == NEXTOU_DTYPE_BF16 ? 2 : 4;
    const TilePlan p = plan_tiles(B, C, S, 16 / esz, aligned16(x) && aligned16(y));
    if (training) {
        NEXTOU_REQUIRE(ws, "norm_act_fwd: null workspace");
        if (ws_bytes < (size_t)C * p.tiles * sizeof(double2))
            return fail(NEXTOU_ENOSPACE, "norm_act_fwd: workspace %zu < %zu bytes", ws_bytes, (size_t)C * p.tiles * sizeof(double2));
    }
    NormArgs a{};
    a.x = x; a.y = y; a.weight = weight; a.bias = bias; a.pre_bias = pre_bias; a.running_mean = running_mean; a.running_var = running_var;
    a.save_mean = save_mean; a.save_invstd = save_invstd; a.partial = (double2*)ws; a.B = B; a.C = C; a.S = S;
    a.wmod = param_period; a.training = training; a.momentum = momentum; a.eps = eps; a.slope = slope;
    norm_dispatch<true>(a, p, dtype, (hipStream_t)stream);
    return check_launch("bn_apply_kernel");
}

extern "C" int nextou_norm_act_bwd(const void* x, const void* gy, const float* weight, const float* bias,
                                   const float* save_mean, const float* save_invstd, void* gx, float* gweight,
                                   float* gbias, void* ws, size_t ws_bytes, int B, int C, int64_t S, int param_period,
                                   int dtype, int training, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && gy && gx && save_mean && save_invstd && ws, "norm_act_bwd: null pointer");
    if (int rc = check_common("norm_act_bwd", B, C, S, param_period, dtype)) return rc;
    const int esz = dtype == NEXTOU_DTYPE_BF16 ? 2 : 4;
    const TilePlan p = plan_tiles(B, C, S, 16 / esz, aligned16(x) && aligned16(gy) && aligned16(gx));
    if (ws_bytes < (size_t)C * p.tiles * sizeof(double2))
        return fail(NEXTOU_ENOSPACE, "norm_act_bwd: workspace %zu < %zu bytes", ws_bytes, (size_t)C * p.tiles * sizeof(double2));
    NormArgs a{};
    a.x = x; a.gy = gy; a.gx = gx; a.weight = weight; a.bias = bias;
    a.save_mean = const_cast<float*>(save_mean); a.save_invstd = const_cast<float*>(save_invstd);
    a.gweight = gweight; a.gbias = gbias; a.partial = (double2*)ws; a.B = B; a.C = C; a.S = S; a.wmod = param_period;
    a.training = training; a.slope = slope;
    norm_dispatch<false>(a, p, dtype, (hipStream_t)stream);
    return check_launch("bn_bwd_apply_kernel");
}
