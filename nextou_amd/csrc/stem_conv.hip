// K9 — the network's first block in one pass each way: convolution of the ONE-channel image (kernel 1x3x3 in 3-D, 3x3 in 2-D, stride 1,
// zero padding 1) -> batch norm -> LeakyReLU.  Reference: NexToU_Encoder_Decoder.py:125-141 (encoder.stages[0]'s first
// ConvDropoutNormReLU: conv_op(input_channels, features[0], kernel_sizes[0], 1, bias) -> norm_op -> nonlin), the autograd of the three.
//
// Why own kernels: the library runs this layer as a GEMM with K = 9 (forward 466 us = 4 % of the MFMA peak, weight
// gradient 2 342 us at cfg 2) and the block around it moves the 881-MB convolution output z = conv(x) through HBM six times (conv write,
// statistics read, apply read, backward reduce read, backward apply read + dz write, weight gradient read) although z is a function of
// nine numbers per voxel.  Here z never exists in memory:
//   forward   (1) stem_moments_kernel   one pass over the image (22 MB): X1[t] = sum_v x_t(v), A[t][t'] = sum_v x_t(v) x_t'(v) over the
//                 nine taps, float64.  The batch statistics of every output channel follow from them by linearity —
//                 sum_v z_c = sum_t w_ct X1[t],  sum_v z_c^2 = sum_tt' w_ct w_ct' A[t][t'] — (2) stem_stats_kernel, then K6's own
//                 finalize (nextou_norm_finalize: mean / invstd / running statistics / folded conv bias);
//             (3) stem_apply_kernel     y[v][c] = leaky(fmaf(z_c(v), scale_c, shift_c)), z_c(v) = the fp32 fma chain over the taps in
//                 ascending tap order; reads the image, writes the channels-last rows once (zero padding channels included).
//   backward  (4) stem_bwd_kernel       one pass over gy (+ the image): dy' = gy * (pre-activation > 0 ? 1 : slope) — the mask is one byte per
//                 (voxel, channel quad) written by (3) — S1[c] = sum dy', S2[c][t] = sum dy' x_t;
//             (5) stem_bwd_finalize_kernel  everything the three ops' autograd returns is linear in (S1, S2) given the moments:
//                 gbeta = S1, ggamma = sum dy' zhat = invstd (sum_t w_ct S2[c][t] - mean S1),
//                 gw[c][t] = scale ( S2[c][t] - (S1/n) X1[t] - (ggamma/n) invstd (sum_t' w_ct' A[t'][t] - mean X1[t]) ).
//                 The image needs no gradient (the caller checks), the folded conv bias gets exactly zero under batch statistics.
// HBM-bound: V (4 + 4 Cpad + Cpad / 8) bytes each way for V voxels.  Sums: fp32 over a thread's own voxels (<= a few
// hundred terms), float64 across threads and workgroups in a fixed order: bit-reproducible.
#include "common.h"
#include <cstdlib>

namespace nextou {
namespace {

constexpr int kTaps = 9;
constexpr int kMom = kTaps + kTaps * (kTaps + 1) / 2;      // X1[9] + upper triangle of A (45)
constexpr int kVox = 32;                                   // voxels of a row chunk per workgroup iteration
constexpr int kMomThreads = 256;
constexpr int kRB = 8;                                     // image rows per batch of the apply / backward kernels = rows per mask dword (4 bits each)

__host__ __device__ constexpr int tri(int a, int b) {      // packed index of A[a][b], a <= b
    return kTaps + a * kTaps - a * (a - 1) / 2 + (b - a);
}

__device__ __forceinline__ float leaky(float v, float slope) { return v > 0.f ? v : v * slope; }

// The nine taps of voxel (row r, column w) are t[ky * 3 + kx] = x[h + ky - 1][w + kx - 1], zero outside the plane (cross-correlation, as
// ATen).  Every streaming kernel below walks DOWN the image rows at a fixed column, so the 3 x 3 window slides: per voxel three new values
// (row r + 2, one iteration ahead of their use) instead of nine.  All loads are UNCONDITIONAL on clamped coordinates (the image is dense, a
// clamped neighbour is always inside it) and the zero padding is a select at the use: written as `cond ? p[i] : 0` the compiler turned
// every load into a branch with a wait — nine dependent round trips per voxel (the first version: 314 / 446 / 129 us where HBM allows 150; batches of 8 rows: apply 236 -> 210 us against batches of 4).
struct Cols { int l, c, r; };                                       // clamped column indices of (w - 1, w, w + 1)
__device__ __forceinline__ Cols clamp_cols(int w, int W) {
    const int wc = min(w, W - 1);
    return Cols{max(wc - 1, 0), wc, min(wc + 1, W - 1)};
}
__device__ __forceinline__ void load_row3(const float* __restrict__ x, int r, int R, int W, const Cols& k, float (&o)[3]) {
    const float* __restrict__ p = x + (long long)min(max(r, 0), R - 1) * W;
    o[0] = p[k.l];
    o[1] = p[k.c];
    o[2] = p[k.r];
}
// window rows (r - 1, r, r + 1) -> taps; h = r mod H decides the vertical padding, (w, W) the horizontal one
__device__ __forceinline__ void window_taps(const float (&u)[3], const float (&m)[3], const float (&d)[3], int w, int W, bool up, bool down,
                                            bool active, float (&t)[kTaps]) {
    const bool l = active && w > 0, r = active && w + 1 < W;
    t[0] = (up && l) ? u[0] : 0.f;
    t[1] = (up && active) ? u[1] : 0.f;
    t[2] = (up && r) ? u[2] : 0.f;
    t[3] = l ? m[0] : 0.f;
    t[4] = active ? m[1] : 0.f;
    t[5] = r ? m[2] : 0.f;
    t[6] = (down && l) ? d[0] : 0.f;
    t[7] = (down && active) ? d[1] : 0.f;
    t[8] = (down && r) ? d[2] : 0.f;
}

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// (1) moments of the taps.  A workgroup = 256 columns x a segment of image rows; every thread keeps the 54 sums of its own voxels in float64
// (products of two floats are exact in float64).
__global__ __launch_bounds__(kMomThreads) void stem_moments_kernel(const float* __restrict__ x, double* __restrict__ partial, int R, int H, int W,
                                                                   int rows_per_seg, int col_blocks) {
    double acc[kMom];
#pragma unroll
    for (int i = 0; i < kMom; ++i) acc[i] = 0.0;
    const int cb = blockIdx.x % col_blocks, seg = blockIdx.x / col_blocks;
    const int w = cb * kMomThreads + threadIdx.x;
    const bool active = w < W;
    const Cols k = clamp_cols(w, W);
    const int r0 = seg * rows_per_seg, r1 = min(R, r0 + rows_per_seg);
    constexpr int RB = 2;                                   // image rows per iteration: their RB + 2 window rows are loaded as ONE batch
    int h = r0 % H;
    for (int r = r0; r < r1; r += RB) {
        float rows[RB + 2][3];
#pragma unroll
        for (int i = 0; i < RB + 2; ++i) load_row3(x, r - 1 + i, R, W, k, rows[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float t[kTaps];
            window_taps(rows[i], rows[i + 1], rows[i + 2], w, W, h > 0, h + 1 < H, active && r + i < r1, t);
            h = h + 1 == H ? 0 : h + 1;
            double dd[kTaps];
#pragma unroll
            for (int a = 0; a < kTaps; ++a) { dd[a] = (double)t[a]; acc[a] += dd[a]; }
#pragma unroll
            for (int a = 0; a < kTaps; ++a)
#pragma unroll
                for (int b = a; b < kTaps; ++b) acc[tri(a, b)] = fma(dd[a], dd[b], acc[tri(a, b)]);
        }
    }
    __shared__ double red[kMomThreads / 64][kMom];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < kMom; ++i) {
        const double sum = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = sum;
    }
    __syncthreads();
    if (threadIdx.x < kMom) {
        double sum = 0.0;
#pragma unroll
        for (int wv = 0; wv < kMomThreads / 64; ++wv) sum += red[wv][threadIdx.x];
        partial[(size_t)blockIdx.x * kMom + threadIdx.x] = sum;
    }
}

// (2) moments -> per-channel (sum z, sum z^2): nextou_norm_finalize's `partial` with tiles = 1
constexpr int kStatsParts = 16;                            // waves of stem_stats_kernel, each summing every 16th workgroup's partial moments
__global__ __launch_bounds__(64 * kStatsParts) void stem_stats_kernel(const double* __restrict__ partial, int G, const float* __restrict__ weight, int C,
                                                                      double* __restrict__ moments, double2* __restrict__ stats) {
    __shared__ double part[kStatsParts][kMom];
    __shared__ double M[kMom];
    const int t = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (t < kMom) {
        // eight loads in flight per lane (four waves walking G / 4 partials one dependent load at a time took 58 us for ~1 000 workgroups);
        // the order of the sum is fixed by (wave, position): bit-reproducible
        double s = 0.0;
        int g = wave;
        for (; g + 7 * kStatsParts < G; g += 8 * kStatsParts) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(size_t)(g + u * kStatsParts) * kMom + t];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; g < G; g += kStatsParts) s += partial[(size_t)g * kMom + t];
        part[wave][t] = s;
    }
    __syncthreads();
    if (threadIdx.x < kMom) {
        double s = 0.0;
#pragma unroll
        for (int u = 0; u < kStatsParts; ++u) s += part[u][threadIdx.x];
        M[threadIdx.x] = s;
        moments[threadIdx.x] = s;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 64 * kStatsParts) {
        double w[kTaps];
#pragma unroll
        for (int a = 0; a < kTaps; ++a) w[a] = (double)weight[c * kTaps + a];
        double s = 0.0, q = 0.0;
#pragma unroll
        for (int a = 0; a < kTaps; ++a) {
            s = fma(w[a], M[a], s);
            q = fma(w[a] * w[a], M[tri(a, a)], q);
#pragma unroll
            for (int b = a + 1; b < kTaps; ++b) q = fma(2.0 * w[a] * w[b], M[tri(a, b)], q);
        }
        stats[c] = make_double2(s, q);
    }
}

struct Affine {          // a thread's four channels: filter taps + K6's affine (scale = gamma * invstd, shift = fmaf(-mean, scale, beta))
    float w[4][kTaps], scale[4], shift[4];
};

__device__ __forceinline__ void load_affine(Affine& a, int c0, int C, const float* __restrict__ weight, const float* __restrict__ gamma,
                                            const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const bool real = c0 + j < C;
        const int c = min(c0 + j, C - 1);           // (unconditional loads on a clamped channel, then a select: no branch per load)
#pragma unroll
        for (int t = 0; t < kTaps; ++t) { const float wv = weight[c * kTaps + t]; a.w[j][t] = real ? wv : 0.f; }
        const float gv = gamma ? gamma[c] : 1.f, bv = beta ? beta[c] : 0.f, mv = mean[c], iv = invstd[c];
        const float sc = real ? gv * iv : 0.f;
        a.scale[j] = sc;
        a.shift[j] = real ? fmaf(-mv, sc, bv) : 0.f;
    }
}

// the convolution value itself: one fp32 fma chain over the taps in ascending order (forward and backward share it bit for bit)
__device__ __forceinline__ float conv_chain(const float (&w)[kTaps], const float (&t)[kTaps]) {
    float z = w[0] * t[0];
#pragma unroll
    for (int k = 1; k < kTaps; ++k) z = fmaf(w[k], t[k], z);
    return z;
}

// (3) y rows.  A workgroup = kVox columns x Q channel quads, walking down a segment of image rows with the sliding window.
__global__ __launch_bounds__(kVox * 12) void stem_apply_kernel(const float* __restrict__ x, const float* __restrict__ weight, const float* __restrict__ gamma,
                                  const float* __restrict__ beta, const float* __restrict__ mean, const float* __restrict__ invstd,
                                  float* __restrict__ y, uint32_t* __restrict__ act, int R, int H, int W, int C, int Q, int rows_per_seg,
                                  int chunks, float slope) {
    const int q = threadIdx.x % Q, vs = threadIdx.x / Q;
    Affine a;
    load_affine(a, 4 * q, C, weight, gamma, beta, mean, invstd);
    const int Cp = 4 * Q;
    const int chunk = blockIdx.x % chunks, seg = blockIdx.x / chunks;
    const int w = chunk * kVox + vs;
    const bool active = w < W;
    const Cols k = clamp_cols(w, W);
    const int r0 = seg * rows_per_seg, r1 = min(R, r0 + rows_per_seg);
    constexpr int RB = kRB;                                 // image rows per iteration: RB + 2 window rows loaded as ONE batch,
    int h = r0 % H;                                         // so the memory latency is paid once per RB voxels (no cross-iteration register copies)
    float* yp = y + ((long long)r0 * W + k.c) * Cp + 4 * q;
    const long long ystep = (long long)W * Cp;
    // the LeakyReLU mask for the backward: one dword per (block of RB = 8 image rows, column, quad), bit 4 i + j = pre-activation of channel 4q + j
    // at row 4 block + i > 0 — ONE dword store per batch (a byte per voxel cost 70 us, 2-byte words were no better: sub-dword accesses are the slow kind); segments start at
    // multiples of 8 rows (seg_plan), so blocks never straddle workgroups
    uint32_t* mp = act ? act + ((long long)(r0 / kRB) * W + k.c) * Q + q : nullptr;
    for (int r = r0; r < r1; r += RB) {
        float rows[RB + 2][3];
        unsigned bits = 0;
#pragma unroll
        for (int i = 0; i < RB + 2; ++i) load_row3(x, r - 1 + i, R, W, k, rows[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float t[kTaps];
            window_taps(rows[i], rows[i + 1], rows[i + 2], w, W, h > 0, h + 1 < H, active, t);
            h = h + 1 == H ? 0 : h + 1;
            const float p0 = fmaf(conv_chain(a.w[0], t), a.scale[0], a.shift[0]), p1 = fmaf(conv_chain(a.w[1], t), a.scale[1], a.shift[1]);
            const float p2 = fmaf(conv_chain(a.w[2], t), a.scale[2], a.shift[2]), p3 = fmaf(conv_chain(a.w[3], t), a.scale[3], a.shift[3]);
            const float4 o = make_float4(leaky(p0, slope), leaky(p1, slope), leaky(p2, slope), leaky(p3, slope));
            if (active && r + i < r1) *reinterpret_cast<float4*>(yp + i * ystep) = o;
            bits |= ((p0 > 0.f ? 1u : 0u) | (p1 > 0.f ? 2u : 0u) | (p2 > 0.f ? 4u : 0u) | (p3 > 0.f ? 8u : 0u)) << (4 * i);
        }
        yp += RB * ystep;
        if (mp) {
            if (active) *mp = bits;
            mp += (long long)W * Q;
        }
    }
}

// (4) S1 / S2 partial sums: the same walk over the gradient rows; the activation's mask comes from the forward's byte per (voxel, quad)
// (recomputing the pre-activation here cost 40 fma + 36 weight registers per thread: the kernel was instruction-bound at 2.5 waves per SIMD).
__global__ __launch_bounds__(kVox * 12) void stem_bwd_kernel(const float* __restrict__ x, const float* __restrict__ gy, const uint32_t* __restrict__ act,
                                double* __restrict__ partial, int R, int H, int W, int Q, int rows_per_seg, int chunks, float slope) {
    __shared__ float red[kVox * 12 * (kTaps + 1)];          // one channel of every quad at a time: [vs][q][10]
    const int q = threadIdx.x % Q, vs = threadIdx.x / Q;
    const int Cp = 4 * Q;
    float s1[4], s2[4][kTaps];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        s1[j] = 0.f;
#pragma unroll
        for (int t = 0; t < kTaps; ++t) s2[j][t] = 0.f;
    }
    const int chunk = blockIdx.x % chunks, seg = blockIdx.x / chunks;
    const int w = chunk * kVox + vs;
    const bool active = w < W;
    const Cols k = clamp_cols(w, W);
    const int r0 = seg * rows_per_seg, r1 = min(R, r0 + rows_per_seg);
    constexpr int RB = kRB;                                 // as in stem_apply_kernel: RB + 2 window rows, RB gradient pieces and one mask dword per batch
    const long long gstep = (long long)W * Cp;
    const float* gp = gy + ((long long)r0 * W + k.c) * Cp + 4 * q;
    const uint32_t* mp = act + ((long long)(r0 / kRB) * W + k.c) * Q + q;
    int h = r0 % H;
    for (int r = r0; r < r1; r += RB) {
        float rows[RB + 2][3];
        float4 g[RB];
#pragma unroll
        for (int i = 0; i < RB + 2; ++i) load_row3(x, r - 1 + i, R, W, k, rows[i]);
#pragma unroll
        for (int i = 0; i < RB; ++i) g[i] = *reinterpret_cast<const float4*>(gp + (r + i < r1 ? i : 0) * gstep);     // (clamped: a row past the segment re-reads row r)
        const unsigned mk = *mp;
        gp += RB * gstep;
        mp += (long long)W * Q;
#pragma unroll
        for (int i = 0; i < RB; ++i) {
            float t[kTaps];
            const bool on = active && r + i < r1;
            window_taps(rows[i], rows[i + 1], rows[i + 2], w, W, h > 0, h + 1 < H, on, t);
            h = h + 1 == H ? 0 : h + 1;
            const float gj[4] = {g[i].x, g[i].y, g[i].z, g[i].w};
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float dy = (mk >> (4 * i + j)) & 1u ? gj[j] : gj[j] * slope;
                dy = on ? dy : 0.f;
                s1[j] += dy;
#pragma unroll
                for (int kk = 0; kk < kTaps; ++kk) s2[j][kk] = fmaf(dy, t[kk], s2[j][kk]);
            }
        }
    }
    // partial[(4q + j) * 10 + kk]: kk = 0 -> S1, 1 + t -> S2[t]; the workgroup's kVox column lanes are summed in float64, in lane order
    const int per = Cp * (kTaps + 1), qw = Q * (kTaps + 1);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        __syncthreads();
        float* mine = red + vs * qw + q * (kTaps + 1);
        mine[0] = s1[j];
#pragma unroll
        for (int kk = 0; kk < kTaps; ++kk) mine[1 + kk] = s2[j][kk];
        __syncthreads();
        for (int i = threadIdx.x; i < qw; i += blockDim.x) {
            double sum = 0.0;
#pragma unroll 8
            for (int v = 0; v < kVox; ++v) sum += (double)red[v * qw + i];
            const int qq = i / (kTaps + 1), kk = i - qq * (kTaps + 1);
            partial[(size_t)blockIdx.x * per + (size_t)(4 * qq + j) * (kTaps + 1) + kk] = sum;
        }
    }
}

// (5) one wave per real channel: totals in a fixed order, then the parameter gradients in float64
__global__ __launch_bounds__(64) void stem_bwd_finalize_kernel(const double* __restrict__ partial, int G, int Cp, const float* __restrict__ weight,
                                                               const float* __restrict__ gamma, const float* __restrict__ mean,
                                                               const float* __restrict__ invstd, const double* __restrict__ moments, double count,
                                                               float* __restrict__ gweight, float* __restrict__ ggamma, float* __restrict__ gbeta) {
    const int c = blockIdx.x, lane = threadIdx.x;
    const int per = Cp * (kTaps + 1);
    double tot[kTaps + 1];
#pragma unroll
    for (int k = 0; k <= kTaps; ++k) tot[k] = 0.0;
    for (int g = lane; g < G; g += 64) {
        const double* p = partial + (size_t)g * per + (size_t)c * (kTaps + 1);
#pragma unroll
        for (int k = 0; k <= kTaps; ++k) tot[k] += p[k];
    }
#pragma unroll
    for (int k = 0; k <= kTaps; ++k) tot[k] = wave_sum(tot[k]);
    const double S1 = tot[0];
    double w[kTaps];
#pragma unroll
    for (int t = 0; t < kTaps; ++t) w[t] = (double)weight[c * kTaps + t];
    const double m = (double)mean[c], is = (double)invstd[c];
    const double scale = (gamma ? (double)gamma[c] : 1.0) * is;
    double wS2 = 0.0;
#pragma unroll
    for (int t = 0; t < kTaps; ++t) wS2 = fma(w[t], tot[1 + t], wS2);
    const double dot = is * (wS2 - m * S1);            // sum dy' * zhat
    if (lane == 0) {
        if (ggamma) ggamma[c] = (float)dot;
        if (gbeta) gbeta[c] = (float)S1;
    }
    if (lane < kTaps && gweight) {
        const int t = lane;
        double wA = 0.0;
        for (int u = 0; u < kTaps; ++u) wA = fma(w[u], moments[u <= t ? tri(u, t) : tri(t, u)], wA);
        const double zx = is * (wA - m * moments[t]);   // sum_v zhat x_t
        double s2t = 0.0;                               // tot[1 + t] with a runtime t: select without indexing registers dynamically
#pragma unroll
        for (int u = 0; u < kTaps; ++u) s2t = (u == t) ? tot[1 + u] : s2t;
        gweight[c * kTaps + t] = (float)(scale * (s2t - (S1 / count) * moments[t] - (dot / count) * zx));
    }
}

struct SegPlan { int blocks_w, segs, rows_per_seg, grid; };
// a grid of (column blocks) x (row segments) with about `target` workgroups; segments of at least 8 rows (the window's warm-up is 3 loads)
SegPlan seg_plan(int R, int W, int cols_per_wg, int target) {
    SegPlan p;
    p.blocks_w = (W + cols_per_wg - 1) / cols_per_wg;
    int segs = target / p.blocks_w;
    if (segs > (R + 7) / 8) segs = (R + 7) / 8;
    if (segs < 1) segs = 1;
    p.rows_per_seg = ((R + segs - 1) / segs + kRB - 1) / kRB * kRB;      // multiples of kRB: the kernels' row blocks (and the mask's 8-row words) never straddle segments
    p.segs = (R + p.rows_per_seg - 1) / p.rows_per_seg;
    p.grid = p.segs * p.blocks_w;
    return p;
}
SegPlan moments_plan(int R, int W) { return seg_plan(R, W, kMomThreads, 1024); }
SegPlan apply_plan(int R, int W) { return seg_plan(R, W, kVox, 3072); }
SegPlan bwd_plan(int R, int W) { return seg_plan(R, W, kVox, 1536); }

}  // namespace
}  // namespace nextou

using namespace nextou;

extern "C" size_t nextou_stem_workspace_bytes(int B, int D, int H, int W, int Cpad) {
    if (B <= 0 || D <= 0 || H <= 0 || W <= 0 || Cpad <= 0) return 0;
    const long long R64 = (long long)B * D * H;
    if (R64 >= INT32_MAX) return 0;
    const int R = (int)R64;
    const size_t fwd = (size_t)moments_plan(R, W).grid * kMom * sizeof(double) + (size_t)Cpad * sizeof(double2);
    const size_t bwd = (size_t)bwd_plan(R, W).grid * Cpad * (kTaps + 1) * sizeof(double);
    return fwd > bwd ? fwd : bwd;
}

extern "C" int nextou_stem_fwd(const float* x, const float* weight, const float* pre_bias, const float* gamma, const float* beta,
                               float* running_mean, float* running_var, float* y, uint32_t* act_mask, float* save_mean, float* save_invstd,
                               double* moments, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, int Cpad, int training,
                               float momentum, float eps, float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && weight && y && save_mean && save_invstd, "stem_fwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0 && Cpad <= 48, "stem_fwd: 0 < C <= Cpad <= 48, Cpad a multiple of 4");
    NEXTOU_REQUIRE((long long)B * D * H < INT32_MAX, "stem_fwd: more than 2^31 image rows");
    NEXTOU_REQUIRE(((uintptr_t)y & 15) == 0, "stem_fwd: y must be 16-byte aligned");
    NEXTOU_REQUIRE(!training || moments, "stem_fwd: training needs the moments output (the backward reads it)");
    NEXTOU_REQUIRE(!training || workspace_bytes >= nextou_stem_workspace_bytes(B, D, H, W, Cpad), "stem_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long V = (long long)B * D * H * W;
    const int R = B * D * H;
    if (training) {
        const SegPlan mp = moments_plan(R, W);
        const int G = mp.grid;
        double* partial = static_cast<double*>(workspace);
        double2* stats = reinterpret_cast<double2*>(partial + (size_t)G * kMom);
        {
            ProfScope prof(s, kBoundHbm, 4.0 * (double)V, "stem_moments_kernel[B%d S%lld]", B, V / B);
            hipLaunchKernelGGL(stem_moments_kernel, dim3(G), dim3(kMomThreads), 0, s, x, partial, R, H, W, mp.rows_per_seg, mp.blocks_w);
        }
        hipLaunchKernelGGL(stem_stats_kernel, dim3(1), dim3(64 * kStatsParts), 0, s, partial, G, weight, C, moments, stats);
        int rc = check_launch("stem_moments_kernel");
        if (rc) return rc;
        rc = nextou_norm_finalize(reinterpret_cast<const double*>(stats), 1, (double)V, pre_bias, running_mean, running_var, save_mean,
                                  save_invstd, nullptr, nullptr, nullptr, nullptr, C, 1, momentum, eps, stream);
        if (rc) return rc;
    } else {
        const int rc = nextou_norm_finalize(nullptr, 0, (double)V, pre_bias, running_mean, running_var, save_mean, save_invstd, nullptr,
                                            nullptr, nullptr, nullptr, C, 0, momentum, eps, stream);
        if (rc) return rc;
    }
    const int Q = Cpad / 4;
    const SegPlan ap = apply_plan(R, W);
    ProfScope prof(s, kBoundHbm, (double)V * (4.0 + 4.0 * Cpad + (act_mask ? 0.5 * Q : 0.0)), "stem_apply_kernel[B%d C%d S%lld]", B, Cpad, V / B);
    hipLaunchKernelGGL(stem_apply_kernel, dim3(ap.grid), dim3(kVox * Q), 0, s, x, weight, gamma, beta, save_mean, save_invstd, y, act_mask, R, H, W,
                       C, Q, ap.rows_per_seg, ap.blocks_w, slope);
    return check_launch("stem_apply_kernel");
}

extern "C" int nextou_stem_bwd(const float* x, const float* gy, const uint32_t* act_mask, const float* weight, const float* gamma,
                               const float* save_mean, const float* save_invstd, const double* moments, float* gweight, float* ggamma,
                               float* gbeta, void* workspace, size_t workspace_bytes, int B, int D, int H, int W, int C, int Cpad,
                               float slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && gy && act_mask && weight && save_mean && save_invstd && moments && workspace, "stem_bwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && C > 0 && Cpad >= C && Cpad % 4 == 0 && Cpad <= 48, "stem_bwd: 0 < C <= Cpad <= 48, Cpad a multiple of 4");
    NEXTOU_REQUIRE((long long)B * D * H < INT32_MAX, "stem_bwd: more than 2^31 image rows");
    NEXTOU_REQUIRE(((uintptr_t)gy & 15) == 0, "stem_bwd: gy must be 16-byte aligned");
    NEXTOU_REQUIRE(workspace_bytes >= nextou_stem_workspace_bytes(B, D, H, W, Cpad), "stem_bwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    const long long V = (long long)B * D * H * W;
    const int R = B * D * H;
    const int Q = Cpad / 4;
    const SegPlan bp = bwd_plan(R, W);
    const int G = bp.grid;
    double* partial = static_cast<double*>(workspace);
    {
        ProfScope prof(s, kBoundHbm, (double)V * (4.0 + 4.0 * Cpad + 0.5 * Q), "stem_bwd_kernel[B%d C%d S%lld]", B, Cpad, V / B);
        hipLaunchKernelGGL(stem_bwd_kernel, dim3(G), dim3(kVox * Q), 0, s, x, gy, act_mask, partial, R, H, W, Q, bp.rows_per_seg, bp.blocks_w, slope);
    }
    int rc = check_launch("stem_bwd_kernel");
    if (rc) return rc;
    hipLaunchKernelGGL(stem_bwd_finalize_kernel, dim3(C), dim3(64), 0, s, partial, G, Cpad, weight, gamma, save_mean, save_invstd, moments,
                       (double)V, gweight, ggamma, gbeta);
    return check_launch("stem_bwd_finalize_kernel");
}
