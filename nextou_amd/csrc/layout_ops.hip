// K3 / K4 — the data movement around the graph kernels, for gfx950: window shift + partition / reverse, and the
// query max-pool / max-unpool of the Pool-GNN blocks, each fused with the layout change between the dense stages'
// channels-last activations (B, D, H, W, C — the layout MIOpen's CK convolutions run in natively) and the graph kernels'
// channel-major rows (B', C, N).
//
// Reference op sequences replaced (network_architecture/NexToU_Encoder_Decoder.py):
//   window_gather   torch.roll(x, -shift) -> window_partition (einops rearrange + contiguous)          :781-790, :634-660
//   window_scatter  window_reverse -> torch.roll(x, +shift) -> drop_path(x) + shortcut                :807-817, :662-693
//   pool_rows       MaxPool{2,3}d(pool, return_indices=True)                                          :524-530
//   cell_scatter    MaxUnpool{2,3}d(out, cat(indices, indices))  (182 MB of mostly zeros at cfg-2 s2) :536-549
//   cell_gather     the backward of the unpool / the teacher-forced replay of recorded arg-max cells
// Every element has to move once anyway; doing the index math here removes the roll copies, the permute copies, the
// index concatenation, the zero fill and the NCDHW <-> NDHWC transposes that otherwise wrap every convolution of the
// graph stages.  Bound: HBM (pure data movement); algorithmic bytes = one read + one write of the tensor.
//
// Pattern of every kernel: a workgroup owns a tile of T points x Cc channels.  The channels-last side is accessed with
// lanes along the channels (consecutive addresses inside a row of C floats), the channel-major side with lanes along
// the points (consecutive addresses inside a row of N floats); the tile turns around in LDS with an odd row stride
// (conflict-free both ways).
#include "common.h"

namespace nextou {

constexpr int kLayThreads = 256;
constexpr int kTileChannels = 64;   // channels per tile (LDS row = 65 floats)
constexpr int kTilePoints = 64;     // points per tile of the pool / unpool kernels

// (Vol, Win and window_point_row live in common.h: the window map is shared with mr_aggregate.hip)

// grid = (windows per sample, channel tiles, B); LDS = Nw * 65 floats (tile) + Nw ints (row of every window point)
// GATHER:  out_cm[(b * nWin + win), c, p] = x_cl[b, row(win, p), c]
// !GATHER: out_cl[b, row(win, p), c] = src_cm[(b * nWin + win), c, p] (+ residual_cl[b, row, c])
// VEC: C % 4 == 0 and 16-byte aligned bases — the channels-last side moves float4 (16 lanes cover the 256-byte run of a
// tile row); the channel-major side is walked one channel row per wave, lanes along the points.  No integer division in
// the loops: the window-point -> volume-row map is computed once per workgroup into LDS.
template <bool GATHER, bool VEC>
__global__ __launch_bounds__(kLayThreads) void window_move_kernel(const float* __restrict__ src, const float* __restrict__ residual,
                                                                  float* __restrict__ dst, Vol v, Win w, int C, int nH, int nW,
                                                                  int n_win, int Nw) {
    extern __shared__ float tile[];
    int* rows = reinterpret_cast<int*>(tile + (size_t)Nw * (kTileChannels + 1));
    const int win = blockIdx.x, b = blockIdx.z;
    const int c0 = blockIdx.y * kTileChannels;
    const int cc = min(kTileChannels, C - c0);
    const int ld = kTileChannels + 1;
    const long long vol_rows = (long long)v.D * v.H * v.W;
    const size_t cm_off = ((size_t)(b * n_win + win) * C + c0) * Nw;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = kLayThreads >> 6;
    for (int p = threadIdx.x; p < Nw; p += kLayThreads) rows[p] = (int)window_point_row(win, p, v, w, nH, nW);
    __syncthreads();
    const float* cl_in = GATHER ? src : residual;
    const size_t cl_base = (size_t)b * vol_rows * C + c0;
    if (GATHER) {
        if (VEC) {
            // batches of four 16-byte loads in flight per thread (one load per loop iteration left the workgroup waiting on ~10
            // dependent round trips: 0.47 of 8 TB/s at the stage-2 shape)
            constexpr int U = 4;
            const int total = Nw * 16;
            for (int e0 = threadIdx.x; e0 < total; e0 += U * kLayThreads) {
                float4 val[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = min(e0 + u * kLayThreads, total - 1);
                    const int p = e >> 4, c4 = min((e & 15) << 2, (cc - 1) & ~3);
                    val[u] = *reinterpret_cast<const float4*>(cl_in + cl_base + (size_t)rows[p] * C + c4);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int e = e0 + u * kLayThreads;
                    const int p = e >> 4, c4 = (e & 15) << 2;
                    if (e < total && c4 < cc) {
                        float* t = tile + p * ld + c4;
                        t[0] = val[u].x; t[1] = val[u].y; t[2] = val[u].z; t[3] = val[u].w;
                    }
                }
            }
        } else {
            for (int p = wave; p < Nw; p += n_waves)
                if (lane < cc) tile[p * ld + lane] = cl_in[cl_base + (size_t)rows[p] * C + lane];
        }
        __syncthreads();
        float* cm = dst + cm_off;
        for (int c = wave; c < cc; c += n_waves)
            for (int p = lane; p < Nw; p += 64) cm[(size_t)c * Nw + p] = tile[p * ld + c];
    } else {
        const float* cm = src + cm_off;
        // eight channel rows in flight per wave (see the gather side)
        constexpr int U = 8;
        for (int p = lane; p < Nw; p += 64) {
            for (int cb = wave; cb < cc; cb += U * n_waves) {
                float val[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = min(cb + u * n_waves, cc - 1);
                    val[u] = cm[(size_t)c * Nw + p];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int c = cb + u * n_waves;
                    if (c < cc) tile[p * ld + c] = val[u];
                }
            }
        }
        __syncthreads();
        if (VEC) {
            for (int e = threadIdx.x; e < Nw * 16; e += kLayThreads) {
                const int p = e >> 4, c4 = (e & 15) << 2;
                if (c4 < cc) {
                    const float* t = tile + p * ld + c4;
                    float4 val = make_float4(t[0], t[1], t[2], t[3]);
                    const size_t off = cl_base + (size_t)rows[p] * C + c4;
                    if (residual != nullptr) {
                        const float4 r = *reinterpret_cast<const float4*>(residual + off);
                        val.x += r.x; val.y += r.y; val.z += r.z; val.w += r.w;
                    }
                    *reinterpret_cast<float4*>(dst + off) = val;
                }
            }
        } else {
            for (int p = wave; p < Nw; p += n_waves)
                if (lane < cc) {
                    const size_t off = cl_base + (size_t)rows[p] * C + lane;
                    float val = tile[p * ld + lane];
                    if (residual != nullptr) val += residual[off];
                    dst[off] = val;
                }
        }
    }
}

// ------------------------------------------------------------------------------------------------------
// pooled points: n = (dz * H2 + hy) * W2 + wx over the pooled volume; cell k = (kd * ph + kh) * pw + kw
// ------------------------------------------------------------------------------------------------------
struct Pool {
    int pd, ph, pw, D2, H2, W2;
};
__device__ __forceinline__ long long cell_row(int n, int k, const Vol& v, const Pool& q) {
    const int wx = n % q.W2, t = n / q.W2;
    const int hy = t % q.H2, dz = t / q.H2;
    const int kw = k % q.pw, t2 = k / q.pw;
    const int kh = t2 % q.ph, kd = t2 / q.ph;
    return ((long long)(dz * q.pd + kd) * v.H + (hy * q.ph + kh)) * v.W + (wx * q.pw + kw);
}

// The three pool kernels share one tile walk: grid = (point tiles, channel tiles, B); the volume row of every
// (tile point, cell position) is computed once into LDS; VEC (C2 % 4 == 0, C % 4 == 0, aligned bases) moves float4 /
// uchar4 on the channels-last side with 16 lanes per point, the channel-major side goes one channel row per wave.
constexpr int kMaxCells = 8;

__device__ __forceinline__ void fill_cell_rows(int* rows, int n0, int tn, int cells, const Vol& v, const Pool& q) {
    for (int e = threadIdx.x; e < tn * cells; e += kLayThreads) {
        const int i = e / cells, k = e - i * cells;
        rows[e] = (int)cell_row(n0 + i, k, v, q);
    }
}

// Max pool of channels-last rows -> channel-major values + the winning cell (uint8, points-major (B, N, C)).
// First maximum in (d, h, w) scan order wins, NaN wins (ATen's max_pool3d_with_indices rule).
template <bool VEC>
__global__ __launch_bounds__(kLayThreads) void pool_rows_kernel(const float* __restrict__ x_cl, float* __restrict__ val_cm,
                                                                uint8_t* __restrict__ cell, Vol v, Pool q, int C, int N) {
    __shared__ float tile[kTilePoints * (kTileChannels + 1)];
    __shared__ int rows[kTilePoints * kMaxCells];
    const int b = blockIdx.z, c0 = blockIdx.y * kTileChannels, n0 = blockIdx.x * kTilePoints;
    const int cc = min(kTileChannels, C - c0), tn = min(kTilePoints, N - n0);
    const int ld = kTileChannels + 1, cells = q.pd * q.ph * q.pw;
    const long long vol_rows = (long long)v.D * v.H * v.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = kLayThreads >> 6;
    fill_cell_rows(rows, n0, tn, cells, v, q);
    __syncthreads();
    const float* base = x_cl + (size_t)b * vol_rows * C + c0;
    if (VEC) {
        for (int e = threadIdx.x; e < tn * 16; e += kLayThreads) {
            const int i = e >> 4, c4 = (e & 15) << 2;
            if (c4 >= cc) continue;
            const int* r = rows + i * cells;
            float4 best = *reinterpret_cast<const float4*>(base + (size_t)r[0] * C + c4);
            unsigned a0 = 0, a1 = 0, a2 = 0, a3 = 0;
            for (int k = 1; k < cells; ++k) {
                const float4 val = *reinterpret_cast<const float4*>(base + (size_t)r[k] * C + c4);
                if (val.x > best.x || val.x != val.x) { best.x = val.x; a0 = k; }
                if (val.y > best.y || val.y != val.y) { best.y = val.y; a1 = k; }
                if (val.z > best.z || val.z != val.z) { best.z = val.z; a2 = k; }
                if (val.w > best.w || val.w != val.w) { best.w = val.w; a3 = k; }
            }
            float* t = tile + i * ld + c4;
            t[0] = best.x; t[1] = best.y; t[2] = best.z; t[3] = best.w;
            *reinterpret_cast<unsigned*>(cell + ((size_t)b * N + n0 + i) * C + c0 + c4) = a0 | (a1 << 8) | (a2 << 16) | (a3 << 24);
        }
    } else {
        for (int i = wave; i < tn; i += n_waves) {
            if (lane >= cc) continue;
            const int* r = rows + i * cells;
            float best = base[(size_t)r[0] * C + lane];
            int arg = 0;
            for (int k = 1; k < cells; ++k) {
                const float val = base[(size_t)r[k] * C + lane];
                if (val > best || val != val) { best = val; arg = k; }
            }
            tile[i * ld + lane] = best;
            cell[((size_t)b * N + n0 + i) * C + c0 + lane] = (uint8_t)arg;
        }
    }
    __syncthreads();
    for (int c = wave; c < cc; c += n_waves)
        if (lane < tn) val_cm[((size_t)b * C + c0 + c) * N + n0 + lane] = tile[lane * ld + c];
}

// out_cm[b, c2, n] = x_cl[b, cell_row(n, cell[b, n, c2 mod C]), c2]      (C2 = C or 2C channels)
__global__ __launch_bounds__(kLayThreads) void cell_gather_kernel(const float* __restrict__ x_cl, const uint8_t* __restrict__ cell,
                                                                  float* __restrict__ out_cm, Vol v, Pool q, int C2, int C, int N) {
    __shared__ float tile[kTilePoints * (kTileChannels + 1)];
    __shared__ int rows[kTilePoints * kMaxCells];
    const int b = blockIdx.z, c0 = blockIdx.y * kTileChannels, n0 = blockIdx.x * kTilePoints;
    const int cc = min(kTileChannels, C2 - c0), tn = min(kTilePoints, N - n0);
    const int ld = kTileChannels + 1, cells = q.pd * q.ph * q.pw;
    const long long vol_rows = (long long)v.D * v.H * v.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = kLayThreads >> 6;
    fill_cell_rows(rows, n0, tn, cells, v, q);
    __syncthreads();
    const float* base = x_cl + (size_t)b * vol_rows * C2;
    if (lane < cc) {                                    // lanes along the channels: every lane its own cell of the point
        const int c2 = c0 + lane;
        const uint8_t* cb = cell + ((size_t)b * N + n0) * C + (c2 >= C ? c2 - C : c2);
        constexpr int U = 8;                            // eight points of the wave in flight: cell bytes first, then the values they select
        for (int i0 = wave; i0 < tn; i0 += U * n_waves) {
            int k[U];
            float val[U];
#pragma unroll
            for (int u = 0; u < U; ++u) k[u] = cb[(size_t)min(i0 + u * n_waves, tn - 1) * C];
#pragma unroll
            for (int u = 0; u < U; ++u) val[u] = base[(size_t)rows[min(i0 + u * n_waves, tn - 1) * cells + k[u]] * C2 + c2];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (i0 + u * n_waves < tn) tile[(i0 + u * n_waves) * ld + lane] = val[u];
        }
    }
    __syncthreads();
    for (int c = wave; c < cc; c += n_waves)
        if (lane < tn) out_cm[((size_t)b * C2 + c0 + c) * N + n0 + lane] = tile[lane * ld + c];
}

// out_cl[b, cell_row(n, k), c2] = (cell[b, n, c2 mod C] == k) ? src_cm[b, c2, n] : 0   for every cell k: the full tensor is
// written exactly once, zeros included (no memset, no index concatenation)
template <bool VEC>
__global__ __launch_bounds__(kLayThreads) void cell_scatter_kernel(const float* __restrict__ src_cm, const uint8_t* __restrict__ cell,
                                                                   float* __restrict__ out_cl, Vol v, Pool q, int C2, int C, int N) {
    __shared__ float tile[kTilePoints * (kTileChannels + 1)];
    __shared__ int rows[kTilePoints * kMaxCells];
    const int b = blockIdx.z, c0 = blockIdx.y * kTileChannels, n0 = blockIdx.x * kTilePoints;
    const int cc = min(kTileChannels, C2 - c0), tn = min(kTilePoints, N - n0);
    const int ld = kTileChannels + 1, cells = q.pd * q.ph * q.pw;
    const long long vol_rows = (long long)v.D * v.H * v.W;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, n_waves = kLayThreads >> 6;
    fill_cell_rows(rows, n0, tn, cells, v, q);
    if (lane < tn) {
        constexpr int U = 8;                            // eight channel rows of the wave in flight
        for (int cb = wave; cb < cc; cb += U * n_waves) {
            float val[U];
#pragma unroll
            for (int u = 0; u < U; ++u) val[u] = src_cm[((size_t)b * C2 + c0 + min(cb + u * n_waves, cc - 1)) * N + n0 + lane];
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (cb + u * n_waves < cc) tile[lane * ld + cb + u * n_waves] = val[u];
        }
    }
    __syncthreads();
    float* base = out_cl + (size_t)b * vol_rows * C2 + c0;
    if (VEC) {      // tile channel blocks never straddle C (C % 64 == 0 is not needed: c2 mod C is taken per 4-channel group)
        for (int e = threadIdx.x; e < tn * 16; e += kLayThreads) {
            const int i = e >> 4, c4 = (e & 15) << 2;
            if (c4 >= cc) continue;
            const int c2 = c0 + c4;
            const unsigned packed = *reinterpret_cast<const unsigned*>(cell + ((size_t)b * N + n0 + i) * C + (c2 >= C ? c2 - C : c2));
            const float* t = tile + i * ld + c4;
            const float4 val = make_float4(t[0], t[1], t[2], t[3]);
            const unsigned a0 = packed & 255u, a1 = (packed >> 8) & 255u, a2 = (packed >> 16) & 255u, a3 = packed >> 24;
            const int* r = rows + i * cells;
            for (unsigned k = 0; k < (unsigned)cells; ++k)
                *reinterpret_cast<float4*>(base + (size_t)r[k] * C2 + c4) =
                    make_float4(k == a0 ? val.x : 0.f, k == a1 ? val.y : 0.f, k == a2 ? val.z : 0.f, k == a3 ? val.w : 0.f);
        }
    } else {
        for (int i = wave; i < tn; i += n_waves) {
            if (lane >= cc) continue;
            const int c2 = c0 + lane;
            const int arg = cell[((size_t)b * N + n0 + i) * C + (c2 >= C ? c2 - C : c2)];
            const float val = tile[i * ld + lane];
            const int* r = rows + i * cells;
            for (int k = 0; k < cells; ++k) base[(size_t)r[k] * C2 + lane] = (k == arg) ? val : 0.f;
        }
    }
}

// out[b, d, hw, kd * C + c] = x[b, d + kd - 1, hw, c], zero outside the volume: the three depth taps of a [3,k,k] convolution as
// input channels, which turns its weight gradient into a 2-D problem (graph_ops._ConvDgradAsForward).  One workgroup per
// (plane, chunk of rows); a thread moves 16 bytes; 32-bit index arithmetic only.
constexpr int kUnrollRows = 32;
__global__ __launch_bounds__(kLayThreads) void depth_unroll_kernel(const float4* __restrict__ x, float4* __restrict__ out, int C4, int D, int HW) {
    const int plane = blockIdx.y;                    // b * D + d
    const int d = plane % D;
    const int row0 = blockIdx.x * kUnrollRows;
    const int rows = min(kUnrollRows, HW - row0);
    const int w3 = 3 * C4;
    const float4* xp = x + ((long)plane * HW + row0) * C4;
    float4* op = out + ((long)plane * HW + row0) * w3;
    const long plane_stride = (long)HW * C4;
    for (int e = threadIdx.x; e < rows * w3; e += kLayThreads) {
        const int r = e / w3, j = e - r * w3;
        const int kd = j / C4, c4 = j - kd * C4;
        const int ds = d + kd - 1;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (ds >= 0 && ds < D) v = xp[(long)(kd - 1) * plane_stride + (long)r * C4 + c4];
        op[e] = v;
    }
}

static int check_vol(const char* who, int B, int C, int D, int H, int W) {
    NEXTOU_REQUIRE(B > 0 && C > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, "%s: bad size B=%d C=%d D=%d H=%d W=%d", who, B, C, D, H, W);
    NEXTOU_REQUIRE((long long)D * H * W < (1ll << 31) && cdiv(C, kTileChannels) <= 65535, "%s: volume / channel count out of range", who);
    return 0;
}

static int launch_window(bool gather, const float* src, const float* residual, float* dst, int B, int C, int D, int H, int W,
                         int wd, int wh, int ww, int sd, int sh, int sw, hipStream_t s) {
    const char* who = gather ? "window_gather" : "window_scatter";
    NEXTOU_REQUIRE(src && dst, "%s: null pointer", who);
    if (int e = check_vol(who, B, C, D, H, W)) return e;
    NEXTOU_REQUIRE(wd > 0 && wh > 0 && ww > 0 && D % wd == 0 && H % wh == 0 && W % ww == 0,
                   "%s: window (%d,%d,%d) does not tile the volume (%d,%d,%d)", who, wd, wh, ww, D, H, W);
    NEXTOU_REQUIRE(sd >= 0 && sd < D && sh >= 0 && sh < H && sw >= 0 && sw < W, "%s: shift (%d,%d,%d) out of range", who, sd, sh, sw);
    const int Nw = wd * wh * ww;
    const size_t lds = (size_t)Nw * (kTileChannels + 1) * sizeof(float) + (size_t)Nw * sizeof(int);
    if (lds > 160 * 1024) return fail(NEXTOU_ENOTSUP, "%s: a window of %d points does not fit the LDS tile", who, Nw);
    const int nD = D / wd, nH = H / wh, nW = W / ww, n_win = nD * nH * nW;
    const Vol v{D, H, W};
    const Win w{wd, wh, ww, sd, sh, sw};
    const dim3 grid(n_win, cdiv(C, kTileChannels), B);
    const double bytes = (gather || residual == nullptr ? 2.0 : 3.0) * 4.0 * B * (double)C * D * H * W;
    const bool vec = (C % 4 == 0) && (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) |
                                        reinterpret_cast<uintptr_t>(residual)) & 15u) == 0);
    ProfScope prof(s, kBoundHbm, bytes, "%s[B%d C%d %dx%dx%d win %dx%dx%d]", gather ? "window_gather_kernel" : "window_scatter_kernel",
                   B, C, D, H, W, wd, wh, ww);
#define NEXTOU_WINDOW(G, V)                                                                                              \
    do {                                                                                                                 \
        if (lds > 64 * 1024)                                                                                             \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&window_move_kernel<G, V>),                          \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);                             \
        hipLaunchKernelGGL((window_move_kernel<G, V>), grid, dim3(kLayThreads), lds, s, src, residual, dst, v, w, C, nH, nW, \
                           n_win, Nw);                                                                                   \
    } while (0)
    if (gather && vec) NEXTOU_WINDOW(true, true);
    else if (gather) NEXTOU_WINDOW(true, false);
    else if (vec) NEXTOU_WINDOW(false, true);
    else NEXTOU_WINDOW(false, false);
#undef NEXTOU_WINDOW
    return check_launch(who);
}

static int make_pool(const char* who, int D, int H, int W, int pd, int ph, int pw, Pool* q) {
    NEXTOU_REQUIRE(pd > 0 && ph > 0 && pw > 0 && pd * ph * pw <= kMaxCells && D % pd == 0 && H % ph == 0 && W % pw == 0,
                   "%s: pool (%d,%d,%d) does not tile the volume (%d,%d,%d)", who, pd, ph, pw, D, H, W);
    *q = Pool{pd, ph, pw, D / pd, H / ph, W / pw};
    return 0;
}

}  // namespace nextou

using namespace nextou;

extern "C" int nextou_window_gather(const float* x_cl, float* out_cm, int B, int C, int D, int H, int W, int wd, int wh, int ww,
                                    int sd, int sh, int sw, nextou_stream_t stream) {
    return launch_window(true, x_cl, nullptr, out_cm, B, C, D, H, W, wd, wh, ww, sd, sh, sw, (hipStream_t)stream);
}

extern "C" int nextou_window_scatter(const float* src_cm, const float* residual_cl, float* out_cl, int B, int C, int D, int H, int W,
                                     int wd, int wh, int ww, int sd, int sh, int sw, nextou_stream_t stream) {
    return launch_window(false, src_cm, residual_cl, out_cl, B, C, D, H, W, wd, wh, ww, sd, sh, sw, (hipStream_t)stream);
}

extern "C" int nextou_pool_rows(const float* x_cl, float* values_cm, uint8_t* cell, int B, int C, int D, int H, int W, int pd, int ph,
                                int pw, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x_cl && values_cm && cell, "pool_rows: null pointer");
    if (int e = check_vol("pool_rows", B, C, D, H, W)) return e;
    Pool q;
    if (int e = make_pool("pool_rows", D, H, W, pd, ph, pw, &q)) return e;
    const int N = q.D2 * q.H2 * q.W2;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, 4.0 * B * (double)C * D * H * W + 5.0 * B * (double)C * N, "pool_rows_kernel[B%d C%d %dx%dx%d pool %dx%dx%d]",
                   B, C, D, H, W, pd, ph, pw);
    const dim3 grid(cdiv(N, kTilePoints), cdiv(C, kTileChannels), B);
    const bool vec = (C % 4 == 0) && (((reinterpret_cast<uintptr_t>(x_cl) | reinterpret_cast<uintptr_t>(cell)) & 15u) == 0);
    if (vec) hipLaunchKernelGGL(pool_rows_kernel<true>, grid, dim3(kLayThreads), 0, s, x_cl, values_cm, cell, Vol{D, H, W}, q, C, N);
    else hipLaunchKernelGGL(pool_rows_kernel<false>, grid, dim3(kLayThreads), 0, s, x_cl, values_cm, cell, Vol{D, H, W}, q, C, N);
    return check_launch("pool_rows_kernel");
}

extern "C" int nextou_cell_gather(const float* x_cl, const uint8_t* cell, float* out_cm, int B, int C2, int C, int D, int H, int W, int pd,
                                  int ph, int pw, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x_cl && cell && out_cm, "cell_gather: null pointer");
    NEXTOU_REQUIRE(C2 == C || C2 == 2 * C, "cell_gather: C2=%d must be C=%d or 2C", C2, C);
    if (int e = check_vol("cell_gather", B, C2, D, H, W)) return e;
    Pool q;
    if (int e = make_pool("cell_gather", D, H, W, pd, ph, pw, &q)) return e;
    const int N = q.D2 * q.H2 * q.W2;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, 4.0 * B * (double)C2 * D * H * W + 4.0 * B * (double)C2 * N + 1.0 * B * (double)C * N,
                   "cell_gather_kernel[B%d C%d %dx%dx%d pool %dx%dx%d]", B, C2, D, H, W, pd, ph, pw);
    hipLaunchKernelGGL(cell_gather_kernel, dim3(cdiv(N, kTilePoints), cdiv(C2, kTileChannels), B), dim3(kLayThreads), 0, s, x_cl, cell, out_cm,
                       Vol{D, H, W}, q, C2, C, N);
    return check_launch("cell_gather_kernel");
}

extern "C" int nextou_cell_scatter(const float* src_cm, const uint8_t* cell, float* out_cl, int B, int C2, int C, int D, int H, int W, int pd,
                                   int ph, int pw, nextou_stream_t stream) {
    NEXTOU_REQUIRE(src_cm && cell && out_cl, "cell_scatter: null pointer");
    NEXTOU_REQUIRE(C2 == C || C2 == 2 * C, "cell_scatter: C2=%d must be C=%d or 2C", C2, C);
    if (int e = check_vol("cell_scatter", B, C2, D, H, W)) return e;
    Pool q;
    if (int e = make_pool("cell_scatter", D, H, W, pd, ph, pw, &q)) return e;
    const int N = q.D2 * q.H2 * q.W2;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, 4.0 * B * (double)C2 * D * H * W + 4.0 * B * (double)C2 * N + 1.0 * B * (double)C * N,
                   "cell_scatter_kernel[B%d C%d %dx%dx%d pool %dx%dx%d]", B, C2, D, H, W, pd, ph, pw);
    const dim3 grid(cdiv(N, kTilePoints), cdiv(C2, kTileChannels), B);
    // a 4-channel group must not straddle the C boundary of the duplicated index set: C % 4 == 0 covers it
    const bool vec = (C % 4 == 0) && (C2 % 4 == 0) && (((reinterpret_cast<uintptr_t>(out_cl) | reinterpret_cast<uintptr_t>(cell)) & 15u) == 0);
    if (vec) hipLaunchKernelGGL(cell_scatter_kernel<true>, grid, dim3(kLayThreads), 0, s, src_cm, cell, out_cl, Vol{D, H, W}, q, C2, C, N);
    else hipLaunchKernelGGL(cell_scatter_kernel<false>, grid, dim3(kLayThreads), 0, s, src_cm, cell, out_cl, Vol{D, H, W}, q, C2, C, N);
    return check_launch("cell_scatter_kernel");
}

// out[p, :] = [a[p, :] + bias, b[p, :]] over channels-last rows: the decoder's torch.cat((up-convolution output, skip), 1)
// (reference NexToU_Encoder_Decoder.py:311-337) with the up-convolution's bias folded in, so that the convolution runs bias-free
// and ATen's separate bias-add pass over its output (288 us at stage 0) never runs.  One read of a and b, one write.
namespace nextou {
__global__ __launch_bounds__(256) void cat_bias_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ bias,
                                                            const float4* __restrict__ b, float4* __restrict__ out, long long P,
                                                            int c1q, int c2q, int rows_per_pass) {
    const int cq = c1q + c2q;
    const int r_local = threadIdx.x / cq, q = threadIdx.x - r_local * cq;
    if (r_local >= rows_per_pass) return;
    const bool first = q < c1q;
    const float4 bv = (first && bias) ? bias[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long step = (long long)gridDim.x * rows_per_pass;
    for (long long row = (long long)blockIdx.x * rows_per_pass + r_local; row < P; row += step) {
        float4 v;
        if (first) {
            v = a[row * c1q + q];
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        } else {
            v = b[row * c2q + (q - c1q)];
        }
        out[row * cq + q] = v;
    }
}

// cat_bias_rows_kernel whose first operand is the (P_in, T * C1) product of the up-convolution's input with its filter (K7, nextou_pw_rows):
// output row p takes the C1 channels of tap t of input point p_in (upconv_row) — the transposed convolution's "pixel shuffle" happens in the
// pass that concatenates anyway.
__global__ __launch_bounds__(256) void upconv_cat_rows_kernel(const float4* __restrict__ a, const float4* __restrict__ bias,
                                                              const float4* __restrict__ b, float4* __restrict__ out, long long P,
                                                              int c1q, int c2q, int rows_per_pass, UpShuffle u) {
    const int cq = c1q + c2q;
    const int r_local = threadIdx.x / cq, q = threadIdx.x - r_local * cq;
    if (r_local >= rows_per_pass) return;
    const bool first = q < c1q;
    const float4 bv = (first && bias) ? bias[q] : make_float4(0.f, 0.f, 0.f, 0.f);
    const long long step = (long long)gridDim.x * rows_per_pass;
    for (long long row = (long long)blockIdx.x * rows_per_pass + r_local; row < P; row += step) {
        float4 v;
        if (first) {
            if (a == nullptr) continue;            // the up-sampled half is already in place (nextou_pw_rows_up wrote it): only the skip half
            v = a[upconv_row(row, u) * c1q + q];
            v.x += bv.x; v.y += bv.y; v.z += bv.z; v.w += bv.w;
        } else {
            v = b[row * c2q + (q - c1q)];
        }
        out[row * cq + q] = v;
    }
}

// The skip half alone (nextou_upconv_cat_rows with y2 == NULL: the up-sampled half was stored in place by nextou_pw_rows_up): every thread
// moves a 16-byte piece of the skip rows into channels C1 .. C1 + C2 of the wide rows — no idle lanes (run through upconv_cat_rows_kernel
// with half of every wave skipping its rows this copy took as long as the whole two-operand pass).
__global__ __launch_bounds__(256) void cat_skip_half_kernel(const float4* __restrict__ b, float4* __restrict__ out, long long P, int c1q, int c2q,
                                                            int rows_per_pass) {
    const int r_local = threadIdx.x / c2q, q = threadIdx.x - r_local * c2q;
    if (r_local >= rows_per_pass) return;
    const int cq = c1q + c2q;
    const long long step = (long long)gridDim.x * rows_per_pass;
    long long row = (long long)blockIdx.x * rows_per_pass + r_local;
    for (; row + step < P; row += 2 * step) {
        const float4 v0 = b[row * c2q + q], v1 = b[(row + step) * c2q + q];
        out[row * cq + c1q + q] = v0;
        out[(row + step) * cq + c1q + q] = v1;
    }
    if (row < P) out[row * cq + c1q + q] = b[row * c2q + q];
}

// out[ci][kd][kh][kw][co] (channels-last memory of the (Ci, Co, Kd, Kh, Kw) filter) = w[co][ci][Kd-1-kd][Kh-1-kh][Kw-1-kw]: the filter of the
// forward convolution that computes a stride-1 convolution's data gradient, from the forward filter in either memory layout (element
// strides).  One 32 x 32 (co, ci) tile per tap through LDS: reads run along ci (contiguous in a channels-last filter), writes along co.
__global__ __launch_bounds__(256) void filter_flip_t_kernel(const float* __restrict__ w, float* __restrict__ out, int Co, int Ci, int Kh,
                                                            int Kw, int Kd, long long s_co, long long s_ci, long long s_kd, long long s_kh,
                                                            long long s_kw) {
    __shared__ float tile[32][33];
    const int T = Kd * Kh * Kw, t = blockIdx.z;
    const int kd = t / (Kh * Kw), kh = (t / Kw) % Kh, kw = t % Kw;
    const long long off = (long long)(Kd - 1 - kd) * s_kd + (long long)(Kh - 1 - kh) * s_kh + (long long)(Kw - 1 - kw) * s_kw;
    const int ci0 = blockIdx.x * 32, co0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int co = co0 + ty + 8 * j, ci = ci0 + tx;
        if (co < Co && ci < Ci) tile[ty + 8 * j][tx] = w[(long long)co * s_co + (long long)ci * s_ci + off];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int ci = ci0 + ty + 8 * j, co = co0 + tx;
        if (co < Co && ci < Ci) out[((size_t)ci * T + t) * Co + co] = tile[tx][ty + 8 * j];
    }
}
}  // namespace nextou

extern "C" int nextou_filter_flip_t(const float* w, float* out, int Co, int Ci, int Kd, int Kh, int Kw, int64_t s_co, int64_t s_ci,
                                    int64_t s_kd, int64_t s_kh, int64_t s_kw, nextou_stream_t stream) {
    NEXTOU_REQUIRE(w && out, "filter_flip_t: null pointer");
    NEXTOU_REQUIRE(Co > 0 && Ci > 0 && Kd > 0 && Kh > 0 && Kw > 0 && (long long)Kd * Kh * Kw <= 65535 && Co <= (1 << 20) && Ci <= (1 << 20),
                   "filter_flip_t: bad size Co=%d Ci=%d kernel (%d,%d,%d)", Co, Ci, Kd, Kh, Kw);
    NEXTOU_REQUIRE(s_co >= 0 && s_ci >= 0 && s_kd >= 0 && s_kh >= 0 && s_kw >= 0, "filter_flip_t: negative stride");
    hipStream_t s = (hipStream_t)stream;
    const int T = Kd * Kh * Kw;
    ProfScope prof(s, kBoundHbm, 8.0 * (double)Co * Ci * T, "filter_flip_t_kernel[%dx%d k%dx%dx%d]", Co, Ci, Kd, Kh, Kw);
    hipLaunchKernelGGL(nextou::filter_flip_t_kernel, dim3((unsigned)((Ci + 31) / 32), (unsigned)((Co + 31) / 32), (unsigned)T), dim3(256), 0, s,
                       w, out, Co, Ci, Kh, Kw, Kd, (long long)s_co, (long long)s_ci, (long long)s_kd, (long long)s_kh, (long long)s_kw);
    return check_launch("filter_flip_t_kernel");
}

extern "C" int nextou_cat_bias_rows(const float* a, const float* bias, const float* b, float* out, int64_t P, int C1, int C2,
                                    nextou_stream_t stream) {
    NEXTOU_REQUIRE(a && b && out, "cat_bias_rows: null pointer");
    NEXTOU_REQUIRE(P > 0 && C1 > 0 && C2 > 0 && C1 % 4 == 0 && C2 % 4 == 0 && (C1 + C2) / 4 <= 256,
                   "cat_bias_rows: bad size P=%lld C1=%d C2=%d (multiples of 4, C1 + C2 <= 1024)", (long long)P, C1, C2);
    NEXTOU_REQUIRE(((reinterpret_cast<uintptr_t>(a) | reinterpret_cast<uintptr_t>(b) | reinterpret_cast<uintptr_t>(out) |
                     reinterpret_cast<uintptr_t>(bias)) & 15u) == 0, "cat_bias_rows: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const int cq = (C1 + C2) / 4, rpp = 256 / cq;
    long long blocks = (P + rpp - 1) / rpp;
    if (blocks > 16384) blocks = 16384;
    ProfScope prof(s, kBoundHbm, 8.0 * (double)P * (C1 + C2), "cat_bias_rows_kernel[P%lld C%d+%d]", (long long)P, C1, C2);
    hipLaunchKernelGGL(nextou::cat_bias_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(a),
                       reinterpret_cast<const float4*>(bias), reinterpret_cast<const float4*>(b), reinterpret_cast<float4*>(out),
                       (long long)P, C1 / 4, C2 / 4, rpp);
    return check_launch("cat_bias_rows_kernel");
}

extern "C" int nextou_upconv_cat_rows(const float* y2, const float* bias, const float* skip, float* out, int B, int D, int H, int W, int sd,
                                      int sh, int sw, int C1, int C2, nextou_stream_t stream) {
    NEXTOU_REQUIRE(skip && out, "upconv_cat_rows: null pointer");          // y2 == NULL: fill the skip half only (the first C1 channels stay as they are)
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && sd >= 1 && sh >= 1 && sw >= 1 && sd <= 4 && sh <= 4 && sw <= 4 && C1 > 0 && C2 > 0 &&
                   C1 % 4 == 0 && C2 % 4 == 0 && (C1 + C2) / 4 <= 256,
                   "upconv_cat_rows: bad size B=%d (%d,%d,%d) stride (%d,%d,%d) C %d+%d (multiples of 4, C1 + C2 <= 1024)", B, D, H, W, sd, sh,
                   sw, C1, C2);
    NEXTOU_REQUIRE(((reinterpret_cast<uintptr_t>(y2) | reinterpret_cast<uintptr_t>(skip) | reinterpret_cast<uintptr_t>(out) |
                     reinterpret_cast<uintptr_t>(bias)) & 15u) == 0, "upconv_cat_rows: tensors must be 16-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    const nextou::UpShuffle u{D * sd, H * sh, W * sw, sd, sh, sw};
    const long long P = (long long)B * u.D2 * u.H2 * u.W2;
    if (y2 == nullptr && C2 / 4 <= 256) {
        const int rpp2 = 256 / (C2 / 4);
        long long blocks2 = (P + rpp2 - 1) / rpp2;
        if (blocks2 > 16384) blocks2 = 16384;
        ProfScope prof(s, kBoundHbm, 8.0 * (double)P * C2, "cat_skip_half_kernel[P%lld C%d+%d]", P, C1, C2);
        hipLaunchKernelGGL(nextou::cat_skip_half_kernel, dim3((unsigned)blocks2), dim3(256), 0, s, reinterpret_cast<const float4*>(skip),
                           reinterpret_cast<float4*>(out), P, C1 / 4, C2 / 4, rpp2);
        return check_launch("cat_skip_half_kernel");
    }
    const int cq = (C1 + C2) / 4, rpp = 256 / cq;
    long long blocks = (P + rpp - 1) / rpp;
    if (blocks > 16384) blocks = 16384;
    ProfScope prof(s, kBoundHbm, 8.0 * (double)P * (y2 ? C1 + C2 : C2), "upconv_cat_rows_kernel%s[P%lld C%d+%d s%dx%dx%d]", y2 ? "" : "<skip half>", P, C1,
                   C2, sd, sh, sw);
    hipLaunchKernelGGL(nextou::upconv_cat_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, s, reinterpret_cast<const float4*>(y2),
                       reinterpret_cast<const float4*>(bias), reinterpret_cast<const float4*>(skip), reinterpret_cast<float4*>(out), P,
                       C1 / 4, C2 / 4, rpp, u);
    return check_launch("upconv_cat_rows_kernel");
}

extern "C" int nextou_depth_unroll(const float* x_cl, float* out_cl, int B, int C, int D, int H, int W, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x_cl && out_cl, "depth_unroll: null pointer");
    if (int e = check_vol("depth_unroll", B, C, D, H, W)) return e;
    NEXTOU_REQUIRE(C % 4 == 0 && (((reinterpret_cast<uintptr_t>(x_cl) | reinterpret_cast<uintptr_t>(out_cl)) & 15u) == 0),
                   "depth_unroll: C=%d must be a multiple of 4 and the tensors 16-byte aligned", C);
    NEXTOU_REQUIRE((long long)B * D <= 65535, "depth_unroll: B*D=%lld planes exceed the grid", (long long)B * D);
    hipStream_t s = (hipStream_t)stream;
    const double bytes = 4.0 * B * (double)C * D * H * W;
    ProfScope prof(s, kBoundHbm, 4.0 * bytes, "depth_unroll_kernel[B%d C%d %dx%dx%d]", B, C, D, H, W);     // one read (neighbour planes from L2) + three writes
    hipLaunchKernelGGL(depth_unroll_kernel, dim3(cdiv(H * W, kUnrollRows), B * D), dim3(kLayThreads), 0, s,
                       reinterpret_cast<const float4*>(x_cl), reinterpret_cast<float4*>(out_cl), C / 4, D, H * W);
    return check_launch("depth_unroll_kernel");
}
