// K8 — the segmentation heads: 1x1 convolutions WITH bias from C feature channels to L class logits on channels-last rows,
// forward, data gradient, weight + bias gradient (gfx950).
//
// Reference: the deep-supervision `seg_layers` of the decoder (NexToU_Encoder_Decoder.py:253-258 build, :311-337 forward:
// `self.seg_layers[s](x)` — nn.Conv3d(features, num_classes, 1, 1, 0, bias=True)).  PyTorch-ROCm hands them to MIOpen as
// convolutions; the backward of the full-resolution head is the call the averaged (N > 1) eager step died in with a GPU
// memory fault (profiles/r04_sgd_fused.md, tools/conv_bwd_fault_repro.py), so the heads moved onto kernels of this library.
//
// The problem is HBM-bound by a wide margin (cfg 2, full resolution: 5.5 M points x (40 -> 14): 216 B and 1 120 flop per
// point), L <= 16 per tile is far below any GEMM tile, and the rows (160 B and 56 B) are shorter than a wave's access, so
// none of K7's kernels fit.  All three kernels use v_mfma_f32_16x16x4_f32 (exact f32, a k-ordered fma chain) only to keep
// the VALU out of the way; what matters is that every global access of a wave covers whole, adjacent rows:
//
//   forward   Y^T (L x 16 points) = W (L x C) . X^T: a lane (q = lane & 15, g = lane >> 4) loads the float4
//             x[p0 + q][16 j + 4 g ..] (a wave: 16 rows x 64 contiguous bytes), the same float4 of W[q] comes from LDS,
//             4 MFMAs per 16 channels; the lane ends up with 4 consecutive logits of ONE point (8-byte stores).
//   dgrad     GX^T (C x 16 points) = W^T . GY^T: the lane loads gy[p0 + q][4 g ..] (a wave: 16 whole rows = 896
//             contiguous bytes for L = 14), W^T from LDS, and stores one float4 of gx per 16 channels.
//   wgrad     GW (L x C) = GY^T . X over the points (the MFMA's k index): 4 + 4 scalar row loads per 16 points and
//             channel tile, a `ones` column after the last channel gives the bias gradient from the same MFMAs; the waves
//             of a workgroup are summed through LDS in wave order, workgroups through `workspace` by a fixed-order reduce
//             kernel: bit-reproducible.
#include "common.h"
#include <cstdlib>

namespace nextou {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kHeadWaves = 4;
constexpr int kHeadThreads = 64 * kHeadWaves;
constexpr int kHeadPT = 2;        // 16-point tiles a wave has in flight (forward, dgrad)
constexpr int kHeadJB = 3;        // 16-channel chunks loaded before the first MFMA of a batch (forward)
constexpr int kHeadCG = 4;        // channel tiles per workgroup row of the weight-gradient kernel (grid.y)
constexpr int kHeadMaxLT = 4;     // class tiles of 16 the data-gradient kernel keeps in registers: L <= 64

__device__ __forceinline__ f32x4 mfma4(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// y[p, l] = bias[l] + sum_c x[p, c] w[l, c], classes lt*16 .. lt*16 + 15 (blockIdx.y = lt)
template <bool VEC>
__global__ __launch_bounds__(kHeadThreads) void head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ bias, float* __restrict__ y, long P, int L,
                                                                int C, long ldx, long ldy, int KT, long tiles) {
    extern __shared__ float wl[];                       // [16][ldw]: this class tile's weights, zero padded
    const int lt = blockIdx.y;
    const int ldw = KT * 16 + 4;
    for (int e = threadIdx.x; e < 16 * ldw; e += kHeadThreads) {
        const int l = lt * 16 + e / ldw, c = e % ldw;
        wl[e] = (l < L && c < C) ? w[(long)l * C + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    f32x4 binit;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int l = lt * 16 + 4 * g + r;
        binit[r] = (bias != nullptr && l < L) ? bias[l] : 0.f;
    }
    const long n_waves = (long)gridDim.x * kHeadWaves;
    for (long t0 = ((long)blockIdx.x * kHeadWaves + wave) * kHeadPT; t0 < tiles; t0 += n_waves * kHeadPT) {
        f32x4 acc[kHeadPT];
#pragma unroll
        for (int pt = 0; pt < kHeadPT; ++pt) acc[pt] = binit;
        for (int jb = 0; jb < KT; jb += kHeadJB) {
            float4 xv[kHeadPT][kHeadJB];
#pragma unroll
            for (int pt = 0; pt < kHeadPT; ++pt) {
                const long p = (t0 + pt) * 16 + q;
#pragma unroll
                for (int jj = 0; jj < kHeadJB; ++jj) {
                    const int c = (jb + jj) * 16 + 4 * g;
                    xv[pt][jj] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (p < P && c < C) {
                        const float* src = x + p * ldx + c;
                        if (VEC) {
                            xv[pt][jj] = *reinterpret_cast<const float4*>(src);
                        } else {
                            xv[pt][jj].x = src[0];
                            if (c + 1 < C) xv[pt][jj].y = src[1];
                            if (c + 2 < C) xv[pt][jj].z = src[2];
                            if (c + 3 < C) xv[pt][jj].w = src[3];
                        }
                    }
                }
            }
#pragma unroll
            for (int jj = 0; jj < kHeadJB; ++jj) {
                if (jb + jj < KT) {
                    const float4 wv = *reinterpret_cast<const float4*>(&wl[q * ldw + (jb + jj) * 16 + 4 * g]);
#pragma unroll
                    for (int pt = 0; pt < kHeadPT; ++pt) {
                        acc[pt] = mfma4(wv.x, xv[pt][jj].x, acc[pt]);
                        acc[pt] = mfma4(wv.y, xv[pt][jj].y, acc[pt]);
                        acc[pt] = mfma4(wv.z, xv[pt][jj].z, acc[pt]);
                        acc[pt] = mfma4(wv.w, xv[pt][jj].w, acc[pt]);
                    }
                }
            }
        }
#pragma unroll
        for (int pt = 0; pt < kHeadPT; ++pt) {
            const long p = (t0 + pt) * 16 + q;
            const int l0 = lt * 16 + 4 * g;
            if (p < P && l0 < L) {
                float* dst = y + p * ldy + l0;
                if (((L | ldy) & 1) == 0) {             // even row length: 8-byte stores
                    *reinterpret_cast<float2*>(dst) = make_float2(acc[pt][0], acc[pt][1]);
                    if (l0 + 2 < L) *reinterpret_cast<float2*>(dst + 2) = make_float2(acc[pt][2], acc[pt][3]);
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (l0 + r < L) dst[r] = acc[pt][r];
                }
            }
        }
    }
}

// gx[p, c] = sum_l gy[p, l] w[l, c]
template <bool VEC>
__global__ __launch_bounds__(kHeadThreads) void head_dgrad_kernel(const float* __restrict__ gy, const float* __restrict__ w,
                                                                  float* __restrict__ gx, long P, int L, int C, long ldg, long ldx, int CT,
                                                                  int LT, long tiles) {
    extern __shared__ float wl[];                       // [LT * 16][ldw], zero padded
    const int ldw = CT * 16 + 4;
    for (int e = threadIdx.x; e < LT * 16 * ldw; e += kHeadThreads) {
        const int l = e / ldw, c = e % ldw;
        wl[e] = (l < L && c < C) ? w[(long)l * C + c] : 0.f;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const bool pair = ((L | ldg) & 1) == 0;
    const long n_waves = (long)gridDim.x * kHeadWaves;
    for (long t0 = ((long)blockIdx.x * kHeadWaves + wave) * kHeadPT; t0 < tiles; t0 += n_waves * kHeadPT) {
        float gv[kHeadPT][kHeadMaxLT][4];
#pragma unroll
        for (int pt = 0; pt < kHeadPT; ++pt) {
            const long p = (t0 + pt) * 16 + q;
#pragma unroll
            for (int lt = 0; lt < kHeadMaxLT; ++lt) {
#pragma unroll
                for (int r = 0; r < 4; ++r) gv[pt][lt][r] = 0.f;
                const int l0 = lt * 16 + 4 * g;
                if (lt < LT && p < P && l0 < L) {
                    const float* src = gy + p * ldg + l0;
                    if (pair) {
                        const float2 a = *reinterpret_cast<const float2*>(src);
                        gv[pt][lt][0] = a.x;
                        gv[pt][lt][1] = a.y;
                        if (l0 + 2 < L) {
                            const float2 b = *reinterpret_cast<const float2*>(src + 2);
                            gv[pt][lt][2] = b.x;
                            gv[pt][lt][3] = b.y;
                        }
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (l0 + r < L) gv[pt][lt][r] = src[r];
                    }
                }
            }
        }
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 acc[kHeadPT];
#pragma unroll
            for (int pt = 0; pt < kHeadPT; ++pt) acc[pt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int lt = 0; lt < kHeadMaxLT; ++lt) {
                if (lt < LT) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const float wv = wl[(lt * 16 + 4 * g + r) * ldw + ct * 16 + q];     // A[i = channel q][k = class 4g + r]
#pragma unroll
                        for (int pt = 0; pt < kHeadPT; ++pt) acc[pt] = mfma4(wv, gv[pt][lt][r], acc[pt]);
                    }
                }
            }
            const int c0 = ct * 16 + 4 * g;
#pragma unroll
            for (int pt = 0; pt < kHeadPT; ++pt) {
                const long p = (t0 + pt) * 16 + q;
                if (p < P && c0 < C) {
                    float* dst = gx + p * ldx + c0;
                    if (VEC) {
                        *reinterpret_cast<float4*>(dst) = make_float4(acc[pt][0], acc[pt][1], acc[pt][2], acc[pt][3]);
                    } else {
#pragma unroll
                        for (int r = 0; r < 4; ++r)
                            if (c0 + r < C) dst[r] = acc[pt][r];
                    }
                }
            }
        }
    }
}

// part[blockIdx.x][lt][l][c] = sum over this workgroup's points of gy[p, lt*16 + l] * xe[p, c], xe = [x, 1]: column C is the bias gradient.
// grid (blocks, channel-tile groups, class tiles); CW = 16 * (channel tiles incl. the ones column)
__global__ __launch_bounds__(kHeadThreads) void head_wgrad_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  float* __restrict__ part, long P, int L, int C, long ldg, long ldx,
                                                                  int CTtot, int LT, long tiles) {
    __shared__ float red[kHeadWaves][kHeadCG][4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int ct0 = blockIdx.y * kHeadCG, lt = blockIdx.z;
    const int l = lt * 16 + q;
    f32x4 acc[kHeadCG];
#pragma unroll
    for (int j = 0; j < kHeadCG; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long n_waves = (long)gridDim.x * kHeadWaves;
    for (long t = (long)blockIdx.x * kHeadWaves + wave; t < tiles; t += n_waves) {
        float a[4], b[kHeadCG][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long p = t * 16 + 4 * g + r;               // MFMA r sums over the points {4g + r}
            a[r] = (p < P && l < L) ? gy[p * ldg + l] : 0.f;
#pragma unroll
            for (int j = 0; j < kHeadCG; ++j) {
                const int c = (ct0 + j) * 16 + q;
                b[j][r] = (p < P) ? (c < C ? x[p * ldx + c] : (c == C ? 1.f : 0.f)) : 0.f;
            }
        }
#pragma unroll
        for (int j = 0; j < kHeadCG; ++j) {
            if (ct0 + j < CTtot) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[j] = mfma4(a[r], b[j][r], acc[j]);
            }
        }
    }
    // waves in wave order (fixed), then one float per (class 4g' + r, channel q) of every channel tile
#pragma unroll
    for (int j = 0; j < kHeadCG; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) red[wave][j][r][lane] = acc[j][r];
    __syncthreads();
    const int CW = CTtot * 16;
    for (int e = threadIdx.x; e < kHeadCG * 4 * 64; e += kHeadThreads) {
        const int ln = e & 63, r = (e >> 6) & 3, j = e >> 8;
        if (ct0 + j >= CTtot) continue;
        const float s = ((red[0][j][r][ln] + red[1][j][r][ln]) + red[2][j][r][ln]) + red[3][j][r][ln];
        const int row = 4 * (ln >> 4) + r, col = (ct0 + j) * 16 + (ln & 15);
        part[(((long)blockIdx.x * LT + lt) * 16 + row) * CW + col] = s;
    }
}

// ---------------------------------------------------------------------------------------------
// LDS-staged backward for dense rows (ldg == L, ldx == C, C % 4 == 0, L <= 16, C <= 104): the direct kernels above read gy twice and
// touch every 128-byte line of gx in 64-byte pieces from different instructions (data gradient 0.41, weight gradient 0.54 of 8 TB/s
// at cfg 2's full-resolution head, profiles/r05_head_bench_v1.txt).  Here a wave copies a tile of 32 WHOLE rows of gy and x — one
// contiguous range of global memory each — into its private LDS region with 16-byte accesses, BOTH gradients' MFMA operands come
// from LDS in whatever order they need, and gx leaves the same way: 637 -> 385 us at that head (0.67 of 8 TB/s on 376 B per point).
// Wave-private regions: LDS instructions of one wave execute in order, so no barrier is needed inside the tile loop, and the gx
// tile may overwrite the x tile it was computed beside.  (The same staging for the forward was measured and dropped: 243 us
// against the direct kernel's 228 — the forward has one operand and its 64-byte pieces already merge in L2.)
// ---------------------------------------------------------------------------------------------
constexpr int kHeadTileRows = 32;      // rows per wave and iteration (two MFMA point tiles)
constexpr int kHeadMaxCT = 7;          // channel tiles incl. the bias column: C + 1 <= 112

// A wave's view of one tile of whole rows: NV 16-byte pieces per lane, fetched into REGISTERS one iteration ahead (the loads of
// tile t + 1 are in flight while tile t multiplies — written as "load the tile, then use it" every tile cost the wave one exposed HBM
// round trip per piece: 662 us for the first head_bwd_lds_kernel where the two direct kernels it replaces took 637) and committed to
// the wave's private LDS region at the top of the next iteration.  `base`: start of the tensor (a valid 16-byte read for lanes that
// have no piece), n_floats: floats of this tile (a multiple of 4 except possibly in the tensor's last tile: scalar tail).
__device__ __forceinline__ int head_rows_of(long P, long tt) {
    const long left = P - tt * kHeadTileRows;
    return (int)(left < kHeadTileRows ? left : kHeadTileRows);
}
template <int NV>
__device__ __forceinline__ void tile_fetch(f32x4 (&v)[NV], const float* __restrict__ base, const float* __restrict__ src, int n_floats, int lane) {
    const int n4 = n_floats >> 2;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = u * 64 + lane;
        const long off = i < n4 ? (src - base) + 4 * (long)i : 0;        // (an offset, not a select of two pointers)
        v[u] = *reinterpret_cast<const f32x4*>(base + off);
    }
}
template <int NV>
__device__ __forceinline__ void tile_commit(const f32x4 (&v)[NV], float* dst, const float* __restrict__ src, int n_floats, int lane) {
    const int n4 = n_floats >> 2;
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = u * 64 + lane;
        if (i < n4) reinterpret_cast<f32x4*>(dst)[i] = v[u];
    }
    const int rest = n_floats & 3;
    if (lane < rest) dst[4 * n4 + lane] = src[4 * n4 + lane];       // (the tensor's last, ragged tile only)
}
__device__ __forceinline__ void tile_out(float* __restrict__ dst, const float* src, int n_floats, int lane) {
    const int n4 = n_floats >> 2;
    for (int i = lane; i < n4; i += 64) reinterpret_cast<float4*>(dst)[i] = reinterpret_cast<const float4*>(src)[i];
    const int rest = n_floats & 3;
    if (lane < rest) dst[4 * n4 + lane] = src[4 * n4 + lane];
}

// gx = gy W (rows), part[block] = [gy^T x | gy^T 1] of this workgroup's rows: the data gradient and the weight + bias gradient
// from ONE read of gy and x
template <int NXV>
__global__ __launch_bounds__(kHeadThreads) void head_bwd_lds_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                    const float* __restrict__ w, float* __restrict__ gx,
                                                                    float* __restrict__ part, long P, int L, int C, int CT, int CTtot,
                                                                    long tiles) {
    extern __shared__ __attribute__((aligned(16))) float hsm[];
    const int ldw = CT * 16 + 4;
    float* wl = hsm;                                                    // [16][ldw]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float* gt = hsm + 16 * ldw + (size_t)wave * kHeadTileRows * (C + 16);   // [32][L] (<= 32 x 16)
    float* xt = gt + kHeadTileRows * 16;                                    // [32][C]: x, then gx
    for (int e = threadIdx.x; e < 16 * ldw; e += kHeadThreads) {
        const int l = e / ldw, c = e % ldw;
        wl[e] = (l < L && c < C) ? w[(long)l * C + c] : 0.f;
    }
    __syncthreads();
    const int q = lane & 15, g = lane >> 4;
    f32x4 wacc[kHeadMaxCT];
#pragma unroll
    for (int j = 0; j < kHeadMaxCT; ++j) wacc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    const long n_waves = (long)gridDim.x * kHeadWaves;
    long t = (long)blockIdx.x * kHeadWaves + wave;
    f32x4 xr[NXV], gr[2];
    if (t < tiles) {
        tile_fetch(gr, gy, gy + t * kHeadTileRows * L, head_rows_of(P, t) * L, lane);
        tile_fetch(xr, x, x + t * kHeadTileRows * C, head_rows_of(P, t) * C, lane);
    }
    for (; t < tiles; t += n_waves) {
        const long p0 = t * kHeadTileRows;
        const int rows = head_rows_of(P, t);
        tile_commit(gr, gt, gy + p0 * L, rows * L, lane);
        tile_commit(xr, xt, x + p0 * C, rows * C, lane);
        const long nx = t + n_waves < tiles ? t + n_waves : t;
        tile_fetch(gr, gy, gy + nx * kHeadTileRows * L, head_rows_of(P, nx) * L, lane);
        tile_fetch(xr, x, x + nx * kHeadTileRows * C, head_rows_of(P, nx) * C, lane);
        __builtin_amdgcn_wave_barrier();
        // weight + bias gradient: A[i = class q][k = point], B[k = point][j = channel q]; MFMA r of a point tile sums the points {4g + r}
        float a[2][4];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = pt * 16 + 4 * g + r;
                a[pt][r] = (row < rows && q < L) ? gt[row * L + q] : 0.f;
            }
#pragma unroll
        for (int j = 0; j < kHeadMaxCT; ++j) {
            if (j < CTtot) {
                const int c = j * 16 + q;
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = pt * 16 + 4 * g + r;
                        const float b = row < rows ? (c < C ? xt[row * C + c] : (c == C ? 1.f : 0.f)) : 0.f;
                        wacc[j] = mfma4(a[pt][r], b, wacc[j]);
                    }
            }
        }
        // data gradient: A[i = channel q][k = class 4g + r] = W[4g + r][ct * 16 + q], B[k][j = point q] = gy[q][4g + r]
        float gv[2][4];
#pragma unroll
        for (int pt = 0; pt < 2; ++pt)
#pragma unroll
            for (int r = 0; r < 4; ++r) gv[pt][r] = (pt * 16 + q < rows && 4 * g + r < L) ? gt[(pt * 16 + q) * L + 4 * g + r] : 0.f;
        __builtin_amdgcn_wave_barrier();
        for (int ct = 0; ct < CT; ++ct) {
            f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float wv = wl[(4 * g + r) * ldw + ct * 16 + q];
#pragma unroll
                for (int pt = 0; pt < 2; ++pt) acc[pt] = mfma4(wv, gv[pt][r], acc[pt]);
            }
            const int c0 = ct * 16 + 4 * g;
            if (c0 < C) {
#pragma unroll
                for (int pt = 0; pt < 2; ++pt)          // over the x tile: every read of it by this wave has been issued (in-order LDS)
                    *reinterpret_cast<float4*>(&xt[(pt * 16 + q) * C + c0]) = make_float4(acc[pt][0], acc[pt][1], acc[pt][2], acc[pt][3]);
            }
        }
        __builtin_amdgcn_wave_barrier();
        tile_out(gx + p0 * C, xt, rows * C, lane);
        __builtin_amdgcn_wave_barrier();
    }
    // waves in wave order, then one partial per workgroup (the layout head_wgrad_reduce_kernel sums)
    __syncthreads();
    float* red = hsm;                                                   // [waves][CTtot][4][64] over everything (all tiles are dead)
#pragma unroll
    for (int j = 0; j < kHeadMaxCT; ++j)
        if (j < CTtot) {
#pragma unroll
            for (int r = 0; r < 4; ++r) red[((wave * CTtot + j) * 4 + r) * 64 + lane] = wacc[j][r];
        }
    __syncthreads();
    const int CW = CTtot * 16;
    for (int e = threadIdx.x; e < CTtot * 4 * 64; e += kHeadThreads) {
        const int ln = e & 63, r = (e >> 6) & 3, j = e >> 8;
        float s = red[((0 * CTtot + j) * 4 + r) * 64 + ln];
#pragma unroll
        for (int wv = 1; wv < kHeadWaves; ++wv) s += red[((wv * CTtot + j) * 4 + r) * 64 + ln];
        const int row = 4 * (ln >> 4) + r, col = j * 16 + (ln & 15);
        part[((long)blockIdx.x * 16 + row) * CW + col] = s;
    }
}

// gw[l, c] = sum_blocks part[b][l][c] (c < C), gb[l] = sum_blocks part[b][l][C]: 16 slices of blocks (mod 16), each in ascending
// order with four loads in flight, combined by a fixed tree — bit-reproducible
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb,
                                                                int blocks, int L, int C, int LT, int CW) {
    __shared__ float partial[16][16];
    const int slice = threadIdx.x >> 4, ln = threadIdx.x & 15;
    const long e = (long)blockIdx.x * 16 + ln;              // over L x (C + 1)
    const long elems = (long)L * (C + 1);
    float s = 0.f;
    int l = 0, c = 0;
    if (e < elems) {
        l = (int)(e / (C + 1));
        c = (int)(e % (C + 1));
        const long stride = (long)LT * 16 * CW;
        const float* src = part + (long)l * CW + c;
        int i = slice;
        for (; i + 48 < blocks; i += 64) {
            const float v0 = src[(long)i * stride], v1 = src[(long)(i + 16) * stride], v2 = src[(long)(i + 32) * stride],
                        v3 = src[(long)(i + 48) * stride];
            s = (((s + v0) + v1) + v2) + v3;
        }
        for (; i < blocks; i += 16) s += src[(long)i * stride];
    }
    partial[slice][ln] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        float t[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) t[k] = partial[k][ln];
#pragma unroll
        for (int h = 8; h >= 1; h >>= 1)
#pragma unroll
            for (int k = 0; k < h; ++k) t[k] = t[k] + t[k + h];
        if (c < C) {
            if (gw != nullptr) gw[(long)l * C + c] = t[0];
        } else if (gb != nullptr) {
            gb[l] = t[0];
        }
    }
}

int head_cus() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

int check_head(const char* what, int64_t P, int L, int C, int64_t ldrow_l, int64_t ldrow_c) {
    NEXTOU_REQUIRE(P > 0 && L > 0 && C > 0, "%s: P=%lld L=%d C=%d must be positive", what, (long long)P, L, C);
    NEXTOU_REQUIRE(ldrow_l >= L && ldrow_c >= C, "%s: row strides (%lld, %lld) shorter than the rows (%d, %d)", what, (long long)ldrow_l,
                   (long long)ldrow_c, L, C);
    NEXTOU_REQUIRE(P < (int64_t)1 << 40, "%s: P=%lld too large", what, (long long)P);
    return 0;
}

int wgrad_blocks(int64_t tiles) {
    // every wave should see >= 8 tiles of 16 points; at most 2 workgroups per CU
    const int64_t want = (tiles + 8 * kHeadWaves - 1) / (8 * kHeadWaves);
    const int64_t cap = 2 * (int64_t)head_cus();
    return (int)(want < 1 ? 1 : (want > cap ? cap : want));
}

// LDS-staged kernels: bytes of dynamic LDS (0 = shape not eligible) and the grid
struct LdsPlan { size_t lds; int blocks; };
LdsPlan plan_bwd_lds(int64_t P, int L, int C) {
    const int CTtot = cdiv(C + 1, 16);
    if (C % 4 != 0 || L > 16 || C > 104 || CTtot > kHeadMaxCT || P < 4) return {0, 0};
    size_t lds = ((size_t)16 * (cdiv(C, 16) * 16 + 4) + (size_t)kHeadWaves * kHeadTileRows * (C + 16)) * sizeof(float);
    const size_t red = (size_t)kHeadWaves * CTtot * 4 * 64 * sizeof(float);
    if (red > lds) lds = red;
    if (lds > 64 * 1024) return {0, 0};
    const int64_t tiles = cdiv64(P, kHeadTileRows);
    int per_cu = (int)((160 * 1024) / lds);
    if (per_cu > 4) per_cu = 4;
    // every wave should see >= 4 tiles of 32 rows (the partials cost a reduce pass)
    const int64_t want = cdiv64(tiles, 4 * kHeadWaves), cap = (int64_t)per_cu * head_cus();
    return {lds, (int)(want < 1 ? 1 : (want > cap ? cap : want))};
}
bool lds_path_off() {
    static const bool off = [] { const char* e = getenv("NEXTOU_HEAD_LDS"); return e != nullptr && e[0] == '0'; }();
    return off;
}

}  // namespace
}  // namespace nextou

using namespace nextou;

extern "C" int nextou_head_rows_fwd(const float* x, const float* w, const float* bias, float* y, int64_t P, int L, int C, int64_t ldx,
                                    int64_t ldy, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && w && y, "head_rows_fwd: null pointer");
    if (int e = check_head("head_rows_fwd", P, L, C, ldy, ldx)) return e;
    const int KT = cdiv(C, 16), LT = cdiv(L, 16);
    const size_t lds = (size_t)16 * (KT * 16 + 4) * sizeof(float);
    if (lds > 64 * 1024) return fail(NEXTOU_ENOTSUP, "head_rows_fwd: C=%d needs %zu bytes of LDS", C, lds);
    const int64_t tiles = cdiv64(P, 16);
    const int64_t want = cdiv64(tiles, (int64_t)kHeadWaves * kHeadPT);
    const int64_t cap = 8 * (int64_t)head_cus();
    const dim3 grid((unsigned)(want > cap ? cap : want), (unsigned)LT);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = C % 4 == 0 && ldx % 4 == 0 && aligned16(x);
    ProfScope prof(s, kBoundHbm, 4.0 * (double)P * (C + L), "head_fwd_kernel[P%lld C%d L%d]", (long long)P, C, L);
    if (vec)
        hipLaunchKernelGGL(head_fwd_kernel<true>, grid, dim3(kHeadThreads), lds, s, x, w, bias, y, (long)P, L, C, (long)ldx, (long)ldy, KT,
                           (long)tiles);
    else
        hipLaunchKernelGGL(head_fwd_kernel<false>, grid, dim3(kHeadThreads), lds, s, x, w, bias, y, (long)P, L, C, (long)ldx, (long)ldy, KT,
                           (long)tiles);
    return check_launch("head_rows_fwd");
}

extern "C" int nextou_head_rows_bwd_workspace(int64_t P, int L, int C, size_t* bytes) {
    NEXTOU_REQUIRE(bytes != nullptr, "head_rows_bwd_workspace: null pointer");
    if (int e = check_head("head_rows_bwd_workspace", P, L, C, L, C)) return e;
    const int CTtot = cdiv(C + 1, 16), LT = cdiv(L, 16);
    int blocks = wgrad_blocks(cdiv64(P, 16));
    const LdsPlan lp = plan_bwd_lds(P, L, C);                // the caller's strides / alignment decide later: room for either plan
    if (lp.lds != 0 && lp.blocks > blocks) blocks = lp.blocks;
    *bytes = (size_t)blocks * LT * 16 * CTtot * 16 * sizeof(float);
    return 0;
}

extern "C" int nextou_head_rows_bwd(const float* gy, const float* x, const float* w, float* gx, float* gw, float* gb, float* workspace,
                                    size_t workspace_bytes, int64_t P, int L, int C, int64_t ldg, int64_t ldx, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gy != nullptr, "head_rows_bwd: null pointer");
    if (int e = check_head("head_rows_bwd", P, L, C, ldg, ldx)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = cdiv64(P, 16);
    const int LT = cdiv(L, 16);
    const LdsPlan lp = plan_bwd_lds(P, L, C);
    if (gx != nullptr && (gw != nullptr || gb != nullptr) && lp.lds != 0 && !lds_path_off() && ldg == L && ldx == C && x != nullptr &&
        w != nullptr && workspace != nullptr && aligned16(gy) && aligned16(x) && aligned16(gx)) {
        // both gradients from one read of gy and x
        const int CT = cdiv(C, 16), CTtot = cdiv(C + 1, 16);
        const size_t need = (size_t)lp.blocks * 16 * CTtot * 16 * sizeof(float);
        if (workspace_bytes < need) return fail(NEXTOU_ENOSPACE, "head_rows_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
        {
            ProfScope prof(s, kBoundHbm, 4.0 * (double)P * (2 * C + L), "head_bwd_lds_kernel[P%lld C%d L%d]", (long long)P, C, L);
            const long t32 = (long)cdiv64(P, kHeadTileRows);
            const dim3 grid((unsigned)lp.blocks), block(kHeadThreads);
            if (C <= 16) hipLaunchKernelGGL(head_bwd_lds_kernel<2>, grid, block, lp.lds, s, gy, x, w, gx, workspace, (long)P, L, C, CT, CTtot, t32);
            else if (C <= 40) hipLaunchKernelGGL(head_bwd_lds_kernel<5>, grid, block, lp.lds, s, gy, x, w, gx, workspace, (long)P, L, C, CT, CTtot, t32);
            else if (C <= 72) hipLaunchKernelGGL(head_bwd_lds_kernel<9>, grid, block, lp.lds, s, gy, x, w, gx, workspace, (long)P, L, C, CT, CTtot, t32);
            else hipLaunchKernelGGL(head_bwd_lds_kernel<13>, grid, block, lp.lds, s, gy, x, w, gx, workspace, (long)P, L, C, CT, CTtot, t32);
            if (int e = check_launch("head_rows_bwd (fused)")) return e;
        }
        const long elems = (long)L * (C + 1);
        ProfScope prof(s, kBoundHbm, (double)need, "head_wgrad_reduce_kernel[L%d C%d x%d]", L, C, lp.blocks);
        hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((unsigned)cdiv64(elems, 16)), dim3(256), 0, s, workspace, gw, gb, lp.blocks, L, C, 1,
                           CTtot * 16);
        return check_launch("head_rows_bwd (reduce)");
    }
    if (gx != nullptr) {
        NEXTOU_REQUIRE(w != nullptr, "head_rows_bwd: the data gradient needs the weights");
        const int CT = cdiv(C, 16);
        if (LT > kHeadMaxLT) return fail(NEXTOU_ENOTSUP, "head_rows_bwd: L=%d > %d classes", L, 16 * kHeadMaxLT);
        const size_t lds = (size_t)LT * 16 * (CT * 16 + 4) * sizeof(float);
        if (lds > 96 * 1024) return fail(NEXTOU_ENOTSUP, "head_rows_bwd: L=%d C=%d need %zu bytes of LDS", L, C, lds);
        const int64_t want = cdiv64(tiles, (int64_t)kHeadWaves * kHeadPT);
        const int64_t cap = 8 * (int64_t)head_cus();
        const dim3 grid((unsigned)(want > cap ? cap : want));
        const bool vec = C % 4 == 0 && ldx % 4 == 0 && aligned16(gx);
        ProfScope prof(s, kBoundHbm, 4.0 * (double)P * (C + L), "head_dgrad_kernel[P%lld C%d L%d]", (long long)P, C, L);
        if (vec)
            hipLaunchKernelGGL(head_dgrad_kernel<true>, grid, dim3(kHeadThreads), lds, s, gy, w, gx, (long)P, L, C, (long)ldg, (long)ldx, CT, LT,
                               (long)tiles);
        else
            hipLaunchKernelGGL(head_dgrad_kernel<false>, grid, dim3(kHeadThreads), lds, s, gy, w, gx, (long)P, L, C, (long)ldg, (long)ldx, CT, LT,
                               (long)tiles);
        if (int e = check_launch("head_rows_bwd (data gradient)")) return e;
    }
    if (gw != nullptr || gb != nullptr) {
        NEXTOU_REQUIRE(x != nullptr && workspace != nullptr, "head_rows_bwd: the weight / bias gradient needs x and a workspace");
        const int CTtot = cdiv(C + 1, 16);
        const int blocks = wgrad_blocks(tiles);
        const size_t need = (size_t)blocks * LT * 16 * CTtot * 16 * sizeof(float);
        if (workspace_bytes < need) return fail(NEXTOU_ENOSPACE, "head_rows_bwd: workspace %zu < %zu bytes", workspace_bytes, need);
        {
            ProfScope prof(s, kBoundHbm, 4.0 * (double)P * (C + L), "head_wgrad_kernel[P%lld C%d L%d]", (long long)P, C, L);
            hipLaunchKernelGGL(head_wgrad_kernel, dim3((unsigned)blocks, (unsigned)cdiv(CTtot, kHeadCG), (unsigned)LT), dim3(kHeadThreads), 0, s, gy, x,
                               workspace, (long)P, L, C, (long)ldg, (long)ldx, CTtot, LT, (long)tiles);
            if (int e = check_launch("head_rows_bwd (weight gradient)")) return e;
        }
        const long elems = (long)L * (C + 1);
        ProfScope prof(s, kBoundHbm, (double)need, "head_wgrad_reduce_kernel[L%d C%d x%d]", L, C, blocks);
        hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((unsigned)cdiv64(elems, 16)), dim3(256), 0, s, workspace, gw, gb, blocks, L, C, LT,
                           CTtot * 16);
        if (int e = check_launch("head_rows_bwd (reduce)")) return e;
    }
    return 0;
}
