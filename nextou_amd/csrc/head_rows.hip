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

// gw[l, c] = sum_blocks part[b][l][c] (c < C), gb[l] = sum_blocks part[b][l][C]; blocks in ascending order per slice, slices
// (mod 4) combined as (s0 + s1) + (s2 + s3): the order pw_wgrad_reduce_kernel uses
__global__ __launch_bounds__(256) void head_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ gw, float* __restrict__ gb,
                                                                int blocks, int L, int C, int LT, int CW) {
    __shared__ float partial[4][64];
    const int slice = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * 64 + lane;            // over L x (C + 1)
    const long elems = (long)L * (C + 1);
    float s = 0.f;
    int l = 0, c = 0;
    if (e < elems) {
        l = (int)(e / (C + 1));
        c = (int)(e % (C + 1));
        const long stride = (long)LT * 16 * CW;
        const float* src = part + (long)l * CW + c;
        for (int i = slice; i < blocks; i += 4) s += src[(long)i * stride];
    }
    partial[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        const float t = (partial[0][lane] + partial[1][lane]) + (partial[2][lane] + partial[3][lane]);
        if (c < C) {
            if (gw != nullptr) gw[(long)l * C + c] = t;
        } else if (gb != nullptr) {
            gb[l] = t;
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
    *bytes = (size_t)wgrad_blocks(cdiv64(P, 16)) * LT * 16 * CTtot * 16 * sizeof(float);
    return 0;
}

extern "C" int nextou_head_rows_bwd(const float* gy, const float* x, const float* w, float* gx, float* gw, float* gb, float* workspace,
                                    size_t workspace_bytes, int64_t P, int L, int C, int64_t ldg, int64_t ldx, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gy != nullptr, "head_rows_bwd: null pointer");
    if (int e = check_head("head_rows_bwd", P, L, C, ldg, ldx)) return e;
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles = cdiv64(P, 16);
    const int LT = cdiv(L, 16);
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
        hipLaunchKernelGGL(head_wgrad_reduce_kernel, dim3((unsigned)cdiv64(elems, 64)), dim3(256), 0, s, workspace, gw, gb, blocks, L, C, LT,
                           CTtot * 16);
        if (int e = check_launch("head_rows_bwd (reduce)")) return e;
    }
    return 0;
}
