// K2 — max-relative aggregation (MRConv's gather / sub / max / interleave) and the plain
// neighbour gather, for gfx950.
//
// Reference op sequence replaced (network_architecture/NexToU_Encoder_Decoder.py:401-409,
// torch_nn.py:94-115): two batched_index_select calls that materialise (B,C,N,K) tensors,
// a subtraction, a max over K and a cat/reshape channel interleave.  Here a workgroup stages a
// chunk of source channel rows in LDS once and every lane gathers its K neighbours from LDS;
// the (B,C,N,K) tensors never exist.  HBM traffic is the algorithmic minimum:
//   fwd  4*B*C*(N+M_y) + 4*B*N*K(idx) + 4*B*2C*N        bwd  fwd + 4*B*C*(N+M_y)
//
// Layout: features (B,C,N) channel-major, idx (B,N,idx_stride) int32, out (B,2C,N) with
// channels interleaved [x_0, mr_0, x_1, mr_1, ...] (reference :409).
#include "common.h"
#include <cmath>
#include <cstdlib>

namespace nextou {

// linear copy of `count` floats (HBM -> LDS staging of consecutive channel rows, or LDS -> HBM)
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src,
                                           int count) {
    const bool vec = (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst)) & 15u) == 0) &&
                     ((count & 3) == 0);
    if (vec) {
        const float4* s4 = reinterpret_cast<const float4*>(src);
        float4* d4 = reinterpret_cast<float4*>(dst);
        for (int e = threadIdx.x; e < (count >> 2); e += blockDim.x) d4[e] = s4[e];
    } else {
        for (int e = threadIdx.x; e < count; e += blockDim.x) dst[e] = src[e];
    }
}

// ---------------------------------------------------------------------------------------------
// forward.  grid = (n_tiles, c_chunks, B); LDS = chunk * M floats.
// ---------------------------------------------------------------------------------------------
// WITH_ARG: also emit arg[b,c,n] = the source id that won the max (first max over the rounded
// differences, the element autograd's max backward routes the gradient to), as uint16, so that the
// backward pass is a pure scatter (mr_bwd_arg_kernel) instead of a second gather + arg-max.
template <int KB, bool SELF, bool WITH_ARG>
__global__ __launch_bounds__(512) void mr_fwd_lds_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const int32_t* __restrict__ idx,
    float* __restrict__ out, uint16_t* __restrict__ arg, int C, int N, int M, int K, int idx_stride,
    int idx_step, int chunk, int n_per_block) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * chunk;
    const int nc = (C - c0 < chunk) ? (C - c0) : chunk;
    stage_rows(lds, src + ((size_t)b * C + c0) * M, nc * M);
    __syncthreads();
    const int n_begin = blockIdx.x * n_per_block;
    int n_end = n_begin + n_per_block;
    if (n_end > N) n_end = N;
    for (int n = n_begin + threadIdx.x; n < n_end; n += blockDim.x) {
        int id[KB];
        const int32_t* irow = idx + ((size_t)b * N + n) * idx_stride;
#pragma unroll
        for (int j = 0; j < KB; ++j) id[j] = irow[(j < K ? j : 0) * idx_step];
        for (int c = 0; c < nc; ++c) {
            const float* row = lds + c * M;
            const float xv = SELF ? row[n] : x[((size_t)b * C + c0 + c) * N + n];
            float* o = out + ((size_t)b * 2 * C + 2 * (c0 + c)) * N + n;
            if (WITH_ARG) {
                float mx = row[id[0]] - xv;
                int am = id[0];
#pragma unroll
                for (int j = 1; j < KB; ++j) {
                    const float v = row[id[j]] - xv;
                    if (v > mx) { mx = v; am = id[j]; }  // strict: first max wins
                }
                o[0] = xv;
                o[N] = mx;
                arg[((size_t)b * C + c0 + c) * N + n] = (uint16_t)am;
            } else {
                float mx = row[id[0]];
#pragma unroll
                for (int j = 1; j < KB; ++j) mx = fmaxf(mx, row[id[j]]);
                o[0] = xv;
                o[N] = mx - xv;  // == max_j (src_j - x): rounding is monotone
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward, channel quads.  Same decomposition as above, but the LDS tile interleaves FOUR channels per source point:
// tile[q][m] is a float4 holding channels c0+4q .. c0+4q+3 of point m, so one ds_read_b128 gathers four channels of a
// neighbour (the random gather is the cost of this op: 16-B gathers spread 16 lanes over 16 bank groups — measured
// SQ_LDS_BANK_CONFLICT of the dword version: 1.4e7 of 2e7 LDS cycles on Pool s3) and the index arithmetic is shared by the
// four.  Staging stays coalesced (lanes along m, four row loads, one conflict-free ds_write_b128), the stores stay
// coalesced (lanes along n).  grid = (n_tiles, quad blocks, B); LDS = quads * M float4.
// ---------------------------------------------------------------------------------------------
template <int KB, bool SELF, bool WITH_ARG>
__global__ __launch_bounds__(512) void mr_fwd_q4_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const int32_t* __restrict__ idx,
    float* __restrict__ out, uint16_t* __restrict__ arg, int C, int N, int M, int K, int idx_stride,
    int idx_step, int quads, int n_per_block) {
    extern __shared__ __attribute__((aligned(16))) float4 tile4[];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * quads * 4;
    int nq = (C - c0 + 3) >> 2;
    if (nq > quads) nq = quads;
    const float* sb = src + ((size_t)b * C + c0) * M;
    for (int e = threadIdx.x; e < nq * M; e += blockDim.x) {
        const int q = e / M, m = e - q * M;
        const int c = 4 * q;
        const float* p = sb + (size_t)c * M + m;
        float4 v;
        v.x = p[0];
        v.y = (c0 + c + 1 < C) ? p[(size_t)M] : 0.f;
        v.z = (c0 + c + 2 < C) ? p[(size_t)2 * M] : 0.f;
        v.w = (c0 + c + 3 < C) ? p[(size_t)3 * M] : 0.f;
        tile4[e] = v;
    }
    __syncthreads();
    const int n_begin = blockIdx.x * n_per_block;
    int n_end = n_begin + n_per_block;
    if (n_end > N) n_end = N;
    for (int n = n_begin + threadIdx.x; n < n_end; n += blockDim.x) {
        int id[KB];
        const int32_t* irow = idx + ((size_t)b * N + n) * idx_stride;
        if (idx_step == 1) {    // immediate offsets: 32 scalar offset registers would otherwise stay live across the loop
#pragma unroll
            for (int j = 0; j < KB; ++j) id[j] = irow[j < K ? j : 0];
        } else {
#pragma unroll
            for (int j = 0; j < KB; ++j) id[j] = irow[(j < K ? j : 0) * idx_step];
        }
        for (int q = 0; q < nq; ++q) {
            const float4* row = tile4 + q * M;
            const int c = c0 + 4 * q;
            const bool v1 = c + 1 < C, v2 = c + 2 < C, v3 = c + 3 < C;
            float4 xv;
            if (SELF) {
                xv = row[n];
            } else {
                const float* xp = x + ((size_t)b * C + c) * N + n;
                xv.x = xp[0];
                xv.y = v1 ? xp[(size_t)N] : 0.f;
                xv.z = v2 ? xp[(size_t)2 * N] : 0.f;
                xv.w = v3 ? xp[(size_t)3 * N] : 0.f;
            }
            float4 mx;
            int a0, a1, a2, a3;
            {
                const float4 s = row[id[0]];
                mx.x = s.x - xv.x; mx.y = s.y - xv.y; mx.z = s.z - xv.z; mx.w = s.w - xv.w;
                a0 = a1 = a2 = a3 = id[0];
            }
            // groups of four gathers in flight: 16 VGPRs of payload and 16 compare masks at a time (the fully unrolled
            // K = 32 loop kept 128 payload VGPRs and spilled 269 SGPRs of masks)
#pragma unroll
            for (int j0 = 1; j0 < KB; j0 += 4) {
                float4 sv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) sv[u] = row[id[(j0 + u < KB) ? j0 + u : 0]];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (j0 + u >= KB) continue;
                    const int idj = id[(j0 + u < KB) ? j0 + u : 0];
                    const float d0 = sv[u].x - xv.x, d1 = sv[u].y - xv.y, d2 = sv[u].z - xv.z, d3 = sv[u].w - xv.w;
                    if (WITH_ARG) {     // strict >: the first maximum of the rounded differences wins (autograd's max)
                        if (d0 > mx.x) { mx.x = d0; a0 = idj; }
                        if (d1 > mx.y) { mx.y = d1; a1 = idj; }
                        if (d2 > mx.z) { mx.z = d2; a2 = idj; }
                        if (d3 > mx.w) { mx.w = d3; a3 = idj; }
                    } else {
                        mx.x = fmaxf(mx.x, d0); mx.y = fmaxf(mx.y, d1); mx.z = fmaxf(mx.z, d2); mx.w = fmaxf(mx.w, d3);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the groups apart: do not hoist the next group's gathers
            }
            float* o = out + ((size_t)b * 2 * C + 2 * c) * N + n;
            o[0] = xv.x;
            o[(size_t)N] = mx.x;
            if (v1) { o[(size_t)2 * N] = xv.y; o[(size_t)3 * N] = mx.y; }
            if (v2) { o[(size_t)4 * N] = xv.z; o[(size_t)5 * N] = mx.z; }
            if (v3) { o[(size_t)6 * N] = xv.w; o[(size_t)7 * N] = mx.w; }
            if (WITH_ARG) {
                uint16_t* ap = arg + ((size_t)b * C + c) * N + n;
                ap[0] = (uint16_t)a0;
                if (v1) ap[(size_t)N] = (uint16_t)a1;
                if (v2) ap[(size_t)2 * N] = (uint16_t)a2;
                if (v3) ap[(size_t)3 * N] = (uint16_t)a3;
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// forward, channel quads with the NEIGHBOUR LIST split over the four lanes of a DPP quad (round 3).  The pooled graphs
// (K = 14 ... 32 ids per query) kept the dword kernel in round 2 because 28 ids + their gathered values per lane cost
// 2 waves per SIMD; its counters (profiles/r03_sq_counters_k2_pool_s3.md) say where the time goes: 70 % of the LDS cycles are
// bank conflicts of the random ds_read_b32 gathers (6.9 cycles per wave instruction against 2 conflict-free) behind 0.9 waves per
// SIMD.  Here lane (query, kg) owns the KL = ceil(K / 4) neighbours j = kg * KL .. kg * KL + KL - 1 of its query: KL ids and KL
// 16-byte gathers in flight (the quad tile of mr_fwd_q4_kernel: one ds_read_b128 moves four channels, ~1.5x fewer LDS cycles per
// element than the dword gather under random conflicts), ~60 VGPRs, and the four partial (max, arg) pairs meet in two
// quad_perm steps.  First-maximum-wins is preserved: the lanes' blocks are ascending in j, blocks after the first start from
// -inf with the strict compare (so they skip what the sequential scan skips), and a pair keeps the lower lane's candidate
// unless the higher one is strictly greater.
// grid = (n_tiles, quad blocks, B), 256 threads = 64 queries per pass; LDS = quads * M float4.
// ---------------------------------------------------------------------------------------------
template <int CTRL>
__device__ __forceinline__ float dpp_f(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_update_dpp(0, v, CTRL, 0xF, 0xF, true); }

constexpr int kKqMaxQuads = 4;       // channel quads per workgroup (the planner never asks for more)

template <int KL, bool SELF, bool WITH_ARG>
__global__ __launch_bounds__(256) void mr_fwd_kq_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const int32_t* __restrict__ idx,
    float* __restrict__ out, uint16_t* __restrict__ arg, int C, int N, int M, int K, int idx_stride,
    int idx_step, int quads, int n_per_block) {
    extern __shared__ __attribute__((aligned(16))) float4 tile4[];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * quads * 4;
    int nq = (C - c0 + 3) >> 2;
    if (nq > quads) nq = quads;
    const float* sb = src + ((size_t)b * C + c0) * M;
    for (int e = threadIdx.x; e < nq * M; e += blockDim.x) {
        const int q = e / M, m = e - q * M;
        const int c = 4 * q;
        const float* p = sb + (size_t)c * M + m;
        float4 v;
        v.x = p[0];
        v.y = (c0 + c + 1 < C) ? p[(size_t)M] : 0.f;
        v.z = (c0 + c + 2 < C) ? p[(size_t)2 * M] : 0.f;
        v.w = (c0 + c + 3 < C) ? p[(size_t)3 * M] : 0.f;
        tile4[e] = v;
    }
    __syncthreads();
    const int kg = threadIdx.x & 3, ql = threadIdx.x >> 2;          // 64 queries per pass, 4 lanes each
    const int n_begin = blockIdx.x * n_per_block;
    int n_end = n_begin + n_per_block;
    if (n_end > N) n_end = N;
    const int j_first = kg * KL;
    // The ids and centre values of pass p + 1 are fetched while pass p computes: un-prefetched, every pass began with a global load
    // whose latency nothing covered (the first version of this kernel: 74 us on Pool s3 for ~20 us of LDS / VALU work).
    int idn[KL];
    float xn[kKqMaxQuads];
    auto fetch = [&](int nb) __attribute__((always_inline)) {
        int n = nb + ql;
        if (n >= n_end) n = n_end - 1;                                // clamped: every lane of a quad takes part in the DPP steps
        const int32_t* irow = idx + ((size_t)b * N + n) * idx_stride;
#pragma unroll
        for (int t = 0; t < KL; ++t) idn[t] = irow[(size_t)((j_first + t < K) ? j_first + t : 0) * idx_step];
        if (!SELF) {
#pragma unroll
            for (int q = 0; q < kKqMaxQuads; ++q) {
                const int c = c0 + 4 * q + kg;
                xn[q] = (q < nq && c < C) ? x[((size_t)b * C + c) * N + n] : 0.f;
            }
        }
    };
    fetch(n_begin);
    for (int nb = n_begin; nb < n_end; nb += 64) {
        const int n = nb + ql;
        const bool live = n < n_end;
        const int nc = live ? n : n_end - 1;
        int id[KL];
        float xc[kKqMaxQuads];
#pragma unroll
        for (int t = 0; t < KL; ++t) id[t] = idn[t];
#pragma unroll
        for (int q = 0; q < kKqMaxQuads; ++q) xc[q] = xn[q];
        if (nb + 64 < n_end) fetch(nb + 64);
#pragma unroll
        for (int q = 0; q < kKqMaxQuads; ++q) {
            if (q >= nq) break;
            const float4* row = tile4 + q * M;
            const int c = c0 + 4 * q;
            // the centre values of the quad's four channels: lane kg holds channel c + kg, quad_perm broadcasts spread them
            float xm;
            if (SELF) {
                const float4 r = row[nc];
                xm = kg == 0 ? r.x : (kg == 1 ? r.y : (kg == 2 ? r.z : r.w));
            } else {
                xm = xc[q];
            }
            float4 xv;
            xv.x = dpp_f<0x00>(xm);     // quad_perm [0,0,0,0]
            xv.y = dpp_f<0x55>(xm);     // [1,1,1,1]
            xv.z = dpp_f<0xAA>(xm);     // [2,2,2,2]
            xv.w = dpp_f<0xFF>(xm);     // [3,3,3,3]
            float4 sv[KL];
#pragma unroll
            for (int t = 0; t < KL; ++t) sv[t] = row[id[t]];
            float4 mx;
            int a0, a1, a2, a3;
            if (kg == 0) {              // the scan's first element initialises the maximum (NaN included, as in the sequential form)
                mx.x = sv[0].x - xv.x; mx.y = sv[0].y - xv.y; mx.z = sv[0].z - xv.z; mx.w = sv[0].w - xv.w;
            } else {
                mx.x = mx.y = mx.z = mx.w = -INFINITY;
            }
            a0 = a1 = a2 = a3 = id[0];
#pragma unroll
            for (int t = 0; t < KL; ++t) {
                if (j_first + t >= K) continue;          // (uniform per lane; the padded slots hold id[0] of the row)
                const float d0 = sv[t].x - xv.x, d1 = sv[t].y - xv.y, d2 = sv[t].z - xv.z, d3 = sv[t].w - xv.w;
                if (d0 > mx.x) { mx.x = d0; a0 = id[t]; }
                if (d1 > mx.y) { mx.y = d1; a1 = id[t]; }
                if (d2 > mx.z) { mx.z = d2; a2 = id[t]; }
                if (d3 > mx.w) { mx.w = d3; a3 = id[t]; }
            }
            // pairs (kg ^ 1), then (kg ^ 2): the lower lane's candidate stays unless the higher one is strictly greater
#define NEXTOU_KQ_MERGE(CTRL, BIT, V, A)                                          \
            {                                                                     \
                const float pv = dpp_f<CTRL>(V);                                  \
                const int pa = dpp_i<CTRL>(A);                                    \
                const bool take = (kg & BIT) ? !(V > pv) : (pv > V);              \
                V = take ? pv : V;                                                \
                A = take ? pa : A;                                                \
            }
            NEXTOU_KQ_MERGE(0xB1, 1, mx.x, a0) NEXTOU_KQ_MERGE(0xB1, 1, mx.y, a1) NEXTOU_KQ_MERGE(0xB1, 1, mx.z, a2) NEXTOU_KQ_MERGE(0xB1, 1, mx.w, a3)
            NEXTOU_KQ_MERGE(0x4E, 2, mx.x, a0) NEXTOU_KQ_MERGE(0x4E, 2, mx.y, a1) NEXTOU_KQ_MERGE(0x4E, 2, mx.z, a2) NEXTOU_KQ_MERGE(0x4E, 2, mx.w, a3)
#undef NEXTOU_KQ_MERGE
            // lane kg writes channel c + kg of its query: 16 queries x 4 channel rows per wave instruction
            const float mv = kg == 0 ? mx.x : (kg == 1 ? mx.y : (kg == 2 ? mx.z : mx.w));
            const int av = kg == 0 ? a0 : (kg == 1 ? a1 : (kg == 2 ? a2 : a3));
            if (live && c + kg < C) {
                float* o = out + ((size_t)b * 2 * C + 2 * (c + kg)) * N + n;
                o[0] = xm;
                o[(size_t)N] = mv;
                if (WITH_ARG) arg[((size_t)b * C + c + kg) * N + n] = (uint16_t)av;
            }
        }
    }
}

// generic forward: arbitrary centre ids and/or source rows too long for LDS; gathers from
// global memory (L2-resident rows).  One thread per (b, c, n).
__global__ __launch_bounds__(256) void mr_fwd_global_kernel(
    const float* __restrict__ x, const float* __restrict__ src, const int32_t* __restrict__ idx,
    const int32_t* __restrict__ ctr, float* __restrict__ out, int C, int N, int M, int K,
    int idx_stride, int idx_step) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int b = blockIdx.z;
    if (n >= N) return;
    const float* xrow = x + ((size_t)b * C + c) * N;
    const float* srow = src + ((size_t)b * C + c) * M;
    const size_t ioff = ((size_t)b * N + n) * idx_stride;
    float mx = -INFINITY;
    for (int j = 0; j < K; ++j) {
        const float xc = ctr ? xrow[ctr[ioff + (size_t)j * idx_step]] : xrow[n];
        const float v = srow[idx[ioff + (size_t)j * idx_step]] - xc;
        mx = (j == 0) ? v : fmaxf(mx, v);
    }
    float* o = out + ((size_t)b * 2 * C + 2 * c) * N + n;
    o[0] = xrow[n];
    o[N] = mx;
}

// ---------------------------------------------------------------------------------------------
// backward, LDS accumulate.  grid = (n_tiles, c_chunks, B); LDS = 2 * chunk * M floats
// (source rows + gradient accumulators).
//   SELF  (src == x, n_tiles == 1): accumulators start at g_x - g_mr, receive the scattered g_mr
//         and are stored to dx — one pass, no atomics outside LDS.
//   !SELF (src == y): dx = g_x - g_mr is written directly; the accumulators are flushed into the
//         pre-zeroed dy with one global atomic per touched (c, m).
// ---------------------------------------------------------------------------------------------
template <int KB, bool SELF>
__global__ __launch_bounds__(512) void mr_bwd_lds_kernel(
    const float* __restrict__ gout, const float* __restrict__ x, const float* __restrict__ src,
    const int32_t* __restrict__ idx, float* __restrict__ dx, float* __restrict__ dsrc, int C, int N,
    int M, int K, int idx_stride, int idx_step, int chunk, int n_per_block) {
    extern __shared__ float lds[];
    const int b = blockIdx.z;
    const int c0 = blockIdx.y * chunk;
    const int nc = (C - c0 < chunk) ? (C - c0) : chunk;
    float* vals = lds;               // [nc][M] source values
    float* accg = lds + chunk * M;   // [nc][M] gradient accumulators
    stage_rows(vals, src + ((size_t)b * C + c0) * M, nc * M);
    if (SELF) {
        for (int e = threadIdx.x; e < nc * M; e += blockDim.x) {
            const int c = e / M, m = e - c * M;
            const float* g = gout + ((size_t)b * 2 * C + 2 * (c0 + c)) * N + m;
            accg[e] = g[0] - g[N];
        }
    } else {
        for (int e = threadIdx.x; e < nc * M; e += blockDim.x) accg[e] = 0.f;
    }
    __syncthreads();
    const int n_begin = blockIdx.x * n_per_block;
    int n_end = n_begin + n_per_block;
    if (n_end > N) n_end = N;
    for (int n = n_begin + threadIdx.x; n < n_end; n += blockDim.x) {
        int id[KB];
        const int32_t* irow = idx + ((size_t)b * N + n) * idx_stride;
#pragma unroll
        for (int j = 0; j < KB; ++j) id[j] = irow[(j < K ? j : 0) * idx_step];
        for (int c = 0; c < nc; ++c) {
            const float* row = vals + c * M;
            const float xv = SELF ? row[n] : x[((size_t)b * C + c0 + c) * N + n];
            float mx = row[id[0]] - xv;
            int am = id[0];
#pragma unroll
            for (int j = 1; j < KB; ++j) {
                const float v = row[id[j]] - xv;  // the rounded difference autograd's max saw
                if (v > mx) { mx = v; am = id[j]; }  // strict: first max wins
            }
            const float* g = gout + ((size_t)b * 2 * C + 2 * (c0 + c)) * N + n;
            const float gm = g[N];
            atomicAdd(&accg[c * M + am], gm);
            if (!SELF) dx[((size_t)b * C + c0 + c) * N + n] = g[0] - gm;
        }
    }
    __syncthreads();
    if (SELF) {
        stage_rows(dx + ((size_t)b * C + c0) * N, accg, nc * M);  // LDS -> HBM, same linear copy
    } else {
        float* drow = dsrc + ((size_t)b * C + c0) * M;
        for (int e = threadIdx.x; e < nc * M; e += blockDim.x) {
            const float v = accg[e];
            if (v != 0.f) atomicAdd(&drow[e], v);
        }
    }
}

// backward from the saved arg-max ids: a pure scatter-add, no gathers, no ids, no source rows.
//   grid = (c_chunks, B); LDS = chunk * M accumulators.  The workgroup owns its channels for ALL
//   points, so dx / dy rows are written exactly once with plain stores (no global atomics).
//   One pass over (c, n), 4 independent elements in flight per lane.  SELF: the identity-branch
//   term g_x - g_mr is added to the accumulator too and the accumulator is stored as dx.
//   !SELF: dx = g_x - g_mr is written on the way and the accumulator is stored as dy.
template <bool SELF, bool VEC4>
__global__ __launch_bounds__(256) void mr_bwd_arg_kernel(const float* __restrict__ gout,
                                                         const uint16_t* __restrict__ arg,
                                                         float* __restrict__ dx, float* __restrict__ dy,
                                                         int C, int N, int M, int chunk, unsigned magic) {
    // VEC4 (N % 4 == 0, 16-B aligned rows): a lane owns 4 consecutive points of one channel row and
    // moves them with 16-B loads / stores (8 B for the 4 arg ids).
    // LDS float atomics retire ~0.5 lane per clock per CU on gfx950 (rocprofv3: SQ_LDS_IDX_ACTIVE ~ 120
    // cycles per ds_add_f32 wave-instruction, profiles/r01_sq_counters.md), so only the scattered g_mr
    // term goes through them; the identity term g_x - g_mr is added with plain arithmetic on the way out.
    constexpr int W = VEC4 ? 4 : 1;   // points per item
    constexpr int U = VEC4 ? 2 : 4;   // independent items in flight per lane
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * chunk;
    const int nc = (C - c0 < chunk) ? (C - c0) : chunk;
    for (int e = threadIdx.x; e < nc * M; e += blockDim.x) lds[e] = 0.f;
    __syncthreads();
    const unsigned row_items = (unsigned)N / W;          // items per channel row
    const unsigned total = (unsigned)nc * row_items;
    const float* gbase = gout + ((size_t)b * 2 * C + 2 * c0) * N;
    const uint16_t* abase = arg + ((size_t)b * C + c0) * N;
    float* dxbase = dx + ((size_t)b * C + c0) * N;
    // phase 1: scatter g_mr into the accumulators (and, for the pooled graph, write dx = g_x - g_mr)
    for (unsigned base = 0; base < total; base += blockDim.x * U) {
        float g0[U][W], g1[U][W];
        unsigned short a[U][W];
        unsigned cc[U], nn[U];
        bool ok[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned it = base + u * blockDim.x + threadIdx.x;
            ok[u] = it < total;
            const unsigned c = (row_items == 1) ? it : __umulhi(it, magic);  // it / row_items (exact, see host)
            const unsigned n = (it - c * row_items) * W;
            cc[u] = c;
            nn[u] = n;
            const float* g = gbase + (size_t)2 * c * N + n;
            const uint16_t* ap = abase + (size_t)c * N + n;
            if (VEC4) {
                float4 v0 = make_float4(0.f, 0.f, 0.f, 0.f), v1 = v0;
                uint2 av = make_uint2(0u, 0u);
                if (ok[u]) {
                    if (!SELF) v0 = *reinterpret_cast<const float4*>(g);
                    v1 = *reinterpret_cast<const float4*>(g + N);
                    av = *reinterpret_cast<const uint2*>(ap);
                }
                g0[u][0] = v0.x; g0[u][1 % W] = v0.y; g0[u][2 % W] = v0.z; g0[u][3 % W] = v0.w;
                g1[u][0] = v1.x; g1[u][1 % W] = v1.y; g1[u][2 % W] = v1.z; g1[u][3 % W] = v1.w;
                a[u][0] = (unsigned short)(av.x & 0xffffu); a[u][1 % W] = (unsigned short)(av.x >> 16);
                a[u][2 % W] = (unsigned short)(av.y & 0xffffu); a[u][3 % W] = (unsigned short)(av.y >> 16);
            } else {
                g0[u][0] = (ok[u] && !SELF) ? g[0] : 0.f;
                g1[u][0] = ok[u] ? g[N] : 0.f;
                a[u][0] = ok[u] ? ap[0] : (unsigned short)0;
            }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!ok[u]) continue;
            float* acc = lds + cc[u] * M;
#pragma unroll
            for (int w = 0; w < W; ++w) atomicAdd(&acc[a[u][w]], g1[u][w]);
            if (!SELF) {
                float* o = dxbase + (size_t)cc[u] * N + nn[u];
                if (VEC4)
                    *reinterpret_cast<float4*>(o) = make_float4(g0[u][0] - g1[u][0], g0[u][1 % W] - g1[u][1 % W],
                                                                g0[u][2 % W] - g1[u][2 % W], g0[u][3 % W] - g1[u][3 % W]);
                else
                    o[0] = g0[u][0] - g1[u][0];
            }
        }
    }
    __syncthreads();
    if (!SELF) {
        stage_rows(dy + ((size_t)b * C + c0) * M, lds, nc * M);
        return;
    }
    // phase 2 (self graph): dx = scattered + (g_x - g_mr); g_mr is re-read (L2-hot)
    for (unsigned base = 0; base < total; base += blockDim.x * U) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const unsigned it = base + u * blockDim.x + threadIdx.x;
            if (it >= total) continue;
            const unsigned c = (row_items == 1) ? it : __umulhi(it, magic);
            const unsigned n = (it - c * row_items) * W;
            const float* g = gbase + (size_t)2 * c * N + n;
            const float* acc = lds + c * M + n;
            float* o = dxbase + (size_t)c * N + n;
            if (VEC4) {
                const float4 v0 = *reinterpret_cast<const float4*>(g);
                const float4 v1 = *reinterpret_cast<const float4*>(g + N);
                const float4 s4 = *reinterpret_cast<const float4*>(acc);
                *reinterpret_cast<float4*>(o) = make_float4(s4.x + (v0.x - v1.x), s4.y + (v0.y - v1.y),
                                                            s4.z + (v0.z - v1.z), s4.w + (v0.w - v1.w));
            } else {
                o[0] = acc[0] + (g[0] - g[N]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// backward of a SELF window graph (N = M <= 512) as a GATHER over reverse neighbour lists — no float atomics.
//   dx[c, m] = (g_x - g_mr)[c, m] + sum over the queries n that list m among their neighbours of [arg[c, n] == m] * g_mr[c, n]
// The scatter version above spends 81 % of its wave cycles waiting on ds_add_f32 (rocprofv3 SQ_WAIT_INST_LDS 4.9e8 of 6.05e8,
// ~118 LDS cycles per wave-instruction, profiles/r01_sq_counters.md).  Here a workgroup (one window, a block of channel
// quads) first builds the reverse lists of the window in LDS — count (integer LDS atomics, N*K of them instead of C*N float
// ones), scan, fill, then every lane sorts its own short segment, which makes the summation order fixed: gradients are
// bit-reproducible run to run — and then lane m walks its list once per channel quad, reading the quad's g_mr (float4) and
// arg ids (4 x uint16) of each listed query from LDS.  A query that lists m twice (possible for hand-made index tensors,
// not for kNN output) counts once, as the forward's first-maximum rule does.
// grid = (quad blocks, B); LDS = quads * N * 24 B + (N + 1) * 4 + N * 4 + N * K * 2.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void mr_bwd_rev_kernel(const float* __restrict__ gout, const uint16_t* __restrict__ arg,
                                                         const int32_t* __restrict__ idx, float* __restrict__ dx, int C, int N,
                                                         int K, int idx_stride, int idx_step, int quads) {
    extern __shared__ __attribute__((aligned(16))) unsigned char rev_lds[];
    const int b = blockIdx.y;
    const int c0 = blockIdx.x * quads * 4;
    int nq = (C - c0 + 3) >> 2;
    if (nq > quads) nq = quads;
    float4* g4 = reinterpret_cast<float4*>(rev_lds);                                  // [quads][N]
    uint2* a4 = reinterpret_cast<uint2*>(rev_lds + (size_t)quads * N * 16);           // [quads][N]  4 x uint16
    int* off = reinterpret_cast<int*>(rev_lds + (size_t)quads * N * 24);              // [N + 1]
    int* cur = off + (N + 1);                                                         // [N]
    uint16_t* ent = reinterpret_cast<uint16_t*>(cur + N);                             // [N * K]
    const int tid = threadIdx.x;
    // ---- stage g_mr and arg, quad-interleaved (lanes along n: coalesced row reads, conflict-free 16 / 8-byte writes)
    const float* gb = gout + ((size_t)b * 2 * C + 2 * c0) * N;
    const uint16_t* ab = arg + ((size_t)b * C + c0) * N;
    for (int e = tid; e < nq * N; e += blockDim.x) {
        const int q = e / N, n = e - q * N;
        const int c = 4 * q;
        const bool v1 = c0 + c + 1 < C, v2 = c0 + c + 2 < C, v3 = c0 + c + 3 < C;
        const float* gp = gb + (size_t)(2 * c + 1) * N + n;
        const uint16_t* ap = ab + (size_t)c * N + n;
        float4 g;
        g.x = gp[0];
        g.y = v1 ? gp[(size_t)2 * N] : 0.f;
        g.z = v2 ? gp[(size_t)4 * N] : 0.f;
        g.w = v3 ? gp[(size_t)6 * N] : 0.f;
        const unsigned a0 = ap[0], a1 = v1 ? ap[(size_t)N] : 0xffffu, a2 = v2 ? ap[(size_t)2 * N] : 0xffffu,
                       a3 = v3 ? ap[(size_t)3 * N] : 0xffffu;
        g4[e] = g;
        a4[e] = make_uint2(a0 | (a1 << 16), a2 | (a3 << 16));
    }
    // ---- reverse lists of this window
    for (int m = tid; m <= N; m += blockDim.x) off[m] = 0;
    for (int m = tid; m < N; m += blockDim.x) cur[m] = 0;
    __syncthreads();
    const int32_t* ib = idx + (size_t)b * N * idx_stride;
    for (int e = tid; e < N * K; e += blockDim.x) {
        const int n = e / K, j = e - n * K;
        atomicAdd(&off[ib[(size_t)n * idx_stride + (size_t)j * idx_step] + 1], 1);
    }
    __syncthreads();
    if (tid < 64) {        // inclusive scan of off[1..N] by one wave: N <= 512 -> 8 values per lane
        const int per = (N + 63) / 64;
        int local = 0;
        for (int i = 0; i < per; ++i) { const int m = tid * per + i; if (m < N) local += off[m + 1]; }
        int run = local;
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) { const int t = __shfl_up(run, d, 64); if (tid >= d) run += t; }
        int base = run - local;
        for (int i = 0; i < per; ++i) { const int m = tid * per + i; if (m < N) { base += off[m + 1]; off[m + 1] = base; } }
    }
    __syncthreads();
    for (int e = tid; e < N * K; e += blockDim.x) {
        const int n = e / K, j = e - n * K;
        const int m = ib[(size_t)n * idx_stride + (size_t)j * idx_step];
        ent[off[m] + atomicAdd(&cur[m], 1)] = (uint16_t)n;
    }
    __syncthreads();
    for (int m = tid; m < N; m += blockDim.x) {     // insertion sort of the lane's own segment (a handful of entries)
        const int lo = off[m], hi = off[m + 1];
        for (int i = lo + 1; i < hi; ++i) {
            const uint16_t v = ent[i];
            int k = i - 1;
            while (k >= lo && ent[k] > v) { ent[k + 1] = ent[k]; --k; }
            ent[k + 1] = v;
        }
    }
    __syncthreads();
    // ---- gather: lane m, every channel quad of the block
    for (int m = tid; m < N; m += blockDim.x) {
        const int lo = off[m], hi = off[m + 1];
        const unsigned um = (unsigned)m;
        for (int q = 0; q < nq; ++q) {
            const float4* gq = g4 + q * N;
            const uint2* aq = a4 + q * N;
            float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
            int prev = -1;
            for (int e = lo; e < hi; ++e) {
                const int n = ent[e];
                if (n == prev) continue;
                prev = n;
                const uint2 a = aq[n];
                const float4 g = gq[n];
                s0 += ((a.x & 0xffffu) == um) ? g.x : 0.f;
                s1 += ((a.x >> 16) == um) ? g.y : 0.f;
                s2 += ((a.y & 0xffffu) == um) ? g.z : 0.f;
                s3 += ((a.y >> 16) == um) ? g.w : 0.f;
            }
            const int c = c0 + 4 * q;
            const float4 gm = gq[m];
            const float* gx = gout + ((size_t)b * 2 * C + 2 * c) * N + m;
            float* o = dx + ((size_t)b * C + c) * N + m;
            o[0] = s0 + (gx[0] - gm.x);
            if (c + 1 < C) o[(size_t)N] = s1 + (gx[(size_t)2 * N] - gm.y);
            if (c + 2 < C) o[(size_t)2 * N] = s2 + (gx[(size_t)4 * N] - gm.z);
            if (c + 3 < C) o[(size_t)3 * N] = s3 + (gx[(size_t)6 * N] - gm.w);
        }
    }
}

// generic backward: global atomics, arbitrary centre ids.  dx / dsrc pre-zeroed by the host.
__global__ __launch_bounds__(256) void mr_bwd_global_kernel(
    const float* __restrict__ gout, const float* __restrict__ x, const float* __restrict__ src,
    const int32_t* __restrict__ idx, const int32_t* __restrict__ ctr, float* __restrict__ dx,
    float* __restrict__ dsrc, int C, int N, int M, int K, int idx_stride, int idx_step) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y;
    const int b = blockIdx.z;
    if (n >= N) return;
    const float* xrow = x + ((size_t)b * C + c) * N;
    const float* srow = src + ((size_t)b * C + c) * M;
    const size_t ioff = ((size_t)b * N + n) * idx_stride;
    float mx = 0.f;
    int am = 0, ac = n;
    for (int j = 0; j < K; ++j) {
        const int cj = ctr ? ctr[ioff + (size_t)j * idx_step] : n;
        const int sj = idx[ioff + (size_t)j * idx_step];
        const float v = srow[sj] - xrow[cj];
        if (j == 0 || v > mx) { mx = v; am = sj; ac = cj; }
    }
    const float* g = gout + ((size_t)b * 2 * C + 2 * c) * N + n;
    const float gm = g[N];
    atomicAdd(&dx[((size_t)b * C + c) * N + n], g[0]);
    atomicAdd(&dx[((size_t)b * C + c) * N + ac], -gm);
    atomicAdd(&dsrc[((size_t)b * C + c) * M + am], gm);
}

// ---------------------------------------------------------------------------------------------
// batched_index_select forward / backward
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gather_fwd_kernel(const float* __restrict__ src,
                                                         const int32_t* __restrict__ idx,
                                                         float* __restrict__ out, int C, int M,
                                                         long long NK) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // over N*K
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= NK) return;
    out[((size_t)b * C + c) * NK + e] = src[((size_t)b * C + c) * M + idx[(size_t)b * NK + e]];
}

__global__ __launch_bounds__(256) void gather_bwd_kernel(const float* __restrict__ gout,
                                                         const int32_t* __restrict__ idx,
                                                         float* __restrict__ dsrc, int C, int M,
                                                         long long NK) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const int c = blockIdx.y, b = blockIdx.z;
    if (e >= NK) return;
    atomicAdd(&dsrc[((size_t)b * C + c) * M + idx[(size_t)b * NK + e]],
              gout[((size_t)b * C + c) * NK + e]);
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
struct MrPlan {
    int chunk, c_chunks, n_tiles, n_per_block, threads;
    size_t lds;
};

// rows_per_channel: LDS floats needed per channel (M forward, 2M backward)
static bool plan_lds(int B, int C, int N, int M, int floats_per_channel, bool tile_n, MrPlan* p) {
    const int budget = kGatherLdsBytes / (int)sizeof(float);
    int max_chunk = budget / floats_per_channel;
    if (max_chunk < 1) return false;
    if (max_chunk > C) max_chunk = C;
    int threads = ((N < 256 ? N : 256) + 63) / 64 * 64;
    int n_per_block = N, n_tiles = 1;
    if (tile_n && N > 1024) {
        n_per_block = 1024;
        n_tiles = cdiv(N, n_per_block);
        // pooled graphs: the 64 KB tile allows two workgroups per CU; 512 threads each give the random LDS gathers
        // 16 waves per CU to hide behind instead of 8 (NEXTOU_MR_THREADS=256 restores round 1 for A/B)
        const char* e = getenv("NEXTOU_MR_THREADS");
        threads = (e && atoi(e) == 256) ? 256 : 512;
    }
    // shrink the channel chunk until the grid has >= 512 workgroups; keep >= 4 channels per
    // workgroup when N is tiled so the idx registers are reused across channels
    int chunk = max_chunk;
    while (chunk > 4 && (long long)cdiv(C, chunk) * n_tiles * B < 512) chunk = (chunk + 1) / 2;
    if (!tile_n) {
        while (chunk > 1 && (long long)cdiv(C, chunk) * n_tiles * B < 512) chunk = (chunk + 1) / 2;
    }
    chunk = cdiv(C, cdiv(C, chunk));  // balance the last chunk
    p->chunk = chunk;
    p->c_chunks = cdiv(C, chunk);
    p->n_tiles = n_tiles;
    p->n_per_block = n_per_block;
    p->threads = threads;
    p->lds = (size_t)chunk * floats_per_channel * sizeof(float);
    return p->c_chunks <= 65535 && B <= 65535;
}

// Work decomposition of mr_fwd_q4_kernel.  One channel quad costs 16 * M bytes of LDS.  Small source sets (windows) take
// ~20 KB tiles so that many workgroups share a CU; long ones (pooled candidate sets of 1344 / 3072 points) take what fits
// 48 KB, and the query range is cut into tiles until the grid has a few workgroups per CU.
struct Q4Plan {
    int quads, q_blocks, n_tiles, n_per_block, threads;
    size_t lds;
};
static bool plan_q4(int B, int C, int N, int M, int K, bool self, Q4Plan* p) {
    // Measured (profiles/r02_kernel_bench_k2.md): the quad kernel wins where the LDS gather is the bound and the id list
    // is short — the cfg-2 stage-2 windows, K = 7 (79.8 -> 66.2 us = 61 % of 8 TB/s) — is level at K = 14 and loses beyond
    // (K = 16 windows of 384 points 233 -> 319 us; pooled graphs with K = 28 / 32 ids + 32 float4 in flight = 208 VGPRs,
    // 2 waves per SIMD: Pool s3 73 -> 95 us), which keep the dword kernel.
    // NEXTOU_MR_FWD=v1 | q4 forces one of them for A/B runs.
    const char* force = getenv("NEXTOU_MR_FWD");
    if (force && force[0] == 'v') return false;
    if (!(force && force[0] == 'q') && !(self && K <= 8 && N <= 512)) return false;
    const size_t per_quad = (size_t)M * 16;
    if (per_quad > 152 * 1024) return false;
    const int total_quads = (C + 3) / 4;
    size_t budget = M <= 512 ? 20 * 1024 : 48 * 1024;
    int quads = (int)(budget / per_quad);
    if (quads < 1) quads = 1;
    if (quads > total_quads) quads = total_quads;
    while (quads > 1 && (long long)cdiv(total_quads, quads) * B < 512 && N <= 1024) quads = (quads + 1) / 2;   // fill the chip
    quads = cdiv(total_quads, cdiv(total_quads, quads));      // balance the last block
    p->quads = quads;
    p->q_blocks = cdiv(total_quads, quads);
    p->lds = (size_t)quads * per_quad;
    int threads = N >= 512 ? 512 : ((N + 63) / 64) * 64;
    int n_per_block = N, n_tiles = 1;
    if (N > 1024) {
        // >= ~3 workgroups per CU, at least 2 queries per lane so that the staged tile is reused
        long long want = cdiv64(768, (long long)p->q_blocks * B);
        long long max_tiles = N / (2 * threads);
        if (want > max_tiles) want = max_tiles;
        if (want < 1) want = 1;
        n_tiles = (int)want;
        n_per_block = cdiv(cdiv(N, n_tiles), 64) * 64;
        n_tiles = cdiv(N, n_per_block);
    }
    p->threads = threads;
    p->n_tiles = n_tiles;
    p->n_per_block = n_per_block;
    return p->q_blocks <= 65535 && B <= 65535;
}

// Work decomposition of mr_fwd_kq_kernel: the quad tile of plan_q4 (16 * M bytes per channel quad; ~20 KB tiles for windows, what
// fits 48 KB for the pooled candidate sets), 256 threads = 64 queries per pass, the query range cut until the grid has ~4
// workgroups per CU while every workgroup still makes >= 2 passes over its staged tile.
static bool plan_kq(int B, int C, int N, int M, int K, bool self, Q4Plan* p) {
    const char* force = getenv("NEXTOU_MR_FWD");
    if (force && (force[0] == 'v' || force[0] == 'q')) return false;
    if (!(force && force[0] == 'k') && K <= 16) return false;      // measured (profiles/r03_kernel_bench_cfg2.md): short lists keep the
                                                                   // quad kernel (K <= 8 windows, 61 % of HBM) / the dword kernel (K = 14:
                                                                   // 22.3 vs 23.8 us pooled, 32.2 vs 39.2 us windows); K = 28 / 32 take this one
    if (M > 65536) return false;
    const size_t per_quad = (size_t)M * 16;
    if (per_quad > 152 * 1024) return false;
    const int total_quads = (C + 3) / 4;
    size_t budget = M <= 512 ? 20 * 1024 : 48 * 1024;
    int quads = (int)(budget / per_quad);
    if (quads < 1) quads = 1;
    if (quads > kKqMaxQuads) quads = kKqMaxQuads;
    if (quads > total_quads) quads = total_quads;
    while (quads > 1 && (long long)cdiv(total_quads, quads) * B * cdiv(N, 128) < 1024) quads = (quads + 1) / 2;     // fill the chip
    quads = cdiv(total_quads, cdiv(total_quads, quads));      // balance the last block
    p->quads = quads;
    p->q_blocks = cdiv(total_quads, quads);
    p->lds = (size_t)quads * per_quad;
    long long want = cdiv64(1024, (long long)p->q_blocks * B);
    const long long max_tiles = N >= 128 ? N / 128 : 1;
    if (want > max_tiles) want = max_tiles;
    if (want < 1) want = 1;
    p->n_per_block = cdiv(cdiv(N, (int)want), 64) * 64;
    p->n_tiles = cdiv(N, p->n_per_block);
    p->threads = 256;
    return p->q_blocks <= 65535 && B <= 65535;
}

static int check_mr_args(const char* who, const void* a, const void* b, const void* c, int B, int C,
                         int N, int M, int K, int idx_stride, int idx_step) {
    NEXTOU_REQUIRE(a && b && c, "%s: null pointer", who);
    NEXTOU_REQUIRE(B > 0 && C > 0 && N > 0 && M > 0 && K > 0, "%s: non-positive size B=%d C=%d N=%d M=%d K=%d",
                   who, B, C, N, M, K);
    NEXTOU_REQUIRE(idx_step > 0 && idx_stride >= (K - 1) * idx_step + 1,
                   "%s: idx_stride=%d too small for K=%d step=%d", who, idx_stride, K, idx_step);
    NEXTOU_REQUIRE(B <= 65535 && C <= 65535, "%s: B=%d / C=%d exceed the grid limit 65535", who, B, C);
    return 0;
}

}  // namespace nextou

using namespace nextou;

extern "C" int nextou_mr_aggregate_fwd(const float* x, const float* y, const int32_t* nn_idx,
                                       const int32_t* center_idx, float* out, uint16_t* arg_out, int B,
                                       int C, int N, int M, int K, int idx_stride, int idx_step,
                                       nextou_stream_t stream) {
    if (int e = check_mr_args("mr_aggregate_fwd", x, nn_idx, out, B, C, N, M, K, idx_stride, idx_step)) return e;
    NEXTOU_REQUIRE(y != nullptr || M == N, "mr_aggregate_fwd: y == NULL needs M == N (N=%d M=%d)", N, M);
    NEXTOU_REQUIRE(arg_out == nullptr || (center_idx == nullptr && M <= 65536),
                   "mr_aggregate_fwd: arg_out needs identity centres and M <= 65536 (M=%d)", M);
    hipStream_t s = (hipStream_t)stream;
    const float* src = y ? y : x;
    const bool self = (y == nullptr);
    // algorithmic HBM bytes: read x (+y), read idx (int32), write the 2C-channel output (+ arg)
    const double fwd_bytes = 4.0 * B * C * ((double)N + (y ? M : 0)) + 4.0 * B * (double)N * K + 8.0 * B * C * (double)N +
                             (arg_out ? 2.0 * B * C * (double)N : 0.0);
    Q4Plan qp;
    if (center_idx == nullptr && K <= 32 && plan_q4(B, C, N, M, K, self, &qp)) {
        dim3 grid(qp.n_tiles, qp.q_blocks, B), block(qp.threads);
        const int kb = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
        ProfScope prof(s, kBoundHbm, fwd_bytes, "mr_fwd_q4_kernel<%d,%s,%s>[B%d C%d N%d M%d K%d]", kb,
                       self ? "self" : "xy", arg_out ? "arg" : "noarg", B, C, N, M, K);
#define NEXTOU_MR_Q4(KB, SELF, ARG)                                                                          \
    do {                                                                                                     \
        if (qp.lds > 64 * 1024)                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_fwd_q4_kernel<KB, SELF, ARG>),       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)qp.lds);              \
        hipLaunchKernelGGL((mr_fwd_q4_kernel<KB, SELF, ARG>), grid, block, qp.lds, s, x, src, nn_idx, out, arg_out, \
                           C, N, M, K, idx_stride, idx_step, qp.quads, qp.n_per_block);                      \
    } while (0)
#define NEXTOU_MR_Q4_KB(KB)                                                      \
    do {                                                                         \
        if (self && arg_out) NEXTOU_MR_Q4(KB, true, true);                       \
        else if (self) NEXTOU_MR_Q4(KB, true, false);                            \
        else if (arg_out) NEXTOU_MR_Q4(KB, false, true);                         \
        else NEXTOU_MR_Q4(KB, false, false);                                     \
    } while (0)
        if (kb == 8) NEXTOU_MR_Q4_KB(8);
        else if (kb == 16) NEXTOU_MR_Q4_KB(16);
        else NEXTOU_MR_Q4_KB(32);
#undef NEXTOU_MR_Q4_KB
#undef NEXTOU_MR_Q4
        return check_launch("mr_fwd_q4_kernel");
    }
    if (center_idx == nullptr && K <= 32 && plan_kq(B, C, N, M, K, self, &qp)) {
        dim3 grid(qp.n_tiles, qp.q_blocks, B), block(qp.threads);
        const int kl = K <= 8 ? 2 : (K <= 16 ? 4 : (K <= 28 ? 7 : 8));
        ProfScope prof(s, kBoundHbm, fwd_bytes, "mr_fwd_kq_kernel<%d,%s,%s>[B%d C%d N%d M%d K%d]", kl,
                       self ? "self" : "xy", arg_out ? "arg" : "noarg", B, C, N, M, K);
#define NEXTOU_MR_KQ(KL, SELF, ARG)                                                                          \
    do {                                                                                                     \
        if (qp.lds > 64 * 1024)                                                                              \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_fwd_kq_kernel<KL, SELF, ARG>),       \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)qp.lds);              \
        hipLaunchKernelGGL((mr_fwd_kq_kernel<KL, SELF, ARG>), grid, block, qp.lds, s, x, src, nn_idx, out, arg_out, \
                           C, N, M, K, idx_stride, idx_step, qp.quads, qp.n_per_block);                      \
    } while (0)
#define NEXTOU_MR_KQ_KL(KL)                                                      \
    do {                                                                         \
        if (self && arg_out) NEXTOU_MR_KQ(KL, true, true);                       \
        else if (self) NEXTOU_MR_KQ(KL, true, false);                            \
        else if (arg_out) NEXTOU_MR_KQ(KL, false, true);                         \
        else NEXTOU_MR_KQ(KL, false, false);                                     \
    } while (0)
        if (kl == 2) NEXTOU_MR_KQ_KL(2);
        else if (kl == 4) NEXTOU_MR_KQ_KL(4);
        else if (kl == 7) NEXTOU_MR_KQ_KL(7);
        else NEXTOU_MR_KQ_KL(8);
#undef NEXTOU_MR_KQ_KL
#undef NEXTOU_MR_KQ
        return check_launch("mr_fwd_kq_kernel");
    }
    MrPlan p;
    if (center_idx == nullptr && K <= 32 && plan_lds(B, C, N, M, M, true, &p)) {
        dim3 grid(p.n_tiles, p.c_chunks, B), block(p.threads);
        const int kb = K <= 8 ? 8 : (K <= 16 ? 16 : 32);
        ProfScope prof(s, kBoundHbm, fwd_bytes, "mr_fwd_lds_kernel<%d,%s,%s>[B%d C%d N%d M%d K%d]", kb,
                       self ? "self" : "xy", arg_out ? "arg" : "noarg", B, C, N, M, K);
#define NEXTOU_MR_FWD(KB, SELF, ARG)                                                                      \
    hipLaunchKernelGGL((mr_fwd_lds_kernel<KB, SELF, ARG>), grid, block, p.lds, s, x, src, nn_idx, out, arg_out, \
                       C, N, M, K, idx_stride, idx_step, p.chunk, p.n_per_block)
#define NEXTOU_MR_FWD_KB(KB)                                                     \
    do {                                                                         \
        if (self && arg_out) NEXTOU_MR_FWD(KB, true, true);                      \
        else if (self) NEXTOU_MR_FWD(KB, true, false);                           \
        else if (arg_out) NEXTOU_MR_FWD(KB, false, true);                        \
        else NEXTOU_MR_FWD(KB, false, false);                                    \
    } while (0)
        if (kb == 8) NEXTOU_MR_FWD_KB(8);
        else if (kb == 16) NEXTOU_MR_FWD_KB(16);
        else NEXTOU_MR_FWD_KB(32);
#undef NEXTOU_MR_FWD_KB
#undef NEXTOU_MR_FWD
        return check_launch("mr_fwd_lds_kernel");
    }
    if (arg_out != nullptr)
        return fail(NEXTOU_ENOTSUP, "mr_aggregate_fwd: arg_out is only produced by the LDS kernel (K <= 32, rows <= %d floats)",
                    kGatherLdsBytes / 4);
    hipLaunchKernelGGL(mr_fwd_global_kernel, dim3(cdiv(N, 256), C, B), dim3(256), 0, s, x, src, nn_idx,
                       center_idx, out, C, N, M, K, idx_stride, idx_step);
    return check_launch("mr_fwd_global_kernel");
}

// 1 if nextou_mr_aggregate_fwd can fill arg_out for this shape (lets the caller decide what to save)
extern "C" int nextou_mr_aggregate_has_arg(int B, int C, int N, int M, int K) {
    MrPlan p;
    Q4Plan q;
    return (M <= 65536 && K <= 32 && B > 0 && C > 0 && N > 0 && M > 0 &&
            (plan_q4(B, C, N, M, K, M == N, &q) || plan_kq(B, C, N, M, K, M == N, &q) || plan_lds(B, C, N, M, M, true, &p))) ? 1 : 0;
}

// 1 if the caller should keep nn_idx alive and take nextou_mr_aggregate_bwd_arg_idx for a self graph of this shape.
// Policy, not capability: measured on MI355X (profiles/r02_kernel_bench_k2.md) the reverse-list gather is SLOWER than the
// LDS-atomic scatter it was meant to replace (Swin s2 200 us vs 133 us; cfg-5 Swin s2 677 vs 282 us — every workgroup of
// a window rebuilds the lists, and lanes of one wave walk lists of different lengths), so it is opt-in
// (NEXTOU_MR_BWD=rev) for runs that want bit-reproducible gradients.
static bool rev_shape_ok(int B, int C, int N, int K) { return B > 0 && C > 0 && N > 0 && N <= 512 && K > 0 && K <= 64; }
extern "C" int nextou_mr_aggregate_bwd_wants_idx(int B, int C, int N, int K) {
    const char* e = getenv("NEXTOU_MR_BWD");
    return (e != nullptr && e[0] == 'r' && rev_shape_ok(B, C, N, K)) ? 1 : 0;
}

static int launch_bwd_rev(const float* gout, const uint16_t* arg, const int32_t* nn_idx, float* dx, int B, int C, int N, int K,
                          int idx_stride, int idx_step, hipStream_t s) {
    const int total_quads = (C + 3) / 4;
    const size_t fixed = (size_t)(2 * N + 1) * 4 + (size_t)N * K * 2 + 16;
    int quads = (int)((28 * 1024) / ((size_t)N * 24));       // ~28 KB of g_mr / arg tiles per workgroup
    if (quads < 1) quads = 1;
    if (quads > total_quads) quads = total_quads;
    quads = cdiv(total_quads, cdiv(total_quads, quads));
    size_t lds = (size_t)quads * N * 24 + fixed;
    lds = (lds + 15) & ~(size_t)15;
    const int threads = ((N < 512 ? N : 512) + 63) / 64 * 64;
    const double bytes = 8.0 * B * C * (double)N + 2.0 * B * C * (double)N + 4.0 * B * C * (double)N + 4.0 * B * (double)N * K;
    ProfScope prof(s, kBoundHbm, bytes, "mr_bwd_rev_kernel<self>[B%d C%d N%d K%d]", B, C, N, K);
    if (lds > 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&mr_bwd_rev_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipLaunchKernelGGL(mr_bwd_rev_kernel, dim3(cdiv(total_quads, quads), B), dim3(threads), lds, s, gout, arg, nn_idx, dx, C, N, K,
                       idx_stride, idx_step, quads);
    return check_launch("mr_bwd_rev_kernel");
}

extern "C" int nextou_mr_aggregate_bwd_arg_idx(const float* gout, const uint16_t* arg, const int32_t* nn_idx, float* dx, int B,
                                               int C, int N, int K, int idx_stride, int idx_step, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gout && arg && nn_idx && dx, "mr_aggregate_bwd_arg_idx: null pointer");
    NEXTOU_REQUIRE(B > 0 && C > 0 && N > 0 && K > 0 && B <= 65535, "mr_aggregate_bwd_arg_idx: bad size B=%d C=%d N=%d K=%d", B, C, N, K);
    NEXTOU_REQUIRE(idx_step > 0 && idx_stride >= (K - 1) * idx_step + 1, "mr_aggregate_bwd_arg_idx: idx_stride=%d too small for K=%d step=%d",
                   idx_stride, K, idx_step);
    if (!rev_shape_ok(B, C, N, K))
        return fail(NEXTOU_ENOTSUP, "mr_aggregate_bwd_arg_idx: self graphs of N <= 512 points only (N=%d)", N);
    return launch_bwd_rev(gout, arg, nn_idx, dx, B, C, N, K, idx_stride, idx_step, (hipStream_t)stream);
}

extern "C" int nextou_mr_aggregate_bwd_arg(const float* gout, const uint16_t* arg, float* dx, float* dy, int B,
                                           int C, int N, int M, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gout && arg && dx, "mr_aggregate_bwd_arg: null pointer");
    NEXTOU_REQUIRE(B > 0 && C > 0 && N > 0 && M > 0 && B <= 65535, "mr_aggregate_bwd_arg: bad size B=%d C=%d N=%d M=%d", B, C, N, M);
    NEXTOU_REQUIRE(dy != nullptr || M == N, "mr_aggregate_bwd_arg: dy == NULL (self graph) needs M == N");
    hipStream_t s = (hipStream_t)stream;
    const bool self = (dy == nullptr);
    const int budget = 32 * 1024 / (int)sizeof(float);  // <= 32 KB of accumulators per workgroup
    int chunk = budget / M;
    if (chunk < 1) return fail(NEXTOU_ENOTSUP, "mr_aggregate_bwd_arg: M=%d rows do not fit the LDS accumulators", M);
    if (chunk > C) chunk = C;
    while (chunk > 1 && (long long)cdiv(C, chunk) * B < 1024) chunk = (chunk + 1) / 2;
    chunk = cdiv(C, cdiv(C, chunk));
    while (chunk > 1 && (unsigned long long)chunk * N * N >= 0x100000000ull) --chunk;  // keeps e / N by umulhi exact
    const int threads = 256;
    const bool vec4 = (N % 4 == 0) && (((reinterpret_cast<uintptr_t>(gout) | reinterpret_cast<uintptr_t>(dx)) & 15u) == 0) &&
                      ((reinterpret_cast<uintptr_t>(arg) & 7u) == 0);
    const unsigned row_items = (unsigned)(vec4 ? N / 4 : N);
    const unsigned magic = (unsigned)((0x100000000ull + row_items - 1) / row_items);  // it / row_items by umulhi
    const double bytes = 8.0 * B * C * (double)N + 2.0 * B * C * (double)N + 4.0 * B * C * ((double)N + (self ? 0 : M));
    ProfScope prof(s, kBoundHbm, bytes, "mr_bwd_arg_kernel<%s>[B%d C%d N%d M%d]", self ? "self" : "xy", B, C, N, M);
    dim3 grid(cdiv(C, chunk), B);
    const size_t lds = (size_t)chunk * M * sizeof(float);
#define NEXTOU_MR_BWD_ARG(SELF, VEC)                                                                            \
    hipLaunchKernelGGL((mr_bwd_arg_kernel<SELF, VEC>), grid, dim3(threads), lds, s, gout, arg, dx, dy, C, N, M, chunk, \
                       magic)
    if (self && vec4) NEXTOU_MR_BWD_ARG(true, true);
    else if (self) NEXTOU_MR_BWD_ARG(true, false);
    else if (vec4) NEXTOU_MR_BWD_ARG(false, true);
    else NEXTOU_MR_BWD_ARG(false, false);
#undef NEXTOU_MR_BWD_ARG
    return check_launch("mr_bwd_arg_kernel");
}

extern "C" int nextou_mr_aggregate_bwd(const float* gout, const float* x, const float* y,
                                       const int32_t* nn_idx, const int32_t* center_idx, float* dx,
                                       float* dy, int B, int C, int N, int M, int K, int idx_stride,
                                       int idx_step, nextou_stream_t stream) {
    if (int e = check_mr_args("mr_aggregate_bwd", gout, nn_idx, dx, B, C, N, M, K, idx_stride, idx_step)) return e;
    NEXTOU_REQUIRE(x != nullptr, "mr_aggregate_bwd: x is null");
    NEXTOU_REQUIRE((y == nullptr) == (dy == nullptr), "mr_aggregate_bwd: dy must be given iff y is");
    NEXTOU_REQUIRE(y != nullptr || M == N, "mr_aggregate_bwd: y == NULL needs M == N (N=%d M=%d)", N, M);
    hipStream_t s = (hipStream_t)stream;
    const bool self = (y == nullptr);
    const float* src = self ? x : y;
    MrPlan p;
    // algorithmic HBM bytes: read gout (2C), x (+y), idx; write dx (+dy)
    const double bwd_bytes = 8.0 * B * C * (double)N + 4.0 * B * C * ((double)N + (y ? M : 0)) + 4.0 * B * (double)N * K +
                             4.0 * B * C * ((double)N + (y ? M : 0));
    if (center_idx == nullptr && K <= 32 && plan_lds(B, C, N, M, 2 * M, !self, &p)) {
        dim3 grid(p.n_tiles, p.c_chunks, B), block(p.threads);
        ProfScope prof(s, kBoundHbm, bwd_bytes, "mr_bwd_lds_kernel<%d,%s>[B%d C%d N%d M%d K%d]", K <= 8 ? 8 : (K <= 16 ? 16 : 32),
                       self ? "self" : "xy", B, C, N, M, K);
        if (!self) {
            hipError_t e = hipMemsetAsync(dy, 0, (size_t)B * C * M * sizeof(float), s);
            if (e != hipSuccess) return fail((int)e, "mr_aggregate_bwd: memset dy: %s", hipGetErrorString(e));
        }
#define NEXTOU_MR_BWD(KB)                                                                            \
    do {                                                                                             \
        if (self)                                                                                    \
            hipLaunchKernelGGL((mr_bwd_lds_kernel<KB, true>), grid, block, p.lds, s, gout, x, src,   \
                               nn_idx, dx, dx, C, N, M, K, idx_stride, idx_step, p.chunk,            \
                               p.n_per_block);                                                       \
        else                                                                                         \
            hipLaunchKernelGGL((mr_bwd_lds_kernel<KB, false>), grid, block, p.lds, s, gout, x, src,  \
                               nn_idx, dx, dy, C, N, M, K, idx_stride, idx_step, p.chunk,            \
                               p.n_per_block);                                                       \
    } while (0)
        if (K <= 8) NEXTOU_MR_BWD(8);
        else if (K <= 16) NEXTOU_MR_BWD(16);
        else NEXTOU_MR_BWD(32);
#undef NEXTOU_MR_BWD
        return check_launch("mr_bwd_lds_kernel");
    }
    // generic path: everything through global atomics
    hipError_t e = hipMemsetAsync(dx, 0, (size_t)B * C * N * sizeof(float), s);
    if (e != hipSuccess) return fail((int)e, "mr_aggregate_bwd: memset dx: %s", hipGetErrorString(e));
    if (!self) {
        e = hipMemsetAsync(dy, 0, (size_t)B * C * M * sizeof(float), s);
        if (e != hipSuccess) return fail((int)e, "mr_aggregate_bwd: memset dy: %s", hipGetErrorString(e));
    }
    hipLaunchKernelGGL(mr_bwd_global_kernel, dim3(cdiv(N, 256), C, B), dim3(256), 0, s, gout, x, src,
                       nn_idx, center_idx, dx, self ? dx : dy, C, N, M, K, idx_stride, idx_step);
    return check_launch("mr_bwd_global_kernel");
}

extern "C" int nextou_gather_fwd(const float* src, const int32_t* idx, float* out, int B, int C,
                                 int M, int N, int K, nextou_stream_t stream) {
    NEXTOU_REQUIRE(src && idx && out, "gather_fwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && C > 0 && M > 0 && N > 0 && K > 0 && B <= 65535 && C <= 65535,
                   "gather_fwd: bad size B=%d C=%d M=%d N=%d K=%d", B, C, M, N, K);
    const long long NK = (long long)N * K;
    hipLaunchKernelGGL(gather_fwd_kernel, dim3((unsigned)cdiv64(NK, 256), C, B), dim3(256), 0,
                       (hipStream_t)stream, src, idx, out, C, M, NK);
    return check_launch("gather_fwd_kernel");
}

extern "C" int nextou_gather_bwd(const float* gout, const int32_t* idx, float* dsrc, int B, int C,
                                 int M, int N, int K, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gout && idx && dsrc, "gather_bwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && C > 0 && M > 0 && N > 0 && K > 0 && B <= 65535 && C <= 65535,
                   "gather_bwd: bad size B=%d C=%d M=%d N=%d K=%d", B, C, M, N, K);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipMemsetAsync(dsrc, 0, (size_t)B * C * M * sizeof(float), s);
    if (e != hipSuccess) return fail((int)e, "gather_bwd: memset: %s", hipGetErrorString(e));
    const long long NK = (long long)N * K;
    hipLaunchKernelGGL(gather_bwd_kernel, dim3((unsigned)cdiv64(NK, 256), C, B), dim3(256), 0, s, gout,
                       idx, dsrc, C, M, NK);
    return check_launch("gather_bwd_kernel");
}
