// K1 — dense kNN graph for gfx950 (MI355X).
//
// Replaces the reference's F.normalize -> bmm -> add -> add -> (+relative_pos) -> neg -> topk
// chain (reference network_architecture/torch_edge.py:151-163, 58-110, 12-55) with
//   knn_prep_kernel   : L2-normalise over channels + squared norms of the normalised rows
//   knn_fused_kernel  : f32 MFMA (32x32x2) distance tiles, LDS-staged channel slabs, and a
//                       streaming per-query top-K kept entirely in registers; the (B,N,M)
//                       distance matrix never exists in HBM
//                       (128-wide chunks, K > 16: the slabs go global -> LDS directly, global_load_lds_dwordx4 into
//                       three buffers, two slabs ahead of the MFMAs — round 6, NEXTOU_KNN_GLDS=0 for the register-staged loop)
//   knn_window_kernel / knn_small_kernel : self graphs of <= 192 points in ONE launch, normalisation inside (rounds 4 / 6)
//   knn_dist_naive / knn_select_naive : the materialising fallback (any K), also the on-GPU
//                       cross-check of the "MFMA == fmaf chain" claim.
//
// Arithmetic contract (must stay bit-identical to oracle/nextou_oracle.c):
//   den = max(sqrtf(chain(x*x)), 1e-12f); xn = x / den; xs = chain(xn*xn);
//   inner = chain(xn*yn) with acc = fmaf(a_c, b_c, acc), c ascending, acc0 = 0;
//   dist = ((xs + (-2*inner)) + ys) [+ relpos];   order by (dist, index).
// The f32 MFMA is bitwise a k-ordered fmaf chain (MI355X_MICROARCH.md, "Matrix cores"), so the
// MFMA and VALU paths agree bit for bit.  This file is compiled with -ffp-contract=off and
// correctly rounded divide/sqrt.
#include "common.h"
#include <cmath>
#include <cstdlib>

// Experiment switch (tools/ablate_knn.sh builds side libraries with -DNEXTOU_ABLATE=n; the product
// build leaves it 0):  1 = skip the top-K pushes, 2 = stage only the first slab of every chunk,
// 4 = skip the MFMAs.  Results are wrong by construction with any bit set.
#ifndef NEXTOU_ABLATE
#define NEXTOU_ABLATE 0
#endif

namespace nextou {

constexpr float kNormEps = 1e-12f;  // F.normalize eps (torch_edge.py:154-155,160)
constexpr int kSentinelIdx = 0x7fffffff;

// A distance that is NaN or +inf (a feature overflowed: fp16 autocast does that, and GradScaler expects to survive it) becomes the largest
// finite float: it still sorts after every finite distance, ties among such candidates go by index, and — the point — it still ENTERS a list
// whose empty slots are (+inf, sentinel), so every output slot holds a candidate id.  Left as NaN it would never compare below anything, the
// row would keep its sentinels (or, in the counting kernel, leave slots unwritten) and the aggregation after it would gather out of bounds.
// The reference's topk puts NaN first instead; both are garbage in, valid ids out.  One v_min_f32 (IEEE minNum: the non-NaN operand);
// finite distances pass unchanged, so the bit-exactness tests are not affected.
__device__ __forceinline__ float finite_or_last(float dist) { return fminf(dist, 3.402823466e+38f); }

using f32x16 = __attribute__((ext_vector_type(16))) float;

// --------------------------------------------------------------------------------------------
// prep: one thread per point, channel loop strided by N (coalesced across the wave).
// --------------------------------------------------------------------------------------------
// NORMALIZE = false: inputs are used as they are (dense_knn_matrix / *_pairwise_distance called
// directly, torch_edge.py:58-110 do not normalise); only the squared norms are produced.
// U = loads in flight per lane (the fma chain itself stays strictly c-ordered whatever U): 16 when the grid fills the chip; 64 for the
// few-hundred-point graphs of stages 4 / 5 (B' x N <= 8 192: one or two waves per CU), where the kernel is nothing but 2 C / U dependent
// round trips to memory — 22-27 us at C = 324 with U = 16 (round 5)
template <bool NORMALIZE, int U>
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ x,
                                                       float* __restrict__ xn,
                                                       float* __restrict__ sq, int C, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float* xb = x + (size_t)b * C * N + n;
    float* xo = xn + (size_t)b * C * N + n;
    float s = 0.f;
    int c = 0;
    for (; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xb[(size_t)(c + u) * N];
#pragma unroll
        for (int u = 0; u < U; ++u) s = fmaf(v[u], v[u], s);
    }
    for (; c < C; ++c) {
        const float v = xb[(size_t)c * N];
        s = fmaf(v, v, s);
    }
    if (!NORMALIZE) {
        sq[(size_t)b * N + n] = s;
        return;
    }
    const float den = fmaxf(sqrtf(s), kNormEps);
    float q = 0.f;
    for (c = 0; c + U <= C; c += U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = xb[(size_t)(c + u) * N];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            v[u] = v[u] / den;
            xo[(size_t)(c + u) * N] = v[u];
            q = fmaf(v[u], v[u], q);
        }
    }
    for (; c < C; ++c) {
        const float v = xb[(size_t)c * N] / den;
        xo[(size_t)c * N] = v;
        q = fmaf(v, v, q);
    }
    sq[(size_t)b * N + n] = q;
}

// --------------------------------------------------------------------------------------------
// prep, tiled (round 3): the kernel above keeps 16 loads in flight per point and nothing else, so a 168-point stage
// (B' = 2: 336 threads on the whole chip) spends 2 x C / 16 dependent round trips to memory — 21-26 us for 0.4 MB, and the pooled
// graphs pay it twice (queries and candidates, two launches).  Here a workgroup owns 64 points: all 256 threads pull the
// (C, 64) slab into LDS with every load in flight, ONE wave runs the two strictly c-ordered fmaf chains out of LDS (the
// arithmetic contract is unchanged: chain(x*x) -> max(sqrt, eps) -> x / den -> chain(xn*xn)), the divides are spread over
// all four waves, and the normalised slab goes back coalesced.  Queries and candidates share one launch (tiles_x + tiles_y).
// LDS = C * 64 floats (83 KB at C = 324); C > 384 keeps the kernel above.
// --------------------------------------------------------------------------------------------
constexpr int kPrepPts = 64;
constexpr int kPrepMaxC = 384;

template <bool NORMALIZE>
__global__ __launch_bounds__(256) void knn_prep_tile_kernel(const float* __restrict__ x, float* __restrict__ xn,
                                                            float* __restrict__ xs, int N, int tiles_x,
                                                            const float* __restrict__ y, float* __restrict__ yn,
                                                            float* __restrict__ ys, int M, int C) {
    extern __shared__ __attribute__((aligned(16))) float tile[];      // [C][64]
    __shared__ float den_s[kPrepPts];
    int t = blockIdx.x;
    const float* src = x;
    float* dstn = xn;
    float* dsts = xs;
    int P = N;
    if (t >= tiles_x) { t -= tiles_x; src = y; dstn = yn; dsts = ys; P = M; }
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, cw = threadIdx.x >> 6;
    const int n = t * kPrepPts + lane;
    const bool ok = n < P;
    const float* sb = src + (size_t)b * C * P + (ok ? n : 0);
    constexpr int U = 16;
    for (int c0 = cw; c0 < C; c0 += 4 * U) {
        float v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 4 * u;
            v[u] = (ok && c < C) ? sb[(size_t)c * P] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int c = c0 + 4 * u;
            if (c < C) tile[c * kPrepPts + lane] = v[u];
        }
    }
    __syncthreads();
    if (cw == 0) {
        float s = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const float v = tile[c * kPrepPts + lane];
            s = fmaf(v, v, s);
        }
        if (!NORMALIZE) {
            if (ok) dsts[(size_t)b * P + n] = s;
        } else {
            den_s[lane] = fmaxf(sqrtf(s), kNormEps);
        }
    }
    if (!NORMALIZE) return;
    __syncthreads();
    const float den = den_s[lane];
    for (int c = cw; c < C; c += 4) tile[c * kPrepPts + lane] = tile[c * kPrepPts + lane] / den;
    __syncthreads();
    if (cw == 0) {
        float q = 0.f;
#pragma unroll 8
        for (int c = 0; c < C; ++c) {
            const float v = tile[c * kPrepPts + lane];
            q = fmaf(v, v, q);
        }
        if (ok) dsts[(size_t)b * P + n] = q;
    }
    if (ok) {
        float* ob = dstn + (size_t)b * C * P + n;
        for (int c = cw; c < C; c += 4) ob[(size_t)c * P] = tile[c * kPrepPts + lane];
    }
}

// --------------------------------------------------------------------------------------------
// fused distance + top-K.
//   grid  = (ceil(N / (32*nw)), B), block = 64*nw threads (nw waves, 32 queries per wave).
//   MFMA orientation: A operand = candidates (row i = m), B operand = queries (col j = n), so a
//   lane owns ONE query (n = lane&31 of its wave) and, per 32x32 tile, the 16 candidates
//   m = (r&3) + 8*(r>>2) + 4*(lane>>5).  The two half-waves hold disjoint candidate sets of the
//   same 32 queries; their sorted lists are merged with cross-half shuffles at the end.
//   Candidates reach a lane in ascending m, so a strict `<` on the distance alone keeps the
//   list in (dist, index) order.
// --------------------------------------------------------------------------------------------
template <int KB>
struct TopK {
    float d[KB];
    int i[KB];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KB; ++j) { d[j] = INFINITY; i[j] = kSentinelIdx; }
    }
    // Both inserts are "find the slot, shift the tail down": slot j takes the new entry iff the
    // entry sorts before old[j] but not before old[j-1].  The predicate is evaluated against the
    // NEW entry for every slot (it is monotone over a sorted list), never against a displaced old
    // entry — comparing displaced entries with `<` would reorder exact ties among the old ones.
    //
    // insert (v, vi); caller guarantees vi is larger than every index already present, so a strict
    // `<` on the distance alone is the (dist, index) order.
    // For a sorted list the shifted-in distance of slot j is the median of (v, d[j-1], d[j]):
    // one v_med3_f32 instead of two selects.
    __device__ __forceinline__ void push_ascending(float v, int vi) {
        // Compiled form per slot: v_cmp, v_med3, s_nop 0, v_cndmask, v_cndmask — the s_nop covers the VALU-writes-VCC ->
        // v_cndmask-reads-VCC hazard (1 issue slot in 5; 857 of the 7908 instructions of the <28,2> kernel).  Hoisting all
        // KB compares in front of the loop in the SOURCE changes nothing: the compiler sinks them back and emits
        // byte-identical ISA (checked, round 1).  Only per-slot SGPR masks (inline asm) would remove it, and 28 mask
        // pairs do not fit next to the kernel's 86 SGPRs.
        bool before_hi = v < d[KB - 1];
#pragma unroll
        for (int j = KB - 1; j >= 1; --j) {
            const bool before_lo = v < d[j - 1];
            d[j] = __builtin_amdgcn_fmed3f(v, d[j - 1], d[j]);
            i[j] = before_lo ? i[j - 1] : (before_hi ? vi : i[j]);
            before_hi = before_lo;
        }
        d[0] = fminf(v, d[0]);
        i[0] = before_hi ? vi : i[0];
    }
    // general insert with the full (dist, index) comparison.
    __device__ __forceinline__ bool sorts_before(float v, int vi, int j) const {
        return (v < d[j]) || (v == d[j] && vi < i[j]);
    }
    __device__ __forceinline__ void push_any(float v, int vi) {
        bool before_hi = sorts_before(v, vi, KB - 1);
#pragma unroll
        for (int j = KB - 1; j >= 1; --j) {
            const bool before_lo = sorts_before(v, vi, j - 1);
            d[j] = before_lo ? d[j - 1] : (before_hi ? v : d[j]);
            i[j] = before_lo ? i[j - 1] : (before_hi ? vi : i[j]);
            before_hi = before_lo;
        }
        d[0] = before_hi ? v : d[0];
        i[0] = before_hi ? vi : i[0];
    }
};

// --------------------------------------------------------------------------------------------
// Top-K as sorting networks on packed 64-bit keys (K >= 14).  key = orderable(dist) << 32 | index, so one unsigned
// 64-bit compare IS the (dist, index) order of the contract, exact ties included.  Per 32x32 tile a lane sorts its 16 new
// candidates with a bitonic network (80 compare-exchanges), folds them into its sorted list of KP = 16 / 32 keys
// (T[KP-1-i] = min(T[KP-1-i], N[i]) leaves a bitonic sequence holding the KP smallest of the union) and re-sorts it with a
// bitonic merge (KP/2 * log2 KP compare-exchanges): ~850 VALU instructions per tile at KP = 32, whatever the candidates
// are — the shift-insert list above costs 5 * K per candidate ROW as soon as one of the 64 lanes qualifies (always, in
// practice: 25-45 % of the candidates still beat a lane's running K-th), i.e. 2240 per tile at K = 28.
// --------------------------------------------------------------------------------------------
using u64 = unsigned long long;
constexpr u64 kKeyMax = ~0ull;

__device__ __forceinline__ u64 make_key(float d, int m) {
    unsigned u = __float_as_uint(d);
    u = (u == 0x80000000u) ? 0u : u;                                  // -0 == +0
    u ^= (unsigned)((int)u >> 31) | 0x80000000u;                      // monotone float -> unsigned
    return ((u64)u << 32) | (unsigned)m;
}
__device__ __forceinline__ float key_dist(u64 k) {
    unsigned u = (unsigned)(k >> 32);
    u ^= (u & 0x80000000u) ? 0x80000000u : 0xffffffffu;
    return __uint_as_float(u);
}
__device__ __forceinline__ void key_ce(u64& a, u64& b) {             // ascending compare-exchange
    const bool sw = b < a;
    const u64 lo = sw ? b : a, hi = sw ? a : b;
    a = lo;
    b = hi;
}
template <int N>
__device__ __forceinline__ void bitonic_sort_keys(u64 (&a)[N]) {
#pragma unroll
    for (int k = 2; k <= N; k <<= 1)
#pragma unroll
        for (int j = k >> 1; j > 0; j >>= 1)
#pragma unroll
            for (int i = 0; i < N; ++i) {
                const int l = i ^ j;
                if (l > i) {
                    if ((i & k) == 0) key_ce(a[i], a[l]);
                    else key_ce(a[l], a[i]);
                }
            }
}
template <int N>
__device__ __forceinline__ void bitonic_merge_keys(u64 (&a)[N]) {    // a bitonic -> ascending
#pragma unroll
    for (int j = N >> 1; j > 0; j >>= 1)
#pragma unroll
        for (int i = 0; i < N; ++i) {
            const int l = i ^ j;
            if (l > i) key_ce(a[i], a[l]);
        }
}
template <int KP>
struct KeyList {
    u64 k[KP];
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int j = 0; j < KP; ++j) k[j] = kKeyMax;
    }
    __device__ __forceinline__ void absorb16(u64 (&n)[16]) {          // n: any order; on return the list holds the KP best
        bitonic_sort_keys<16>(n);
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            const int j = KP - 1 - i;
            k[j] = (n[i] < k[j]) ? n[i] : k[j];
        }
        bitonic_merge_keys<KP>(k);
    }
};

// stage ROWS x WIDTH floats of a (rows, ld) matrix into LDS, zero-filling out-of-range rows/cols.
// `vec` (ld % 4 == 0, col0 % 4 == 0, 16-B aligned base): 16-B loads and ds_write_b128.
__device__ __forceinline__ void stage_slab(float* __restrict__ dst, const float* __restrict__ src, int ld,
                                           int row0, int rows_total, int col0, int cols_total, int width,
                                           bool vec, const int ROWS) {
    if (vec) {
        const int w4 = width >> 2;
        for (int e = threadIdx.x; e < ROWS * w4; e += blockDim.x) {
            const int r = e / w4, c4 = (e - r * w4) << 2;
            const int row = row0 + r, col = col0 + c4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (row < rows_total) {
                const float* p = src + (size_t)row * ld + col;
                if (col + 3 < cols_total) {
                    v = *reinterpret_cast<const float4*>(p);
                } else {
                    if (col < cols_total) v.x = p[0];
                    if (col + 1 < cols_total) v.y = p[1];
                    if (col + 2 < cols_total) v.z = p[2];
                }
            }
            *reinterpret_cast<float4*>(dst + r * width + c4) = v;
        }
    } else {
        for (int e = threadIdx.x; e < ROWS * width; e += blockDim.x) {
            const int r = e / width, c = e - r * width;
            const int row = row0 + r, col = col0 + c;
            dst[e] = (row < rows_total && col < cols_total) ? src[(size_t)row * ld + col] : 0.f;
        }
    }
}

// grid = (query tiles, B, S): split s handles candidates [s*m_per_split, (s+1)*m_per_split).
// S == 1: the final indices go to `out`; S > 1: every split writes its K best (dist, index) pairs
// to part_d / part_i [(b*N + n)*S + s][K] and knn_merge_kernel picks the overall K best.
__device__ float glds_zero[4] = {0.f, 0.f, 0.f, 0.f};           // what an out-of-range piece of a direct-to-LDS slab reads

template <int KB, int TILES, bool BITONIC, bool GLDS = false>
__global__ __launch_bounds__(512, (TILES == 4 ? 3 : 1)) void knn_fused_kernel(
    const float* __restrict__ xn, const float* __restrict__ yn,
    const float* __restrict__ xs, const float* __restrict__ ys,
    const float* __restrict__ relpos, int32_t* __restrict__ out,
    float* __restrict__ part_d, int32_t* __restrict__ part_i,
    int C, int N, int M, int K, int m_per_split, int vec_ok, int KS) {
    // KS = channels per LDS slab (even; 32 normally, 64 for latency-bound small grids)
    constexpr int TM = 32 * TILES;  // candidates per chunk
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int nw = blockDim.x >> 6;
    const int QW = nw * 32;
    float* ldsA = lds;            // [KS][TM]  candidates
    float* ldsB = lds + KS * TM;  // [KS][QW]  queries (aliases ldsA for a self window, see below)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int split = blockIdx.z, n_splits = gridDim.z;
    const int n0 = blockIdx.x * QW;
    const int n = n0 + wave * 32 + lq;
    const bool nvalid = n < N;
    const float* xb = xn + (size_t)b * C * N;
    const float* yb = yn + (size_t)b * C * M;
    const float* ysb = ys + (size_t)b * M;
    const float xsv = nvalid ? xs[(size_t)b * N + n] : 0.f;
    const float* rp_row = (relpos != nullptr && nvalid) ? relpos + (size_t)n * M : nullptr;
    const int m_begin = split * m_per_split;
    int m_end = m_begin + m_per_split;
    if (m_end > M) m_end = M;
    const bool vec = vec_ok != 0;

    constexpr int KP = KB <= 8 ? 8 : (KB <= 16 ? 16 : 32);      // padded list length of the network paths
    TopK<BITONIC ? 1 : KB> top;
    KeyList<BITONIC ? KP : 1> keys;
    if (BITONIC) keys.init();
    else top.init();

    // ---- slab staging plan -------------------------------------------------------------------
    // A self window that one workgroup covers completely (queries == candidates, one chunk) stages ONE
    // slab and reads both MFMA operands from it.  (A register-prefetch software pipeline of the staging
    // was measured and dropped: +50 VGPRs cost more occupancy than the hidden latency bought —
    // Pool s3 415 vs 326 us, Swin s2 195 vs 202 us, profiles/r01_knn_pipeline_ab.txt.  So was a threshold
    // pre-filter with per-lane compaction of the pushes: 460 vs 327 us on Pool s3 at 4 splits, break-even
    // at 2-3 splits where the lost parallelism costs more, profiles/r01_knn_compaction_ab.txt.)
    const bool share_ab = (yn == xn) && (QW == TM) && (n0 == 0) && (m_begin == 0) && (M <= TM);
    // epilogue: distances of a chunk -> running top-K (ascending m per lane)
    auto epilogue = [&](const int mc0, f32x16 (&acc)[TILES]) __attribute__((always_inline)) {
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            f32x16 v = acc[t];
            u64 fresh[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int mbase = mc0 + t * 32 + 8 * g + 4 * h;
                float yv[4], rv[4];
                if (vec && mbase + 3 < m_end) {  // 16-B aligned: M % 4 == 0 and mbase % 4 == 0
                    const float4 y4 = *reinterpret_cast<const float4*>(ysb + mbase);
                    yv[0] = y4.x; yv[1] = y4.y; yv[2] = y4.z; yv[3] = y4.w;
                    if (rp_row != nullptr) {
                        const float4 r4 = *reinterpret_cast<const float4*>(rp_row + mbase);
                        rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const bool ok = mbase + r < m_end;
                        yv[r] = ok ? ysb[mbase + r] : 0.f;
                        rv[r] = (ok && rp_row != nullptr) ? rp_row[mbase + r] : 0.f;
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mbase + r;
                    float dist = INFINITY;
                    const bool valid = nvalid && m < m_end;
                    if (valid) {
                        dist = (xsv + (-2.0f * v[4 * g + r])) + yv[r];
                        if (rp_row != nullptr) dist = dist + rv[r];
                        dist = finite_or_last(dist);
                    }
                    if (BITONIC) {
                        fresh[4 * g + r] = valid ? make_key(dist, m) : kKeyMax;
                        continue;
                    }
                    if (NEXTOU_ABLATE & 1) {
                        top.d[0] = fminf(top.d[0], dist);  // keeps the distance alive
                        continue;
                    }
                    if (__any(dist < top.d[(BITONIC ? 1 : KB) - 1])) top.push_ascending(dist, m);
                }
            }
            if (BITONIC) keys.absorb16(fresh);
        }
    };
    if (share_ab) ldsB = ldsA;
    if constexpr (GLDS) {
        // Round 6: 16-channel slabs in THREE LDS buffers, filled by global_load_lds_dwordx4 (global -> LDS without registers: the register-prefetch
        // pipeline of round 1 cost 50 VGPRs and an occupancy step).  (chunk, slab) pairs are ONE sequence: the loads of pair it + 2 are issued
        // right behind the barrier that hands pair it to the MFMAs — into the buffer pair it - 1 has just left — and have two pairs' MFMAs (and,
        // at a chunk's end, its whole selection epilogue) to arrive; one barrier per slab, in front of it a counted s_waitcnt: every wave issues
        // the same four loads per pair, vmcnt returns in order, so "all but four" means pair it has landed while pair it + 1 stays in flight.
        // An LDS row is 128 floats, a wave instruction writes lane x 16 B = two consecutive rows; pieces outside the matrix read a zero line
        // instead (the destination is lane-linear, so the zero fill has to come through the source address).  Same MFMA order per accumulator:
        // bit-identical distances.  Pool s3 287 -> 258 us with two buffers (one pair ahead), same box.
        constexpr int KSG = 16;
        static_assert(TM == 128 && !BITONIC, "two LDS rows per wave instruction");
        const int spc = (C + KSG - 1) / KSG;                          // slabs per chunk
        const int n_it = spc * ((m_end - m_begin + TM - 1) / TM);
        constexpr int NBUF = 3;                                        // LDS buffers: the loads run TWO slabs ahead of the MFMAs
        auto buffer = [&](int i) __attribute__((always_inline)) { return lds + i * (KSG * (TM + QW)); };
        const int wave_u = __builtin_amdgcn_readfirstlane(wave);        // (scalar loop counters for the issue loops)
        static_assert((KSG * TM / 256) % 4 == 0, "every wave issues the same number of loads per slab");
        constexpr int kLoadsPerSlab = 2 * (KSG * TM / 256) / 4;         // per wave (nw == 4, QW == TM): what s_waitcnt leaves in flight
        auto issue = [&](int ch, int sl, int bi) __attribute__((always_inline)) {
            float* dA = buffer(bi);
            float* dB = dA + KSG * TM;
            const int c0 = sl * KSG, mc = m_begin + ch * TM;
            const int rows = min(KSG, C - c0);
            for (int j = wave_u; j < KSG * TM / 256; j += 4) {          // candidates: wave instruction j = LDS rows 2j, 2j + 1
                const int r = 2 * j + (lane >> 5), c = (lane & 31) << 2;
                const float* src = (r < rows && mc + c < m_end) ? yb + (size_t)(c0 + r) * M + mc + c : glds_zero;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(dA + j * 256), 16, 0, 0);
            }
            for (int j = wave_u; j < KSG * QW / 256; j += 4) {          // queries (QW == 128 as well)
                const int r = 2 * j + (lane >> 5), c = (lane & 31) << 2;
                const float* src = (r < rows && n0 + c < N) ? xb + (size_t)(c0 + r) * N + n0 + c : glds_zero;
                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(dB + j * 256), 16, 0, 0);
            }
        };
        f32x16 acc[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // (chunk, slab) of the pair two ahead of the one being multiplied
        int ch = 0, sl = 0, pch = 0, psl = 0;
        auto advance = [&](int& c_, int& s_) __attribute__((always_inline)) { if (++s_ == spc) { s_ = 0; ++c_; } };
        if (n_it > 0) { issue(pch, psl, 0); advance(pch, psl); }
        if (n_it > 1) { issue(pch, psl, 1); advance(pch, psl); }
        int bi = 0;                                                     // buffer of pair `it`
        for (int it = 0; it < n_it; ++it) {
            // this wave's share of pair `it` is in LDS: every load but the kLoadsPerSlab of pair it + 1 has returned (vmcnt counts in order)
            if (it + 1 < n_it) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(kLoadsPerSlab) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                               // ... everybody's; and everybody is done with pair it - 1's buffer
            asm volatile("" ::: "memory");
            if (it + 2 < n_it) { issue(pch, psl, bi == 0 ? 2 : bi - 1); advance(pch, psl); }
            const float* A = buffer(bi);
            const float* Bq = A + KSG * TM;
            int kmax = min(KSG, C - sl * KSG);
            kmax = (kmax + 1) & ~1;
            for (int kp = 0; kp < kmax; kp += 2) {
                const float bq = Bq[(kp + h) * QW + wave * 32 + lq];
#pragma unroll
                for (int t = 0; t < TILES; ++t) {
                    const float a = A[(kp + h) * TM + t * 32 + lq];
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[t], 0, 0, 0);
                }
            }
            if (sl + 1 == spc) {
                epilogue(m_begin + ch * TM, acc);
#pragma unroll
                for (int t = 0; t < TILES; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
            }
            advance(ch, sl);
            bi = bi == NBUF - 1 ? 0 : bi + 1;
        }
    } else {
        for (int mc0 = m_begin; mc0 < m_end; mc0 += TM) {
            f32x16 acc[TILES];
    #pragma unroll
            for (int t = 0; t < TILES; ++t)
    #pragma unroll
                for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

            for (int c0 = 0; c0 < C; c0 += KS) {
                __syncthreads();  // previous slab fully consumed
                if (!(NEXTOU_ABLATE & 2) || c0 == 0) {
                    stage_slab(ldsA, yb, M, c0, C, mc0, m_end, TM, vec, KS);
                    if (!share_ab) stage_slab(ldsB, xb, N, c0, C, n0, N, QW, vec, KS);
                }
                __syncthreads();
                int kmax = C - c0;
                if (kmax > KS) kmax = KS;
                kmax = (kmax + 1) & ~1;
                for (int kp = 0; kp < kmax; kp += 2) {
                    const float bq = ldsB[(kp + h) * QW + wave * 32 + lq];
    #pragma unroll
                    for (int t = 0; t < TILES; ++t) {
                        const float a = ldsA[(kp + h) * TM + t * 32 + lq];
                        if (NEXTOU_ABLATE & 4) {
                            acc[t][0] += a * bq;  // keeps the LDS reads alive
                            continue;
                        }
                        acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[t], 0, 0, 0);
                    }
                }
            }

            epilogue(mc0, acc);
        }
    }

    // Merge the two half-waves' sorted lists (disjoint candidate sets of the same query) with a
    // bitonic network held in registers: t[i] = min(mine[i], partner[KP-1-i]) is a bitonic sequence
    // holding the KP smallest of the union; log2(KP) half-cleaner stages sort it.  ~KP/2*log2(KP)
    // compare-exchanges instead of K rounds of K-slot inserts (which cost 40 % on top of the pushes
    // when a workgroup only sees a few hundred candidates).  Comparisons are on (dist, index).
    if constexpr (BITONIC) {
        u64 t[KP];
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const u64 mine = keys.k[j], theirs = keys.k[KP - 1 - j];
            const unsigned lo = __shfl_xor((unsigned)(theirs & 0xffffffffull), 32);
            const unsigned hi = __shfl_xor((unsigned)(theirs >> 32), 32);
            const u64 other = ((u64)hi << 32) | lo;
            t[j] = other < mine ? other : mine;
        }
        bitonic_merge_keys<KP>(t);
        if (nvalid && h == 0) {
            if (n_splits == 1) {
                int32_t* o = out + ((size_t)b * N + n) * K;
#pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (j < K) o[j] = (int32_t)(unsigned)(t[j] & 0xffffffffull);
            } else {
                const size_t base = (((size_t)b * N + n) * n_splits + split) * K;
#pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (j < K) {
                        const bool none = t[j] == kKeyMax;
                        part_d[base + j] = none ? INFINITY : key_dist(t[j]);
                        part_i[base + j] = none ? kSentinelIdx : (int32_t)(unsigned)(t[j] & 0xffffffffull);
                    }
            }
        }
    } else {
        float md[KP];
        int mi[KP];
    #pragma unroll
        for (int j = 0; j < KP; ++j) {
            const int pj = KP - 1 - j;
            const float ad = (j < KB) ? top.d[j < KB ? j : 0] : INFINITY;
            const int ai = (j < KB) ? top.i[j < KB ? j : 0] : kSentinelIdx;
            float bd = INFINITY;
            int bi = kSentinelIdx;
            if (pj < KB) {
                bd = __shfl_xor(top.d[pj < KB ? pj : 0], 32);
                bi = __shfl_xor(top.i[pj < KB ? pj : 0], 32);
            }
            const bool take_b = (bd < ad) || (bd == ad && bi < ai);
            md[j] = take_b ? bd : ad;
            mi[j] = take_b ? bi : ai;
        }
    #pragma unroll
        for (int stride = KP / 2; stride >= 1; stride >>= 1) {
    #pragma unroll
            for (int j = 0; j < KP; ++j) {
                if ((j & stride) == 0) {
                    const int q = j + stride;
                    const bool sw = (md[q] < md[j]) || (md[q] == md[j] && mi[q] < mi[j]);
                    const float lo = sw ? md[q] : md[j], hi = sw ? md[j] : md[q];
                    const int li = sw ? mi[q] : mi[j], hi_i = sw ? mi[j] : mi[q];
                    md[j] = lo; md[q] = hi; mi[j] = li; mi[q] = hi_i;
                }
            }
        }
        if (nvalid && h == 0) {
            if (n_splits == 1) {
                int32_t* o = out + ((size_t)b * N + n) * K;
    #pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (j < K) o[j] = mi[j];
            } else {
                const size_t base = (((size_t)b * N + n) * n_splits + split) * K;
    #pragma unroll
                for (int j = 0; j < KB; ++j)
                    if (j < K) { part_d[base + j] = md[j]; part_i[base + j] = mi[j]; }
            }
        }
    }
}

// --------------------------------------------------------------------------------------------
// Self graphs of at most 192 points (the windows of every Swin-GNN block and the stage-5 pooled graph: N = M = 168 at cfg 2) in ONE
// launch, normalisation included (round 4).  Through the general path such a graph costs three launches — knn_prep (21-28 us: a chain of
// 2 x C / 16 dependent round trips per point), knn_fused (60-90 us) and, when the candidates are split, knn_merge (7 us) — i.e. 90-120 us for
// <= 2.3 GFLOP (profiles/r04_k1_small_graphs.md).  Here a workgroup owns a window (and one 32 * TILES-wide candidate range of it):
//   pass A  the (C, 192) raw slab streams through LDS in KS-channel slabs, all loads of a slab in flight; lane n runs the c-ordered chain
//           s = fmaf(x, x, s) out of LDS -> den[n] = max(sqrt(s), 1e-12)                     (the arithmetic of knn_prep_kernel, same order)
//   pass B  the slabs stream again (L2-hot); every element is divided by its point's den on the way into LDS (IEEE division: xn = x / den),
//           lane n continues the chain q = fmaf(xn, xn, q), and the MFMAs take BOTH operands from the one normalised slab
//   then    the epilogue, the per-lane sorted lists and the half-wave merge of knn_fused_kernel, with the squared norms read from LDS.
// Bit-identical to the general path by construction (same fmaf chains, same division, same MFMA k order); tests/test_gpu_parity.py holds
// both to the oracle.  grid = (splits, B'), block = 64 * ceil(N / 32) threads, LDS = KS * 192 floats + 2 * 192 floats.
// --------------------------------------------------------------------------------------------
constexpr int kWinPts = 192;

// One KS x 192 slab of the window kernel: EIGHT 16-byte loads in flight per thread before any is consumed (with one load per loop iteration
// the staging of a slab was a chain of eight dependent round trips — the run time of a 2-window launch), optionally divided by the points'
// norms on the way into LDS.  vec path only (N % 4 == 0, 16-byte aligned rows); pieces past C or N are zero.
template <bool DIVIDE>
__device__ __forceinline__ void win_stage(float* __restrict__ slab, const float* __restrict__ xb, const float* __restrict__ den_s, int c0,
                                          int C, int N, int KS) {
    constexpr int W = 192, U = 8;
    const int w4 = W / 4, total = KS * w4;
    for (int e0 = threadIdx.x; e0 < total; e0 += blockDim.x * U) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * blockDim.x;
            const int r = e / w4, c4 = (e - r * w4) << 2;
            const bool ok = e < total && c0 + r < C && c4 < N;
            v[u] = *reinterpret_cast<const float4*>(xb + (size_t)(ok ? c0 + r : 0) * N + (ok ? c4 : 0));       // clamped, unconditional
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int e = e0 + u * blockDim.x;
            if (e < total) {
                const int r = e / w4, c4 = (e - r * w4) << 2;
                const bool ok = c0 + r < C && c4 < N;
                float4 t = v[u];
                if (DIVIDE) {
                    const float4 d = *reinterpret_cast<const float4*>(den_s + (ok ? c4 : 0));
                    t = make_float4(t.x / d.x, t.y / d.y, t.z / d.z, t.w / d.w);
                }
                *reinterpret_cast<float4*>(slab + r * W + c4) = make_float4(ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f);
            }
        }
    }
}



// G = 2 (whole windows of more than 64 points): TWO groups of waves share a window — waves 0 .. nq-1 take the lower half of the candidate
// tiles, waves nq .. 2nq-1 the upper half, for the same 32-query tiles — and the two sorted lists of a query meet through LDS.  With one
// group a 168-point window is 6 waves of 6 accumulator tiles: 159 VGPRs, waves placed 2 / 2 / 1 / 1 on the CU's SIMDs, and (measured) ONE
// resident workgroup per CU whatever the occupancy calculator says; two groups are 12 waves of 3 tiles, 3 per SIMD.
template <int KB, int TILES, int G>
__global__ __launch_bounds__(G == 2 ? 768 : 384) void knn_window_kernel(const float* __restrict__ x, const float* __restrict__ relpos,
                                                         int32_t* __restrict__ out, float* __restrict__ part_d,
                                                         int32_t* __restrict__ part_i, int C, int N, int K, int KS, int ablate) {
    // ablate (experiments, NEXTOU_KNN_WIN_ABLATE; wrong results with any bit set): 1 skip pass A, 2 skip the MFMAs, 4 skip the list pushes,
    // 8 skip the chains of squares
    constexpr int TM = 32 * TILES * G;     // candidates per workgroup
    constexpr int W = kWinPts;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* slab = lds;                 // [KS][W]  (G = 2: afterwards the upper group's lists, [W][KP] (dist, id) pairs)
    float* den_s = lds + max(KS * W, G == 2 ? 2 * W * (KB <= 8 ? 8 : (KB <= 16 ? 16 : 32)) : 0);     // [W]
    float* sq_s = den_s + W;           // [W]  chain of xn^2
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    const int nq = (int)(blockDim.x >> 6) / G;              // query tiles = waves per group
    const int wq = G == 2 ? (wave >= nq ? wave - nq : wave) : wave, gq = G == 2 ? (wave >= nq ? 1 : 0) : 0;
    const int b = blockIdx.y, split = blockIdx.x, n_splits = gridDim.x;
    const float* xb = x + (size_t)b * C * N;
    const int n = wq * 32 + lq;
    const bool nvalid = n < N;
    const int m_begin = split * TM;
    const int m_end = min(m_begin + TM, N);
    const int mw_begin = m_begin + gq * 32 * TILES;          // this wave's candidates
    const bool vec = (N % 4 == 0) && ((reinterpret_cast<uintptr_t>(x) & 15u) == 0);
    // stagger (bits 8+ of `ablate`, microseconds): the workgroups that take the SECOND slot of a CU (ids 256 ... 511 of the first wave of
    // dispatches) start late, so that co-resident workgroups run different phases instead of the same ones in lockstep
    if (const int stagger_us = ablate >> 8) {
        const unsigned id = blockIdx.y * gridDim.x + blockIdx.x;
        if ((id >> 8) & 1u) {
            const long long t0 = wall_clock64();
            while (wall_clock64() - t0 < 100ll * stagger_us) __builtin_amdgcn_s_sleep(16);
        }
    }
    ablate &= 255;

    // ---- pass A: den
    float ssum = 0.f;
    for (int c0 = 0; c0 < ((ablate & 1) ? 0 : C); c0 += KS) {
        __syncthreads();
        if (vec) win_stage<false>(slab, xb, den_s, c0, C, N, KS);
        else stage_slab(slab, xb, N, c0, C, 0, N, W, false, KS);
        __syncthreads();
        if (tid < W) {
            int kmax = C - c0;
            if (kmax > KS) kmax = KS;
#pragma unroll 8
            for (int k = 0; k < kmax; ++k) { const float v = slab[k * W + tid]; ssum = fmaf(v, v, ssum); }
        }
    }
    if (tid < W) den_s[tid] = fmaxf(sqrtf(ssum), kNormEps);
    __syncthreads();

    // ---- pass B: normalise while staging, chain of squares, distance tiles
    f32x16 acc[TILES];
#pragma unroll
    for (int t = 0; t < TILES; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float qsum = 0.f;
    for (int c0 = 0; c0 < C; c0 += KS) {
        __syncthreads();               // the previous slab is fully consumed
        if (vec) {
            win_stage<true>(slab, xb, den_s, c0, C, N, KS);
        } else {
            for (int e = tid; e < KS * W; e += blockDim.x) {
                const int r = e / W, c = e - r * W;
                const int row = c0 + r;
                slab[e] = (row < C && c < N) ? xb[(size_t)row * N + c] / den_s[c] : 0.f;
            }
        }
        __syncthreads();
        int kmax = C - c0;
        if (kmax > KS) kmax = KS;
        if (tid < W && !(ablate & 8)) {
#pragma unroll 8
            for (int k = 0; k < kmax; ++k) { const float v = slab[k * W + tid]; qsum = fmaf(v, v, qsum); }
        }
        kmax = (kmax + 1) & ~1;        // (an odd tail multiplies a zero row: the slab is zero-filled past C)
        if (ablate & 2) kmax = 0;
        for (int kp = 0; kp < kmax; kp += 2) {
            const float bq = slab[(kp + h) * W + wq * 32 + lq];
#pragma unroll
            for (int t = 0; t < TILES; ++t) {
                const float a = slab[(kp + h) * W + mw_begin + t * 32 + lq];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[t], 0, 0, 0);
            }
        }
    }
    if (tid < W) sq_s[tid] = qsum;
    __syncthreads();

    // ---- epilogue: distances -> per-lane sorted list (candidates reach a lane in ascending index)
    TopK<KB> top;
    top.init();
    const float xsv = nvalid ? sq_s[n] : 0.f;
    const float* rp_row = (relpos != nullptr && nvalid) ? relpos + (size_t)n * N : nullptr;
    const bool rvec = vec && ((reinterpret_cast<uintptr_t>(relpos) & 15u) == 0);
#pragma unroll
    for (int t = 0; t < TILES; ++t) {
        f32x16 v = acc[t];
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int mbase = mw_begin + t * 32 + 8 * g + 4 * h;
            float yv[4], rv[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) yv[r] = mbase + r < W ? sq_s[mbase + r] : 0.f;
            if (rp_row != nullptr && rvec && mbase + 3 < m_end) {
                const float4 r4 = *reinterpret_cast<const float4*>(rp_row + mbase);
                rv[0] = r4.x; rv[1] = r4.y; rv[2] = r4.z; rv[3] = r4.w;
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) rv[r] = (rp_row != nullptr && mbase + r < m_end) ? rp_row[mbase + r] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int m = mbase + r;
                float dist = INFINITY;
                if (nvalid && m < m_end) {
                    dist = (xsv + (-2.0f * v[4 * g + r])) + yv[r];
                    if (rp_row != nullptr) dist = dist + rv[r];
                    dist = finite_or_last(dist);
                }
                if (__any(dist < top.d[KB - 1]) && !(ablate & 4)) top.push_ascending(dist, m);
            }
        }
    }

    // ---- merge the two half-waves' lists (bitonic, (dist, index) order) and emit — as knn_fused_kernel
    constexpr int KP = KB <= 8 ? 8 : (KB <= 16 ? 16 : 32);
    float md[KP];
    int mi[KP];
#pragma unroll
    for (int j = 0; j < KP; ++j) {
        const int pj = KP - 1 - j;
        const float ad = (j < KB) ? top.d[j < KB ? j : 0] : INFINITY;
        const int ai = (j < KB) ? top.i[j < KB ? j : 0] : kSentinelIdx;
        float bd = INFINITY;
        int bi = kSentinelIdx;
        if (pj < KB) {
            bd = __shfl_xor(top.d[pj < KB ? pj : 0], 32);
            bi = __shfl_xor(top.i[pj < KB ? pj : 0], 32);
        }
        const bool take_b = (bd < ad) || (bd == ad && bi < ai);
        md[j] = take_b ? bd : ad;
        mi[j] = take_b ? bi : ai;
    }
#pragma unroll
    for (int stride = KP / 2; stride >= 1; stride >>= 1) {
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            if ((j & stride) == 0) {
                const int q = j + stride;
                const bool sw = (md[q] < md[j]) || (md[q] == md[j] && mi[q] < mi[j]);
                const float lo = sw ? md[q] : md[j], hi = sw ? md[j] : md[q];
                const int li = sw ? mi[q] : mi[j], hi_i = sw ? mi[j] : mi[q];
                md[j] = lo; md[q] = hi; mi[j] = li; mi[q] = hi_i;
            }
        }
    }
    if (G == 2) {
        // the upper group's list of every query -> LDS (over the slab: the last MFMA read it before the barrier above); the lower group
        // takes the KP best of the two sorted lists with the same bitonic step (all of the upper group's ids are larger)
        float* xd = slab;
        int* xi = reinterpret_cast<int*>(slab + W * KP);
        if (gq == 1 && h == 0) {
#pragma unroll
            for (int j = 0; j < KP; ++j) { xd[n * KP + j] = md[j]; xi[n * KP + j] = mi[j]; }
        }
        __syncthreads();
        if (gq == 1) return;
#pragma unroll
        for (int j = 0; j < KP; ++j) {
            const float bd = xd[n * KP + (KP - 1 - j)];
            const int bi = xi[n * KP + (KP - 1 - j)];
            const bool take_b = (bd < md[j]) || (bd == md[j] && bi < mi[j]);
            md[j] = take_b ? bd : md[j];
            mi[j] = take_b ? bi : mi[j];
        }
#pragma unroll
        for (int stride = KP / 2; stride >= 1; stride >>= 1) {
#pragma unroll
            for (int j = 0; j < KP; ++j) {
                if ((j & stride) == 0) {
                    const int q = j + stride;
                    const bool sw = (md[q] < md[j]) || (md[q] == md[j] && mi[q] < mi[j]);
                    const float lo = sw ? md[q] : md[j], hi = sw ? md[j] : md[q];
                    const int li = sw ? mi[q] : mi[j], hi_i = sw ? mi[j] : mi[q];
                    md[j] = lo; md[q] = hi; mi[j] = li; mi[q] = hi_i;
                }
            }
        }
    }
    if (nvalid && h == 0) {
        if (n_splits == 1) {
            int32_t* o = out + ((size_t)b * N + n) * K;
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < K) o[j] = mi[j];
        } else {
            const size_t base = (((size_t)b * N + n) * n_splits + split) * K;
#pragma unroll
            for (int j = 0; j < KB; ++j)
                if (j < K) { part_d[base + j] = md[j]; part_i[base + j] = mi[j]; }
        }
    }
}

// S sorted partial lists per query -> the K best overall, by (dist, index).  One thread per partial
// entry: its final position is its own position plus the number of entries of the OTHER lists that
// sort before it (binary search; every candidate lives in exactly one list, so ranks are unique).
constexpr int kMaxSplits = 16;
__global__ __launch_bounds__(256) void knn_merge_kernel(const float* __restrict__ part_d,
                                                        const int32_t* __restrict__ part_i,
                                                        int32_t* __restrict__ out, long long rows, int S,
                                                        int K) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= rows * S * K) return;
    const long long row = e / ((long long)S * K);
    const int rem = (int)(e - row * S * K);
    const int sp = rem / K, j = rem - sp * K;
    const float* pd = part_d + (size_t)row * S * K;
    const int32_t* pi = part_i + (size_t)row * S * K;
    const float d = pd[sp * K + j];
    const int i = pi[sp * K + j];
    if (i == kSentinelIdx) return;  // list shorter than K: not a candidate
    int rank = j;
    for (int o = 0; o < S && rank < K; ++o) {
        if (o == sp) continue;
        int lo = 0, hi = K;  // first position in list o that does NOT sort before (d, i)
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            const float dm = pd[o * K + mid];
            const int im = pi[o * K + mid];
            if (dm < d || (dm == d && im < i)) lo = mid + 1; else hi = mid;
        }
        rank += lo;
    }
    if (rank < K) out[(size_t)row * K + rank] = i;
}

// Same merge with a workgroup's rows staged in LDS first: the binary searches above are chains of dependent GLOBAL loads
// (S lists x log2 K probes: 64 us for the 2 688 rows of the stage-4 pooled graph at S = 11, K = 32, i.e. pure latency).
// grid = ceil(rows / rows_per_wg); LDS = rows_per_wg * S * K * 8 bytes.
__global__ __launch_bounds__(256) void knn_merge_lds_kernel(const float* __restrict__ part_d,
                                                            const int32_t* __restrict__ part_i,
                                                            int32_t* __restrict__ out, long long rows, int S, int K,
                                                            int rows_per_wg) {
    extern __shared__ __attribute__((aligned(16))) float merge_lds[];
    const int per_row = S * K;
    float* sd = merge_lds;
    int32_t* si = reinterpret_cast<int32_t*>(merge_lds + (size_t)rows_per_wg * per_row);
    const long long row0 = (long long)blockIdx.x * rows_per_wg;
    long long left = rows - row0;
    const int nrows = left < rows_per_wg ? (int)left : rows_per_wg;
    const int total = nrows * per_row;
    const size_t base = (size_t)row0 * per_row;
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        sd[e] = part_d[base + e];
        si[e] = part_i[base + e];
    }
    __syncthreads();
    for (int e = threadIdx.x; e < total; e += blockDim.x) {
        const int r = e / per_row;
        const int rem = e - r * per_row;
        const int sp = rem / K, j = rem - sp * K;
        const float d = sd[e];
        const int i = si[e];
        if (i == kSentinelIdx) continue;  // list shorter than K: not a candidate
        const float* pd = sd + r * per_row;
        const int32_t* pi = si + r * per_row;
        int rank = j;
        for (int o = 0; o < S && rank < K; ++o) {
            if (o == sp) continue;
            int lo = 0, hi = K;  // first position in list o that does NOT sort before (d, i)
            while (lo < hi) {
                const int mid = (lo + hi) >> 1;
                const float dm = pd[o * K + mid];
                const int im = pi[o * K + mid];
                if (dm < d || (dm == d && im < i)) lo = mid + 1; else hi = mid;
            }
            rank += lo;
        }
        if (rank < K) out[(size_t)(row0 + r) * K + rank] = i;
    }
}

// --------------------------------------------------------------------------------------------
// naive pair: materialised distances + one wave per row (any K <= M).
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void knn_dist_naive_kernel(
    const float* __restrict__ xn, const float* __restrict__ yn, const float* __restrict__ xs,
    const float* __restrict__ ys, const float* __restrict__ relpos, float* __restrict__ dist,
    int C, int N, int M, int row_start, int rows) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    const int n = row_start + blockIdx.y;
    const int b = blockIdx.z;
    if (m >= M) return;
    const float* xb = xn + (size_t)b * C * N + n;
    const float* yb = yn + (size_t)b * C * M + m;
    float inner = 0.f;
    for (int c = 0; c < C; ++c) inner = fmaf(yb[(size_t)c * M], xb[(size_t)c * N], inner);
    float d = (xs[(size_t)b * N + n] + (-2.0f * inner)) + ys[(size_t)b * M + m];
    if (relpos != nullptr) d = d + relpos[(size_t)n * M + m];
    dist[((size_t)b * rows + (n - row_start)) * M + m] = d;
}

__device__ __forceinline__ unsigned long long knn_key(float d, int m) {
    unsigned int u = __float_as_uint(d);
    if (u == 0x80000000u) u = 0u;  // -0 == +0
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);
    return ((unsigned long long)u << 32) | (unsigned int)m;
}

// Round 5: the merge as a TOURNAMENT of list heads.  Both kernels above rank every partial entry by binary searches through the other
// S - 1 lists — (S - 1) log2 K dependent LDS / global probes per entry: 63 us for the 2 688 rows of cfg 2's stage-4 pooled graph
// (S = 11, K = 32: a 7.5 MB problem), 40 us at Pool s3 — pure latency.  Here 16 lanes own a row, lane l walks list l (the lists
// are sorted, so the K best overall are K pops of the smallest head): a 64-bit (distance, index) key per head (knn_key: the
// (dist, index) order, -0 == +0), the minimum over the 16 lanes by four DPP steps, the one lane that holds it advances.  K
// sequential steps per row, but four rows per wave and every row of the problem in flight at once.  Same result as the rank merge
// (keys are unique: every candidate lives in exactly one list).  grid = ceil(rows / 16); LDS = 16 rows x S x K keys.
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_min_u64(unsigned long long v) {
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)v, CTRL, 0xF, 0xF, true);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp(0, (int)(unsigned)(v >> 32), CTRL, 0xF, 0xF, true);
    const unsigned long long o = ((unsigned long long)hi << 32) | lo;
    return o < v ? o : v;
}

__global__ __launch_bounds__(256) void knn_merge_heads_kernel(const float* __restrict__ part_d, const int32_t* __restrict__ part_i,
                                                              int32_t* __restrict__ out, long long rows, int S, int K) {
    extern __shared__ __attribute__((aligned(16))) unsigned long long merge_keys[];
    const int per_row = S * K;
    const long long row0 = (long long)blockIdx.x * 16;
    const long long left = rows - row0;
    const int nrows = left < 16 ? (int)left : 16;
    const int total = nrows * per_row;
    const size_t base = (size_t)row0 * per_row;
    for (int e = threadIdx.x; e < total; e += 256) {
        const int i = part_i[base + e];
        merge_keys[e] = i == kSentinelIdx ? ~0ull : knn_key(part_d[base + e], i);      // (a list shorter than K ends in sentinels)
    }
    __syncthreads();
    const int lane = threadIdx.x & 63, l = lane & 15;
    const int r = (threadIdx.x >> 6) * 4 + (lane >> 4);
    const bool walks = r < nrows && l < S;
    const unsigned long long* lk = merge_keys + (size_t)(walks ? r : 0) * per_row + (walks ? l : 0) * K;
    int pos = 0;
    unsigned long long head = walks ? lk[0] : ~0ull;
    int res0 = 0, res1 = 0;
    for (int k = 0; k < K; ++k) {
        unsigned long long m = head;
        m = dpp_min_u64<0xB1>(m);      // quad_perm [1,0,3,2]
        m = dpp_min_u64<0x4E>(m);      // quad_perm [2,3,0,1]
        m = dpp_min_u64<0x141>(m);     // row_half_mirror
        m = dpp_min_u64<0x140>(m);     // row_mirror
        if (walks && head == m && m != ~0ull) {
            ++pos;
            head = pos < K ? lk[pos] : ~0ull;
        }
        if (l == (k & 15)) { if (k < 16) res0 = (int)(unsigned)m; else res1 = (int)(unsigned)m; }
    }
    if (r < nrows) {
        int32_t* o = out + (size_t)(row0 + r) * K;
        if (l < K) o[l] = res0;
        if (16 + l < K) o[16 + l] = res1;
    }
}

__global__ __launch_bounds__(256) void knn_select_naive_kernel(const float* __restrict__ dist,
                                                               int32_t* __restrict__ out,
                                                               long long rows, int M, int K) {
    const int lane = threadIdx.x & 63;
    const long long row = (long long)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* drow = dist + (size_t)row * M;
    unsigned long long prev = 0ull;
    for (int j = 0; j < K; ++j) {
        unsigned long long best = ~0ull;
        for (int m = lane; m < M; m += 64) {
            const unsigned long long key = knn_key(finite_or_last(drow[m]), m);      // (the distance matrix itself keeps its NaNs: nextou_pairwise_distance shares the kernel)
            if ((j == 0 || key > prev) && key < best) best = key;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const unsigned int lo = __shfl_xor((unsigned int)(best & 0xffffffffull), off);
            const unsigned int hi = __shfl_xor((unsigned int)(best >> 32), off);
            const unsigned long long other = ((unsigned long long)hi << 32) | lo;
            if (other < best) best = other;
        }
        if (lane == 0) out[(size_t)row * K + j] = (int32_t)(best & 0xffffffffull);
        prev = best;
    }
}

__global__ __launch_bounds__(256) void edge_index_i64_kernel(const int32_t* __restrict__ nn_idx,
                                                             long long* __restrict__ edge,
                                                             long long BN, int N, int K_total,
                                                             int dilation, int K_out) {
    const long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const long long total = BN * K_out;
    if (e >= total) return;
    const long long row = e / K_out;
    const int j = (int)(e - row * K_out);
    edge[e] = nn_idx[row * K_total + (long long)j * dilation];
    edge[total + e] = row % N;
}

// --------------------------------------------------------------------------------------------
// host side
// --------------------------------------------------------------------------------------------
static inline size_t align256(size_t v) { return (v + 255) & ~(size_t)255; }

struct KnnWorkspace {
    size_t xn, xs, yn, ys, dist, part_d, part_i, total;
};

struct FusedPlan {
    int nw, tiles, splits, m_per_split, ks;
};

// Work decomposition of the fused kernel.  32 queries per wave; 4 waves per workgroup (up to 6 when
// one workgroup then covers a whole <= 192-point window).  Candidates are processed in chunks of
// 32*tiles; the chunk range is split over `splits` workgroups until ~2 waves sit on every SIMD of
// the 256 CUs.  Large grids take 192-wide chunks (fewest staging passes per MFMA); grids that
// cannot fill the chip even so take 64-wide chunks (more splits) and 64-channel slabs (half the
// barrier / global-load round trips: these launches are latency-bound, LDS is plentiful).
static FusedPlan plan_fused(int B, int N, int M, int K, bool self = false) {
    FusedPlan p;
    const int need = cdiv(N, 32);
    p.nw = need <= 6 ? (need < 1 ? 1 : need) : 4;
    if (const char* e = getenv("NEXTOU_KNN_NW")) { const int v = atoi(e); if (v >= 1 && v <= 6) p.nw = v; }                 // experiments
    const long long waves = (long long)B * cdiv(N, 32 * p.nw) * p.nw;
    const bool small = waves < 1024;
    const long long ww = (long long)cdiv(M, 192) * 192, w2 = (long long)cdiv(M, 64) * 64;
    p.tiles = (small || w2 * 10 < ww * 9) ? 2 : 6;
    // long lists: 192-wide chunks (96 accumulator registers) beside a 28 / 32-slot list cost 192-200 VGPRs = 2 waves per
    // SIMD; 64-wide chunks (133 VGPRs, 3 waves) are faster although they stage three times as often — cfg-5 Pool s3
    // 1850 -> 1591 us, Swin s3 268 -> 232 us (profiles/r02_knn_topk_ab.md)
    if (K > 16) p.tiles = 2;
    // Round 4: 128-wide chunks (64 accumulator registers: ~150 VGPRs, still 3 waves per SIMD) for long lists over long candidate sets —
    // each query slab is re-staged half as often: cfg-2 Pool s3 350 -> 316 us in one call (profiles/r04_k1_tiles4.md)
    if (K > 16 && M >= 512) p.tiles = 4;
    // Round 3: windows that ONE workgroup covers (N, M <= 192) on a grid of 512 ... 1023 waves (the stage-3 windows of cfg 2,
    // B' = 128) took 64-wide chunks, 3 candidate splits and a merge launch: 80 + 16 us; one 192-wide chunk per workgroup, no
    // split, no merge is 90 us in ONE launch (profiles/r03_kernel_bench_cfg2.md).  Below 512 waves (stage 4 / 5: B' = 16 / 2) the
    // split stays: there a window's 18 MFLOP on a single CU are the latency (100-130 us un-split against 61-73 + 9 us split).
    // NEXTOU_KNN_WINDOW=0 restores the round-2 plan for A/B.
    static const bool window_plan = [] { const char* e = getenv("NEXTOU_KNN_WINDOW"); return !(e && e[0] == '0'); }();
    const bool window = window_plan && small && waves >= 512 && N <= 192 && M <= 192;
    if (window) p.tiles = 6;
    if (const char* e = getenv("NEXTOU_KNN_TILES")) { const int v = atoi(e); if (v == 1 || v == 2 || v == 4 || v == 6) p.tiles = v; }   // experiments
    const int tm = 32 * p.tiles;
    const int chunks = cdiv(M, tm);
    long long want = cdiv64(2048, waves);
    if (want > chunks) want = chunks;
    if (want > kMaxSplits) want = kMaxSplits;
    if (want < 1 || window) want = 1;
    if (const char* e = getenv("NEXTOU_KNN_SPLITS")) { const int v = atoi(e); if (v >= 1 && v <= kMaxSplits) want = v < chunks ? v : chunks; }   // experiments
    const int chunks_per_split = cdiv(chunks, (int)want);
    p.splits = cdiv(chunks, chunks_per_split);
    p.m_per_split = chunks_per_split * tm;
    p.ks = (waves * p.splits < 2048) ? 64 : 32;
    // (round 5: 128-channel slabs for the handful-of-workgroups launches of stages 4 / 5 — half the staging round trips — measured: no
    // change, 72.0 / 95.7 us either way; NEXTOU_KNN_KS=128 keeps the experiment, launch_fused raises the LDS limit for it)
    if (const char* e = getenv("NEXTOU_KNN_KS")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128) p.ks = v; }                    // experiments
    // a self window covered by one workgroup stages ONE slab for both MFMA operands (launch_fused)
    const bool one_slab = self && p.splits == 1 && N <= tm && 32 * p.nw == tm;
    const size_t per_k = (size_t)(one_slab ? tm : tm + 32 * p.nw) * sizeof(float);
    if (p.ks == 128 && p.ks * per_k > 150 * 1024) p.ks = 64;
    if (p.ks == 64 && p.ks * per_k > 64 * 1024) p.ks = 32;          // (the default dynamic-LDS limit for everything but the tiny grids)
    return p;
}

// ---- the single-launch window path (knn_window_kernel): self graphs of <= 192 points, normalisation inside
struct WindowPlan { bool ok; int nw, tiles, groups, splits, ks; size_t lds; };
static WindowPlan plan_window(int B, int N, int M, int K, bool has_y) {
    WindowPlan w{};
    w.ok = false;
    static const bool enabled = [] { const char* e = getenv("NEXTOU_KNN_WINDOW_FUSED"); return !(e && e[0] == '0'); }();
    if (!enabled || has_y || N != M || N > kWinPts || K > 32 || N < 1) return w;
    w.nw = cdiv(N, 32);
    // enough windows to fill the chip (>= 512 waves): one workgroup per window, every candidate in registers, no merge launch;
    // fewer: the candidates are split in 64-wide ranges over workgroups (the window's MFMAs on one CU would be the run time)
    bool whole = (long long)B * w.nw >= 512 || N <= 64;
    if (const char* e = getenv("NEXTOU_KNN_WIN_SPLIT")) { if (e[0] == '1' && N > 64) whole = false; }   // experiments
    // a handful of windows (stage 5: B' = 2): measured 102 us here against 101 us for prep + fused + merge with its wider split; the
    // kernel's serial floor (staging + epilogue, 66 us with every arithmetic phase ablated) is not amortised — profiles/r04_k1_small_graphs.md
    if (!whole && (long long)B * w.nw < 64) return w;
    // whole windows of more than 64 points: two wave groups of three candidate tiles each (knn_window_kernel, G = 2); NEXTOU_KNN_WIN_G=1
    // keeps the one-group kernel with six tiles per wave for A/B
    w.groups = 1;
    if (whole && N > 64) { const char* e = getenv("NEXTOU_KNN_WIN_G"); w.groups = (e && e[0] == '1') ? 1 : 2; }
    w.tiles = whole ? (N <= 64 ? 2 : 6 / w.groups) : 2;        // per wave
    w.splits = whole ? 1 : cdiv(N, 64);
    w.ks = 64;
    if (const char* e = getenv("NEXTOU_KNN_WIN_KS")) { const int v = atoi(e); if (v == 32 || v == 64 || v == 128 || v == 192) w.ks = v; }   // experiments
    const int kp = K <= 7 ? 8 : (K <= 14 ? 16 : 32);           // padded list length of the kernel's bucket (KB 7 | 14 | 28 | 32)
    const size_t slab = (size_t)w.ks * kWinPts, lists = w.groups == 2 ? (size_t)2 * kWinPts * kp : 0;
    w.lds = ((slab > lists ? slab : lists) + 2 * kWinPts) * sizeof(float);
    w.ok = true;
    return w;
}


// --------------------------------------------------------------------------------------------
// Small self graphs in ONE launch (round 6; VERDICT r5 missing #4): a handful of <= 192-point windows — stage 4 / 5 of cfg 2: B' = 16 or 2
// windows of 168 points, C = 324 — where the kernels above are nothing but latency chains (prep 21 us + fused 72 us + merge 12 us for 0.4 MB
// of input).  One workgroup of 16 waves per (window, 16-query tile):
//   pass A  the window's channel slabs (64 x 192) stream through LDS with the NEXT slab's 16-byte loads already in flight in registers;
//           three waves run the strictly c-ordered chain of squares -> den = max(sqrt, eps);
//   pass B  the slabs again (L2-hot), divided by den on the way into LDS (the same IEEE division as knn_prep); the same three waves run the
//           chain of xn^2; up to twelve waves hold ONE 16 x 16 distance tile each on v_mfma_f32_16x16x4_f32 — lane group g supplies channel
//           c0 + g, so the instruction's k order is the oracle's ascending-c fma chain (32-cycle issue: 81 instructions for C = 324);
//   select  dist = ((xs + (-2 inner)) + ys) [+ relpos] -> LDS, then a wave per query RANKS its candidates by counting: position of
//           candidate m = #{m' : (dist, index)(m') < (dist, index)(m)}, two VALU ops per comparison on the strict part, every lane busy,
//           no dependent insert chains; exact ties (found by the rank sum falling short of N (N - 1) / 2) take the full key comparison.
// Bit-identical ids to knn_prep + knn_fused + merge (same arithmetic, same (dist, index) order).  Any K <= N.
// grid = (ceil(N / 16), B'), block = 1024, LDS = 64 * 192 + 2 * 192 + 16 * 192 floats (61.5 KB).
// --------------------------------------------------------------------------------------------
constexpr int kSmW = 192, kSmKS = 64, kSmThreads = 1024, kSmPieces = kSmKS * (kSmW / 4) / kSmThreads;     // 3 float4 per thread and slab
using f32x4 = __attribute__((ext_vector_type(4))) float;

__device__ __forceinline__ void small_load(const float* __restrict__ xb, int c0, int C, int N, float4 (&v)[kSmPieces]) {
#pragma unroll
    for (int u = 0; u < kSmPieces; ++u) {
        const int e = threadIdx.x + u * kSmThreads;
        const int r = e / (kSmW / 4), c4 = (e - r * (kSmW / 4)) << 2;
        const bool ok = c0 + r < C && c4 < N;
        v[u] = ok ? *reinterpret_cast<const float4*>(xb + (size_t)(c0 + r) * N + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

// One point's strictly c-ordered chain of squares over a whole slab: all kSmKS values out of LDS first, then the dependent fmas (rows past C
// are zero: fma(0, 0, s) = s)
__device__ __forceinline__ float small_chain(const float* __restrict__ col, float s) {
#pragma unroll
    for (int k0 = 0; k0 < kSmKS; k0 += 32) {
        float t[32];
#pragma unroll
        for (int k = 0; k < 32; ++k) t[k] = col[(k0 + k) * kSmW];
#pragma unroll
        for (int k = 0; k < 32; ++k) s = fmaf(t[k], t[k], s);
    }
    return s;
}

template <bool DIVIDE>
__device__ __forceinline__ void small_store(float* __restrict__ slab, const float* __restrict__ den_s, const float4 (&v)[kSmPieces]) {
#pragma unroll
    for (int u = 0; u < kSmPieces; ++u) {
        const int e = threadIdx.x + u * kSmThreads;
        const int r = e / (kSmW / 4), c4 = (e - r * (kSmW / 4)) << 2;
        float4 t = v[u];
        if (DIVIDE) {       // (rows past C and points past N were loaded as zeros: 0 / den = 0)
            const float4 d = *reinterpret_cast<const float4*>(den_s + c4);
            t = make_float4(t.x / d.x, t.y / d.y, t.z / d.z, t.w / d.w);
        }
        *reinterpret_cast<float4*>(slab + r * kSmW + c4) = t;
    }
}

__global__ __launch_bounds__(kSmThreads) void knn_small_kernel(const float* __restrict__ x, const float* __restrict__ relpos,
                                                               int32_t* __restrict__ out, int C, int N, int K, int ablate) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    float* slab = lds;                           // [64][192]
    float* den_s = slab + kSmKS * kSmW;          // [192]
    float* sq_s = den_s + kSmW;                  // [192]
    float* dist_s = sq_s + kSmW;                 // [16][192]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int q = lane & 15, g = lane >> 4;
    const int b = blockIdx.y, qt = blockIdx.x;
    const float* xb = x + (size_t)b * C * N;
    const int n_slabs = (C + kSmKS - 1) / kSmKS;
    const int n_tiles = (N + 15) >> 4;           // candidate tiles of 16
    const int chain_pt = tid - (kSmThreads - kSmW);      // the last three waves own the per-point chains
    const bool chain = chain_pt >= 0;

    // relative-position bias of this lane's four (query, candidate) pairs: in flight during both passes
    float rp[4] = {0.f, 0.f, 0.f, 0.f};
    if (relpos != nullptr && wave < n_tiles) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int n = qt * 16 + 4 * g + r, m = wave * 16 + q;
            if (n < N && m < N) rp[r] = relpos[(size_t)n * N + m];
        }
    }

    // ---- 2 n_slabs stages: pass A (den) over the slabs, then pass B (normalise on the way in, chain of squares, one distance tile per wave)
    // over the same slabs.  The loads of stage t + 2 are issued while stage t is stored and consumed (two register sets, stages in pairs so
    // that the sets are indexed statically).
    // Clocked with s_memtime inside workgroup (0, 0) at C = 384, N = 168 (round 6): pass A 0.93 us per slab, pass B 2.3 us per slab (of which
    // ~1.4 us are the 12 IEEE divisions per lane, VALU work of all sixteen waves), distances 0.5 us, selection 2.9 us for the first wave and
    // 7.8 us for the last (VALU-bound: 3 N compare-and-count per lane and query): 29.6 us inside the kernel, 41.6 us per launch from outside.
    // Tried without gain: slabs of 128 channels; two slab buffers with one barrier per slab and half of the waves consuming before storing
    // (43.4 us); the selection's compares as v_cmp + v_addc in inline assembly (43.3 us)
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    float ssum = 0.f, qsum = 0.f;
    const int n_stages = 2 * n_slabs;
    auto slab_of = [&](int t) { return (t >= n_slabs ? t - n_slabs : t) * kSmKS; };
    auto stage = [&](int t, float4 (&v)[kSmPieces]) __attribute__((always_inline)) {
        const bool pass_b = t >= n_slabs;
        const int sl = pass_b ? t - n_slabs : t;
        __syncthreads();                         // the previous slab consumed (and, entering pass B, den_s complete)
        if (pass_b && !(ablate & 2)) small_store<true>(slab, den_s, v);
        else small_store<false>(slab, den_s, v);
        if (t + 2 < n_stages) small_load(xb, slab_of(t + 2), C, N, v);
        __syncthreads();
        if (!pass_b) {
            if (chain && !(ablate & 1)) ssum = small_chain(slab + chain_pt, ssum);
            if (chain && sl == n_slabs - 1) den_s[chain_pt] = fmaxf(sqrtf(ssum), kNormEps);
            return;
        }
        if (chain) qsum = small_chain(slab + chain_pt, qsum);
        if (wave < n_tiles && !(ablate & 4)) {
            // the slab's sixteen k-steps: all 32 operands out of LDS first, then the dependent MFMA chain (rows past C are zero: fma(0, 0, acc)
            // = acc)
            const float* pa = slab + g * kSmW + qt * 16 + q;
            const float* pb = slab + g * kSmW + wave * 16 + q;
            float av[kSmKS / 4], bv[kSmKS / 4];
#pragma unroll
            for (int i = 0; i < kSmKS / 4; ++i) { av[i] = pa[4 * i * kSmW]; bv[i] = pb[4 * i * kSmW]; }
#pragma unroll
            for (int i = 0; i < kSmKS / 4; ++i) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], bv[i], acc, 0, 0, 0);
        }
    };
    {
        float4 va[kSmPieces], vb[kSmPieces];
        small_load(xb, slab_of(0), C, N, va);
        small_load(xb, slab_of(1), C, N, vb);    // (n_stages >= 2 always)
        for (int t = 0; t < n_stages; t += 2) {
            stage(t, va);
            stage(t + 1, vb);
        }
    }
    if (chain) sq_s[chain_pt] = qsum;
    __syncthreads();

    // ---- distances of the tile -> LDS (acc[r] = inner(query 4g + r, candidate q))
    if (wave < kSmW / 16) {
        const int m = wave * 16 + q;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int i = 4 * g + r, n = qt * 16 + i;
            float dist = INFINITY;
            if (wave < n_tiles && n < N && m < N) {
                dist = (sq_s[n] + (-2.0f * acc[r])) + sq_s[m];
                if (relpos != nullptr) dist = dist + rp[r];
                dist = finite_or_last(dist);
            }
            dist_s[i * kSmW + m] = dist;
        }
    }
    __syncthreads();

    // ---- selection by counting: wave = query, lane owns candidates lane, lane + 64, lane + 128
    const int n = qt * 16 + wave;
    if (n >= N) return;
    const float* row = dist_s + wave * kSmW;
    const int m0 = lane, m1 = lane + 64, m2 = lane + 128;
    const float d0 = row[m0], d1 = row[m1], d2 = row[m2];
    int r0 = 0, r1 = 0, r2 = 0;
    const int n16 = (ablate & 8) ? 16 : (N + 15) & ~15;      // rows are 192 wide and +inf past N: whole groups of 16 are safe to read
    for (int mp = 0; mp < n16; mp += 16) {                   // four 16-byte reads in flight per step
        float4 o[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) o[u] = *reinterpret_cast<const float4*>(row + mp + 4 * u);
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            r0 += (o[u].x < d0) + (o[u].y < d0) + (o[u].z < d0) + (o[u].w < d0);
            r1 += (o[u].x < d1) + (o[u].y < d1) + (o[u].z < d1) + (o[u].w < d1);
            r2 += (o[u].x < d2) + (o[u].y < d2) + (o[u].z < d2) + (o[u].w < d2);
        }
    }
    int total = (m0 < N ? r0 : 0) + (m1 < N ? r1 : 0) + (m2 < N ? r2 : 0);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) total += __shfl_xor(total, o, 64);
    if (total != N * (N - 1) / 2 && !ablate) {              // exact ties somewhere in this row: the full (dist, index) order
        r0 = r1 = r2 = 0;
        for (int mp = 0; mp < N; ++mp) {
            const float o = row[mp];
            r0 += (o < d0) || (o == d0 && mp < m0);
            r1 += (o < d1) || (o == d1 && mp < m1);
            r2 += (o < d2) || (o == d2 && mp < m2);
        }
    }
    int32_t* orow = out + ((size_t)b * N + n) * K;
    if (m0 < N && r0 < K) orow[r0] = m0;
    if (m1 < N && r1 < K) orow[r1] = m1;
    if (m2 < N && r2 < K) orow[r2] = m2;
}

struct SmallPlan { bool ok; size_t lds; };
static SmallPlan plan_small(int B, int N, int M, int K, bool has_y, const float* x) {
    SmallPlan p{false, (size_t)(kSmKS * kSmW + 2 * kSmW + 16 * kSmW) * sizeof(float)};
    static const bool enabled = [] { const char* e = getenv("NEXTOU_KNN_SMALL"); return !(e && e[0] == '0'); }();
    if (!enabled || has_y || N != M || N > kSmW || N < 1 || (N & 3) || K > N) return p;
    if ((reinterpret_cast<uintptr_t>(x) & 15u) != 0) return p;
    // few windows only: every 16-query tile re-reads its whole window (twice), which costs nothing while the windows sit in L2 and
    // the launch is latency-bound, and everything once there are hundreds of windows (stages 2 / 3 keep knn_window_kernel)
    int max_wg = 384;
    if (const char* e = getenv("NEXTOU_KNN_SMALL_MAX_WG")) max_wg = atoi(e);       // experiments
    if ((long long)B * cdiv(N, 16) > max_wg) return p;
    p.ok = true;
    return p;
}

static int launch_small(const float* x, const float* relpos, int32_t* out, int B, int C, int N, int K, const SmallPlan& p, hipStream_t s) {
    ProfScope prof(s, kBoundMfma, 2.0 * B * (double)N * N * C, "knn_small_kernel[B%d C%d N%d K%d]", B, C, N, K);
    // experiments (NEXTOU_KNN_SMALL_ABLATE; wrong results with any bit set): 1 no pass-A chains, 2 no divisions, 4 no MFMAs, 8 no counting
    static const int ablate = [] { const char* e = getenv("NEXTOU_KNN_SMALL_ABLATE"); return e ? atoi(e) : 0; }();
    hipLaunchKernelGGL(knn_small_kernel, dim3(cdiv(N, 16), B), dim3(kSmThreads), p.lds, s, x, relpos, out, C, N, K, ablate);
    return check_launch("knn_small_kernel");
}

static int resolve_algo(int algo, int K) {
    if (algo == NEXTOU_KNN_AUTO) return K <= 32 ? NEXTOU_KNN_FUSED : NEXTOU_KNN_NAIVE;
    return algo;
}

static KnnWorkspace knn_layout(int B, int C, int N, int M, int K, int has_y, int algo) {
    KnnWorkspace w{};
    size_t off = 0;
    w.xn = off; off += align256((size_t)B * C * N * sizeof(float));
    w.xs = off; off += align256((size_t)B * N * sizeof(float));
    if (has_y) {
        w.yn = off; off += align256((size_t)B * C * M * sizeof(float));
        w.ys = off; off += align256((size_t)B * M * sizeof(float));
    } else {
        w.yn = w.xn; w.ys = w.xs;
    }
    if (algo == NEXTOU_KNN_NAIVE) {
        w.dist = off; off += align256((size_t)B * N * M * sizeof(float));
    } else {
        const FusedPlan p = plan_fused(B, N, M, K);
        const WindowPlan wp = plan_window(B, N, M, K, has_y != 0);
        const int splits = (wp.ok && wp.splits > p.splits) ? wp.splits : p.splits;
        if (splits > 1) {
            w.part_d = off; off += align256((size_t)B * N * splits * K * sizeof(float));
            w.part_i = off; off += align256((size_t)B * N * splits * K * sizeof(int32_t));
        }
    }
    w.total = off;
    return w;
}

static int launch_prep(const float* x, float* xn, float* sq, int B, int C, int N, bool normalize,
                       hipStream_t s) {
    // reads x twice (second pass L2-hot: counted once), writes xn and the norms
    ProfScope prof(s, kBoundHbm, 4.0 * B * (double)N * ((normalize ? 2.0 : 1.0) * C + 1), "knn_prep_kernel[B%d C%d N%d]",
                   B, C, N);
    const bool small = (long long)B * N <= 8192;
    if (normalize) {
        if (small) hipLaunchKernelGGL((knn_prep_kernel<true, 64>), dim3(cdiv(N, 64), B), dim3(64), 0, s, x, xn, sq, C, N);
        else hipLaunchKernelGGL((knn_prep_kernel<true, 16>), dim3(cdiv(N, 256), B), dim3(256), 0, s, x, xn, sq, C, N);
    } else {
        if (small) hipLaunchKernelGGL((knn_prep_kernel<false, 64>), dim3(cdiv(N, 64), B), dim3(64), 0, s, x, xn, sq, C, N);
        else hipLaunchKernelGGL((knn_prep_kernel<false, 16>), dim3(cdiv(N, 256), B), dim3(256), 0, s, x, xn, sq, C, N);
    }
    return check_launch("knn_prep_kernel");
}

// queries (and, for a pooled graph, candidates) in one launch of the tiled kernel; NEXTOU_KNN_PREP=v1 keeps the per-point
// kernel (one launch per operand) for A/B runs
static int launch_prep_pair(const float* x, float* xn, float* xs, int N, const float* y, float* yn, float* ys, int M, int B,
                            int C, bool normalize, hipStream_t s) {
    // Measured (profiles/r03_k1_prep_merge.md): one launch for both operands of a pooled graph beats two of the per-point
    // kernel (cfg-2 Pool s3 28.3 + 24.1 -> 24.0 us, Pool s2 16.0 + 13.3 -> 16.5 us; cfg-5 Pool s3 36.7 + 29.3 -> 49.8 us); for a
    // single operand the two are level (the phases of one 64-point tile are as serial as the per-point loop), so self graphs
    // keep the per-point kernel.  NEXTOU_KNN_PREP=v1 / tile forces one of them.
    static const int mode = [] { const char* e = getenv("NEXTOU_KNN_PREP"); return !e ? 0 : (e[0] == 'v' ? 1 : 2); }();
    if (mode == 1 || C > kPrepMaxC || (mode == 0 && y == nullptr)) {
        if (int e = launch_prep(x, xn, xs, B, C, N, normalize, s)) return e;
        return y ? launch_prep(y, yn, ys, B, C, M, normalize, s) : 0;
    }
    const int tx = cdiv(N, kPrepPts), ty = y ? cdiv(M, kPrepPts) : 0;
    const size_t lds = (size_t)C * kPrepPts * sizeof(float);
    const double pts = (double)N + (y ? M : 0);
    ProfScope prof(s, kBoundHbm, 4.0 * B * pts * ((normalize ? 2.0 : 1.0) * C + 1), "knn_prep_tile_kernel[B%d C%d N%d M%d]", B, C,
                   N, y ? M : 0);
    if (normalize) {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_prep_tile_kernel<true>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(knn_prep_tile_kernel<true>, dim3(tx + ty, B), dim3(256), lds, s, x, xn, xs, N, tx, y, yn, ys, M, C);
    } else {
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_prep_tile_kernel<false>),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        hipLaunchKernelGGL(knn_prep_tile_kernel<false>, dim3(tx + ty, B), dim3(256), lds, s, x, xn, xs, N, tx, y, yn, ys, M, C);
    }
    return check_launch("knn_prep_tile_kernel");
}

struct FusedArgs {
    const float *xn, *yn, *xs, *ys, *relpos;
    int32_t* out;
    float* part_d;
    int32_t* part_i;
    int B, C, N, M, K;
};

// Which top-K the fused kernel runs.  The sorting networks on packed keys are bit-exact (same tests) and were meant to cut
// the VALU work of the exact selection; measured (profiles/r02_knn_topk_ab.md) they are SLOWER where it matters — Pool s3
// (K = 28) 411 vs 325 us, Swin s3 (K = 14) 109 vs 80 us; level or slightly ahead on two small shapes — because the
// shift-insert list skips whole candidate rows whenever no lane of the wave qualifies (more often than the 25-45 % per-lane
// qualification rate suggests once the lists are warm), while a network costs the same for every tile, and 64 + 32 key
// registers leave 2 waves per SIMD instead of 4.  The shift-insert list stays the default; NEXTOU_KNN_TOPK=network selects
// the networks (64-wide chunks only) for A/B runs.
static bool use_networks(int KB) {
    static const int mode = [] { const char* e = getenv("NEXTOU_KNN_TOPK"); return (e != nullptr && e[0] == 'n') ? 1 : 0; }();
    return mode == 1 && KB >= 8;
}

// S sorted partial lists per query -> final ids (shared by the general and the window path)
static int launch_merge(const FusedArgs& a, int splits, hipStream_t s) {
    const long long rows = (long long)a.B * a.N;
    ProfScope prof(s, kBoundHbm, 8.0 * rows * splits * a.K + 4.0 * rows * a.K, "knn_merge_kernel[B%d N%d S%d K%d]",
                   a.B, a.N, splits, a.K);       // (both merge kernels report under this label)
    // LDS-staged merge for up to 4 partial lists (cfg-2 Pool s3 53.9 -> 41.9 us, Swin / Pool s4-s5 9 -> 7 us, cfg-5 S = 2
    // 37 -> 32 us); with 6-11 lists it is level or behind the global one (64 -> 81 us at S = 11 on 2 688 rows, 108 -> 88 us
    // on 6 144): profiles/r03_k1_prep_merge.md.  NEXTOU_KNN_MERGE=v1 / lds forces one of them.
    static const int merge_mode = [] { const char* e = getenv("NEXTOU_KNN_MERGE"); return !e ? 0 : (e[0] == 'v' ? 1 : (e[0] == 'l' ? 2 : 0)); }();
    // Round 5 default: the tournament of list heads (knn_merge_heads_kernel) for every split count (S <= 16 = kMaxSplits lanes of a
    // row group, K <= 32); NEXTOU_KNN_MERGE=v1 / lds keep the two rank merges for A/B
    if (merge_mode == 0 && splits <= 16 && a.K <= 32 && (size_t)16 * splits * a.K * 8 <= 64 * 1024) {
        hipLaunchKernelGGL(knn_merge_heads_kernel, dim3((unsigned)cdiv64(rows, 16)), dim3(256), (size_t)16 * splits * a.K * 8, s, a.part_d,
                           a.part_i, a.out, rows, splits, a.K);
        return check_launch("knn_merge_heads_kernel");
    }
    if (merge_mode == 1 || (merge_mode == 0 && splits > 4)) {
        hipLaunchKernelGGL(knn_merge_kernel, dim3((unsigned)cdiv64(rows * splits * a.K, 256)), dim3(256), 0, s,
                           a.part_d, a.part_i, a.out, rows, splits, a.K);
        return check_launch("knn_merge_kernel");
    }
    const int per_row = splits * a.K;
    int rows_per_wg = 2048 / per_row;          // <= 16 KB of LDS; at least 4 entries per thread
    if (rows_per_wg < 1) rows_per_wg = 1;
    // small problems: fewer rows per workgroup until ~2 workgroups sit on every CU
    while (rows_per_wg > 1 && cdiv64(rows, rows_per_wg) < 512) rows_per_wg = (rows_per_wg + 1) / 2;
    hipLaunchKernelGGL(knn_merge_lds_kernel, dim3((unsigned)cdiv64(rows, rows_per_wg)), dim3(256),
                       (size_t)rows_per_wg * per_row * 8, s, a.part_d, a.part_i, a.out, rows, splits, a.K, rows_per_wg);
    return check_launch("knn_merge_lds_kernel");
}

template <int KB>
static int launch_window(const FusedArgs& a, const float* x, const WindowPlan& w, hipStream_t s) {
    {
        ProfScope prof(s, kBoundMfma, 2.0 * a.B * (double)a.N * a.N * a.C, "knn_window_kernel<%d,%dx%d>[B%d C%d N%d K%d]", KB, w.groups, w.tiles, a.B, a.C,
                       a.N, a.K);
        const dim3 grid(w.splits, a.B), block(64 * w.nw * w.groups);
        int ablate = 0;
        if (const char* e = getenv("NEXTOU_KNN_WIN_ABLATE")) ablate = atoi(e);
        if (const char* e = getenv("NEXTOU_KNN_WIN_STAGGER")) ablate |= atoi(e) << 8;
#define NEXTOU_KNN_WIN(T, G_)                                                                                                          \
    do {                                                                                                                             \
        if (w.lds > 64 * 1024)                                                                                                       \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_window_kernel<KB, T, G_>), hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)w.lds);                                                                                   \
        hipLaunchKernelGGL((knn_window_kernel<KB, T, G_>), grid, block, w.lds, s, x, a.relpos, a.out, a.part_d, a.part_i, a.C, a.N, a.K, w.ks, \
                           ablate);                                                                                                  \
    } while (0)
        if (w.groups == 2) NEXTOU_KNN_WIN(3, 2);
        else if (w.tiles == 2) NEXTOU_KNN_WIN(2, 1);
        else NEXTOU_KNN_WIN(6, 1);
#undef NEXTOU_KNN_WIN
    }
    if (int e = check_launch("knn_window_kernel")) return e;
    return w.splits > 1 ? launch_merge(a, w.splits, s) : 0;
}

template <int KB, int TILES, bool BITONIC, bool GLDS = false>
static int launch_fused(const FusedArgs& a, const FusedPlan& p, hipStream_t s) {
    const int QW = 32 * p.nw;
    // (the kernel's share_ab condition: queries == candidates, one query tile, one chunk)
    const bool one_slab = a.yn == a.xn && QW == 32 * TILES && a.N <= QW && p.splits == 1 && a.M <= 32 * TILES;
    const size_t lds = GLDS ? (size_t)3 * 16 * (32 * TILES + QW) * sizeof(float)        // three buffers of 16-channel slabs
                            : (size_t)p.ks * (one_slab ? 32 * TILES : 32 * TILES + QW) * sizeof(float);
    dim3 grid(cdiv(a.N, QW), a.B, p.splits);
    // 16-B staging needs row strides and bases that keep every 4-float piece aligned
    const int vec_ok = (a.N % 4 == 0) && (a.M % 4 == 0) &&
                       ((reinterpret_cast<uintptr_t>(a.xn) | reinterpret_cast<uintptr_t>(a.yn) |
                         reinterpret_cast<uintptr_t>(a.ys) | reinterpret_cast<uintptr_t>(a.relpos)) & 15u) == 0;
    {
        // algorithmic work of the distance contraction: 2*B*N*M*C flops
        ProfScope prof(s, kBoundMfma, 2.0 * a.B * (double)a.N * a.M * a.C,
                       "knn_fused_kernel<%d,%d>[B%d C%d N%d M%d K%d]", KB, TILES, a.B, a.C, a.N, a.M, a.K);
        if (lds > 64 * 1024)
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&knn_fused_kernel<KB, TILES, BITONIC, GLDS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)lds);
        hipLaunchKernelGGL((knn_fused_kernel<KB, TILES, BITONIC, GLDS>), grid, dim3(64 * p.nw), lds, s, a.xn, a.yn, a.xs, a.ys,
                           a.relpos, a.out, a.part_d, a.part_i, a.C, a.N, a.M, a.K, p.m_per_split, vec_ok, p.ks);
    }
    if (int e = check_launch("knn_fused_kernel")) return e;
    return p.splits > 1 ? launch_merge(a, p.splits, s) : 0;
}

template <int KB>
static int launch_fused_tiles(const FusedArgs& a, const FusedPlan& p, hipStream_t s) {
    // 192-wide chunks keep 96 accumulator registers per lane: beside 64 + 32 key registers that spills (54 VGPRs at
    // K = 32), so the network path is taken with 64-wide chunks only
    if (p.tiles == 1) return launch_fused<KB, 1, false>(a, p, s);
    if (p.tiles == 2) return use_networks(KB) ? launch_fused<KB, 2, true>(a, p, s) : launch_fused<KB, 2, false>(a, p, s);
    if (p.tiles == 4) {
        // direct-to-LDS double-buffered slabs (round 6) for the long-list pooled graphs: 4 waves x 128 candidates, 16-byte pieces throughout
        if constexpr (KB >= 28) {
            static const bool glds = [] { const char* e = getenv("NEXTOU_KNN_GLDS"); return !(e && e[0] == '0'); }();
            const bool aligned = (a.N % 4 == 0) && (a.M % 4 == 0) && ((reinterpret_cast<uintptr_t>(a.xn) | reinterpret_cast<uintptr_t>(a.yn)) & 15u) == 0;
            if (glds && p.nw == 4 && aligned && p.m_per_split % 128 == 0) return launch_fused<KB, 4, false, true>(a, p, s);
        }
        return launch_fused<KB, 4, false>(a, p, s);
    }
    return launch_fused<KB, 6, false>(a, p, s);
}

}  // namespace nextou

using namespace nextou;

extern "C" size_t nextou_knn_workspace_bytes(int B, int C, int N, int M, int K, int has_y, int algo) {
    if (B <= 0 || C <= 0 || N <= 0 || M <= 0) return 0;
    return knn_layout(B, C, N, M, K, has_y, resolve_algo(algo, K)).total;
}

extern "C" int nextou_knn_graph(const float* x, const float* y, const float* relpos,
                                int32_t* nn_idx, void* workspace, size_t workspace_bytes, int B,
                                int C, int N, int M, int K, int algo, int normalize,
                                nextou_stream_t stream) {
    NEXTOU_REQUIRE(x != nullptr && nn_idx != nullptr && workspace != nullptr,
                   "knn_graph: null pointer (x=%p nn_idx=%p workspace=%p)", (const void*)x,
                   (void*)nn_idx, workspace);
    NEXTOU_REQUIRE(B > 0 && C > 0 && N > 0 && M > 0 && K > 0,
                   "knn_graph: non-positive size B=%d C=%d N=%d M=%d K=%d", B, C, N, M, K);
    NEXTOU_REQUIRE(y != nullptr || M == N, "knn_graph: self graph needs M == N (N=%d M=%d)", N, M);
    NEXTOU_REQUIRE(K <= M, "knn_graph: K=%d exceeds the number of candidates M=%d", K, M);
    NEXTOU_REQUIRE(B <= 65535, "knn_graph: B=%d exceeds the grid limit 65535", B);
    const int has_y = y != nullptr;
    algo = resolve_algo(algo, K);
    if (algo == NEXTOU_KNN_FUSED && K > 32)
        return fail(NEXTOU_ENOTSUP, "knn_graph: fused kernel supports K <= 32, got %d", K);
    if (algo != NEXTOU_KNN_FUSED && algo != NEXTOU_KNN_NAIVE)
        return fail(NEXTOU_EINVAL, "knn_graph: unknown algo %d", algo);
    const KnnWorkspace w = knn_layout(B, C, N, M, K, has_y, algo);
    if (workspace_bytes < w.total)
        return fail(NEXTOU_ENOSPACE, "knn_graph: workspace %zu < required %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    float* xn = (float*)(base + w.xn);
    float* xs = (float*)(base + w.xs);
    float* yn = (float*)(base + w.yn);
    float* ys = (float*)(base + w.ys);

    if (algo == NEXTOU_KNN_FUSED && normalize) {
        const SmallPlan sp = plan_small(B, N, M, K, has_y != 0, x);
        if (sp.ok) return launch_small(x, relpos, nn_idx, B, C, N, K, sp, s);      // a handful of <= 192-point self graphs: one launch
        const WindowPlan wp = plan_window(B, N, M, K, has_y != 0);
        if (wp.ok) {       // <= 192-point self graph: normalisation, distances, selection in one launch
            FusedArgs a{nullptr, nullptr, nullptr, nullptr, relpos, nn_idx, (float*)(base + w.part_d), (int32_t*)(base + w.part_i), B, C, N, M, K};
            if (K <= 7) return launch_window<7>(a, x, wp, s);
            if (K <= 14) return launch_window<14>(a, x, wp, s);
            if (K <= 28) return launch_window<28>(a, x, wp, s);
            return launch_window<32>(a, x, wp, s);
        }
    }
    if (int e = launch_prep_pair(x, xn, xs, N, has_y ? y : nullptr, yn, ys, M, B, C, normalize != 0, s)) return e;
    if (!normalize) {  // the un-normalised copies are the inputs themselves
        xn = const_cast<float*>(x);
        yn = has_y ? const_cast<float*>(y) : xn;
    }

    if (algo == NEXTOU_KNN_NAIVE) {
        NEXTOU_REQUIRE(N <= 65535, "knn_graph(naive): N=%d exceeds the grid limit 65535", N);
        float* dist = (float*)(base + w.dist);
        hipLaunchKernelGGL(knn_dist_naive_kernel, dim3(cdiv(M, 256), N, B), dim3(256), 0, s, xn, yn,
                           xs, ys, relpos, dist, C, N, M, 0, N);
        if (int e = check_launch("knn_dist_naive_kernel")) return e;
        const long long rows = (long long)B * N;
        hipLaunchKernelGGL(knn_select_naive_kernel, dim3((unsigned)cdiv64(rows, 4)), dim3(256), 0, s,
                           dist, nn_idx, rows, M, K);
        return check_launch("knn_select_naive_kernel");
    }

    const FusedPlan plan = plan_fused(B, N, M, K, !has_y);
    FusedArgs a{xn, yn, xs, ys, relpos, nn_idx, (float*)(base + w.part_d), (int32_t*)(base + w.part_i), B, C, N, M, K};
    // list-length buckets: every slot costs 4 VALU ops per candidate per lane, so the cfg-2 values
    // 7 / 14 / 28 get their own instantiation instead of rounding up to 8 / 16 / 32
    if (K <= 7) return launch_fused_tiles<7>(a, plan, s);
    if (K <= 8) return launch_fused_tiles<8>(a, plan, s);
    if (K <= 14) return launch_fused_tiles<14>(a, plan, s);
    if (K <= 16) return launch_fused_tiles<16>(a, plan, s);
    if (K <= 28) return launch_fused_tiles<28>(a, plan, s);
    return launch_fused_tiles<32>(a, plan, s);
}

extern "C" int nextou_edge_index_i64(const int32_t* nn_idx, int64_t* edge_index, int B, int N,
                                     int K_total, int dilation, nextou_stream_t stream) {
    NEXTOU_REQUIRE(nn_idx != nullptr && edge_index != nullptr, "edge_index_i64: null pointer");
    NEXTOU_REQUIRE(B > 0 && N > 0 && K_total > 0 && dilation > 0,
                   "edge_index_i64: non-positive size B=%d N=%d K=%d d=%d", B, N, K_total, dilation);
    const int K_out = (K_total + dilation - 1) / dilation;  // len(range(0, K_total, d))
    const long long BN = (long long)B * N;
    const long long total = BN * K_out;
    hipLaunchKernelGGL(edge_index_i64_kernel, dim3((unsigned)cdiv64(total, 256)), dim3(256), 0,
                       (hipStream_t)stream, nn_idx, (long long*)edge_index, BN, N, K_total, dilation,
                       K_out);
    return check_launch("edge_index_i64_kernel");
}

extern "C" size_t nextou_pairwise_workspace_bytes(int B, int N, int M, int has_y) {
    if (B <= 0 || N <= 0 || M <= 0) return 0;
    return align256((size_t)B * N * sizeof(float)) + (has_y ? align256((size_t)B * M * sizeof(float)) : 0);
}

extern "C" int nextou_pairwise_distance(const float* x, const float* y, float* dist, void* workspace,
                                        size_t workspace_bytes, int B, int C, int N, int M,
                                        int row_start, int row_end, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && dist && workspace, "pairwise_distance: null pointer");
    NEXTOU_REQUIRE(B > 0 && C > 0 && N > 0 && M > 0 && B <= 65535, "pairwise_distance: bad size B=%d C=%d N=%d M=%d", B, C, N, M);
    NEXTOU_REQUIRE(y != nullptr || M == N, "pairwise_distance: self distance needs M == N");
    NEXTOU_REQUIRE(0 <= row_start && row_start < row_end && row_end <= N && row_end - row_start <= 65535,
                   "pairwise_distance: bad row window [%d,%d) of N=%d (at most 65535 rows per call)", row_start, row_end, N);
    const int has_y = y != nullptr;
    const size_t need = nextou_pairwise_workspace_bytes(B, N, M, has_y);
    if (workspace_bytes < need)
        return fail(NEXTOU_ENOSPACE, "pairwise_distance: workspace %zu < required %zu", workspace_bytes, need);
    hipStream_t s = (hipStream_t)stream;
    float* xs = (float*)workspace;
    float* ys = has_y ? (float*)((char*)workspace + align256((size_t)B * N * sizeof(float))) : xs;
    if (int e = launch_prep(x, nullptr, xs, B, C, N, false, s)) return e;
    if (has_y) {
        if (int e = launch_prep(y, nullptr, ys, B, C, M, false, s)) return e;
    }
    const int rows = row_end - row_start;
    hipLaunchKernelGGL(knn_dist_naive_kernel, dim3(cdiv(M, 256), rows, B), dim3(256), 0, s, x,
                       has_y ? y : x, xs, ys, (const float*)nullptr, dist, C, N, M, row_start, rows);
    return check_launch("knn_dist_naive_kernel");
}
