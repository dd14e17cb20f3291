// K1 — dense kNN graph for gfx950 (MI355X).
//
// Replaces the reference's F.normalize -> bmm -> add -> add -> (+relative_pos) -> neg -> topk
// chain (reference network_architecture/torch_edge.py:151-163, 58-110, 12-55) with
//   knn_prep_kernel   : L2-normalise over channels + squared norms of the normalised rows
//   knn_fused_kernel  : f32 MFMA (32x32x2) distance tiles, LDS-staged channel slabs, and a
//                       streaming per-query top-K kept entirely in registers; the (B,N,M)
//                       distance matrix never exists in HBM
//   knn_dist_naive / knn_select_naive : the materialising fallback (any K), also the on-GPU
//                       cross-check of the "MFMA == fmaf chain" claim.
//
// Arithmetic contract (must stay bit-identical to oracle/knn_canonical.c):
//   den = max(sqrtf(chain(x*x)), 1e-12f); xn = x / den; xs = chain(xn*xn);
//   inner = chain(xn*yn) with acc = fmaf(a_c, b_c, acc), c ascending, acc0 = 0;
//   dist = ((xs + (-2*inner)) + ys) [+ relpos];   order by (dist, index).
// The f32 MFMA is bitwise a k-ordered fmaf chain (MI355X_MICROARCH.md, "Matrix cores"), so the
// MFMA and VALU paths agree bit for bit.  This file is compiled with -ffp-contract=off and
// correctly rounded divide/sqrt.
#include "common.h"
#include <cmath>

namespace nextou {

constexpr float kNormEps = 1e-12f;  // F.normalize eps (torch_edge.py:154-155,160)
constexpr int kSentinelIdx = 0x7fffffff;

using f32x16 = __attribute__((ext_vector_type(16))) float;

// --------------------------------------------------------------------------------------------
// prep: one thread per point, channel loop strided by N (coalesced across the wave).
// --------------------------------------------------------------------------------------------
// NORMALIZE = false: inputs are used as they are (dense_knn_matrix / *_pairwise_distance called
// directly, torch_edge.py:58-110 do not normalise); only the squared norms are produced.
template <bool NORMALIZE>
__global__ __launch_bounds__(256) void knn_prep_kernel(const float* __restrict__ x,
                                                       float* __restrict__ xn,
                                                       float* __restrict__ sq, int C, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (n >= N) return;
    const float* xb = x + (size_t)b * C * N + n;
    float* xo = xn + (size_t)b * C * N + n;
    float s = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * N];
        s = fmaf(v, v, s);
    }
    if (!NORMALIZE) {
        sq[(size_t)b * N + n] = s;
        return;
    }
    const float den = fmaxf(sqrtf(s), kNormEps);
    float q = 0.f;
    for (int c = 0; c < C; ++c) {
        const float v = xb[(size_t)c * N] / den;
        xo[(size_t)c * N] = v;
        q = fmaf(v, v, q);
    }
    sq[(size_t)b * N + n] = q;
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
    __device__ __forceinline__ void push_ascending(float v, int vi) {
        bool before_hi = v < d[KB - 1];
#pragma unroll
        for (int j = KB - 1; j >= 1; --j) {
            const bool before_lo = v < d[j - 1];
            d[j] = before_lo ? d[j - 1] : (before_hi ? v : d[j]);
            i[j] = before_lo ? i[j - 1] : (before_hi ? vi : i[j]);
            before_hi = before_lo;
        }
        d[0] = before_hi ? v : d[0];
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

template <int KB, int TILES>
__global__ __launch_bounds__(512) void knn_fused_kernel(
    const float* __restrict__ xn, const float* __restrict__ yn,
    const float* __restrict__ xs, const float* __restrict__ ys,
    const float* __restrict__ relpos, int32_t* __restrict__ out,
    int C, int N, int M, int K) {
    constexpr int KS = 32;          // channels per LDS slab
    constexpr int TM = 32 * TILES;  // candidates per chunk
    extern __shared__ float lds[];
    const int nw = blockDim.x >> 6;
    const int QW = nw * 32;
    float* ldsA = lds;            // [KS][TM]  candidates
    float* ldsB = lds + KS * TM;  // [KS][QW]  queries
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lq = lane & 31, h = lane >> 5;
    const int b = blockIdx.y;
    const int n0 = blockIdx.x * QW;
    const int n = n0 + wave * 32 + lq;
    const bool nvalid = n < N;
    const float* xb = xn + (size_t)b * C * N;
    const float* yb = yn + (size_t)b * C * M;
    const float* ysb = ys + (size_t)b * M;
    const float xsv = nvalid ? xs[(size_t)b * N + n] : 0.f;
    const float* rp_row = (relpos != nullptr && nvalid) ? relpos + (size_t)n * M : nullptr;

    TopK<KB> top;
    top.init();

    for (int mc0 = 0; mc0 < M; mc0 += TM) {
        f32x16 acc[TILES];
#pragma unroll
        for (int t = 0; t < TILES; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

        for (int c0 = 0; c0 < C; c0 += KS) {
            __syncthreads();  // previous slab fully consumed
            for (int c = wave; c < KS; c += nw) {
                const int cc = c0 + c;
                const bool cvalid = cc < C;
                for (int col = lane; col < TM; col += 64) {
                    const int m = mc0 + col;
                    ldsA[c * TM + col] = (cvalid && m < M) ? yb[(size_t)cc * M + m] : 0.f;
                }
                for (int col = lane; col < QW; col += 64) {
                    const int q = n0 + col;
                    ldsB[c * QW + col] = (cvalid && q < N) ? xb[(size_t)cc * N + q] : 0.f;
                }
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
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, bq, acc[t], 0, 0, 0);
                }
            }
        }

        // epilogue: distances of this chunk -> running top-K (ascending m per lane)
#pragma unroll
        for (int t = 0; t < TILES; ++t) {
            f32x16 v = acc[t];
            for (int g = 0; g < 4; ++g) {
                const int mbase = mc0 + t * 32 + 8 * g + 4 * h;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = mbase + r;
                    float dist = INFINITY;
                    if (nvalid && m < M) {
                        dist = (xsv + (-2.0f * v[r])) + ysb[m];
                        if (rp_row != nullptr) dist = dist + rp_row[m];
                    }
                    if (__any(dist < top.d[KB - 1])) top.push_ascending(dist, m);
                }
                // rotate the next 4 accumulator rows into v[0..3]
                v = __builtin_shufflevector(v, v, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 0, 1, 2, 3);
            }
        }
    }

    // merge the two half-waves' lists into the h == 0 lanes.  Each round the h == 1 lanes hand
    // over the head of their (ascending) list and pop it; they only ever see (+inf, sentinel)
    // pushes themselves, which leave a list untouched.  Kept as a rolled loop: code size O(KB).
#pragma unroll 1
    for (int round = 0; round < KB; ++round) {
        float pd = __shfl_xor(top.d[0], 32);
        int pi = __shfl_xor(top.i[0], 32);
        if (h != 0) {
            pd = INFINITY;
            pi = kSentinelIdx;
#pragma unroll
            for (int j = 0; j + 1 < KB; ++j) { top.d[j] = top.d[j + 1]; top.i[j] = top.i[j + 1]; }
            top.d[KB - 1] = INFINITY;
            top.i[KB - 1] = kSentinelIdx;
        }
        const bool enters = (pd < top.d[KB - 1]) || (pd == top.d[KB - 1] && pi < top.i[KB - 1]);
        if (!__any(enters)) break;  // partner entries only grow from here on
        top.push_any(pd, pi);
    }
    if (nvalid && h == 0) {
        int32_t* o = out + ((size_t)b * N + n) * K;
#pragma unroll
        for (int j = 0; j < KB; ++j)
            if (j < K) o[j] = top.i[j];
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
            const unsigned long long key = knn_key(drow[m], m);
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
    size_t xn, xs, yn, ys, dist, total;
};

static int resolve_algo(int algo, int K) {
    if (algo == NEXTOU_KNN_AUTO) return K <= 32 ? NEXTOU_KNN_FUSED : NEXTOU_KNN_NAIVE;
    return algo;
}

static KnnWorkspace knn_layout(int B, int C, int N, int M, int has_y, int algo) {
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
    }
    w.total = off;
    return w;
}

static int launch_prep(const float* x, float* xn, float* sq, int B, int C, int N, bool normalize,
                       hipStream_t s) {
    // reads x twice (second pass L2-hot: counted once), writes xn and the norms
    ProfScope prof(s, kBoundHbm, 4.0 * B * (double)N * ((normalize ? 2.0 : 1.0) * C + 1), "knn_prep_kernel[B%d C%d N%d]",
                   B, C, N);
    if (normalize)
        hipLaunchKernelGGL(knn_prep_kernel<true>, dim3(cdiv(N, 256), B), dim3(256), 0, s, x, xn, sq, C, N);
    else
        hipLaunchKernelGGL(knn_prep_kernel<false>, dim3(cdiv(N, 256), B), dim3(256), 0, s, x, xn, sq, C, N);
    return check_launch("knn_prep_kernel");
}

template <int KB, int TILES>
static int launch_fused(const float* xn, const float* yn, const float* xs, const float* ys,
                        const float* relpos, int32_t* out, int B, int C, int N, int M, int K,
                        int nw, hipStream_t s) {
    const int QW = 32 * nw;
    const size_t lds = (size_t)32 * (32 * TILES + QW) * sizeof(float);
    dim3 grid(cdiv(N, QW), B);
    // algorithmic work of the distance contraction: 2*B*N*M*C flops
    ProfScope prof(s, kBoundMfma, 2.0 * B * (double)N * M * C, "knn_fused_kernel<%d,%d>[B%d C%d N%d M%d K%d]",
                   KB, TILES, B, C, N, M, K);
    hipLaunchKernelGGL((knn_fused_kernel<KB, TILES>), grid, dim3(64 * nw), lds, s, xn, yn, xs, ys,
                       relpos, out, C, N, M, K);
    return check_launch("knn_fused_kernel");
}

template <int KB>
static int launch_fused_tiles(int tiles, const float* xn, const float* yn, const float* xs,
                              const float* ys, const float* relpos, int32_t* out, int B, int C,
                              int N, int M, int K, int nw, hipStream_t s) {
    if (tiles == 2) return launch_fused<KB, 2>(xn, yn, xs, ys, relpos, out, B, C, N, M, K, nw, s);
    return launch_fused<KB, 6>(xn, yn, xs, ys, relpos, out, B, C, N, M, K, nw, s);
}

// waves per workgroup: cover N with 32-query waves, prefer >= 512 workgroups in the grid.
static int pick_waves(int B, int N) {
    const int need = cdiv(N, 32);
    if (need <= 6) return need < 1 ? 1 : need;
    int nw = 6;
    while (nw > 2 && (long long)cdiv(N, 32 * nw) * B < 512) nw -= 2;
    return nw;
}

// candidate tiles per chunk: 6 tiles (192 wide) unless a 64-wide chunk wastes >10 % fewer MFMAs on
// padding (M = 168, 384, 1344, 3072 of cfg 2 / cfg 5 are all multiples or near-multiples of 192).
static int pick_tiles(int M, int K) {
    (void)K;
    const int wide = 192;
    const long long ww = (long long)cdiv(M, wide) * wide, w2 = (long long)cdiv(M, 64) * 64;
    return (w2 * 10 < ww * 9) ? 2 : 6;
}

}  // namespace nextou

using namespace nextou;

extern "C" size_t nextou_knn_workspace_bytes(int B, int C, int N, int M, int K, int has_y, int algo) {
    if (B <= 0 || C <= 0 || N <= 0 || M <= 0) return 0;
    return knn_layout(B, C, N, M, has_y, resolve_algo(algo, K)).total;
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
    const KnnWorkspace w = knn_layout(B, C, N, M, has_y, algo);
    if (workspace_bytes < w.total)
        return fail(NEXTOU_ENOSPACE, "knn_graph: workspace %zu < required %zu", workspace_bytes, w.total);
    hipStream_t s = (hipStream_t)stream;
    char* base = (char*)workspace;
    float* xn = (float*)(base + w.xn);
    float* xs = (float*)(base + w.xs);
    float* yn = (float*)(base + w.yn);
    float* ys = (float*)(base + w.ys);

    if (int e = launch_prep(x, xn, xs, B, C, N, normalize != 0, s)) return e;
    if (has_y) {
        if (int e = launch_prep(y, yn, ys, B, C, M, normalize != 0, s)) return e;
    }
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

    const int nw = pick_waves(B, N);
    const int tiles = pick_tiles(M, K);
    if (K <= 8) return launch_fused_tiles<8>(tiles, xn, yn, xs, ys, relpos, nn_idx, B, C, N, M, K, nw, s);
    if (K <= 16) return launch_fused_tiles<16>(tiles, xn, yn, xs, ys, relpos, nn_idx, B, C, N, M, K, nw, s);
    return launch_fused_tiles<32>(tiles, xn, yn, xs, ys, relpos, nn_idx, B, C, N, M, K, nw, s);
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
