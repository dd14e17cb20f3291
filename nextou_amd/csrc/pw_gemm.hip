// K7 — point-wise (1x1[x1]) convolutions of the graph stages on channels-last rows (gfx950, f32 MFMA).
//
// The Grapher / FFN blocks of the path are chains of point-wise convolutions (reference
// NexToU_Encoder_Decoder.py:368-390 FFN fc1/fc2, :710-720 / :833-842 the graphers' fc1 / fc2, torch_nn.py:66-92 the
// MRConv's grouped BasicConv) and the decoder ends in the 1x1 segmentation heads (:298).  On a channels-last volume
// every one of them is a GEMM over the (points, channels) matrix that is already in memory:
//
//   pw_rows    y[p, n]  = sum_k x[p, k] * w[n, k] (+ bias[n])        forward; data gradient with w transposed
//   pw_wgrad   dw[n, k] = sum_p gy[p, n] * x[p, k]                   weight gradient: reduction over ALL points
//
// Channel counts are 44 ... 1296 (multiples of 4, rarely of 32) and the point count is 336 ... 5.5 M; CK's 3-D kernels
// run these at 20-50 % (forward, data gradient) and 2-24 % (weight gradient) of the f32 MFMA peak
// (profiles/r02_conv_evidence_padding_ab.md), MIOpen's 2-D assembly kernels on the depth-flat view at 45-60 % — which is
// where these kernels are as well (profiles/r02_pw_gemm.md: level stand-alone, not yet ahead in the step, hence opt-in via
// NEXTOU_PW_GEMM=1).  Both kernels here use v_mfma_f32_16x16x4_f32 (exact f32, bitwise
// an fmaf chain; 16-granular tiles waste <= 9 % on these channel counts), keep the channel dimension in the lane's
// four accumulator registers so that results leave as 16-byte stores, and map workgroups to XCDs so that the
// workgroups which share an input tile share an L2.
//
// pw_rows:  workgroup = 4 waves stacked over points, each TM point tiles x all TN channel tiles of the workgroup; both
// operands are K-contiguous in memory, so one ds_read_b128 per lane feeds FOUR MFMAs (lane group g = lane >> 4 takes
// k = 16 r + 4 g + {0..3}; which k a lane group holds is free as long as A and B agree).  LDS rows are 20 floats
// (4 * odd): the 16 lanes of one ds_read_b128 phase hit 16 distinct 4-bank groups.
//
// pw_wgrad: the reduction index is the memory row, so the MFMA operands are rows of the tiles as they lie in memory:
// lane (c = lane & 15, g = lane >> 4) reads tile[row + g][col + c] — ds_read_b32, conflict-free with a row stride
// = 16 (mod 32) floats.  Split over points (deterministic: partial tiles to a workspace, summed in a fixed order by
// pw_wgrad_reduce_kernel).
#include "common.h"
#include <cstdlib>

namespace nextou {
namespace {

using f32x4 = __attribute__((ext_vector_type(4))) float;

// NEXTOU_PW_ABLATE (experiments, tools/pw_ablate.sh — never in the product build): 1 = no MFMAs (LDS reads kept), 2 = global loads and LDS
// stores only for the first stage of a tile, 4 = no result stores
#ifndef NEXTOU_PW_ABLATE
#define NEXTOU_PW_ABLATE 0
#endif
#ifndef NEXTOU_PW_STAGGER
#define NEXTOU_PW_STAGGER 0          // x 8128 cycles of start delay for every second resident generation of workgroups (experiment)
#endif
constexpr int kPwKC = 16;          // k per LDS stage of pw_rows: one MFMA round (4 lane groups x float4)
constexpr int kPwLd = kPwKC + 4;   // 20 floats = 80 B = 4 * odd
constexpr int kXcds = 8;

// blockIdx.x -> work item such that consecutive work items run on ONE XCD (hardware deals workgroups round-robin
// over the 8 XCDs): the items that share an input tile are neighbours in item order, so they share an L2.
__device__ __forceinline__ int xcd_item(int block, int grid) { return (block % kXcds) * (grid / kXcds) + block / kXcds; }

__device__ __forceinline__ float4 ld4(const float* p) { return *reinterpret_cast<const float4*>(p); }

// v or zeros, component-wise: `ok ? a : b` on two float4 lvalues is a select of ADDRESSES, which parks the register arrays
// the operands live in on the stack (scratch)
__device__ __forceinline__ float4 keep_if(bool ok, float4 v) {
    return make_float4(ok ? v.x : 0.f, ok ? v.y : 0.f, ok ? v.z : 0.f, ok ? v.w : 0.f);
}

// What a fused launch of pw_rows adds to the plain GEMM (the Grapher / FFN blocks' norm + activation inside the
// 1x1 convolutions; reference torch_nn.py:84-90, NexToU_Encoder_Decoder.py:384-390, :710-720, :833-842):
//   PRO = 1  operand prologue: the x tile is normalised + activated on its way into LDS,
//            a[p, k] = leaky(fmaf(x[p, k], scale[k], shift[k]), slope) with scale = w * invstd, shift = fmaf(-mean, scale, b) —
//            bit for bit K6's apply, so the activated tensor never exists in HBM;
//   EPI = 1  statistics epilogue: per output channel (sum y, sum y^2) of the tile's points -> `partial` (K6's finalize input);
//   EPI = 2  gradient-statistics epilogue for a data-gradient GEMM whose output is d(activated) : with h = the forward's
//            pre-norm tensor at the output's coordinates, z = h * scale + shift, dz = y * (z > 0 ? 1 : slope),
//            xhat = (h - mean) * invstd: (sum dz, sum dz * xhat) -> `partial` (K6's backward-finalize input).  y itself is
//            written unchanged (K6's backward apply recomputes dz from it).
// Partials: a wave reduces its 32 points in fp32 (fixed DPP tree), the workgroup's four waves are combined in float64 in wave
// order, one double2 per (channel, point tile) goes to partial[channel * tiles + tile] — K6's layout; fixed order throughout
// (bit-reproducible).
struct PwFuse {
    const float *pro_scale, *pro_shift;                     // prologue affine over the K input channels (per group: index g*K + k):
    float pro_slope;                                        //   scale = weight * invstd, shift = bias - mean * scale (nextou_norm_finalize)
    double2* partial;                                       // [groups * N][tiles]
    const float* h;                                         // EPI = 2: (P, groups * N) rows, stride ldh
    long ldh;
    const float *epi_w, *epi_b, *epi_mean, *epi_invstd;     // EPI = 2: the norm whose backward statistics are collected
    float epi_slope;
    // up_cout > 0 (pw_rows_kernel, EPI = 0 only): the product is a transposed convolution with kernel == stride (every power-of-two
    // stride up to 4) — column n = t * up_cout + co of input point p goes to channel co of the OUTPUT point the tap t of p lands on,
    // row pitch ldy (the concatenation buffer: the up-sampled half is written where torch.cat would copy it); up_bias[co] added
    int up_cout;
    UpShuffle up;
    const float* up_bias;
};

// sum over the 16 lanes of a DPP row (lanes that share lane >> 4), result in every lane; fixed tree
__device__ __forceinline__ float row16_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));    // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));    // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true));   // row_mirror
    return v;
}

__device__ __forceinline__ float leaky_f(float z, float slope) { return z > 0.f ? z : z * slope; }

// ------------------------------------------------------------------------------------------------------------
// y[p, n] = sum_k x[p, k] w[n, k] + bias[n]
// ------------------------------------------------------------------------------------------------------------
template <int TM, int TN, int PRO, int EPI>
__global__ __launch_bounds__(256, 2) void pw_rows_kernel(const float* __restrict__ X, const float* __restrict__ Wt,
                                                      const float* __restrict__ bias, float* __restrict__ Y, int P, int N, int K,
                                                      long ldx, long ldw, long ldy, int nb_n, int items, int vec_store, PwFuse fz) {
    constexpr int BM = 64 * TM, BN = 16 * TN;
    constexpr int XV = BM * (kPwKC / 4) / 256;                // float4 per thread per stage, X tile
    constexpr int WV = (BN * (kPwKC / 4) + 255) / 256;        // ... W tile (last one predicated)
    extern __shared__ float4 pw_smem4[];
    float* Xs = reinterpret_cast<float*>(pw_smem4);           // [2][BM][kPwLd]
    float* Ws = Xs + 2 * BM * kPwLd;                          // [2][BN][kPwLd]
    float2* red = reinterpret_cast<float2*>(Ws + 2 * BN * kPwLd);   // EPI != 0: [4 waves][BN] (sum, sum') of a wave's 32 points

    // persistent workgroups: XCD x owns the contiguous item range [x * per_xcd, (x + 1) * per_xcd) — items are ordered
    // channel block fastest, so the workgroups resident on one XCD at any time share x tiles (and w) in its L2 — and
    // slot j of the XCD walks it with stride slots.  A workgroup that finishes a tile has the next tile's first two stages
    // already in flight while it writes its results: neither the load latency nor the store drain leaves the matrix
    // cores idle (with one tile per workgroup and K = 132 they were busy 53 % of the time, profiles/r02_pw_gemm.md).
    const int per_xcd = (items + kXcds - 1) / kXcds;
    const int xcd = blockIdx.x % kXcds, slots = gridDim.x / kXcds;
    int walk = blockIdx.x / kXcds;
    auto item_at = [&](int j) { const int it = xcd * per_xcd + j; return (j < per_xcd && it < items) ? it : -1; };
    const int g = blockIdx.y;
    X += (long)g * K;
    Wt += (long)g * N * ldw;
    Y += (long)g * N;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;

    // register staging, prefetch distance 2: while stage s is multiplied out of LDS, stage s + 1 sits in (or is on its way
    // to) one register set and the loads of stage s + 2 are issued into the other — one stage of MFMAs (~1.2 us) does not
    // cover the memory latency under load, two do
    float4 xr[2][XV], wr[2][WV];
    // PRO: the thread's k offset inside a stage is the same for all its X pieces (f % 4 == tid % 4 because 256 % 4 == 0), so one
    // (scale, shift) float4 pair per stage and register set covers them; they are loaded with the stage (prefetched alike)
    float4 psc[2], psh[2];
    auto load_pro = [&](int k0, float4& sc, float4& sh) __attribute__((always_inline)) {
        if constexpr (PRO == 1) {
            const long k = (long)g * K + min(k0 + (tid & 3) * 4, K - 4);
            sc = ld4(fz.pro_scale + k);
            sh = ld4(fz.pro_shift + k);
        }
    };
    // loads are UNCONDITIONAL and their results are not touched until the LDS store one stage later (addresses clamped into the
    // tensor; out-of-range pieces are zeroed by a select in store_stage): a predicated load is a branch, and a select right
    // behind the load is a use — either way the compiler waits (s_waitcnt vmcnt(0)) for loads it has just issued, which
    // defeats the prefetch.  Like this it waits with vmcnt(n), n = the loads of the younger stage still in flight.
    auto load_stage = [&](int p0, int n0, int k0, float4 (&xq)[XV], float4 (&wq)[WV], float4& sc, float4& sh) __attribute__((always_inline)) {
        load_pro(k0, sc, sh);
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + i * 256, row = f / (kPwKC / 4), c4 = f % (kPwKC / 4);
            xq[i] = ld4(X + (long)min(p0 + row, P - 1) * ldx + min(k0 + c4 * 4, K - 4));
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int f = tid + i * 256, row = f / (kPwKC / 4), c4 = f % (kPwKC / 4);
            wq[i] = ld4(Wt + (long)min(n0 + row, N - 1) * ldw + min(k0 + c4 * 4, K - 4));
        }
    };
    auto store_stage = [&](int buf, int p0, int n0, int k0, const float4 (&xq)[XV], const float4 (&wq)[WV], const float4& sc,
                           const float4& sh) __attribute__((always_inline)) {
        float* xs = Xs + buf * BM * kPwLd;
        float* ws = Ws + buf * BN * kPwLd;
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + i * 256, row = f / (kPwKC / 4), c4 = f % (kPwKC / 4);
            const bool ok = p0 + row < P && k0 + c4 * 4 < K;
            float4 v = xq[i];
            if constexpr (PRO == 1)        // K6's apply, bit for bit: leaky(fmaf(x, scale, shift)); padding stays exactly zero (keep_if below)
                v = make_float4(leaky_f(fmaf(v.x, sc.x, sh.x), fz.pro_slope), leaky_f(fmaf(v.y, sc.y, sh.y), fz.pro_slope),
                                leaky_f(fmaf(v.z, sc.z, sh.z), fz.pro_slope), leaky_f(fmaf(v.w, sc.w, sh.w), fz.pro_slope));
            *reinterpret_cast<float4*>(xs + row * kPwLd + c4 * 4) = keep_if(ok, v);
        }
#pragma unroll
        for (int i = 0; i < WV; ++i) {
            const int f = tid + i * 256, row = f / (kPwKC / 4), c4 = f % (kPwKC / 4);
            const bool ok = n0 + row < N && k0 + c4 * 4 < K;
            if (row < BN) *reinterpret_cast<float4*>(ws + row * kPwLd + c4 * 4) = keep_if(ok, wq[i]);
        }
    };

    f32x4 acc[TM][TN];
    // (a K that is not a multiple of 16 still runs all four k sub-steps of its last stage: lane group kg holds k = 16 r + 4 kg +
    // {0..3}, so a 4-channel tail sits in lane group 0 of EVERY sub-step; pw_rows_sw_kernel gives the tail its own lane mapping)
    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* xs = Xs + buf * BM * kPwLd + (wave * TM * 16 + r16) * kPwLd + kg * 4;
        const float* ws = Ws + buf * BN * kPwLd + r16 * kPwLd + kg * 4;
        float4 b[TM];
#pragma unroll
        for (int i = 0; i < TM; ++i) b[i] = ld4(xs + i * 16 * kPwLd);
        // the four k sub-steps of one accumulator are a dependent chain (40-cycle latency, 32-cycle issue): run them over
        // 2 channel tiles x TM point tiles at a time, so that a chain's next link is >= 4 MFMAs away; the weight fragments of
        // the NEXT pair are read from LDS before the MFMAs of the current one (the compiler, left alone, reads them right in
        // front of their first use: ~100 cycles of LDS latency exposed per 16 MFMAs)
        float4 a0 = ld4(ws), a1 = TN > 1 ? ld4(ws + 16 * kPwLd) : a0;
#pragma unroll
        for (int j = 0; j < TN; j += 2) {
            float4 n0f = a0, n1f = a1;
            if (j + 2 < TN) n0f = ld4(ws + (j + 2) * 16 * kPwLd);
            if (j + 3 < TN) n1f = ld4(ws + (j + 3) * 16 * kPwLd);
            __builtin_amdgcn_sched_barrier(0);      // the scheduler would sink the two reads to just before their first use
#if NEXTOU_PW_ABLATE & 1
#define NEXTOU_PW_STEP(c) _Pragma("unroll") for (int i = 0; i < TM; ++i) acc[i][j][0] += a0.c + a1.c + b[i].c;
#else
#define NEXTOU_PW_STEP(c)                                                                                                   \
    _Pragma("unroll") for (int i = 0; i < TM; ++i) {                                                                       \
        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0.c, b[i].c, acc[i][j], 0, 0, 0);                               \
        if (j + 1 < TN) acc[i][j + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1.c, b[i].c, acc[i][j + 1], 0, 0, 0);       \
    }
#endif
            NEXTOU_PW_STEP(x)
            NEXTOU_PW_STEP(y)
            NEXTOU_PW_STEP(z)
            NEXTOU_PW_STEP(w)
#undef NEXTOU_PW_STEP
            a0 = n0f;
            a1 = n1f;
        }
    };

    auto epilogue = [&](int p0, int n0) __attribute__((always_inline)) {
        if (EPI == 0 && fz.up_cout > 0) {
            // transposed-convolution store: rows of the output volume (see PwFuse).  The input point's coordinates once per point tile,
            // the tap's offset per channel tile with shifts (strides are 1, 2 or 4)
            const UpShuffle& u = fz.up;
            const int Wi = u.W2 / u.sw, Hi = u.H2 / u.sh, Di = u.D2 / u.sd;
            const int lw = u.sw >> 1, lh = u.sh >> 1;                        // log2 of 1 / 2 / 4 is 0 / 1 / 2 = s >> 1
            long long rowbase[TM];
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int p = min(p0 + (wave * TM + i) * 16 + r16, P - 1);
                const int w_ = p % Wi, q1 = p / Wi;
                const int h_ = q1 % Hi, q2 = q1 / Hi;
                const int d_ = q2 % Di, b_ = q2 / Di;
                rowbase[i] = (((long long)b_ * u.D2 + (long long)d_ * u.sd) * u.H2 + (long long)h_ * u.sh) * u.W2 + (long long)w_ * u.sw;
            }
            const float inv_cout = 1.0f / (float)fz.up_cout;
#pragma unroll
            for (int j = 0; j < TN; ++j) {
                const int n = n0 + j * 16 + kg * 4;
                if (n >= N) continue;
                const int t = (int)(((float)n + 0.5f) * inv_cout);           // exact for n < 2^20 (n, up_cout integers; four channels never straddle a tap)
                const int co = n - t * fz.up_cout;
                const int tw = t & (u.sw - 1), th = (t >> lw) & (u.sh - 1), td = t >> (lw + lh);
                const long long toff = ((long long)td * u.H2 + th) * u.W2 + tw;
                const float4 bv = fz.up_bias ? ld4(fz.up_bias + co) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const int p = p0 + (wave * TM + i) * 16 + r16;
                    if (p >= P) continue;
                    *reinterpret_cast<float4*>(Y + (rowbase[i] + toff) * ldy + co) =
                        make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
                }
            }
            return;
        }
        // D tile: column = lane & 15 = point, row = 4 (lane >> 4) + reg = channel -> one 16-byte store per tile and lane
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int n = n0 + j * 16 + kg * 4;
            if (n >= N) continue;
            float4 bv = make_float4(0.f, 0.f, 0.f, 0.f);
            if (bias) {
                const float* bp = bias + (long)g * N + n;
                bv.x = bp[0];
                if (n + 1 < N) bv.y = bp[1];
                if (n + 2 < N) bv.z = bp[2];
                if (n + 3 < N) bv.w = bp[3];
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) {
                const int p = p0 + (wave * TM + i) * 16 + r16;
                if (p >= P) continue;
#if NEXTOU_PW_ABLATE & 4
                if (acc[i][j][0] == acc[i][j][0]) continue;      // stores only for NaNs: keeps the accumulators alive
#endif
                float* yp = Y + (long)p * ldy + n;
                const float4 v = make_float4(acc[i][j][0] + bv.x, acc[i][j][1] + bv.y, acc[i][j][2] + bv.z, acc[i][j][3] + bv.w);
                if (vec_store && n + 3 < N) {
                    *reinterpret_cast<float4*>(yp) = v;
                } else {
                    yp[0] = v.x;
                    if (n + 1 < N) yp[1] = v.y;
                    if (n + 2 < N) yp[2] = v.z;
                    if (n + 3 < N) yp[3] = v.w;
                }
            }
        }
        if constexpr (EPI != 0) {
            // per-channel partial sums of this tile (fused launches: bias == NULL, N % 4 == 0; rows past P are exact zeros and
            // add nothing).  A lane holds 4 channels x TM points per channel tile, the 16 lanes of its DPP row the other points.
            constexpr int JB = EPI == 2 ? 3 : 1;       // EPI 2: h loads of JB channel tiles are issued together (latency batches)
#pragma unroll
            for (int j0 = 0; j0 < TN; j0 += JB) {
                float4 hq[JB][TM];
                if constexpr (EPI == 2) {
#pragma unroll
                    for (int jj = 0; jj < JB; ++jj)
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const int n = min(n0 + (j0 + jj) * 16 + kg * 4, N - 4);
                            const int p = min(p0 + (wave * TM + i) * 16 + r16, P - 1);
                            hq[jj][i] = ld4(fz.h + (long)p * fz.ldh + (long)g * N + n);
                        }
                }
#pragma unroll
                for (int jj = 0; jj < JB; ++jj) {
                    const int j = j0 + jj;
                    if (j >= TN) break;
                    const int n = n0 + j * 16 + kg * 4;
                    if (n >= N) continue;                       // uniform over the DPP row (same kg)
                    float s[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
                    if constexpr (EPI == 1) {
#pragma unroll
                        for (int i = 0; i < TM; ++i)
#pragma unroll
                            for (int r = 0; r < 4; ++r) { const float v = acc[i][j][r]; s[r] += v; q[r] = fmaf(v, v, q[r]); }
                    } else {
                        const long c = (long)g * N + n;
                        const float4 w = ld4(fz.epi_w + c), b = ld4(fz.epi_b + c), m = ld4(fz.epi_mean + c), is = ld4(fz.epi_invstd + c);
                        const float wv[4] = {w.x, w.y, w.z, w.w}, bv4[4] = {b.x, b.y, b.z, b.w}, mv[4] = {m.x, m.y, m.z, m.w},
                                    iv[4] = {is.x, is.y, is.z, is.w};
#pragma unroll
                        for (int i = 0; i < TM; ++i) {
                            const float hv[4] = {hq[jj][i].x, hq[jj][i].y, hq[jj][i].z, hq[jj][i].w};
#pragma unroll
                            for (int r = 0; r < 4; ++r) {           // K6's backward reduce, term for term
                                const float scale = wv[r] * iv[r], shift = fmaf(-mv[r], scale, bv4[r]);
                                const float z = fmaf(hv[r], scale, shift);
                                const float gq = acc[i][j][r];
                                const float dz = z > 0.f ? gq : gq * fz.epi_slope;
                                const float xh = (hv[r] - mv[r]) * iv[r];
                                s[r] += dz;
                                q[r] = fmaf(dz, xh, q[r]);
                            }
                        }
                    }
#pragma unroll
                    for (int r = 0; r < 4; ++r) { s[r] = row16_sum(s[r]); q[r] = row16_sum(q[r]); }
                    if (r16 == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) red[wave * BN + j * 16 + kg * 4 + r] = make_float2(s[r], q[r]);
                    }
                }
            }
            __syncthreads();
            const int tiles_p = items / nb_n, tile_p = p0 / BM;
            for (int c = tid; c < BN; c += 256) {
                if (n0 + c >= N) continue;
                const double S = (((double)red[c].x + (double)red[BN + c].x) + (double)red[2 * BN + c].x) + (double)red[3 * BN + c].x;
                const double Q = (((double)red[c].y + (double)red[BN + c].y) + (double)red[2 * BN + c].y) + (double)red[3 * BN + c].y;
                fz.partial[((long)g * N + n0 + c) * tiles_p + tile_p] = make_double2(S, Q);
            }
            // `red` is next written in the next tile's epilogue, behind the barriers of its main loop
        }
    };

    const int stages = (K + kPwKC - 1) / kPwKC;
    int item = item_at(walk);
    int p0 = 0, n0 = 0;
#if NEXTOU_PW_STAGGER
    // the workgroups that share a CU start together and stay in lockstep: they all multiply, then they all store — the matrix
    // pipes idle while the stores drain and HBM idles while they multiply (ablation: 223 us compute-only + 66 us stores + ~40 us
    // loads = the 322 us measured, no overlap).  Delaying the second resident generation by about half a tile period puts its
    // store phase under the first generation's multiply phase.
    if ((blockIdx.x / (kXcds * 32)) & 1)
        for (int i = 0; i < NEXTOU_PW_STAGGER; ++i) __builtin_amdgcn_s_sleep(127);
#endif
    if (item >= 0) {
        p0 = (item / nb_n) * BM;
        n0 = (item % nb_n) * BN;
        load_stage(p0, n0, 0, xr[0], wr[0], psc[0], psh[0]);
        load_stage(p0, n0, stages > 1 ? kPwKC : 0, xr[1], wr[1], psc[1], psh[1]);
    }
    while (item >= 0) {
        __syncthreads();                       // the previous tile's last LDS reads are done
        store_stage(0, p0, n0, 0, xr[0], wr[0], psc[0], psh[0]);
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
        // stages in pairs (LDS buffer 0, then 1) so that the register sets are indexed statically; no exit inside the pair —
        // an exit there makes the compiler shuttle every accumulator between AGPRs and VGPRs once per iteration
        for (int s = 0; s + 1 < stages; s += 2) {
#if NEXTOU_PW_ABLATE & 2
            compute(0);
            __syncthreads();
            compute(0);
            __syncthreads();
#else
            load_stage(p0, n0, min(s + 2, stages - 1) * kPwKC, xr[0], wr[0], psc[0], psh[0]);     // past the end: a re-read nobody stores
            compute(0);
            store_stage(1, p0, n0, (s + 1) * kPwKC, xr[1], wr[1], psc[1], psh[1]);
            __syncthreads();
            load_stage(p0, n0, min(s + 3, stages - 1) * kPwKC, xr[1], wr[1], psc[1], psh[1]);
            compute(1);
            store_stage(0, p0, n0, (s + 2) * kPwKC, xr[0], wr[0], psc[0], psh[0]);     // past the end: zeros nobody reads (no branch: keeps vmcnt exact)
            __syncthreads();
#endif
        }
        if (stages & 1) compute(0);
        walk += slots;
        const int next = item_at(walk);
        const int np0 = next >= 0 ? (next / nb_n) * BM : p0, nn0 = next >= 0 ? (next % nb_n) * BN : n0;
        load_stage(np0, nn0, 0, xr[0], wr[0], psc[0], psh[0]);                                    // no next tile: a re-read of this one
        load_stage(np0, nn0, stages > 1 ? kPwKC : 0, xr[1], wr[1], psc[1], psh[1]);
        epilogue(p0, n0);
        item = next;
        p0 = np0;
        n0 = nn0;
    }
}

// ------------------------------------------------------------------------------------------------------------
// pw_rows, STATIONARY WEIGHTS (round 3).  y[p, n] = sum_k A[p, k] w[n, k] for the stage-2 shapes of the path, where the point count
// is long (172 032) and K short (132 ... 528): each wave keeps the weights of its TNW x 16 output channels in REGISTERS for the whole
// launch and the workgroup — ceil(N / (16 TNW)) waves, all N channels — streams 64-point tiles of x through a double-buffered LDS
// slab.  Against pw_rows_kernel (weights re-staged through LDS for every tile, K-staged loop with two barriers per 16 k):
//   * x is read from HBM / MALL ONCE (not once per channel block), the weights never touch LDS, one barrier per 64-point slab;
//   * the loop body is ONE static instruction sequence per tile (the k loop is unrolled over the register-resident weights), so the
//     in-order vmcnt counter the compiler waits on is exact: the result stores of tile t stay in flight under the multiply of tile
//     t + 1 (in pw_rows_kernel the loop-top wait merged the entry path with the back edge and drained the stores before the next
//     tile could start: its multiply, store and load phases added up — profiles/r02_pw_gemm.md ablation, r03_pw_rows_sw.md);
//   * statistics epilogue without cross-wave traffic: a channel belongs to exactly one wave.
// Same MFMA (16x16x4 f32), same k order inside a 16-k chunk (lane group kg holds k = 16 c + 4 kg + {0..3}), k chunks ascending: for
// K = 132 the results are bit-identical to pw_rows_kernel's; for K = 264 / 528 the slabs cut the chain at multiples of 132
// instead of 16 (another order of the same fp32 fma chain).
// Shape-specialised: <TNW, NW waves, NCH full 16-k chunks per slab, TAIL extra 4-k sub-steps (one slab only), SLABS>,
// K = SLABS * 16 * NCH + 4 * TAIL; plan_rows_sw() knows the instantiated shapes, everything else takes pw_rows_kernel.
// ------------------------------------------------------------------------------------------------------------
// Slabs are 132 channels wide — every K of the path's stage-2 point-wise convolutions is a multiple of 132 (132, 264, 528) —
// i.e. eight 16-k chunks and one 4-k tail sub-step: K = SLABS * 132.
constexpr int kSwNch = 8, kSwSk = 16 * kSwNch + 4, kSwSk4 = kSwSk / 4;     // 132 floats = 33 16-byte slots (odd: conflict-light ds_read_b128)

// TNW channel tiles per wave, NW waves; PASSES: the 64-point tile is multiplied in PASSES sequential slices (accumulators:
// 4 / PASSES point tiles x TNW channel tiles); WSTREAM: the weights of slab s + 1 are re-fetched (L2) into a second register set
// while slab s multiplies — K = 528 would otherwise hold 132 weight registers per lane (3 waves per SIMD allow 168 in all).
template <int TNW, int NW, int SLABS, int WSTREAM, int PRO, int EPI>
__global__ __launch_bounds__(64 * NW) void pw_rows_sw_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y,
                                                           int P, int N, int K, long ldx, long ldy, int tiles64, PwFuse fz) {
    constexpr int NT = 64 * NW, SK = kSwSk, SK4 = kSwSk4, LDK = kSwSk, NCH = kSwNch;
    constexpr int NV = (64 * SK4 + NT - 1) / NT;                           // float4 per thread per slab
    constexpr int ROWS = (NV * NT + SK4 - 1) / SK4;                        // LDS rows incl. the ones only the staging's overhang touches:
                                                                           // every thread loads and stores NV pieces UNCONDITIONALLY (a
                                                                           // predicated piece is sunk into its branch, behind the multiply)
    constexpr int PASSES = SLABS > 1 ? 1 : (TNW >= 3 ? 4 : (TNW == 2 ? 2 : 1));
    constexpr int PTS = 4 / PASSES;
    constexpr int WSETS = WSTREAM ? 2 : SLABS;
    static_assert(SLABS == 1 || TNW == 1, "several slabs keep all four point tiles' accumulators: one channel tile per wave");
    extern __shared__ float4 pw_smem4[];
    float* xs = reinterpret_cast<float*>(pw_smem4);                        // [2][ROWS][LDK], rows 0..63 are the tile
    float* psc = xs + 2 * ROWS * LDK;                                      // PRO: [K] scale, [K] shift
    float* psh = psc + (PRO ? K : 0);
    // EPI: per-channel (sum, sum') of everything this workgroup has multiplied so far, float64, one private slot per channel (a
    // channel belongs to one wave): ONE partial per channel and workgroup leaves the kernel instead of one per tile and pass
    double2* wg_stats = reinterpret_cast<double2*>(psc + (PRO ? 2 * K : 0));         // 16-byte aligned: every piece before it is
    // EPI 2: scale / shift / mean / invstd of the norm whose backward statistics are collected, for this workgroup's channels — read
    // from LDS in the epilogue (as global loads they were 12 exposed L2 round trips per pass: 381 us against 245 without them)
    float* ecoef = reinterpret_cast<float*>(wg_stats + (EPI ? NW * TNW * 16 : 0));     // [4][NW * TNW * 16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;

    // ---- the wave's weights: w[set][j][c] = W[n0 + 16 j + r16][132 s + 16 c + 4 kg + {0..3}], tail w_t[set][j] = W[..][132 s + 128 + kg]
    f32x4 wreg[WSETS][TNW][NCH];
    float wtail[WSETS][TNW];
    const int n_wave = wave * TNW * 16;
    auto load_weights = [&](int set, int sl) __attribute__((always_inline)) {
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = n_wave + j * 16 + r16;
            const float* wp = Wt + (long)min(n, N - 1) * K + sl * SK;
            const bool ok = n < N;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                const float4 v = ld4(wp + 16 * c + 4 * kg);
                wreg[set][j][c] = ok ? f32x4{v.x, v.y, v.z, v.w} : f32x4{0.f, 0.f, 0.f, 0.f};
            }
            wtail[set][j] = ok ? wp[16 * NCH + kg] : 0.f;
        }
    };
    if constexpr (WSTREAM) {
        load_weights(0, 0);
    } else {
#pragma unroll
        for (int sl = 0; sl < SLABS; ++sl) load_weights(sl, sl);
    }
    if constexpr (PRO == 1) {
        for (int k = tid; k < K; k += NT) { psc[k] = fz.pro_scale[k]; psh[k] = fz.pro_shift[k]; }
    }
    if constexpr (EPI != 0) {
        for (int c = tid; c < NW * TNW * 16; c += NT) wg_stats[c] = make_double2(0.0, 0.0);
    }
    if constexpr (EPI == 2) {
        constexpr int NC = NW * TNW * 16;
        for (int c = tid; c < NC; c += NT) {
            const int cc = min(c, N - 1);
            const float is = fz.epi_invstd[cc], m = fz.epi_mean[cc], sc = fz.epi_w[cc] * is;
            ecoef[c] = sc;
            ecoef[NC + c] = fmaf(-m, sc, fz.epi_b[cc]);
            ecoef[2 * NC + c] = m;
            ecoef[3 * NC + c] = is;
        }
    }

    // ---- slab staging: global -> registers (in flight during the multiply) -> LDS
    float4 xr[NV];
    auto load_slab = [&](int tile, int sl) __attribute__((always_inline)) {
        const long p0 = (long)tile * 64;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT, row = f / SK4, c4 = f - row * SK4;
            const long p = min(p0 + min(row, 63), (long)P - 1);
            xr[i] = ld4(X + p * ldx + sl * SK + c4 * 4);
        }
    };
    auto store_slab = [&](int buf, int tile, int sl) __attribute__((always_inline)) {
        float* dst = xs + buf * ROWS * LDK;
        const long p0 = (long)tile * 64;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT, row = f / SK4, c4 = f - row * SK4;
            float4 v = xr[i];
            if constexpr (PRO == 1) {
                const float4 sc = *reinterpret_cast<const float4*>(psc + sl * SK + c4 * 4), sh = *reinterpret_cast<const float4*>(psh + sl * SK + c4 * 4);
                v = make_float4(leaky_f(fmaf(v.x, sc.x, sh.x), fz.pro_slope), leaky_f(fmaf(v.y, sc.y, sh.y), fz.pro_slope),
                                leaky_f(fmaf(v.z, sc.z, sh.z), fz.pro_slope), leaky_f(fmaf(v.w, sc.w, sh.w), fz.pro_slope));
            }
            *reinterpret_cast<float4*>(dst + row * LDK + c4 * 4) = keep_if(p0 + row < P, v);
        }
    };

    f32x4 acc[PTS][TNW];
    auto multiply = [&](int buf, int set, int pass) __attribute__((always_inline)) {
        const float* base = xs + buf * ROWS * LDK + (pass * PTS * 16 + r16) * LDK + 4 * kg;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            float4 b[PTS];
#pragma unroll
            for (int pt = 0; pt < PTS; ++pt) b[pt] = ld4(base + pt * 16 * LDK + 16 * c);
#define NEXTOU_SW_STEP(cmp, idx)                                                                                           \
    _Pragma("unroll") for (int pt = 0; pt < PTS; ++pt) _Pragma("unroll") for (int j = 0; j < TNW; ++j)                    \
        acc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[set][j][c][idx], b[pt].cmp, acc[pt][j], 0, 0, 0);
            NEXTOU_SW_STEP(x, 0)
            NEXTOU_SW_STEP(y, 1)
            NEXTOU_SW_STEP(z, 2)
            NEXTOU_SW_STEP(w, 3)
#undef NEXTOU_SW_STEP
        }
        float bt[PTS];
#pragma unroll
        for (int pt = 0; pt < PTS; ++pt) bt[pt] = base[pt * 16 * LDK + 16 * NCH - 3 * kg];                   // column 128 + kg
#pragma unroll
        for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[pt][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[set][j], bt[pt], acc[pt][j], 0, 0, 0);
    };

    // EPI 2: the h values of a pass's output positions, fetched BEFORE the pass multiplies (in flight under ~100 MFMAs per wave)
    float4 hq[EPI == 2 ? PTS : 1][EPI == 2 ? TNW : 1];
    auto prefetch_h = [&](int tile, int pass) __attribute__((always_inline)) {
        if constexpr (EPI == 2) {
            const long p0 = (long)tile * 64 + pass * PTS * 16;
#pragma unroll
            for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
                for (int j = 0; j < TNW; ++j)
                    hq[pt][j] = ld4(fz.h + min(p0 + pt * 16 + r16, (long)P - 1) * fz.ldh + min(n_wave + j * 16 + kg * 4, N - 4));
        }
    };
    auto epilogue = [&](int tile, int pass) __attribute__((always_inline)) {
        const long p0 = (long)tile * 64 + pass * PTS * 16;
#pragma unroll
        for (int j = 0; j < TNW; ++j) {
            const int n = n_wave + j * 16 + kg * 4;
            if (n >= N) continue;                                           // uniform over the DPP row
#pragma unroll
            for (int pt = 0; pt < PTS; ++pt) {
                const long p = p0 + pt * 16 + r16;
                if (p < P) *reinterpret_cast<float4*>(Y + p * ldy + n) = make_float4(acc[pt][j][0], acc[pt][j][1], acc[pt][j][2], acc[pt][j][3]);
            }
            if constexpr (EPI != 0) {
                float sm[4] = {0.f, 0.f, 0.f, 0.f}, q[4] = {0.f, 0.f, 0.f, 0.f};
                if constexpr (EPI == 1) {
#pragma unroll
                    for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { const float v = acc[pt][j][r]; sm[r] += v; q[r] = fmaf(v, v, q[r]); }
                } else {
                    constexpr int NC = NW * TNW * 16;
                    const float4 sc4 = *reinterpret_cast<const float4*>(ecoef + n), sh4 = *reinterpret_cast<const float4*>(ecoef + NC + n),
                                 m4 = *reinterpret_cast<const float4*>(ecoef + 2 * NC + n), is4 = *reinterpret_cast<const float4*>(ecoef + 3 * NC + n);
                    const float scv[4] = {sc4.x, sc4.y, sc4.z, sc4.w}, shv[4] = {sh4.x, sh4.y, sh4.z, sh4.w}, mv[4] = {m4.x, m4.y, m4.z, m4.w},
                                iv[4] = {is4.x, is4.y, is4.z, is4.w};
#pragma unroll
                    for (int pt = 0; pt < PTS; ++pt) {
                        const float hv[4] = {hq[pt][j].x, hq[pt][j].y, hq[pt][j].z, hq[pt][j].w};
#pragma unroll
                        for (int r = 0; r < 4; ++r) {                   // K6's backward reduce, term for term (rows past P: g = 0)
                            const float z = fmaf(hv[r], scv[r], shv[r]);
                            const float gq = acc[pt][j][r];
                            const float dz = z > 0.f ? gq : gq * fz.epi_slope;
                            const float xh = (hv[r] - mv[r]) * iv[r];
                            sm[r] += dz;
                            q[r] = fmaf(dz, xh, q[r]);
                        }
                    }
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) { sm[r] = row16_sum(sm[r]); q[r] = row16_sum(q[r]); }
                if (r16 == 0) {         // 32 (or 64) points summed in fp32 by the fixed DPP tree, float64 from here on, fixed order
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        double2 t = wg_stats[n + r];
                        t.x += (double)sm[r];
                        t.y += (double)q[r];
                        wg_stats[n + r] = t;
                    }
                }
            }
        }
    };

    // ---- persistent loop over the 64-point tiles; ONE static body per tile
    int tile = blockIdx.x;
    if (tile < tiles64) load_slab(tile, 0);
    __syncthreads();                                                        // psc / psh visible
    if (tile < tiles64) store_slab(0, tile, 0);
    __syncthreads();
    int buf = 0;
    for (; tile < tiles64; tile += gridDim.x) {
        const int next = tile + gridDim.x < tiles64 ? tile + gridDim.x : tile;          // no next tile: a re-read nobody uses
        // one slab step: fetch the next slab (and, streaming, its weights), multiply this one, then move the fetched slab into LDS
        // The in-order vmcnt counter and where the waits fall (the compiler's waits count the LOADS it tracks; the hardware
        // counter also holds every store issued before them): the next slab's loads are issued BEHIND the result stores of all
        // passes but the last, multiply under the last pass (one pass ~ 5 us: more than the load latency), and are consumed —
        // the wait — before the last pass's stores go out.  So a wait only ever covers stores that are at least a pass old, and
        // the last pass's stores have the whole next tile to complete.  (First version: loads first, one wait behind twelve fresh
        // stores per tile — 63 % of the MFMA peak with the matrix pipe idle in every tile's store drain.)
        auto step = [&](int sl, int set_now, int set_next) __attribute__((always_inline)) {
            const int nsl = sl + 1 < SLABS ? sl + 1 : 0;
            const int ntile = sl + 1 < SLABS ? tile : next;
            if constexpr (SLABS == 1) {
#pragma unroll
                for (int pass = 0; pass < PASSES; ++pass) {
                    if (pass == PASSES - 1) {
                        load_slab(ntile, nsl);
                        __builtin_amdgcn_sched_barrier(0);          // ... and the loads in front of it (the scheduler sinks them otherwise)
                    }
#pragma unroll
                    for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
                        for (int j = 0; j < TNW; ++j) acc[pt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                    prefetch_h(tile, pass);
                    multiply(buf, 0, pass);
                    if (pass == PASSES - 1) {
                        __builtin_amdgcn_sched_barrier(0);          // keep the wait for the slab behind the multiply
                        store_slab(buf ^ 1, ntile, nsl);
                    }
                    epilogue(tile, pass);
                }
            } else {
                load_slab(ntile, nsl);
                if constexpr (WSTREAM) load_weights(set_next, nsl);             // in flight during this multiply
                __builtin_amdgcn_sched_barrier(0);
                if (sl == 0) {
#pragma unroll
                    for (int pt = 0; pt < PTS; ++pt)
#pragma unroll
                        for (int j = 0; j < TNW; ++j) acc[pt][j] = f32x4{0.f, 0.f, 0.f, 0.f};
                }
                if (sl == SLABS - 1) prefetch_h(tile, 0);
                multiply(buf, set_now, 0);
                __builtin_amdgcn_sched_barrier(0);
                store_slab(buf ^ 1, ntile, nsl);
                if (sl == SLABS - 1) epilogue(tile, 0);
            }
            __syncthreads();
            buf ^= 1;
        };
        if constexpr (WSTREAM) {
            // two register sets alternate: the slab loop runs in PAIRS and is not unrolled further (unrolled over four slabs the
            // scheduler hoisted every slab's LDS reads and weight loads: 87 spilled registers)
            static_assert(SLABS % 2 == 0, "weight streaming alternates two register sets");
#pragma unroll 1
            for (int sl = 0; sl < SLABS; sl += 2) {
                step(sl, 0, 1);
                step(sl + 1, 1, 0);
            }
        } else {
#pragma unroll
            for (int sl = 0; sl < SLABS; ++sl) step(sl, sl, 0);
        }
    }
    if constexpr (EPI != 0) {
        __syncthreads();
        for (int c = tid; c < N; c += NT) fz.partial[(long)c * gridDim.x + blockIdx.x] = wg_stats[c];
    }
}

// ------------------------------------------------------------------------------------------------------------
// pw_rows, K-SPLIT stationary weights (round 5): the K = 528 direction of the stage-2 FFN (fc2 forward 528 -> 132 and the data gradient of
// fc1), which the kernel above cannot hold — 132 x 528 weights are 279 KB, more than a wave's registers — and which therefore ran on the
// LDS-tiled pw_rows_kernel at 0.53-0.57 of the fp32 MFMA peak (weights re-staged for every tile, two barriers per 16 k).  Here the FOUR
// waves of a workgroup split K: wave w keeps W[:, 132 w .. 132 w + 131] for all nine 16-channel tiles in registers (297 VGPRs, one
// wave per SIMD) for the whole launch, the workgroup streams 16-point tiles of x (16 x 528 floats) through a double-buffered LDS slab,
// every wave multiplies ITS k slice of the tile (297 MFMAs, nine independent accumulators), the four partial products meet in a
// double-buffered LDS area and are summed in the fixed order (w0 + w1) + (w2 + w3) by threads that own (point, channel quad) pairs laid
// out so that the 16 points of a quad sit in one DPP row — the statistics epilogue is a row reduction, no extra barrier.  ONE barrier
// per tile.  x is read once, weights never touch LDS.  Another (fixed) order of the fp32 sum than pw_rows_kernel's: the chain is cut at
// multiples of 132.  PRO / EPI as above (EPI = 2 is not instantiated: the K = 528 direction never carries it).
// ------------------------------------------------------------------------------------------------------------
constexpr int kKsPts = 16, kKsK = 4 * kSwSk, kKsLdk = kKsK + 4, kKsTn = 9, kKsN4 = kKsTn * 4;
constexpr int kKsTnw = 3, kKsWaves = 4 * (kKsTn / kKsTnw), kKsThreads = 64 * kKsWaves;      // 12 waves: (k slab 0..3) x (three channel tiles each)

template <int PRO, int EPI>
__global__ __launch_bounds__(kKsThreads) void pw_rows_ks_kernel(const float* __restrict__ X, const float* __restrict__ Wt, float* __restrict__ Y, int P,
                                                        int N, long ldx, long ldy, int tiles16, PwFuse fz) {
    constexpr int K = kKsK, LDK = kKsLdk, TN = kKsTn, NCH = kSwNch, SK = kSwSk, ROW4 = K / 4;
    constexpr int NT = kKsThreads, TNW = kKsTnw;
    constexpr int NV = (kKsPts * ROW4 + NT - 1) / NT;                      // float4 per thread per tile (3)
    extern __shared__ float4 pw_smem4[];
    float* xs = reinterpret_cast<float*>(pw_smem4);                        // [2][16][LDK]
    f32x4* red = reinterpret_cast<f32x4*>(xs + 2 * kKsPts * LDK);          // [2][4 waves][16 points][36 quads]
    float* psc = reinterpret_cast<float*>(red + 2 * 4 * kKsPts * kKsN4);   // PRO: [K] scale, [K] shift
    float* psh = psc + (PRO ? K : 0);
    double2* wg_stats = reinterpret_cast<double2*>(psh + (PRO ? K : 0));   // EPI: [144] per-channel (sum, sum of squares) of this workgroup
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, kg = lane >> 4;
    const int ks = wave & 3, j0 = (wave >> 2) * TNW;                       // this wave's k slab and first channel tile

    // ---- the wave's weights: wreg[j][c] = W[16 (j0 + j) + r16][132 ks + 16 c + 4 kg + {0..3}], wtail[j] = W[..][132 ks + 128 + kg]
    // (99 registers: the first version — four waves, nine tiles each, 297 — went over the 256 architectural VGPRs, the compiler parked
    // weights in AGPRs and paid a v_accvgpr_read in front of 233 of the 297 MFMAs of every tile: 300-400 us)
    f32x4 wreg[TNW][NCH];
    float wtail[TNW];
#pragma unroll
    for (int j = 0; j < TNW; ++j) {
        const int n = (j0 + j) * 16 + r16;
        const float* wp = Wt + (long)min(n, N - 1) * K + ks * SK;
        const bool ok = n < N;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 v = ld4(wp + 16 * c + 4 * kg);
            wreg[j][c] = ok ? f32x4{v.x, v.y, v.z, v.w} : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        wtail[j] = ok ? wp[16 * NCH + kg] : 0.f;
    }
    if constexpr (PRO == 1) {
        for (int k = tid; k < K; k += NT) { psc[k] = fz.pro_scale[k]; psh[k] = fz.pro_shift[k]; }
    }
    if constexpr (EPI != 0) {
        for (int c = tid; c < TN * 16; c += NT) wg_stats[c] = make_double2(0.0, 0.0);
    }

    float4 xr[NV];
    auto load_tile = [&](int tile) __attribute__((always_inline)) {
        const long p0 = (long)tile * kKsPts;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = min(tid + i * NT, kKsPts * ROW4 - 1), row = f / ROW4, c4 = f - row * ROW4;
            const long p = min(p0 + row, (long)P - 1);
            xr[i] = ld4(X + p * ldx + c4 * 4);
        }
    };
    auto store_tile = [&](int buf, int tile) __attribute__((always_inline)) {
        float* dst = xs + buf * kKsPts * LDK;
        const long p0 = (long)tile * kKsPts;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int f = tid + i * NT;
            if (f < kKsPts * ROW4) {
                const int row = f / ROW4, c4 = f - row * ROW4;
                float4 v = xr[i];
                if constexpr (PRO == 1) {
                    const float4 sc = *reinterpret_cast<const float4*>(psc + c4 * 4), sh = *reinterpret_cast<const float4*>(psh + c4 * 4);
                    v = make_float4(leaky_f(fmaf(v.x, sc.x, sh.x), fz.pro_slope), leaky_f(fmaf(v.y, sc.y, sh.y), fz.pro_slope),
                                    leaky_f(fmaf(v.z, sc.z, sh.z), fz.pro_slope), leaky_f(fmaf(v.w, sc.w, sh.w), fz.pro_slope));
                }
                *reinterpret_cast<float4*>(dst + row * LDK + c4 * 4) = keep_if(p0 + row < P, v);
            }
        }
    };

    int tile = blockIdx.x;
    if (tile < tiles16) load_tile(tile);
    __syncthreads();                                                        // psc / psh / wg_stats visible
    if (tile < tiles16) store_tile(0, tile);
    __syncthreads();
    int buf = 0, rb = 0;
    const int n4s = N >> 2;
    for (; tile < tiles16; tile += gridDim.x) {
        const int next = tile + gridDim.x < tiles16 ? tile + gridDim.x : tile;           // no next tile: a re-read nobody uses
        load_tile(next);
        __builtin_amdgcn_sched_barrier(0);
        // ---- this wave's k slice of the tile
        f32x4 acc[TNW];
#pragma unroll
        for (int j = 0; j < TNW; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* base = xs + buf * kKsPts * LDK + r16 * LDK + ks * SK + 4 * kg;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const float4 b = ld4(base + 16 * c);
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][c][0], b.x, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][c][1], b.y, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][c][2], b.z, acc[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wreg[j][c][3], b.w, acc[j], 0, 0, 0);
        }
        {
            const float bt = base[16 * NCH - 3 * kg];                       // column 128 + kg of the slice
#pragma unroll
            for (int j = 0; j < TNW; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wtail[j], bt, acc[j], 0, 0, 0);
        }
        // ---- partial products of (point r16, channels 16 j + 4 kg ..) -> LDS; the next tile -> the other x buffer
        f32x4* mine = red + ((rb * 4 + ks) * kKsPts + r16) * kKsN4 + 4 * j0 + kg;
#pragma unroll
        for (int j = 0; j < TNW; ++j) mine[4 * j] = acc[j];
        __builtin_amdgcn_sched_barrier(0);
        store_tile(buf ^ 1, next);
        __syncthreads();
        // ---- (s0 + s1) + (s2 + s3) over the four k slabs, rows out, statistics: thread = (point tid & 15, quad tid >> 4)
        const long p = (long)tile * kKsPts + (tid & 15);
        const f32x4* rbase = red + (rb * 4 * kKsPts + (tid & 15)) * kKsN4;
        {
            const int n4 = tid >> 4;
            if (n4 < n4s) {                                                 // (uniform over a DPP row)
                const f32x4 a0 = rbase[n4], a1 = rbase[kKsPts * kKsN4 + n4], a2 = rbase[2 * kKsPts * kKsN4 + n4],
                            a3 = rbase[3 * kKsPts * kKsN4 + n4];
                f32x4 v = (a0 + a1) + (a2 + a3);
                if (p < P) *reinterpret_cast<float4*>(Y + p * ldy + 4 * n4) = make_float4(v[0], v[1], v[2], v[3]);
                if constexpr (EPI == 1) {
                    if (p >= P) v = f32x4{0.f, 0.f, 0.f, 0.f};
                    float sm[4], q[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) { sm[r] = row16_sum(v[r]); q[r] = row16_sum(v[r] * v[r]); }
                    if ((tid & 15) == 0) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            double2 t = wg_stats[4 * n4 + r];
                            t.x += (double)sm[r];
                            t.y += (double)q[r];
                            wg_stats[4 * n4 + r] = t;
                        }
                    }
                }
            }
        }
        buf ^= 1;
        rb ^= 1;
    }
    if constexpr (EPI != 0) {
        __syncthreads();
        for (int c = tid; c < N; c += kKsThreads) fz.partial[(long)c * gridDim.x + blockIdx.x] = wg_stats[c];
    }
}

// ------------------------------------------------------------------------------------------------------------
// part[split, n, k] = sum_{p in split} gy[p, n] x[p, k]
// ------------------------------------------------------------------------------------------------------------
constexpr int kWgPC = 16;   // points per LDS stage

constexpr int wgrad_stride(int cols) { return cols + ((16 - cols % 32) + 32) % 32; }   // = 16 (mod 32) floats

// wave tile TN x TK 16x16 tiles; WN x WK waves per workgroup (all compile-time: the staging loops and LDS strides are
// constants, which keeps the (3,3) kernel at ~100 VGPRs instead of 256 + SGPR spills with run-time shapes)
// PRO = 1: the x operand is normalised + activated on its way into LDS (see PwFuse / pw_rows_kernel): the weight gradient of a
// convolution whose input was only ever produced inside the forward GEMM's operand load.
template <int TN, int TK, int WN, int WK, int PRO>
__global__ __launch_bounds__(64 * WN * WK, 2) void pw_wgrad_kernel(const float* __restrict__ G, const float* __restrict__ X,
                                                                float* __restrict__ part, int P, int N, int K, long ldg, long ldx,
                                                                int nb_n, int nb_k, int rows_per_split, int items, PwFuse fz) {
    constexpr int NT = 64 * WN * WK;
    constexpr int BN = WN * TN * 16, BK = WK * TK * 16;
    constexpr int SG = wgrad_stride(BN), SX = wgrad_stride(BK);
    constexpr int GQ = BN / 4, XQ = BK / 4;                             // float4 groups per tile row
    constexpr int GV = (kWgPC * GQ + NT - 1) / NT, XV = (kWgPC * XQ + NT - 1) / NT;
    extern __shared__ float4 pw_smem4[];
    float* Gs = reinterpret_cast<float*>(pw_smem4);        // [2][kWgPC][SG]
    float* Xs = Gs + 2 * kWgPC * SG;                       // [2][kWgPC][SX]
    float* Psc = Xs + 2 * kWgPC * SX;                      // PRO: [BK] scale, then [BK] shift of this workgroup's k block
    float* Psh = Psc + BK;

    const int item = xcd_item(blockIdx.x, gridDim.x);
    if (item >= items) return;
    const int tiles = nb_n * nb_k;
    const int split = item / tiles, tile = item - split * tiles;
    const int bn = tile / nb_k, bk = tile - bn * nb_k;
    const int g = blockIdx.y;
    G += (long)g * N;
    X += (long)g * K;
    const int n0 = bn * BN, k0 = bk * BK;
    const int p_begin = split * rows_per_split;
    const int p_end = min(P, p_begin + rows_per_split);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, pg = lane >> 4;
    const int wave_n = wave / WK, wave_k = wave - wave_n * WK;
    if constexpr (PRO == 1) {
        for (int c = tid; c < BK; c += NT) {
            const long k = (long)g * K + min(k0 + c, K - 1);
            Psc[c] = fz.pro_scale[k];
            Psh[c] = fz.pro_shift[k];
        }
    }

    // register staging with prefetch distance 2 (see pw_rows_kernel)
    float4 gr[2][GV], xr[2][XV];
    auto load_stage = [&](int p_row0, float4 (&gq)[GV], float4 (&xq)[XV]) __attribute__((always_inline)) {          // unconditional, clamped, untouched until the store: see pw_rows_kernel
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int f = tid + i * NT, row = f / GQ, c = (f - row * GQ) * 4;
            gq[i] = ld4(G + (long)min(p_row0 + row, P - 1) * ldg + min(n0 + c, N - 4));
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + i * NT, row = f / XQ, c = (f - row * XQ) * 4;
            xq[i] = ld4(X + (long)min(p_row0 + row, P - 1) * ldx + min(k0 + c, K - 4));
        }
    };
    auto store_stage = [&](int buf, int p_row0, const float4 (&gq)[GV], const float4 (&xq)[XV]) __attribute__((always_inline)) {
        float* gs = Gs + buf * kWgPC * SG;
        float* xs = Xs + buf * kWgPC * SX;
#pragma unroll
        for (int i = 0; i < GV; ++i) {
            const int f = tid + i * NT, row = f / GQ, c = (f - row * GQ) * 4;
            const bool ok = p_row0 + row < p_end && n0 + c < N;
            if (row < kWgPC) *reinterpret_cast<float4*>(gs + row * SG + c) = keep_if(ok, gq[i]);
        }
#pragma unroll
        for (int i = 0; i < XV; ++i) {
            const int f = tid + i * NT, row = f / XQ, c = (f - row * XQ) * 4;
            const bool ok = p_row0 + row < p_end && k0 + c < K;
            float4 v = xq[i];
            if constexpr (PRO == 1) {
                if (row < kWgPC) {
                    const float4 sc = *reinterpret_cast<const float4*>(Psc + c), sh = *reinterpret_cast<const float4*>(Psh + c);
                    v = make_float4(leaky_f(fmaf(v.x, sc.x, sh.x), fz.pro_slope), leaky_f(fmaf(v.y, sc.y, sh.y), fz.pro_slope),
                                    leaky_f(fmaf(v.z, sc.z, sh.z), fz.pro_slope), leaky_f(fmaf(v.w, sc.w, sh.w), fz.pro_slope));
                }
            }
            if (row < kWgPC) *reinterpret_cast<float4*>(xs + row * SX + c) = keep_if(ok, v);
        }
    };

    f32x4 acc[TN][TK];
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    auto compute = [&](int buf) __attribute__((always_inline)) {
        const float* gs = Gs + buf * kWgPC * SG + pg * SG + wave_n * TN * 16 + c16;
        const float* xs = Xs + buf * kWgPC * SX + pg * SX + wave_k * TK * 16 + c16;
#pragma unroll
        for (int r = 0; r < kWgPC; r += 4) {
            float a[TN], b[TK];
#pragma unroll
            for (int i = 0; i < TN; ++i) a[i] = gs[r * SG + i * 16];
#pragma unroll
            for (int j = 0; j < TK; ++j) b[j] = xs[r * SX + j * 16];
#pragma unroll
            for (int i = 0; i < TN; ++i)
#pragma unroll
                for (int j = 0; j < TK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    const int stages = (p_end - p_begin + kWgPC - 1) / kWgPC;
    if (stages > 0) {
        load_stage(p_begin, gr[0], xr[0]);
        load_stage(p_begin + kWgPC, gr[1], xr[1]);
    }
    if constexpr (PRO == 1) __syncthreads();           // Psc / Psh are read by the first store_stage
    if (stages > 0) store_stage(0, p_begin, gr[0], xr[0]);
    __syncthreads();
    for (int s = 0; s + 1 < stages; s += 2) {      // pairs, no exit inside, loads always issued: see pw_rows_kernel
        load_stage(p_begin + (s + 2) * kWgPC, gr[0], xr[0]);       // past p_end: zeros (addresses clamped)
        compute(0);
        store_stage(1, p_begin + (s + 1) * kWgPC, gr[1], xr[1]);
        __syncthreads();
        load_stage(p_begin + (s + 3) * kWgPC, gr[1], xr[1]);
        compute(1);
        store_stage(0, p_begin + (s + 2) * kWgPC, gr[0], xr[0]);          // past p_end: zeros nobody reads
        __syncthreads();
    }
    if (stages & 1) compute(0);

    // D tile: row = 4 (lane >> 4) + reg = n, column = lane & 15 = k
    float* out = part + ((long)split * gridDim.y + g) * (long)N * K;
#pragma unroll
    for (int i = 0; i < TN; ++i)
#pragma unroll
        for (int j = 0; j < TK; ++j) {
            const int k = k0 + (wave_k * TK + j) * 16 + c16;
            if (k >= K) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int n = n0 + (wave_n * TN + i) * 16 + pg * 4 + reg;
                if (n < N) out[(long)n * K + k] = acc[i][j][reg];
            }
        }
}

// ------------------------------------------------------------------------------------------------------------
// pw_wgrad, STATIONARY OUTPUT (round 3): dw[n, k] = sum_p gy[p, n] x[p, k] for the stage-2 shapes.  One workgroup per CU holds the WHOLE
// (N x K) product in its waves' accumulators (wave (wn, wk): A x BT tiles of 16 x 16) and streams a contiguous range of 16-point
// slabs of gy and x through a double-buffered LDS tile: both operands are read from HBM exactly once, nothing is stored inside the
// loop, one barrier per slab.  pw_wgrad_kernel cuts the product into (tile, point-split) work items instead: every item re-reads
// its gy / x columns through L2 and the 121-245 splits leave partial tiles for the reduction.  Here the partials are one (N x K)
// block per workgroup, summed by the same fixed-order pw_wgrad_reduce_kernel (bit-reproducible).  Shape-specialised; the MFMA
// operands are rows of the slabs as they lie in memory (lane (c16, pg): tile[4 step + pg][col + c16], row stride = 16 mod 32).
// ------------------------------------------------------------------------------------------------------------
constexpr int kSoPS = 16;        // points per slab

template <int N, int K, int A, int BT, int WN, int WK, int PRO>
__global__ __launch_bounds__(64 * WN * WK) void pw_wgrad_so_kernel(const float* __restrict__ G, const float* __restrict__ X, float* __restrict__ part,
                                                                 int P, long ldg, long ldx, int slabs, int slabs_per_wg, PwFuse fz) {
    constexpr int NT = 64 * WN * WK;
    constexpr int SG = wgrad_stride(WN * A * 16), SX = wgrad_stride(WK * BT * 16);
    constexpr int N4 = N / 4, K4 = K / 4;
    constexpr int NVG = (kSoPS * N4 + NT - 1) / NT, NVX = (kSoPS * K4 + NT - 1) / NT;
    constexpr int RG = (NVG * NT + N4 - 1) / N4, RX = (NVX * NT + K4 - 1) / K4;       // LDS rows incl. the staging overhang (see pw_rows_sw_kernel)
    static_assert(N % 4 == 0 && K % 4 == 0 && WN * A * 16 >= N && WK * BT * 16 >= K, "tile cover");
    extern __shared__ float4 pw_smem4[];
    float* Gs = reinterpret_cast<float*>(pw_smem4);        // [2][RG][SG]
    float* Xs = Gs + 2 * RG * SG;                          // [2][RX][SX]
    float* psc = Xs + 2 * RX * SX;                         // PRO: [K] scale, [K] shift
    float* psh = psc + (PRO ? K : 0);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int c16 = lane & 15, pg = lane >> 4;
    const int wn = wave / WK, wk = wave - wn * WK;
    if constexpr (PRO == 1) {
        for (int k = tid; k < K; k += NT) { psc[k] = fz.pro_scale[k]; psh[k] = fz.pro_shift[k]; }
    }
    const int s_begin = blockIdx.x * slabs_per_wg;
    const int s_end = min(slabs, s_begin + slabs_per_wg);

    float4 gr[NVG], xr[NVX];
    auto load_slab = [&](int sl) __attribute__((always_inline)) {
        const long p0 = (long)sl * kSoPS;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const int f = tid + i * NT, row = f / N4, c4 = f - row * N4;
            gr[i] = ld4(G + min(p0 + min(row, kSoPS - 1), (long)P - 1) * ldg + c4 * 4);
        }
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            const int f = tid + i * NT, row = f / K4, c4 = f - row * K4;
            xr[i] = ld4(X + min(p0 + min(row, kSoPS - 1), (long)P - 1) * ldx + c4 * 4);
        }
    };
    auto store_slab = [&](int buf, int sl) __attribute__((always_inline)) {
        const long p0 = (long)sl * kSoPS;
        float* gs = Gs + buf * RG * SG;
        float* xs = Xs + buf * RX * SX;
#pragma unroll
        for (int i = 0; i < NVG; ++i) {
            const int f = tid + i * NT, row = f / N4, c4 = f - row * N4;
            *reinterpret_cast<float4*>(gs + row * SG + c4 * 4) = keep_if(p0 + row < P, gr[i]);      // points past P add nothing
        }
#pragma unroll
        for (int i = 0; i < NVX; ++i) {
            const int f = tid + i * NT, row = f / K4, c4 = f - row * K4;
            float4 v = xr[i];
            if constexpr (PRO == 1) {
                const float4 sc = *reinterpret_cast<const float4*>(psc + c4 * 4), sh = *reinterpret_cast<const float4*>(psh + c4 * 4);
                v = make_float4(leaky_f(fmaf(v.x, sc.x, sh.x), fz.pro_slope), leaky_f(fmaf(v.y, sc.y, sh.y), fz.pro_slope),
                                leaky_f(fmaf(v.z, sc.z, sh.z), fz.pro_slope), leaky_f(fmaf(v.w, sc.w, sh.w), fz.pro_slope));
            }
            *reinterpret_cast<float4*>(xs + row * SX + c4 * 4) = v;      // (gy's rows past P are zero: the products vanish)
        }
    };

    f32x4 acc[A][BT];
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

    if (s_begin < s_end) load_slab(s_begin);
    __syncthreads();                                       // psc / psh visible
    if (s_begin < s_end) store_slab(0, s_begin);
    __syncthreads();
    int buf = 0;
    for (int sl = s_begin; sl < s_end; ++sl) {
        load_slab(sl + 1 < s_end ? sl + 1 : sl);           // last slab: a re-read nobody uses
        __builtin_amdgcn_sched_barrier(0);
        const float* gs = Gs + buf * RG * SG + pg * SG + wn * A * 16 + c16;
        const float* xs = Xs + buf * RX * SX + pg * SX + wk * BT * 16 + c16;
#pragma unroll
        for (int st = 0; st < kSoPS / 4; ++st) {
            float a[A], b[BT];
#pragma unroll
            for (int i = 0; i < A; ++i) a[i] = gs[st * 4 * SG + i * 16];
#pragma unroll
            for (int j = 0; j < BT; ++j) b[j] = xs[st * 4 * SX + j * 16];
#pragma unroll
            for (int i = 0; i < A; ++i)
#pragma unroll
                for (int j = 0; j < BT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        store_slab(buf ^ 1, sl + 1 < s_end ? sl + 1 : sl);
        __syncthreads();
        buf ^= 1;
    }
    // D tile: row = 4 (lane >> 4) + reg = n, column = lane & 15 = k
    float* out = part + (long)blockIdx.x * N * K;
#pragma unroll
    for (int i = 0; i < A; ++i)
#pragma unroll
        for (int j = 0; j < BT; ++j) {
            const int k = (wk * BT + j) * 16 + c16;
            if (k >= K) continue;
#pragma unroll
            for (int reg = 0; reg < 4; ++reg) {
                const int n = (wn * A + i) * 16 + pg * 4 + reg;
                if (n < N) out[(long)n * K + k] = acc[i][j][reg];
            }
        }
}

// dw[e] = (accumulate ? dw[e] : 0) + sum_split part[split][e] in a fixed order (deterministic): 4 threads per element take
// the splits = slice (mod 4) in ascending order, then (s0 + s1) + (s2 + s3)
__global__ __launch_bounds__(256) void pw_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, long elems, int splits,
                                                              int accumulate) {
    __shared__ float partial[4][64];
    const int slice = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long e = (long)blockIdx.x * 64 + lane;
    float s = 0.f;
    if (e < elems)
        for (int i = slice; i < splits; i += 4) s += part[(long)i * elems + e];
    partial[slice][lane] = s;
    __syncthreads();
    if (slice == 0 && e < elems) {
        const float t = (partial[0][lane] + partial[1][lane]) + (partial[2][lane] + partial[3][lane]);
        dw[e] = accumulate ? dw[e] + t : t;
    }
}

// ------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------
int cu_count() {
    static int n = 0;
    if (n == 0) {
        int dev = 0, v = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && v > 0) n = v;
        else n = 256;
    }
    return n;
}

struct RowsPlan { int tm, tn, nb_n, nb_p, items, grid; size_t lds; };

inline int tiles16(int c) { return (c + 15) / 16; }

// channel tiles per workgroup: the divisor-like choice that wastes least; ties -> the larger (fewer re-reads of x)
RowsPlan plan_rows(int P, int N, int groups, bool allow_tm1 = false) {
    static const int kTn[] = {1, 3, 6, 7, 9, 11};
    const int t = tiles16(N);
    int best = 1;
    long best_cost = -1;
    for (int tn : kTn) {
        const long padded = (long)((t + tn - 1) / tn) * tn;
        // cost: padded MFMA work, plus a small penalty per extra pass over x (more workgroup columns)
        const long cost = padded * 100 + (long)((t + tn - 1) / tn) * 3 * 16;
        if (best_cost < 0 || cost < best_cost || (cost == best_cost && tn > best)) { best_cost = cost; best = tn; }
    }
    if (const char* e = getenv("NEXTOU_PW_ROWS_TN")) {          // experiment: force the channel tiles per workgroup
        const int v = atoi(e);
        for (int tn : kTn)
            if (tn == v) best = v;
    }
    RowsPlan q{};
    q.tm = 2;
    if (const char* e = getenv("NEXTOU_PW_ROWS_TM"))                                     // experiment: 64-point workgroup tiles
        q.tm = (atoi(e) == 1 && allow_tm1 && (best == 9 || best == 11)) ? 1 : 2;
    q.tn = best;
    q.nb_n = (t + best - 1) / best;
    q.nb_p = (P + 64 * q.tm - 1) / (64 * q.tm);
    q.items = q.nb_n * q.nb_p;
    q.lds = (size_t)2 * (64 * q.tm + 16 * q.tn) * kPwLd * sizeof(float);
    // persistent grid: what is resident at once (registers: 2 workgroups per CU from 9 channel tiles up, more below)
    const int per_cu = q.tm == 1 ? (best >= 9 ? 3 : 5) : (best >= 9 ? 2 : best >= 6 ? 3 : best >= 3 ? 5 : 8);
    int grid = cu_count() * per_cu / (groups > 0 ? groups : 1);
    if (grid > q.items) grid = q.items;
    q.grid = (grid + kXcds - 1) / kXcds * kXcds;
    return q;
}

struct WgradPlan { int cfg, tn, tk, wn, wk, nb_n, nb_k, splits, rows_per_split, items, grid; size_t lds; };

// the instantiated (wave tile, wave arrangement) pairs: X(TN, TK, WN, WK); kWgradPerCu: workgroups resident per CU (registers)
#define NEXTOU_WGRAD_CONFIGS(X) \
    X(3, 9, 4, 1) X(9, 3, 1, 4) X(3, 9, 3, 1) X(9, 3, 1, 3) X(3, 3, 1, 1) X(3, 3, 2, 2)
static const int kWgradPerCu[] = {2, 2, 2, 2, 12, 4};

// Cost model (seconds, crude).  Matrix cores: a wave's share is (rows / 4) * TN * TK MFMAs of 32 cycles; a CU runs its
// resident workgroups' waves on 4 pipes; workgroups beyond what is resident at once run in further rounds — the split count
// is chosen so that the LAST round is not a nearly empty one (558 workgroups on 512 slots cost two rounds: measured 364 us
// where one round takes ~200).  Every workgroup column re-reads gy and every row re-reads x (through L2); the partial tiles
// are written and read once more by the reduction.  Measured choices: profiles/r02_pw_gemm.md.
WgradPlan plan_wgrad(int P, int N, int K, int groups) {
    static const int kCfg[][4] = {
#define X(a, b, c, d) {a, b, c, d},
        NEXTOU_WGRAD_CONFIGS(X)
#undef X
    };
    const int tn_all = tiles16(N), tk_all = tiles16(K);
    int force[4] = {0, 0, 0, 0};
    const char* env = getenv("NEXTOU_PW_WGRAD_TILE");      // experiment: "TN,TK,WN,WK"
    const bool forced = env && sscanf(env, "%d,%d,%d,%d", &force[0], &force[1], &force[2], &force[3]) == 4;
    const int max_splits = (P + 4 * kWgPC - 1) / (4 * kWgPC);
    const int cus = cu_count();
    WgradPlan best{};
    double best_cost = -1.0;
    int id = 0;
    for (const auto& c : kCfg) {
        const int cfg = id++;
        if (forced && (force[0] != c[0] || force[1] != c[1] || force[2] != c[2] || force[3] != c[3])) continue;
        const int bn = c[0] * c[2], bk = c[1] * c[3], waves = c[2] * c[3];
        const int nb_n = (tn_all + bn - 1) / bn, nb_k = (tk_all + bk - 1) / bk;
        const int tiles = nb_n * nb_k * groups;
        const long capacity = (long)cus * kWgradPerCu[cfg];
        const double t_l2 = 4.0 * P * ((double)nb_k * N + (double)nb_n * K) * groups / 8e12;
        const double t_in = 4.0 * P * (double)(N + K) * groups / 4.5e12;
        for (int splits = 1; splits <= max_splits; splits = splits < 16 ? splits + 1 : splits + splits / 16) {
            const long blocks = (long)tiles * splits;
            if (blocks > 8 * capacity) break;
            const int rows = ((P + splits - 1) / splits + kWgPC - 1) / kWgPC * kWgPC;
            const double wave_s = (rows / 4.0) * c[0] * c[1] * 32.0 / 2.4e9 * (1.0 + 0.3 * (c[0] + c[1]) / (double)(c[0] * c[1]));
            const long full = blocks / capacity, rem = blocks - full * capacity;
            // a SIMD with ONE wave keeps its matrix pipe ~60 % busy (LDS reads, staging and the barrier are not hidden), with
            // two or more ~95 % (measured: 512 co-resident 4-wave workgroups ran at 126 TFLOP/s, 246 at 79)
            auto round_time = [&](double waves_per_simd) {
                return waves_per_simd <= 1.0 ? 1.0 / 0.6 : waves_per_simd / 0.95;
            };
            const double share_full = kWgradPerCu[cfg] * waves / 4.0;
            const double share_rem = (double)((rem + cus - 1) / cus) * waves / 4.0;
            const double t_mfma = wave_s * (full * round_time(share_full) + (rem > 0 ? round_time(share_rem) : 0.0));
            double cost = t_mfma > t_l2 ? t_mfma : t_l2;
            if (t_in > cost) cost = t_in;
            cost += 8.0 * splits * (double)N * K * groups / 3e12;      // partial tiles: written, then read by the reduction
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best = WgradPlan{cfg, c[0], c[1], c[2], c[3], nb_n, nb_k, splits, 0, 0, 0, 0};
            }
        }
    }
    if (best_cost < 0) best = WgradPlan{5, 3, 3, 2, 2, (tn_all + 5) / 6, (tk_all + 5) / 6, 1, 0, 0, 0, 0};   // forced tile unknown
    int splits = best.splits;
    int rows = (P + splits - 1) / splits;
    rows = (rows + kWgPC - 1) / kWgPC * kWgPC;
    splits = (P + rows - 1) / rows;
    best.splits = splits;
    best.rows_per_split = rows;
    best.items = best.nb_n * best.nb_k * splits;
    best.grid = (best.items + kXcds - 1) / kXcds * kXcds;
    best.lds = (size_t)2 * kWgPC * (wgrad_stride(best.tn * best.wn * 16) + wgrad_stride(best.tk * best.wk * 16)) * sizeof(float);
    return best;
}

template <typename Kern>
int allow_lds(Kern kern, size_t bytes) {
    if (bytes <= 64 * 1024) return 0;
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e != hipSuccess) return fail((int)e, "pw: hipFuncSetAttribute(%zu B LDS): %s", bytes, hipGetErrorString(e));
    return 0;
}

template <int TM, int TN, int PRO, int EPI>
int launch_rows(const RowsPlan& q, const float* x, const float* w, const float* bias, float* y, int P, int N, int K, int groups, long ldx,
                long ldw, long ldy, int vec_store, const PwFuse& fz, hipStream_t s) {
    static size_t allowed = 0;          // per instantiation; the attribute is sticky
    const size_t lds = q.lds + (EPI != 0 ? (size_t)4 * 16 * TN * sizeof(float2) : 0);
    if (lds > allowed) {
        if (int e = allow_lds(pw_rows_kernel<TM, TN, PRO, EPI>, lds)) return e;
        allowed = lds;
    }
    hipLaunchKernelGGL((pw_rows_kernel<TM, TN, PRO, EPI>), dim3(q.grid, groups), dim3(256), lds, s, x, w, bias, y, P, N, K, ldx, ldw, ldy,
                       q.nb_n, q.items, vec_store, fz);
    return check_launch("pw_rows_kernel");
}

// ---- stationary-weights kernel: the instantiated shapes -------------------------------------------------------------
// X(id, TNW, NW, SLABS, WSTREAM): K = 132 * SLABS, up to 16 * TNW * NW output channels
#define NEXTOU_SW_CONFIGS(X) \
    X(0, 3, 11, 1, 0) /* K 132 -> N <= 528: FFN fc1 (132 -> 528) and the data gradient of FFN fc2            */ \
    X(1, 2, 9, 1, 0)  /* K 132 -> N <= 288: the data gradient of the graphers' fc2 (132 -> 264)               */ \
    X(2, 1, 9, 1, 0)  /* K 132 -> N <= 144: the graphers' fc1 (132 -> 132) and its data gradient              */ \
    X(3, 1, 9, 2, 0)  /* K 264 -> N <= 144: the graphers' fc2 (264 -> 132)                                    */ \
    X(4, 1, 9, 4, 1)  /* K 528 -> N <= 144: FFN fc2 (528 -> 132) and the data gradient of FFN fc1             */

struct SwPlan { int cfg, tnw, nw, tiles64, passes, grid; size_t lds; };

// cfg = -1 unless an instantiated shape fits: one group, enough 64-point tiles to give every CU several (the stage-2 volumes)
SwPlan plan_rows_sw(int64_t P, int N, int K, int groups, bool pro) {
    SwPlan q{};
    q.cfg = -1;
    const char* env = getenv("NEXTOU_PW_SW");          // read per call: tests and A/B runs flip it inside one process
    const int mode = env ? atoi(env) : 1;
    if (mode == 0 || groups != 1 || N % 4 != 0 || P < 64 * 4 * (int64_t)cu_count()) return q;
    const int t = tiles16(N);
    if (K == 132) q.cfg = t <= 9 ? 2 : (t <= 18 ? 1 : (t <= 33 ? 0 : -1));
    else if (K == 264 && t <= 9) q.cfg = 3;
    else if (K == 528 && t <= 9 && mode >= 2) q.cfg = 4;      // streamed weights: 346 us against pw_rows_kernel's 292 (r03_pw_rows_sw.md): opt-in
    if (q.cfg < 0) return q;
    q.tnw = q.cfg == 0 ? 3 : (q.cfg == 1 ? 2 : 1);
    q.nw = q.cfg == 0 ? 11 : 9;
    q.passes = q.cfg == 0 ? 4 : (q.cfg == 1 ? 2 : 1);
    q.tiles64 = (int)((P + 63) / 64);
    q.grid = q.tiles64 < cu_count() ? q.tiles64 : cu_count();
    const int nt = 64 * q.nw, nv = (64 * kSwSk4 + nt - 1) / nt, rows = (nv * nt + kSwSk4 - 1) / kSwSk4;      // as in the kernel
    q.lds = (size_t)2 * rows * kSwSk * sizeof(float) + (pro ? (size_t)2 * K * sizeof(float) : 0) + (size_t)q.nw * q.tnw * 16 * (sizeof(double2) + 4 * sizeof(float));
    return q;
}

template <int TNW, int NW, int SLABS, int WSTREAM, int PRO, int EPI>
int launch_rows_sw(const SwPlan& q, const float* x, const float* w, float* y, int P, int N, int K, long ldx, long ldy, const PwFuse& fz,
                   hipStream_t s) {
    static size_t allowed = 0;
    if (q.lds > allowed) {
        if (int e = allow_lds(pw_rows_sw_kernel<TNW, NW, SLABS, WSTREAM, PRO, EPI>, q.lds)) return e;
        allowed = q.lds;
    }
    hipLaunchKernelGGL((pw_rows_sw_kernel<TNW, NW, SLABS, WSTREAM, PRO, EPI>), dim3(q.grid), dim3(64 * NW), q.lds, s, x, w, y, P, N, K, ldx, ldy,
                       q.tiles64, fz);
    return check_launch("pw_rows_sw_kernel");
}

template <int PRO, int EPI>
int dispatch_rows_sw(const SwPlan& q, const float* x, const float* w, float* y, int P, int N, int K, long ldx, long ldy, const PwFuse& fz,
                     hipStream_t s) {
#define X(id, tnw, nw, slabs, wstream) \
    if (q.cfg == id) return launch_rows_sw<tnw, nw, slabs, wstream, PRO, EPI>(q, x, w, y, P, N, K, ldx, ldy, fz, s);
    NEXTOU_SW_CONFIGS(X)
#undef X
    return fail(NEXTOU_EINVAL, "pw_rows_sw: no kernel for plan %d", q.cfg);
}

// ---- K-split stationary weights (pw_rows_ks_kernel): K = 528, N <= 144, dense rows of x, one group
struct KsPlan { bool ok; int tiles16, grid; size_t lds; };
KsPlan plan_rows_ks(int64_t P, int N, int K, int groups, int64_t ldx, bool pro, int epi) {
    KsPlan q{};
    // Measured at the stage-2 FFN of cfg 2 (P = 172 032, profiles/r05_k7_k_split.md): plain (the data gradient of fc1) 249 us against
    // pw_rows_kernel's 266; with the normalise + activate prologue and the statistics epilogue (fc2 forward) 289 against 285 — level.  So
    // the kernel is taken WITHOUT a prologue only; NEXTOU_PW_KS=2 takes it for every instantiated variant (tests, A/B), 0 never.
    const char* env = getenv("NEXTOU_PW_KS");           // read per call
    const int mode = env ? atoi(env) : 1;
    if (mode == 0 || (pro && mode < 2) || groups != 1 || K != kKsK || N % 4 != 0 || N > 16 * kKsTn || ldx != K || epi == 2 ||
        P < 64 * 4 * (int64_t)cu_count())
        return q;
    q.tiles16 = (int)((P + kKsPts - 1) / kKsPts);
    q.grid = q.tiles16 < cu_count() ? q.tiles16 : cu_count();
    q.lds = ((size_t)2 * kKsPts * kKsLdk + (size_t)2 * 4 * kKsPts * kKsN4 * 4 + (pro ? (size_t)2 * kKsK : 0)) * sizeof(float) +
            (epi ? (size_t)kKsTn * 16 * sizeof(double2) : 0);
    q.ok = true;
    return q;
}

template <int PRO, int EPI>
int launch_rows_ks(const KsPlan& q, const float* x, const float* w, float* y, int P, int N, long ldx, long ldy, const PwFuse& fz, hipStream_t s) {
    static size_t allowed = 0;
    if (q.lds > allowed) {
        if (int e = allow_lds(pw_rows_ks_kernel<PRO, EPI>, q.lds)) return e;
        allowed = q.lds;
    }
    hipLaunchKernelGGL((pw_rows_ks_kernel<PRO, EPI>), dim3(q.grid), dim3(kKsThreads), q.lds, s, x, w, y, P, N, ldx, ldy, q.tiles16, fz);
    return check_launch("pw_rows_ks_kernel");
}

template <int PRO, int EPI>
int dispatch_rows(const RowsPlan& q, const float* x, const float* w, const float* bias, float* y, int P, int N, int K, int groups, long ldx,
                  long ldy, int vec_store, const PwFuse& fz, hipStream_t s) {
    if constexpr (PRO == 0 && EPI == 0) {
        if (q.tm == 1 && q.tn == 9) return launch_rows<1, 9, 0, 0>(q, x, w, bias, y, P, N, K, groups, ldx, (long)K, ldy, vec_store, fz, s);
        if (q.tm == 1 && q.tn == 11) return launch_rows<1, 11, 0, 0>(q, x, w, bias, y, P, N, K, groups, ldx, (long)K, ldy, vec_store, fz, s);
    }
#define NEXTOU_PW_ROWS(TN_) \
    case TN_: return launch_rows<2, TN_, PRO, EPI>(q, x, w, bias, y, P, N, K, groups, ldx, (long)K, ldy, vec_store, fz, s)
    switch (q.tn) {
        NEXTOU_PW_ROWS(1);
        NEXTOU_PW_ROWS(3);
        NEXTOU_PW_ROWS(6);
        NEXTOU_PW_ROWS(7);
        NEXTOU_PW_ROWS(9);
        NEXTOU_PW_ROWS(11);
    }
#undef NEXTOU_PW_ROWS
    return fail(NEXTOU_EINVAL, "pw_rows: no kernel for %d channel tiles", q.tn);
}

template <int TN, int TK, int WN, int WK, int PRO>
int launch_wgrad(const WgradPlan& q, const float* gy, const float* x, float* part, int P, int N, int K, int groups, long ldg, long ldx,
                 const PwFuse& fz, hipStream_t s) {
    static size_t allowed = 0;
    const size_t lds = q.lds + (PRO ? (size_t)2 * WK * TK * 16 * sizeof(float) : 0);
    if (lds > allowed) {
        if (int e = allow_lds(pw_wgrad_kernel<TN, TK, WN, WK, PRO>, lds)) return e;
        allowed = lds;
    }
    hipLaunchKernelGGL((pw_wgrad_kernel<TN, TK, WN, WK, PRO>), dim3(q.grid, groups), dim3(64 * WN * WK), lds, s, gy, x, part, P, N, K, ldg,
                       ldx, q.nb_n, q.nb_k, q.rows_per_split, q.items, fz);
    return check_launch("pw_wgrad_kernel");
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

int check_pw(const char* who, int64_t P, int N, int K, int groups, int64_t lda, int64_t ldb, int cols_a, int cols_b) {
    NEXTOU_REQUIRE(P > 0 && P < (int64_t(1) << 31) && N > 0 && K > 0 && groups > 0 && groups < 65536,
                   "%s: bad sizes P=%lld N=%d K=%d groups=%d", who, (long long)P, N, K, groups);
    NEXTOU_REQUIRE(lda >= (int64_t)groups * cols_a && ldb >= (int64_t)groups * cols_b, "%s: row strides %lld / %lld shorter than the rows",
                   who, (long long)lda, (long long)ldb);
    return 0;
}


// ---------------------------------------------------------------------------------------------------------------------------
// Grouped 1x1 convolution with SMALL groups (MRConv's BasicConv on a channels-last volume: 264 -> 264 in 6 groups of 44 at
// stage 2 of cfg 2; reference torch_nn.py:66-92, groups 4 | 6).  pw_rows_kernel hands every (row tile, group) to its own
// workgroup: 176-byte row segments in and out, a 128 x 48 x 44 product per workgroup — 170 us for 363 MB of traffic and
// 4 GFLOP (a quarter of what HBM allows; profiles/r03_pw_rows_grouped.md).  Here a workgroup has ONE WAVE PER GROUP and walks
// 64-row slabs: the slab's full rows come in and go out as contiguous 16-byte pieces (all groups * K channels of a point), the
// group's N x K weights sit in the wave's registers for the whole kernel (NT x KSTEPS B operands of v_mfma_f32_16x16x4_f32), the
// A operand is read from the LDS slab (row stride = 4 mod 8 floats: conflict-free), and — N == K — the result is written over
// the wave's own columns of the slab, so the store is the mirror image of the load.  EPI = 1: (sum, sum of squares) of every
// output column, fp32 over a slab's 64 rows, float64 across the workgroup's slabs, one partial per workgroup (the `partial`
// input of nextou_norm_finalize with tiles = gridDim.x).  Arithmetic: one k-ordered fp32 MFMA chain per output element.
// Measured (profiles/r03_pw_rows_grouped.md): 170 -> 128 us plain, 168 -> 144 us with the statistics epilogue.
// ---------------------------------------------------------------------------------------------------------------------------
constexpr int kGrpRows = 64;
constexpr int kGrpMaxV = 12;          // 16-byte pieces of a slab per thread

template <int NT, int KSTEPS, bool EXACT, int EPI>
__global__ __launch_bounds__(512) void pw_rows_grp_kernel(const float* __restrict__ x, const float* __restrict__ w, float* __restrict__ y,
                                                          int P, int N, int K, int groups, long ldx, long ldy, int ld_lds, int slabs,
                                                          double2* __restrict__ partial) {
    // EXACT: N == K == 4 * KSTEPS (no bound checks on the operand reads / result writes)
    extern __shared__ __attribute__((aligned(16))) float grp_lds[];        // [64][ld_lds]
    const int tid = threadIdx.x, lane = tid & 63, g = tid >> 6;             // wave g owns group g
    const int C = groups * K;                                               // == groups * N
    const int ln = lane & 15, lk = lane >> 4;
    // the group's weights: B operand of tile (nt, ks) = w[g * N + nt * 16 + ln][4 * ks + lk]
    float wreg[NT][KSTEPS];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            const int n = nt * 16 + ln, k = 4 * ks + lk;
            const int nc = n < N ? n : N - 1, kc = k < K ? k : K - 1;
            const float t = w[(size_t)(g * N + nc) * K + kc];
            wreg[nt][ks] = (n < N && k < K) ? t : 0.f;
        }
    double s1[NT], s2[NT];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) s1[nt] = s2[nt] = 0.0;
    // a slab = 64 rows of C / 4 16-byte pieces; thread t moves pieces t, t + T, ... (at most kGrpMaxV per thread: plan_rows_grp).
    // Piece e sits at (row e / c4, quad e % c4); the thread's first one is divided out once, the others follow by addition.
    const int c4 = C >> 2, T = blockDim.x, total = kGrpRows * c4;
    const int r0 = tid / c4, q0 = tid - r0 * c4, dr = T / c4, dq = T - dr * c4;
    for (int slab = blockIdx.x; slab < slabs; slab += gridDim.x) {
        const int p0 = slab * kGrpRows;
        __syncthreads();                                                    // the previous slab has left the LDS
        {
            // in two batches: half of the thread's loads in flight, then their LDS writes.  Branch-free: rows past the end (and
            // the pieces past `total`, never stored) re-read row P - 1; such a row's product is computed and dropped
            int r = r0, q = q0;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                f32x4 v[kGrpMaxV / 2];
                int rl = r, ql = q;
#pragma unroll
                for (int i = 0; i < kGrpMaxV / 2; ++i) {
                    const int rr = (tid + (h * (kGrpMaxV / 2) + i) * T < total) ? rl : 0;
                    const int row = p0 + rr < P ? p0 + rr : P - 1;
                    v[i] = *reinterpret_cast<const f32x4*>(x + (size_t)row * ldx + 4 * ql);
                    rl += dr; ql += dq;
                    if (ql >= c4) { ql -= c4; ++rl; }
                }
#pragma unroll
                for (int i = 0; i < kGrpMaxV / 2; ++i) {
                    if (tid + (h * (kGrpMaxV / 2) + i) * T < total) *reinterpret_cast<f32x4*>(grp_lds + r * ld_lds + 4 * q) = v[i];
                    r += dr; q += dq;
                    if (q >= c4) { q -= c4; ++r; }
                }
            }
        }
        __syncthreads();
        const bool full = p0 + kGrpRows <= P;
#pragma unroll 1
        for (int mt = 0; mt < kGrpRows / 16; ++mt) {
            f32x4 acc[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
            const float* arow = grp_lds + (mt * 16 + ln) * ld_lds + g * K + lk;
            float a[KSTEPS];
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks) a[ks] = arow[(EXACT || 4 * ks + lk < K) ? 4 * ks : 0];    // (a clamped read meets a zero weight)
#pragma unroll
            for (int ks = 0; ks < KSTEPS; ++ks)
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], wreg[nt][ks], acc[nt], 0, 0, 0);
            // C layout: lane holds rows 4 * lk + {0..3} of column nt * 16 + ln; written over the wave's own input columns
            float* orow = grp_lds + (mt * 16 + 4 * lk) * ld_lds + g * N + ln;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                if (nt * 16 + ln < N) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) orow[r * ld_lds + nt * 16] = acc[nt][r];
                }
                if (EPI == 1) {                                             // four rows in fp32, then float64 per lane
                    f32x4 t = acc[nt];
                    if (!full) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) t[r] = (p0 + mt * 16 + 4 * lk + r < P) ? t[r] : 0.f;
                    }
                    const float u = (t[0] + t[1]) + (t[2] + t[3]);
                    const float q2 = fmaf(t[3], t[3], fmaf(t[2], t[2], fmaf(t[1], t[1], t[0] * t[0])));
                    s1[nt] += (double)u;
                    s2[nt] += (double)q2;
                }
            }
        }
        __syncthreads();                                                    // every group's columns are in place
        {
            int r = r0, q = q0;
#pragma unroll
            for (int i = 0; i < kGrpMaxV; ++i) {
                if (tid + i * T < total && p0 + r < P)
                    *reinterpret_cast<f32x4*>(y + (size_t)(p0 + r) * ldy + 4 * q) = *reinterpret_cast<const f32x4*>(grp_lds + r * ld_lds + 4 * q);
                r += dr; q += dq;
                if (q >= c4) { q -= c4; ++r; }
            }
        }
    }
    if (EPI == 1) {
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {                                   // the rows held by the other three lane groups
            double u = s1[nt], q2 = s2[nt];
            u += __shfl_xor(u, 16); q2 += __shfl_xor(q2, 16);
            u += __shfl_xor(u, 32); q2 += __shfl_xor(q2, 32);
            if (lk == 0 && nt * 16 + ln < N) partial[(size_t)(g * N + nt * 16 + ln) * gridDim.x + blockIdx.x] = make_double2(u, q2);
        }
    }
}

struct GrpPlan { int ok, nt, ksteps, grid, slabs, ld; size_t lds; };

// groups of N == K <= 64 channels, 2 ... 8 groups (one wave each), enough rows to give every CU several slabs
GrpPlan plan_rows_grp(int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy) {
    GrpPlan q{};
    const char* env = getenv("NEXTOU_PW_GRP");          // read per call (tests / A-B)
    if (env && atoi(env) == 0) return q;
    const int C = groups * K;
    if (groups < 2 || groups > 8 || N != K || K % 4 != 0 || K > 64 || ldx < C || ldy < C || P < 64 * 2 * (int64_t)cu_count() ||
        P > 0x7fffffff)
        return q;
    q.nt = K == 44 ? 3 : 4;
    q.ksteps = K == 44 ? 11 : 16;
    q.ld = C + ((C % 8 == 4) ? 0 : 4);                  // row stride = 4 mod 8 floats
    q.lds = (size_t)kGrpRows * q.ld * sizeof(float);
    if (q.lds > 150 * 1024) return q;
    if ((kGrpRows * (C / 4) + 64 * groups - 1) / (64 * groups) > kGrpMaxV) return q;      // pieces of a slab per thread
    q.slabs = (int)((P + kGrpRows - 1) / kGrpRows);
    const int per_cu = (int)(160 * 1024 / q.lds) < 2 ? 1 : 2;
    const int max_grid = cu_count() * per_cu;
    q.grid = q.slabs <= max_grid ? q.slabs : (q.slabs + (q.slabs + max_grid - 1) / max_grid - 1) / ((q.slabs + max_grid - 1) / max_grid);
    q.ok = 1;
    return q;
}

template <int EPI>
int dispatch_rows_grp(const GrpPlan& q, const float* x, const float* w, float* y, int P, int N, int K, int groups, long ldx, long ldy,
                      double2* partial, hipStream_t s) {
#define NEXTOU_GRP_LAUNCH(NT, KS, EX)                                                                                                \
    do {                                                                                                                           \
        if (q.lds > 64 * 1024)                                                                                                     \
            (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&pw_rows_grp_kernel<NT, KS, EX, EPI>),                         \
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)q.lds);                                     \
        hipLaunchKernelGGL((pw_rows_grp_kernel<NT, KS, EX, EPI>), dim3(q.grid), dim3(64 * groups), q.lds, s, x, w, y, P, N, K, groups, \
                           ldx, ldy, q.ld, q.slabs, partial);                                                                      \
    } while (0)
    if (q.nt == 3 && q.ksteps == 11) NEXTOU_GRP_LAUNCH(3, 11, true);
    else NEXTOU_GRP_LAUNCH(4, 16, false);
#undef NEXTOU_GRP_LAUNCH
    return check_launch("pw_rows_grp_kernel");
}

}  // namespace
}  // namespace nextou

using namespace nextou;

extern "C" int nextou_pw_rows(const float* x, const float* w, const float* bias, float* y, int64_t P, int N, int K, int groups, int64_t ldx,
                              int64_t ldy, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && w && y, "pw_rows: null pointer");
    if (int e = check_pw("pw_rows", P, N, K, groups, ldx, ldy, K, N)) return e;
    NEXTOU_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && aligned16(x) && aligned16(w),
                   "pw_rows: K=%d and ldx=%lld must be multiples of 4 and x, w 16-byte aligned", K, (long long)ldx);
    const RowsPlan q = plan_rows((int)P, N, groups, true);
    const int vec_store = (N % 4 == 0 && ldy % 4 == 0 && aligned16(y)) ? 1 : 0;
    hipStream_t s = (hipStream_t)stream;
    if (bias == nullptr && vec_store && groups > 1) {
        const GrpPlan gp = plan_rows_grp(P, N, K, groups, ldx, ldy);
        if (gp.ok) {
            ProfScope prof(s, kBoundHbm, 8.0 * (double)P * groups * K, "pw_rows_grp_kernel<%d,%d|plain>[P%lld N%d K%d g%d]", gp.nt, gp.ksteps,
                           (long long)P, N, K, groups);
            return dispatch_rows_grp<0>(gp, x, w, y, (int)P, N, K, groups, (long)ldx, (long)ldy, nullptr, s);
        }
    }
    if (bias == nullptr && vec_store) {
        const KsPlan ks = plan_rows_ks(P, N, K, groups, ldx, false, 0);
        if (ks.ok) {
            ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_rows_ks_kernel<plain>[P%lld N%d K%d]", (long long)P, N, K);
            return launch_rows_ks<0, 0>(ks, x, w, y, (int)P, N, (long)ldx, (long)ldy, PwFuse{}, s);
        }
    }
    if (bias == nullptr && vec_store && ldx == K) {
        const SwPlan sw = plan_rows_sw(P, N, K, groups, false);
        if (sw.cfg >= 0) {
            ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_rows_sw_kernel<%d,%d|plain>[P%lld N%d K%d]", sw.tnw, sw.nw, (long long)P, N, K);
            return dispatch_rows_sw<0, 0>(sw, x, w, y, (int)P, N, K, (long)ldx, (long)ldy, PwFuse{}, s);
        }
    }
    ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K * groups, "pw_rows_kernel<%d,%d>[P%lld N%d K%d g%d]", q.tm, q.tn, (long long)P, N, K,
                   groups);
    return dispatch_rows<0, 0>(q, x, w, bias, y, (int)P, N, K, groups, (long)ldx, (long)ldy, vec_store, PwFuse{}, s);
}

// nextou_pw_rows for a transposed convolution whose kernel equals its stride, writing straight into the concatenation buffer: see PwFuse::up_cout.
extern "C" int nextou_pw_rows_up(const float* x, const float* w, const float* bias, float* out, int N, int K, int64_t ldx, int64_t ld_out, int B,
                                 int D, int H, int W, int sd, int sh, int sw, int cout, nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && w && out, "pw_rows_up: null pointer");
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && (long long)B * D * H * W < 0x7fffffffLL, "pw_rows_up: bad input volume");
    const int64_t P = (int64_t)B * D * H * W;
    auto pow2_le4 = [](int v) { return v == 1 || v == 2 || v == 4; };
    NEXTOU_REQUIRE(pow2_le4(sd) && pow2_le4(sh) && pow2_le4(sw), "pw_rows_up: strides must be 1, 2 or 4 (got %d, %d, %d)", sd, sh, sw);
    NEXTOU_REQUIRE(cout > 0 && cout % 4 == 0 && N == cout * sd * sh * sw && N < (1 << 20), "pw_rows_up: N = %d is not cout (%d, a multiple of 4) x taps", N, cout);
    if (int e = check_pw("pw_rows_up", P, N, K, 1, ldx, N, K, N)) return e;
    NEXTOU_REQUIRE(K % 4 == 0 && ldx % 4 == 0 && ld_out % 4 == 0 && ld_out >= cout && aligned16(x) && aligned16(w) && aligned16(out) &&
                   (!bias || aligned16(bias)), "pw_rows_up: K, ldx, ld_out multiples of 4 and 16-byte aligned tensors");
    const RowsPlan q = plan_rows((int)P, N, 1, false);
    hipStream_t s = (hipStream_t)stream;
    PwFuse fz{};
    fz.up_cout = cout;
    fz.up = UpShuffle{D * sd, H * sh, W * sw, sd, sh, sw};
    fz.up_bias = bias;
    ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_rows_kernel<%d,%d|up>[P%lld N%d K%d s%dx%dx%d]", q.tm, q.tn, (long long)P, N, K, sd, sh, sw);
    return dispatch_rows<0, 0>(q, x, w, nullptr, out, (int)P, N, K, 1, (long)ldx, (long)ld_out, 1, fz, s);
}

// Which kernel a fused launch takes and how many statistics partials per channel it writes — ONE function for the launch and for the
// size query (the query assumed dense rows and no prologue while the launch re-planned with the caller's strides, so a strided
// caller of the C-ABI was told the stationary kernel's count and got pw_rows_kernel's many more partials written past its buffer).
struct FusedChoice { int kind, tiles; GrpPlan gp; SwPlan sw; RowsPlan q; KsPlan ks; };     // kind: 0 grouped rows, 1 stationary weights, 2 LDS-tiled, 3 K-split
static FusedChoice choose_rows_fused(int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy, bool pro, int epi) {
    FusedChoice c{};
    c.q = plan_rows((int)P, N, groups);
    if (!pro && epi != 2 && groups > 1) {
        c.gp = plan_rows_grp(P, N, K, groups, ldx, ldy);
        if (c.gp.ok) { c.kind = 0; c.tiles = c.gp.grid; return c; }
    }
    c.ks = plan_rows_ks(P, N, K, groups, ldx, pro, epi);
    if (c.ks.ok) { c.kind = 3; c.tiles = c.ks.grid; return c; }
    c.sw = ldx == (int64_t)groups * K ? plan_rows_sw(P, N, K, groups, pro) : SwPlan{-1, 0, 0, 0, 0, 0, 0};
    if (c.sw.cfg >= 0) { c.kind = 1; c.tiles = c.sw.grid; return c; }
    c.kind = 2;
    c.tiles = c.q.nb_p;
    return c;
}

extern "C" int nextou_pw_rows_tiles(int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy, int prologue, int grad_stats) {
    if (P <= 0 || N <= 0 || K <= 0 || groups <= 0) return 0;
    return choose_rows_fused(P, N, K, groups, ldx, ldy, prologue != 0, grad_stats ? 2 : 1).tiles;
}

extern "C" int nextou_pw_rows_fused(const float* x, const float* w, float* y, int64_t P, int N, int K, int groups, int64_t ldx, int64_t ldy,
                                    const float* pro_scale, const float* pro_shift, float pro_slope, double* stats_partial, int stats_tiles,
                                    const float* bwd_h, int64_t ldh, const float* bwd_weight,
                                    const float* bwd_bias, const float* bwd_mean, const float* bwd_invstd, float bwd_slope,
                                    nextou_stream_t stream) {
    NEXTOU_REQUIRE(x && w && y, "pw_rows_fused: null pointer");
    if (int e = check_pw("pw_rows_fused", P, N, K, groups, ldx, ldy, K, N)) return e;
    NEXTOU_REQUIRE(K % 4 == 0 && N % 4 == 0 && ldx % 4 == 0 && ldy % 4 == 0 && aligned16(x) && aligned16(w) && aligned16(y),
                   "pw_rows_fused: K=%d, N=%d and the row strides must be multiples of 4 and x, w, y 16-byte aligned", K, N);
    const bool pro = pro_scale != nullptr;
    NEXTOU_REQUIRE(!pro || (pro_shift && aligned16(pro_scale) && aligned16(pro_shift)),
                   "pw_rows_fused: the operand prologue needs scale and shift (16-byte aligned)");
    const int epi = bwd_h ? 2 : (stats_partial ? 1 : 0);
    NEXTOU_REQUIRE(epi != 2 || (stats_partial && bwd_weight && bwd_bias && bwd_mean && bwd_invstd && ldh % 4 == 0 && aligned16(bwd_h) &&
                                ldh >= (int64_t)groups * N && aligned16(bwd_weight) && aligned16(bwd_bias) && aligned16(bwd_mean) &&
                                aligned16(bwd_invstd)),
                   "pw_rows_fused: the gradient-statistics epilogue needs partials, h (row stride %lld) and the norm's four vectors", (long long)ldh);
    NEXTOU_REQUIRE(!(pro && epi == 2), "pw_rows_fused: prologue + gradient-statistics epilogue is not an instantiated combination");
    const FusedChoice ch = choose_rows_fused(P, N, K, groups, ldx, ldy, pro, epi);
    NEXTOU_REQUIRE(epi == 0 || stats_tiles == ch.tiles,
                   "pw_rows_fused: stats_partial was sized for %d partials per channel, this launch writes %d — size it with "
                   "nextou_pw_rows_tiles(P, N, K, groups, ldx, ldy, prologue, grad_stats) of the SAME arguments", stats_tiles, ch.tiles);
    const RowsPlan& q = ch.q;
    PwFuse fz{};
    fz.pro_scale = pro_scale; fz.pro_shift = pro_shift; fz.pro_slope = pro_slope;
    fz.partial = reinterpret_cast<double2*>(stats_partial);
    fz.h = bwd_h; fz.ldh = (long)ldh; fz.epi_w = bwd_weight; fz.epi_b = bwd_bias; fz.epi_mean = bwd_mean; fz.epi_invstd = bwd_invstd;
    fz.epi_slope = bwd_slope;
    hipStream_t s = (hipStream_t)stream;
    const int Pi = (int)P;
    if (ch.kind == 0) {
        const GrpPlan& gp = ch.gp;
        ProfScope prof(s, kBoundHbm, 8.0 * (double)P * groups * K, "pw_rows_grp_kernel<%d,%d|%s>[P%lld N%d K%d g%d]", gp.nt, gp.ksteps,
                       epi == 1 ? "stats" : "plain", (long long)P, N, K, groups);
        if (epi == 1) return dispatch_rows_grp<1>(gp, x, w, y, Pi, N, K, groups, (long)ldx, (long)ldy, fz.partial, s);
        return dispatch_rows_grp<0>(gp, x, w, y, Pi, N, K, groups, (long)ldx, (long)ldy, nullptr, s);
    }
    if (ch.kind == 3) {
        ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_rows_ks_kernel<%s%s>[P%lld N%d K%d]", pro ? "norm-act," : "", epi == 1 ? "stats" : "plain",
                       (long long)P, N, K);
        if (!pro && epi == 1) return launch_rows_ks<0, 1>(ch.ks, x, w, y, Pi, N, (long)ldx, (long)ldy, fz, s);
        if (pro && epi == 1) return launch_rows_ks<1, 1>(ch.ks, x, w, y, Pi, N, (long)ldx, (long)ldy, fz, s);
        if (pro && epi == 0) return launch_rows_ks<1, 0>(ch.ks, x, w, y, Pi, N, (long)ldx, (long)ldy, fz, s);
        return launch_rows_ks<0, 0>(ch.ks, x, w, y, Pi, N, (long)ldx, (long)ldy, fz, s);
    }
    if (ch.kind == 1) {
        const SwPlan& sw = ch.sw;
        ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_rows_sw_kernel<%d,%d|%s%s>[P%lld N%d K%d]", sw.tnw, sw.nw, pro ? "norm-act," : "",
                       epi == 2 ? "grad-stats" : (epi == 1 ? "stats" : "plain"), (long long)P, N, K);
        if (!pro && epi == 1) return dispatch_rows_sw<0, 1>(sw, x, w, y, Pi, N, K, (long)ldx, (long)ldy, fz, s);
        if (pro && epi == 1) return dispatch_rows_sw<1, 1>(sw, x, w, y, Pi, N, K, (long)ldx, (long)ldy, fz, s);
        if (pro && epi == 0) return dispatch_rows_sw<1, 0>(sw, x, w, y, Pi, N, K, (long)ldx, (long)ldy, fz, s);
        if (!pro && epi == 2) return dispatch_rows_sw<0, 2>(sw, x, w, y, Pi, N, K, (long)ldx, (long)ldy, fz, s);
        return dispatch_rows_sw<0, 0>(sw, x, w, y, Pi, N, K, (long)ldx, (long)ldy, fz, s);
    }
    ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K * groups, "pw_rows_kernel<%d,%d|%s%s>[P%lld N%d K%d g%d]", q.tm, q.tn,
                   pro ? "norm-act," : "", epi == 2 ? "grad-stats" : (epi == 1 ? "stats" : "plain"), (long long)P, N, K, groups);
    if (!pro && epi == 1) return dispatch_rows<0, 1>(q, x, w, nullptr, y, Pi, N, K, groups, (long)ldx, (long)ldy, 1, fz, s);
    if (pro && epi == 1) return dispatch_rows<1, 1>(q, x, w, nullptr, y, Pi, N, K, groups, (long)ldx, (long)ldy, 1, fz, s);
    if (pro && epi == 0) return dispatch_rows<1, 0>(q, x, w, nullptr, y, Pi, N, K, groups, (long)ldx, (long)ldy, 1, fz, s);
    if (!pro && epi == 2) return dispatch_rows<0, 2>(q, x, w, nullptr, y, Pi, N, K, groups, (long)ldx, (long)ldy, 1, fz, s);
    return dispatch_rows<0, 0>(q, x, w, nullptr, y, Pi, N, K, groups, (long)ldx, (long)ldy, 1, fz, s);
}

// ---- stationary-output weight gradient: the instantiated shapes: X(id, N, K, A, BT, WN, WK)
#define NEXTOU_SO_CONFIGS(X) \
    X(0, 528, 132, 3, 9, 11, 1) /* FFN fc1: gy 528 wide, x 132                       */ \
    X(1, 132, 528, 9, 3, 1, 11) /* FFN fc2: gy 132 wide, x (hidden, prologue) 528    */ \
    X(2, 132, 132, 1, 9, 9, 1)  /* the graphers' fc1                                  */ \
    X(3, 132, 264, 1, 17, 9, 1) /* the graphers' fc2: gy 132 wide, x 264              */

struct SoPlan { int cfg, grid, slabs, slabs_per_wg; size_t lds; };

SoPlan plan_wgrad_so(int64_t P, int N, int K, int groups, int64_t ldg, int64_t ldx, bool pro) {
    SoPlan q{};
    q.cfg = -1;
    const char* env = getenv("NEXTOU_PW_SO");
    if ((env && atoi(env) == 0) || groups != 1 || ldg != N || ldx != K || P < 64 * 4 * (int64_t)cu_count()) return q;
    int a = 0, bt = 0, wn = 0, wk = 0;
#define X(id, n, k, A_, BT_, WN_, WK_) if (N == n && K == k) { q.cfg = id; a = A_; bt = BT_; wn = WN_; wk = WK_; }
    NEXTOU_SO_CONFIGS(X)
#undef X
    if (q.cfg < 0) return q;
    q.slabs = (int)((P + kSoPS - 1) / kSoPS);
    q.grid = q.slabs < cu_count() ? q.slabs : cu_count();
    q.slabs_per_wg = (q.slabs + q.grid - 1) / q.grid;
    q.grid = (q.slabs + q.slabs_per_wg - 1) / q.slabs_per_wg;
    const int nt = 64 * wn * wk, n4 = N / 4, k4 = K / 4;
    const int nvg = (kSoPS * n4 + nt - 1) / nt, nvx = (kSoPS * k4 + nt - 1) / nt;
    const int rg = (nvg * nt + n4 - 1) / n4, rx = (nvx * nt + k4 - 1) / k4;
    q.lds = (size_t)2 * (rg * wgrad_stride(wn * a * 16) + rx * wgrad_stride(wk * bt * 16)) * sizeof(float) + (pro ? (size_t)2 * K * sizeof(float) : 0);
    return q;
}

template <int N, int K, int A, int BT, int WN, int WK, int PRO>
int launch_wgrad_so(const SoPlan& q, const float* gy, const float* x, float* part, int P, long ldg, long ldx, const PwFuse& fz, hipStream_t s) {
    static size_t allowed = 0;
    if (q.lds > allowed) {
        if (int e = allow_lds(pw_wgrad_so_kernel<N, K, A, BT, WN, WK, PRO>, q.lds)) return e;
        allowed = q.lds;
    }
    hipLaunchKernelGGL((pw_wgrad_so_kernel<N, K, A, BT, WN, WK, PRO>), dim3(q.grid), dim3(64 * WN * WK), q.lds, s, gy, x, part, P, ldg, ldx, q.slabs,
                       q.slabs_per_wg, fz);
    return check_launch("pw_wgrad_so_kernel");
}

extern "C" int nextou_pw_wgrad_workspace(int64_t P, int N, int K, int groups, size_t* bytes) {
    NEXTOU_REQUIRE(bytes, "pw_wgrad_workspace: null pointer");
    if (int e = check_pw("pw_wgrad_workspace", P, N, K, groups, (int64_t)groups * N, (int64_t)groups * K, N, K)) return e;
    const WgradPlan q = plan_wgrad((int)P, N, K, groups);
    *bytes = (size_t)q.splits * groups * N * K * sizeof(float);
    const SoPlan so = plan_wgrad_so(P, N, K, groups, (int64_t)groups * N, (int64_t)groups * K, false);      // either kernel may take the call
    if (so.cfg >= 0 && (size_t)so.grid * N * K * sizeof(float) > *bytes) *bytes = (size_t)so.grid * N * K * sizeof(float);
    return 0;
}

static int pw_wgrad_impl(const float* gy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int64_t P, int N, int K,
                         int groups, int64_t ldg, int64_t ldx, int accumulate, const PwFuse* fz, nextou_stream_t stream) {
    NEXTOU_REQUIRE(gy && x && dw && workspace, "pw_wgrad: null pointer");
    if (int e = check_pw("pw_wgrad", P, N, K, groups, ldg, ldx, N, K)) return e;
    const SoPlan so = plan_wgrad_so(P, N, K, groups, ldg, ldx, fz != nullptr);
    if (so.cfg >= 0 && aligned16(gy) && aligned16(x)) {
        const size_t need_so = (size_t)so.grid * N * K * sizeof(float);
        NEXTOU_REQUIRE(workspace_bytes >= need_so, "pw_wgrad: workspace %zu B < %zu B (nextou_pw_wgrad_workspace)", workspace_bytes, need_so);
        hipStream_t s = (hipStream_t)stream;
        int rc = NEXTOU_EINVAL;
        {
            ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K, "pw_wgrad_so_kernel[P%lld N%d K%d%s WG%d]", (long long)P, N, K, fz ? "|norm-act" : "",
                           so.grid);
#define X(id, n, k, A_, BT_, WN_, WK_)                                                                                                     \
            if (so.cfg == id)                                                                                                              \
                rc = fz ? launch_wgrad_so<n, k, A_, BT_, WN_, WK_, 1>(so, gy, x, workspace, (int)P, (long)ldg, (long)ldx, *fz, s)          \
                        : launch_wgrad_so<n, k, A_, BT_, WN_, WK_, 0>(so, gy, x, workspace, (int)P, (long)ldg, (long)ldx, PwFuse{}, s);
            NEXTOU_SO_CONFIGS(X)
#undef X
        }
        if (rc) return rc;
        const long elems = (long)N * K;
        ProfScope prof(s, kBoundHbm, 4.0 * elems * (so.grid + 1), "pw_wgrad_reduce_kernel[%ld x S%d]", elems, so.grid);
        hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, s, workspace, dw, elems, so.grid, accumulate);
        return check_launch("pw_wgrad_reduce_kernel");
    }
    const WgradPlan q = plan_wgrad((int)P, N, K, groups);
    const size_t need = (size_t)q.splits * groups * N * K * sizeof(float);
    NEXTOU_REQUIRE(workspace_bytes >= need, "pw_wgrad: workspace %zu B < %zu B (nextou_pw_wgrad_workspace)", workspace_bytes, need);
    NEXTOU_REQUIRE(N % 4 == 0 && K % 4 == 0 && ldg % 4 == 0 && ldx % 4 == 0 && aligned16(gy) && aligned16(x),
                   "pw_wgrad: N=%d, K=%d and the row strides must be multiples of 4 and gy, x 16-byte aligned", N, K);
    hipStream_t s = (hipStream_t)stream;
    int rc = NEXTOU_EINVAL;
    {
        ProfScope prof(s, kBoundMfma, 2.0 * (double)P * N * K * groups, "pw_wgrad_kernel<%d,%d|%dx%d%s>[P%lld N%d K%d g%d S%d]", q.tn, q.tk, q.wn,
                       q.wk, fz ? "|norm-act" : "", (long long)P, N, K, groups, q.splits);
        int id = 0;
#define X(a, b, c, d)                                                                                                              \
        if (q.cfg == id++)                                                                                                         \
            rc = fz ? launch_wgrad<a, b, c, d, 1>(q, gy, x, workspace, (int)P, N, K, groups, (long)ldg, (long)ldx, *fz, s)         \
                    : launch_wgrad<a, b, c, d, 0>(q, gy, x, workspace, (int)P, N, K, groups, (long)ldg, (long)ldx, PwFuse{}, s);
        NEXTOU_WGRAD_CONFIGS(X)
#undef X
    }
    if (rc) return rc;
    const long elems = (long)groups * N * K;
    ProfScope prof(s, kBoundHbm, 4.0 * elems * (q.splits + 1), "pw_wgrad_reduce_kernel[%ld x S%d]", elems, q.splits);
    hipLaunchKernelGGL(pw_wgrad_reduce_kernel, dim3((unsigned)((elems + 63) / 64)), dim3(256), 0, s, workspace, dw, elems, q.splits, accumulate);
    return check_launch("pw_wgrad_reduce_kernel");
}

extern "C" int nextou_pw_wgrad(const float* gy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int64_t P, int N, int K,
                               int groups, int64_t ldg, int64_t ldx, int accumulate, nextou_stream_t stream) {
    return pw_wgrad_impl(gy, x, dw, workspace, workspace_bytes, P, N, K, groups, ldg, ldx, accumulate, nullptr, stream);
}

extern "C" int nextou_pw_wgrad_fused(const float* gy, const float* x, float* dw, float* workspace, size_t workspace_bytes, int64_t P, int N,
                                     int K, int groups, int64_t ldg, int64_t ldx, int accumulate, const float* pro_scale,
                                     const float* pro_shift, float pro_slope, nextou_stream_t stream) {
    NEXTOU_REQUIRE(pro_scale && pro_shift, "pw_wgrad_fused: the operand prologue needs scale and shift");
    PwFuse fz{};
    fz.pro_scale = pro_scale; fz.pro_shift = pro_shift; fz.pro_slope = pro_slope;
    return pw_wgrad_impl(gy, x, dw, workspace, workspace_bytes, P, N, K, groups, ldg, ldx, accumulate, &fz, stream);
}
