// K5 — BTI critical-voxel map for gfx950.
//
// The reference (loss/bti_loss.py:76-117) runs, per interaction, two isin() masks, two float64
// conv3d with an all-ones 3x3x3 kernel used as a binary dilation, three where() thresholds and
// a logical_or — 24 float64 convolutions per deep-supervision scale for the Synapse list.  The
// whole loop is integer/bit logic: with one bit per interaction,
//     a = lut_a[label], c = lut_c[label]
//     critical = ((OR_nbhd c) & a) | ((OR_nbhd a) & c) != 0
// which is one pass over the label volume.  HBM-bound: 1 B read + 1 B written per voxel for the
// map, 4*L B per voxel for the arg-max that produces the labels.
#include "common.h"
#include <cstdlib>

namespace nextou {

// exp(t), t <= 0, in double by a fixed fma sequence — the same sequence as oracle/nextou_oracle.c::oracle_exp_neg, so that the rare
// near-tie voxels below resolve to the same label on both sides bit for bit (IEEE fma / rint / ldexp only).
__device__ __forceinline__ double exp_neg_f64(double t) {
    if (t < -110.0) return 0.0;
    const double k = rint(t * 1.4426950408889634);
    double r = fma(-k, 0.693147180369123816490, t);
    r = fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;
    p = fma(p, r, 1.0 / 479001600.0);
    p = fma(p, r, 1.0 / 39916800.0);
    p = fma(p, r, 1.0 / 3628800.0);
    p = fma(p, r, 1.0 / 362880.0);
    p = fma(p, r, 1.0 / 40320.0);
    p = fma(p, r, 1.0 / 5040.0);
    p = fma(p, r, 1.0 / 720.0);
    p = fma(p, r, 1.0 / 120.0);
    p = fma(p, r, 1.0 / 24.0);
    p = fma(p, r, 1.0 / 6.0);
    p = fma(p, r, 0.5);
    p = fma(p, r, 1.0);
    p = fma(p, r, 1.0);
    return ldexp(p, (int)k);
}

// The reference labels a voxel with argmax(softmax(x)) (bti_loss.py:132-134).  In float32 the softmax values of two logits closer than
// ~2.4e-7 can round to the same number and torch.argmax returns the FIRST: the earlier class wins although its logit is the smaller one.
// Slow path for a voxel whose arg-max `arg` replaced a running maximum within 2^-21 of it (about one voxel in 1e6): the canonical
// restatement — e_k = float32(exp(x_k - m)), s = e_0 + e_1 + ... in class order, q_k = e_k / s, first k with q_k == q_max.
__device__ __noinline__ int softmax_first_tie(const float* __restrict__ x, long long stride, int L, float m, int arg) {
    float s = 0.0f;
    for (int l = 0; l < L; ++l) s = s + (float)exp_neg_f64((double)(x[(size_t)l * stride] - m));
    const float qmax = 1.0f / s;
    for (int l = 0; l < arg; ++l) {
        const float d = x[(size_t)l * stride] - m;
        if (d < -0x1p-21f) continue;
        const float e = (float)exp_neg_f64((double)d);
        if (e / s == qmax) return l;
    }
    return arg;
}

// labels[b, v] = argmax_l softmax(logits[b, :, v]) with torch's first-index rule on equal softmax values (see above); for all but
// ~1e-6 of the voxels that is the first arg-max of the logits.  `near` = the running maximum the final one replaced lay within 2^-21
// of it (only then can an earlier class tie: running maxima increase, so the last replaced one is the closest earlier logit).
// One thread per 4 consecutive voxels (16-B loads per class plane) when V % 4 == 0.
template <bool VEC4>
__global__ __launch_bounds__(256) void argmax_labels_kernel(const float* __restrict__ logits,
                                                            uint8_t* __restrict__ labels, int L,
                                                            long long V) {
    const int b = blockIdx.y;
    const float* lb = logits + (size_t)b * L * V;
    uint8_t* ob = labels + (size_t)b * V;
    const long long stride = (long long)gridDim.x * blockDim.x;
    constexpr float kBand = 0x1p-21f;
    if (VEC4) {
        const long long V4 = V >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V4; i += stride) {
            float4 best = reinterpret_cast<const float4*>(lb)[i];
            uchar4 arg = make_uchar4(0, 0, 0, 0);
            bool nx = false, ny = false, nz = false, nw = false;
            for (int l = 1; l < L; ++l) {
                const float4 v = reinterpret_cast<const float4*>(lb + (size_t)l * V)[i];
                if (v.x > best.x) { nx = v.x - best.x <= kBand; best.x = v.x; arg.x = (uint8_t)l; }
                if (v.y > best.y) { ny = v.y - best.y <= kBand; best.y = v.y; arg.y = (uint8_t)l; }
                if (v.z > best.z) { nz = v.z - best.z <= kBand; best.z = v.z; arg.z = (uint8_t)l; }
                if (v.w > best.w) { nw = v.w - best.w <= kBand; best.w = v.w; arg.w = (uint8_t)l; }
            }
            if (nx | ny | nz | nw) {
                const float* p = lb + 4 * i;
                if (nx) arg.x = (uint8_t)softmax_first_tie(p + 0, V, L, best.x, arg.x);
                if (ny) arg.y = (uint8_t)softmax_first_tie(p + 1, V, L, best.y, arg.y);
                if (nz) arg.z = (uint8_t)softmax_first_tie(p + 2, V, L, best.z, arg.z);
                if (nw) arg.w = (uint8_t)softmax_first_tie(p + 3, V, L, best.w, arg.w);
            }
            reinterpret_cast<uchar4*>(ob)[i] = arg;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
            float best = lb[i];
            uint8_t arg = 0;
            bool near = false;
            for (int l = 1; l < L; ++l) {
                const float v = lb[(size_t)l * V + i];
                if (v > best) { near = v - best <= kBand; best = v; arg = (uint8_t)l; }
            }
            if (near) arg = (uint8_t)softmax_first_tie(lb + i, V, L, best, arg);
            ob[i] = arg;
        }
    }
}

// The same labels from channels-last logits: element (row, l) at row * L + l — what the network's heads emit (the planes kernel above
// needed a (B, L, V) copy of them: 308 MB each way at cfg 4's full resolution).  A lane owns one voxel and reads its row as 8-byte pairs
// (rows of a wave are one contiguous 64 * L * 4-byte run, served out of L1 between the L / 2 loads — ce_mean_fwd_kernel's pattern).
template <int LMAX>
__global__ __launch_bounds__(256) void argmax_labels_rows_kernel(const float* __restrict__ logits, uint8_t* __restrict__ labels, int L,
                                                                 long long rows) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    constexpr float kBand = 0x1p-21f;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < rows; p += stride) {
        const float* base = logits + (size_t)p * L;
        float x[LMAX];
#pragma unroll
        for (int l = 0; l < LMAX; l += 2) {
            float2 t = make_float2(-INFINITY, -INFINITY);
            if (l + 1 < L) t = *reinterpret_cast<const float2*>(base + l);
            else if (l < L) t.x = base[l];
            x[l] = t.x;
            if (l + 1 < LMAX) x[l + 1] = t.y;
        }
        float best = x[0];
        int arg = 0;
        bool near = false;
#pragma unroll
        for (int l = 1; l < LMAX; ++l)
            if (l < L && x[l] > best) { near = x[l] - best <= kBand; best = x[l]; arg = l; }
        if (near) arg = softmax_first_tie(base, 1, L, best, arg);
        labels[p] = (uint8_t)arg;
    }
}

// Target label maps as the uint8 volumes the K5 kernels read, with the range check of the reference's CrossEntropyLoss (which raises
// for a target outside [0, L), bti_loss.py:141) done on the way: ONE pass over the target instead of ATen's aminmax + .to(uint8), and a
// device-side flag instead of a host read, so the (B)TI losses validate their targets inside a captured hipGraph step as well.
// SRC: 0 float32, 1 int64, 2 uint8.  flag[0] |= 1 when any value is outside [0, L) (integer atomic: deterministic).
template <int SRC>
__global__ __launch_bounds__(256) void labels_u8_kernel(const void* __restrict__ src, uint8_t* __restrict__ out, long long n, int L,
                                                        unsigned* __restrict__ flag) {
    bool bad = false;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        long long v;
        if (SRC == 0) {
            const float f = static_cast<const float*>(src)[i];
            v = (long long)f;                                   // truncation, as Tensor.long() / .to(uint8) do
            bad |= !(f > -1.0f && f < (float)L);                // NaN and everything that truncates outside [0, L)
        } else if (SRC == 1) {
            v = static_cast<const long long*>(src)[i];
            bad |= v < 0 || v >= L;
        } else {
            v = static_cast<const uint8_t*>(src)[i];
            bad |= v >= L;
        }
        out[i] = (uint8_t)v;
    }
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1u);
}

// Round-1 kernel, kept as the fallback for min_thick > 3: one thread per voxel; the two LUTs live in LDS; neighbour labels come
// through L1/L2 (each label byte is touched by at most 27 threads of neighbouring rows).  81 memory instructions per voxel at
// connectivity 26: 96 us for the 2 x 64 x 224 x 192 volume of cfg 4 = 1.4 % of the HBM roofline (profiles/r03_kernel_bench_k5_start.md).
// full_box: box neighbourhood of radius `rad` (connectivity 26 / 8); otherwise the 6 / 4 cross.
__global__ __launch_bounds__(256) void bti_critical_naive_kernel(
    const uint8_t* __restrict__ labels, const uint32_t* __restrict__ lut_a,
    const uint32_t* __restrict__ lut_c, int n_labels, uint8_t* __restrict__ critical, int D, int H,
    int W, int full_box, int rad) {
    __shared__ uint32_t la[256], lc[256];
    for (int i = threadIdx.x; i < 256; i += blockDim.x) {
        la[i] = i < n_labels ? lut_a[i] : 0u;
        lc[i] = i < n_labels ? lut_c[i] : 0u;
    }
    __syncthreads();
    const int b = blockIdx.z;
    const long long HW = (long long)H * W;
    const long long V = HW * D;
    const uint8_t* lb = labels + (size_t)b * V;
    const long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const int w = (int)(v % W);
    const int hh = (int)((v / W) % H);
    const int d = (int)(v / HW);
    const uint8_t self = lb[v];
    const uint32_t a = la[self], c = lc[self];
    uint32_t na = a, nc = c;
    if (full_box) {
        for (int dz = -rad; dz <= rad; ++dz) {
            const int z = d + dz;
            if (z < 0 || z >= D) continue;
            for (int dy = -rad; dy <= rad; ++dy) {
                const int y = hh + dy;
                if (y < 0 || y >= H) continue;
                const uint8_t* row = lb + (size_t)z * HW + (size_t)y * W;
                for (int dx = -rad; dx <= rad; ++dx) {
                    const int xx = w + dx;
                    if (xx < 0 || xx >= W) continue;
                    const uint8_t l = row[xx];
                    na |= la[l];
                    nc |= lc[l];
                }
            }
        }
    } else {
#define NEXTOU_TAP(cond, off)                      \
    if (cond) {                                    \
        const uint8_t l = lb[v + (off)];           \
        na |= la[l];                               \
        nc |= lc[l];                               \
    }
        NEXTOU_TAP(w > 0, -1)
        NEXTOU_TAP(w < W - 1, 1)
        NEXTOU_TAP(hh > 0, -(long long)W)
        NEXTOU_TAP(hh < H - 1, (long long)W)
        NEXTOU_TAP(d > 0, -HW)
        NEXTOU_TAP(d < D - 1, HW)
#undef NEXTOU_TAP
    }
    critical[(size_t)b * V + v] = ((nc & a) | (na & c)) ? 1 : 0;
}

// Round 3: the same map as a SEPARABLE bitwise-OR dilation over LDS tiles.  With m = (lut_a[label], lut_c[label]) per voxel,
// the box neighbourhood's OR factorises into an OR along x, then y, then z; the 6 / 4 cross is (in-plane cross) | centre(z-1) |
// centre(z+1).  A workgroup owns a 64 x 8 column of voxels and walks the depth: per plane it looks every label of the halo tile
// up ONCE (one ds_read_b64 of the packed LUT), ORs along x into a second LDS plane and along y out of it, and keeps the last
// 2R+1 plane values of its voxels in registers — ~10 LDS instructions per voxel instead of 81 L1 / LDS instructions, label bytes
// read from HBM once per tile (+ halo), critical bytes written once, 64 consecutive bytes per wave.
constexpr int kCritTX = 64, kCritTY = 8, kCritTZ = 14, kCritG = 4;     // TZ + 2 planes of halo = 4 prefetch groups at R = 1

template <int R, bool BOX>
__global__ __launch_bounds__(256) void bti_critical_kernel(const uint8_t* __restrict__ labels, const uint32_t* __restrict__ lut_a,
                                                          const uint32_t* __restrict__ lut_c, int n_labels,
                                                          uint8_t* __restrict__ critical, int D, int H, int W, int z_chunks, int tz) {
    constexpr int HX = kCritTX + 2 * R, HY = kCritTY + 2 * R;
    __shared__ uint2 lut[256];
    __shared__ uint2 M[HY][HX];              // masks of the halo plane
    __shared__ uint2 P1[HY][kCritTX];        // BOX: OR along x;  cross: unused
    const int tid = threadIdx.x;
    for (int i = tid; i < 256; i += 256) lut[i] = i < n_labels ? make_uint2(lut_a[i], lut_c[i]) : make_uint2(0u, 0u);
    const int b = blockIdx.z / z_chunks, zc = blockIdx.z - b * z_chunks;
    const int x0 = blockIdx.x * kCritTX, y0 = blockIdx.y * kCritTY, z0 = zc * tz;
    const int z1 = min(D, z0 + tz);
    const long long HW = (long long)H * W;
    const uint8_t* lb = labels + (size_t)b * HW * D;
    uint8_t* cb = critical + (size_t)b * HW * D;
    const int tx = tid & 63, ty = tid >> 6;                 // the thread's voxels: rows ty and ty + 4 of the tile, column tx
    // rings over the last 2R+1 planes: BOX: the in-plane OR; cross: [0] / [1] / [2] = centre masks of planes z-2, z-1, z and the
    // in-plane cross of plane z-1
    uint2 ring[2][2 * R + 1], centre[2][R + 1], cross_q[2];
#pragma unroll
    for (int v = 0; v < 2; ++v) {
#pragma unroll
        for (int k = 0; k < 2 * R + 1; ++k) ring[v][k] = make_uint2(0u, 0u);
#pragma unroll
        for (int k = 0; k < R + 1; ++k) centre[v][k] = make_uint2(0u, 0u);
        cross_q[v] = make_uint2(0u, 0u);
    }
    // label bytes of the halo planes, prefetched a GROUP of kCritG planes ahead: a plane's LDS work is ~0.3 us, a global load ~1-2 us,
    // so un-prefetched the walk over TZ + 2R planes was a chain of exposed load latencies (30 us even for a 16 x 28 x 24 volume)
    constexpr int NL = (HY * HX + 255) / 256;
    uint8_t nxt[kCritG][NL], cur[kCritG][NL];
    auto fetch = [&](int zbase, uint8_t (&dst)[kCritG][NL]) __attribute__((always_inline)) {
#pragma unroll
        for (int g = 0; g < kCritG; ++g) {
            const int z = zbase + g;
            const bool plane_ok = z >= 0 && z < D && z < z1 + R;
#pragma unroll
            for (int k = 0; k < NL; ++k) {
                const int i = tid + k * 256;
                const int yy = i / HX, xx = i - yy * HX;
                const int y = y0 + yy - R, x = x0 + xx - R;
                const bool ok = plane_ok && i < HY * HX && y >= 0 && y < H && x >= 0 && x < W;
                dst[g][k] = ok ? lb[(size_t)z * HW + (size_t)y * W + x] : (uint8_t)0;
            }
        }
    };
    fetch(z0 - R, nxt);
    __syncthreads();
    for (int zg = z0 - R; zg < z1 + R; zg += kCritG) {
#pragma unroll
        for (int g = 0; g < kCritG; ++g)
#pragma unroll
            for (int k = 0; k < NL; ++k) cur[g][k] = nxt[g][k];
        if (zg + kCritG < z1 + R) fetch(zg + kCritG, nxt);
#pragma unroll
      for (int g = 0; g < kCritG; ++g) {
        const int z = zg + g;
        if (z >= z1 + R) break;
        const bool plane_ok = z >= 0 && z < D;
#pragma unroll
        for (int k = 0; k < NL; ++k) {
            const int i = tid + k * 256;
            if (i >= HY * HX) break;
            const int yy = i / HX, xx = i - yy * HX;
            const int y = y0 + yy - R, x = x0 + xx - R;
            uint2 m = make_uint2(0u, 0u);
            if (plane_ok && y >= 0 && y < H && x >= 0 && x < W) m = lut[cur[g][k]];
            M[yy][xx] = m;
        }
        __syncthreads();
        if constexpr (BOX) {
            for (int i = tid; i < HY * kCritTX; i += 256) {
                const int yy = i >> 6, xx = i & 63;
                uint2 o = M[yy][xx];
#pragma unroll
                for (int dx = 1; dx <= 2 * R; ++dx) { const uint2 t = M[yy][xx + dx]; o.x |= t.x; o.y |= t.y; }
                P1[yy][xx] = o;
            }
            __syncthreads();
        }
#pragma unroll
        for (int v = 0; v < 2; ++v) {
            const int row = ty + 4 * v;
            uint2 o;
            if constexpr (BOX) {
                o = P1[row][tx];
#pragma unroll
                for (int dy = 1; dy <= 2 * R; ++dy) { const uint2 t = P1[row + dy][tx]; o.x |= t.x; o.y |= t.y; }
            } else {
                const uint2 c0 = M[row + 1][tx + 1], l = M[row + 1][tx], r = M[row + 1][tx + 2], u = M[row][tx + 1], d = M[row + 2][tx + 1];
                o = make_uint2(c0.x | l.x | r.x | u.x | d.x, c0.y | l.y | r.y | u.y | d.y);
            }
            // slide the window: after this step ring[.][2R] belongs to plane z, centre[.][R] to plane z
#pragma unroll
            for (int k = 0; k < 2 * R; ++k) ring[v][k] = ring[v][k + 1];
            ring[v][2 * R] = o;
#pragma unroll
            for (int k = 0; k < R; ++k) centre[v][k] = centre[v][k + 1];
            centre[v][R] = M[row + R][tx + R];
            const int zo = z - R;                           // the plane whose neighbourhood is now complete
            const int y = y0 + row, x = x0 + tx;
            if (zo >= z0 && zo < z1 && y < H && x < W) {
                uint2 n;
                uint2 self;
                if constexpr (BOX) {
                    n = ring[v][0];
#pragma unroll
                    for (int k = 1; k < 2 * R + 1; ++k) { n.x |= ring[v][k].x; n.y |= ring[v][k].y; }
                    self = centre[v][0];
                } else {
                    // R == 1: ring[.][1] = in-plane cross of plane zo; centre[.][0] = centre of plane zo; the z taps are the centres
                    // of planes zo - 1 (kept in cross_q) and zo + 1 (= this plane's centre)
                    n = make_uint2(ring[v][1].x | cross_q[v].x | centre[v][1].x, ring[v][1].y | cross_q[v].y | centre[v][1].y);
                    self = centre[v][0];
                }
                cb[(size_t)zo * HW + (size_t)y * W + x] = ((n.y & self.x) | (n.x & self.y)) ? 1 : 0;
            }
            if constexpr (!BOX) cross_q[v] = centre[v][0];  // becomes "centre of plane zo - 1" for the next output plane
        }
        __syncthreads();        // M / P1 are rewritten by the next plane
      }
    }
}

// ---------------------------------------------------------------------------------------------
// critical-voxel cross-entropy in float64 (reference loss/bti_loss.py:141-143):
//   ce[b,v] = logsumexp_l(double(x[b,l,v])) - double(x[b,target,v]);  loss[b] = sum_v critical * ce
// Only critical voxels are touched: a wave whose 64 voxels are all non-critical skips its L loads
// (critical voxels hug the organ interfaces, so most waves do).  Partial sums are written per block
// and added up by the caller in a fixed order (no float64 atomics: bit-reproducible).
// ---------------------------------------------------------------------------------------------
constexpr int kCeBlocks = 1024;  // partial sums per batch element

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        const long long bits = __double_as_longlong(v);
        const int lo = __shfl_xor((int)(bits & 0xffffffffll), off);
        const int hi = __shfl_xor((int)(bits >> 32), off);
        v += __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
    }
    return v;
}

// LMAX > 0: L <= LMAX class planes, held in registers — every logit is read once and every exp evaluated once (the generic
// LMAX = 0 form re-reads the planes for the max, the sum and, backward, the probabilities, and evaluates exp twice per class
// in the backward).  Same float64 arithmetic in the same order either way.
template <int LMAX>
__global__ __launch_bounds__(256) void bti_ce_fwd_kernel(const float* __restrict__ logits,
                                                         const uint8_t* __restrict__ target,
                                                         const uint8_t* __restrict__ critical,
                                                         double* __restrict__ partial, int L, long long V, long long sl,
                                                         long long sv) {
    // element (b, l, v) at b * L * V + l * sl + v * sv: (sl, sv) = (V, 1) planes, (1, L) channels-last rows
    __shared__ double wsum[4];
    const int b = blockIdx.y;
    const float* lb = logits + (size_t)b * L * V;
    const uint8_t* tb = target + (size_t)b * V;
    const uint8_t* cb = critical + (size_t)b * V;
    double acc = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long v_end = (V + blockDim.x - 1) / blockDim.x * blockDim.x;  // keep waves converged for __any
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < v_end; v += stride) {
        const bool crit = v < V && cb[v] != 0;
        if (!__any(crit)) continue;
        if (crit) {
            const int y = tb[v];
            if constexpr (LMAX > 0) {
                float x[LMAX];
#pragma unroll
                for (int l = 0; l < LMAX; ++l) x[l] = l < L ? lb[(size_t)l * sl + (size_t)v * sv] : -INFINITY;
                float mf = x[0];
#pragma unroll
                for (int l = 1; l < LMAX; ++l) mf = fmaxf(mf, x[l]);          // max of floats == max of their doubles
                const double m = (double)mf;
                double s = 0.0, xy = 0.0;
#pragma unroll
                for (int l = 0; l < LMAX; ++l) {
                    if (l < L) s += exp((double)x[l] - m);
                    if (l == y) xy = (double)x[l];
                }
                if (y < L) acc += (m + log(s)) - xy;
            } else {
                double m = (double)lb[(size_t)v * sv], xy = (y == 0) ? m : 0.0;
                for (int l = 1; l < L; ++l) {
                    const double x = (double)lb[(size_t)l * sl + (size_t)v * sv];
                    if (l == y) xy = x;
                    m = fmax(m, x);
                }
                double s = 0.0;
                for (int l = 0; l < L; ++l) s += exp((double)lb[(size_t)l * sl + (size_t)v * sv] - m);
                if (y < L) acc += (m + log(s)) - xy;
            }
        }
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partial[(size_t)b * gridDim.x + blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// grad[b,l,v] = scale * critical[b,v] * (softmax_l(double x) - [l == target]), written for every voxel.
template <int LMAX>
__global__ __launch_bounds__(256) void bti_ce_bwd_kernel(const float* __restrict__ logits,
                                                         const uint8_t* __restrict__ target,
                                                         const uint8_t* __restrict__ critical,
                                                         const double* __restrict__ scale_dev,
                                                         float* __restrict__ grad, int L, long long V, long long sl, long long sv) {
    const int b = blockIdx.y;
    const float* lb = logits + (size_t)b * L * V;
    float* gb = grad + (size_t)b * L * V;
    const uint8_t* tb = target + (size_t)b * V;
    const uint8_t* cb = critical + (size_t)b * V;
    const double scale = scale_dev[b];  // upstream gradient of this sample's sum
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
        if (cb[v] == 0) {
            for (int l = 0; l < L; ++l) gb[(size_t)l * sl + (size_t)v * sv] = 0.f;
            continue;
        }
        const int y = tb[v];
        if constexpr (LMAX > 0) {
            float x[LMAX];
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = l < L ? lb[(size_t)l * sl + (size_t)v * sv] : -INFINITY;
            float mf = x[0];
#pragma unroll
            for (int l = 1; l < LMAX; ++l) mf = fmaxf(mf, x[l]);
            const double m = (double)mf;
            double e[LMAX], s = 0.0;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                e[l] = l < L ? exp((double)x[l] - m) : 0.0;
                if (l < L) s += e[l];
            }
            const double inv = (y < L) ? scale / s : 0.0;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) {
                if (l >= L) break;
                const double p = e[l] * inv;
                gb[(size_t)l * sl + (size_t)v * sv] = (float)(l == y ? p - ((y < L) ? scale : 0.0) : p);
            }
        } else {
            double m = (double)lb[(size_t)v * sv];
            for (int l = 1; l < L; ++l) m = fmax(m, (double)lb[(size_t)l * sl + (size_t)v * sv]);
            double s = 0.0;
            for (int l = 0; l < L; ++l) s += exp((double)lb[(size_t)l * sl + (size_t)v * sv] - m);
            const double inv = (y < L) ? scale / s : 0.0;
            for (int l = 0; l < L; ++l) {
                const double p = exp((double)lb[(size_t)l * sl + (size_t)v * sv] - m) * inv;
                gb[(size_t)l * sl + (size_t)v * sv] = (float)(l == y ? p - ((y < L) ? scale : 0.0) : p);
            }
        }
    }
}

}  // namespace nextou

using namespace nextou;

namespace nextou {

// --------------------------------------------------------------------------------------------
// Mean cross-entropy of the segmentation logits (the deep-supervision CE every NexToU trainer's loss contains:
// nnUNetTrainer_NexToU*.py -> nnU-Net's RobustCrossEntropyLoss = torch.nn.CrossEntropyLoss, mean over the voxels whose target
// is not `ignore_index`), fp32 arithmetic as ATen's (max-subtracted log-sum-exp), float64 partial sums in a fixed order.
// ATen runs it as log_softmax -> nll_loss on NCDHW tensors: with the network's logits channels-last that is a layout copy in, two
// passes forward, two backward and a copy of the gradient back (~3 ms of the cfg-2 step for 99 M logits).  Here: one pass each way
// over the logits where they lie — element (b, l, v) at b * L * V + l * sl + v * sv, i.e. (sl, sv) = (1, L) for channels-last rows,
// (V, 1) for NCDHW planes — and the gradient is written in the same layout.
//   fwd: partial[block] = (sum of -log p[target], number of counted voxels)
//   bwd: grad[b, l, v] = scale * (p[l] - [l == target]) for counted voxels, 0 otherwise; scale = upstream gradient / count (device)
// --------------------------------------------------------------------------------------------
constexpr int kCeMeanBlocks = 2048;

template <int LMAX, bool ROWS>
__global__ __launch_bounds__(256) void ce_mean_fwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          double2* __restrict__ partial, int L, long long total, long long V,
                                                          long long sl, long long sv, long long ignore_index) {
    // (forward, ROWS: every lane reads its own row as 8-byte pairs — the rows of a wave are one contiguous 64 * L * 4-byte run that the
    // L / 2 loads cover between them out of L1: 117 us for cfg 2's four heads against 153 us with the LDS staging the backward uses)
    __shared__ double2 wsum[4];
    double acc = 0.0, cnt = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long p = (long long)blockIdx.x * blockDim.x + threadIdx.x; p < total; p += stride) {
        const long long y = target[p];
        if (y == ignore_index) continue;
        if (y < 0 || y >= L) {         // a label outside [0, L) that is not the ignore index: torch.nn.CrossEntropyLoss raises a device
            acc += (double)NAN;        // assert; here the LOSS turns NaN (loud, capturable, no host sync) instead of the voxel silently
            continue;                  // dropping out of the mean
        }
        float x[LMAX];
        if constexpr (ROWS) {
            const float* base = logits + (size_t)p * L;
#pragma unroll
            for (int l = 0; l < LMAX; l += 2) {
                float2 t = make_float2(-INFINITY, -INFINITY);
                if (l + 1 < L) t = *reinterpret_cast<const float2*>(base + l);
                x[l] = t.x;
                if (l + 1 < LMAX) x[l + 1] = t.y;
            }
        } else {
            const long long b = p / V, v = p - b * V;
            const float* base = logits + (size_t)b * L * V + (size_t)v * sv;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = l < L ? base[(size_t)l * sl] : -INFINITY;
        }
        float m = x[0];
#pragma unroll
        for (int l = 1; l < LMAX; ++l) m = fmaxf(m, x[l]);
        float ssum = 0.f, xy = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            if (l < L) ssum += expf(x[l] - m);
            if (l == (int)y) xy = x[l];
        }
        acc += (double)((m + logf(ssum)) - xy);
        cnt += 1.0;
    }
    acc = wave_sum(acc);
    cnt = wave_sum(cnt);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = make_double2(acc, cnt);
    __syncthreads();
    if (threadIdx.x == 0)
        partial[blockIdx.x] = make_double2((wsum[0].x + wsum[1].x) + (wsum[2].x + wsum[3].x), (wsum[0].y + wsum[1].y) + (wsum[2].y + wsum[3].y));
}

// backward, ROWS: a workgroup's 256 voxels are one contiguous run of 256 * L floats that moves through LDS with 16-byte accesses both
// ways (every lane writing its own 56-byte row: 410 us for cfg 2's four heads = 2 TB/s; staged: 189 us).
template <int LMAX, bool ROWS>
__global__ __launch_bounds__(256) void ce_mean_bwd_kernel(const float* __restrict__ logits, const long long* __restrict__ target,
                                                          const float* __restrict__ scale_dev, float* __restrict__ grad, int L,
                                                          long long total, long long V, long long sl, long long sv,
                                                          long long ignore_index) {
    __shared__ __attribute__((aligned(16))) float tile[ROWS ? 256 * LMAX : 4];
    const float scale = scale_dev[0];
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long p_end = (total + 255) / 256 * 256;
    for (long long p0 = (long long)blockIdx.x * 256; p0 < p_end; p0 += stride) {
        const long long p = p0 + threadIdx.x;
        const bool live = p < total;
        const long long y = live ? target[p] : ignore_index;
        const bool counted = live && !(y == ignore_index || y < 0 || y >= L);
        float x[LMAX];
        int floats = 0;
        size_t off = 0;
        if constexpr (ROWS) {
            const long long n = total - p0 < 256 ? total - p0 : 256;
            floats = (int)n * L;
            const float* src = logits + (size_t)p0 * L;
            __syncthreads();
            for (int i = threadIdx.x * 4; i < floats; i += 1024) {
                if (i + 3 < floats) *reinterpret_cast<float4*>(tile + i) = *reinterpret_cast<const float4*>(src + i);
                else for (int k = i; k < floats; ++k) tile[k] = src[k];
            }
            __syncthreads();
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = (l < L && live) ? tile[threadIdx.x * L + l] : -INFINITY;
        } else {
            const long long b = live ? p / V : 0, v = live ? p - b * V : 0;
            off = (size_t)b * L * V + (size_t)v * sv;
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = (l < L && live) ? logits[off + (size_t)l * sl] : -INFINITY;
        }
        float m = x[0];
#pragma unroll
        for (int l = 1; l < LMAX; ++l) m = fmaxf(m, x[l]);
        float e[LMAX], ssum = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] = (l < L && live) ? expf(x[l] - m) : 0.f;
            ssum += e[l];
        }
        const float inv = counted ? scale / ssum : 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) e[l] = e[l] * inv - ((counted && l == (int)y) ? scale : 0.f);
        if constexpr (ROWS) {
            __syncthreads();                                             // everyone has read its row
            if (live) {
#pragma unroll
                for (int l = 0; l < LMAX; ++l)
                    if (l < L) tile[threadIdx.x * L + l] = e[l];
            }
            __syncthreads();
            float* dst = grad + (size_t)p0 * L;
            for (int i = threadIdx.x * 4; i < floats; i += 1024) {
                if (i + 3 < floats) *reinterpret_cast<float4*>(dst + i) = *reinterpret_cast<const float4*>(tile + i);
                else for (int k = i; k < floats; ++k) dst[k] = tile[k];
            }
        } else if (live) {
#pragma unroll
            for (int l = 0; l < LMAX; ++l)
                if (l < L) grad[off + (size_t)l * sl] = e[l];
        }
    }
}


// --------------------------------------------------------------------------------------------
// K5d — soft-Dice statistics of the segmentation logits (round 4).  Every NexToU trainer's loss holds nnU-Net's (Memory-
// Efficient)SoftDiceLoss next to the cross-entropy (reference loss/compound_bti_loss.py:29-30, :53-55 -> nnunetv2 dice.py): softmax over
// the classes, a one-hot scatter of the target, three products and three reductions over the volume — ~8 ms of the cfg-4 step as
// generic ATen passes over the 99 M logits of the four weighted heads (profiles/r03_cfg4_step_kernel_trace_start.md).  What the loss
// needs from the volume are three sums per (sample, class):
//   intersect[b,l] = sum_v w p[b,l,v] [y = l],   sum_pred[b,l] = sum_v w p[b,l,v],   sum_gt[b,l] = sum_v w [y = l]
// (p = softmax over l in fp32 as ATen computes it, w = loss mask or 1), after which Dice is arithmetic on (B, L) numbers that stays
// in PyTorch (do_bg, batch_dice, smooth, the DDP all-gather: nnU-Net's own formula on the sums).
//   fwd: one pass over the logits where they lie -> partial[b][block][l][3] doubles (fixed order; the caller adds the blocks);
//   bwd: with gi = dLoss/d intersect, gp = dLoss/d sum_pred (device (B, L) doubles): G_l = w (gp[l] + gi[l][y = l]),
//        grad[k] = p_k (G_k - sum_l G_l p_l) — softmax's Jacobian applied once, written in the logits' layout.
// Logits element (b, l, v) at b L V + l sl + v sv: (V, 1) planes or (1, L) channels-last rows.  target: uint8 labels; mask: uint8 or NULL.
// --------------------------------------------------------------------------------------------
constexpr int kDiceBlocks = 512;

template <int LMAX>
__global__ __launch_bounds__(256) void dice_stats_fwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ target,
                                                             const uint8_t* __restrict__ mask, double* __restrict__ partial, int L,
                                                             long long V, long long sl, long long sv) {
    const bool rows = sl == 1 && sv == L && V > 1;          // (V == 1: both layouts are the same bytes -> the strided form)
    __shared__ float red[4][3 * LMAX];
    const int b = blockIdx.y;
    const float* lb = logits + (size_t)b * L * V;
    const uint8_t* tb = target + (size_t)b * V;
    const uint8_t* mb = mask ? mask + (size_t)b * V : nullptr;
    float si[LMAX], sp[LMAX], sg[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) si[l] = sp[l] = sg[l] = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
        float x[LMAX];
        if (rows) {
            const float* base = lb + (size_t)v * sv;
#pragma unroll
            for (int l = 0; l < LMAX; l += 2) {
                float2 t = make_float2(-INFINITY, -INFINITY);
                if (l + 1 < L) t = *reinterpret_cast<const float2*>(base + l);
                x[l] = t.x;
                if (l + 1 < LMAX) x[l + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = l < L ? lb[(size_t)l * sl + (size_t)v * sv] : -INFINITY;
        }
        const int y = tb[v];
        const float w = mb ? (mb[v] ? 1.f : 0.f) : 1.f;
        float m = x[0];
#pragma unroll
        for (int l = 1; l < LMAX; ++l) m = fmaxf(m, x[l]);
        float e[LMAX], ssum = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { e[l] = l < L ? expf(x[l] - m) : 0.f; ssum += e[l]; }
        const float inv = w / ssum;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            const float p = e[l] * inv;            // w * softmax
            sp[l] += p;
            if (l == y) { si[l] += p; sg[l] += w; }
        }
    }
    // lanes -> wave (fp32 tree over <= a few dozen addends each), waves -> block in double, fixed order
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int l = 0; l < LMAX; ++l) {
        float a = si[l], c = sp[l], d = sg[l];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) { a += __shfl_xor(a, off); c += __shfl_xor(c, off); d += __shfl_xor(d, off); }
        if (lane == 0) { red[wave][3 * l] = a; red[wave][3 * l + 1] = c; red[wave][3 * l + 2] = d; }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * L; i += blockDim.x)
        partial[((size_t)b * gridDim.x + blockIdx.x) * 3 * L + i] =
            (((double)red[0][i] + (double)red[1][i]) + (double)red[2][i]) + (double)red[3][i];
}

template <int LMAX>
__global__ __launch_bounds__(256) void dice_stats_bwd_kernel(const float* __restrict__ logits, const uint8_t* __restrict__ target,
                                                             const uint8_t* __restrict__ mask, const double* __restrict__ g_inter,
                                                             const double* __restrict__ g_pred, float* __restrict__ grad, int L,
                                                             long long V, long long sl, long long sv) {
    const bool rows = sl == 1 && sv == L && V > 1;
    __shared__ float gi_s[LMAX], gp_s[LMAX];
    const int b = blockIdx.y;
    if (threadIdx.x < LMAX) {
        gi_s[threadIdx.x] = threadIdx.x < L ? (float)g_inter[(size_t)b * L + threadIdx.x] : 0.f;
        gp_s[threadIdx.x] = threadIdx.x < L ? (float)g_pred[(size_t)b * L + threadIdx.x] : 0.f;
    }
    __syncthreads();
    float gi[LMAX], gp[LMAX];
#pragma unroll
    for (int l = 0; l < LMAX; ++l) { gi[l] = gi_s[l]; gp[l] = gp_s[l]; }
    const float* lb = logits + (size_t)b * L * V;
    float* gb = grad + (size_t)b * L * V;
    const uint8_t* tb = target + (size_t)b * V;
    const uint8_t* mb = mask ? mask + (size_t)b * V : nullptr;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < V; v += stride) {
        float x[LMAX];
        if (rows) {
            const float* base = lb + (size_t)v * sv;
#pragma unroll
            for (int l = 0; l < LMAX; l += 2) {
                float2 t = make_float2(-INFINITY, -INFINITY);
                if (l + 1 < L) t = *reinterpret_cast<const float2*>(base + l);
                x[l] = t.x;
                if (l + 1 < LMAX) x[l + 1] = t.y;
            }
        } else {
#pragma unroll
            for (int l = 0; l < LMAX; ++l) x[l] = l < L ? lb[(size_t)l * sl + (size_t)v * sv] : -INFINITY;
        }
        const int y = tb[v];
        const float w = mb ? (mb[v] ? 1.f : 0.f) : 1.f;
        float m = x[0];
#pragma unroll
        for (int l = 1; l < LMAX; ++l) m = fmaxf(m, x[l]);
        float e[LMAX], ssum = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) { e[l] = l < L ? expf(x[l] - m) : 0.f; ssum += e[l]; }
        const float inv = 1.f / ssum;
        float dot = 0.f;
#pragma unroll
        for (int l = 0; l < LMAX; ++l) {
            e[l] *= inv;                                            // p_l
            const float G = w * (gp[l] + (l == y ? gi[l] : 0.f));
            x[l] = G;
            dot = fmaf(G, e[l], dot);
        }
        if (rows) {
            float* base = gb + (size_t)v * sv;
#pragma unroll
            for (int l = 0; l < LMAX; l += 2)
                if (l + 1 < L) *reinterpret_cast<float2*>(base + l) = make_float2(e[l] * (x[l] - dot), e[l + 1 < LMAX ? l + 1 : l] * (x[l + 1 < LMAX ? l + 1 : l] - dot));
        } else {
#pragma unroll
            for (int l = 0; l < LMAX; ++l)
                if (l < L) gb[(size_t)l * sl + (size_t)v * sv] = e[l] * (x[l] - dot);
        }
    }
}

}  // namespace nextou

extern "C" int nextou_bti_ce_partials(void) { return kCeBlocks; }

static int check_strides(const char* who, int L, long long V, long long sl, long long sv) {
    NEXTOU_REQUIRE((sl == 1 && sv == L) || (sl == V && sv == 1), "%s: strides (%lld, %lld) are neither channels-last rows (1, L) nor planes (V, 1)",
                   who, sl, sv);
    return 0;
}

extern "C" int nextou_bti_ce_fwd(const float* logits, const uint8_t* target, const uint8_t* critical,
                                 double* partial, int B, int L, int64_t V, int64_t stride_l, int64_t stride_v, nextou_stream_t stream) {
    NEXTOU_REQUIRE(logits && target && critical && partial, "bti_ce_fwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && L > 0 && L <= 256 && V > 0 && B <= 65535, "bti_ce_fwd: bad size B=%d L=%d V=%lld", B, L, (long long)V);
    if (int e = check_strides("bti_ce_fwd", L, V, stride_l, stride_v)) return e;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, (4.0 * L + 2.0) * B * (double)V, "bti_ce_fwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    if (L <= 16)
        hipLaunchKernelGGL(bti_ce_fwd_kernel<16>, dim3(kCeBlocks, B), dim3(256), 0, s, logits, target, critical, partial, L, (long long)V,
                           (long long)stride_l, (long long)stride_v);
    else
        hipLaunchKernelGGL(bti_ce_fwd_kernel<0>, dim3(kCeBlocks, B), dim3(256), 0, s, logits, target, critical, partial, L, (long long)V,
                           (long long)stride_l, (long long)stride_v);
    return check_launch("bti_ce_fwd_kernel");
}

extern "C" int nextou_bti_ce_bwd(const float* logits, const uint8_t* target, const uint8_t* critical,
                                 const double* scale_dev, float* grad_logits, int B, int L, int64_t V, int64_t stride_l, int64_t stride_v,
                                 nextou_stream_t stream) {
    NEXTOU_REQUIRE(logits && target && critical && scale_dev && grad_logits, "bti_ce_bwd: null pointer");
    NEXTOU_REQUIRE(B > 0 && L > 0 && L <= 256 && V > 0 && B <= 65535, "bti_ce_bwd: bad size B=%d L=%d V=%lld", B, L, (long long)V);
    if (int e = check_strides("bti_ce_bwd", L, V, stride_l, stride_v)) return e;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, (8.0 * L + 2.0) * B * (double)V, "bti_ce_bwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    long long blocks = cdiv64(V, 256);
    if (blocks > 8192) blocks = 8192;
    if (L <= 16)
        hipLaunchKernelGGL(bti_ce_bwd_kernel<16>, dim3((unsigned)blocks, B), dim3(256), 0, s, logits, target, critical, scale_dev, grad_logits, L,
                           (long long)V, (long long)stride_l, (long long)stride_v);
    else
        hipLaunchKernelGGL(bti_ce_bwd_kernel<0>, dim3((unsigned)blocks, B), dim3(256), 0, s, logits, target, critical, scale_dev, grad_logits, L,
                           (long long)V, (long long)stride_l, (long long)stride_v);
    return check_launch("bti_ce_bwd_kernel");
}

extern "C" int nextou_argmax_labels(const float* logits, uint8_t* labels, int B, int L, int64_t V, int64_t stride_l, int64_t stride_v,
                                    nextou_stream_t stream) {
    NEXTOU_REQUIRE(logits && labels, "argmax_labels: null pointer");
    NEXTOU_REQUIRE(B > 0 && L > 0 && L <= 256 && V > 0 && B <= 65535,
                   "argmax_labels: bad size B=%d L=%d V=%lld (L <= 256)", B, L, (long long)V);
    if (int e = check_strides("argmax_labels", L, V, stride_l, stride_v)) return e;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, (4.0 * L + 1.0) * B * (double)V, "argmax_labels_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 && stride_v == L && L > 1 && V > 1 ? "rows" : "planes");
    if (stride_l == 1 && stride_v == L && L > 1 && V > 1) {     // channels-last rows (L = 1 or V = 1: both layouts are the same bytes)
        NEXTOU_REQUIRE(L <= 32 && L % 2 == 0 && (reinterpret_cast<uintptr_t>(logits) & 7u) == 0,
                       "argmax_labels: channels-last rows need an even class count <= 32 and 8-byte aligned logits (L=%d)", L);
        const long long rows = (long long)B * V;
        long long blocks = cdiv64(rows, 256);
        if (blocks > 16384) blocks = 16384;
        if (L <= 16) hipLaunchKernelGGL(argmax_labels_rows_kernel<16>, dim3((unsigned)blocks), dim3(256), 0, s, logits, labels, L, rows);
        else hipLaunchKernelGGL(argmax_labels_rows_kernel<32>, dim3((unsigned)blocks), dim3(256), 0, s, logits, labels, L, rows);
        return check_launch("argmax_labels_rows_kernel");
    }
    const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(labels) & 3u) == 0);
    const long long items = vec ? V / 4 : V;
    long long blocks = cdiv64(items, 256);
    if (blocks > 8192) blocks = 8192;  // grid-stride the rest
    if (vec)
        hipLaunchKernelGGL(argmax_labels_kernel<true>, dim3((unsigned)blocks, B), dim3(256), 0, s, logits,
                           labels, L, (long long)V);
    else
        hipLaunchKernelGGL(argmax_labels_kernel<false>, dim3((unsigned)blocks, B), dim3(256), 0, s,
                           logits, labels, L, (long long)V);
    return check_launch("argmax_labels_kernel");
}

extern "C" int nextou_labels_u8(const void* src, int src_dtype, uint8_t* out, int64_t n, int n_classes, unsigned* flag,
                                nextou_stream_t stream) {
    NEXTOU_REQUIRE(src && out && flag, "labels_u8: null pointer");
    NEXTOU_REQUIRE(n > 0 && n_classes > 0 && n_classes <= 256 && src_dtype >= 0 && src_dtype <= 2,
                   "labels_u8: bad arguments n=%lld classes=%d dtype=%d", (long long)n, n_classes, src_dtype);
    hipStream_t s = (hipStream_t)stream;
    const double in_bytes = src_dtype == 0 ? 4.0 : (src_dtype == 1 ? 8.0 : 1.0);
    ProfScope prof(s, kBoundHbm, (in_bytes + 1.0) * (double)n, "labels_u8_kernel[n%lld]", (long long)n);
    long long blocks = cdiv64(n, 256 * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks), block(256);
    if (src_dtype == 0) hipLaunchKernelGGL(labels_u8_kernel<0>, grid, block, 0, s, src, out, (long long)n, n_classes, flag);
    else if (src_dtype == 1) hipLaunchKernelGGL(labels_u8_kernel<1>, grid, block, 0, s, src, out, (long long)n, n_classes, flag);
    else hipLaunchKernelGGL(labels_u8_kernel<2>, grid, block, 0, s, src, out, (long long)n, n_classes, flag);
    return check_launch("labels_u8_kernel");
}

extern "C" int nextou_bti_critical_map(const uint8_t* labels, const uint32_t* lut_a,
                                       const uint32_t* lut_c, int n_labels, uint8_t* critical, int B,
                                       int D, int H, int W, int connectivity, int min_thick,
                                       nextou_stream_t stream) {
    NEXTOU_REQUIRE(labels && lut_a && lut_c && critical, "bti_critical_map: null pointer");
    NEXTOU_REQUIRE(B > 0 && D > 0 && H > 0 && W > 0 && B <= 65535, "bti_critical_map: bad size B=%d D=%d H=%d W=%d",
                   B, D, H, W);
    NEXTOU_REQUIRE(n_labels > 0 && n_labels <= 256, "bti_critical_map: n_labels=%d not in [1,256]", n_labels);
    int full_box;
    if (connectivity == 26 || connectivity == 8) full_box = 1;
    else if (connectivity == 6 || connectivity == 4) full_box = 0;
    else return fail(NEXTOU_EINVAL, "bti_critical_map: connectivity %d not in {4,8,6,26}", connectivity);
    NEXTOU_REQUIRE(!full_box || min_thick >= 1, "bti_critical_map: min_thick=%d must be >= 1", min_thick);
    NEXTOU_REQUIRE(!((connectivity == 8 || connectivity == 4) && D != 1),
                   "bti_critical_map: 2-D connectivity %d needs D == 1 (got %d)", connectivity, D);
    const long long V = (long long)D * H * W;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, 2.0 * B * (double)V, "bti_critical_kernel[B%d %dx%dx%d c%d]", B, D, H, W, connectivity);
    const int rad = full_box ? min_thick : 1;
    // depth per workgroup: kCritTZ planes for volumes that fill the chip; a low-resolution deep-supervision scale (cfg 4: 32x56x48 and
    // smaller, a few dozen workgroups) is a chain of tz + 2R dependent plane steps (~1 us each: 17-19 us whatever the volume, round 4's
    // floor) — shorter columns (6, then 2 planes) trade a little halo work for a chain of 8 / 4 steps on 3-7x the workgroups
    int tz = kCritTZ;
    {
        const long long per_z = (long long)cdiv(W, kCritTX) * cdiv(H, kCritTY) * B;
        if (per_z * cdiv(D, tz) < 512) tz = 6;
        if (per_z * cdiv(D, tz) < 512) tz = 2;
        if (const char* e = getenv("NEXTOU_BTI_TZ")) { const int v = atoi(e); if (v >= 1 && v <= kCritTZ) tz = v; }      // experiments
    }
    const int z_chunks = cdiv(D, tz);
    const dim3 grid(cdiv(W, kCritTX), cdiv(H, kCritTY), (unsigned)(B * z_chunks));
    if (rad > 3 || (long long)B * z_chunks > 65535 || cdiv(H, kCritTY) > 65535) {
        hipLaunchKernelGGL(bti_critical_naive_kernel, dim3((unsigned)cdiv64(V, 256), 1, B), dim3(256), 0, s, labels, lut_a, lut_c, n_labels,
                           critical, D, H, W, full_box, rad);
        return check_launch("bti_critical_naive_kernel");
    }
#define NEXTOU_CRIT(R_, BOX_)                                                                                                      \
    hipLaunchKernelGGL((bti_critical_kernel<R_, BOX_>), grid, dim3(256), 0, s, labels, lut_a, lut_c, n_labels, critical, D, H, W, z_chunks, tz)
    if (!full_box) NEXTOU_CRIT(1, false);
    else if (rad == 1) NEXTOU_CRIT(1, true);
    else if (rad == 2) NEXTOU_CRIT(2, true);
    else NEXTOU_CRIT(3, true);
#undef NEXTOU_CRIT
    return check_launch("bti_critical_kernel");
}

extern "C" int nextou_dice_stats_partials(void) { return kDiceBlocks; }

static int check_dice(const char* who, const void* a, const void* b, const void* c, int B, int L, long long V, long long sl, long long sv) {
    NEXTOU_REQUIRE(a && b && c, "%s: null pointer", who);
    NEXTOU_REQUIRE(B > 0 && B <= 65535 && L > 1 && L <= 32 && V > 0, "%s: bad size B=%d L=%d V=%lld (2 <= L <= 32)", who, B, L, V);
    NEXTOU_REQUIRE((sl == 1 && sv == L && L % 2 == 0 && (reinterpret_cast<uintptr_t>(a) & 7u) == 0) || (sl == V && sv == 1),
                   "%s: strides (%lld, %lld) are neither channels-last rows (1, L; L even, 8-byte aligned) nor planes (V, 1)", who, sl, sv);
    return 0;
}

extern "C" int nextou_dice_stats_fwd(const float* logits, const uint8_t* target, const uint8_t* mask, double* partial, int B, int L,
                                     int64_t V, int64_t stride_l, int64_t stride_v, nextou_stream_t stream) {
    if (int e = check_dice("dice_stats_fwd", logits, target, partial, B, L, V, stride_l, stride_v)) return e;
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, (4.0 * L + (mask ? 2.0 : 1.0)) * B * (double)V, "dice_stats_fwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    const dim3 grid(kDiceBlocks, B), block(256);
    if (L <= 16)
        hipLaunchKernelGGL(dice_stats_fwd_kernel<16>, grid, block, 0, s, logits, target, mask, partial, L, (long long)V, (long long)stride_l,
                           (long long)stride_v);
    else
        hipLaunchKernelGGL(dice_stats_fwd_kernel<32>, grid, block, 0, s, logits, target, mask, partial, L, (long long)V, (long long)stride_l,
                           (long long)stride_v);
    return check_launch("dice_stats_fwd_kernel");
}

extern "C" int nextou_dice_stats_bwd(const float* logits, const uint8_t* target, const uint8_t* mask, const double* g_intersect,
                                     const double* g_sum_pred, float* grad_logits, int B, int L, int64_t V, int64_t stride_l,
                                     int64_t stride_v, nextou_stream_t stream) {
    if (int e = check_dice("dice_stats_bwd", logits, target, grad_logits, B, L, V, stride_l, stride_v)) return e;
    NEXTOU_REQUIRE(g_intersect && g_sum_pred, "dice_stats_bwd: null gradient");
    hipStream_t s = (hipStream_t)stream;
    ProfScope prof(s, kBoundHbm, (8.0 * L + (mask ? 2.0 : 1.0)) * B * (double)V, "dice_stats_bwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    long long blocks = cdiv64(V, 256 * 4);
    if (blocks > 4096) blocks = 4096;
    if (blocks < 1) blocks = 1;
    const dim3 grid((unsigned)blocks, B), block(256);
    if (L <= 16)
        hipLaunchKernelGGL(dice_stats_bwd_kernel<16>, grid, block, 0, s, logits, target, mask, g_intersect, g_sum_pred, grad_logits, L,
                           (long long)V, (long long)stride_l, (long long)stride_v);
    else
        hipLaunchKernelGGL(dice_stats_bwd_kernel<32>, grid, block, 0, s, logits, target, mask, g_intersect, g_sum_pred, grad_logits, L,
                           (long long)V, (long long)stride_l, (long long)stride_v);
    return check_launch("dice_stats_bwd_kernel");
}

extern "C" int nextou_ce_mean_partials(void) { return kCeMeanBlocks; }

static int check_ce_mean(const char* who, const void* a, const void* b, const void* c, int B, int L, long long V, long long sl, long long sv) {
    NEXTOU_REQUIRE(a && b && c, "%s: null pointer", who);
    NEXTOU_REQUIRE(B > 0 && L > 0 && L <= 32 && V > 0, "%s: bad size B=%d L=%d V=%lld (L <= 32)", who, B, L, V);
    NEXTOU_REQUIRE((sl == 1 && sv == L) || (sl == V && sv == 1), "%s: strides (%lld, %lld) are neither channels-last rows (1, L) nor planes (V, 1)",
                   who, sl, sv);
    return 0;
}

extern "C" int nextou_ce_mean_fwd(const float* logits, const int64_t* target, double* partial, int B, int L, int64_t V,
                                  int64_t stride_l, int64_t stride_v, int64_t ignore_index, nextou_stream_t stream) {
    if (int e = check_ce_mean("ce_mean_fwd", logits, target, partial, B, L, V, stride_l, stride_v)) return e;
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)B * V;
    const bool rows = stride_l == 1 && L % 2 == 0 && (reinterpret_cast<uintptr_t>(logits) & 15u) == 0;
    ProfScope prof(s, kBoundHbm, (4.0 * L + 8.0) * (double)total, "ce_mean_fwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    const dim3 grid(kCeMeanBlocks), block(256);
#define NEXTOU_CE_FWD(LM, R)                                                                                                       \
    hipLaunchKernelGGL((ce_mean_fwd_kernel<LM, R>), grid, block, 0, s, logits, (const long long*)target, (double2*)partial, L, total, \
                       (long long)V, (long long)stride_l, (long long)stride_v, (long long)ignore_index)
    if (L <= 16) { if (rows) NEXTOU_CE_FWD(16, true); else NEXTOU_CE_FWD(16, false); }
    else { if (rows) NEXTOU_CE_FWD(32, true); else NEXTOU_CE_FWD(32, false); }
#undef NEXTOU_CE_FWD
    return check_launch("ce_mean_fwd_kernel");
}

extern "C" int nextou_ce_mean_bwd(const float* logits, const int64_t* target, const float* scale_dev, float* grad_logits, int B, int L,
                                  int64_t V, int64_t stride_l, int64_t stride_v, int64_t ignore_index, nextou_stream_t stream) {
    if (int e = check_ce_mean("ce_mean_bwd", logits, target, grad_logits, B, L, V, stride_l, stride_v)) return e;
    NEXTOU_REQUIRE(scale_dev != nullptr, "ce_mean_bwd: null scale");
    hipStream_t s = (hipStream_t)stream;
    const long long total = (long long)B * V;
    const bool rows = stride_l == 1 && L % 2 == 0 && ((reinterpret_cast<uintptr_t>(logits) | reinterpret_cast<uintptr_t>(grad_logits)) & 15u) == 0;
    ProfScope prof(s, kBoundHbm, (8.0 * L + 8.0) * (double)total, "ce_mean_bwd_kernel[B%d L%d V%lld %s]", B, L, (long long)V,
                   stride_l == 1 ? "rows" : "planes");
    long long blocks = cdiv64(total, 256);
    if (blocks > 8192) blocks = 8192;
    const dim3 grid((unsigned)blocks), block(256);
#define NEXTOU_CE_BWD(LM, R)                                                                                                     \
    hipLaunchKernelGGL((ce_mean_bwd_kernel<LM, R>), grid, block, 0, s, logits, (const long long*)target, scale_dev, grad_logits, L, \
                       total, (long long)V, (long long)stride_l, (long long)stride_v, (long long)ignore_index)
    if (L <= 16) { if (rows) NEXTOU_CE_BWD(16, true); else NEXTOU_CE_BWD(16, false); }
    else { if (rows) NEXTOU_CE_BWD(32, true); else NEXTOU_CE_BWD(32, false); }
#undef NEXTOU_CE_BWD
    return check_launch("ce_mean_bwd_kernel");
}
