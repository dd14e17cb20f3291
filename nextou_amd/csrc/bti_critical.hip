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

namespace nextou {

// labels[b, v] = first arg-max over L class planes of logits (B, L, V).
// One thread per 4 consecutive voxels (16-B loads per class plane) when V % 4 == 0.
template <bool VEC4>
__global__ __launch_bounds__(256) void argmax_labels_kernel(const float* __restrict__ logits,
                                                            uint8_t* __restrict__ labels, int L,
                                                            long long V) {
    const int b = blockIdx.y;
    const float* lb = logits + (size_t)b * L * V;
    uint8_t* ob = labels + (size_t)b * V;
    const long long stride = (long long)gridDim.x * blockDim.x;
    if (VEC4) {
        const long long V4 = V >> 2;
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V4; i += stride) {
            float4 best = reinterpret_cast<const float4*>(lb)[i];
            uchar4 arg = make_uchar4(0, 0, 0, 0);
            for (int l = 1; l < L; ++l) {
                const float4 v = reinterpret_cast<const float4*>(lb + (size_t)l * V)[i];
                if (v.x > best.x) { best.x = v.x; arg.x = (uint8_t)l; }
                if (v.y > best.y) { best.y = v.y; arg.y = (uint8_t)l; }
                if (v.z > best.z) { best.z = v.z; arg.z = (uint8_t)l; }
                if (v.w > best.w) { best.w = v.w; arg.w = (uint8_t)l; }
            }
            reinterpret_cast<uchar4*>(ob)[i] = arg;
        }
    } else {
        for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < V; i += stride) {
            float best = lb[i];
            uint8_t arg = 0;
            for (int l = 1; l < L; ++l) {
                const float v = lb[(size_t)l * V + i];
                if (v > best) { best = v; arg = (uint8_t)l; }
            }
            ob[i] = arg;
        }
    }
}

// One thread per voxel; the two LUTs live in LDS; neighbour labels come through L1/L2 (each label
// byte is touched by at most 27 threads of neighbouring rows).
// full_box: box neighbourhood of radius `rad` (connectivity 26 / 8); otherwise the 6 / 4 cross.
__global__ __launch_bounds__(256) void bti_critical_kernel(
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

}  // namespace nextou

using namespace nextou;

extern "C" int nextou_argmax_labels(const float* logits, uint8_t* labels, int B, int L, int64_t V,
                                    nextou_stream_t stream) {
    NEXTOU_REQUIRE(logits && labels, "argmax_labels: null pointer");
    NEXTOU_REQUIRE(B > 0 && L > 0 && L <= 256 && V > 0 && B <= 65535,
                   "argmax_labels: bad size B=%d L=%d V=%lld (L <= 256)", B, L, (long long)V);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (V % 4 == 0) && ((reinterpret_cast<uintptr_t>(logits) & 15u) == 0) &&
                     ((reinterpret_cast<uintptr_t>(labels) & 3u) == 0);
    ProfScope prof(s, kBoundHbm, (4.0 * L + 1.0) * B * (double)V, "argmax_labels_kernel[B%d L%d V%lld]", B, L, (long long)V);
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
    ProfScope prof((hipStream_t)stream, kBoundHbm, 2.0 * B * (double)V, "bti_critical_kernel[B%d %dx%dx%d c%d]", B, D, H, W,
                   connectivity);
    hipLaunchKernelGGL(bti_critical_kernel, dim3((unsigned)cdiv64(V, 256), 1, B), dim3(256), 0,
                       (hipStream_t)stream, labels, lut_a, lut_c, n_labels, critical, D, H, W, full_box,
                       full_box ? min_thick : 1);
    return check_launch("bti_critical_kernel");
}
