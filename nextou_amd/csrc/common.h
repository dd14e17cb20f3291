// Shared host-side helpers for libnextou_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdio>
#include <cstdint>
#include "../../include/nextou_hip.h"

namespace nextou {

// Thread-local message returned by nextou_last_error().
char* error_buffer();
int fail(int code, const char* fmt, ...);

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail((int)e, "%s: %s", what, hipGetErrorString(e));
    return 0;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }
inline int64_t cdiv64(int64_t a, int64_t b) { return (a + b - 1) / b; }

// Optional per-launch timing with HIP events on the launch stream (bench.py's roofline object).
// Disabled by default: a disabled scope costs one branch and records nothing, so captured
// hipGraphs never see an event record.
enum ProfBound { kBoundHbm = 0, kBoundMfma = 1 };
struct ProfScope {
    ProfScope(hipStream_t s, int bound, double work, const char* fmt, ...);
    ~ProfScope();
    int slot;
    hipStream_t stream;
};

// Channels-last volume and Swin window geometry of the layout kernels (layout_ops.hip) and of the window-fused K2 (mr_aggregate.hip)
struct Vol {
    int D, H, W;    // channels-last volume
};
struct Win {
    int wd, wh, ww, sd, sh, sw;   // window size, cyclic shift
};

// row index (without batch) of point p of window `win` after the cyclic shift: the partitioned tensor is roll(x, -shift),
// i.e. window coordinate (d, h, w) reads x at ((d + sd) mod D, ...)
__device__ __forceinline__ long long window_point_row(int win, int p, const Vol& v, const Win& w, int nH, int nW) {
    const int wi_w = win % nW, t = win / nW;
    const int wi_h = t % nH, wi_d = t / nH;
    const int pw = p % w.ww, t2 = p / w.ww;
    const int ph = t2 % w.wh, pd = t2 / w.wh;
    int d = wi_d * w.wd + pd + w.sd, h = wi_h * w.wh + ph + w.sh, x = wi_w * w.ww + pw + w.sw;
    if (d >= v.D) d -= v.D;
    if (h >= v.H) h -= v.H;
    if (x >= v.W) x -= v.W;
    return ((long long)d * v.H + h) * v.W + x;
}

// Transposed convolution with kernel == stride (the decoder's up-convolutions): output volume (B, D2, H2, W2) = input volume times the stride
struct UpShuffle {
    int D2, H2, W2, sd, sh, sw;
};
__device__ __forceinline__ long long upconv_row(long long row, const UpShuffle& u) {       // output row -> p_in * T + t
    const unsigned w2 = (unsigned)(row % u.W2);
    const long long r1 = row / u.W2;
    const unsigned h2 = (unsigned)(r1 % u.H2);
    const long long r2 = r1 / u.H2;
    const unsigned d2 = (unsigned)(r2 % u.D2);
    const long long b = r2 / u.D2;
    const unsigned t = ((d2 % u.sd) * u.sh + h2 % u.sh) * u.sw + w2 % u.sw;
    const long long p_in = ((b * (u.D2 / u.sd) + d2 / u.sd) * (u.H2 / u.sh) + h2 / u.sh) * (u.W2 / u.sw) + w2 / u.sw;
    return p_in * (u.sd * u.sh * u.sw) + t;
}

// LDS budget one workgroup of the gather kernels may claim (keeps >= 2 workgroups per CU).
constexpr int kGatherLdsBytes = 64 * 1024;

}  // namespace nextou

#define NEXTOU_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::nextou::fail(NEXTOU_EINVAL, __VA_ARGS__); \
    } while (0)
