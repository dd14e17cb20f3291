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

// LDS budget one workgroup of the gather kernels may claim (keeps >= 2 workgroups per CU).
constexpr int kGatherLdsBytes = 64 * 1024;

}  // namespace nextou

#define NEXTOU_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::nextou::fail(NEXTOU_EINVAL, __VA_ARGS__); \
    } while (0)
