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

// LDS budget one workgroup of the gather kernels may claim (keeps >= 2 workgroups per CU).
constexpr int kGatherLdsBytes = 64 * 1024;

}  // namespace nextou

#define NEXTOU_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::nextou::fail(NEXTOU_EINVAL, __VA_ARGS__); \
    } while (0)
