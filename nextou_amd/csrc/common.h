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

// LDS budget one workgroup of the gather kernels may claim (keeps >= 2 workgroups per CU).
constexpr int kGatherLdsBytes = 64 * 1024;

}  // namespace nextou

#define NEXTOU_REQUIRE(cond, ...)                                   \
    do {                                                            \
        if (!(cond)) return ::nextou::fail(NEXTOU_EINVAL, __VA_ARGS__); \
    } while (0)
