// Read-only streaming bandwidth on MI355X for the access patterns of K6's statistics kernels (why does bn_cl_stats sit at 0.57 of 8 TB/s?).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/read_bw.hip -o tools/micro/read_bw && tools/micro/read_bw
// Variants over one 880 MB fp32 buffer (the stage-0 activation of cfg 2: 2 x 40 x 64 x 224 x 192):
//   A  256 threads, 16-byte loads, 1 KB-aligned wave pieces, fp32 sums, U loads in flight
//   B  as A with float64 sums of v and v*v (the statistics kernel's arithmetic)
//   C  250 active threads of 256 (tact = floor(256 / C) * C for C = 40): 4 000-byte pieces, wave loads straddle lines
//   D  as C with float64 sums (= bn_cl_stats_kernel's inner loop)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int U, bool F64, int TACT>
__global__ __launch_bounds__(256) void read_kernel(const float4* __restrict__ x, double* __restrict__ out, long long n4, long long span4) {
    double s = 0.0, q = 0.0;
    float fs = 0.f;
    if ((int)threadIdx.x < TACT) {
        const long long base = (long long)blockIdx.x * span4;
        const long long end = min(n4, base + span4);
        long long e = base + threadIdx.x;
        for (; e + (U - 1) * TACT < end; e += (long long)U * TACT) {
            float4 p[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                typedef float v4f __attribute__((ext_vector_type(4)));
                const v4f t = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(x + e + u * TACT));
                p[u] = make_float4(t.x, t.y, t.z, t.w);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (F64) {
                    const double a = p[u].x, b = p[u].y, c = p[u].z, d = p[u].w;
                    s += a; q = fma(a, a, q); s += b; q = fma(b, b, q); s += c; q = fma(c, c, q); s += d; q = fma(d, d, q);
                } else {
                    fs += (p[u].x + p[u].y) + (p[u].z + p[u].w);
                }
            }
        }
    }
    if (s + q + fs == 12345.678) out[blockIdx.x] = s;
}

template <int U, bool F64, int TACT>
static void run(const char* name, const float4* x, double* out, long long n4, int iters_per_block) {
    const long long span4 = (long long)iters_per_block * U * TACT;
    const int blocks = (int)((n4 + span4 - 1) / span4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) hipLaunchKernelGGL((read_kernel<U, F64, TACT>), dim3(blocks), dim3(256), 0, 0, x, out, n4, span4);
    hipEventRecord(e0);
    const int reps = 10;
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((read_kernel<U, F64, TACT>), dim3(blocks), dim3(256), 0, 0, x, out, n4, span4);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    printf("%-44s blocks %6d  %8.1f us  %7.1f GB/s  %.3f of 8 TB/s\n", name, blocks, us, n4 * 16.0 / us / 1e3, n4 * 16.0 / us / 1e3 / 8000.0);
}

int main() {
    const long long n = 2ll * 40 * 64 * 224 * 192, n4 = n / 4;
    float4* x;
    double* out;
    hipMalloc(&x, n * 4);
    hipMalloc(&out, 1 << 20);
    hipMemset(x, 0, n * 4);
    for (int ipb : {5, 20}) {
        printf("-- %d unrolled iterations per workgroup\n", ipb);
        run<4, false, 256>("A  aligned, fp32 sums, 4 loads in flight", x, out, n4, ipb);
        run<8, false, 256>("A  aligned, fp32 sums, 8 loads in flight", x, out, n4, ipb);
        run<4, true, 256>("B  aligned, float64 sums, 4 in flight", x, out, n4, ipb);
        run<8, true, 256>("B  aligned, float64 sums, 8 in flight", x, out, n4, ipb);
        run<4, false, 250>("C  250 of 256 threads, fp32 sums, 4 in flight", x, out, n4, ipb);
        run<4, true, 250>("D  250 of 256 threads, float64 sums, 4 in flight", x, out, n4, ipb);
        run<8, true, 250>("D  250 of 256 threads, float64 sums, 8 in flight", x, out, n4, ipb);
    }
    return 0;
}
