// Micro-benchmark: throughput of LDS atomics on gfx950 — ds_add_f32 vs ds_add_u32 vs ds_add_u64 — under the address patterns of the
// K2 backward scatter (profiles/r04_lds_atomics.md).  Each wave issues ROUNDS atomic instructions on a table of M accumulators.
//   pattern 0: random addresses (uniform over M)       pattern 1: clustered (runs of 1..8 lanes share an address, as arg-max winners do)
//   pattern 2: conflict-free (lane l -> bank l)        build: hipcc --offload-arch=gfx950 -O3 -munsafe-fp-atomics
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <typename T>
__global__ __launch_bounds__(256) void bench(const unsigned short* __restrict__ addr, int M, int rounds, T* out) {
    extern __shared__ unsigned long long smem[];
    T* acc = reinterpret_cast<T*>(smem);
    for (int i = threadIdx.x; i < M; i += blockDim.x) acc[i] = T(0);
    __syncthreads();
    const unsigned short* a = addr + (size_t)blockIdx.x * rounds * 256;
    T v = T(threadIdx.x + 1);
    for (int r = 0; r < rounds; r += 4) {
        const unsigned short a0 = a[(r + 0) * 256 + threadIdx.x], a1 = a[(r + 1) * 256 + threadIdx.x];
        const unsigned short a2 = a[(r + 2) * 256 + threadIdx.x], a3 = a[(r + 3) * 256 + threadIdx.x];
        atomicAdd(&acc[a0], v);
        atomicAdd(&acc[a1], v);
        atomicAdd(&acc[a2], v);
        atomicAdd(&acc[a3], v);
    }
    __syncthreads();
    T s = T(0);
    for (int i = threadIdx.x; i < M; i += blockDim.x) s += acc[i];
    if (s == T(12345)) out[blockIdx.x] = s;
}

template <typename T>
float run(const unsigned short* d_addr, int M, int rounds, int blocks, T* d_out) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int w = 0; w < 2; ++w) bench<T><<<blocks, 256, M * sizeof(T)>>>(d_addr, M, rounds, d_out);
    hipEventRecord(e0);
    for (int w = 0; w < 5; ++w) bench<T><<<blocks, 256, M * sizeof(T)>>>(d_addr, M, rounds, d_out);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0; hipEventElapsedTime(&ms, e0, e1);
    return ms / 5;
}

int main() {
    const int blocks = 2048, rounds = 256;
    unsigned long long* d_out; hipMalloc(&d_out, blocks * 8);
    for (int M : {168, 1344}) {
        for (int pattern = 0; pattern < 3; ++pattern) {
            std::vector<unsigned short> h((size_t)blocks * rounds * 256);
            unsigned s = 12345u;
            auto rnd = [&]() { s = s * 1664525u + 1013904223u; return s >> 8; };
            for (size_t i = 0; i < h.size();) {
                if (pattern == 0) { h[i++] = rnd() % M; }
                else if (pattern == 1) { int run_len = 1 + rnd() % 8; unsigned short m = rnd() % M; for (int k = 0; k < run_len && i < h.size(); ++k) h[i++] = m; }
                else { h[i] = (i % 256) % M; ++i; }
            }
            unsigned short* d_addr; hipMalloc(&d_addr, h.size() * 2);
            hipMemcpy(d_addr, h.data(), h.size() * 2, hipMemcpyHostToDevice);
            const double instr = (double)blocks * rounds * 4;       // wave-instructions
            float tf = run<float>(d_addr, M, rounds, blocks, (float*)d_out);
            float tu = run<unsigned int>(d_addr, M, rounds, blocks, (unsigned int*)d_out);
            float tl = run<unsigned long long>(d_addr, M, rounds, blocks, d_out);
            printf("M %4d pattern %d: f32 %.3f ms  u32 %.3f ms  u64 %.3f ms   (%.0f wave-atomics; per CU-cycle @2.4GHz,256CU: f32 %.1f u32 %.1f u64 %.1f cycles/instr)\n",
                   M, pattern, tf, tu, tl, instr, tf * 1e-3 * 2.4e9 * 256 / instr, tu * 1e-3 * 2.4e9 * 256 / instr, tl * 1e-3 * 2.4e9 * 256 / instr);
            hipFree(d_addr);
        }
    }
    return 0;
}
