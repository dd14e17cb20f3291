// How many workgroups of a given dynamic-LDS size are resident per CU on this GPU?  A kernel that only spins for a fixed number of clock
// cycles is launched with grids of 1x, 2x, 3x ... the CU count; the first multiple at which the time doubles is the residency.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/lds_occupancy.hip -o /tmp/lds_occ && /tmp/lds_occ
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void spin_kernel(float* out, long long cycles) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = (float)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && lds[1] < 0.f) out[0] = lds[2];
}

int main() {
    hipDeviceProp_t p;
    hipGetDeviceProperties(&p, 0);
    const int cus = p.multiProcessorCount;
    printf("%s: %d CUs, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", p.name, cus, p.sharedMemPerBlock,
           p.maxSharedMemoryPerMultiProcessor);
    float* out;
    hipMalloc(&out, 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const long long cycles = 2000000;      // 100 MHz wall clock -> 20 ms?  (reported below in ms)
    for (int threads : {384, 512}) {
        for (int kb : {8, 16, 24, 32, 40, 50, 64, 80, 100, 160}) {
            const size_t lds = (size_t)kb * 1024;
            if (lds > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void*>(&spin_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            printf("threads %d lds %3d KB:", threads, kb);
            float base = 0.f;
            for (int mult = 1; mult <= 6; ++mult) {
                hipEventRecord(e0);
                hipLaunchKernelGGL(spin_kernel, dim3(cus * mult), dim3(threads), lds, 0, out, 20000LL);
                hipEventRecord(e1);
                hipEventSynchronize(e1);
                float ms;
                hipEventElapsedTime(&ms, e0, e1);
                if (mult == 1) base = ms;
                printf("  x%d %.2f", mult, ms / base);
            }
            printf("   (base %.3f ms)%s\n", base, hipGetLastError() == hipSuccess ? "" : "  LAUNCH ERROR");
        }
    }
    return 0;
}
