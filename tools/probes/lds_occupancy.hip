// How many workgroups of T threads with B bytes of LDS (and V VGPRs' worth of nothing: the kernel is tiny) does the chip hold at once?
// Every workgroup counts itself in, records the largest count it saw, waits ~150 us, counts itself out; the grid is far larger than what fits.
// concurrent / 256 CUs = workgroups per CU for that LDS size.     hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_occupancy tools/probes/lds_occupancy.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void k_hold(int* cur, int* mx, long long spin)
{
    extern __shared__ char s[];
    s[threadIdx.x] = (char)threadIdx.x;
    __syncthreads();
    if (threadIdx.x == 0) {
        const int c = atomicAdd(cur, 1) + 1; atomicMax(mx, c);
        const long long t0 = wall_clock64(); while (wall_clock64() - t0 < spin) { atomicMax(mx, __hip_atomic_load(cur, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); __builtin_amdgcn_s_sleep(64); }
        atomicSub(cur, 1);
    }
    __syncthreads();
    if (s[threadIdx.x] == 99 && spin < 0) mx[1] = 1;
}
int main()
{
    int *cur, *mx; hipMalloc(&cur, 4); hipMalloc(&mx, 8);
    hipDeviceProp_t pr; hipGetDeviceProperties(&pr, 0);
    printf("%s: %d CUs, sharedMemPerMultiprocessor %zu, sharedMemPerBlock %zu, maxSharedMemoryPerMultiProcessor %zu\n", pr.name, pr.multiProcessorCount, pr.sharedMemPerMultiprocessor, pr.sharedMemPerBlock, (size_t)pr.maxSharedMemoryPerMultiProcessor);
    hipFuncSetAttribute(reinterpret_cast<const void*>(k_hold), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int sizes[] = { 8192, 12288, 16384, 20480, 24416, 26624, 27306, 28672, 32416, 32768, 36864, 40272, 40416, 40960, 45056, 49152, 53248, 54272, 54784, 57344, 63272, 65536, 81920, 98304, 131072, 163840 };
    for (int threads : { 256, 512 })
        for (int b : sizes) {
            hipMemset(cur, 0, 4); hipMemset(mx, 0, 8);
            hipLaunchKernelGGL(k_hold, dim3(256 * 40), dim3(threads), b, 0, cur, mx, 15000LL);      // wall clock: 100 MHz -> 150 us
            if (hipDeviceSynchronize() != hipSuccess) { printf("threads %d lds %6d: launch failed (%s)\n", threads, b, hipGetErrorString(hipGetLastError())); continue; }
            int m = 0; hipMemcpy(&m, mx, 4, hipMemcpyDeviceToHost);
            printf("threads %d lds %6d B: %5d workgroups at once = %.2f per CU\n", threads, b, m, m / (double)pr.multiProcessorCount);
        }
    return 0;
}
