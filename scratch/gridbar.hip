// Micro-benchmark: cost of a grid-wide barrier (agent-scope release/acquire on one counter) vs a kernel boundary.
// hipcc --offload-arch=gfx950 -O3 scratch/gridbar.hip -o gpurun_out/gridbar && ./gridbar
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__device__ __forceinline__ bool grid_barrier(unsigned *ctr, unsigned &target, unsigned nwg)
{
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        target += nwg;
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        long spins = 0;
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > 20000000) { ok = false; break; }      // never hang the box
        }
    }
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void k_bar(unsigned *ctr, int *data, int iters, int *bad)
{
    unsigned target = 0;
    const int g = blockIdx.x, n = gridDim.x;
    for (int it = 0; it < iters; it++) {
        if (threadIdx.x == 0) data[g] = it * 7 + g;
        if (!grid_barrier(ctr, target, n)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
        const int nb = (g + 1 + it) % n;
        if (threadIdx.x == 0 && data[nb] != it * 7 + nb) atomicAdd(bad, 1);
        if (!grid_barrier(ctr, target, n)) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; }
    }
}

__global__ __launch_bounds__(256) void k_step(int *data, int it) { if (threadIdx.x == 0) data[blockIdx.x] = it; }

int main()
{
    unsigned *ctr; int *data, *bad;
    CK(hipMalloc(&ctr, 256)); CK(hipMalloc(&data, 1 << 20)); CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int grids[] = {64, 256, 512, 1024};
    for (int n : grids) {
        const int iters = 2000;
        CK(hipMemset(ctr, 0, 256)); CK(hipMemset(bad, 0, 4)); CK(hipMemset(data, 0, 1 << 20));
        k_bar<<<n, 256>>>(ctr, data, 10, bad); CK(hipDeviceSynchronize());
        CK(hipMemset(ctr, 0, 256));
        CK(hipEventRecord(e0)); k_bar<<<n, 256>>>(ctr, data, iters, bad); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("grid %4d: %.2f us per barrier (2 per iteration), stale reads %d\n", n, ms * 1000.0 / (2 * iters), hb);
    }
    for (int n : grids) {
        const int iters = 2000;
        CK(hipEventRecord(e0));
        for (int it = 0; it < iters; it++) k_step<<<n, 256>>>(data, it);
        CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("grid %4d: %.2f us per back-to-back kernel\n", n, ms * 1000.0 / iters);
    }
    return 0;
}
