// Micro-benchmark: one-way latency of an agent-scope release -> acquire hand-over between two workgroups on different
// XCDs (the producer writes `bytes` of payload, then a flag; the consumer polls the flag and reads the payload back).
// hipcc --offload-arch=gfx950 -O3 scratch/handoff.hip -o scratch/_bin/handoff && scratch/_bin/handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(64) void k_pingpong(unsigned *flag, unsigned long long *buf, int iters, int words, int *bad)
{
    const int me = blockIdx.x;                       // 0 and 1: consecutive workgroups go to different XCDs
    if (me > 1) return;
    unsigned long long *mine = buf + me * 4096, *theirs = buf + (1 - me) * 4096;
    for (int it = 1; it <= iters; it++) {
        const unsigned turn = 2u * it - (me == 0 ? 1u : 0u);          // 1, 3, 5 .. written by 0; 2, 4, 6 .. by 1
        if (me == 1 || it > 1 || true) {
            if (me == 0) {
                for (int w = threadIdx.x; w < words; w += 64) mine[w] = (unsigned long long)turn * 1000003ull + w;
                if (threadIdx.x == 0) __hip_atomic_store(flag, turn, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
                // wait for the answer
                long spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != turn + 1u) { if (++spins > 50000000) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; } }
                for (int w = threadIdx.x; w < words; w += 64) if (theirs[w] != (unsigned long long)(turn + 1u) * 1000003ull + w) atomicAdd(bad, 1);
            } else {
                const unsigned want = 2u * it - 1u;
                long spins = 0;
                while (__hip_atomic_load(flag, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) != want) { if (++spins > 50000000) { if (threadIdx.x == 0) atomicAdd(bad, 1000000); return; } }
                for (int w = threadIdx.x; w < words; w += 64) if (theirs[w] != (unsigned long long)want * 1000003ull + w) atomicAdd(bad, 1);
                for (int w = threadIdx.x; w < words; w += 64) mine[w] = (unsigned long long)(want + 1u) * 1000003ull + w;
                if (threadIdx.x == 0) __hip_atomic_store(flag, want + 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
}

int main()
{
    unsigned *flag; unsigned long long *buf; int *bad;
    CK(hipMalloc(&flag, 256)); CK(hipMalloc(&buf, 2 * 4096 * 8)); CK(hipMalloc(&bad, 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int sizes[] = {0, 16, 128, 2048};
    for (int words : sizes) {
        const int iters = 20000;
        CK(hipMemset(flag, 0, 256)); CK(hipMemset(bad, 0, 4));
        k_pingpong<<<2, 64>>>(flag, buf, 100, words, bad); CK(hipDeviceSynchronize());
        CK(hipMemset(flag, 0, 256));
        CK(hipEventRecord(e0)); k_pingpong<<<2, 64>>>(flag, buf, iters, words, bad); CK(hipEventRecord(e1)); CK(hipDeviceSynchronize());
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        int hb; CK(hipMemcpy(&hb, bad, 4, hipMemcpyDeviceToHost));
        printf("payload %5d B: %.2f us per one-way hand-over (write + release, poll + acquire, read), stale / failed %d\n", words * 8, ms * 1000.0 / (2 * iters), hb);
    }
    return 0;
}
