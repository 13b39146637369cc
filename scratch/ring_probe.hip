// How fast does a pageable buffer reach the GPU through a ring of pinned staging buffers?  (scratch/ring_probe.hip;
// hipcc -O2 -o ring_probe ring_probe.hip -lpthread)  Variants: slots, piece size, staging threads, pinned-memory flags,
// waiting by hipEventSynchronize or by polling, one or two DMA streams.
#include <hip/hip_runtime.h>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static double ring(const char *src, char *dst, size_t bytes, int slots, size_t piece, int nthreads, unsigned flags, bool poll, int nstreams)
{
    char *ring = nullptr;
    if (hipHostMalloc((void **)&ring, (size_t)slots * piece, flags) != hipSuccess) return -1;
    memset(ring, 1, (size_t)slots * piece);
    hipStream_t st[2];
    for (int i = 0; i < 2; i++) hipStreamCreateWithFlags(&st[i], hipStreamNonBlocking);
    std::vector<hipEvent_t> ev(slots);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const size_t n = (bytes + piece - 1) / piece;
    std::vector<std::atomic<int>> staged(n);
    for (auto &f : staged) f.store(0);
    std::atomic<long long> released(slots), next(0);
    const double t0 = now();
    auto worker = [&]() {
        for (;;) {
            const long long i = next.fetch_add(1);
            if (i >= (long long)n) return;
            while (released.load(std::memory_order_acquire) <= i) std::this_thread::yield();
            const size_t off = (size_t)i * piece, m = std::min(piece, bytes - off);
            memcpy(ring + (size_t)(i % slots) * piece, src + off, m);
            staged[(size_t)i].store(1, std::memory_order_release);
        }
    };
    std::vector<std::thread> pool;
    for (int t = 0; t < nthreads; t++) pool.emplace_back(worker);
    const int inflight = slots / 2;
    for (size_t i = 0; i < n; i++) {
        while (!staged[i].load(std::memory_order_acquire)) std::this_thread::yield();
        const size_t off = i * piece, m = std::min(piece, bytes - off);
        const int slot = (int)(i % slots);
        hipMemcpyAsync(dst + off, ring + (size_t)slot * piece, m, hipMemcpyHostToDevice, st[i % nstreams]);
        hipEventRecord(ev[slot], st[i % nstreams]);
        if ((int)i + 1 >= inflight) {
            const size_t done = i + 1 - inflight;
            if (poll) while (hipEventQuery(ev[done % slots]) == hipErrorNotReady) {}
            else hipEventSynchronize(ev[done % slots]);
            released.store((long long)(done + 1 + slots), std::memory_order_release);
        }
    }
    released.store((long long)n + slots, std::memory_order_release);
    for (auto &t : pool) t.join();
    hipStreamSynchronize(st[0]); hipStreamSynchronize(st[1]);
    const double dt = now() - t0;
    for (auto &e : ev) hipEventDestroy(e);
    hipStreamDestroy(st[0]); hipStreamDestroy(st[1]);
    hipHostFree(ring);
    return dt;
}

int main()
{
    const size_t bytes = (size_t)1382400000;
    char *src = (char *)malloc(bytes);
    for (size_t i = 0; i < bytes; i += 4096) src[i] = (char)i;
    char *dst = nullptr;
    hipMalloc((void **)&dst, bytes);
    for (int rep = 0; rep < 3; rep++) {
        const double t0 = now();
        hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice);
        const double dt = now() - t0;
        printf("hipMemcpy pageable: %.1f ms  %.1f GB/s\n", dt * 1e3, bytes / dt / 1e9);
    }
    struct V { const char *name; int slots; size_t piece; int threads; unsigned flags; bool poll; int streams; };
    const V vs[] = {
        {"library: 8 x 4 MB, 4 threads, sync", 8, 4u << 20, 4, hipHostMallocDefault, false, 1},
        {"8 x 4 MB, 4 threads, poll", 8, 4u << 20, 4, hipHostMallocDefault, true, 1},
        {"8 x 4 MB, 8 threads, poll", 8, 4u << 20, 8, hipHostMallocDefault, true, 1},
        {"16 x 4 MB, 8 threads, poll", 16, 4u << 20, 8, hipHostMallocDefault, true, 1},
        {"16 x 4 MB, 8 threads, poll, 2 streams", 16, 4u << 20, 8, hipHostMallocDefault, true, 2},
        {"16 x 2 MB, 8 threads, poll, 2 streams", 16, 2u << 20, 8, hipHostMallocDefault, true, 2},
        {"8 x 4 MB, 4 threads, sync, non-coherent", 8, 4u << 20, 4, hipHostMallocNonCoherent, false, 1},
        {"8 x 4 MB, 4 threads, sync, write-combined", 8, 4u << 20, 4, hipHostMallocWriteCombined, false, 1},
        {"16 x 4 MB, 8 threads, poll, 2 streams, non-coherent", 16, 4u << 20, 8, hipHostMallocNonCoherent, true, 2},
        {"16 x 8 MB, 12 threads, poll, 2 streams", 16, 8u << 20, 12, hipHostMallocDefault, true, 2},
        {"32 x 4 MB, 16 threads, poll, 2 streams", 32, 4u << 20, 16, hipHostMallocDefault, true, 2},
    };
    for (const V &v : vs) {
        double best = 1e9;
        for (int rep = 0; rep < 3; rep++) { const double dt = ring(src, dst, bytes, v.slots, v.piece, v.threads, v.flags, v.poll, v.streams); if (dt > 0 && dt < best) best = dt; }
        printf("%-55s %.1f ms  %.1f GB/s\n", v.name, best * 1e3, bytes / best / 1e9);
    }
    // memcpy alone: how fast do N threads stage?
    for (int nt : {1, 4, 8, 16}) {
        char *p = nullptr;
        hipHostMalloc((void **)&p, (size_t)256 << 20, hipHostMallocDefault);
        memset(p, 1, (size_t)256 << 20);
        const double t0 = now();
        std::vector<std::thread> pool;
        for (int t = 0; t < nt; t++) pool.emplace_back([&, t]() {
            const size_t per = bytes / nt;
            for (size_t off = 0; off < per; off += (4u << 20)) memcpy(p + ((size_t)t * (16u << 20)) % ((size_t)240 << 20), src + t * per + off, std::min((size_t)(4u << 20), per - off));
        });
        for (auto &t : pool) t.join();
        const double dt = now() - t0;
        printf("memcpy pageable -> pinned, %d threads: %.1f GB/s\n", nt, bytes / dt / 1e9);
        hipHostFree(p);
    }
    return 0;
}
