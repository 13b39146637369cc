"""How fast can a pageable numpy array reach the GPU on this box? (torch only as a convenient HIP front end)"""
import time, numpy as np, torch
n = 138_240_000 // 8 * 8
a = np.random.rand(n // 8)
t = torch.from_numpy(a)
d = torch.empty_like(t, device="cuda")
torch.cuda.synchronize()
def tm(f, reps=5):
    ts = []
    for _ in range(reps):
        torch.cuda.synchronize(); t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return min(ts), sorted(ts)[len(ts) // 2]
gb = a.nbytes / 1e9
m, med = tm(lambda: d.copy_(t)); print("pageable -> device: %.1f ms (%.1f GB/s)" % (m * 1e3, gb / m))
p = torch.empty_like(t).pin_memory()
m, med = tm(lambda: p.copy_(t)); print("memcpy pageable -> pinned (1 thread): %.1f ms (%.1f GB/s)" % (m * 1e3, gb / m))
m, med = tm(lambda: d.copy_(p, non_blocking=True)); print("pinned -> device: %.1f ms (%.1f GB/s)" % (m * 1e3, gb / m))
torch.set_num_threads(8)
m, med = tm(lambda: p.copy_(t)); print("memcpy pageable -> pinned (torch, 8 threads): %.1f ms (%.1f GB/s)" % (m * 1e3, gb / m))
t0 = time.perf_counter(); r = torch.cuda.cudart().cudaHostRegister(t.data_ptr(), a.nbytes, 0); t1 = time.perf_counter()
print("hipHostRegister: rc %s %.1f ms" % (r, (t1 - t0) * 1e3))
m, med = tm(lambda: d.copy_(t, non_blocking=True)); print("registered -> device: %.1f ms (%.1f GB/s)" % (m * 1e3, gb / m))
t0 = time.perf_counter(); torch.cuda.cudart().cudaHostUnregister(t.data_ptr()); print("unregister %.1f ms" % ((time.perf_counter() - t0) * 1e3))
