#!/usr/bin/env python3
"""Lab: the same question inside torch's capture machinery (its HIP runtime instance, torch.cuda.graph's private pool, the
caching allocator churning between replays): hipMemsetAsync through ctypes on torch tensors + an atomic consumer."""
import ctypes, torch
hip = ctypes.CDLL("libamdhip64.so")          # the runtime torch has already loaded (same SONAME -> same instance)
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
dev = "cuda:0"
sizes = [768, 1024, 2304, 3072, 20, 8, 768 * 768, 1]
idx = [torch.arange(n, device=dev) for n in sizes]
ones = [torch.ones(n, device=dev) for n in sizes]
g = torch.cuda.CUDAGraph()
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    bufs = [torch.empty(n, device=dev) for n in sizes]           # allocated OUTSIDE the capture (as parameters' .grad are)
    inner = None
    torch.cuda.synchronize()
    with torch.cuda.graph(g, stream=side):
        inner = [torch.empty(n, device=dev) for n in sizes]      # ... and INSIDE it (the graph's private pool)
        for b, i, o in zip(bufs + inner, idx + idx, ones + ones):
            assert hip.hipMemsetAsync(b.data_ptr(), 0, b.numel() * 4, torch.cuda.current_stream().cuda_stream) == 0
            b.index_add_(0, i, o)                                # atomicAdd consumer
bad = 0
for r in range(1000):
    junk = [torch.full((64 << (r % 14),), 1.2345e30, device=dev) for _ in range(3)]      # allocator churn + poison between replays
    del junk
    if r % 3 == 0:
        for b in bufs + inner:
            b.fill_(1.2345e30)                                   # stale values in the targets
    g.replay()
    torch.cuda.synchronize()
    bad += sum(int((b != 1.0).sum()) for b in bufs + inner)
print(f"graph memset repro (torch capture + ctypes hipMemsetAsync): 1000 replays, wrong words: {bad} -> "
      + ("MEMSET NODES DO NOT CLEAR" if bad else "memset nodes clear every time"))
