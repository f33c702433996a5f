#!/usr/bin/env python3
"""Lab: the same question inside torch's capture machinery (its HIP runtime instance, torch.cuda.graph's private pool, the
caching allocator churning between replays): hipMemsetAsync through ctypes on torch tensors + an atomic consumer."""
import ctypes, os, torch
ZERO = os.environ.get("REPRO_ZERO", "memset")      # memset: hipMemsetAsync through ctypes | fill: torch's zero_() (a fill kernel)
STREAM = os.environ.get("REPRO_STREAM", "default")   # default: replays on torch's default (NULL) stream | side: on a created stream
torch.cuda.init()
# the HIP runtime INSTANCE torch itself uses: the file mapped into this process (a bare dlopen("libamdhip64.so") finds
# /opt/rocm's copy -- another file, hence a second runtime whose hipMemsetAsync knows nothing of torch's capturing stream;
# the first version of this script did that and "found" 150 M stale words)
path = next(ln.split()[-1] for ln in open("/proc/self/maps") if "libamdhip64" in ln)
hip = ctypes.CDLL(path)
print("HIP runtime:", path)
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
            if ZERO == "fill":
                b.zero_()
            else:
                assert hip.hipMemsetAsync(b.data_ptr(), 0, b.numel() * 4, torch.cuda.current_stream().cuda_stream) == 0
            b.index_add_(0, i, o)                                # atomicAdd consumer
run_stream = torch.cuda.Stream() if STREAM == "side" else torch.cuda.current_stream()
torch.cuda.set_stream(run_stream)
bad = 0
per = [0] * (2 * len(sizes))
first_bad = {}
for r in range(1000):
    junk = [torch.full((64 << (r % 14),), 1.2345e30, device=dev) for _ in range(3)]      # allocator churn + poison between replays
    del junk
    if r % 3 == 0:
        for b in bufs + inner:
            b.fill_(1.2345e30)                                   # stale values in the targets
    g.replay()
    torch.cuda.synchronize()
    for j, b in enumerate(bufs + inner):
        w = (b != 1.0).nonzero().flatten()
        if w.numel():
            per[j] += w.numel()
            first_bad.setdefault(j, (r, int(w[0]), int(w[-1]), w.numel(), float(b[w[0]])))
    bad = sum(per)
for j, n in enumerate(per):
    print(f"  buffer {j} ({'outside' if j < len(sizes) else 'inside '} the capture, {sizes[j % len(sizes)]} floats): {n} wrong words"
          + (f"; first at replay {first_bad[j][0]}: words {first_bad[j][1]}..{first_bad[j][2]} ({first_bad[j][3]} of them), value {first_bad[j][4]:.4g}" if j in first_bad else ""))
print(f"graph memset repro (torch capture, zeroing by {ZERO}, replays on the {STREAM} stream): 1000 replays, wrong words: {bad} -> "
      + ("MEMSET NODES DO NOT CLEAR" if bad else "memset nodes clear every time"))
