#!/usr/bin/env python3
"""Where do the fill / zero kernels of a structure training step come from?  (torch.profiler with stacks.)"""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_train  # noqa: E402
from torch.profiler import profile, ProfilerActivity  # noqa: E402

# reuse bench_train's setup by running it with a profiler around the timed steps
orig_sync = torch.cuda.synchronize
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    bench_train.run("structure", steps=1, warmup=1)
counts = collections.Counter()
for ev in prof.events():
    if ev.name in ("aten::zero_", "aten::fill_", "aten::zeros", "aten::zeros_like", "aten::new_zeros"):
        stack = [s for s in (ev.stack or []) if "e3-invaraint" in s or "tools/" in s or "torch/optim" in s or "clip_grad" in s or "autograd" in s]
        counts[(ev.name, tuple(stack[:3]))] += 1
for (name, stack), n in counts.most_common(25):
    print(n, name, " <- ".join(s.split("/")[-1] for s in stack))
