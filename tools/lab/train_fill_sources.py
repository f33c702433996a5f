#!/usr/bin/env python3
"""Where do the fill / zero launches of a training step come from?  Python-level: zero_ / fill_ / zeros / zeros_like / full /
new_zeros are wrapped for ONE eager step and counted by their first frame inside this repository (fills made by C++ autograd
internals do not pass through Python and are not seen here: compare the total with the kernel trace).
    E3D_TRAIN_GRAPH=0 python tools/lab/train_fill_sources.py [structure|sequence]"""
import collections, os, sys, traceback
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["E3D_TRAIN_GRAPH"] = "0"
import bench_train  # noqa: E402

counts = collections.Counter()
active = [False]


def site():
    for fr in reversed(traceback.extract_stack()[:-2]):
        if ROOT in fr.filename and "train_fill_sources" not in fr.filename:
            return f"{os.path.relpath(fr.filename, ROOT)}:{fr.lineno} {fr.name}"
    fr = traceback.extract_stack()[-3]
    return f"(outside the repo) {fr.filename.split('/')[-1]}:{fr.lineno}"


def wrap(owner, name):
    orig = getattr(owner, name)

    def f(*a, **k):
        if active[0]:
            counts[(name, site())] += 1
        return orig(*a, **k)
    setattr(owner, name, f)


for owner, names in ((torch.Tensor, ("zero_", "fill_", "new_zeros", "new_full")), (torch, ("zeros", "zeros_like", "full", "full_like", "ones", "ones_like"))):
    for n in names:
        wrap(owner, n)

# one eager step after warm-up, instrumented
orig_run = bench_train.run
model = sys.argv[1] if len(sys.argv) > 1 else "structure"
import time
real_perf = time.perf_counter
state = {"n": 0}


def perf():
    # bench_train.run calls perf_counter right before its single host-timing step: switch the counting on for that step
    state["n"] += 1
    active[0] = state["n"] == 1
    return real_perf()


time.perf_counter = perf
bench_train.time.perf_counter = perf
bench_train.run(model, steps=1, warmup=2, graph=False)
time.perf_counter = real_perf
tot = sum(counts.values())
print(f"{model}: {tot} Python-level fill / zero calls in one eager step")
for (name, where), n in counts.most_common(30):
    print(f"{n:5d}  {name:11s} {where}")
