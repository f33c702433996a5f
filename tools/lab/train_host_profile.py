#!/usr/bin/env python3
"""Lab: cProfile of the Python side of training steps (structure model, bench configuration): where does the host spend
the ~25 ms it needs to enqueue one step?"""
import cProfile, os, pstats, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_train as BT  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "structure"
BT.run(name, steps=2)             # warm: kernels loaded, caches built
pr = cProfile.Profile()
pr.enable()
BT.run(name, steps=6, warmup=1)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(45)
st.sort_stats("cumtime").print_stats(60)
