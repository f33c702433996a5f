#!/usr/bin/env python3
"""Lab: two ranks on ONE GPU (gloo), the data-parallel training step of tools/bench_train.py several times in one process,
padded / trimmed in the order given: python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1
tools/lab/ddp_trim_check.py 0 0 1 1   (0 = padded, 1 = trimmed).  Each run ends with bench_train's "ranks diverged" check."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from e3diff_amd import sharding  # noqa: E402
import bench_train  # noqa: E402

rank, world, _ = sharding.init_distributed("gloo")
for i, flag in enumerate(sys.argv[1:]):
    try:
        r = bench_train.run("sequence", batch=4, seq_len=64, steps=5, warmup=2, device="cuda:0", ddp=True, layers=2, seed=rank,
                            trim=flag == "1", graph=None if os.environ.get("EAGER") != "1" else False)
        msg = f"ok frame={r['frame']} graph={r['graph_replay']} loss={r['loss']:.5f}"
    except AssertionError as e:
        msg = f"FAILED {e}"
    if rank == 0:
        print(f"run {i} trim={flag}: {msg}", flush=True)
