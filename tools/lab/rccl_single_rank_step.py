#!/usr/bin/env python3
"""The data-parallel training step on RCCL with the ONE GPU a builder's box has: a process group of one rank over the "nccl"
backend (= RCCL on ROCm), E3D_DDP_SINGLE_RANK=1 so that GradientAverager / GraphedDDPStep take their multi-rank path -- flat
gradient buckets, post-accumulate hooks, asynchronous ``all_reduce`` calls on RCCL's stream, the two graph segments around
them -- and the same steps without a process group (training.GraphedStep) next to it: a group of one averages nothing, so
losses and weights must agree.  Prints one JSON object.  (Scaling is the driver's to measure on an 8-GPU node.)

    python tools/lab/rccl_single_rank_step.py [layers] [steps]"""
import json
import os
import sys

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29531")
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
os.environ["E3D_DDP_SINGLE_RANK"] = "1"

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import bench_train  # noqa: E402

layers = int(sys.argv[1]) if len(sys.argv) > 1 else 6
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5


def run(ddp, graph):
    torch.manual_seed(0)
    return bench_train.run("sequence", batch=64, seq_len=128, steps=steps, warmup=4, ddp=ddp, layers=layers, seed=0, graph=graph)


out = {"what": "sequence-model training step (config 4's per-rank batch) on a one-rank RCCL process group vs no process group"}
single = run(False, True)
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
probe = torch.arange(8, device="cuda:0", dtype=torch.float32)
dist.all_reduce(probe)
torch.cuda.synchronize()
out["backend"] = dist.get_backend()
out["all_reduce_probe_ok"] = bool(torch.equal(probe.cpu(), torch.arange(8, dtype=torch.float32)))
ddp_graph = run(True, True)
ddp_eager = run(True, False)
dist.barrier()
dist.destroy_process_group()
for name, r in (("single_process_graph", single), ("rccl_one_rank_graph_segments", ddp_graph), ("rccl_one_rank_eager", ddp_eager)):
    out[name] = {k: r[k] for k in ("ms_per_step", "host_enqueue_ms", "graph_replay", "loss", "ranks", "backend", "buckets",
                                   "gradient_MB_per_step") if k in r}
# same seeds, same batches, a group of one: the three step sequences see the same numbers up to the summation order of the
# flat-bucket path
out["loss_agreement"] = {"graph_segments_vs_single": abs(ddp_graph["loss"] - single["loss"]),
                         "eager_vs_single": abs(ddp_eager["loss"] - single["loss"])}
out["ok"] = bool(out["all_reduce_probe_ok"] and out["backend"] == "nccl" and ddp_graph.get("buckets", 0) > 0
                 and out["loss_agreement"]["graph_segments_vs_single"] < 2e-3 * abs(single["loss"])
                 and out["loss_agreement"]["eager_vs_single"] < 2e-3 * abs(single["loss"]))
print(json.dumps(out))
sys.exit(0 if out["ok"] else 1)
