import os, sys, collections, torch
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
os.environ["E3D_TRAIN_GRAPH"] = "0"
import bench_train
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU]) as prof:
    bench_train.run(sys.argv[1] if len(sys.argv) > 1 else "structure", steps=1, warmup=1, graph=False)
cnt = collections.Counter()
for e in prof.events():
    if e.name in ("aten::fill_", "aten::zero_"):
        chain = []
        p = e.cpu_parent
        while p is not None and len(chain) < 4:
            chain.append(p.name)
            p = p.cpu_parent
        cnt[" <- ".join(chain)] += 1
tot = sum(cnt.values())
print("fill_/zero_ events (3 steps incl. warm-up):", tot)
for k, v in cnt.most_common(25):
    print(f"{v:5d}  {k[:200]}")
