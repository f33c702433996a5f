#!/usr/bin/env python3
"""Lab: which host ops issue the small device-to-device copies / fills of one headline sampling step (B=256, L=256)?
torch.profiler over two steps, memcpy / memset / fill / copy kernels grouped by their aten op and first Python frames."""
import os, sys, collections
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import synthetic_pockets  # noqa: E402
B, L = int(os.environ.get("B", "256")), 256
dev = torch.device("cuda:0")
model, pkg = bench.build_model(L, dev)
from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
from e3diff_amd.structure_model import sample as S
pk = {k: v.to(dev) for k, v in synthetic_pockets(B, L, seed=1000).items() if torch.is_tensor(v)}
tab = CosineTables(1000)
x = modulo_with_wrapped_range(torch.randn(B, L, 8, device=dev)).contiguous()
nxt = torch.empty_like(x)
pkg.hip.lib().e3d_attn_skip_padded_tiles(0)


def step(i, x, out):
    return S._reverse_step(model, pk["ligand_attn_mask"], x, pk["receptor_seq"], pk["receptor_attn_mask"], pk["receptor_angles"], i, tab, None,
                           None, out, wrap=True)


with torch.no_grad():
    for i in range(2):
        x, nxt = step(999 - i, x, nxt), x
    torch.cuda.synchronize()
    from torch.profiler import profile, ProfilerActivity
    with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
        for i in range(2):
            x, nxt = step(997 - i, x, nxt), x
        torch.cuda.synchronize()
ev = prof.events()
cnt = collections.Counter()
for e in ev:
    n = e.name
    if any(k in n for k in ("copy_", "aten::fill_", "aten::zero_", "aten::cat", "aten::contiguous", "aten::clone", "aten::to", "aten::_to_copy", "aten::zeros", "aten::full")) and e.device_type == torch.autograd.DeviceType.CPU:
        st = [s for s in (e.stack or []) if "e3-invaraint" in s or "bench" in s or "tests/" in s][:2]
        cnt[(n, tuple(st))] += 1
for (n, st), c in cnt.most_common(25):
    print(f"{c / 2:6.1f} per step  {n:22s} {' <- '.join(s.split('/')[-1] for s in st)}")
