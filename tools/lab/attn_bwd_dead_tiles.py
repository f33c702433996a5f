#!/usr/bin/env python3
"""How many 32-row query tiles of each fused attention backward see an all-zero dO, and how many key tiles are all padding,
in ONE eager training step on the synthetic BioLiP-shaped batch (the data attn_bwd_coop_kernel's dead-tile skip finds by itself).
    python tools/lab/attn_bwd_dead_tiles.py [structure|sequence]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from e3diff_amd import autograd  # noqa: E402

rows = []
orig = autograd._Attention.backward


def spy(ctx, dout):
    q_src, kv_src, dist_emb, key_mask, out, lse = ctx.saved_tensors
    B, nh, Lq, Lk, _ = ctx.dims
    d = dout.reshape(B, Lq, -1)
    live_rows = (d != 0).any(-1)                                  # [B, Lq]
    qt = live_rows.reshape(B, -1, 32).any(-1)                     # [B, q tiles]
    km = key_mask.reshape(B, Lk) if key_mask is not None else torch.ones(B, Lk, device=d.device)
    kt = (km != 0).reshape(B, -1, 32).any(-1)
    rows.append(dict(kind="self" if kv_src is None else "cross", relkey=dist_emb is not None, Lq=Lq, Lk=Lk,
                     live_q_rows=float(live_rows.float().mean()), live_q_tiles=float(qt.float().mean()),
                     live_k_tiles=float(kt.float().mean()), dout_absmax=float(d.abs().max()),
                     smallest_nonzero=float(d.abs()[d != 0].min()) if (d != 0).any() else 0.0))
    return orig(ctx, dout)


autograd._Attention.backward = staticmethod(spy)
import bench_train  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "structure"
os.environ["E3D_TRAIN_GRAPH"] = "0"
r = bench_train.run(name, steps=1, warmup=1, graph=False)
n = len(rows)
print(json.dumps(dict(model=name, launches=n, first=rows[:3], last=rows[-3:],
                      mean_live_q_tiles=sum(x["live_q_tiles"] for x in rows) / n,
                      mean_live_k_tiles=sum(x["live_k_tiles"] for x in rows) / n)))
