#!/usr/bin/env python3
"""BASELINE configs[2] END TO END, once: structure sampling, 256 x 256-residue synthetic pockets, the whole T = 1000 chain
through the product entry point (structure_model/sample.py::p_sample_loop: pocket encoder once, then 1000 decoder steps +
DDPM update + wrap, trajectory [1000, 256, 256, 8] = 2 GB kept on the device and copied to the host once) -- the tests
check this size through a few steps and size-independent properties; this is the full run: wall time, finiteness, range.

    python tools/lab/config3_full_chain.py [T] [B]"""
import json, os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
from helpers import synthetic_pockets  # noqa: E402
T = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
L = 256
dev = torch.device("cuda:0")
model, pkg = bench.build_model(L, dev)
from e3diff_amd.structure_model import sample as S
from e3diff_amd.structure_model.utils import cosine_beta_schedule, modulo_with_wrapped_range
pk = {k: v.to(dev) for k, v in synthetic_pockets(B, L, seed=1000).items() if torch.is_tensor(v)}
torch.manual_seed(0)
x_T = modulo_with_wrapped_range(torch.randn(B, L, 8, device=dev)).contiguous()
betas = cosine_beta_schedule(T)
out = {"workload": f"p_sample_loop, {B} x {L}-residue pockets, T = {T}, f16x3, padded frames (trim_padding=False), encoder cached per chain"}
with torch.no_grad():
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    traj = S.p_sample_loop(model, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"], pk["receptor_angles"], T, betas,
                           disable_pbar=True, return_device=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    host = traj.cpu()
    t2 = time.perf_counter()
out.update(chain_s=t1 - t0, ms_per_step=1e3 * (t1 - t0) / T, pocket_steps_per_s=B * T / (t1 - t0), trajectory_shape=list(traj.shape),
           trajectory_GiB=traj.numel() * 4 / 2 ** 30, copy_out_s=t2 - t1,
           all_finite=bool(torch.isfinite(host).all()), max_abs=float(host.abs().max()), within_pi=bool(float(host.abs().max()) <= 3.1415927 + 1e-6),
           last_step_std=float(host[-1][pk["ligand_attn_mask"].cpu().bool()].std()), peak_mem_GiB=torch.cuda.max_memory_allocated() / 2 ** 30)
print(json.dumps(out))
sys.exit(0 if out["all_finite"] and out["within_pi"] else 1)
