#!/usr/bin/env python3
"""Lab: is the SECOND GraphedStep of a process as good as the first?  The structure twin-model comparison of
tests/test_training_gpu.py (same batches, eager vs graph-replayed) with the graphed run done twice, before and after the eager one."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
import test_training_gpu as T  # noqa: E402
from e3diff_amd import autograd, ops, training  # noqa: E402

batches = [b for i, b in enumerate(T._structure_batches(10)) if i != 4]
res = []
for graphed in (True, False, True, True):
    model = T._small_structure_model()
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = training.GraphedStep(model, optim, params, 1.0) if graphed else None
    losses = []
    with ops.arithmetic("bf16x3"):
        for k, batch in enumerate(batches):
            if graphed:
                losses.append(float(stepper.step(batch)))
            else:
                loss = model.training_step(batch)
                optim.zero_grad(set_to_none=True)
                with autograd.deferred_weight_grads():
                    loss.backward()
                training.clip_and_step(params, optim, 1.0)
                losses.append(float(loss))
    res.append((graphed, losses, [p.detach().clone() for p in params]))
    del model, optim, stepper
ref = res[1]
for i, (g, losses, ps) in enumerate(res):
    apart = sum(float((a - b).abs().sum()) for a, b in zip(ps, ref[2]))
    print(f"run {i} graphed={g}: last losses {['%.6f' % v for v in losses[-3:]]}  sum|param - eager param| = {apart:.4e}", flush=True)
