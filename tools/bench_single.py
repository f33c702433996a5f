#!/usr/bin/env python3
"""BASELINE configs[0] on the GPU: ONE synthetic 64-residue pocket, 50 reverse steps of the full 12+12-layer structure
model (encoder cached, as `structure_model/sample.py` runs it), default arithmetic.  Prints ms per reverse step.

    python tools/bench_single.py [--seq-len 64] [--steps 50] [--graph]
"""
import argparse
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402

pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402

DEV = "cuda:0"


def run(seq_len=64, batch=1, steps=50, graph=None, chains=3):
    """Best of ``chains`` full reverse chains; returns a dict (also the ``single_pocket`` key of bench.py's line)."""
    from e3diff_amd.structure_model import sample as S
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    L, B, T = seq_len, batch, steps
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12, max_position_embeddings=L)
    torch.manual_seed(0)
    model = ConditionalBertForDiffusionBase(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), 8).eval().to(DEV)
    pk = {k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=1).items() if torch.is_tensor(v)}
    x_T = modulo_with_wrapped_range(torch.randn(B, L, 8, device=DEV))
    tab = CosineTables(T)

    def chain():
        return S.p_sample_loop(model, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"],
                               pk["receptor_angles"], T, tab, disable_pbar=True, return_device=True, step=1,
                               use_graph=None if graph is None else bool(graph))

    with pkg.ops.arithmetic(S.ARITHMETIC):
        chain()
        torch.cuda.synchronize()
        best = None
        for _ in range(chains):
            t0 = time.perf_counter()
            out = chain()
            torch.cuda.synchronize()
            dt = time.perf_counter() - t0
            best = dt if best is None else min(best, dt)
        mode = pkg.ops.GEMM_MODE
    assert bool(torch.isfinite(out).all())
    return {"batch": B, "seq_len": L, "timesteps": T, "arithmetic": mode, "graph_replay": "sampler default" if graph is None else bool(graph),
            "ms_per_chain": best * 1e3, "ms_per_step": best / T * 1e3,
            "note": "structure_model/sample.py p_sample_loop: encoder + cross K/V once per chain, 12-layer decoder + DDPM update per step"}


def run_sequence(seq_len=64, batch=1, steps=50, graph=None, chains=3):
    """The sequence stage for ONE pocket: 50 reverse steps of the 6-layer amino-acid denoiser (sequence_model/sample.py
    ``denoise``, BLOSUM transition, categorical draws); best of ``chains``."""
    import contextlib
    import io
    from e3diff_amd.sequence_model import sample as Q
    from e3diff_amd.sequence_model.model import PeptideDiff
    from e3diff_amd.sequence_model.utils import BlosumTransition, PredefinedNoiseScheduleDiscrete
    L, B, T = seq_len, batch, steps
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=6, max_position_embeddings=L)
    torch.manual_seed(0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                        loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=T).eval().to(DEV)
    sched, trans = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV), BlosumTransition(x_classes=20)
    pk = synthetic_pockets(B, L, seed=1, with_ligand_seq=True)
    best = None
    for _ in range(chains + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        with contextlib.redirect_stdout(io.StringIO()):
            Q.denoise(pk, model, sched, trans, True, timesteps=T, use_graph=None if graph is None else bool(graph))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"batch": B, "seq_len": L, "timesteps": T, "arithmetic": pkg.ops.GEMM_MODE,
            "graph_replay": "sampler default" if graph is None else bool(graph), "ms_per_chain": best * 1e3, "ms_per_step": best / T * 1e3,
            "note": "sequence_model/sample.py denoise: 6-layer decoder + discrete posterior draw per step, host to sequences included"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seq-len", type=int, default=64)
    ap.add_argument("--batch", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--graph", type=int, default=None, choices=(0, 1), help="1: replay one captured HIP graph per step, 0: eager launches (default: the sampler's own choice)")
    a = ap.parse_args()
    r = run(a.seq_len, a.batch, a.steps, a.graph)
    print(f"single-pocket sampling B={r['batch']} L={r['seq_len']} T={r['timesteps']} ({r['arithmetic']}, skinny GEMM M<={pkg.ops.SKINNY_MAX_M}, "
          f"graph={a.graph}): {r['ms_per_chain']:.1f} ms per chain = {r['ms_per_step']:.3f} ms per reverse step (encoder cached)", flush=True)


if __name__ == "__main__":
    main()
