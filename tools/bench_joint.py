#!/usr/bin/env python3
"""BASELINE config 5 (configs[4]), one GPU's share: joint structure -> sequence sampling of 128 synthetic pockets
(L = 128): stage 1 = structure p_sample_loop over T = 1000 steps, stage 2 = sequence denoise over 50 steps
(DiscreteUniformTransition, diverse=True) on the generated last-step angles, handed over on the device.

    python tools/bench_joint.py [--batch 128] [--seq-len 128] [--t-structure 1000] [--t-sequence 50]
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


def run(batch=128, seq_len=128, t_structure=1000, t_sequence=50, trim=False, device=DEV):
    """One GPU's share of BASELINE config 5 (sequence_model/sample_by_generated_angles.py:196-278): structure chain ->
    hand-over on the device -> sequence chain.  Returns a dict (also the ``joint`` key of bench.py's line)."""
    B, L = batch, seq_len
    from e3diff_amd.structure_model import sample as SS
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as SM
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    from e3diff_amd.sequence_model import sample_by_generated_angles as QJ
    from e3diff_amd.sequence_model.model import PeptideDiff
    from e3diff_amd.sequence_model.utils import DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete

    def cfgs(layers):
        c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers,
                 max_position_embeddings=L, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
        return BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True)

    torch.manual_seed(0)
    smodel = SM(*cfgs(12), feature_names=list("abcdefgh"), loss_func=[SM.diheral_loss_func] * 8).eval().to(device)
    qmodel = PeptideDiff(*cfgs(6), feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                         noise_schedule="cosine", timesteps=t_sequence).eval().to(device)
    pk = synthetic_pockets(B, L, seed=0, with_ligand_seq=True)
    dpk = {k: v.to(device) for k, v in pk.items() if torch.is_tensor(v)}
    tab = CosineTables(t_structure)
    x_T = modulo_with_wrapped_range(torch.randn(B, L, 8, device=device))

    def sync():
        torch.cuda.synchronize()
        return time.perf_counter()

    # warm-up (first launches, allocator)
    SS.p_sample_loop(smodel, dpk["ligand_attn_mask"], x_T, dpk["receptor_seq"], dpk["receptor_attn_mask"],
                     dpk["receptor_angles"], 4, CosineTables(4), disable_pbar=True, return_device=True, step=1, trim_padding=trim)
    t0 = sync()
    traj = SS.p_sample_loop(smodel, dpk["ligand_attn_mask"], x_T, dpk["receptor_seq"], dpk["receptor_attn_mask"],
                            dpk["receptor_angles"], t_structure, tab, disable_pbar=True, return_device=True, step=1,
                            trim_padding=trim)
    t1 = sync()
    angles = QJ.angles_from_trajectory(traj, dpk["ligand_attn_mask"])
    schedule = PredefinedNoiseScheduleDiscrete("cosine", t_sequence).to(device)
    ids, true_s, pred_s, rec = QJ.denoise(pk, angles, qmodel, schedule, DiscreteUniformTransition(20), True, trim_padding=trim,
                                          timesteps=t_sequence)
    t2 = sync()
    assert len(pred_s) == B and bool(torch.isfinite(traj[-1]).all())
    mib = traj.numel() * 4 / 2 ** 20
    del traj, smodel, qmodel
    torch.cuda.empty_cache()
    return {"pockets": B, "seq_len": L, "frames": "trimmed to the longest ligand / pocket" if trim else "padded to seq_len",
            "structure_steps": t_structure, "structure_s": t1 - t0, "structure_pocket_steps_per_s": B * t_structure / (t1 - t0),
            "sequence_steps": t_sequence, "sequence_s": t2 - t1, "sequence_pocket_steps_per_s": B * t_sequence / (t2 - t1),
            "total_s": t2 - t0, "pockets_per_s": B / (t2 - t0), "trajectory_MiB_on_device": mib}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=128)
    ap.add_argument("--seq-len", type=int, default=128)
    ap.add_argument("--t-structure", type=int, default=1000)
    ap.add_argument("--t-sequence", type=int, default=50)
    ap.add_argument("--trim", action="store_true", help="trim_padding=True in both chains (frame of the longest ligand / pocket)")
    a = ap.parse_args()
    r = run(a.batch, a.seq_len, a.t_structure, a.t_sequence, a.trim)
    print(f"joint sampling ({r['frames']}), {r['pockets']} pockets x L={r['seq_len']} on one GPU: structure {r['structure_steps']} steps "
          f"{r['structure_s']:.2f} s ({r['structure_pocket_steps_per_s']:.0f} pocket-steps/s, encoder cached), sequence "
          f"{r['sequence_steps']} steps {r['sequence_s']:.2f} s ({r['sequence_pocket_steps_per_s']:.0f} pocket-steps/s); total "
          f"{r['total_s']:.2f} s = {r['pockets_per_s']:.1f} pockets/s; trajectory kept on device ({r['trajectory_MiB_on_device']:.0f} MiB)",
          flush=True)


if __name__ == "__main__":
    main()
