#!/usr/bin/env python3
"""Lab: sequence reverse chain (6-layer model, T = 50) with and without graph replay of the step, by batch size."""
import os, sys, time
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.sequence_model import sample as S  # noqa: E402
from e3diff_amd.sequence_model.model import PeptideDiff  # noqa: E402
from e3diff_amd.sequence_model.utils import BlosumTransition, PredefinedNoiseScheduleDiscrete  # noqa: E402
import contextlib, io
DEV, L, T = "cuda:0", 64, 50
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=6, max_position_embeddings=L,
         hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
torch.manual_seed(0)
model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                    loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=T).eval().to(DEV)
sched, trans = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV), BlosumTransition(x_classes=20)
for B in (1, 4, 8, 16, 32):
    pk = synthetic_pockets(B, L, seed=1, with_ligand_seq=True)
    row = f"B={B:3d} x L={L}:"
    for g in (False, True):
        ts = []
        for rep in range(3):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            with contextlib.redirect_stdout(io.StringIO()):
                S.denoise(pk, model, sched, trans, True, timesteps=T, use_graph=g)
            torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        row += f"  graph={g}: {min(ts) * 1e3:7.1f} ms per chain ({min(ts) / T * 1e3:.2f} ms per step)"
    print(row, flush=True)
