"""VERDICT r02 item 1, CPU leg: conditioning of the margin-test regimes.  The oracle in fp64 (Fourier features kept in
fp32 -- their argument rounding is part of the reference arithmetic) against the oracle in fp32, on the state dicts of
tests/test_full_configs_gpu.py::test_bf16x3_margin_elementwise_at_L256_full_depth.  Also dumps per-layer statistics
of the attention logits (sharpness) so the regimes can be described."""
import os, sys, math
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from helpers import FULL_STRUCT, seeded_state_dict, synthetic_pockets, rescaled_state_dict, rel_err, elementwise_err
from oracle import structure as ostr

pkg = ge.load_package()
from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
from e3diff_amd.bert import BertConfig
L, B = 256, 2
common = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12, max_position_embeddings=L)
model = ConditionalBertForDiffusionBase(BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
shapes = {k: v.shape for k, v in model.state_dict().items()}
torch.set_num_threads(8)
for scale in (1.0, 2.0, 4.0):
    sd = seeded_state_dict(shapes, seed=71)
    if scale != 1.0:
        sd = rescaled_state_dict(sd, scale, (0.5, 2.0), seed=72)
    pk = synthetic_pockets(B, L, seed=73, lig_range=(180, 256), rec_range=(150, 256))
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=torch.Generator().manual_seed(74)))
    t = torch.tensor([999, 3])
    cfg = {"num_heads": 12, "max_pos": L}
    want32 = ostr.forward(sd, cfg, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
    sd64 = {k: v.double() for k, v in sd.items()}
    orig = ostr.fourier_projection
    ostr.fourier_projection = lambda s, p, tt: orig(sd, p, tt).double()
    want64 = ostr.forward(sd64, cfg, t, x_t.double(), pk["ligand_attn_mask"].double(), pk["receptor_seq"].double(),
                          pk["receptor_angles"].double(), pk["receptor_attn_mask"].double())
    ostr.fourier_projection = orig
    m = pk["ligand_attn_mask"].bool()
    print(f"scale x{scale}: fp32-CPU vs fp64-CPU max-norm {rel_err(want32[m], want64[m].float()):.3e}  "
          f"elementwise p99.9/max {elementwise_err(want32[m], want64[m])}", flush=True)
