"""VERDICT r02 item 1, GPU leg: where do the split arithmetics leave the fp64 oracle in the weights-x4 regime?
(1) whole forward per (GEMM mode x attention mode); (2) teacher-forced per stage: every stage of the product is fed the
fp64 oracle's input of that stage and compared with the oracle's output of that stage."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge
from helpers import seeded_state_dict, synthetic_pockets, rescaled_state_dict, rel_err
from oracle import structure as ostr, bert as obert

pkg = ge.load_package()
from e3diff_amd import ops, bert
from e3diff_amd.blocks import flat2d
from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
from e3diff_amd.bert import BertConfig
DEV = "cuda:0"
SCALE = float(os.environ.get("SCALE", "4"))
L, B = 256, 2
common = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=12, max_position_embeddings=L)
model = ConditionalBertForDiffusionBase(BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
shapes = {k: v.shape for k, v in model.state_dict().items()}
sd = seeded_state_dict(shapes, seed=71)
if SCALE != 1.0:
    sd = rescaled_state_dict(sd, SCALE, (0.5, 2.0), seed=72)
model.load_state_dict(sd)
model = model.eval().to(DEV)
pk = synthetic_pockets(B, L, seed=73, lig_range=(180, 256), rec_range=(150, 256))
d = {k: v.to(DEV) for k, v in pk.items() if torch.is_tensor(v)}
x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=torch.Generator().manual_seed(74)))
t = torch.tensor([999, 3])
torch.set_num_threads(16)

# ---- fp64 oracle with every stage boundary recorded
sd64 = {k: v.double() for k, v in sd.items()}
nh, mp = 12, L
D = lambda v: v.double()
stages = {}   # name -> (input, output)
lig_bias, rec_bias = ostr.extend_mask(D(pk["ligand_attn_mask"])), ostr.extend_mask(D(pk["receptor_attn_mask"]))
rec_a = ostr.embeddings(sd64, "receptor_angle_emb", D(pk["receptor_angles"]))
rec_s = ostr.embeddings(sd64, "receptor_seq_emb", D(pk["receptor_seq"]))
x = ostr.se_layer(sd64, "receptor_emb", rec_a, rec_s, rec_bias, nh, mp)
stages["receptor_emb"] = ((rec_a, rec_s), x)
for i in range(12):
    y = obert.bert_layer(sd64, f"encoder.layer.{i}", x, rec_bias, nh, mp)
    stages[f"enc{i}"] = (x, y); x = y
enc = x
lig = ostr.embeddings(sd64, "ligand_angle_emb", D(x_t))
temb = ostr.fourier_projection(sd, "timestep_projector", t).double().unsqueeze(1)
x = ostr.se_layer(sd64, "timestep_emb", lig, temb, lig_bias, nh, mp)
stages["timestep_emb"] = ((lig, temb), x)
for i in range(12):
    y = obert.bert_layer(sd64, f"decoder.layer.{i}", x, lig_bias, nh, mp, enc, rec_bias)
    stages[f"dec{i}"] = (x, y); x = y
want64 = ostr.predictor(sd64, "angles_predictor", x)
stages["predictor"] = (x, want64)
m = pk["ligand_attn_mask"].bool()
mr = pk["receptor_attn_mask"].bool()

def err(got, want, mask):
    got = got.detach().cpu().view(B, L, -1)[mask]; want = want.view(B, L, -1)[mask]
    return ((got.double() - want).abs().max() / want.abs().max()).item()

MODES = tuple(os.environ.get("MODES", "f32,bf16x6,f16x3,bf16x3").split(","))
if os.environ.get("NOSKIP") == "1":
    print("padded-tile skip OFF, previous =", pkg.hip.lib().e3d_attn_skip_padded_tiles(0))
xq = obert.linear(sd64, "encoder.layer.0.attention.self.query", stages["enc0"][0]); xk = obert.linear(sd64, "encoder.layer.0.attention.self.key", stages["enc0"][0])
print("enc0 input absmax", stages["enc0"][0].abs().max().item(), "q absmax", xq.abs().max().item(), "k absmax", xk.abs().max().item(), "enc1 q absmax", obert.linear(sd64, "encoder.layer.1.attention.self.query", stages["enc1"][0]).abs().max().item(), flush=True)
print(f"=== whole forward vs fp64 oracle, weights x{SCALE}", flush=True)
for gm in MODES:
    for am in MODES:
        ops.set_gemm_mode(gm); ops.set_attn_mode(am)
        with torch.no_grad():
            got = model(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
        print(f"gemm {gm:7s} attn {am:7s}: max-norm {err(got, want64, m):.3e}", flush=True)

if os.environ.get("CPU32", "1") == "1":
    # the CPU oracle in fp32, stage by stage on the same teacher-forced inputs: the yardstick of "fp32 grade" per stage
    F = lambda v: v.float()
    row = []
    with torch.no_grad():
        (a, s_), y = stages["receptor_emb"]
        row.append(("receptor_emb", err(ostr.se_layer(sd, "receptor_emb", F(a), F(s_), F(rec_bias), nh, mp), y, mr)))
        for i in range(12):
            xi, y = stages[f"enc{i}"]
            row.append((f"enc{i}", err(obert.bert_layer(sd, f"encoder.layer.{i}", F(xi), F(rec_bias), nh, mp), y, mr)))
        (a, c), y = stages["timestep_emb"]
        row.append(("timestep_emb", err(ostr.se_layer(sd, "timestep_emb", F(a), F(c), F(lig_bias), nh, mp), y, m)))
        for i in range(12):
            xi, y = stages[f"dec{i}"]
            row.append((f"dec{i}", err(obert.bert_layer(sd, f"decoder.layer.{i}", F(xi), F(lig_bias), nh, mp, F(enc), F(rec_bias)), y, m)))
        xi, y = stages["predictor"]
        row.append(("predictor", err(ostr.predictor(sd, "angles_predictor", F(xi)), y, m)))
        whole = ostr.forward(sd, {"num_heads": nh, "max_pos": mp}, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
    print("cpu32 whole forward: max-norm %.3e" % err(whole, want64, m), flush=True)
    print("cpu32", " ".join(f"{n}:{e:.1e}" for n, e in row), flush=True)
print("=== teacher-forced per stage (input = fp64 oracle's, cast to fp32)", flush=True)
G = lambda v: v.float().to(DEV)
lm, rm = d["ligand_attn_mask"].contiguous(), d["receptor_attn_mask"].contiguous()
for mode in MODES:
    ops.set_gemm_mode(mode); ops.set_attn_mode(mode)
    row = []
    with torch.no_grad():
        (a, s), y = stages["receptor_emb"]
        got = model.receptor_emb.run(flat2d(G(a)), flat2d(G(s)), rm, B, L)
        row.append(("receptor_emb", err(got, y, mr)))
        for i in range(12):
            xi, y = stages[f"enc{i}"]
            got = bert.run_layer(model.encoder.layer[i], flat2d(G(xi)), rm, B, L)
            row.append((f"enc{i}", err(got, y, mr)))
        (a, c), y = stages["timestep_emb"]
        got = model.timestep_emb.run(flat2d(G(a)), G(c.squeeze(1)).contiguous(), lm, B, L)
        row.append(("timestep_emb", err(got, y, m)))
        encg = flat2d(G(enc))
        for i in range(12):
            xi, y = stages[f"dec{i}"]
            layer = model.decoder.layer[i]
            kv = bert.project_cross_kv(layer.crossattention, encg)
            got = bert.run_layer(layer, flat2d(G(xi)), lm, B, L, kv, rm, L)
            row.append((f"dec{i}", err(got, y, m)))
        xi, y = stages["predictor"]
        got = model.angles_predictor.run(flat2d(G(xi)))
        row.append(("predictor", err(got, y, m)))
    print(mode, " ".join(f"{n}:{e:.1e}" for n, e in row), flush=True)
