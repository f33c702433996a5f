"""GPU parity of the sequence model's denoising path (HIP via the C-ABI) against the
reference-generated golden fixtures and the CPU oracle.  Logits: 1e-4 relative fp32; the BLOSUM
table index: bit-exact; sampled classes: identical wherever the oracle's decision margin exceeds
the fp tolerance."""
import os

import pytest
import torch
import torch.nn.functional as F

from helpers import FULL_SEQ, GOLDEN, rel_err, seeded_state_dict, synthetic_pockets
from oracle import sequence as oseq

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def build(pkg, cfg, L, seed, relkey=True, wrapper=False):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import ConditionalBertForDiffusionBase, PeptideDiff
    common = dict(hidden_size=cfg["hidden_size"], num_attention_heads=cfg["num_heads"],
                  intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                  max_position_embeddings=L, position_embedding_type="relative_key" if relkey else "absolute")
    enc, dec = BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True)
    if wrapper:
        model = PeptideDiff(enc, dec, feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                            loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=50)
    else:
        model = ConditionalBertForDiffusionBase(enc, dec, 20)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=seed)
    model.load_state_dict(sd, strict=True)
    return model.eval().to(DEV), sd


def to_dev(d):
    return {k: v.to(DEV) for k, v in d.items() if torch.is_tensor(v)}


def blosum():
    return torch.load(os.path.join(GOLDEN, "blosum_substitute.pt"), weights_only=True)


def test_forward_matches_reference_golden(pkg, hip):
    fx = torch.load(os.path.join(GOLDEN, "sequence_forward_tiny.pt"), weights_only=False)
    cfg = dict(fx["cfg"])
    model, sd = build(pkg, cfg, cfg["max_seq_len"], fx["seed"], relkey=False, wrapper=True)
    assert {k: tuple(v.shape) for k, v in sd.items()} == fx["shapes"]      # checkpoint keys == the reference's
    pk = to_dev(fx["pockets"])
    for tag, (t, want) in fx["outs"].items():
        got = model(t.to(DEV), fx["x_t"].to(DEV), pk["ligand_angles"], pk["ligand_attn_mask"], pk["receptor_seq"],
                    pk["receptor_angles"], pk["receptor_attn_mask"])
        assert rel_err(got, want) < TOL, tag
    # losses of the reference on its own noised sample (A25)
    loss = model.get_loss(pk, fx["loss_t_norm"].to(DEV), fx["aa_noised_mixed"].to(DEV))
    for g, w in zip(loss, fx["loss"]):
        assert g.item() == pytest.approx(w.item(), rel=2e-4, abs=1e-5)
    # forward noising argmax path == the reference's (A21), padding rows -> class 0
    from e3diff_amd.sequence_model.model import onehot_to_index
    ab = model.discrete_noise_schedule.get_alpha_bar(t_normalized=fx["aa_t_int"].to(DEV) / 50)
    qtb = model.aa_transition_model.get_Qt_bar(ab, device=DEV).contiguous()
    idx = pkg.ops.discrete_q_sample(onehot_to_index(pk["ligand_seq"]).contiguous(), qtb, None)
    assert torch.equal(F.one_hot(idx.long(), 20).float().cpu(), fx["aa_noised_argmax"])


@pytest.mark.parametrize("layers,B,L", [(2, 3, 64), (6, 2, 128)])
def test_forward_matches_oracle_full_width_with_relkey(pkg, hip, layers, B, L):
    cfg = dict(FULL_SEQ, num_hidden_layers=layers)
    model, sd = build(pkg, cfg, L, seed=30 + layers)
    ocfg = {"num_heads": 12, "max_pos": L}
    pk = synthetic_pockets(B, L, seed=L + 1, with_ligand_seq=True)
    x_t = F.one_hot(torch.randint(0, 20, (B, L), generator=torch.Generator().manual_seed(1)), 20).float()
    d = to_dev(pk)
    for t in (torch.full((B, 1), 17.0), torch.rand(B, 1, generator=torch.Generator().manual_seed(2))):
        want = oseq.forward(sd, ocfg, t, x_t, pk["ligand_angles"], pk["ligand_attn_mask"], pk["receptor_seq"],
                            pk["receptor_angles"], pk["receptor_attn_mask"])
        got = model(t.to(DEV), x_t.to(DEV), d["ligand_angles"], d["ligand_attn_mask"], d["receptor_seq"],
                    d["receptor_angles"], d["receptor_attn_mask"])
        assert got.shape == (B, L, 20)
        assert rel_err(got, want) < TOL


def test_transitions_on_device_bit_exact_index(pkg, hip):
    from e3diff_amd.sequence_model.utils import (BlosumTransition, DiscreteUniformTransition,
                                                PredefinedNoiseScheduleDiscrete)
    fx = torch.load(os.path.join(GOLDEN, "sequence_utils.pt"), weights_only=False)
    sched = PredefinedNoiseScheduleDiscrete("cosine", 50).to(DEV)
    t_norm = (torch.arange(51).float() / 50).unsqueeze(1).to(DEV)
    ab = sched.get_alpha_bar(t_normalized=t_norm)
    # the alpha-bar table is exp(cumsum(log(.))) on the HOST: libm/SIMD width may differ in the last
    # bit between the fixture's CPU and this box's, the integer index below may not
    assert torch.allclose(ab.cpu(), fx["alpha_bar_of_t"], rtol=2e-6, atol=0)
    bl = BlosumTransition(x_classes=20)
    assert torch.equal(bl.table_index(ab).cpu(), fx["blosum_t_index"])          # INT index: bit-exact
    assert torch.equal(bl.table_index(fx["round_probe_in"].to(DEV)).cpu(), fx["round_probe_idx"])
    assert rel_err(bl.get_Qt_bar(ab, DEV), fx["blosum_Qtb"]) < 1e-6
    assert rel_err(DiscreteUniformTransition(20).get_Qt_bar(ab, DEV), fx["uniform_Qtb"]) < 1e-6


def test_reverse_sampler_matches_reference_golden(pkg, hip):
    from e3diff_amd.sequence_model.sample import sample_p_zs_given_zt_discrete
    from e3diff_amd.sequence_model.utils import (BlosumTransition, DiscreteUniformTransition,
                                                PredefinedNoiseScheduleDiscrete)
    fx = torch.load(os.path.join(GOLDEN, "sequence_sampler.pt"), weights_only=False)
    T = fx["T"]
    sched = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV)
    trans = {"blosum": BlosumTransition(x_classes=20), "uniform": DiscreteUniformTransition(20)}
    x_t, logits = fx["x_t"].to(DEV), fx["logits"].to(DEV)
    for (name, s_int), case in fx["cases"].items():
        s = (s_int * torch.ones(2, 1) / T).to(DEV)
        t = ((s_int + 1) * torch.ones(2, 1) / T).to(DEV)
        x_s = sample_p_zs_given_zt_discrete(t, s, x_t, logits, sched, trans[name], False, False).cpu()
        prob = case["prob_X"]
        top2 = prob.topk(2, -1).values
        clear = ((top2[:, 0] - top2[:, 1]) > 1e-5).reshape(2, -1)
        assert torch.equal(x_s[clear], case["argmax_onehot"][clear]), (name, s_int)
        assert clear.float().mean() > 0.9
    assert sample_p_zs_given_zt_discrete(None, None, x_t, logits, sched, trans["blosum"], True, True) is logits


def test_denoise_loop_matches_oracle_teacher_forced(pkg, hip):
    """Whole reverse chain, argmax and injected-uniform paths.  The chain is discrete: one flipped
    class changes every later step, so each step is restarted from the oracle's state."""
    from e3diff_amd.sequence_model.sample import sample_p_zs_given_zt_discrete
    from e3diff_amd.sequence_model.utils import BlosumTransition, PredefinedNoiseScheduleDiscrete
    cfg = dict(FULL_SEQ, num_hidden_layers=2)
    B, L, T = 2, 64, 6
    model, sd = build(pkg, cfg, L, seed=77)
    ocfg = {"num_heads": 12, "max_pos": L}
    pk = synthetic_pockets(B, L, seed=5, with_ligand_seq=True)
    d = to_dev(pk)
    gen = torch.Generator().manual_seed(9)
    x = F.one_hot(torch.randint(0, 20, (B, L), generator=gen), 20).float()
    us = torch.rand(T, B, L, generator=gen)
    osched, otrans = oseq.NoiseScheduleDiscrete(T), oseq.BlosumTransition(blosum())
    sched, trans = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV), BlosumTransition(x_classes=20)
    agree = []
    for n, s_int in enumerate(reversed(range(T))):
        s_arr = s_int * torch.ones(B, 1)
        t_arr = s_arr + 1
        logits = oseq.forward(sd, ocfg, s_arr, x, pk["ligand_angles"], pk["ligand_attn_mask"], pk["receptor_seq"],
                              pk["receptor_angles"], pk["receptor_attn_mask"])
        got_logits = model(s_arr.to(DEV), x.to(DEV), d["ligand_angles"], d["ligand_attn_mask"], d["receptor_seq"],
                           d["receptor_angles"], d["receptor_attn_mask"])
        assert rel_err(got_logits, logits) < TOL
        want = oseq.sample_p_zs_given_zt_discrete(t_arr / T, s_arr / T, x, logits, osched, otrans, True,
                                                  s_int == 0, u=us[n])
        got = sample_p_zs_given_zt_discrete((t_arr / T).to(DEV), (s_arr / T).to(DEV), x.to(DEV), got_logits, sched,
                                            trans, True, s_int == 0, u=us[n].to(DEV)).cpu()
        if s_int == 0:
            assert rel_err(got, want) < TOL                # last step returns the logits
        else:
            agree.append((got.argmax(-1) == want.argmax(-1)).float().mean().item())
            assert bool((got.sum(-1) == 1).all())
        x = want
    assert min(agree) > 0.97, agree


def test_apply_aa_noise_matches_oracle(pkg, hip):
    cfg = dict(FULL_SEQ, num_hidden_layers=1)
    model, _ = build(pkg, cfg, 64, seed=3, wrapper=True)
    pk = synthetic_pockets(4, 64, seed=2, with_ligand_seq=True)
    t_int = torch.tensor([[0.0], [7.0], [33.0], [50.0]])
    u = torch.rand(4, 64, generator=torch.Generator().manual_seed(1))
    want = oseq.apply_aa_noise(pk["ligand_seq"], t_int, oseq.NoiseScheduleDiscrete(50), oseq.BlosumTransition(blosum()), u=u)
    got = model.apply_aa_noise(pk["ligand_seq"].to(DEV), t_int.to(DEV), u=u.to(DEV)).cpu()
    assert (got.argmax(-1) == want.argmax(-1)).float().mean() > 0.995
    pad = pk["ligand_attn_mask"] == 0
    assert bool((got.argmax(-1)[pad] == 0).all()) and bool((got.sum(-1) == 1).all())
    # statistical check of the device RNG path: empirical class frequencies at t=T follow Qtb's column
    seq = F.one_hot(torch.full((1, 20000), 3), 20).float().to(DEV)
    draws = model.apply_aa_noise(seq, torch.tensor([[50.0]], device=DEV)).argmax(-1).reshape(-1).cpu()
    prob = oseq.aa_noise_prob(seq.cpu()[:, :1], torch.tensor([[50.0]]), oseq.NoiseScheduleDiscrete(50),
                              oseq.BlosumTransition(blosum()))[0]
    freq = torch.bincount(draws, minlength=20).float() / draws.numel()
    assert (freq - prob / prob.sum()).abs().max() < 0.015


def test_joint_handover_helpers(pkg, hip):
    from e3diff_amd.sequence_model.sample_by_generated_angles import angles_from_trajectory
    traj = torch.randn(3, 2, 16, 8, device=DEV)
    mask = (torch.arange(16)[None] < torch.tensor([[5], [9]])).float().to(DEV)
    out = angles_from_trajectory(traj, mask)
    assert torch.equal(out[0, :5], traj[-1, 0, :5]) and float(out[0, 5:].abs().sum()) == 0.0


def test_seeded_sampling_chain_is_the_same_replayed_or_launched(pkg, hip):
    """a chain that DRAWS its categorical samples from torch's device generator (diverse=True, no injected
    uniforms) must not depend on whether its steps are replayed from a graph -- a size threshold decides that.  The graph's
    warm-up pass puts the generator back where it found it; equal seeds then give equal sequences on both paths (same kernels,
    same philox offsets per step), or at the very least the same per-position class statistics."""
    from e3diff_amd.sequence_model import sample as S
    from e3diff_amd.sequence_model.utils import BlosumTransition, PredefinedNoiseScheduleDiscrete
    cfg = dict(FULL_SEQ, num_hidden_layers=2)
    B, L, T = 4, 64, 10
    model, _ = build(pkg, cfg, L, seed=22)
    pk = synthetic_pockets(B, L, seed=7, with_ligand_seq=True)
    x_T = F.one_hot(torch.randint(0, 20, (B, L), generator=torch.Generator().manual_seed(5)), 20).float()
    sched, trans = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV), BlosumTransition(x_classes=20)
    outs = []
    for g_ in (True, False):
        torch.manual_seed(1234)
        outs.append(S.denoise(pk, model, sched, trans, True, x_T=x_T, timesteps=T, use_graph=g_))
    assert outs[0][2] == outs[1][2], "seeded chains differ between graph replay and eager launches"


@pytest.mark.parametrize("diverse", [False, True])
def test_denoise_graph_replay_is_bit_identical_to_eager_launches(pkg, hip, diverse):
    """sequence_model/sample.py::denoise with the reverse step replayed from a HIP graph (GraphedDenoiseStep: default for
    small chains) against kernel-by-kernel launches: the same sequences and recovery rates for the argmax chain and for the
    categorical chain with injected uniforms (a discrete chain: ONE different class would change every later step), and
    the graph really is what ran."""
    from e3diff_amd.sequence_model import sample as S
    from e3diff_amd.sequence_model.utils import BlosumTransition, PredefinedNoiseScheduleDiscrete
    cfg = dict(FULL_SEQ, num_hidden_layers=2)
    B, L, T = 4, 64, 12
    model, _ = build(pkg, cfg, L, seed=21)
    pk = synthetic_pockets(B, L, seed=6, with_ligand_seq=True)
    gen = torch.Generator().manual_seed(4)
    x_T = F.one_hot(torch.randint(0, 20, (B, L), generator=gen), 20).float()
    us = [u.to(DEV) for u in torch.rand(T, B, L, generator=gen)] if diverse else None
    sched, trans = PredefinedNoiseScheduleDiscrete("cosine", T).to(DEV), BlosumTransition(x_classes=20)
    made = []
    orig = S.GraphedDenoiseStep.__init__

    def spy(self, *a, **k):
        orig(self, *a, **k)
        made.append(self)
    S.GraphedDenoiseStep.__init__ = spy
    try:
        outs = [S.denoise(pk, model, sched, trans, diverse, x_T=x_T, us=us, timesteps=T, use_graph=g_) for g_ in (True, False, None)]
    finally:
        S.GraphedDenoiseStep.__init__ = orig
    assert len(made) == 2                                    # use_graph=True and the default (256 token rows) replay
    assert outs[0][2] == outs[1][2] == outs[2][2] and outs[0][3] == outs[1][3]
    assert all(len(seq) == int(n) for seq, n in zip(outs[0][2], pk["ligand_attn_mask"].sum(1)))
