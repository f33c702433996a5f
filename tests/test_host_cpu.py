"""CPU: the product's host logic (schedules, datasets, checkpoint keys, entry-point plumbing)
against the reference-generated fixtures, and the C-ABI library's load/exports (no compute)."""
import os
import re

import numpy as np
import pytest
import torch

from helpers import GOLDEN, synthetic_pockets

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(name):
    return torch.load(os.path.join(GOLDEN, name), weights_only=False)


# ------------------------------------------------------------------------------- C-ABI
def test_library_loads_and_exports_every_declared_symbol(pkg):
    header = open(os.path.join(ROOT, "include", "e3d_hip.h")).read()
    declared = set(re.findall(r"\b(e3d_[a-z0-9_]+)\s*\(", header))
    assert declared == set(pkg.hip.EXPORTS), declared ^ set(pkg.hip.EXPORTS)
    lib = pkg.hip.lib()
    for name in declared:
        assert getattr(lib, name) is not None
    assert lib.e3d_abi_version() == pkg.hip.ABI_VERSION
    # argument validation happens before any launch: callable without a GPU
    assert lib.e3d_gemm_bias_act_f32(None, 0, None, None, None, 0, 1, 128, 32, 0, None) < 0
    assert b"null pointer" in lib.e3d_last_error()


def test_product_does_not_import_the_oracle():
    pkg_dir = os.path.join(ROOT, "e3-invaraint-diffusion-model_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".sh")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, re.M), f
                assert "/root/reference" not in src, f


def test_cpu_inputs_fail_loudly(pkg):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=1,
             max_position_embeddings=16)
    m = ConditionalBertForDiffusionBase(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), 8)
    pk = synthetic_pockets(1, 16, seed=0, lig_range=(3, 9), rec_range=(6, 16))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        m(torch.zeros(1, dtype=torch.long), torch.zeros(1, 16, 8), pk["ligand_attn_mask"], pk["receptor_seq"],
          pk["receptor_angles"], pk["receptor_attn_mask"])


# ------------------------------------------------------------------------------- structure host side
def test_structure_utils_bit_exact(pkg):
    from e3diff_amd.structure_model import utils as U
    fx = load("structure_utils.pt")
    for T in (50, 1000):
        got = U.compute_alphas(U.cosine_beta_schedule(T))
        assert set(got) == set(fx[f"alphas_T{T}"])
        for k, v in fx[f"alphas_T{T}"].items():
            assert torch.equal(got[k], v), (T, k)
        tab = U.CosineTables(T)
        assert torch.equal(tab.sqrt_recip_alphas, 1.0 / torch.sqrt(fx[f"alphas_T{T}"]["alphas"]))
    assert torch.equal(U.modulo_with_wrapped_range(fx["wrap_in"]), fx["wrap_out"])
    assert U.modulo_with_wrapped_range(3, -2, 2) == -1
    a, b = fx["loss_in"]
    assert torch.equal(U.radian_l1_loss(a, b), fx["radian_l1"])
    assert torch.equal(U.radian_smooth_l1_loss(a, b, beta=torch.pi / 10), fx["radian_smooth_l1_b0.314"])
    assert torch.equal(U.radian_smooth_l1_loss(torch.tensor(-17.0466), torch.tensor(-1.3888), beta=0.1), fx["doc_smooth"])
    assert U.tolerant_comparison_check(-3.1415927410125732, ">=", -np.pi) is True    # utils.py:115-116
    assert U.tolerant_comparison_check([0.5, 4.0], "<=", np.pi) is False


def test_structure_checkpoint_keys_match_reference(pkg):
    """state_dict keys/shapes == the reference's (fixture 'shapes' came from the reference model
    under transformers 5.15, i.e. without distance_embedding) + the 4.38.2 rel-key tables."""
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion
    fx = load("structure_forward_tiny.pt")
    cfg = fx["cfg"]
    c = dict(hidden_size=cfg["hidden_size"], num_attention_heads=cfg["num_heads"],
             intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
             max_position_embeddings=cfg["max_seq_len"])
    m = ConditionalBertForDiffusion(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
                                    feature_names=list("abcdefgh"), loss_func=[])
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    extra = {k for k in shapes if k not in fx["shapes"]}
    assert {k: v for k, v in shapes.items() if k not in extra} == fx["shapes"]
    assert extra and all(k.endswith("self.distance_embedding.weight") for k in extra)
    P = cfg["max_seq_len"]
    assert all(shapes[k] == (2 * P - 1, 64) for k in extra)
    # 2 SELayers + 2 enc + 2 dec self-attentions carry a table; cross-attentions do not
    assert len(extra) == 2 + 2 * cfg["num_hidden_layers"]
    assert not any("crossattention" in k for k in extra)
    # SELayer.adaLN_modulation[0] starts at zero (model.py:50-51)
    assert float(m.receptor_emb.adaLN_modulation[0].weight.abs().sum()) == 0.0


def test_structure_dataset_layout_matches_reference(pkg):
    from e3diff_amd.structure_model.dataset import LigandBindingSiteDataset, NoisedAnglesDataset
    fx = load("structure_dataset.pt")
    ds = LigandBindingSiteDataset(None, "train", max_len=32, pocket_ext=1, records=fx["records"])
    assert len(ds) == fx["n_train"]
    for split, n in fx["split_sizes"].items():
        assert len(LigandBindingSiteDataset(None, split, 32, 1, records=fx["records"])) == n
    item = ds[0]
    assert item["structure_ids"] == fx["item0_ids"]
    assert set(item) - {"structure_ids"} == set(fx["item0"])
    for k, v in fx["item0"].items():
        if torch.is_tensor(v):
            assert torch.equal(item[k], v) and item[k].dtype == v.dtype, k
        else:
            assert item[k] == v, k
    nds = NoisedAnglesDataset(ds, timesteps=100)
    real = torch.randn_like
    torch.randn_like = lambda x, **k: fx["injected_randn"]
    try:
        nitem = nds.__getitem__(1, use_timestep=37)
    finally:
        torch.randn_like = real
    for k, v in fx["noised_item1"].items():
        if torch.is_tensor(v):
            assert torch.equal(nitem[k], v), k
    with pytest.raises(RuntimeError, match="Length exceed"):
        LigandBindingSiteDataset(None, "train", max_len=4, pocket_ext=1, records=fx["records"])[0]
    with pytest.raises(IndexError):
        ds[len(ds)]


def test_structure_loss_terms_match_reference(pkg):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion
    fx = load("structure_forward_tiny.pt")
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=1,
             max_position_embeddings=16)
    m = ConditionalBertForDiffusion(
        BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
        loss_func=[ConditionalBertForDiffusion.diheral_loss_func] * 4 + [ConditionalBertForDiffusion.angle_loss_func] * 4,
        epochs=350, lr_scheduler="LinearWarmup", l2_lambda=0.1)
    got = m.loss_terms_from_prediction(fx["loss_pred"], fx["known_noise"], fx["pockets"]["ligand_attn_mask"])
    assert torch.allclose(got, fx["loss_terms"], rtol=1e-6, atol=1e-7)
    opt = m.configure_optimizers()
    assert isinstance(opt["optimizer"], torch.optim.AdamW) and opt["lr_scheduler"]["interval"] == "epoch"
    sch = opt["lr_scheduler"]["scheduler"]
    lrs = []
    for _ in range(40):
        lrs.append(sch.get_last_lr()[0])
        opt["optimizer"].step()
        sch.step()
    assert lrs[0] == 0.0 and lrs[35] == pytest.approx(5e-5) and lrs[36] < 5e-5     # warm-up = 10 % of 350 epochs


# ------------------------------------------------------------------------------- sequence host side
def test_sequence_utils_bit_exact(pkg):
    from e3diff_amd.sequence_model import utils as U
    fx = load("sequence_utils.pt")
    sched = U.PredefinedNoiseScheduleDiscrete("cosine", 50)
    assert torch.equal(sched.betas, fx["betas"]) and torch.equal(sched.alphas_bar, fx["alphas_bar"])
    t_norm = (torch.arange(51).float() / 50).unsqueeze(1)
    ab = sched.get_alpha_bar(t_normalized=t_norm)
    assert torch.equal(ab, fx["alpha_bar_of_t"])
    assert torch.equal(sched(t_normalized=t_norm), fx["betas"][torch.arange(51)].unsqueeze(1))
    bl = U.BlosumTransition(x_classes=20)
    assert torch.equal(bl.temperature_list, fx["blosum_temperature_501"])
    assert torch.equal(bl.table_index(ab), fx["blosum_t_index"])
    assert torch.equal(bl.table_index(fx["round_probe_in"]), fx["round_probe_idx"])      # round-half-even
    assert torch.equal(bl.get_Qt_bar(ab, "cpu"), fx["blosum_Qtb"])
    assert torch.equal(U.DiscreteUniformTransition(20).get_Qt_bar(ab, "cpu"), fx["uniform_Qtb"])
    assert torch.equal(U.elbo_loss(*fx["elbo_in"]), fx["elbo"])
    assert bl.get_Qt(ab, "cpu").shape == (51, 20, 20)
    assert torch.allclose(U.DiscreteUniformTransition(20).get_Qt(torch.tensor([[0.25]]), "cpu").sum(-1),
                          torch.ones(1, 20))


def test_sequence_checkpoint_keys_and_init_match_reference(pkg):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    fx = load("sequence_forward_tiny.pt")
    cfg = fx["cfg"]
    c = dict(hidden_size=cfg["hidden_size"], num_attention_heads=cfg["num_heads"],
             intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
             max_position_embeddings=cfg["max_seq_len"])
    m = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
                    feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                    noise_schedule="cosine", timesteps=50, max_epochs=150, lr_scheduler="LinearWarmup")
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    extra = {k for k in shapes if k not in fx["shapes"]}
    assert {k: v for k, v in shapes.items() if k not in extra} == fx["shapes"]
    assert extra and all(k.endswith("self.distance_embedding.weight") for k in extra)
    assert any(k.startswith("receptor_feature_emb.") for k in shapes)        # dead weight kept (quirk)
    sd = m.state_dict()
    assert float(sd["decoder_normalize.adaLN_modulation.0.weight"].abs().sum()) == 0.0
    assert float(sd["ligand_feature_emb.adaLN_modulation.0.weight"].abs().sum()) > 0.0   # xavier, model.py:183-198
    assert float(sd["decoder.layer.0.output.dense.bias"].abs().sum()) == 0.0
    assert m.configure_optimizers()["lr_scheduler"]["interval"] == "epoch"


def test_sequence_dataset_adds_ligand_seq(pkg):
    from e3diff_amd.sequence_model.dataset import LigandBindingSiteDataset
    fx = load("structure_dataset.pt")
    ds = LigandBindingSiteDataset(None, "train", max_len=32, pocket_ext=1, records=fx["records"])
    item = ds[0]
    assert ds.feature_names == list("ACDEFGHIKLMNPQRSTVWY")
    assert item["ligand_seq"].shape == (32, 20)
    n = int(item["ligand_length"])
    assert bool((item["ligand_seq"][:n].sum(-1) == 1).all()) and float(item["ligand_seq"][n:].abs().sum()) == 0
    for k, v in fx["item0"].items():
        if torch.is_tensor(v):
            assert torch.equal(item[k], v), k


def test_load_generated_angles_roundtrip(pkg, tmp_path):
    import pickle
    from e3diff_amd.sequence_model.sample_by_generated_angles import load_generated_angles
    arrs = [np.random.randn(7, 5, 8).astype(np.float32), np.random.randn(9, 8).astype(np.float32),
            np.random.randn(3, 8).astype(np.float32)]
    p = tmp_path / "out.pkl"
    pickle.dump(arrs, open(p, "wb"))
    chunks = load_generated_angles(str(p), max_seq_len=16, batch_size=2)
    assert [c.shape for c in chunks] == [(2, 16, 8), (1, 16, 8)]
    assert np.allclose(chunks[0][0, :5].numpy(), arrs[0][-1]) and float(chunks[0][0, 5:].abs().sum()) == 0
    assert np.allclose(chunks[1][0, :3].numpy(), arrs[2])


def test_biolip_file_is_loaded_without_executing_anything(tmp_path):
    """A user-supplied biolip.pt goes through torch.load(weights_only=True): the schema's plain containers and
    tensors load, a pickle that would run code is refused with a schema error (never executed)."""
    import pickle

    from e3diff_amd import biolip
    from e3diff_amd.structure_model.dataset import LigandBindingSiteDataset
    good = tmp_path / "biolip.pt"
    biolip.write_synthetic(str(good), 6, seed=3)
    assert len(biolip.load(str(good))) == 6
    assert len(LigandBindingSiteDataset(str(good), None, max_len=64).data) == 6

    marker = tmp_path / "executed"

    class Evil:
        def __reduce__(self):
            return (open, (str(marker), "w"))

    bad = tmp_path / "evil.pt"
    with open(bad, "wb") as f:
        pickle.dump([Evil()], f)
    with pytest.raises(biolip.BiolipSchemaError):
        biolip.load(str(bad))
    with pytest.raises(biolip.BiolipSchemaError):
        LigandBindingSiteDataset(str(bad), None)
    assert not marker.exists()


def test_joint_handover_matches_reference_load_generated_angles(pkg, tmp_path):
    """Pins SURVEY 8(f) rank 1 against the reference itself: tests/golden/joint_handover.pt holds the chunks the
    reference's ``load_generated_angles`` (sequence_model/sample_by_generated_angles.py:54-66, ast-extracted and run
    by tests/golden/make_joint_fixture.py) produced from a pickle of ragged per-ligand [l_i, 8] arrays.
    (1) the product's file loader returns the same chunks bit for bit; (2) the on-device hand-over
    ``angles_from_trajectory`` -- the path the joint sampler really takes, no pickle -- produces exactly that padded
    tensor from a structure trajectory whose padding rows hold arbitrary values."""
    import pickle
    from e3diff_amd.sequence_model.sample_by_generated_angles import angles_from_trajectory, load_generated_angles
    fx = torch.load(os.path.join(GOLDEN, "joint_handover.pt"), weights_only=True)
    cfg, lengths = fx["config"], fx["lengths"]
    p = tmp_path / "output.pkl"
    with open(p, "wb") as f:
        pickle.dump([a.numpy() for a in fx["arrays"]], f)
    got = load_generated_angles(str(p), max_seq_len=cfg["max_seq_len"], batch_size=cfg["batch_size"])
    assert len(got) == len(fx["chunks"])
    for a, b in zip(got, fx["chunks"]):
        assert a.dtype == b.dtype and torch.equal(a, b)
    # the device path: a [T,B,L,8] trajectory whose last step carries the generated angles on the valid rows and
    # garbage on the padding rows (the sampler noises padding too)
    want = torch.cat(fx["chunks"])
    n, L = want.shape[0], cfg["max_seq_len"]
    mask = (torch.arange(L)[None, :] < torch.tensor(lengths)[:, None]).float()
    g = torch.Generator().manual_seed(2)
    traj = torch.randn(3, n, L, 8, generator=g)
    for i, a in enumerate(fx["arrays"]):
        traj[-1, i, :lengths[i]] = a
    assert torch.equal(angles_from_trajectory(traj, mask), want)
    assert torch.equal(angles_from_trajectory(traj[-1], mask), want)


def test_weight_caches_follow_fused_optimizer_steps(pkg):
    """torch's fused AdamW updates parameters WITHOUT bumping ``_version`` (checked here, so that the day torch changes
    that the comment in ops.py can go); the derived-weight caches (packed QKV, W^T, distance-table planes) must
    nevertheless be rebuilt after an optimizer step: their key carries ops.PARAM_GENERATION, bumped by a global
    optimizer post-step hook."""
    from e3diff_amd import autograd as AG, bert, ops
    w = torch.nn.Parameter(torch.randn(8, 4))
    opt = torch.optim.AdamW([w], lr=1e-2, fused=True)
    wt0 = AG._transposed_weight(w)
    assert AG._transposed_weight(w) is wt0                       # cached while nothing changes
    wt0_values = wt0.clone()                                     # (the buffer is refreshed in place)
    att = bert.BertSelfAttention(bert.BertConfig(hidden_size=256, num_attention_heads=4, max_position_embeddings=8))
    opt2 = torch.optim.AdamW(att.parameters(), lr=1e-2, fused=True)
    with torch.no_grad():
        pack0 = bert.qkv_weights(att)[0]
        assert bert.qkv_weights(att)[0] is pack0
    w.grad = torch.ones_like(w)
    v0, gen0 = w._version, ops.PARAM_GENERATION
    opt.step()
    assert ops.PARAM_GENERATION > gen0
    fused_bumps_version = w._version != v0
    wt1 = AG._transposed_weight(w)
    assert torch.equal(wt1, w.detach().t()) and not torch.equal(wt1, wt0_values)
    for p in att.parameters():
        p.grad = torch.ones_like(p)
    opt2.step()
    with torch.no_grad():
        pack1 = bert.qkv_weights(att)[0]
    assert pack1 is not pack0 and torch.equal(pack1[:256], att.query.weight.detach())
    print("fused AdamW bumps _version:", fused_bumps_version)


def test_deferred_weight_gradient_queue_reports_parameters_in_stages(pkg):
    """autograd.deferred_weight_grads with a gradient averager listening: the queue (backward order = last layers first)
    is flushed in slices and every parameter is reported once, after the slice of its LAST use (host logic only: the
    launches themselves are replaced by a recorder; the kernels are covered by tests/test_backward_gpu.py)."""
    from e3diff_amd.autograd import deferred_weight_grads
    ws = [torch.nn.Parameter(torch.zeros(2, 2)) for _ in range(7)]
    index = {id(w): i for i, w in enumerate(ws)}
    calls = []
    q = deferred_weight_grads(on_param=lambda p: calls.append(("ready", index[id(p)])), stages=3)

    def fake_slice(items):
        calls.append(("slice", [index[id(it[2])] for it in items]))
        return {id(it[2]): it[2] for it in items}

    q._flush_slice = fake_slice
    for i in (6, 5, 4, 3, 2, 1, 0, 6):      # weight 6 is used twice (first and last in backward order)
        q.add(None, None, ws[i], None)
    q.flush()
    assert calls == [("slice", [6, 5, 4]), ("ready", 5), ("ready", 4), ("slice", [3, 2, 1]), ("ready", 3), ("ready", 2),
                     ("ready", 1), ("slice", [0, 6]), ("ready", 0), ("ready", 6)]
    assert q.pending == []
    # nobody listening: one slice (the largest groups)
    calls.clear()
    q2 = deferred_weight_grads()
    q2._flush_slice = fake_slice
    for i in (2, 1, 0):
        q2.add(None, None, ws[i], None)
    q2.flush()
    assert calls == [("slice", [2, 1, 0])]


def test_clip_adamw_on_cpu_parameters_is_torch_adamw(pkg):
    """optim.ClipAdamW with parameters the HIP kernels do not cover (here: CPU tensors) behaves exactly as
    clip_grad_norm_ + torch.optim.AdamW -- the host logic around the kernels (step_clipped, fallback, state layout)."""
    from e3diff_amd.optim import ClipAdamW
    from e3diff_amd.training import clip_and_step
    torch.manual_seed(0)
    base = [torch.randn(17, 5), torch.randn(33)]
    outs = []
    for impl in ("torch", "ours"):
        ps = [torch.nn.Parameter(b.clone()) for b in base]
        opt = torch.optim.AdamW(ps, lr=1e-2, weight_decay=0.1) if impl == "torch" else ClipAdamW(ps, lr=1e-2, weight_decay=0.1)
        for step in range(3):
            for i, p in enumerate(ps):
                p.grad = torch.full_like(p, 3.0 * (i + 1) * (step + 1))
            if impl == "torch":
                n = torch.nn.utils.clip_grad_norm_(ps, 1.0)
                opt.step()
            else:
                n = clip_and_step(ps, opt, 1.0)
        outs.append((ps, float(n), opt.state_dict()))
    assert outs[0][1] == outs[1][1]
    assert all(torch.equal(a, b) for a, b in zip(outs[0][0], outs[1][0]))
    assert outs[0][2]["param_groups"][0].keys() == outs[1][2]["param_groups"][0].keys()


def test_training_packs_qkv_by_tying_parameter_storage(pkg):
    """bert.qkv_weights in training mode: query / key / value parameters become row blocks of ONE buffer (no cat per
    step); the packed pair aliases it, follows in-place updates for free, routes gradients to the parameters when the
    weight gradients are not deferred, survives a parameter being moved, and leaves names / shapes / state_dict alone."""
    from e3diff_amd import bert
    torch.manual_seed(0)
    att = bert.BertSelfAttention(bert.BertConfig(hidden_size=128, num_attention_heads=2, max_position_embeddings=8))
    want = torch.cat([att.query.weight, att.key.weight, att.value.weight]).detach().clone()
    keys0 = list(att.state_dict().keys())
    w, b = bert.qkv_weights(att)
    assert torch.equal(w.detach(), want) and w.requires_grad and b.requires_grad
    assert w.data_ptr() == att.query.weight.data_ptr() and att.value.weight.data_ptr() == w.data_ptr() + 4 * 2 * 128 * 128
    assert w._e3d_parts[1] is att.key.weight and list(att.state_dict().keys()) == keys0
    assert att.key.weight.shape == (128, 128) and att.key.weight.is_contiguous() and att.key.weight.is_leaf
    (2 * w.sum() + 3 * b.sum()).backward()
    assert torch.equal(att.key.weight.grad, torch.full((128, 128), 2.0)) and torch.equal(att.value.bias.grad, torch.full((128,), 3.0))
    opt = torch.optim.SGD(att.parameters(), lr=0.5)
    opt.step()                                                    # in place: the packed buffer follows without a copy
    w2, _ = bert.qkv_weights(att)
    assert w2.data_ptr() == w.data_ptr() and torch.equal(w2.detach(), want - 1.0)
    att.key.weight.data = att.key.weight.data.clone() + 5.0       # a parameter moved: re-tied, values kept
    w3, _ = bert.qkv_weights(att)
    assert torch.equal(w3.detach()[128:256], want[128:256] - 1.0 + 5.0) and att.key.weight.data_ptr() == w3.data_ptr() + 4 * 128 * 128
    with torch.no_grad():                                         # the inference cache sees the same values
        assert torch.equal(bert.qkv_weights(att)[0], w3.detach())
    sd = att.state_dict()
    att2 = bert.BertSelfAttention(bert.BertConfig(hidden_size=128, num_attention_heads=2, max_position_embeddings=8))
    att2.load_state_dict(sd, strict=True)
    assert torch.equal(att2.value.weight, att.value.weight)


def test_training_losses_without_index_lists_equal_the_indexed_forms(pkg):
    """The training steps take their means over the un-padded / noised positions as masked sums (no ``torch.where(mask)``
    or boolean-mask indexing: those cost a device-to-host synchronisation per step): same values and same gradients as the
    reference's indexed forms (structure_model/model.py:290-303, sequence_model/model.py:313-345), in fp64 on the CPU."""
    from e3diff_amd.sequence_model import model as SQ
    from e3diff_amd.structure_model import model as ST
    torch.manual_seed(0)
    B, L = 6, 16
    mask = torch.arange(L)[None] < torch.tensor([3, 16, 1, 9, 12, 5])[:, None]

    class S:
        loss_func = [ST.ConditionalBertForDiffusion.diheral_loss_func] * 4 + [ST.ConditionalBertForDiffusion.angle_loss_func] * 4
    pred = (torch.randn(B, L, 8, dtype=torch.float64) * 3).requires_grad_(True)
    noise = torch.randn(B, L, 8, dtype=torch.float64) * 3
    outs = []
    for masked in (True, False):
        ST.MASKED_LOSS = masked
        t = ST.ConditionalBertForDiffusion.loss_terms_from_prediction(S(), pred, noise, mask.float())
        outs.append((t, torch.autograd.grad(t.mean(), pred)[0]))
    ST.MASKED_LOSS = True
    assert (outs[0][0] - outs[1][0]).abs().max() < 1e-14 and (outs[0][1] - outs[1][1]).abs().max() < 1e-14
    # a loss callable this package does not know keeps the indexed path
    S.loss_func = [lambda a, b: (a - b).abs().mean()] * 8
    t = ST.ConditionalBertForDiffusion.loss_terms_from_prediction(S(), pred, noise, mask.float())
    assert torch.allclose(t[0], (pred - noise).abs()[..., 0][mask].mean())

    class Q:
        loss_function = torch.nn.CrossEntropyLoss()

        def forward(self, *a):
            return logits
    true = torch.randint(0, 20, (B, L))
    seq = torch.nn.functional.one_hot(true, 20).double() * mask[..., None]
    noised_idx = torch.where(torch.rand(B, L) < 0.5, torch.randint(0, 20, (B, L)), true)
    noised = torch.nn.functional.one_hot(noised_idx, 20).double() * mask[..., None]
    logits = torch.randn(B, L, 20, dtype=torch.float64, requires_grad=True)
    batch = dict(ligand_attn_mask=mask.float(), ligand_seq=seq, ligand_angles=None, receptor_seq=None, receptor_angles=None,
                 receptor_attn_mask=None)
    outs = []
    for masked in (True, False):
        SQ.MASKED_LOSS = masked
        six = SQ.PeptideDiff.get_loss(Q(), batch, None, noised)
        outs.append((torch.stack([x.double() for x in six]).detach(), torch.autograd.grad(six[0], logits)[0]))
    SQ.MASKED_LOSS = True
    assert (outs[0][0] - outs[1][0]).abs().max() < 1e-14 and (outs[0][1] - outs[1][1]).abs().max() < 1e-14
    assert not SQ._is_plain_cross_entropy(torch.nn.CrossEntropyLoss(label_smoothing=0.1))
