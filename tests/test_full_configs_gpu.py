"""The BASELINE.json configs at their FULL sizes through the HIP path, plus the
arithmetic-margin evidence for the default bf16x3 mode (item 1b).

Kernel selection depends on shape (128-row vs persistent 256x256 GEMM, cached-W^T dgrad, split-K wgrad,
cooperative attention with 4 or 8 waves), so reduced-size oracle parity does not exercise what these configs
launch.  Pattern (tests/test_structure_gpu.py::test_full_bench_size_properties): the first items of the full batch
against the same items run alone, that small run against the CPU oracle; gradients of a full-size step against
the fp32-grade kernels at the same size and against the oracle at full M with fewer layers.

  config 1  structure sampling, 1 x 64-residue pocket, 50 timesteps          test_config1_*
  config 2  structure training step, 32 x 128-residue pockets                test_config2_*
  config 3  (256 x 256 sampling step: tests/test_structure_gpu.py::test_full_bench_size_properties)
  config 4  sequence (BLOSUM) training step, 64 x 128 per rank               test_config4_*
  config 5  joint structure -> sequence sampling, 128 x 128 per GPU          test_config5_*
"""
import os

import pytest
import torch

from helpers import (FULL_SEQ, FULL_STRUCT, GOLDEN, elementwise_err, heavy_tailed_state_dict, rel_err, rescaled_state_dict,
                     reverse_step_tolerance, seeded_state_dict, synthetic_pockets)
from oracle import sequence as oseq
from oracle import structure as ostr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4
# bf16x3 against bf16x6 gradients of a full-size step (see grad_errors): whole-gradient L2, per-parameter L2, per-parameter max-norm
GLOBAL_L2_TOL, PARAM_L2_TOL, PARAM_MAX_TOL = 1e-4, 5e-2, 3e-2   # measured: 2.1e-5 / 1.9e-2 / 9.7e-3 (config 2)


def to_dev(d):
    return {k: v.to(DEV) for k, v in d.items() if torch.is_tensor(v)}


def gen(seed):
    return torch.Generator().manual_seed(seed)


def struct_trainer(L, layers, seed, dropout=0.0):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers,
             max_position_embeddings=L, hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
              loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed)
    model.load_state_dict(sd)
    return model.train().to(DEV), sd


def seq_trainer(L, layers, seed, T=50):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers,
             max_position_embeddings=L, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
                        feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                        noise_schedule="cosine", timesteps=T, l2_lambda=0.1)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=seed)
    model.load_state_dict(sd)
    return model.train().to(DEV), sd


def grads_of(model):
    return {k: p.grad.detach().clone() for k, p in model.named_parameters() if p.grad is not None}


def grad_errors(got, want, skip=("self.key.bias",)):
    """Per-parameter errors of a gradient set against a reference set, two ways:
      "max": max|d| / max(|want|max, 1e-3 * median gradient magnitude)  (the figure tests/test_backward_gpu.py bounds)
      "l2":  ||d||_2 / max(||want||_2, 1e-3 * median norm)              (norm-wise: what an optimizer step sees)
    plus "global": ||d||_2 / ||want||_2 over ALL parameters.  Key biases have an exactly-zero gradient (softmax shift
    invariance) and are only checked to be noise-sized."""
    scale = torch.stack([v.abs().max() for v in want.values()]).median().item()
    nscale = torch.stack([v.float().norm() for v in want.values()]).median().item()
    mx, l2, num, den = {}, {}, 0.0, 0.0
    for k, w in want.items():
        if any(k.endswith(s) for s in skip):
            assert float(got[k].abs().max()) < 1e-2 * scale, k
            continue
        d = got[k].double().cpu() - w.double().cpu()
        mx[k] = (d.abs().max() / max(w.abs().max().item(), 1e-3 * scale)).item()
        l2[k] = (d.norm() / max(w.double().norm().item(), 1e-3 * nscale)).item()
        num += float(d.pow(2).sum())
        den += float(w.double().pow(2).sum())
    return {"max": mx, "l2": l2, "global": (num / den) ** 0.5}


def report_grad_errors(tag, errs, capsys):
    top = lambda d: ", ".join(f"{k.replace('attention', 'att')}={v:.1e}" for k, v in sorted(d.items(), key=lambda kv: -kv[1])[:3])  # noqa: E731
    with capsys.disabled():
        print(f"\n[{tag}] gradient error bf16x3 vs bf16x6 at full size: global L2 {errs['global']:.2e}; "
              f"per-parameter L2 worst: {top(errs['l2'])}; per-parameter max-norm worst: {top(errs['max'])}")


# ------------------------------------------------------------------------------------ config 1
_CONFIG1_ORACLE = {}


def _config1_oracle(sd, pk, x_T, noises, betas, T, L):
    """The CPU oracle's 50-step chain: the same for every arithmetic mode of the test below (same seeds), ~35 s of host
    time -- computed once per session instead of once per mode."""
    if "want" not in _CONFIG1_ORACLE:
        fn = lambda t, x, lm, rs, ra, rm: ostr.forward(sd, {"num_heads": 12, "max_pos": L}, t, x, lm, rs, ra, rm)  # noqa: E731
        _CONFIG1_ORACLE["want"] = ostr.p_sample_loop(fn, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"],
                                                     pk["receptor_angles"], T, betas, noises=noises)
    return _CONFIG1_ORACLE["want"]


@pytest.mark.parametrize("mode", ["bf16x6", "bf16x3", "f16x3"])
def test_config1_single_pocket_64_residues_50_steps(pkg, hip, mode, capsys):
    """BASELINE configs[0] at its full size on the GPU: ONE 64-residue pocket, the full 12+12-layer model, all 50
    reverse steps of a T=50 schedule, teacher-forced against the CPU oracle's chain (each step restarted from the
    oracle's previous state: one step amplifies an eps error up to 100x, helpers.reverse_step_tolerance).
    B=1 launches the small-grid kernels: 128-row GEMM tiles, the 2-query-tile attention groups."""
    from test_structure_gpu import build, teacher_forced_steps
    L, B, T = 64, 1, 50
    model, sd = build(pkg, FULL_STRUCT, L, seed=31)
    pk = synthetic_pockets(B, L, seed=32)
    g = gen(33)
    x_T = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g))
    noises = torch.randn(T, B, L, 8, generator=g)
    betas = ostr.cosine_beta_schedule(T)
    want = _config1_oracle(sd, pk, x_T, noises, betas, T, L)
    with pkg.ops.arithmetic(mode, respect_env=False):
        teacher_forced_steps(model, to_dev(pk), x_T, want, noises, betas, T)
        # the loop entry point itself at this size: same chain, free-running, finite and wrapped
        from e3diff_amd.structure_model.sample import p_sample_loop
        d = to_dev(pk)
        traj = p_sample_loop(model, d["ligand_attn_mask"], x_T.to(DEV), d["receptor_seq"], d["receptor_attn_mask"],
                             d["receptor_angles"], T, betas, disable_pbar=True, noises=noises.to(DEV), step=1)
    assert traj.shape == (T, B, L, 8) and torch.isfinite(traj).all() and traj.abs().max() <= 3.1416
    # the first step (t = T-1) of the free-running chain IS a teacher-forced step
    d0 = ostr.modulo_with_wrapped_range(traj[0] - want[0]).abs().max().item()
    assert d0 < reverse_step_tolerance(betas, T - 1, eps_scale=4.0, rel=TOL)


# ------------------------------------------------------------------------------------ config 2
def _structure_batch(B, L, seed):
    pk = synthetic_pockets(B, L, seed=seed)
    g = gen(seed + 1)
    batch = dict(pk, timestep=torch.randint(0, 1000, (B, 1), generator=g),
                 noised_ligand_angle=ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g)),
                 known_noise=ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g)))
    return pk, batch


def test_config2_structure_training_step_32x128_full_depth(pkg, hip, capsys):
    """BASELINE configs[1]: one structure training step at B=32 x L=128 (M=4096 rows), 12+12 layers, bf16x3.
    (a) predictions of the first 2 items (train-mode graph, dropout 0) against the CPU oracle;
    (b) EVERY parameter gradient of the bf16x3 step against the fp32-grade (bf16x6) kernels at the same size -- other
        kernels for the same maths (whose small-size forms are oracle-pinned in tests/test_backward_gpu.py);
    (c) the step is usable: finite loss / gradients, gradient-norm clip + fused AdamW step, loss changes."""
    B, L = 32, 128
    model, sd = struct_trainer(L, 12, seed=41)
    pk, batch = _structure_batch(B, L, seed=42)
    db = to_dev(batch)
    grads, losses = {}, {}
    for mode in ("bf16x3", "bf16x6"):
        with pkg.ops.arithmetic(mode, respect_env=False):
            model.zero_grad(set_to_none=True)
            loss = model.training_step(db)
            loss.backward()
            grads[mode], losses[mode] = grads_of(model), float(loss.detach())
            if mode == "bf16x3":
                with torch.no_grad():
                    pred = model(db["timestep"][:2], db["noised_ligand_angle"][:2], db["ligand_attn_mask"][:2],
                                 db["receptor_seq"][:2], db["receptor_angles"][:2], db["receptor_attn_mask"][:2])
    want = ostr.forward(sd, {"num_heads": 12, "max_pos": L}, batch["timestep"][:2], batch["noised_ligand_angle"][:2],
                        pk["ligand_attn_mask"][:2], pk["receptor_seq"][:2], pk["receptor_angles"][:2],
                        pk["receptor_attn_mask"][:2])
    assert rel_err(pred, want) < TOL
    assert losses["bf16x3"] == pytest.approx(losses["bf16x6"], rel=2e-4)
    assert set(grads["bf16x3"]) == set(grads["bf16x6"]) and all(torch.isfinite(v).all() for v in grads["bf16x3"].values())
    errs = grad_errors(grads["bf16x3"], grads["bf16x6"])
    report_grad_errors(f"config 2, B=32 L=128 12+12, loss {losses['bf16x3']:.5f}", errs, capsys)
    # What an optimizer step sees is the norm-wise error.  The max-norm figure of single parameters is larger for the
    # query / key projections of the deepest layers: their gradient passes through dS = P * (dP - rowsum(P dP)), a
    # cancellation that is near-total while the softmax rows are close to uniform (random init), so a 2^-17 product
    # error is amplified ~1000x there.  (The reference trains at torch's "medium" matmul precision -- bf16 products,
    # 2^-8 per product, structure_model/train_model.py:120 -- i.e. 500x coarser than bf16x3.)
    assert errs["global"] < GLOBAL_L2_TOL and max(errs["l2"].values()) < PARAM_L2_TOL and max(errs["max"].values()) < PARAM_MAX_TOL, errs["global"]
    # (d) the step as training.fit runs it: weight / bias gradients of all linear layers computed at the end of the
    # backward pass, grouped by shape (autograd.deferred_weight_grads) -- the same sums in another order (whole
    # reductions instead of split-K atomics), so they agree with the layer-by-layer gradients far inside the
    # arithmetic's own error; and the grouped launches really ran (12+12 layers: several groups of >= 192 tiles)
    from e3diff_amd.autograd import deferred_weight_grads
    with pkg.ops.arithmetic("bf16x3", respect_env=False):
        model.zero_grad(set_to_none=True)
        loss = model.training_step(db)
        with deferred_weight_grads() as q:
            loss.backward()
            n_queued = len(q.pending)
        grouped = grads_of(model)
    assert n_queued > 200 and set(grouped) == set(grads["bf16x3"])
    errs_g = grad_errors(grouped, grads["bf16x3"])
    report_grad_errors("config 2, grouped weight gradients vs layer by layer (both bf16x3)", errs_g, capsys)
    assert errs_g["global"] < 2e-6 and max(errs_g["l2"].values()) < 2e-5, errs_g["global"]
    optim = model.configure_optimizers()["optimizer"]
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.requires_grad], 1.0)
    optim.step()
    with pkg.ops.arithmetic("bf16x3", respect_env=False):
        after = float(model.training_step(db).detach())
    assert after == after and after != losses["bf16x3"]


@pytest.mark.parametrize("mode,tol", [("bf16x6", 2e-5), ("bf16x3", 2e-4)])
def test_config2_m4096_gradients_match_oracle_and_wt_cache_survives_optimizer_step(pkg, hip, mode, tol, capsys):
    """M = 4096 rows (B=32 x L=128), 2+2 layers: every parameter gradient against CPU autograd of the oracle at the
    shapes config 2 launches (cached-W^T input-gradient GEMM for M >= 1024, split-K weight gradients for K >= 1024).
    Then an optimizer step (fused AdamW) and a second backward: the W^T cache, keyed on (version, data_ptr), must be
    rebuilt -- dx through the cached path must equal the K-major path that reads W itself."""
    from test_backward_gpu import _grad_compare
    from e3diff_amd import autograd as AG
    B, L = 32, 128
    model, sd = struct_trainer(L, 2, seed=43)
    pk, batch = _structure_batch(B, L, seed=44)
    db = to_dev(batch)
    with pkg.ops.arithmetic(mode, respect_env=False):
        loss = model.training_step(db)

        def ref_loss(rsd):
            pred = ostr.forward(rsd, {"num_heads": 12, "max_pos": L}, batch["timestep"], batch["noised_ligand_angle"],
                                pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
            return ostr.loss_terms(pred, batch["known_noise"], pk["ligand_attn_mask"]).mean()

        worst = _grad_compare(model, sd, loss, ref_loss, tol)
        with capsys.disabled():
            print(f"\n[config 2 shapes, M=4096, 2+2 layers, {mode}] worst per-parameter gradient error vs oracle {worst:.2e}")
        # ---- W^T cache across an optimizer step
        w = model.decoder.layer[0].output.dense.weight      # [768, 1024]: dx = dz . W through the cached W^T
        wt_before = AG._transposed_weight(w).clone()        # (the cached buffer is refreshed in place)
        assert torch.equal(wt_before, w.detach().t())
        optim = model.configure_optimizers()["optimizer"]
        optim.step()                                    # in-place update: bumps w._version
        dz = torch.randn(B * L, w.shape[0], device=DEV, generator=torch.Generator(device=DEV).manual_seed(1))
        wt_after = AG._transposed_weight(w)
        assert torch.equal(wt_after, w.detach().t()) and not torch.equal(wt_after, wt_before)
        dx_cached = pkg.ops.gemm(dz, wt_after, None)
        dx_direct = AG.gemm_general(dz, False, w.detach(), True, B * L, w.shape[1], w.shape[0])
        assert rel_err(dx_cached, dx_direct) < (2e-6 if mode == "bf16x6" else 2e-5)
        # and through autograd: a second full backward after the step gives finite, different gradients
        g_before = grads_of(model)
        model.zero_grad(set_to_none=True)
        model.training_step(db).backward()
        g_after = grads_of(model)
        assert all(torch.isfinite(v).all() for v in g_after.values())
        assert any(not torch.equal(g_after[k], g_before[k]) for k in g_after)


# ------------------------------------------------------------------------------------ config 4
def _sequence_batch(B, L, seed, T=50):
    pk = synthetic_pockets(B, L, seed=seed, with_ligand_seq=True)
    g = gen(seed + 1)
    t_int = torch.randint(1, T, (B, 1), generator=g).float()
    blosum = torch.load(os.path.join(GOLDEN, "blosum_substitute.pt"), weights_only=True)
    noised = oseq.apply_aa_noise(pk["ligand_seq"], t_int, oseq.NoiseScheduleDiscrete(T), oseq.BlosumTransition(blosum),
                                 u=torch.rand(B * L, generator=g))
    return pk, t_int, noised


def test_config4_sequence_training_step_64x128_full_depth(pkg, hip, capsys):
    """BASELINE configs[3], one rank's share: the sequence model's BLOSUM training loss at B=64 x L=128 (M=8192),
    6 decoder layers.  Same three checks as config 2; gradient averaging over ranks is tests/test_sharding_cpu.py."""
    B, L, T = 64, 128, 50
    model, sd = seq_trainer(L, 6, seed=51, T=T)
    pk, t_int, noised = _sequence_batch(B, L, seed=52, T=T)
    dpk = to_dev(pk)
    grads, losses = {}, {}
    for mode in ("bf16x3", "bf16x6"):
        with pkg.ops.arithmetic(mode, respect_env=False):
            model.zero_grad(set_to_none=True)
            loss = model.get_loss(dpk, (t_int / T).to(DEV), noised.to(DEV))[0]
            loss.backward()
            grads[mode], losses[mode] = grads_of(model), float(loss.detach())
            if mode == "bf16x3":
                with torch.no_grad():
                    pred = model.forward((t_int / T)[:2].to(DEV), noised[:2].to(DEV), dpk["ligand_angles"][:2],
                                         dpk["ligand_attn_mask"][:2], dpk["receptor_seq"][:2], dpk["receptor_angles"][:2],
                                         dpk["receptor_attn_mask"][:2])
    want = oseq.forward(sd, {"num_heads": 12, "max_pos": L}, (t_int / T)[:2], noised[:2], pk["ligand_angles"][:2],
                        pk["ligand_attn_mask"][:2], pk["receptor_seq"][:2], pk["receptor_angles"][:2],
                        pk["receptor_attn_mask"][:2])
    assert rel_err(pred, want) < TOL
    assert losses["bf16x3"] == pytest.approx(losses["bf16x6"], rel=2e-4)
    errs = grad_errors(grads["bf16x3"], grads["bf16x6"])
    report_grad_errors(f"config 4, B=64 L=128 6 layers, loss {losses['bf16x3']:.5f}", errs, capsys)
    assert errs["global"] < GLOBAL_L2_TOL and max(errs["l2"].values()) < PARAM_L2_TOL and max(errs["max"].values()) < PARAM_MAX_TOL, errs["global"]
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("receptor_feature_emb."))
    optim = model.configure_optimizers()["optimizer"]
    torch.nn.utils.clip_grad_norm_([p for p in model.parameters() if p.grad is not None], 1.0)
    optim.step()
    with pkg.ops.arithmetic("bf16x3", respect_env=False):
        after = float(model.get_loss(dpk, (t_int / T).to(DEV), noised.to(DEV))[0].detach())
    assert after == after and after != losses["bf16x3"]


def test_config4_m8192_gradients_match_oracle(pkg, hip, capsys):
    """M = 8192 rows (B=64 x L=128), 2 layers, fp32-grade mode: every parameter gradient of the BLOSUM loss against
    CPU autograd of the oracle at config 4's per-rank shapes."""
    from test_backward_gpu import _grad_compare
    B, L, T = 64, 128, 50
    model, sd = seq_trainer(L, 2, seed=53, T=T)
    pk, t_int, noised = _sequence_batch(B, L, seed=54, T=T)
    dpk = to_dev(pk)
    with pkg.ops.arithmetic("bf16x6", respect_env=False):
        loss = model.get_loss(dpk, (t_int / T).to(DEV), noised.to(DEV))[0]

        def ref_loss(rsd):
            pred = oseq.forward(rsd, {"num_heads": 12, "max_pos": L}, t_int / T, noised, pk["ligand_angles"],
                                pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
            return oseq.get_loss(pred, pk, noised)[0]

        worst = _grad_compare(model, sd, loss, ref_loss, 5e-5)
    with capsys.disabled():
        print(f"\n[config 4 shapes, M=8192, 2 layers, bf16x6] worst per-parameter gradient error vs oracle {worst:.2e}")


# ------------------------------------------------------------------------------------ config 5
def test_config5_joint_chain_128_pockets_x128(pkg, hip, capsys):
    """BASELINE configs[4], one GPU's share: 128 pockets x L=128 through structure sampling (full 12+12 model) ->
    device hand-over -> sequence sampling (6 layers), short chains.  The first 2 pockets of the batch of 128 against
    the same 2 run alone (other GEMM / attention launch geometry), the small structure run against the oracle step by
    step, the first sequence forward against the oracle, and the deterministic (argmax) sequences identical."""
    from test_structure_gpu import build as build_struct
    from test_sequence_gpu import build as build_seq
    from e3diff_amd.structure_model.sample import p_sample, p_sample_loop
    from e3diff_amd.structure_model.utils import CosineTables
    from e3diff_amd.sequence_model import sample as QS
    from e3diff_amd.sequence_model.sample_by_generated_angles import angles_from_trajectory, denoise
    from e3diff_amd.sequence_model.utils import DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete
    B, L, NS, TQ = 128, 128, 2, 4
    smodel, ssd = build_struct(pkg, FULL_STRUCT, L, seed=61)
    qmodel, qsd = build_seq(pkg, FULL_SEQ, L, seed=62, wrapper=True)
    pk = synthetic_pockets(B, L, seed=63, with_ligand_seq=True)
    d = to_dev(pk)
    g = gen(64)
    x_T = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g))
    order = [668, 334, 0]                    # 3 steps of the T=1000 schedule (stride 334): moderate amplification
    noises = torch.randn(len(order), B, L, 8, generator=g)
    tab = CosineTables(1000)
    betas = ostr.cosine_beta_schedule(1000)

    def chain(sl):
        return p_sample_loop(smodel, d["ligand_attn_mask"][sl], x_T[sl].to(DEV), d["receptor_seq"][sl],
                             d["receptor_attn_mask"][sl], d["receptor_angles"][sl], 1000, tab, disable_pbar=True,
                             noises=noises[:, sl].to(DEV), return_device=True, step=334)

    full, small = chain(slice(0, B)), chain(slice(0, NS))
    assert full.shape == (3, B, L, 8) and torch.isfinite(full).all()
    valid = d["ligand_attn_mask"][:NS].bool()[None, :, :, None].expand(3, NS, L, 8)
    dd = ostr.modulo_with_wrapped_range((full[:, :NS] - small)[valid]).abs().max().item()
    assert dd < 1e-4, dd
    # the small run against the oracle, step by step from the oracle's previous state
    fn = lambda t, x, lm, rs, ra, rm: ostr.forward(ssd, {"num_heads": 12, "max_pos": L}, t, x, lm, rs, ra, rm)  # noqa: E731
    prev = x_T[:NS]
    for n, t in enumerate(order):
        want = ostr.modulo_with_wrapped_range(ostr.p_sample(
            fn, pk["ligand_attn_mask"][:NS], prev, pk["receptor_seq"][:NS], pk["receptor_attn_mask"][:NS],
            pk["receptor_angles"][:NS], torch.full((NS,), t), betas, noise=noises[n, :NS] if t > 0 else None))
        got = p_sample(smodel, d["ligand_attn_mask"][:NS], prev.to(DEV), d["receptor_seq"][:NS], d["receptor_attn_mask"][:NS],
                       d["receptor_angles"][:NS], torch.full((NS,), t, device=DEV), betas,
                       noise=noises[n, :NS].to(DEV), wrap=True).cpu()
        err = ostr.modulo_with_wrapped_range(got - want).abs().max().item()
        assert err < reverse_step_tolerance(betas, t, eps_scale=4.0, rel=TOL), (t, err)
        prev = want
    # hand-over on the device, then the sequence chain (deterministic argmax path, uniform transition)
    angles = angles_from_trajectory(full, d["ligand_attn_mask"])
    assert float((angles * (1 - d["ligand_attn_mask"])[..., None]).abs().sum()) == 0.0
    sched = PredefinedNoiseScheduleDiscrete("cosine", TQ).to(DEV)
    torch.manual_seed(7)
    xq = QS.generate_discrete_noise(B, L, 20, DEV)
    batch = dict(pk, structure_ids=None)
    ids, true_s, pred_full, rec = denoise(batch, angles, qmodel, sched, DiscreteUniformTransition(20), False, x_T=xq,
                                          timesteps=TQ)
    sub = {k: (v[:NS] if torch.is_tensor(v) else v) for k, v in batch.items()}
    _, _, pred_small, _ = denoise(sub, angles[:NS], qmodel, sched, DiscreteUniformTransition(20), False, x_T=xq[:NS],
                                  timesteps=TQ)
    assert len(pred_full) == B and pred_full[:NS] == pred_small
    # first sequence forward of the chain against the oracle on the 2 items
    s0 = torch.full((NS, 1), float(TQ - 1))
    with torch.no_grad():
        logits = qmodel.forward(s0.to(DEV), xq[:NS], angles[:NS], d["ligand_attn_mask"][:NS], d["receptor_seq"][:NS],
                                d["receptor_angles"][:NS], d["receptor_attn_mask"][:NS])
    want = oseq.forward(qsd, {"num_heads": 12, "max_pos": L}, s0, xq[:NS].cpu(), angles[:NS].cpu(), pk["ligand_attn_mask"][:NS],
                        pk["receptor_seq"][:NS], pk["receptor_angles"][:NS], pk["receptor_attn_mask"][:NS])
    assert rel_err(logits, want) < TOL


# ------------------------------------------------------------------------------------ arithmetic margins
def _oracle_fp64(sd, cfg, t, x_t, pk):
    """The oracle in fp64 -- except the Fourier features, whose fp32 argument rounding (t * W * 2 pi at ~1e5 rad) is part
    of the reference arithmetic -- as the ground truth the arithmetics are ranked against."""
    sd64 = {k: v.double() for k, v in sd.items()}
    orig = ostr.fourier_projection
    ostr.fourier_projection = lambda s_, p_, tt: orig(sd, p_, tt).double()
    try:
        return ostr.forward(sd64, cfg, t, x_t.double(), pk["ligand_attn_mask"].double(), pk["receptor_seq"].double(),
                            pk["receptor_angles"].double(), pk["receptor_attn_mask"].double())
    finally:
        ostr.fourier_projection = orig


@pytest.mark.parametrize("regime,scale", [("random-init", 1.0), ("weights x2, gamma 0.5-2", 2.0), ("weights x4, gamma 0.5-2", 4.0),
                                          ("heavy-tailed rows, outlier entries and LayerNorm channels", -1.0)])
def test_bf16x3_margin_elementwise_at_L256_full_depth(pkg, hip, regime, scale, capsys):
    """Every arithmetic's distance from the 1e-4 contract at the bench's sequence length, 12+12 layers, in three
    weight regimes, reported element-wise (|d| / max(|ref|, 1e-3 rms): 99.9th percentile and max) beside the max-norm
    figure the other tests assert on.

    The scaled regimes sharpen the softmax rows (attention logits x4 / x16) and are ill-conditioned for ANY fp32
    implementation: the CPU oracle in fp32 is itself ~3e-5..8e-5 (x2) / ~4e-4..7e-4 (x4, depending on the thread count's
    summation order) away from the same oracle in fp64 (tools/lab/margin_cpu.py).  They are ranked against the fp64 oracle.
    x2 and the heavy-tailed regime are asserted ABSOLUTELY for the inference default (f16x3 <= 1e-4 against the fp32 oracle
    and against fp64); x4 relative to what exact fp32 achieves there (<= 10x) (measured on MI355X in round 3, vs fp64: x2 f32 1.5e-4 -- 5e-5
    since round 4's blocked accumulation --, f16x3 8.0e-5, bf16x6 1.2e-4,
    bf16x3 6.5e-4; x4 f32 8.0e-4, f16x3 5.5e-4, bf16x6 2.6e-3, bf16x3 0.12 -- at x4 the first encoder layer sees
    activations of ~5000 and logits of ~1e7, where a 2^-17 product error is an absolute logit error of ~100).  (Round 2 reported 0.47 for every split arithmetic at x4 and called it
    ill-conditioning; it was a defect -- the attention kernels skipped all-padding key tiles although the first encoder
    layer's |q|, |k| ~ 7000 put padded keys above the -10000 mask, tools/lab/margin_bisect.py -- fixed by the
    element-bound guard of e3d_relkey_attn_fwd_split_ex.)"""
    from test_structure_gpu import build
    L, B = 256, 2
    model, sd = build(pkg, FULL_STRUCT, L, seed=71)
    if scale < 0:
        sd = heavy_tailed_state_dict(sd, seed=75)
        model.load_state_dict(sd)
    elif scale != 1.0:
        sd = rescaled_state_dict(sd, scale, (0.5, 2.0), seed=72)
        model.load_state_dict(sd)
    pk = synthetic_pockets(B, L, seed=73, lig_range=(180, 256), rec_range=(150, 256))
    d = to_dev(pk)
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=gen(74)))
    t = torch.tensor([999, 3])
    cfg = {"num_heads": 12, "max_pos": L}
    want = ostr.forward(sd, cfg, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"],
                        pk["receptor_attn_mask"])
    want64 = _oracle_fp64(sd, cfg, t, x_t, pk)
    m = pk["ligand_attn_mask"].bool()
    rows = {"cpu-f32": (rel_err(want[m], want64[m]),) * 2 + elementwise_err(want[m], want64[m])}
    for mode in ("bf16x3", "f16x3", "bf16x6", "f32"):
        with pkg.ops.arithmetic(mode, respect_env=False), torch.no_grad():
            got = model(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"],
                        d["receptor_attn_mask"]).cpu()
        assert torch.isfinite(got).all()
        rows[mode] = (rel_err(got[m], want[m]), rel_err(got[m], want64[m])) + elementwise_err(got[m], want64[m])
    with capsys.disabled():
        for mode, (mx, mx64, p999, emax) in rows.items():
            print(f"\n[margin, {regime}, L=256 12+12, {mode}] max-norm vs fp32 oracle {mx:.2e}, vs fp64 oracle {mx64:.2e} | "
                  f"element-wise (fp64) p99.9 {p999:.2e} max {emax:.2e}")
    cpu = rows["cpu-f32"][1]                           # the CPU oracle in fp32 against the same oracle in fp64
    f32 = max(rows["f32"][1], cpu)                     # what an exact-fp32 implementation achieves here (GPU or CPU)
    # the exact-fp32 kernels are fp32-grade in EVERY regime: within 2x of the CPU's own fp32 (round 4: blocked accumulation
    # in gemm_nt_f32 -- with one serial chain over K the kernel sat at 4.4x the CPU figure in the x2 regime)
    assert rows["f32"][1] < 2 * cpu + 1e-6, rows
    if scale == 1.0:
        assert rows["bf16x3"][0] < TOL and rows["bf16x6"][0] < TOL and rows["f32"][0] < TOL and rows["f16x3"][0] < TOL
        # element-wise, floor 1e-3 rms: outputs 1000x below the rms carry the same ABSOLUTE error as the large ones, so
        # the percentile sits ~10-20x above the max-norm figure in every arithmetic (fp32 itself: 4e-5 vs 3e-6)
        assert rows["bf16x3"][2] < 2e-3 and rows["bf16x6"][2] < 2e-4
        assert rows["f16x3"][1] < 3 * f32 and rows["bf16x6"][1] < 3 * f32
    elif scale == 2.0:
        # ABSOLUTE (the north star's "within 1e-4 relative fp32"): the inference default against the fp32 oracle and
        # against the fp64 truth; bf16x6 (3 terms of 8 bits: the large first-layer activations cost it more) within 2e-4
        assert rows["f16x3"][0] <= TOL and rows["f16x3"][1] <= TOL and rows["f32"][0] <= TOL, rows
        assert rows["bf16x6"][1] < 2e-4, rows
        assert rows["bf16x3"][1] < 30 * f32, rows
    elif scale < 0:
        # the regime shaped like TRAINED weights (per-row scales, outlier entries, outlier LayerNorm channels): every
        # fp32-grade arithmetic within the contract, absolutely
        assert rows["f16x3"][0] <= TOL and rows["bf16x6"][0] <= TOL and rows["f32"][0] <= TOL, rows
        assert rows["f16x3"][1] < 4 * f32 and rows["bf16x6"][1] < 4 * f32, rows
    else:
        assert rows["f16x3"][1] < 10 * f32 and rows["bf16x6"][1] < 10 * f32, rows


@pytest.mark.parametrize("factor", [1e-2, 1e-4])
def test_f16x3_small_weight_regime(pkg, hip, factor):
    """ADVICE r02 / VERDICT r02 item 2: the regime random-init tests never reach -- small but non-zero Linear weights
    (every projection, the FFNs and BOTH adaLN modulation layers x ``factor``; LayerNorms keep the activations O(1)).
    Before the power-of-two pre-scaling of the f16x3 weights the low fp16 terms of such weights sat in the subnormals
    (absolute floor 2^-25 per element: ~2e-2 relative for |w| ~ 1e-6).  Against the fp32 CPU oracle, L = 128, 4 + 4
    layers; the exact-f32 kernels beside it."""
    from test_structure_gpu import build
    L, B = 128, 2
    cfg4 = dict(FULL_STRUCT, num_hidden_layers=4)
    model, sd = build(pkg, cfg4, L, seed=81)
    sd = {k: (v * factor if (v.dim() == 2 and k.endswith(".weight") and "distance_embedding" not in k) else v.clone())
          for k, v in sd.items()}
    model.load_state_dict(sd)
    pk = synthetic_pockets(B, L, seed=83, lig_range=(20, 128), rec_range=(40, 128))
    d = to_dev(pk)
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=gen(84)))
    t = torch.tensor([500, 7])
    want = ostr.forward(sd, {"num_heads": 12, "max_pos": L}, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"],
                        pk["receptor_angles"], pk["receptor_attn_mask"])
    m = pk["ligand_attn_mask"].bool()
    errs = {}
    for mode in ("f16x3", "f32"):
        with pkg.ops.arithmetic(mode, respect_env=False), torch.no_grad():
            got = model(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"],
                        d["receptor_attn_mask"]).cpu()
        errs[mode] = rel_err(got[m], want[m])
    assert errs["f32"] < 1e-5 and errs["f16x3"] < 1e-5, errs
