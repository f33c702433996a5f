"""GPU parity of the structure model's denoising path (HIP, via the C-ABI) against (1) the
reference-generated golden fixtures and (2) the CPU oracle on the same seeded inputs.
Tolerance: 1e-4 relative fp32 on predicted noise/angles (BASELINE.json north_star)."""
import os

import pytest
import torch

from helpers import (FULL_STRUCT, GOLDEN, rel_err, reverse_step_tolerance, seeded_state_dict,
                     synthetic_pockets)
from oracle import structure as ostr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
TOL = 1e-4


def build(pkg, cfg, L, seed, zero_relkey=False, relkey=True):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    common = dict(hidden_size=cfg["hidden_size"], num_attention_heads=cfg["num_heads"],
                  intermediate_size=cfg["intermediate_size"], num_hidden_layers=cfg["num_hidden_layers"],
                  max_position_embeddings=L,
                  position_embedding_type="relative_key" if relkey else "absolute")
    model = ConditionalBertForDiffusionBase(BertConfig(**common),
                                            BertConfig(**common, is_decoder=True, add_cross_attention=True), 8)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    sd = seeded_state_dict(shapes, seed=seed, zero_relkey=zero_relkey)
    model.load_state_dict(sd, strict=True)
    return model.eval().to(DEV), sd


def to_dev(d):
    return {k: v.to(DEV) for k, v in d.items() if torch.is_tensor(v)}


def test_forward_matches_reference_golden(pkg, hip):
    """The fixture was produced by the reference itself (transformers 5.15 => no relative_key
    parameters): load exactly its state_dict into the product built without the rel-key term."""
    fx = torch.load(os.path.join(GOLDEN, "structure_forward_tiny.pt"), weights_only=False)
    cfg = dict(fx["cfg"])
    model, sd = build(pkg, cfg, cfg["max_seq_len"], fx["seed"], relkey=False)
    assert set(sd) == set(fx["shapes"])           # checkpoint keys == the reference's
    model.load_state_dict(seeded_state_dict(fx["shapes"], seed=fx["seed"]), strict=True)
    pk = to_dev(fx["pockets"])
    for tag, (t, want) in fx["outs"].items():
        got = model(t.to(DEV), fx["x_t"].to(DEV), pk["ligand_attn_mask"], pk["receptor_seq"],
                    pk["receptor_angles"], pk["receptor_attn_mask"])
        assert rel_err(got, want) < TOL, tag


def teacher_forced_steps(model, pk, x_T, traj, noises, betas, T):
    """Each reverse step restarted from the expected previous state (see
    helpers.reverse_step_tolerance for why trajectories are not compared free-running)."""
    from e3diff_amd.structure_model.sample import p_sample
    prev = x_T
    for n, t in enumerate(reversed(range(T))):
        got = p_sample(model, pk["ligand_attn_mask"], prev.to(DEV), pk["receptor_seq"], pk["receptor_attn_mask"],
                       pk["receptor_angles"], torch.full((prev.shape[0],), t, device=DEV), betas,
                       noise=noises[n].to(DEV), wrap=True).cpu()
        err = ostr.modulo_with_wrapped_range(got - traj[n]).abs().max().item()
        assert err < reverse_step_tolerance(betas, t, eps_scale=4.0, rel=TOL), (t, err)
        prev = traj[n]


def test_sampler_matches_reference_golden(pkg, hip):
    from e3diff_amd.structure_model.sample import p_sample_loop
    fx = torch.load(os.path.join(GOLDEN, "structure_forward_tiny.pt"), weights_only=False)
    sx = torch.load(os.path.join(GOLDEN, "structure_sampler_tiny.pt"), weights_only=False)
    cfg = dict(fx["cfg"])
    model, _ = build(pkg, cfg, cfg["max_seq_len"], fx["seed"], relkey=False)
    model.load_state_dict(seeded_state_dict(fx["shapes"], seed=fx["seed"]), strict=True)
    pk = to_dev(fx["pockets"])
    betas = ostr.cosine_beta_schedule(sx["T"])
    teacher_forced_steps(model, pk, sx["x_T"], sx["traj"], sx["noises"], betas, sx["T"])
    traj = p_sample_loop(model, pk["ligand_attn_mask"], sx["x_T"].to(DEV), pk["receptor_seq"],
                         pk["receptor_attn_mask"], pk["receptor_angles"], sx["T"], betas,
                         disable_pbar=True, noises=sx["noises"].to(DEV), step=1)
    assert traj.shape == sx["traj"].shape and traj.device.type == "cpu"
    # free-running: first step exact to the step tolerance, all steps wrapped
    assert ostr.modulo_with_wrapped_range(traj[0] - sx["traj"][0]).abs().max().item() \
        < reverse_step_tolerance(betas, sx["T"] - 1, 4.0, TOL)
    assert traj.min() >= -3.1416 and traj.max() <= 3.1416


@pytest.mark.parametrize("layers,B,L", [(2, 3, 64), (12, 2, 128), (1, 2, 256), (2, 5, 50)])
def test_forward_matches_oracle_full_width_with_relkey(pkg, hip, layers, B, L):
    cfg = dict(FULL_STRUCT, num_hidden_layers=layers)
    model, sd = build(pkg, cfg, max(L, 64), seed=100 + layers)
    ocfg = {"num_heads": 12, "max_pos": max(L, 64)}
    pk = synthetic_pockets(B, L, seed=L)
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=torch.Generator().manual_seed(1)))
    for t in (torch.full((B,), 999), torch.randint(0, 1000, (B, 1), generator=torch.Generator().manual_seed(2))):
        want = ostr.forward(sd, ocfg, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"],
                            pk["receptor_attn_mask"])
        d = to_dev(pk)
        got = model(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"],
                    d["receptor_attn_mask"])
        assert got.shape == (B, L, 8)
        assert rel_err(got, want) < TOL


def test_gemm_arithmetic_modes_end_to_end(pkg, hip, capsys):
    """Full 12+12-layer model, L=128: exact fp32 MFMA vs bf16-split GEMM + attention modes, all against the
    oracle.  bf16x6 is fp32-grade; bf16x3 (the default) must stay inside the 1e-4 budget with margin."""
    cfg = dict(FULL_STRUCT, num_hidden_layers=12)
    B, L = 2, 128
    model, sd = build(pkg, cfg, L, seed=42)
    pk = synthetic_pockets(B, L, seed=11)
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=torch.Generator().manual_seed(1)))
    t = torch.full((B,), 321)
    want = ostr.forward(sd, {"num_heads": 12, "max_pos": L}, t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"],
                        pk["receptor_angles"], pk["receptor_attn_mask"])
    d = to_dev(pk)
    errs = {}
    for mode, bound in (("f32", 2e-5), ("bf16x6", 2e-5), ("bf16x3", TOL / 2), ("f16x3", 2e-5)):
        prev, prev_a = pkg.ops.set_gemm_mode(mode), pkg.ops.set_attn_mode(mode)
        try:
            got = model(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"],
                        d["receptor_attn_mask"])
        finally:
            pkg.ops.set_gemm_mode(prev)
            pkg.ops.set_attn_mode(prev_a)
        errs[mode] = rel_err(got, want)
        assert errs[mode] < bound, (mode, errs)
    with capsys.disabled():
        print(f"\n[gemm modes, 12+12 layers, L=128] rel err vs oracle: {errs}")


def test_cached_receptor_path_is_identical(pkg, hip):
    """encode_receptor once + decode == forward (the sampler's F5 restructuring changes nothing)."""
    cfg = dict(FULL_STRUCT, num_hidden_layers=2)
    model, _ = build(pkg, cfg, 64, seed=5)
    d = to_dev(synthetic_pockets(4, 64, seed=9))
    x_t = torch.randn(4, 64, 8, device=DEV)
    t = torch.full((4,), 17, device=DEV)
    a = model(t, x_t, d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
    rec = model.encode_receptor(d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
    b = model.decode(t, x_t, d["ligand_attn_mask"], rec)
    assert torch.equal(a, b)


def test_timestep_modulation_rows_match_the_per_step_path(pkg, hip):
    """The samplers hoist what depends on the timestep alone (Fourier features -> adaLN modulation) out of the loop:
    ``decode(mod=row)`` against ``decode`` computing it itself -- one shared [1,6H] row, per-item rows, and a row of
    the table p_sample_loop builds for a whole chain (a batched GEMM: other summation order, fp32 rounding apart)."""
    cfg = dict(FULL_STRUCT, num_hidden_layers=2)
    model, _ = build(pkg, cfg, 64, seed=5)
    with torch.no_grad():                                   # the modulation MLP is zero-initialised (model.py:50-51)
        model.timestep_emb.adaLN_modulation[0].weight.normal_(0, 0.05)
        model.timestep_emb.adaLN_modulation[0].bias.normal_(0, 0.05)
    d = to_dev(synthetic_pockets(3, 64, seed=9))
    x_t = torch.randn(3, 64, 8, device=DEV)
    rec = model.encode_receptor(d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
    t = torch.full((3,), 977, device=DEV)
    want = model.decode(t, x_t, d["ligand_attn_mask"], rec)
    per_item = model.timestep_modulation(t)                                     # [3,6H]
    assert torch.equal(model.decode(t, x_t, d["ligand_attn_mask"], rec, mod=per_item), want)
    assert rel_err(model.decode(t, x_t, d["ligand_attn_mask"], rec, mod=per_item[:1].contiguous()), want) < 1e-6
    table = model.timestep_modulation(torch.arange(1000, device=DEV))          # the chain's table: M = 1000 rows
    assert rel_err(table[977:978], per_item[:1]) < 2e-6
    assert rel_err(model.decode(t, x_t, d["ligand_attn_mask"], rec, mod=table[977:978]), want) < 2e-6


def test_sampling_loop_matches_oracle_and_properties(pkg, hip):
    from e3diff_amd.structure_model.sample import p_sample, p_sample_loop
    cfg = dict(FULL_STRUCT, num_hidden_layers=2)
    L, B, T = 64, 2, 5
    model, sd = build(pkg, cfg, L, seed=7)
    ocfg = {"num_heads": 12, "max_pos": L}
    pk = synthetic_pockets(B, L, seed=4)
    gen = torch.Generator().manual_seed(3)
    x_T = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=gen))
    noises = torch.randn(T, B, L, 8, generator=gen)
    betas = ostr.cosine_beta_schedule(T)
    fn = lambda t, x, lm, rs, ra, rm: ostr.forward(sd, ocfg, t, x, lm, rs, ra, rm)  # noqa: E731
    want = ostr.p_sample_loop(fn, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"],
                              pk["receptor_angles"], T, betas, noises=noises)
    d = to_dev(pk)
    teacher_forced_steps(model, d, x_T, want, noises, betas, T)
    got = p_sample_loop(model, d["ligand_attn_mask"], x_T.to(DEV), d["receptor_seq"], d["receptor_attn_mask"],
                        d["receptor_angles"], T, betas, disable_pbar=True, noises=noises.to(DEV), step=1)
    assert got.shape == want.shape
    assert got.min() >= -3.1416 and got.max() <= 3.1416            # every step is wrapped
    # single un-wrapped step, tensor timestep, own RNG at t == 0 (no noise)
    one = p_sample(model, d["ligand_attn_mask"], x_T.to(DEV), d["receptor_seq"], d["receptor_attn_mask"],
                   d["receptor_angles"], torch.zeros(B, dtype=torch.long, device=DEV), betas)
    want1 = ostr.p_sample(fn, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"],
                          pk["receptor_angles"], torch.zeros(B, dtype=torch.long), betas)
    assert rel_err(one, want1) < TOL
    with pytest.raises(AssertionError, match="multiple values"):
        p_sample(model, d["ligand_attn_mask"], x_T.to(DEV), d["receptor_seq"], d["receptor_attn_mask"],
                 d["receptor_angles"], torch.tensor([1, 2], device=DEV), betas)


def test_padding_keys_do_not_leak(pkg, hip):
    """Changing receptor/ligand values at padded positions must not change real-position outputs."""
    cfg = dict(FULL_STRUCT, num_hidden_layers=1)
    model, _ = build(pkg, cfg, 64, seed=8)
    pk = synthetic_pockets(2, 64, seed=6)
    d = to_dev(pk)
    x_t = torch.randn(2, 64, 8, device=DEV)
    t = torch.full((2,), 3, device=DEV)
    a = model(t, x_t, d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
    ra = d["receptor_angles"] + (1 - d["receptor_attn_mask"])[..., None] * 5.0
    xt2 = x_t + (1 - d["ligand_attn_mask"])[..., None] * 5.0
    b = model(t, xt2, d["ligand_attn_mask"], d["receptor_seq"], ra, d["receptor_attn_mask"])
    m = d["ligand_attn_mask"].bool()
    assert rel_err(a[m], b[m]) < 1e-5


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


def test_hip_graph_replay_of_the_reverse_step_is_bit_identical(pkg, hip):
    """structure_model/sample.py::GraphedReverseStep (one captured HIP graph replayed per step, the
    default for launch-bound small batches) against the eager launch sequence, same injected noise."""
    from e3diff_amd.structure_model import sample as S
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    B, L, T = 2, 64, 12
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2,
             max_position_embeddings=L, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
              feature_names=list("abcdefgh"), loss_func=[M.diheral_loss_func] * 8).eval().to("cuda:0")
    pk = {k: v.to("cuda:0") for k, v in synthetic_pockets(B, L, seed=3).items() if torch.is_tensor(v)}
    tab = CosineTables(T)
    g = torch.Generator().manual_seed(5)
    x_T = modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g)).to("cuda:0")
    noises = torch.randn(T, B, L, 8, generator=g).to("cuda:0")
    args = (model, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"], pk["receptor_angles"], T, tab)
    eager = S.p_sample_loop(*args, noises=noises, return_device=True, step=1, use_graph=False)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter("error")           # a silent fall-back to eager launches would make this test vacuous
        graph = S.p_sample_loop(*args, noises=noises, return_device=True, step=1, use_graph=True)
    assert torch.equal(eager, graph)
    # and the self-drawing graph (production path) produces a finite, wrapped chain
    free = S.p_sample_loop(*args, return_device=True, step=1, use_graph=True)
    assert torch.isfinite(free).all() and free.abs().max() <= 3.1416


def test_trimmed_sampling_chain_equals_the_padded_one_on_valid_positions(pkg, hip):
    """p_sample_loop(trim_padding=True): the chain on the rows up to the longest ligand / pocket of the batch
    (32-row granularity) against the full padded frame, same injected noise."""
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model import sample as S
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    B, L, T = 3, 128, 6
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2,
             max_position_embeddings=L, hidden_dropout_prob=0.1, attention_probs_dropout_prob=0.1)
    torch.manual_seed(0)
    model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
              feature_names=list("abcdefgh"), loss_func=[M.diheral_loss_func] * 8).eval().to("cuda:0")
    pk = {k: v.to("cuda:0") for k, v in synthetic_pockets(B, L, seed=3, rec_range=(20, 70)).items() if torch.is_tensor(v)}
    assert S.trimmed_length(pk["ligand_attn_mask"]) == 32 and S.trimmed_length(pk["receptor_attn_mask"]) < L
    g = torch.Generator().manual_seed(5)
    x_T = modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g)).to("cuda:0")
    noises = torch.randn(T, B, L, 8, generator=g).to("cuda:0")
    args = (model, pk["ligand_attn_mask"], x_T, pk["receptor_seq"], pk["receptor_attn_mask"], pk["receptor_angles"], T,
            CosineTables(T))
    full = S.p_sample_loop(*args, noises=noises, return_device=True, step=1)
    trim = S.p_sample_loop(*args, noises=noises, return_device=True, step=1, trim_padding=True)
    valid = pk["ligand_attn_mask"].bool()[None, :, :, None].expand_as(full)
    # the same mathematics on the same rows through different kernel variants by shape (the 32-row frame runs the
    # per-wave fp32-grade attention kernel, the 128-row frame the cooperative one): fp32-level rounding differences
    # (~3e-6 on eps) through 6 steps of the amplifying reverse chain (x100 at t = T-1 of a T = 6 schedule)
    d = modulo_with_wrapped_range((full - trim)[valid])
    assert d.abs().max() < 6e-4, d.abs().max()
    assert (trim[:, :, 32:] == 0).all()
    # a mask that is not a prefix keeps the full frame
    holes = pk["ligand_attn_mask"].clone()
    holes[0, 100] = 1.0
    assert S.trimmed_length(holes) == L


def test_full_bench_size_properties(pkg, hip):
    """BASELINE configs[2] size -- 256 x 256-residue pockets, 12+12 layers x 768: the shape bench.py times, served by
    the persistent 256x256 GEMM and the 8-wave cooperative attention kernel.  Too large for the CPU oracle as a
    whole, so: (1) the first items of the full batch against the same items run alone (general GEMM kernel, other
    launch geometry) AND that small run against the oracle; (2) padding invariance; (3) dense vs padding-skip key
    sweep bit-identical; (4) the sampler's cached-encoder path bit-identical."""
    B, L, NS = 256, 256, 2
    model, sd = build(pkg, FULL_STRUCT, L, seed=21)
    pk = synthetic_pockets(B, L, seed=77)
    d = to_dev(pk)
    g = torch.Generator().manual_seed(5)
    x_t = ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=g))
    t = torch.full((B,), 731)
    args = (d["ligand_attn_mask"], d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
    with torch.no_grad():
        full = model(t.to(DEV), x_t.to(DEV), *args)
        assert full.shape == (B, L, 8) and torch.isfinite(full).all()
        # (1) cross-geometry agreement and oracle parity on the first NS items
        small = model(t[:NS].to(DEV), x_t[:NS].to(DEV), *(a[:NS].contiguous() for a in args))
        assert rel_err(full[:NS], small) < 2e-5
        want = ostr.forward(sd, {"num_heads": 12, "max_pos": L}, t[:NS], x_t[:NS], pk["ligand_attn_mask"][:NS],
                            pk["receptor_seq"][:NS], pk["receptor_angles"][:NS], pk["receptor_attn_mask"][:NS])
        assert rel_err(small, want) < TOL and rel_err(full[:NS], want) < TOL
        # (2) values at padded positions never reach real positions
        ra = d["receptor_angles"] + (1 - d["receptor_attn_mask"])[..., None] * 3.0
        xt2 = x_t.to(DEV) + (1 - d["ligand_attn_mask"])[..., None] * 3.0
        moved = model(t.to(DEV), xt2, d["ligand_attn_mask"], d["receptor_seq"], ra, d["receptor_attn_mask"])
        m = d["ligand_attn_mask"].bool()
        assert rel_err(moved[m], full[m]) < 1e-5
        # (3) the attention sweep that stops after the last valid key tile is the dense sweep, bit for bit
        lib = pkg.hip.lib()
        prev = lib.e3d_attn_skip_padded_tiles(0)
        try:
            dense = model(t.to(DEV), x_t.to(DEV), *args)
        finally:
            lib.e3d_attn_skip_padded_tiles(prev)
        assert torch.equal(dense, full)
        # (4) pocket encoder + cross K/V computed once (the sampler's loop)
        cache = model.encode_receptor(d["receptor_seq"], d["receptor_angles"], d["receptor_attn_mask"])
        assert torch.equal(model.decode(t.to(DEV), x_t.to(DEV), d["ligand_attn_mask"], cache), full)
