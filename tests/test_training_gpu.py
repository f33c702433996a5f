"""GPU: the training entry points end to end on synthetic BioLiP-shaped records (reduced model
width so that it runs in seconds): losses are finite and go down, weights move, checkpoints are
written in the reference's state_dict format and reload strictly."""
import os

import pytest
import torch

from helpers import GOLDEN

pytestmark = pytest.mark.gpu


def _records(n=48):
    import sys
    sys.path.insert(0, GOLDEN)
    fx = torch.load(os.path.join(GOLDEN, "structure_dataset.pt"), weights_only=False)
    recs = fx["records"]
    return [dict(recs[i % len(recs)], structure_ids=dict(recs[i % len(recs)]["structure_ids"], pdb_id=f"s{i:03d}"))
            for i in range(n)]


SMALL = dict(hidden_size=256, num_heads=4, intermediate_size=512, num_hidden_layers=1, max_seq_len=64,
             batch_size=8, max_epochs=6, min_epochs=1, dropout_p=0.0, lr=2e-3, lr_scheduler=None, pocket_ext=1)


@pytest.mark.parametrize("dropout_p", [0.0, 0.1])   # 0.1 = the reference's training value (train_model.py:24)
def test_structure_train_model_runs_and_learns(pkg, hip, tmp_path, monkeypatch, dropout_p):
    from e3diff_amd.structure_model import train_model as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(T, "NUM_THREAD", 0)
    monkeypatch.setattr(T, "CONFIG", dict(T.CONFIG, **dict(SMALL, dropout_p=dropout_p), timesteps=100))
    torch.manual_seed(0)
    train_dl, val_dl = T.get_dataloader(None, records=_records())
    enc, dec = T.build_configs()
    history, model = T.train_model(enc, dec, train_dl, val_dl)
    assert history["steps"] == 6 * len(train_dl)
    assert all(torch.isfinite(torch.tensor(history["train_loss"]))) and all(torch.isfinite(torch.tensor(history["val_loss"])))
    assert history["train_loss"][-1] < history["train_loss"][0]
    assert os.path.exists(tmp_path / "best_val_model.pt")
    sd = torch.load(tmp_path / "best_val_model.pt", weights_only=True)
    fresh = type(model)(enc, dec, feature_names=model.feature_names, loss_func=model.loss_func)
    fresh.load_state_dict(sd, strict=True)
    assert any(k.endswith("attention.self.distance_embedding.weight") for k in sd)


def test_sequence_train_model_runs_and_learns(pkg, hip, tmp_path, monkeypatch):
    from e3diff_amd.sequence_model import train_model as T
    monkeypatch.chdir(tmp_path)
    monkeypatch.setattr(T, "NUM_THREAD", 0)
    monkeypatch.setattr(T, "CONFIG", dict(T.CONFIG, **dict(SMALL, dropout_p=0.1), timesteps=50))
    torch.manual_seed(0)
    train_dl, val_dl = T.get_dataloader(None, records=_records())
    enc, dec = T.build_configs()
    history, model = T.train_model(enc, dec, train_dl, val_dl)
    assert all(torch.isfinite(torch.tensor(history["train_loss"])))
    assert history["train_loss"][-1] < history["train_loss"][0]
    # parameters the forward never uses received no update (and would not desynchronise DDP buckets)
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("receptor_feature_emb."))


def test_on_device_forward_noising_matches_dataset_path(pkg, hip):
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    tab = CosineTables(100)
    x0 = modulo_with_wrapped_range(torch.randn(5, 32, 8) * 2)
    noise = modulo_with_wrapped_range(torch.randn(5, 32, 8))
    t = torch.tensor([[0], [7], [50], [98], [99]])
    out = noise_batch_on_device(x0.cuda(), tab, timestep=t.cuda(), noise=noise.cuda())
    want = torch.stack([modulo_with_wrapped_range(tab.sqrt_alphas_cumprod[int(t[i])] * x0[i]
                                                  + tab.sqrt_one_minus_alphas_cumprod[int(t[i])] * noise[i])
                        for i in range(5)])
    assert modulo_with_wrapped_range(out["noised_ligand_angle"].cpu() - want).abs().max() < 2e-6
    auto = noise_batch_on_device(x0.cuda(), tab)
    assert auto["timestep"].shape == (5, 1) and auto["known_noise"].abs().max() <= 3.1416


def test_bench_two_rank_rehearsal_reports_the_data_parallel_training_leg():
    """`python bench.py --gpus 2` with E3D_BENCH_REHEARSAL=1: both ranks on cuda:0, gloo for the collectives -- the control
    path of the multi-GPU line on a one-GPU box (never a measurement).  Rank 0's single JSON line must carry the headline
    keys AND the ``train_ddp`` extra key (BASELINE config 4, the sequence model's DDP step with the
    overlapped GradientAverager), and the ranks must still hold identical weights after the timed steps (asserted inside
    tools/bench_train.py)."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["E3D_BENCH_REHEARSAL"] = "1"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--batch", "4", "--seq-len", "64", "--only-default-mode", "--train-ddp-batch", "4",
                        "--train-ddp-seq-len", "64", "--train-ddp-layers", "2"],
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["scaling"] == "weak" and out["value"] > 0
    ddp = out["train_ddp"]
    assert ddp["ranks"] == 2 and ddp["global_batch"] == 8 and ddp["backend"] == "gloo" and ddp["ms_per_step"] > 0
    assert ddp["model"] == "sequence" and ddp["buckets"] >= 1 and "train" not in out


def test_two_rank_rehearsal_of_the_data_parallel_step_on_trimmed_frames():
    """Two ranks on cuda:0 over gloo (tools/lab/ddp_trim_check.py under torch.distributed.run): the ranks' batches differ, the
    frame is the maximum over the ranks (training.trim_batch, sharding.max_over_ranks_host), both replay the graph-segment
    step, and their weights are identical afterwards (bench_train's check) -- twice in one process: the second run is the one
    that exposed gloo's unordered device-to-host copy (sharding.GradientAverager._order_for_host_backend)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                        "--master-port", "29641", os.path.join(root, "tools", "lab", "ddp_trim_check.py"), "1", "1"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("run ")]
    assert len(lines) == 2 and all("ok frame=[32, " in ln and "graph=True" in ln for ln in lines), r.stdout[-2000:]


def test_data_parallel_step_runs_on_rccl_with_a_group_of_one():
    """VERDICT r03 "missing 1": an RCCL-executed step.  A builder's box has one GPU, so the process group has one rank over the
    "nccl" backend (= RCCL); E3D_DDP_SINGLE_RANK=1 makes GradientAverager / GraphedDDPStep take their multi-rank path (flat
    buckets, hooks, asynchronous all_reduce calls on RCCL's stream, two graph segments around them).  A group of one averages
    nothing: losses must agree with the single-process replayed step after the same number of steps."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, MASTER_PORT="29537")
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "lab", "rccl_single_rank_step.py"), "2", "3"], capture_output=True,
                       text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["ok"] and out["backend"] == "nccl" and out["rccl_one_rank_graph_segments"]["graph_replay"]
    assert out["rccl_one_rank_graph_segments"]["buckets"] >= 1 and out["rccl_one_rank_eager"]["ranks"] == 1


def test_clip_folded_into_the_fused_adamw_matches_clip_then_step(pkg, hip):
    """training.clip_and_step(fold=True) hands torch's fused AdamW 1 / clip_coef as ``grad_scale``: the parameters after the
    step equal those of clip_grad_norm_ + step (an ulp from g / (1/c) vs g * c), with the clip active and inactive.  (The
    folded form measured slower on MI355X and is not the default: see the function's docstring.)"""
    from e3diff_amd.training import clip_and_step
    adamw = lambda ps, lr, weight_decay: torch.optim.AdamW(ps, lr=lr, weight_decay=weight_decay, fused=True)   # noqa: E731
    for scale in (50.0, 1e-3):           # gradient norm far above / below max_norm = 1
        torch.manual_seed(0)
        base = [torch.randn(257, 33, device="cuda:0"), torch.randn(1000, device="cuda:0")]
        grads = [torch.randn_like(p) * scale for p in base]
        outs = []
        for folded in (False, True):
            ps = [torch.nn.Parameter(p.clone()) for p in base]
            for p, g_ in zip(ps, grads):
                p.grad = g_.clone()
            opt = adamw(ps, lr=1e-2, weight_decay=0.1)
            assert opt.defaults.get("fused")
            for _ in range(2):
                if folded:
                    n = clip_and_step(ps, opt, 1.0, fold=True)
                else:
                    n = torch.nn.utils.clip_grad_norm_(ps, 1.0)
                    opt.step()
            outs.append(([p.detach().clone() for p in ps], float(n)))
        # (the second step sees the gradients the first one left behind: torch's kernel writes the unscaled gradient back,
        #  clip_grad_norm_ scales in place -- the same values up to an ulp)
        assert abs(outs[0][1] - outs[1][1]) <= 1e-5 * outs[0][1]
        for a, b in zip(outs[0][0], outs[1][0]):
            assert torch.allclose(a, b, rtol=1e-6, atol=1e-7)


def test_structure_sample_entry_point_and_its_f16x3_safety_net(pkg, hip, monkeypatch):
    """structure_model/sample.py::sample on a small model: one array [T, l_i, 8] per pocket, finite, in [-pi, pi].  Then
    the one range limit f16x3 has left: a weight pattern that drives an activation past 65504 gives
    non-finite angles in f16x3 -- ``sample`` notices, says so, and returns the bf16x6 result instead."""
    import warnings
    from e3diff_amd.structure_model import sample as S
    from e3diff_amd.structure_model.dataset import LigandBindingSiteDataset, NoisedAnglesDataset
    cfg = dict(S.CONFIG, hidden_size=256, num_heads=4, intermediate_size=512, num_hidden_layers=1, max_seq_len=64, batch_size=4,
               timesteps=6, pocket_ext=1)
    monkeypatch.setattr(S, "CONFIG", cfg)
    monkeypatch.setattr(S, "DEVICE", "cuda:0")
    monkeypatch.setattr(S, "MODEL_PATH", None)
    monkeypatch.delenv("E3D_GEMM_MODE", raising=False)
    ds = NoisedAnglesDataset(LigandBindingSiteDataset(None, "test", cfg["max_seq_len"], cfg["pocket_ext"], records=_records(40)),
                             timesteps=cfg["timesteps"])
    torch.manual_seed(0)
    model = S.load_model(ds)
    with torch.no_grad():       # adaLN_modulation[0] is zero-initialised: give the gated branches weights
        for se in (model.receptor_emb, model.timestep_emb):
            torch.nn.init.normal_(se.adaLN_modulation[0].weight, std=0.02)
    out = S.sample(model, ds)
    assert len(out) == min(4, len(ds)) and all(o.shape[0] == 6 and o.shape[2] == 8 for o in out)
    assert all(bool(torch.isfinite(torch.from_numpy(o)).all()) and float(abs(o).max()) <= 3.1416 for o in out)
    # activations of ~1e6 in the FFN of the decoder layer: beyond the fp16 range (not beyond bf16's)
    with torch.no_grad():
        model.decoder.layer[0].intermediate.dense.weight.mul_(3.0e4)
        model.decoder.layer[0].output.dense.weight.mul_(1.0e-4)
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        out2 = S.sample(model, ds)
    assert any("bf16x6" in str(x.message) for x in w), [str(x.message) for x in w]
    assert all(bool(torch.isfinite(torch.from_numpy(o)).all()) for o in out2)


def test_clip_adamw_three_launch_step_matches_torch(pkg, hip):
    """optim.ClipAdamW (what ``training.adamw`` / ``configure_optimizers`` return on the GPU): gradient-norm clip + AdamW
    in three HIP launches against torch's clip_grad_norm_ + AdamW (single-tensor implementation) over several steps --
    clip active and inactive, element counts that are not multiples of 4 or of the chunk, a parameter that is a view at
    an address that is not 16-byte aligned, a parameter that never receives a gradient, two parameter groups with
    different lr / weight decay, a parameter that starts receiving gradients two steps late (its own step count), and
    the optimizer state moving to a plain torch.optim.AdamW through state_dict."""
    from e3diff_amd.optim import ClipAdamW
    from e3diff_amd.training import adamw, clip_and_step
    dev = "cuda:0"
    torch.manual_seed(0)
    flat = torch.randn(4 * 8192 + 7, device=dev)
    shapes = [(257, 33), (1000,), (3,), (8192 * 3 + 5,), (64, 768)]
    base = [torch.randn(*s, device=dev) for s in shapes]
    idle0 = torch.randn(5, 5, device=dev)
    assert isinstance(adamw([torch.nn.Parameter(base[0].clone())], lr=1e-3, weight_decay=0.1), ClipAdamW)
    for scale in (50.0, 1e-3):
        models = []
        for impl in ("torch", "hip"):
            ps = [torch.nn.Parameter(b.clone()) for b in base]
            odd = torch.nn.Parameter(flat.clone()[1:1 + 8192 + 2])            # data_ptr % 16 == 4
            idle = torch.nn.Parameter(idle0.clone())                          # never gets a gradient
            late = torch.nn.Parameter(base[1].clone())                        # first gradient at step 2
            groups = [dict(params=ps[:3] + [odd, idle], lr=1e-2, weight_decay=0.1), dict(params=ps[3:] + [late], lr=3e-3, weight_decay=0.0)]
            opt = torch.optim.AdamW(groups, foreach=False) if impl == "torch" else ClipAdamW(groups)
            norms = []
            for step in range(5):
                g = torch.Generator(device=dev).manual_seed(100 + step)
                for p in ps + [odd] + ([late] if step >= 2 else []):
                    p.grad = torch.randn(p.shape, device=dev, generator=g) * scale
                params = [p for gr in opt.param_groups for p in gr["params"]]
                if impl == "torch":
                    norms.append(torch.nn.utils.clip_grad_norm_(params, 1.0))
                    opt.step()
                else:
                    norms.append(clip_and_step(params, opt, 1.0).clone())
                opt.zero_grad(set_to_none=True)
            models.append((ps + [odd, late, idle], norms, opt))
        (pa, na, oa), (pb, nb, ob) = models
        for x, y in zip(na, nb):
            assert abs(float(x) - float(y)) <= 2e-6 * float(x)
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7), (x - y).abs().max()
        assert torch.equal(pa[-1], pb[-1])                                    # the idle parameter did not move
        # state round trip: ClipAdamW's state drives a plain torch AdamW to the same next step
        sd = ob.state_dict()
        assert {int(v["step"]) for v in sd["state"].values()} == {5, 3}
        oc = torch.optim.AdamW([dict(params=g_["params"]) for g_ in ob.param_groups], foreach=False)
        oc.load_state_dict(sd)
        for opt, ps in ((oa, pa), (oc, pb)):
            for i, p in enumerate(ps[:-1]):
                p.grad = torch.full_like(p, 0.01 * (i + 1))
            opt.step()
        for x, y in zip(pa, pb):
            assert torch.allclose(x, y, rtol=2e-6, atol=1e-7)


def _small_structure_model(dropout=0.0, seed=0):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=64,
             hidden_dropout_prob=dropout, attention_probs_dropout_prob=dropout)
    torch.manual_seed(seed)
    m = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1, learning_rate=1e-3)
    with torch.no_grad():       # adaLN_modulation[0] is zero-initialised: give the gated branches weights
        for se in (m.receptor_emb, m.timestep_emb):
            torch.nn.init.normal_(se.adaLN_modulation[0].weight, std=0.02)
    return m.train().to("cuda:0")


def _structure_batches(n, B=8, L=64):
    from helpers import synthetic_pockets
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables
    tab = CosineTables(100)
    out = []
    for i in range(n):
        b = B if i != 4 else B - 3                                   # one ragged batch: runs eagerly between replays
        pk = {k: v.to("cuda:0") for k, v in synthetic_pockets(b, L, seed=10 + i).items() if torch.is_tensor(v)}
        g = torch.Generator().manual_seed(500 + i)
        t = torch.randint(0, 100, (b, 1), generator=g).to("cuda:0")
        noise = torch.randn(b, L, 8, generator=g).to("cuda:0")
        out.append(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab, timestep=t, noise=noise)))
    return out


def test_graph_replayed_training_step_matches_the_eager_step(pkg, hip):
    """training.GraphedStep (what ``training.fit`` runs for a single process): two eager steps, capture, replays -- against
    plain eager steps on a twin model with the same batches: per-step losses and the parameters after 10 steps agree, with a
    learning rate AND a beta1 that change every step (device-side scalars: OneCycleLR cycles both), a ragged batch in the middle (eager, then replays again) and
    the optimizer's step counts where torch's AdamW would have them."""
    from e3diff_amd import autograd, ops, training
    batches = _structure_batches(10)
    results = []
    for graphed in (False, True):
        model = _small_structure_model()
        optim = model.configure_optimizers()["optimizer"]
        params = [p for p in model.parameters() if p.requires_grad]
        stepper = training.GraphedStep(model, optim, params, 1.0) if graphed else None
        losses = []
        with ops.arithmetic("bf16x3"):
            for k, batch in enumerate(batches):
                optim.param_groups[0]["lr"] = 1e-3 * (1 + 0.25 * k)
                # ... and a momentum that moves with it, as torch's OneCycleLR (cycle_momentum=True, the reference's
                # lr_scheduler="OneCycleLR": structure_model/model.py:372) rewrites betas[0] every step between 0.85 and 0.95
                optim.param_groups[0]["betas"] = (0.95 - 0.01 * k, 0.999)
                if graphed:
                    # freed caching-allocator blocks full of a poison value between replays: a captured step that reads
                    # memory it does not own (hipMemsetAsync nodes did not clear their destinations reliably inside a
                    # captured graph: the bias / LayerNorm gradients, summed by atomics, then started from whatever the
                    # allocator had there) shows up as a garbage gradient norm
                    junk = [torch.full((n,), 1.2345e30, device="cuda:0") for n in (64, 768, 4096, 65536, 1 << 20, 1 << 22)]
                    del junk
                    torch.cuda.synchronize()
                    losses.append(float(stepper.step(batch)))
                    assert float(optim.last_norm) < 1e3, (k, float(optim.last_norm))
                else:
                    loss = model.training_step(batch)
                    optim.zero_grad(set_to_none=True)
                    with autograd.deferred_weight_grads():
                        loss.backward()
                    training.clip_and_step(params, optim, 1.0)
                    losses.append(float(loss))
                if k == 5:       # optimizer state re-loaded mid-run (resume): a graph must not keep updating the old tensors
                    optim.load_state_dict(optim.state_dict())
        if graphed:
            assert stepper.graph is not None and stepper.failed is None
        assert {int(st["step"]) for st in optim.state.values()} == {10}
        results.append((losses, [p.detach().clone() for p in params]))
    (la, pa), (lb, pb) = results
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(la, lb)), (la, lb)
    # parameters: AdamW turns a gradient element that is zero up to rounding into a +-lr step, so single elements may differ
    # between ANY two runs (small split-K weight-gradient launches sum with atomics); the bulk must agree -- mean difference
    # against the mean distance travelled from the common initial values
    p0 = [p.detach().clone() for p in _small_structure_model().parameters() if p.requires_grad]
    moved = sum(float((a - z).abs().sum()) for a, z in zip(pa, p0))
    apart = sum(float((a - b).abs().sum()) for a, b in zip(pa, pb))
    assert moved > 0 and apart < 0.02 * moved, (apart, moved)


def test_weight_gradients_overlapped_on_a_side_stream_match_the_serial_order(pkg, hip, monkeypatch):
    """autograd.deferred_weight_grads(overlap=True), opt-in (measured: no gain, see the class): slices of the queue leave on a
    side stream while backward runs on.  Every problem of a grouped launch is a whole reduction whatever
    else rides in the launch, so the gradients equal those of the serial order -- bit for bit wherever both orders take the
    grouped launch (a slice with too few tiles falls back to the per-layer split-K kernel, whose atomics are not
    deterministic), to rounding everywhere."""
    from e3diff_amd import autograd, ops
    batch = _structure_batches(1)[0]
    grads = []
    monkeypatch.setattr(autograd.deferred_weight_grads, "OVERLAP_CHUNK", 12)
    for overlap in (False, True):
        model = _small_structure_model()
        with ops.arithmetic("bf16x3"):
            loss = model.training_step(batch)
            with autograd.deferred_weight_grads(overlap=overlap) as q:
                loss.backward()
            assert (q.side_flushes > 0) == overlap and q.queued_total > 12
        torch.cuda.synchronize()
        grads.append({k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None})
    assert grads[0].keys() == grads[1].keys()
    same = 0
    scale = torch.stack([g.abs().max() for g in grads[0].values()]).median().item()   # (key biases: exactly-zero gradients, noise)
    for k, g0 in grads[0].items():
        g1 = grads[1][k]
        assert float((g0 - g1).abs().max()) <= 1e-5 * max(float(g0.abs().max()), 1e-2 * scale), k
        same += int(torch.equal(g0, g1))
    # (biases and the LayerNorm / embedding parameters sum by atomics in both orders: never bit-stable; of the 2+2-layer model's
    #  weights only the groups big enough for the grouped launch in BOTH orders are)
    assert same >= len(grads[0]) // 4, (same, len(grads[0]))


def test_graph_replays_draw_fresh_dropout_decisions(pkg, hip):
    """With the reference's dropout (0.1) every replay of the captured step must drop different elements: the seeds are baked
    into the graph, the device-side epoch word is what moves.  Same batch every step: the losses of consecutive replays
    differ, stay finite, and forward / backward still agree (the step keeps learning on the repeated batch)."""
    from e3diff_amd import ops, training
    batch = _structure_batches(1)[0]
    model = _small_structure_model(dropout=0.1)
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = training.GraphedStep(model, optim, params, 1.0)
    e0 = int(ops.dropout_epoch("cuda:0"))
    with ops.arithmetic("bf16x3"):
        losses = [float(stepper.step(batch)) for _ in range(40)]
    assert stepper.graph is not None and int(ops.dropout_epoch("cuda:0")) == e0 + 40
    assert all(l == l and abs(l) < 1e3 for l in losses)
    assert len({round(l, 6) for l in losses[3:9]}) == 6                 # replays are not copies of each other
    assert sum(losses[-5:]) < sum(losses[2:7])                         # and the model learns the repeated batch


def test_graph_replays_redraw_the_sequence_models_timesteps_and_noise(pkg, hip):
    """PeptideDiff.training_step draws t and the categorical noise with torch's device generator INSIDE the step
    (sequence_model/model.py:347-367): under graph replay the generator's offset must advance per replay -- the same batch
    gives a different loss every replay (dropout off, tiny learning rate), all finite."""
    from helpers import synthetic_pockets
    from e3diff_amd import ops, training
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=64,
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                        loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=50, l2_lambda=0.1, lr=1e-7).train().to("cuda:0")
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    batch = {k: v.to("cuda:0") for k, v in synthetic_pockets(8, 64, seed=3, with_ligand_seq=True).items() if torch.is_tensor(v)}
    stepper = training.GraphedStep(model, optim, params, 1.0)
    with ops.arithmetic("bf16x3"):
        losses = [float(stepper.step(batch)) for _ in range(10)]
    assert stepper.graph is not None and stepper.failed is None
    assert all(l == l and abs(l) < 1e4 for l in losses) and len({round(l, 5) for l in losses[3:]}) >= 6, losses


def test_captured_step_reads_no_memory_it_does_not_own(pkg, hip):
    """After a training step has been captured, EVERY free block of the caching allocator's default pool is claimed and
    filled with a poison value; one replay later the gradient norm must still be sane.  (Found with
    tools/lab/graph_stale_pointer_hunt.py: hipMemsetAsync nodes inside the captured graph did not clear their destinations
    reliably, so the bias / LayerNorm / embedding gradients -- summed by atomics on top of the memset -- picked up whatever
    the allocator had handed out around them: correct for hundreds of replays, then 1e30.  The zero-fills are kernel
    launches now.)  Full model width, two layers: the sizes at which it showed."""
    from helpers import synthetic_pockets
    from e3diff_amd import ops, training
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M
    from e3diff_amd.structure_model.utils import CosineTables
    dev, L, B = "cuda:0", 128, 32
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2, max_position_embeddings=L,
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
              loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1, learning_rate=1e-4).train().to(dev)
    tab = CosineTables(1000)
    optim = model.configure_optimizers()["optimizer"]
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = training.GraphedStep(model, optim, params, 1.0)
    pk = {k: v.to(dev) for k, v in synthetic_pockets(B, L, seed=0).items() if torch.is_tensor(v)}

    def free_blocks():
        out = []
        for seg in torch.cuda.memory_snapshot():
            if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
                continue
            out += [(b["size"], seg.get("stream", 0)) for b in seg["blocks"] if b["state"] == "inactive"]
        return out

    with ops.arithmetic("bf16x3"):
        for _ in range(4):
            stepper.step(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)))
        assert stepper.graph is not None and stepper.failed is None
        sane = float(optim.last_norm)
        batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab))
        torch.cuda.synchronize()
        held = []
        for _ in range(6):
            fb = sorted(free_blocks(), reverse=True)
            if sum(sz for sz, _ in fb) < (1 << 16):
                break
            for size, stream in fb:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()):
                    held.append(torch.empty(max(1, (size - 256) // 4), dtype=torch.float32, device=dev))
        torch.cuda.synchronize()
        for t in held:
            t.fill_(1.2345e30)
        torch.cuda.synchronize()
        assert len(held) > 4
        stepper.step(batch)
        norm = float(optim.last_norm)
    assert 0.1 * sane < norm < 10 * sane, (sane, norm)
    assert all(bool((t == 1.2345e30).all()) for t in held)        # and it wrote into none of them


def _grads_of(model, loss_fn):
    from e3diff_amd import autograd
    for p in model.parameters():
        p.grad = None
    loss = loss_fn()
    with autograd.deferred_weight_grads():
        loss.backward()
    return float(loss), {n: p.grad.detach().clone() for n, p in model.named_parameters() if p.grad is not None}


def _assert_same_gradients(ga, gb, tol):
    assert set(ga) == set(gb)
    # relative to each gradient's largest element (not below 1e-4 of the largest gradient element of the model); the key
    # biases' gradients are zero in exact arithmetic (softmax ignores a shift of all scores of a row): rounding noise, kept apart
    top = max(float(g.abs().max()) for g in ga.values())
    noise = [n for n in ga if n.endswith("key.bias")]
    assert all(float(g[n].abs().max()) < 1e-3 * top for n in noise for g in (ga, gb)), "key-bias gradients are rounding noise"
    worst = max((float((ga[n] - gb[n]).abs().max()) / max(float(ga[n].abs().max()), 1e-4 * top), n) for n in ga if n not in noise)
    assert worst[0] < tol, worst


@pytest.mark.parametrize("rec_hi", [30, 64])   # pockets within one attention tile / over the whole 64-row frame
def test_trimmed_structure_batch_has_the_padded_batchs_loss_and_gradients(pkg, hip, rec_hi):
    """training.trim_batch: the step on the frame of the batch's longest ligand / pocket -- padding can neither reach a valid
    position in the forward pass nor receive a gradient -- gives the loss and EVERY parameter gradient of the padded step up
    to the order of the fp32 sums (other GEMM tile shapes at M/2 or M/4 rows; relative to each gradient's largest element)."""
    from helpers import synthetic_pockets
    from e3diff_amd import ops, training
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables
    model = _small_structure_model()
    pk = {k: v.to("cuda:0") for k, v in synthetic_pockets(8, 64, seed=21, rec_range=(20, rec_hi)).items() if torch.is_tensor(v)}
    g = torch.Generator().manual_seed(7)
    batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], CosineTables(100), timestep=torch.randint(0, 100, (8, 1), generator=g).to("cuda:0"),
                                             noise=torch.randn(8, 64, 8, generator=g).to("cuda:0")))
    small = training.trim_batch(batch)
    assert small["noised_ligand_angle"].shape[1] == 32 and small["known_noise"].shape[1] == 32
    assert small["receptor_seq"].shape[1] == (32 if rec_hi <= 32 else 64) and small["timestep"].shape == batch["timestep"].shape
    with ops.arithmetic("bf16x3"):
        la, ga = _grads_of(model, lambda: model.training_step(batch))
        lb, gb = _grads_of(model, lambda: model.training_step(small))
    assert abs(la - lb) <= 2e-6 * abs(la), (la, lb)
    _assert_same_gradients(ga, gb, 2e-4)


def test_trimmed_sequence_batch_has_the_padded_batchs_loss_and_gradients(pkg, hip):
    """... and PeptideDiff.get_loss on given timesteps and noised residues (training_step draws them per frame position:
    another place of the random stream on another frame, the same distribution)."""
    from helpers import synthetic_pockets
    from e3diff_amd import ops, training
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=64,
             hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    torch.manual_seed(0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("ACDEFGHIKLMNPQRSTVWY"),
                        loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine", timesteps=50, l2_lambda=0.1, lr=1e-4).train().to("cuda:0")
    batch = {k: v.to("cuda:0") for k, v in synthetic_pockets(8, 64, seed=5, with_ligand_seq=True, rec_range=(20, 30)).items() if torch.is_tensor(v)}
    g = torch.Generator().manual_seed(11)
    t_int = torch.randint(1, 51, (8, 1), generator=g).float().to("cuda:0")
    with ops.arithmetic("bf16x3"):
        noised = model.apply_aa_noise(batch["ligand_seq"], t_int, u=torch.rand(8, 64, generator=g).to("cuda:0"))
        small = training.trim_batch(dict(batch, noised=noised))
        assert small["ligand_seq"].shape[1] == 32 and small["receptor_angles"].shape[1] == 32
        la, ga = _grads_of(model, lambda: model.get_loss(batch, t_int / 50, noised)[0])
        lb, gb = _grads_of(model, lambda: model.get_loss(small, t_int / 50, noised[:, :32].contiguous())[0])
    assert abs(la - lb) <= 2e-6 * abs(la), (la, lb)
    _assert_same_gradients(ga, gb, 2e-4)


def test_graphed_step_keeps_one_graph_per_trimmed_frame(pkg, hip):
    """Trimmed batches come in a few frames: GraphedStep captures one graph per batch signature (own pool, own gradient
    tensors, own staging buffer for their addresses) and replays whichever the batch needs -- 12 steps alternating between two
    frames against eager steps on a twin model: the same losses, the same parameters, two graphs, no eager step after warm-up."""
    from helpers import synthetic_pockets
    from e3diff_amd import autograd, ops, training
    from e3diff_amd.structure_model.dataset import noise_batch_on_device
    from e3diff_amd.structure_model.utils import CosineTables
    tab = CosineTables(100)
    batches = []
    for i in range(12):
        pk = {k: v.to("cuda:0") for k, v in synthetic_pockets(8, 64, seed=40 + i, rec_range=(20, 30 if i % 2 else 64)).items()
              if torch.is_tensor(v)}
        pk["receptor_attn_mask"][0, : (30 if i % 2 else 64)] = 1.0            # the frame of the batch is 32 / 64 rows for certain
        g = torch.Generator().manual_seed(900 + i)
        b = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab, timestep=torch.randint(0, 100, (8, 1), generator=g).to("cuda:0"),
                                             noise=torch.randn(8, 64, 8, generator=g).to("cuda:0")))
        batches.append(training.trim_batch(b))
    assert {b["receptor_seq"].shape[1] for b in batches} == {32, 64} and {b["known_noise"].shape[1] for b in batches} == {32}
    results = []
    for graphed in (False, True):
        model = _small_structure_model()
        optim = model.configure_optimizers()["optimizer"]
        params = [p for p in model.parameters() if p.requires_grad]
        stepper = training.GraphedStep(model, optim, params, 1.0) if graphed else None
        losses = []
        with ops.arithmetic("bf16x3"):
            for k, batch in enumerate(batches):
                optim.param_groups[0]["lr"] = 1e-3 * (1 + 0.1 * k)
                if graphed:
                    losses.append(float(stepper.step(batch)))
                    assert float(optim.last_norm) < 1e3, (k, float(optim.last_norm))
                    # ``p.grad`` is the gradient of the step that just ran, whichever graph replayed it
                    seen_norm = torch.sqrt(sum((p.grad.double() ** 2).sum() for p in params if p.grad is not None))
                    assert abs(float(seen_norm) - float(optim.last_norm)) <= 1e-4 * float(optim.last_norm), (k, float(seen_norm))
                else:
                    loss = model.training_step(batch)
                    optim.zero_grad(set_to_none=True)
                    with autograd.deferred_weight_grads():
                        loss.backward()
                    training.clip_and_step(params, optim, 1.0)
                    losses.append(float(loss))
        if graphed:
            assert stepper.failed is None and len(stepper.graphs) == 2 and all(n == 2 for n in stepper.seen.values()), stepper.seen
        assert {int(st["step"]) for st in optim.state.values()} == {12}
        results.append((losses, [p.detach().clone() for p in params]))
    (la, pa), (lb, pb) = results
    assert all(abs(a - b) <= 2e-5 * abs(a) for a, b in zip(la, lb)), (la, lb)
    p0 = [p.detach().clone() for p in _small_structure_model().parameters() if p.requires_grad]
    moved = sum(float((a - z).abs().sum()) for a, z in zip(pa, p0))
    apart = sum(float((a - b).abs().sum()) for a, b in zip(pa, pb))
    assert moved > 0 and apart < 0.02 * moved, (apart, moved)
