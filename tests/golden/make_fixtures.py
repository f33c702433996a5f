#!/usr/bin/env python3
"""Generate the golden fixtures under tests/golden/ by RUNNING THE REFERENCE in the build
container (it does not exist on the GPU box; only these small .pt files travel).

    python tests/golden/make_fixtures.py            # needs /root/reference

How the reference is made importable:
  * ``pytorch_lightning`` is absent -> a throw-away stub package is written to a temp dir
    (LightningModule = nn.Module); it never enters the repo or the runtime path.
  * ``structure_model/sample.py`` / ``sequence_model/sample.py`` have module-level CUDA calls /
    a torch_geometric import, so their sampler FunctionDefs are ast-extracted and exec'd with
    DEVICE='cpu', STEP=1 injected.  Nothing is copied into the repo: only inputs/outputs are saved.
  * transformers here is 5.15: the reference runs WITHOUT the relative_key term (no
    distance_embedding parameters exist), i.e. these forwards pin everything except that term.
RNG: torch.randn_like / Tensor.multinomial are patched while the reference samplers
run so that the injected noise / recorded probabilities are part of the fixture.
"""
import ast
import importlib
import os
import sys
import tempfile

import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.dont_write_bytecode = True

from helpers import TINY, seeded_state_dict, synthetic_pockets  # noqa: E402

STUB = '''
import torch
class LightningModule(torch.nn.Module):
    def log_dict(self, *a, **k): pass
    def all_gather(self, x): return x
class _U:
    @staticmethod
    def rank_zero_info(*a, **k): pass
utilities = _U()
'''


def _stub_dir():
    d = tempfile.mkdtemp(prefix="pl_stub_")
    os.makedirs(os.path.join(d, "pytorch_lightning"))
    with open(os.path.join(d, "pytorch_lightning", "__init__.py"), "w") as f:
        f.write(STUB)
    return d


def _import_ref(subdir):
    """Import reference {model,utils,dataset} from one of its flat script dirs."""
    for name in ("model", "utils", "dataset"):
        sys.modules.pop(name, None)
    path = os.path.join(REF, subdir)
    sys.path.insert(0, path)
    cwd = os.getcwd()
    os.chdir(path)  # './blosum_substitute.pt' relative path, sequence_model/utils.py:274
    try:
        mods = {n: importlib.import_module(n) for n in ("utils", "dataset", "model")}
    finally:
        os.chdir(cwd)
        sys.path.remove(path)
    return mods


def _extract_functions(pyfile, names, env):
    tree = ast.parse(open(pyfile).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert {n.name for n in body} == set(names), [n.name for n in body]
    exec(compile(ast.Module(body=body, type_ignores=[]), pyfile, "exec"), env)
    return env


def _bert_configs(cfg):
    from transformers import BertConfig
    common = dict(max_position_embeddings=cfg["max_seq_len"], num_attention_heads=cfg["num_heads"],
                  hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                  num_hidden_layers=cfg["num_hidden_layers"],
                  position_embedding_type="relative_key", hidden_dropout_prob=0.1,
                  attention_probs_dropout_prob=0.1, use_cache=False)
    enc = BertConfig(**common)
    dec = BertConfig(**common, is_decoder=True, add_cross_attention=True)
    for c in (enc, dec):
        c._attn_implementation = "eager"
    return enc, dec


def save(name, obj):
    path = os.path.join(HERE, name)
    torch.save(obj, path)
    print(f"wrote {name}: {os.path.getsize(path) / 1024:.1f} KiB")


# =============================================================================== structure
def make_structure():
    m = _import_ref("structure_model")
    U, D, M = m["utils"], m["dataset"], m["model"]

    # ---- A12 schedule tables, A13 wrap, A17 losses (+ the docstring known answers)
    fx = {}
    for T in (50, 1000):
        betas = U.cosine_beta_schedule(T)
        fx[f"alphas_T{T}"] = {k: v.clone() for k, v in U.compute_alphas(betas).items()}
    g = torch.Generator().manual_seed(1)
    v = torch.cat([torch.tensor([3.0, -3.0, 0.0, torch.pi, -torch.pi, 7.5, -7.5, 100.0, -100.0,
                                 3.1415927410125732, -3.1415927410125732]),
                   torch.randn(64, generator=g) * 5])
    fx["wrap_in"], fx["wrap_out"] = v, U.modulo_with_wrapped_range(v)
    fx["wrap_m2_2_of_3"] = U.modulo_with_wrapped_range(3, -2, 2)
    a, b = torch.randn(200, generator=g) * 4, torch.randn(200, generator=g) * 4
    fx["loss_in"] = (a, b)
    fx["radian_l1"] = U.radian_l1_loss(a, b)
    fx["radian_smooth_l1_b0.314"] = U.radian_smooth_l1_loss(a, b, beta=torch.pi / 10)
    fx["doc_l1_a"] = U.radian_l1_loss(torch.tensor(0.1), torch.tensor(2 * torch.pi))
    fx["doc_l1_b"] = U.radian_l1_loss(torch.tensor(0.1), torch.tensor(2 * torch.pi - 0.1))
    fx["doc_smooth"] = U.radian_smooth_l1_loss(torch.tensor(-17.0466), torch.tensor(-1.3888), beta=0.1)
    save("structure_utils.pt", fx)

    # ---- A9 forward (tiny config, eval, E==0 because transformers 5.15 has no rel-key branch)
    enc, dec = _bert_configs(TINY)
    model = M.ConditionalBertForDiffusion(
        encoder_config=enc, decoder_config=dec,
        feature_names=D.LigandBindingSiteDataset.feature_names,
        loss_func=[M.ConditionalBertForDiffusion.diheral_loss_func] * 4
        + [M.ConditionalBertForDiffusion.angle_loss_func] * 4)
    ref_sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in ref_sd.items()}
    assert not any("distance_embedding" in k for k in shapes), "unexpected rel-key params"
    model.load_state_dict(seeded_state_dict(shapes, seed=11))
    model.eval()
    L = TINY["max_seq_len"]
    pk = synthetic_pockets(2, L, seed=3, lig_range=(3, 9), rec_range=(6, L))
    g = torch.Generator().manual_seed(5)
    x_t = U.modulo_with_wrapped_range(torch.randn(2, L, 8, generator=g))
    outs = {}
    with torch.no_grad():
        for tag, t in (("t_B", torch.tensor([7, 7])), ("t_B1", torch.tensor([[999], [3]]))):
            outs[tag] = (t, model(t, x_t, pk["ligand_attn_mask"], pk["receptor_seq"],
                                  pk["receptor_angles"], pk["receptor_attn_mask"]))
        # A17 _get_loss_terms through the reference wrapper
        known = U.modulo_with_wrapped_range(torch.randn(2, L, 8, generator=g))
        batch = {"known_noise": known, "timestep": torch.tensor([[12], [400]]),
                 "noised_ligand_angle": x_t, "ligand_attn_mask": pk["ligand_attn_mask"],
                 "receptor_seq": pk["receptor_seq"], "receptor_angles": pk["receptor_angles"],
                 "receptor_attn_mask": pk["receptor_attn_mask"],
                 "ligand_pos_id": None, "receptor_pos_id": None}
        loss_terms = model._get_loss_terms(batch)
        pred_for_loss = model(batch["timestep"], x_t, pk["ligand_attn_mask"], pk["receptor_seq"],
                              pk["receptor_angles"], pk["receptor_attn_mask"])
    save("structure_forward_tiny.pt", {
        "cfg": TINY, "seed": 11, "shapes": shapes, "pockets": pk, "x_t": x_t, "outs": outs,
        "known_noise": known, "loss_timestep": batch["timestep"], "loss_terms": loss_terms,
        "loss_pred": pred_for_loss})

    # ---- A15/A16 sampler with the real tiny model and injected noise
    env = {"torch": torch, "nn": torch.nn, "tqdm": lambda it, **k: it, "DEVICE": "cpu", "STEP": 1,
           "compute_alphas": U.compute_alphas,
           "modulo_with_wrapped_range": U.modulo_with_wrapped_range,
           "ConditionalBertForDiffusion": M.ConditionalBertForDiffusion}
    _extract_functions(os.path.join(REF, "structure_model", "sample.py"),
                       ["p_sample", "p_sample_loop"], env)
    T = 6
    betas = U.cosine_beta_schedule(T)
    noises = torch.randn(T, 2, L, 8, generator=g)
    it = iter(noises)
    real_randn_like = torch.randn_like
    torch.randn_like = lambda x, **k: next(it)
    try:
        traj = env["p_sample_loop"](
            model=model, ligand_mask=pk["ligand_attn_mask"], ligand_angle_noise=x_t,
            receptor_seq=pk["receptor_seq"], receptor_mask=pk["receptor_attn_mask"],
            receptor_angle=pk["receptor_angles"], total_timesteps=T, betas=betas,
            disable_pbar=True)
    finally:
        torch.randn_like = real_randn_like
    save("structure_sampler_tiny.pt", {"T": T, "noises": noises, "x_T": x_t, "traj": traj})

    # ---- A14/A26 dataset layout + forward noising on a synthetic biolip.pt
    recs = synthetic_biolip_records(12, seed=2)
    tmp = os.path.join(tempfile.mkdtemp(), "biolip.pt")
    torch.save(recs, tmp)
    ds = D.LigandBindingSiteDataset(tmp, "train", max_len=32, pocket_ext=1)
    item = ds[0]
    nds = D.NoisedAnglesDataset(ds, timesteps=100)
    inj = torch.randn(32, 8, generator=g)
    torch.randn_like = lambda x, **k: inj
    try:
        nitem = nds.__getitem__(1, use_timestep=37)
    finally:
        torch.randn_like = real_randn_like
    keep = lambda d: {k: v for k, v in d.items() if k != "structure_ids"}  # noqa: E731
    save("structure_dataset.pt", {
        "records": recs, "n_train": len(ds), "item0": keep(item), "item0_ids": item["structure_ids"],
        "noised_item1": keep(nitem), "injected_randn": inj,
        "split_sizes": {s: len(D.LigandBindingSiteDataset(tmp, s, 32, 1))
                        for s in ("train", "validation", "test")}})


def synthetic_biolip_records(n, seed=0):
    """list[dict] in the biolip.pt schema (clean_data/data_preprocessing.py:880-892)."""
    import math
    g = torch.Generator().manual_seed(seed)
    aa = "ACDEFGHIKLMNPQRSTVWY"
    ss = "HBEGITS-"
    recs = []
    for i in range(n):
        n_rec = int(torch.randint(20, 40, (1,), generator=g))
        n_lig = int(torch.randint(4, 12, (1,), generator=g))
        N = n_rec + n_lig
        ligand_mask = torch.zeros(N, dtype=torch.bool)
        ligand_mask[n_rec:] = True
        pocket_mask = torch.zeros(N, dtype=torch.bool)
        pocket_mask[torch.randperm(n_rec, generator=g)[:10]] = True
        ang = (torch.rand(N, 8, generator=g) * 2 - 1) * math.pi
        recs.append({
            "structure_ids": {"pdb_id": f"s{i:03d}", "receptor_chain": "A", "ligand_chain": "B"},
            "coors": torch.randn(N, 3, generator=g),
            "amino_acid": [aa[int(j)] for j in torch.randint(0, 20, (N,), generator=g)],
            "secondary_structure": [ss[int(j)] for j in torch.randint(0, 8, (N,), generator=g)],
            "numerical_features": torch.randn(N, 5, generator=g),
            "angle_features": ang,
            "edge_index": torch.zeros(2, 0, dtype=torch.long),
            "ligand_mask": ligand_mask,
            "ligand_idx": torch.nonzero(ligand_mask).squeeze(-1).int(),
            "pocket_mask": pocket_mask,
            "pocket_idx": torch.nonzero(pocket_mask).squeeze(-1).int(),
        })
    return recs


# =============================================================================== sequence
def make_sequence():
    m = _import_ref("sequence_model")
    U, D, M = m["utils"], m["dataset"], m["model"]
    cwd = os.getcwd()
    os.chdir(os.path.join(REF, "sequence_model"))
    try:
        blosum = U.BlosumTransition(x_classes=20)
        enc, dec = _bert_configs(TINY)
        model = M.PeptideDiff(encoder_config=enc, decoder_config=dec,
                              feature_names=D.LigandBindingSiteDataset.feature_names,
                              loss_func=torch.nn.CrossEntropyLoss(), noise_schedule="cosine",
                              timesteps=50)
    finally:
        os.chdir(cwd)
    uniform = U.DiscreteUniformTransition(20)
    T = 50
    sched = U.PredefinedNoiseScheduleDiscrete("cosine", T)

    # ---- A18/A19/A20 tables
    fx = {"betas": sched.betas.clone(), "alphas_bar": sched.alphas_bar.clone(),
          "blosum_temperature_501": blosum.temperature_list.clone()}
    t_norm = (torch.arange(T + 1).float() / T).unsqueeze(1)
    ab = sched.get_alpha_bar(t_normalized=t_norm)
    fx["alpha_bar_of_t"] = ab.clone()
    fx["blosum_t_index"] = torch.round(ab * blosum.timestep).long()
    fx["blosum_Qtb"] = blosum.get_Qt_bar(ab, "cpu").clone()
    fx["uniform_Qtb"] = uniform.get_Qt_bar(ab, "cpu").clone()
    # half-way inputs for round-half-even
    fx["round_probe_in"] = torch.tensor([[0.001], [0.003], [0.005], [0.007], [0.009], [0.5], [0.25]])
    fx["round_probe_idx"] = torch.round(fx["round_probe_in"] * 500).long()
    fx["elbo_in"] = (torch.randn(7, 20, generator=torch.Generator().manual_seed(4)),
                     torch.nn.functional.one_hot(torch.arange(7) % 20, 20).float())
    fx["elbo"] = U.elbo_loss(*fx["elbo_in"])
    save("sequence_utils.pt", fx)

    # ---- A10 forward
    ref_sd = model.state_dict()
    shapes = {k: tuple(v.shape) for k, v in ref_sd.items()}
    assert not any("distance_embedding" in k for k in shapes)
    model.load_state_dict(seeded_state_dict(shapes, seed=21))
    model.eval()
    L = TINY["max_seq_len"]
    pk = synthetic_pockets(2, L, seed=8, lig_range=(3, 9), rec_range=(6, L), with_ligand_seq=True)
    g = torch.Generator().manual_seed(9)
    x_t = torch.nn.functional.one_hot(torch.randint(0, 20, (2, L), generator=g), 20).float()
    outs = {}
    with torch.no_grad():
        for tag, t in (("raw_s", torch.tensor([[17.0], [17.0]])),
                       ("t_norm", torch.tensor([[0.34], [0.9]]))):
            outs[tag] = (t, model(t, x_t, pk["ligand_angles"], pk["ligand_attn_mask"],
                                  pk["receptor_seq"], pk["receptor_angles"],
                                  pk["receptor_attn_mask"]))

    # ---- A21 apply_aa_noise: multinomial patched to (record prob, return argmax)
    recorded = []
    real_multinomial = torch.Tensor.multinomial

    def fake_multinomial(self, n, *a, **k):
        recorded.append(self.clone())
        return self.argmax().reshape(1)

    t_int = torch.tensor([[5.0], [33.0]])
    torch.Tensor.multinomial = fake_multinomial
    try:
        noised_argmax = model.apply_aa_noise(pk["ligand_seq"], t_int)
    finally:
        torch.Tensor.multinomial = real_multinomial
    aa_prob_rows = torch.stack(recorded)  # only the non-zero (non-padding) rows, in order

    # second draw: every third non-padding row takes its 2nd most likely class, so that the
    # "noised" CE/ELBO terms of get_loss have a non-empty support
    calls = [0]

    def fake_multinomial_mixed(self, n, *a, **k):
        calls[0] += 1
        order = self.argsort(descending=True)
        return order[1 if calls[0] % 3 == 0 else 0].reshape(1)

    torch.Tensor.multinomial = fake_multinomial_mixed
    try:
        noised_mixed = model.apply_aa_noise(pk["ligand_seq"], t_int)
    finally:
        torch.Tensor.multinomial = real_multinomial

    # ---- A25 get_loss
    with torch.no_grad():
        loss = model.get_loss(pk, t_int / T, noised_mixed)
    save("sequence_forward_tiny.pt", {
        "cfg": TINY, "seed": 21, "shapes": shapes, "pockets": pk, "x_t": x_t, "outs": outs,
        "aa_t_int": t_int, "aa_noised_argmax": noised_argmax, "aa_prob_rows": aa_prob_rows,
        "aa_noised_mixed": noised_mixed,
        "loss_t_norm": t_int / T, "loss": [x.clone() for x in loss]})

    # ---- A22/A23/A24 reverse sampler
    env = {"torch": torch, "F": torch.nn.functional, "DEVICE": "cpu"}
    _extract_functions(os.path.join(REF, "sequence_model", "sample.py"),
                       ["generate_discrete_noise", "compute_batched_over0_posterior_distribution",
                        "sample_p_zs_given_zt_discrete"], env)
    fn = env["sample_p_zs_given_zt_discrete"]
    logits = torch.randn(2, L, 20, generator=g) * 2
    cases = {}
    for trans_name, trans in (("blosum", blosum), ("uniform", uniform)):
        for s_int in (0, 1, 24, 48, 49):
            s = s_int * torch.ones(2, 1) / T
            t = (s_int + 1) * torch.ones(2, 1) / T
            recorded.clear()
            torch.Tensor.multinomial = fake_multinomial
            try:
                fn(t, s, x_t.clone(), logits.clone(), sched, trans, True, False)
            finally:
                torch.Tensor.multinomial = real_multinomial
            prob = torch.stack(recorded)
            x_s = fn(t, s, x_t.clone(), logits.clone(), sched, trans, False, False)
            cases[(trans_name, s_int)] = {"prob_X": prob, "argmax_onehot": x_s}
    last = fn(None, None, x_t, logits, sched, blosum, True, True)
    assert last is logits
    post_in = {k: v for k, v in zip(("X_t", "Q_t", "Qsb", "Qtb"),
                                    (x_t.reshape(-1, 20), fx["blosum_Qtb"][3:5], fx["blosum_Qtb"][2:4],
                                     fx["blosum_Qtb"][3:5]))}
    rep = torch.arange(2).repeat_interleave(L)
    post = env["compute_batched_over0_posterior_distribution"](batch=rep, **post_in)
    noise = env["generate_discrete_noise"](4, L, 20)
    assert noise.shape == (4, L, 20) and bool((noise.sum(-1) == 1).all())
    save("sequence_sampler.pt", {"T": T, "logits": logits, "x_t": x_t, "cases": cases,
                                 "posterior_in": post_in, "posterior_out": post})


if __name__ == "__main__":
    assert os.path.isdir(REF), "fixture generation needs the reference checkout"
    sys.path.insert(0, _stub_dir())
    torch.manual_seed(0)
    make_structure()
    make_sequence()
    import shutil
    shutil.copyfile(os.path.join(REF, "sequence_model", "blosum_substitute.pt"),
                    os.path.join(HERE, "blosum_substitute.pt"))  # data file
    print("done")
