"""SURVEY section 8(f) "next" rows: NeRF backbone builder (rank 3), biolip.pt schema + synthetic
generator (rank 4), joint structure->sequence hand-over (rank 1)."""
import os

import numpy as np
import pytest
import torch

from helpers import GOLDEN
from oracle import nerf as onerf


def nerf_fx():
    return torch.load(os.path.join(GOLDEN, "nerf.pt"), weights_only=False)


# ------------------------------------------------------------------------------- CPU
def test_nerf_oracle_matches_reference_bit_exact():
    fx = nerf_fx()
    assert fx["cols"] == onerf.COLS
    for c in fx["cases"]:
        got = onerf.backbone_coords(c["angles"].numpy(), c["center"])
        assert np.array_equal(got, c["coords"].numpy())
    # geometry sanity of the construction: bond lengths are the reference constants
    xyz = onerf.backbone_coords(fx["cases"][-1]["angles"].numpy(), False).reshape(-1, 4, 3)
    d = lambda a, b: np.linalg.norm(a - b, axis=-1)  # noqa: E731
    assert np.allclose(d(xyz[1:, 0], xyz[:-1, 2]), 1.34, atol=1e-6)      # C(i-1)-N(i)
    assert np.allclose(d(xyz[1:, 1], xyz[1:, 0]), 1.46, atol=1e-6)       # N-CA
    assert np.allclose(d(xyz[1:, 2], xyz[1:, 1]), 1.54, atol=1e-6)       # CA-C
    assert np.allclose(d(xyz[:, 3], xyz[:, 2]), 1.22, atol=1e-6)         # C=O


def test_biolip_schema_and_synthetic_generator(pkg, tmp_path):
    from e3diff_amd import biolip
    from e3diff_amd.sequence_model.dataset import LigandBindingSiteDataset
    recs = biolip.synthetic_records(20, seed=3)
    assert biolip.validate(recs) == sum(len(r["amino_acid"]) for r in recs)
    path = biolip.write_synthetic(str(tmp_path / "biolip.pt"), 20, seed=3)
    again = biolip.load(path)
    assert again[7]["structure_ids"] == recs[7]["structure_ids"]
    # the reference-generated fixture records obey the same schema, and the datasets consume the file
    fx = torch.load(os.path.join(GOLDEN, "structure_dataset.pt"), weights_only=False)
    biolip.validate(fx["records"])
    ds = LigandBindingSiteDataset(path, "train", max_len=256, pocket_ext=1)
    item = ds[0]
    assert item["ligand_seq"].shape == (256, 20) and item["receptor_angles"].shape == (256, 8)
    bad = dict(recs[0], ligand_mask=~recs[0]["ligand_mask"])
    with pytest.raises(biolip.BiolipSchemaError, match="trailing block"):
        biolip.validate([bad])
    with pytest.raises(biolip.BiolipSchemaError, match="missing keys"):
        biolip.validate([{k: v for k, v in recs[0].items() if k != "coors"}])
    with pytest.raises(biolip.BiolipSchemaError, match="radians"):
        biolip.validate([dict(recs[0], angle_features=recs[0]["angle_features"] * 57.3)])


def test_pdb_text_layout(pkg):
    from e3diff_amd.structure_model.create_pdb import pdb_text
    xyz = onerf.backbone_coords(nerf_fx()["cases"][2]["angles"].numpy(), True)      # 3 residues
    text = pdb_text(xyz).splitlines()
    atoms = [ln for ln in text if ln.startswith("ATOM")]
    assert len(atoms) == 12 and all(len(ln) == 78 for ln in atoms)
    assert atoms[1][12:16] == " CA " and atoms[1][17:20] == "GLY" and atoms[1][21] == "A" and atoms[5][22:26] == "   2"
    assert float(atoms[4][30:38]) == pytest.approx(xyz[4, 0], abs=5e-4)
    assert atoms[3][76:78] == " O" and text[-1] == "END"
    conect = {int(ln[6:11]): [int(ln[i:i + 5]) for i in range(11, len(ln), 5)] for ln in text if ln.startswith("CONECT")}
    assert conect[3] == [2, 4, 5] and conect[5] == [3, 6]          # C(1) bonds CA, O and N(2)


# ------------------------------------------------------------------------------- GPU
@pytest.mark.gpu
def test_nerf_kernel_matches_reference(pkg, hip):
    from e3diff_amd.structure_model.create_pdb import backbone_from_angles, coords_for_chains
    fx = nerf_fx()
    for center in (True, False):
        cases = [c for c in fx["cases"] if c["center"] == center]
        L = max(c["angles"].shape[0] for c in cases)
        batch = torch.zeros(len(cases), L, 8)
        for i, c in enumerate(cases):
            batch[i, :c["angles"].shape[0]] = c["angles"]
        lens = torch.tensor([c["angles"].shape[0] for c in cases])
        out = backbone_from_angles(batch.cuda(), lens.cuda(), center).cpu()
        assert out.dtype == torch.float64 and out.shape == (len(cases), L, 4, 3)
        for i, c in enumerate(cases):
            l = int(lens[i])
            # float32 trig of the device vs numpy differs in the last ulp; 64 residues amplify it to ~1e-5 A
            assert (out[i, :l].reshape(-1, 3) - c["coords"]).abs().max() < 1e-4
            assert float(out[i, l:].abs().sum()) == 0.0
    chains = [c["angles"].numpy() for c in fx["cases"] if c["center"]]
    for got, c in zip(coords_for_chains(chains, True, device="cuda:0"), [c for c in fx["cases"] if c["center"]]):
        assert np.abs(got - c["coords"].numpy()).max() < 1e-4


@pytest.mark.gpu
def test_write_preds_pdb_folder(pkg, hip, tmp_path):
    from e3diff_amd.structure_model.create_pdb import load_sampled_angles, write_preds_pdb_folder
    import pickle
    fx = nerf_fx()
    traj = [np.stack([c["angles"].numpy()] * 3) for c in fx["cases"][:4:2]]      # [T,l,8] like the sampler's pickle
    pickle.dump(traj, open(tmp_path / "output.pkl", "wb"))
    chains = load_sampled_angles(str(tmp_path / "output.pkl"))
    files = write_preds_pdb_folder(chains, str(tmp_path / "out"))
    assert len(files) == 2 and all(os.path.exists(f) for f in files)
    assert open(files[0]).read().count("ATOM") == 4 * chains[0].shape[0]


@pytest.mark.gpu
def test_joint_structure_to_sequence_pipeline_on_device(pkg, hip):
    """BASELINE config 5 in miniature: structure chain -> last-step angles handed over on the GPU
    (no pickle round trip) -> sequence chain with the uniform transition, as
    sample_by_generated_angles.py does."""
    from helpers import seeded_state_dict, synthetic_pockets
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusionBase
    from e3diff_amd.structure_model.sample import p_sample_loop
    from e3diff_amd.structure_model.utils import CosineTables, modulo_with_wrapped_range
    from e3diff_amd.sequence_model import sample as seq_sample
    from e3diff_amd.sequence_model.model import PeptideDiff
    from e3diff_amd.sequence_model.sample_by_generated_angles import angles_from_trajectory, denoise
    from e3diff_amd.sequence_model.utils import DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete
    L, B = 64, 4
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=1, max_position_embeddings=L)
    enc, dec = BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True)
    smodel = ConditionalBertForDiffusionBase(enc, dec, 8)
    smodel.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in smodel.state_dict().items()}, seed=1))
    smodel = smodel.eval().cuda()
    qmodel = PeptideDiff(enc, dec, feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                         noise_schedule="cosine", timesteps=5)
    qmodel.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in qmodel.state_dict().items()}, seed=2))
    qmodel = qmodel.eval().cuda()
    pk = synthetic_pockets(B, L, seed=3, with_ligand_seq=True)
    d = {k: v.cuda() for k, v in pk.items() if torch.is_tensor(v)}
    x_T = modulo_with_wrapped_range(torch.randn(B, L, 8)).cuda()
    traj = p_sample_loop(smodel, d["ligand_attn_mask"], x_T, d["receptor_seq"], d["receptor_attn_mask"],
                         d["receptor_angles"], 4, CosineTables(4), disable_pbar=True, return_device=True, step=1)
    gen = angles_from_trajectory(traj, d["ligand_attn_mask"])
    assert gen.shape == (B, L, 8) and float((gen * (1 - d["ligand_attn_mask"])[..., None]).abs().sum()) == 0.0
    sched = PredefinedNoiseScheduleDiscrete("cosine", 5).cuda()
    ids, true_seq, pred_seq, rates = denoise(dict(pk, structure_ids=None), gen, qmodel, sched,
                                             DiscreteUniformTransition(20), True, timesteps=5)
    assert len(pred_seq) == B and all(len(p) == int(n) for p, n in zip(pred_seq, pk["ligand_length"]))
    assert all(0.0 <= r <= 1.0 for r in rates) and set("".join(pred_seq)) <= set("ACDEFGHIKLMNPQRSTVWY")
    assert seq_sample.CONFIG["timesteps"] == 50          # the per-call override did not leak


@pytest.mark.gpu
def test_trimmed_sequence_chain_gives_the_same_sequences(pkg, hip):
    """sequence_model/sample.py::denoise(trim_padding=True): deterministic (argmax) chain on the frame trimmed to
    the longest ligand / pocket against the padded frame -- identical predicted sequences."""
    from helpers import seeded_state_dict, synthetic_pockets
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    from e3diff_amd.sequence_model.sample import denoise, generate_discrete_noise
    from e3diff_amd.sequence_model.utils import DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete
    L, B, T = 128, 4, 6
    c = dict(hidden_size=256, num_attention_heads=4, intermediate_size=512, num_hidden_layers=2, max_position_embeddings=L)
    enc, dec = BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True)
    qmodel = PeptideDiff(enc, dec, feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                         noise_schedule="cosine", timesteps=T)
    qmodel.load_state_dict(seeded_state_dict({k: tuple(v.shape) for k, v in qmodel.state_dict().items()}, seed=2))
    qmodel = qmodel.eval().cuda()
    pk = dict(synthetic_pockets(B, L, seed=8, with_ligand_seq=True, rec_range=(20, 60)), structure_ids=None)
    sched = PredefinedNoiseScheduleDiscrete("cosine", T).cuda()
    torch.manual_seed(4)
    x_T = generate_discrete_noise(B, L, 20, "cuda")
    full = denoise(pk, qmodel, sched, DiscreteUniformTransition(20), False, x_T=x_T, timesteps=T)
    trim = denoise(pk, qmodel, sched, DiscreteUniformTransition(20), False, x_T=x_T, timesteps=T, trim_padding=True)
    assert trim[2] == full[2] and trim[1] == full[1] and trim[3] == full[3]
