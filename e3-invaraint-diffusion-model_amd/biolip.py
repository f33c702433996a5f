"""The on-disk input either side of the hot path: ``biolip.pt`` = ``torch.save`` of ``list[dict]``
as written by the reference's offline preprocessing (clean_data/data_preprocessing.py:838-893).
This module validates that schema and generates synthetic BioLiP-shaped files for benchmarks/tests
(SURVEY.md section 8(f) rank 4).  The preprocessing itself (BioLiP meta TSV + PDB/CIF -> DSSP
features) needs BioLiP downloads, Biopython and a DSSP binary and is out of scope.

Schema of one record (receptor residues first, then ligand residues; N = both):
    structure_ids        dict(pdb_id, receptor_chain, ligand_chain)
    coors                f32 [N,3]   C-alpha coordinates
    amino_acid           list[str] of N one-letter codes from "ACDEFGHIKLMNPQRSTVWY"
    secondary_structure  list[str] of N DSSP codes from "HBEGITS-"
    numerical_features   f32 [N,5]
    angle_features       f32 [N,8]   radians.  Columns as the preprocessing stores them:
                         omega, phi, psi, dihedral_o, theta1 (N:CA:C), theta2 (CA:C:1N),
                         theta3 (1C:N:CA), theta_o (CA:C:O) (data_preprocessing.py:720-730); the
                         datasets LABEL the same columns phi, psi, omega, dihedral_o, tau, ...
                         (structure_model/dataset.py:17) -- a reference quirk kept as is
    edge_index           i64 [2,E]   ligand x pocket bipartite pairs (never read by the models)
    ligand_mask          bool [N]    ;  ligand_idx  i32 [n_lig]
    pocket_mask          bool [N]    ;  pocket_idx  i32 [n_pocket]
"""
import math

import torch

AA_VOCAB = "ACDEFGHIKLMNPQRSTVWY"
SS_VOCAB = "HBEGITS-"
STORED_ANGLE_COLUMNS = ("omega", "phi", "psi", "dihedral_o", "theta1", "theta2", "theta3", "theta_o")
_KEYS = ("structure_ids", "coors", "amino_acid", "secondary_structure", "numerical_features", "angle_features",
         "edge_index", "ligand_mask", "ligand_idx", "pocket_mask", "pocket_idx")


class BiolipSchemaError(ValueError):
    pass


def validate_record(rec, index=0):
    def fail(msg):
        raise BiolipSchemaError(f"record {index}: {msg}")

    missing = [k for k in _KEYS if k not in rec]
    if missing:
        fail(f"missing keys {missing}")
    n = len(rec["amino_acid"])
    if not {"pdb_id", "receptor_chain", "ligand_chain"} <= set(rec["structure_ids"]):
        fail("structure_ids needs pdb_id, receptor_chain, ligand_chain")
    if len(rec["secondary_structure"]) != n:
        fail("secondary_structure length != amino_acid length")
    if any(a not in AA_VOCAB for a in rec["amino_acid"]):
        fail("amino_acid outside the 20-letter vocabulary")
    if any(s not in SS_VOCAB for s in rec["secondary_structure"]):
        fail("secondary_structure outside 'HBEGITS-'")
    for key, shape in (("coors", (n, 3)), ("numerical_features", (n, 5)), ("angle_features", (n, 8))):
        t = rec[key]
        if not torch.is_tensor(t) or tuple(t.shape) != shape or not t.is_floating_point():
            fail(f"{key} must be a float tensor of shape {shape}")
    ang = rec["angle_features"]
    if torch.isfinite(ang).all() and float(ang.abs().max()) > math.pi + 1e-3:
        fail("angle_features must be radians in [-pi, pi]")
    for key in ("ligand_mask", "pocket_mask"):
        m = rec[key]
        if not torch.is_tensor(m) or m.dtype != torch.bool or tuple(m.shape) != (n,):
            fail(f"{key} must be bool [{n}]")
    lig = rec["ligand_mask"]
    n_lig = int(lig.sum())
    if n_lig == 0 or not bool(lig[n - n_lig:].all()) or bool(lig[:n - n_lig].any()):
        fail("ligand residues must be the trailing block (receptor first, then ligand)")
    if bool((rec["pocket_mask"] & lig).any()):
        fail("pocket residues must belong to the receptor")
    if rec["ligand_idx"].tolist() != list(range(n - n_lig, n)):
        fail("ligand_idx inconsistent with ligand_mask")
    if sorted(rec["pocket_idx"].tolist()) != torch.nonzero(rec["pocket_mask"]).flatten().tolist():
        fail("pocket_idx inconsistent with pocket_mask")
    e = rec["edge_index"]
    if not torch.is_tensor(e) or e.dtype != torch.int64 or e.dim() != 2 or e.shape[0] != 2:
        fail("edge_index must be int64 [2,E]")
    return n


def validate(records):
    """Raise BiolipSchemaError on the first malformed record; returns the number of residues seen."""
    if not isinstance(records, (list, tuple)) or not records:
        raise BiolipSchemaError("biolip.pt must hold a non-empty list of dicts")
    return sum(validate_record(r, i) for i, r in enumerate(records))


def load_records(path):
    """``torch.load(..., weights_only=True)``: the documented schema is only list / dict / str / tensors, which
    the restricted unpickler accepts, so nothing in a user-supplied biolip.pt is ever executed."""
    try:
        return torch.load(path, map_location="cpu", weights_only=True)
    except Exception as e:   # pickle.UnpicklingError and friends: anything beyond plain containers and tensors
        raise BiolipSchemaError(
            f"{path}: not loadable as plain containers + tensors (weights_only=True refused it: {e}). "
            "A biolip.pt holds list[dict] of str / list[str] / tensors only (clean_data/data_preprocessing.py:880-892)."
        ) from e


def load(path):
    """safe load + validate."""
    records = load_records(path)
    validate(records)
    return records


def synthetic_records(n, seed=0, receptor_len=(40, 200), ligand_len=(5, 30), pocket_size=(8, 40)):
    """BioLiP-shaped synthetic complexes (the benchmark's stand-in for the real file): dihedrals
    ~ U(-pi, pi), bond angles ~ N(1.95, 0.1) rad, uniform residues, a random pocket subset."""
    g = torch.Generator().manual_seed(seed)

    def ri(lo, hi):
        return int(torch.randint(lo, hi + 1, (1,), generator=g))

    out = []
    for i in range(n):
        n_rec, n_lig = ri(*receptor_len), ri(*ligand_len)
        N = n_rec + n_lig
        angles = torch.empty(N, 8)
        angles[:, :4] = (torch.rand(N, 4, generator=g) * 2 - 1) * math.pi
        angles[:, 4:] = (1.95 + 0.1 * torch.randn(N, 4, generator=g)).clamp(0.5, math.pi)
        ligand_mask = torch.zeros(N, dtype=torch.bool)
        ligand_mask[n_rec:] = True
        pocket_mask = torch.zeros(N, dtype=torch.bool)
        pocket_mask[torch.randperm(n_rec, generator=g)[:min(ri(*pocket_size), n_rec)]] = True
        lig_idx = torch.arange(n_rec, N, dtype=torch.int)
        poc_idx = torch.nonzero(pocket_mask).flatten().int()
        out.append({
            "structure_ids": {"pdb_id": f"syn{i:05d}", "receptor_chain": "A", "ligand_chain": "B"},
            "coors": torch.randn(N, 3, generator=g) * 10,
            "amino_acid": [AA_VOCAB[int(j)] for j in torch.randint(0, 20, (N,), generator=g)],
            "secondary_structure": [SS_VOCAB[int(j)] for j in torch.randint(0, 8, (N,), generator=g)],
            "numerical_features": torch.randn(N, 5, generator=g),
            "angle_features": angles,
            "edge_index": torch.cartesian_prod(lig_idx.long(), poc_idx.long()).T.contiguous(),
            "ligand_mask": ligand_mask, "ligand_idx": lig_idx,
            "pocket_mask": pocket_mask, "pocket_idx": poc_idx,
        })
    return out


def write_synthetic(path, n, seed=0, **kw):
    records = synthetic_records(n, seed, **kw)
    validate(records)
    torch.save(records, path)
    return path
