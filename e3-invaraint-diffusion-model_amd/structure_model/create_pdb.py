"""Sampled angles -> backbone coordinates -> PDB files (the step after structure sampling).

Entry point and function names of the reference's structure_model/create_pdb.py.  The NeRF chain
builder (reference lines 40-234: a Python loop of 3 numpy placements per residue, one pocket at a
time) is one HIP launch over the whole batch (``e3d_nerf_backbone``); the PDB text is written
directly (fixed-column ATOM/CONECT records, glycine backbone, as the reference's biotite call
produces: chain A, occupancy 1.00, B-factor 5.00) -- biotite is not needed.

Run as ``python create_pdb.py`` from this directory after editing the constants, like the reference.
"""
if __package__ in (None, ""):
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    import __graft_entry__ as _g
    _g.load_package()
    __package__ = "e3diff_amd.structure_model"

import os
import pickle
from typing import List, Sequence

import numpy as np
import torch

from .. import hip

TRUE_DATA = "./data/biolip.pt"
GENERATED_DATA = "./data/output.pkl"
OUTPUT_FODLER = "./data/output"          # (sic) the reference's constant name
GPU_ID = 0

COLS = ["phi", "psi", "omega", "dihedral_o", "tau", "CA:C:1N", "1C:N:CA", "CA:C:O"]
REQUIRED_DIHEDRALS = ["phi", "psi", "omega", "dihedral_o"]
_ATOMS = (("N", "N"), ("CA", "C"), ("C", "C"), ("O", "O"))


def backbone_from_angles(angles: torch.Tensor, lengths: torch.Tensor, center: bool = True) -> torch.Tensor:
    """angles [B,L,8] fp32 (COLS order, radians) + lengths [B] -> float64 coords [B,L,4,3] on the
    GPU (N, CA, C, O per residue; residues past ``lengths`` are zero).  ``center`` subtracts each
    pocket's mean atom position (NERFBuilder.centered_cartesian_coords)."""
    if not angles.is_cuda:
        raise RuntimeError("backbone_from_angles runs on the HIP kernel only: pass GPU tensors (no CPU fallback)")
    B, L, F = angles.shape
    assert F == 8, angles.shape
    ang = angles.contiguous().float()
    lens = lengths.to(device=ang.device, dtype=torch.int32).contiguous()
    out = torch.empty((B, L, 4, 3), device=ang.device, dtype=torch.float64)
    hip.check(hip.lib().e3d_nerf_backbone(ang.data_ptr(), lens.data_ptr(), out.data_ptr(), int(center), B, L,
                                          torch.cuda.current_stream().cuda_stream), "e3d_nerf_backbone")
    return out


def pdb_text(coords: np.ndarray) -> str:
    """[4n,3] coordinates (N, CA, C, O per residue) -> PDB text: one GLY residue per 4 atoms,
    chain A, single N-CA-C backbone bonds, C=O, peptide bonds C(i)-N(i+1)."""
    assert len(coords) % 4 == 0, f"Expected 4N coords, got {len(coords)}"
    lines = []
    for i, (x, y, z) in enumerate(np.asarray(coords, dtype=np.float64)):
        name, element = _ATOMS[i % 4]
        lines.append("ATOM  %5d %-4s %3s %1s%4d    %8.3f%8.3f%8.3f%6.2f%6.2f          %2s" % (
            i + 1, (" " + name) if len(name) < 4 else name, "GLY", "A", i // 4 + 1, x, y, z, 1.0, 5.0, element))
    n_atoms = len(coords)
    bonds = {k: [] for k in range(1, n_atoms + 1)}
    for r in range(n_atoms // 4):
        n, ca, c, o = (4 * r + k for k in (1, 2, 3, 4))
        for a, b in ((n, ca), (ca, c), (c, o)):
            bonds[a].append(b)
            bonds[b].append(a)
        if r > 0:
            bonds[4 * r - 1].append(n)
            bonds[n].append(4 * r - 1)
    for a in range(1, n_atoms + 1):
        lines.append("CONECT%5d" % a + "".join("%5d" % b for b in sorted(bonds[a])))
    lines.append("END")
    return "\n".join(lines) + "\n"


def write_coords_to_pdb(coords: np.ndarray, out_fname: str) -> str:
    with open(out_fname, "w") as f:
        f.write(pdb_text(coords))
    return out_fname


def _as_angle_array(dists_and_angles) -> np.ndarray:
    """DataFrame with the COLS columns (the reference's input) or an [l,8] array in COLS order."""
    if hasattr(dists_and_angles, "columns"):
        missing = [c for c in REQUIRED_DIHEDRALS if c not in dists_and_angles.columns]
        assert not missing, f"missing dihedral columns {missing}"
        unknown = [c for c in dists_and_angles.columns if c not in COLS]
        if unknown:
            raise ValueError(f"Unrecognized angle: {unknown[0]}")
        return np.stack([np.asarray(dists_and_angles[c], dtype=np.float32) for c in COLS], axis=1)
    arr = np.asarray(dists_and_angles, dtype=np.float32)
    assert arr.ndim == 2 and arr.shape[1] == 8, arr.shape
    return arr


def coords_for_chains(chains: Sequence, center_coords: bool = True, device=None) -> List[np.ndarray]:
    """Batched NeRF for many chains of different length: one padded launch, trimmed results [4 l_i, 3]."""
    device = device or f"cuda:{GPU_ID}"
    arrs = [_as_angle_array(c) for c in chains]
    L = max(a.shape[0] for a in arrs)
    batch = np.zeros((len(arrs), L, 8), dtype=np.float32)
    for i, a in enumerate(arrs):
        batch[i, :a.shape[0]] = a
    lengths = torch.tensor([a.shape[0] for a in arrs], dtype=torch.int32)
    coords = backbone_from_angles(torch.from_numpy(batch).to(device), lengths.to(device), center_coords).cpu().numpy()
    return [coords[i, :a.shape[0]].reshape(-1, 3) for i, a in enumerate(arrs)]


def create_new_chain_nerf(out_fname: str, dists_and_angles, angles_to_set=None, dists_to_set=None,
                          center_coords: bool = True) -> str:
    """One chain -> one PDB file; returns the path, or "" when NaNs appear (reference lines 322-386)."""
    coords = coords_for_chains([dists_and_angles], center_coords)[0]
    if np.any(np.isnan(coords)):
        print(f"Found NaN values, not writing pdb file {out_fname}")
        return ""
    return write_coords_to_pdb(coords, out_fname)


def write_preds_pdb_folder(final_sampled: Sequence, outdir: str, basename_prefix: str = "generated_") -> List[str]:
    """All sampled chains in one NeRF launch, one PDB file each (reference lines 388-408)."""
    os.makedirs(outdir, exist_ok=True)
    files = []
    for i, coords in enumerate(coords_for_chains(final_sampled)):
        fname = os.path.join(outdir, f"{basename_prefix}{i}.pdb")
        if np.any(np.isnan(coords)):
            print(f"Found NaN values, not writing pdb file {fname}")
            files.append("")
        else:
            files.append(write_coords_to_pdb(coords, fname))
    return files


def load_sampled_angles(path: str = None, last_step_only: bool = True) -> List[np.ndarray]:
    """The structure sampler's pickle: list of [T, l_i, 8] trajectories (or [l_i, 8]) -> list of [l_i, 8]."""
    with open(path or GENERATED_DATA, "rb") as f:
        sampled = pickle.load(f)
    return [np.asarray(s[-1] if (np.ndim(s) == 3 and last_step_only) else s, dtype=np.float32) for s in sampled]


if __name__ == "__main__":
    torch.cuda.set_device(GPU_ID)
    print("Loading Angles")
    chains = load_sampled_angles()
    print("Creating PDBs")
    write_preds_pdb_folder(chains, OUTPUT_FODLER)
