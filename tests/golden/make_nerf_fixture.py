#!/usr/bin/env python3
"""Generate tests/golden/nerf.pt by RUNNING the reference's NeRF builder in the build container.

    python tests/golden/make_nerf_fixture.py        # needs /root/reference

structure_model/create_pdb.py imports biotite at module level (absent here), so the numpy-only
parts -- the module constants, ``NERFBuilder`` and ``place_dihedral`` -- are ast-extracted and exec'd;
the keyword mapping of ``create_new_chain_nerf`` (create_pdb.py:340-375: which angle column feeds which
NERFBuilder argument) is applied here exactly as that function does for the 8 columns the sampler emits.
Nothing of the reference is copied into the repo: only angles in / coordinates out are saved.
"""
import ast
import os
from functools import cached_property
from typing import List, Optional, Sequence, Tuple, Union

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/structure_model/create_pdb.py"


def load_reference_nerf():
    tree = ast.parse(open(SRC).read())
    keep = []
    for node in tree.body:
        if isinstance(node, ast.Assign) and any(isinstance(t, ast.Name) and t.id.endswith(("_LENGTH", "_INIT", "COLS"))
                                                for t in node.targets):
            keep.append(node)
        if isinstance(node, ast.ClassDef) and node.name == "NERFBuilder":
            keep.append(node)
        if isinstance(node, ast.FunctionDef) and node.name == "place_dihedral":
            keep.append(node)
    env = {"np": np, "torch": torch, "pd": pd, "cached_property": cached_property, "Union": Union, "Tuple": Tuple,
           "List": List, "Optional": Optional, "Sequence": Sequence}
    exec(compile(ast.Module(body=keep, type_ignores=[]), SRC, "exec"), env)
    return env


def reference_coords(env, angles, center):
    df = pd.DataFrame(angles, columns=env["COLS"])
    b = env["NERFBuilder"](
        phi_dihedrals=df["phi"], psi_dihedrals=df["psi"], omega_dihedrals=df["omega"],
        oxygen_dihedrals=df["dihedral_o"], bond_angle_ca_c=df["tau"], bond_angle_c_n=df["CA:C:1N"],
        bond_angle_n_ca=df["1C:N:CA"], bond_angle_c_o=df["CA:C:O"])
    return np.asarray(b.centered_cartesian_coords if center else b.cartesian_coords)


if __name__ == "__main__":
    env = load_reference_nerf()
    rng = np.random.default_rng(0)
    cases = []
    for l in (2, 3, 9, 30, 64):
        ang = np.empty((l, 8), dtype=np.float32)
        ang[:, :4] = rng.uniform(-np.pi, np.pi, (l, 4))
        ang[:, 4:] = rng.normal(1.95, 0.1, (l, 4))
        for center in (True, False):
            cases.append({"angles": torch.from_numpy(ang.copy()), "center": center,
                          "coords": torch.from_numpy(reference_coords(env, ang, center))})
    torch.save({"cols": env["COLS"], "cases": cases}, os.path.join(HERE, "nerf.pt"))
    print(f"wrote nerf.pt: {len(cases)} cases")
