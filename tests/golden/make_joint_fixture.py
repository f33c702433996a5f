#!/usr/bin/env python3
"""Generate tests/golden/joint_handover.pt by RUNNING the reference's ``load_generated_angles``
(sequence_model/sample_by_generated_angles.py:54-66) in the build container.

    python tests/golden/make_joint_fixture.py        # needs /root/reference

The module has top-level CUDA / torch_geometric-dependent imports, so the one FunctionDef is ast-extracted and
exec'd with the names it reads (CONFIG, pickle, np, torch, pd) injected.  Input: a pickle of per-ligand [l_i, 8]
float32 arrays -- what structure_model/sample.py writes once its "extract last time step" line (sample.py:233)
is enabled, the form line 59's 2-D ``np.pad`` requires.  Only inputs and outputs are stored.
"""
import ast
import os
import pickle
import tempfile

import numpy as np
import pandas as pd
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = "/root/reference/sequence_model/sample_by_generated_angles.py"


def main():
    tree = ast.parse(open(SRC).read())
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "load_generated_angles"]
    assert len(body) == 1
    cfg = {"max_seq_len": 128, "batch_size": 3}
    env = {"CONFIG": cfg, "pickle": pickle, "np": np, "torch": torch, "pd": pd}
    exec(compile(ast.Module(body=body, type_ignores=[]), SRC, "exec"), env)
    g = torch.Generator().manual_seed(11)
    lengths = [5, 30, 17, 128, 1, 64, 9]            # ragged, incl. a full-length and a single-residue ligand
    arrays = [(torch.rand(l, 8, generator=g) * 2 * np.pi - np.pi).numpy().astype(np.float32) for l in lengths]
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "output.pkl")
        with open(path, "wb") as f:
            pickle.dump(arrays, f)
        chunks = env["load_generated_angles"](path)
    assert [tuple(c.shape) for c in chunks] == [(3, 128, 8), (3, 128, 8), (1, 128, 8)]
    out = {"config": cfg, "lengths": lengths, "arrays": [torch.from_numpy(a) for a in arrays],
           "chunks": [c.clone() for c in chunks]}
    torch.save(out, os.path.join(HERE, "joint_handover.pt"))
    print("wrote joint_handover.pt", os.path.getsize(os.path.join(HERE, "joint_handover.pt")), "bytes")


if __name__ == "__main__":
    main()
