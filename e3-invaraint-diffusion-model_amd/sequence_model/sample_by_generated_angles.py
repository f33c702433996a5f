"""Joint structure -> sequence sampling: the sequence model conditioned on ligand angles that
the structure model generated (reference sequence_model/sample_by_generated_angles.py), with the
uniform transition (line 253 there).  ``load_generated_angles`` reads the structure sampler's
pickle; ``angles_from_trajectory`` is the on-device hand-over that skips the pickle round trip
(SURVEY.md section 8(f) rank 1).

Run as ``python sample_by_generated_angles.py`` from this directory, like the reference.
"""
if __package__ in (None, ""):
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    import __graft_entry__ as _g
    _g.load_package()
    __package__ = "e3diff_amd.sequence_model"

import pickle

import numpy as np
import torch

from . import sample as _sample
from .sample import denoise as _denoise, get_dataloader, get_model  # noqa: F401
from .utils import DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete

GPU_ID = 0
DATA_PATH = "./data/biolip.pt"
MODEL_PATH = ""
OUTPUT_PATH = "./data/from_generated_angles/output.pkl"
GENERATED_ANGLE_PATH = "../structure_model/data/output.pkl"

CONFIG = dict(_sample.CONFIG, pocket_ext=4, max_seq_len=128)


def load_generated_angles(file_path, max_seq_len=None, batch_size=None):
    """Pickled list of per-ligand [l_i,8] arrays (the structure sampler's output with the last
    timestep extracted, reference structure_model/sample.py:235) -> zero-padded [n,max_seq_len,8]
    tensor cut into batches (reference lines 54-66)."""
    max_seq_len = max_seq_len or CONFIG["max_seq_len"]
    batch_size = batch_size or CONFIG["batch_size"]
    with open(file_path, "rb") as f:
        arrays = pickle.load(f)
    arrays = [a[-1] if a.ndim == 3 else a for a in arrays]          # accept full [T,l,8] trajectories too
    padded = np.zeros((len(arrays), max_seq_len, arrays[0].shape[-1]), dtype=np.float32)
    for i, a in enumerate(arrays):
        padded[i, :a.shape[0]] = a
    angles = torch.from_numpy(padded)
    return [angles[i:i + batch_size] for i in range(0, len(angles), batch_size)]


def angles_from_trajectory(traj, ligand_mask):
    """Device hand-over: last step of a structure trajectory [T,B,L,8] (or [B,L,8]) with the
    padding rows zeroed, i.e. what load_generated_angles would have produced from the pickle."""
    last = traj[-1] if traj.dim() == 4 else traj
    return last * ligand_mask.to(last.device)[..., None]


def denoise(batch, generated_angles, model, noise_schedule, transition, diverse, **kw):
    """reference lines 196-243: sample.denoise with the ligand angles replaced."""
    return _denoise(batch, model, noise_schedule, transition, diverse, generated_angles=generated_angles, **kw)


if __name__ == "__main__":
    import pandas as pd
    torch.cuda.set_device(GPU_ID)
    _sample.CONFIG.update(CONFIG)
    generated_angles = load_generated_angles(GENERATED_ANGLE_PATH)
    loader = get_dataloader(DATA_PATH)
    model = get_model(len(loader), MODEL_PATH)
    schedule = PredefinedNoiseScheduleDiscrete(CONFIG["noise_schedule"], CONFIG["timesteps"]).to(_sample.DEVICE)
    transition = DiscreteUniformTransition(20)
    cols = ([], [], [], [])
    for idx, batch in enumerate(loader):
        print(f"Generating Batch {idx}")
        for acc, part in zip(cols, denoise(batch, generated_angles[idx], model, schedule, transition, True)):
            acc.extend(part)
    res = pd.DataFrame(zip(*cols), columns=["structure_ids", "true_sequence", "predict_sequence", "recovery_rate"])
    res.to_pickle(OUTPUT_PATH)
    print(res)
