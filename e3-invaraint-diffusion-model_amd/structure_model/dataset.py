"""``biolip.pt`` -> padded per-complex tensors, and the forward noising q(x_t | x_0).

The tensor layout of an item is the drop-in contract with the reference
(structure_model/dataset.py:119-132, 203-209): same keys, shapes, dtypes, zero padding to
``max_len``, 80/10/10 split after ``random.seed(0); shuffle``.  ``NoisedAnglesDataset`` keeps the
reference's per-item host path (DataLoader workers); ``noise_batch_on_device`` is the batched HIP
path for training loops that already hold the batch in HBM.
"""
import random
from typing import Dict, Optional

import numpy as np
import torch
import torch.nn.functional as F
from torch.utils.data import Dataset

from .. import biolip
from .utils import CosineTables, modulo_with_wrapped_range

RANDOM_SEED = 0
AA_VOCAB = "ACDEFGHIKLMNPQRSTVWY"
SS_VOCAB = "HBEGITS-"


def _one_hot(symbols, vocab):
    idx = torch.tensor([vocab.index(ch) for ch in symbols])
    return F.one_hot(idx, num_classes=len(vocab)).float()


def dilate_pocket_mask(pocket_mask: torch.Tensor, ext: int) -> torch.Tensor:
    """Pocket mask OR its copies rolled by +/-ext with the wrapped-around end cleared
    (reference dataset.py:104-109; note: a roll by ``ext``, not a fill of the gap)."""
    left = torch.roll(pocket_mask, ext)
    left[0] = False
    right = torch.roll(pocket_mask, -ext)
    right[-1] = False
    return pocket_mask | left | right


class LigandBindingSiteDataset(Dataset):
    feature_names = ["phi", "psi", "omega", "dihedral_o", "tau", "CA:C:1N", "1C:N:CA", "CA:C:O"]
    with_ligand_seq = False  # the sequence model's dataset adds "ligand_seq"

    def __init__(self, filepath: str, split: Optional[str], max_len: int = 64, pocket_ext: int = 1,
                 records=None) -> None:
        """filepath: a ``biolip.pt`` (torch.save of list[dict], clean_data/data_preprocessing.py:880-892);
        split: "train" | "validation" | "test" | None; ``records`` bypasses the file (synthetic data)."""
        super().__init__()
        self.max_len, self.pocket_ext = max_len, pocket_ext
        if records is None:
            print(f"Loading data from {filepath}")
            records = biolip.load_records(filepath)   # weights_only=True: nothing in the file is executed
        self.data = [dict(r) for r in records]
        for d in self.data:
            d["amino_acid"] = _one_hot("".join(d["amino_acid"]), AA_VOCAB)
            d["secondary_structure"] = _one_hot("".join(d["secondary_structure"]), SS_VOCAB)
        self._split_data(split)

    def _split_data(self, split_name):
        random.seed(RANDOM_SEED)
        random.shuffle(self.data)
        if split_name is None:
            return
        n = len(self.data)
        cut, tenth = int(n * 0.8), int(n * 0.1)
        bounds = {"train": (0, cut), "validation": (cut, cut + tenth), "test": (cut + tenth, n)}
        if split_name in bounds:
            lo, hi = bounds[split_name]
            self.data = self.data[lo:hi]

    def _pad(self, rows):
        if rows.shape[0] > self.max_len:
            raise RuntimeError("Length exceed")
        return F.pad(rows, (0, 0, 0, self.max_len - rows.shape[0]), mode="constant", value=0)

    def __len__(self) -> int:
        return len(self.data)

    def get_structure_id(self, index):
        return self.data[index]["structure_ids"]

    def _length_mask(self, n):
        m = torch.zeros(size=(self.max_len,))
        m[:n] = 1.0
        return m

    def __getitem__(self, index):
        if not 0 <= index < len(self):
            raise IndexError("Index out of range")
        rec = self.data[index]
        lig = rec["ligand_mask"]
        pocket = dilate_pocket_mask(rec["pocket_mask"], self.pocket_ext)
        item = {
            "ligand_angles": self._pad(rec["angle_features"][lig]),
            "ligand_attn_mask": self._length_mask(lig.sum()),
            "ligand_pos_id": 0,
            "receptor_angles": self._pad(rec["angle_features"][pocket]),
            "receptor_attn_mask": self._length_mask(pocket.sum()),
            "receptor_seq": self._pad(rec["amino_acid"][pocket]),
            "receptor_pos_id": 0,
            "ligand_length": lig.sum(),
            "receptor_length": pocket.sum(),
            "structure_ids": rec["structure_ids"],
        }
        if self.with_ligand_seq:
            item["ligand_seq"] = self._pad(rec["amino_acid"][lig])
        return item


class NoisedAnglesDataset(Dataset):
    """Wraps a dataset and adds a random-timestep noised copy of ``ligand_angles``
    (reference dataset.py:134-229).  Padding positions are noised too, as in the reference."""

    def __init__(self, dset: Dataset, timesteps: int = 250) -> None:
        super().__init__()
        self.dset = dset
        self.n_features = len(dset.feature_names)
        self.angular_var_scale = 1.0
        self.timesteps = timesteps
        self.tables = CosineTables(timesteps)
        self.alpha_beta_terms = self.tables.as_dict()

    @property
    def feature_names(self):
        return self.dset.feature_names

    def __len__(self) -> int:
        return len(self.dset)

    def __str__(self) -> str:
        return f"NoisedAnglesDataset({self.dset}, n={len(self)}, cosine-{self.timesteps})"

    def sample_noise(self, vals: torch.Tensor) -> torch.Tensor:
        """N(0, angular_var_scale^2) noise of vals' shape, wrapped into [-pi, pi)."""
        noise = torch.randn_like(vals)
        if self.angular_var_scale != 1.0:
            noise = noise * self.angular_var_scale
        return modulo_with_wrapped_range(noise, -np.pi, np.pi)

    def _add_noise_by_timestep(self, v: torch.Tensor, timestep: torch.Tensor):
        t = timestep.item()
        a = self.alpha_beta_terms["sqrt_alphas_cumprod"][t]
        s = self.alpha_beta_terms["sqrt_one_minus_alphas_cumprod"][t]
        noise = self.sample_noise(v)
        return {"noise": noise,
                "noised_value": modulo_with_wrapped_range(a * v + s * noise, -np.pi, np.pi),
                "sqrt_alphas_cumprod_t": a, "sqrt_one_minus_alphas_cumprod_t": s}

    def __getitem__(self, index: int, use_timestep: Optional[int] = None) -> Dict[str, torch.Tensor]:
        assert 0 <= index < len(self), f"Index {index} out of bounds for {len(self)}"
        item = self.dset.__getitem__(index)
        if use_timestep is None:
            timestep = torch.randint(0, self.timesteps, (1,)).long()
        else:
            timestep = torch.from_numpy(np.clip(np.array([use_timestep]), 0, self.timesteps - 1)).long()
        noised = self._add_noise_by_timestep(item["ligand_angles"], timestep)
        item.update({
            "timestep": timestep,
            "known_noise": noised["noise"],
            "noised_ligand_angle": noised["noised_value"],
            "sqrt_alphas_cumprod_t": noised["sqrt_alphas_cumprod_t"],
            "sqrt_one_minus_alphas_cumprod_t": noised["sqrt_one_minus_alphas_cumprod_t"],
        })
        return item


def noise_batch_on_device(ligand_angles, tables: CosineTables, timestep=None, noise=None):
    """Batched q(x_t | x_0) on the GPU (HIP ``e3d_q_sample_wrap``): the device-side equivalent of
    NoisedAnglesDataset.__getitem__ for a whole [B,L,8] batch.  Returns dict(timestep [B,1],
    known_noise, noised_ligand_angle)."""
    from .. import ops
    B = ligand_angles.shape[0]
    dev = ligand_angles.device
    if timestep is None:
        timestep = torch.randint(0, tables.timesteps, (B, 1), device=dev)
    if noise is None:
        noise = modulo_with_wrapped_range(torch.randn_like(ligand_angles), -np.pi, np.pi)
    x_t = ops.q_sample_wrap(ligand_angles.contiguous().float(), noise.contiguous().float(),
                            timestep.reshape(-1).long().contiguous(),
                            tables.sqrt_alphas_cumprod.to(dev), tables.sqrt_one_minus_alphas_cumprod.to(dev))
    return {"timestep": timestep, "known_noise": noise, "noised_ligand_angle": x_t}
