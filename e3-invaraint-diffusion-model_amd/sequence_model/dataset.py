"""``biolip.pt`` -> padded per-complex tensors for the sequence model: the structure model's
layout plus ``ligand_seq`` f32[L,20] (reference sequence_model/dataset.py:100,118), with
``feature_names`` = the 20 amino-acid letters (reference dataset.py:13)."""
from ..structure_model.dataset import AA_VOCAB, SS_VOCAB, RANDOM_SEED  # noqa: F401
from ..structure_model.dataset import LigandBindingSiteDataset as _AngleDataset


class LigandBindingSiteDataset(_AngleDataset):
    feature_names = list(AA_VOCAB)
    with_ligand_seq = True
