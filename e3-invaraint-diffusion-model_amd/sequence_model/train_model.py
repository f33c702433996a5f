"""Training entry point of the sequence model (reference sequence_model/train_model.py): same
constants-as-config and saved ``state_dict`` format; ``pl.Trainer`` is replaced by
``e3diff_amd.training.fit``.  BASELINE config 4 = this script under
``torchrun --nproc-per-node 8`` (per-rank batch 64, gradient all-reduce over RCCL/xGMI; the
never-used ``receptor_feature_emb`` parameters are handled by the static bucket plan).

Run as ``python train_model.py`` from this directory after editing the constants.
"""
if __package__ in (None, ""):
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    import __graft_entry__ as _g
    _g.load_package()
    __package__ = "e3diff_amd.sequence_model"

import os

import torch
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler

from .. import training
from ..bert import BertConfig
from .dataset import LigandBindingSiteDataset
from .model import PeptideDiff

MODEL_PATH = ""
DATA_FILE = "./data/biolip.pt"
GPU_ID = [0]
NUM_THREAD = 16

CONFIG = {
    "pocket_ext": 4,
    "timesteps": 50,
    "max_seq_len": 128,
    "noise_schedule": "cosine",

    "num_heads": 12,
    "dropout_p": 0.1,
    "hidden_size": 768,
    "num_hidden_layers": 6,
    "intermediate_size": 1024,
    "position_embedding_type": "relative_key",

    "lr": 5e-5,
    "l2_norm": 0.1,
    "loss": "smooth_l1",
    "gradient_clip": 1.0,
    "lr_scheduler": "LinearWarmup",

    "min_epochs": 100,
    "max_epochs": 150,
    "batch_size": 64,
}


def get_dataloader(file_path, records=None):
    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    out = []
    for split, shuffle in (("train", True), ("validation", False)):
        ds = LigandBindingSiteDataset(file_path, split, CONFIG["max_seq_len"], CONFIG["pocket_ext"], records=records)
        sampler = DistributedSampler(ds, num_replicas=world, rank=rank, shuffle=shuffle) if world > 1 else None
        out.append(DataLoader(dataset=ds, batch_size=CONFIG["batch_size"], shuffle=shuffle and sampler is None,
                              sampler=sampler, num_workers=NUM_THREAD))
    return tuple(out)


def build_configs():
    common = dict(max_position_embeddings=CONFIG["max_seq_len"], num_attention_heads=CONFIG["num_heads"],
                  hidden_size=CONFIG["hidden_size"], intermediate_size=CONFIG["intermediate_size"],
                  num_hidden_layers=CONFIG["num_hidden_layers"],
                  position_embedding_type=CONFIG["position_embedding_type"],
                  hidden_dropout_prob=CONFIG["dropout_p"], attention_probs_dropout_prob=CONFIG["dropout_p"],
                  use_cache=False)
    return BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True)


def train_model(encoder_config, decoder_config, train_dataloader, val_dataloader, max_steps=None):
    model = PeptideDiff(
        encoder_config=encoder_config, decoder_config=decoder_config,
        feature_names=LigandBindingSiteDataset.feature_names, max_epochs=CONFIG["max_epochs"],
        lr_scheduler=CONFIG["lr_scheduler"], l2_lambda=CONFIG["l2_norm"],
        steps_per_epoch=len(train_dataloader), learning_rate=CONFIG["lr"],
        loss_func=torch.nn.CrossEntropyLoss(), noise_schedule=CONFIG["noise_schedule"],
        timesteps=CONFIG["timesteps"])
    print(f"Model has {sum(p.numel() for p in model.parameters() if p.requires_grad)} trainable parameters")
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = f"cuda:{GPU_ID[local_rank % len(GPU_ID)] if 'LOCAL_RANK' not in os.environ else local_rank}"
    print("Start training")
    history = training.fit(model, train_dataloader, val_dataloader, min_epochs=CONFIG["min_epochs"],
                           max_epochs=CONFIG["max_epochs"], gradient_clip=CONFIG["gradient_clip"], device=device,
                           checkpoint_path="./best_val_model.pt", checkpoint_mode="max", max_steps=max_steps)
    return history, model


if __name__ == "__main__":
    torch.set_num_threads(NUM_THREAD)
    print("Loading Data")
    train_dataloader, val_dataloader = get_dataloader(DATA_FILE)
    encoder_config, decoder_config = build_configs()
    history, model = train_model(encoder_config, decoder_config, train_dataloader, val_dataloader)
    if int(os.environ.get("RANK", "0")) == 0 and MODEL_PATH:
        torch.save(model.state_dict(), MODEL_PATH)
