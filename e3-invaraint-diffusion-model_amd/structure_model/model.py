"""Angle-space denoiser: pocket encoder + peptide decoder predicting Gaussian noise on the 8
backbone angles.  Same constructor arguments, ``forward`` signature, attribute names and
``state_dict`` keys as the reference's structure_model/model.py:157-231 (so its checkpoints
load strictly), executed as gfx950 HIP kernels (GPU only).
"""
import functools
from typing import List

import torch
from torch import nn

from .. import bert, ops
from ..blocks import (BertEmbeddings, GaussianFourierProjection, Predictor, SELayer, flat2d,
                      require_gpu)
from ..training import adamw
import os

from .utils import elementwise_form, radian_l1_loss, radian_smooth_l1_loss

MASKED_LOSS = os.environ.get("E3D_MASKED_LOSS", "1") == "1"   # 0: the reference's indexed form (a host sync per step)


class ReceptorCache:
    """Everything the decoder needs from the pocket: it depends on neither the timestep nor the
    noised ligand (reference model.py:191-200), so a sampler computes it once per batch instead
    of once per reverse step (41 % of per-step FLOPs + the cross K/V projections)."""

    def __init__(self, encoder_states, cross_kv, mask, B, L):
        self.encoder_states, self.cross_kv, self.mask, self.B, self.L = encoder_states, cross_kv, mask, B, L


class ConditionalBertForDiffusionBase(nn.Module):
    def __init__(self, encoder_config, decoder_config, feature_size: int) -> None:
        super().__init__()
        self.encoder_config = encoder_config
        self.decoder_config = decoder_config
        # pocket side
        self.receptor_seq_emb = BertEmbeddings(20, encoder_config)
        self.receptor_angle_emb = BertEmbeddings(feature_size, encoder_config)
        self.receptor_emb = SELayer(encoder_config)
        self.encoder = bert.BertEncoder(encoder_config)
        # peptide side
        self.ligand_angle_emb = BertEmbeddings(feature_size, decoder_config)
        self.timestep_projector = GaussianFourierProjection(decoder_config.hidden_size)
        self.timestep_emb = SELayer(decoder_config)
        self.decoder = bert.BertEncoder(decoder_config)
        self.angles_predictor = Predictor(decoder_config.hidden_size, feature_size)

    # -- the two halves of forward ------------------------------------------------------------
    def encode_receptor(self, receptor_seq, receptor_angles, receptor_attention_masks,
                        project_cross_kv=True) -> ReceptorCache:
        """reference model.py:191-200 (+ the decoder layers' cross K/V projections)."""
        require_gpu(receptor_seq, receptor_angles, receptor_attention_masks)
        B, L = receptor_angles.shape[:2]
        ops.reset_absmax(receptor_angles.device)   # |Q|, |K| bounds of the attention calls: fresh per batch / chain
        mask = receptor_attention_masks.contiguous().float()
        ang = self.receptor_angle_emb.run(flat2d(receptor_angles))
        seq = self.receptor_seq_emb.run(flat2d(receptor_seq))
        x = self.receptor_emb.run(ang, seq, mask, B, L)
        x = bert.run_encoder(self.encoder, x, mask, B, L)
        kv = None
        if project_cross_kv:
            kv = [bert.project_cross_kv(layer.crossattention, x) for layer in self.decoder.layer]
        return ReceptorCache(x, kv, mask, B, L)

    def timestep_modulation(self, timesteps):
        """The part of the decoder that depends on the timestep alone (reference model.py:205-207: Fourier features ->
        SELayer.adaLN_modulation): [n] timesteps -> [n,6H].  A sampler computes it once for its whole chain and hands
        ``decode`` one row per step (``mod``) instead of ~12 small launches per step."""
        temb = self.timestep_projector(timesteps.reshape(-1)).contiguous()
        return self.timestep_emb.modulation(temb)

    def decode(self, timestep, noised_ligand_angles, ligand_attention_masks, receptor: ReceptorCache, mod=None):
        """reference model.py:202-214.  ``mod`` ([1,6H] or [B,6H]): ``timestep_modulation`` of this step's timestep,
        if the caller has it already (``timestep`` is then not read)."""
        require_gpu(timestep, noised_ligand_angles, ligand_attention_masks)
        B, L = noised_ligand_angles.shape[:2]
        mask = ligand_attention_masks.contiguous().float()
        x = self.ligand_angle_emb.run(flat2d(noised_ligand_angles))
        if mod is None:
            mod = self.timestep_modulation(timestep.squeeze(dim=-1))             # [B,6H]
        x = self.timestep_emb.run(x, None, mask, B, L, mod=mod)
        x = bert.run_encoder(self.decoder, x, mask, B, L, enc=receptor.encoder_states,
                             enc_mask=receptor.mask, Lk=receptor.L, cross_kv=receptor.cross_kv)
        return self.angles_predictor.run(x).view(B, L, -1)

    def forward(self, timestep, noised_ligand_angles, ligand_attention_masks,
                receptor_seq, receptor_angles, receptor_attention_masks,
                ligand_pos_ids=None, receptor_pos_ids=None):
        """Predicted noise [B,L,8].  ``*_pos_ids`` are accepted and ignored, as in the reference
        (model.py:185-186: computed, never used)."""
        rec = self.encode_receptor(receptor_seq, receptor_angles, receptor_attention_masks,
                                   project_cross_kv=False)
        return self.decode(timestep, noised_ligand_angles, ligand_attention_masks, rec)


class ConditionalBertForDiffusion(ConditionalBertForDiffusionBase):
    """Training wrapper (reference model.py:233-403 minus the Lightning logging hooks):
    per-feature wrapped-angle loss terms and the AdamW / LinearWarmup optimizer recipe."""
    diheral_loss_func = radian_l1_loss
    angle_loss_func = functools.partial(radian_smooth_l1_loss, beta=torch.pi / 10)

    def __init__(self, encoder_config, decoder_config, feature_names: List[str], loss_func: List,
                 epochs: int = 1, lr_scheduler=None, l2_lambda: float = 0.0,
                 steps_per_epoch: int = 250, learning_rate: float = 5e-5, **kwargs):
        super().__init__(encoder_config, decoder_config, len(feature_names))
        self.steps_per_epoch = steps_per_epoch
        self.learning_rate = learning_rate
        self.feature_names = feature_names
        self.lr_scheduler = lr_scheduler
        self.loss_func = loss_func
        self.epochs = epochs
        self.l2_lambda = l2_lambda
        self.train_epoch_losses, self.valid_epoch_losses = [], []
        self.train_epoch_counter = 0

    def _get_loss_terms(self, batch) -> torch.Tensor:
        """One loss per angle feature over the un-padded ligand positions (reference model.py:266-303)."""
        known_noise = batch["known_noise"]
        predicted_noise = self.forward(
            timestep=batch["timestep"], noised_ligand_angles=batch["noised_ligand_angle"],
            ligand_attention_masks=batch["ligand_attn_mask"], receptor_seq=batch["receptor_seq"],
            receptor_angles=batch["receptor_angles"], receptor_attention_masks=batch["receptor_attn_mask"])
        assert known_noise.shape == predicted_noise.shape, f"{known_noise.shape} != {predicted_noise.shape}"
        return self.loss_terms_from_prediction(predicted_noise, known_noise, batch["ligand_attn_mask"])

    def loss_terms_from_prediction(self, predicted_noise, known_noise, ligand_attn_mask):
        """``fn(pred[rows, cols, i], noise[rows, cols, i])`` per feature over ``rows, cols = torch.where(mask)``
        (reference model.py:290-303).  For this package's own loss functions the mean over the selected positions is
        taken as a masked sum / count: the same value without the index lists, whose data-dependent length costs a
        device-to-host synchronisation in the middle of every training step (the GPU drains before the backward pass can
        be enqueued, and the step cannot be captured in a graph).  Unknown callables take the indexed path."""
        n_feat = known_noise.shape[-1]
        fns = [self.loss_func[i] if isinstance(self.loss_func, list) else self.loss_func for i in range(n_feat)]
        elems = [elementwise_form(fn) for fn in fns]
        if all(e is not None for e in elems) and MASKED_LOSS:
            sel = (ligand_attn_mask != 0).unsqueeze(-1)                       # [B, L, 1]
            count = sel.sum().to(predicted_noise.dtype)
            groups = {}                                                       # features sharing a function: one pass
            for i, fn in enumerate(fns):
                key = (fn.func, tuple(sorted(fn.keywords.items()))) if isinstance(fn, functools.partial) else (fn, ())
                groups.setdefault(key, []).append(i)
            terms = [None] * n_feat
            for idx in groups.values():
                if idx == list(range(idx[0], idx[-1] + 1)):                   # a run of features: plain slices
                    pn, kn = predicted_noise[..., idx[0]:idx[-1] + 1], known_noise[..., idx[0]:idx[-1] + 1]
                else:                                                         # (index tensors are made once per device)
                    cache = self.__dict__.setdefault("_e3d_loss_idx", {})
                    it = cache.get((tuple(idx), predicted_noise.device))
                    if it is None:
                        it = cache[(tuple(idx), predicted_noise.device)] = torch.tensor(idx, device=predicted_noise.device)
                    pn, kn = predicted_noise.index_select(-1, it), known_noise.index_select(-1, it)
                e = elems[idx[0]](pn, kn)                                     # [B, L, len(idx)]
                sums = torch.where(sel, e, torch.zeros((), dtype=e.dtype, device=e.device)).sum(dim=(0, 1)) / count
                for j, i in enumerate(idx):
                    terms[i] = sums[j]
            return torch.stack(terms)
        rows, cols = torch.where(ligand_attn_mask)
        terms = []
        for i in range(n_feat):
            terms.append(fns[i](predicted_noise[rows, cols, i], known_noise[rows, cols, i]))
        return torch.stack(terms)

    def training_step(self, batch, batch_idx=0):
        return torch.mean(self._get_loss_terms(batch))

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        return {"val_loss": torch.mean(self._get_loss_terms(batch))}

    def configure_optimizers(self):
        """AdamW(lr, weight_decay=l2_lambda) + optional schedule (reference model.py:361-403).
        LinearWarmup counts EPOCHS (warm-up = 10 % of ``epochs``), as the reference does."""
        optim = adamw(self.parameters(), lr=self.learning_rate, weight_decay=self.l2_lambda)
        retval = {"optimizer": optim}
        if self.lr_scheduler == "OneCycleLR":
            retval["lr_scheduler"] = {
                "scheduler": torch.optim.lr_scheduler.OneCycleLR(
                    optim, max_lr=1e-2, epochs=self.epochs, steps_per_epoch=self.steps_per_epoch),
                "interval": "step"}
        elif self.lr_scheduler == "LinearWarmup":
            warmup, total = int(self.epochs * 0.1), self.epochs

            def lr_lambda(step):  # transformers.get_linear_schedule_with_warmup semantics
                if step < warmup:
                    return float(step) / float(max(1, warmup))
                return max(0.0, float(total - step) / float(max(1, total - warmup)))

            retval["lr_scheduler"] = {"scheduler": torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda),
                                      "interval": "epoch"}
        elif self.lr_scheduler:
            raise ValueError(f"Unknown lr scheduler {self.lr_scheduler}")
        return retval
