"""Amino-acid denoiser under discrete (BLOSUM / uniform) diffusion.  Same constructor
arguments, ``forward`` signature, attribute names and ``state_dict`` keys as the reference's
sequence_model/model.py:156-253; executed as gfx950 HIP kernels (GPU only).

Reference quirks that are behaviour and therefore kept: the receptor goes
through ``ligand_feature_emb`` (``receptor_feature_emb`` is dead weight that still lives in the
checkpoint); the timestep embedding is ADDED to both angle embeddings and also drives the final
``decoder_normalize`` block; training feeds t/T, sampling feeds the raw integer step.
"""
import os
from typing import List

import torch
from torch import nn
from torch.nn import functional as F

from .. import bert, ops
from ..blocks import (BertEmbeddings, GaussianFourierProjection, Predictor, SELayer, flat2d,
                      require_gpu)
from ..training import adamw
from .utils import BlosumTransition, PredefinedNoiseScheduleDiscrete, elbo_loss

AA_VOCAB = "ACDEFGHIKLMNPQRSTVWY"
MASKED_LOSS = os.environ.get("E3D_MASKED_LOSS", "1") == "1"   # 0: the reference's boolean-mask indexing (host syncs per step)


def _is_plain_cross_entropy(fn):
    """nn.CrossEntropyLoss() as the reference's train script builds it (sequence_model/train_model.py): mean reduction, no
    class weights, no label smoothing -- the form ``get_loss`` can take as a masked mean."""
    return (type(fn) is nn.CrossEntropyLoss and fn.weight is None and fn.reduction == "mean" and fn.ignore_index == -100
            and float(getattr(fn, "label_smoothing", 0.0)) == 0.0)


class ConditionalBertForDiffusionBase(nn.Module):
    def __init__(self, encoder_config, decoder_config, feature_size: int) -> None:
        super().__init__()
        self.encoder_config = encoder_config
        self.decoder_config = decoder_config
        self.timestep_projector = GaussianFourierProjection(decoder_config.hidden_size)
        self.ligand_seq_embedding = BertEmbeddings(20, encoder_config)
        self.ligand_angle_embedding = BertEmbeddings(8, encoder_config)
        self.ligand_feature_emb = SELayer(encoder_config)
        self.receptor_seq_embedding = BertEmbeddings(20, encoder_config)
        self.receptor_angle_embedding = BertEmbeddings(8, encoder_config)
        self.receptor_feature_emb = SELayer(encoder_config)   # never used by forward (reference quirk)
        self.decoder = bert.BertEncoder(decoder_config)
        self.decoder_normalize = SELayer(decoder_config)
        self.amino_acid_predictor = Predictor(decoder_config.hidden_size, feature_size)
        self.initialize_weights()

    def initialize_weights(self):
        """Xavier-uniform weights / zero biases on every nn.Linear, then zero only
        decoder_normalize.adaLN_modulation[0] (reference model.py:183-198)."""
        for module in self.modules():
            if isinstance(module, nn.Linear):
                nn.init.xavier_uniform_(module.weight)
                if module.bias is not None:
                    nn.init.constant_(module.bias, 0)
        first = self.decoder_normalize.adaLN_modulation[0]
        nn.init.constant_(first.weight, 0)
        nn.init.constant_(first.bias, 0)

    def forward(self, timestep, noised_ligand_seq, ligand_angle, ligand_attention_masks,
                receptor_seq, receptor_angle, receptor_attention_masks,
                ligand_pos_ids=None, receptor_pos_ids=None):
        """Logits [B,L,20] (reference model.py:200-237)."""
        require_gpu(timestep, noised_ligand_seq, ligand_angle, ligand_attention_masks, receptor_seq,
                    receptor_angle, receptor_attention_masks)
        B, L = noised_ligand_seq.shape[:2]
        Lr = receptor_seq.shape[1]
        ops.reset_absmax(noised_ligand_seq.device)   # |Q|, |K| bounds of the attention calls: fresh per forward
        lig_mask = ligand_attention_masks.contiguous().float()
        rec_mask = receptor_attention_masks.contiguous().float()
        temb = self.timestep_projector(timestep.squeeze(dim=-1)).contiguous()            # [B,H]
        lig_seq = self.ligand_seq_embedding.run(flat2d(noised_ligand_seq))
        lig_ang = self.ligand_angle_embedding.run(flat2d(ligand_angle), post_add=temb, rows_per_add=L)
        lig = self.ligand_feature_emb.run(lig_seq, lig_ang, lig_mask, B, L)
        rec_seq = self.receptor_seq_embedding.run(flat2d(receptor_seq))
        rec_ang = self.receptor_angle_embedding.run(flat2d(receptor_angle), post_add=temb, rows_per_add=Lr)
        rec = self.ligand_feature_emb.run(rec_seq, rec_ang, rec_mask, B, Lr)
        x = bert.run_encoder(self.decoder, lig, lig_mask, B, L, enc=rec, enc_mask=rec_mask, Lk=Lr)
        x = self.decoder_normalize.run(x, temb, lig_mask, B, L)
        return self.amino_acid_predictor.run(x).view(B, L, -1)


def onehot_to_index(onehot: torch.Tensor) -> torch.Tensor:
    """[.., C] one-hot (all-zero rows = padding) -> int32 class index, -1 for all-zero rows."""
    idx = onehot.argmax(dim=-1).to(torch.int32)
    return torch.where(onehot.sum(dim=-1) != 0, idx, torch.full_like(idx, -1))


class PeptideDiff(ConditionalBertForDiffusionBase):
    """Training wrapper (reference model.py:256-450 minus Lightning logging): discrete forward
    noising, losses, optimizer recipe."""

    def __init__(self, encoder_config, decoder_config, feature_names: List[str], loss_func,
                 noise_schedule, timesteps, max_epochs: int = 1, lr_scheduler=None,
                 l2_lambda: float = 0.0, steps_per_epoch: int = 250, learning_rate: float = 5e-5, **kwargs):
        super().__init__(encoder_config, decoder_config, len(feature_names))
        self.noise_schedule = noise_schedule
        self.timesteps = timesteps
        self.aa_transition_model = BlosumTransition(x_classes=20)
        self.discrete_noise_schedule = PredefinedNoiseScheduleDiscrete(noise_schedule=noise_schedule,
                                                                       timesteps=timesteps)
        self.loss_function = loss_func
        self.lr, self.l2_lambda, self.lr_scheduler = learning_rate, l2_lambda, lr_scheduler
        self.max_epochs, self.steps_per_epoch = max_epochs, steps_per_epoch
        self.valid_epoch_losses, self.train_epoch_losses = [], []

    def apply_aa_noise(self, ligand_seq, t_int, u=None):
        """x_t ~ Cat(Qtb[b] @ onehot(x_0)) per residue; all-zero (padding) rows -> class 0
        (reference model.py:291-311).  One HIP launch over all B*L rows instead of the reference's
        per-row Python multinomial loop; ``u`` injects the uniforms (default torch.rand on device)."""
        require_gpu(ligand_seq)
        B, L, C = ligand_seq.shape
        t_float = t_int / self.timesteps
        alpha_t_bar = self.discrete_noise_schedule.get_alpha_bar(t_normalized=t_float)
        qtb = self.aa_transition_model.get_Qt_bar(alpha_t_bar, device=ligand_seq.device).contiguous()
        if u is None:
            u = torch.rand(B, L, device=ligand_seq.device)
        idx = ops.discrete_q_sample(onehot_to_index(ligand_seq).contiguous(), qtb, u.contiguous().float())
        return F.one_hot(idx.long(), num_classes=C).float()

    def get_loss(self, batch, t_norm, noised_ligand_seq):
        """CE on noised positions + "elbo"; also CE on kept positions and two rates, returned in
        the reference's order (model.py:313-345)."""
        ligand_mask = batch["ligand_attn_mask"].bool()
        true_idx = batch["ligand_seq"].argmax(dim=-1)
        noised_idx = noised_ligand_seq.argmax(dim=-1)
        noised_mask = noised_idx != true_idx
        pred_aa = self.forward(t_norm, noised_ligand_seq, batch["ligand_angles"], batch["ligand_attn_mask"],
                               batch["receptor_seq"], batch["receptor_angles"], batch["receptor_attn_mask"])
        n_lig = ligand_mask.sum()
        kept = ligand_mask & (~noised_mask)
        if MASKED_LOSS and _is_plain_cross_entropy(self.loss_function):
            # The same six numbers without boolean-mask indexing (whose data-dependent result sizes cost a device-to-host
            # synchronisation each, in the middle of every training step): means over the noised / kept positions as
            # masked sums divided by counts.  An empty selection gives 0 / 0 = NaN, as the indexed means do.
            zero = torch.zeros((), dtype=pred_aa.dtype, device=pred_aa.device)

            def masked_mean(values, mask):
                return torch.where(mask, values, zero).sum() / mask.sum()

            aa_noise_rate = ((noised_idx == true_idx) & ligand_mask).sum() / n_lig
            aa_recovery_rate = ((pred_aa.argmax(dim=-1) == true_idx) & ligand_mask).sum() / n_lig
            nll = -F.log_softmax(pred_aa, dim=-1).gather(-1, true_idx.unsqueeze(-1)).squeeze(-1)      # CE per position
            aa_noised_loss = masked_mean(nll, noised_mask)
            aa_all_loss = masked_mean(nll, kept)
            # elbo_loss(pred[noised], onehot[noised]) (utils.py): mean row entropy + KL "batchmean" = sums over the rows / rows
            p_model = F.softmax(pred_aa, dim=-1)
            logp_model = F.log_softmax(pred_aa + 1e-6, dim=-1)
            p_target = F.softmax(batch["ligand_seq"], dim=-1)
            kl_rows = (p_target * (p_target.log() - logp_model)).sum(dim=-1)
            entropy_rows = -(p_model * logp_model).sum(dim=-1)
            elbo = masked_mean(entropy_rows, noised_mask) + masked_mean(kl_rows, noised_mask)
            return aa_noised_loss + elbo, elbo, aa_noised_loss, aa_all_loss, aa_recovery_rate, aa_noise_rate
        aa_noise_rate = (noised_idx[ligand_mask] == true_idx[ligand_mask]).sum() / n_lig
        aa_recovery_rate = (pred_aa.argmax(dim=-1)[ligand_mask] == true_idx[ligand_mask]).sum() / n_lig
        aa_noised_loss = self.loss_function(pred_aa[noised_mask].view(-1, 20), true_idx[noised_mask].view(-1))
        aa_all_loss = self.loss_function(pred_aa[kept].view(-1, 20), true_idx[kept].view(-1))
        elbo = elbo_loss(pred_aa[noised_mask], batch["ligand_seq"][noised_mask])
        return aa_noised_loss + elbo, elbo, aa_noised_loss, aa_all_loss, aa_recovery_rate, aa_noise_rate

    def _draw_and_score(self, batch):
        B = batch["ligand_seq"].shape[0]
        t_int = torch.randint(0, self.timesteps + 1, size=(B, 1), device=batch["ligand_seq"].device).float()
        noised = self.apply_aa_noise(batch["ligand_seq"], t_int)
        return self.get_loss(batch, t_int / self.timesteps, noised)

    def training_step(self, batch, batch_idx=0):
        return self._draw_and_score(batch)[0]

    @torch.no_grad()
    def validation_step(self, batch, batch_idx=0):
        return torch.mean(self._draw_and_score(batch)[0])

    def configure_optimizers(self):
        """AdamW + optional schedule (reference model.py:405-450); LinearWarmup counts epochs."""
        optim = adamw(self.parameters(), lr=self.lr, weight_decay=self.l2_lambda)
        retval = {"optimizer": optim}
        if self.lr_scheduler == "OneCycleLR":
            retval["lr_scheduler"] = {
                "scheduler": torch.optim.lr_scheduler.OneCycleLR(
                    optim, max_lr=1e-2, epochs=self.max_epochs, steps_per_epoch=self.steps_per_epoch),
                "interval": "step"}
        elif self.lr_scheduler == "LinearWarmup":
            warmup, total = int(self.max_epochs * 0.1), self.max_epochs

            def lr_lambda(step):
                if step < warmup:
                    return float(step) / float(max(1, warmup))
                return max(0.0, float(total - step) / float(max(1, total - warmup)))

            retval["lr_scheduler"] = {"scheduler": torch.optim.lr_scheduler.LambdaLR(optim, lr_lambda),
                                      "interval": "epoch"}
        elif self.lr_scheduler:
            raise ValueError(f"Unknown lr scheduler {self.lr_scheduler}")
        return retval
