"""Discrete reverse sampler of the sequence model -- entry point and function names of the
reference's sequence_model/sample.py.  The posterior (sample.py:120-139), its mixture with the
predicted x0 distribution and the categorical draw (sample.py:141-179: a Python loop over B*L
rows with a device sync per row in the reference) are ONE HIP launch, ``e3d_discrete_posterior_sample``.

Run as ``python sample.py`` from this directory after editing the constants, like the reference.
"""
if __package__ in (None, ""):  # executed as a script from inside this directory
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    import __graft_entry__ as _g
    _g.load_package()
    __package__ = "e3diff_amd.sequence_model"

import os

import torch
from torch.nn import functional as F
from torch.utils.data import DataLoader

from .. import ops
from ..bert import BertConfig
from .dataset import AA_VOCAB, LigandBindingSiteDataset
from .model import PeptideDiff, onehot_to_index
from .utils import BlosumTransition, DiscreteUniformTransition, PredefinedNoiseScheduleDiscrete  # noqa: F401

GPU_ID = 0
DEVICE = torch.device(f"cuda:{GPU_ID}")
THREAD_NUM = 16
DATA_PATH = "./data/biolip.pt"
MODEL_PATH = ""  # trained state_dict (reference checkpoint key names)
OUTPUT_PATH = "./data/from_generated_angles/output.pkl"

CONFIG = {
    "pocket_ext": 0,
    "timesteps": 50,
    "max_seq_len": 64,
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


def get_dataloader(file_path):
    ds = LigandBindingSiteDataset(file_path, "test", CONFIG["max_seq_len"], CONFIG["pocket_ext"])
    return DataLoader(dataset=ds, batch_size=CONFIG["batch_size"], shuffle=False, num_workers=THREAD_NUM)


def build_configs(cfg=None):
    cfg = cfg or CONFIG
    common = dict(max_position_embeddings=cfg["max_seq_len"], num_attention_heads=cfg["num_heads"],
                  hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                  num_hidden_layers=cfg["num_hidden_layers"],
                  position_embedding_type=cfg["position_embedding_type"],
                  hidden_dropout_prob=cfg["dropout_p"], attention_probs_dropout_prob=cfg["dropout_p"],
                  use_cache=False)
    return BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True)


def get_model(steps_per_epoch, model_path=None) -> PeptideDiff:
    encoder_config, decoder_config = build_configs()
    model = PeptideDiff(
        encoder_config=encoder_config, decoder_config=decoder_config,
        feature_names=LigandBindingSiteDataset.feature_names, max_epochs=CONFIG["max_epochs"],
        lr_scheduler=CONFIG["lr_scheduler"], l2_lambda=CONFIG["l2_norm"], steps_per_epoch=steps_per_epoch,
        learning_rate=CONFIG["lr"], loss_func=torch.nn.CrossEntropyLoss(),
        noise_schedule=CONFIG["noise_schedule"], timesteps=CONFIG["timesteps"])
    path = MODEL_PATH if model_path is None else model_path
    if path:
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    model = model.eval().to(DEVICE)
    print(f"Model has {sum(p.numel() for p in model.parameters() if p.requires_grad)} trainable parameters")
    return model


def generate_discrete_noise(batch_size, length, num_classes=20, device=None):
    """Uniform random one-hot [B,L,C] (reference sample.py:112-116), drawn on the device."""
    device = DEVICE if device is None else device
    idx = torch.randint(0, num_classes, (batch_size, length), device=device)
    return F.one_hot(idx, num_classes).float()


def compute_batched_over0_posterior_distribution(X_t, Q_t, Qsb, Qtb, batch):
    """[N,C,C] posterior table  (x_t Qt^T) * Qsb / (Qtb x_t)  (reference sample.py:120-139).
    API-compatibility helper in plain torch: the sampler below never materialises this tensor
    (it is fused into the HIP kernel)."""
    left = X_t.unsqueeze(-2) @ Q_t.transpose(-1, -2)[batch]
    den = Qtb[batch] @ X_t.unsqueeze(2)
    den = torch.where(den == 0, torch.full_like(den, 1e-6), den)
    return left * Qsb[batch] / den


def sample_p_zs_given_zt_discrete(t, s, noised_data, pred_noise, noise_schedule, transition, diverse,
                                  is_last_step, u=None):
    """z_s ~ p(z_s | z_t) for every residue (reference sample.py:141-179).  ``diverse`` draws
    from the categorical (inverse CDF with uniforms ``u`` [B,L], default torch.rand on device),
    otherwise argmax; the last step returns the raw logits, as the reference does."""
    if is_last_step:
        return pred_noise
    B, L, C = noised_data.shape
    dev = noised_data.device
    qtb = transition.get_Qt_bar(noise_schedule.get_alpha_bar(t_normalized=t), dev).contiguous()
    qsb = transition.get_Qt_bar(noise_schedule.get_alpha_bar(t_normalized=s), dev).contiguous()
    if diverse and u is None:
        u = torch.rand(B, L, device=dev)
    idx = ops.discrete_posterior_sample(noised_data.argmax(dim=-1).to(torch.int32).contiguous(),
                                        pred_noise.contiguous().float(), qsb, qtb,
                                        u.contiguous().float() if diverse else None)
    return F.one_hot(idx.long(), num_classes=C).float()


class GraphedDenoiseStep:
    """One reverse step of the sequence chain -- ``model.forward`` + ``sample_p_zs_given_zt_discrete`` -- captured once
    into a HIP graph and replayed per step, as structure_model/sample.py::GraphedReverseStep does for the angle chain.
    What varies between steps lives on the device: the step index ``self.s`` ([B,1] float: both normalised times, the
    schedule look-ups and the timestep embedding are computed from it inside the graph), the state ``self.x`` and the
    uniforms of the categorical draw (drawn inside the graph, or injected through ``self.u``).  The last step of a chain
    (which returns the raw logits) is not replayed.  Small chains are host-bound when launched kernel by kernel (about
    150 launches per step): default for at most ``GRAPH_MAX_ROWS`` token rows, ``use_graph`` / E3D_SAMPLE_GRAPH override."""

    def __init__(self, model, x_like, ligand_angles, ligand_mask, receptor_seq, receptor_angles, receptor_mask, noise_schedule,
                 transition, diverse, T, inject_u=False):
        dev = x_like.device
        self.args = (ligand_angles, ligand_mask, receptor_seq, receptor_angles, receptor_mask)
        self.model, self.schedule, self.transition, self.diverse, self.T = model, noise_schedule, transition, diverse, T
        self.x = x_like.clone()
        self.s = torch.zeros((x_like.shape[0], 1), device=dev)
        self.u = torch.zeros(x_like.shape[:2], device=dev) if (inject_u and diverse) else None
        self.out = None
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        # the warm-up draws uniforms when the chain samples (diverse, no injected u): put the device generator back
        # afterwards, so that a seeded chain sees the same random numbers whether its steps are replayed or launched one
        # by one (which path runs is decided by a size threshold)
        rng = torch.cuda.get_rng_state(dev)
        with torch.cuda.stream(side):       # warm-up off the capture: first-launch attribute calls, caches, allocator
            self._body()
        torch.cuda.current_stream(dev).wait_stream(side)
        torch.cuda.synchronize(dev)
        torch.cuda.set_rng_state(rng, dev)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = self._body()

    def _body(self):
        logits = self.model.forward(self.s, self.x, *self.args)
        return sample_p_zs_given_zt_discrete((self.s + 1) / self.T, self.s / self.T, self.x, logits, self.schedule, self.transition,
                                             self.diverse, is_last_step=False, u=self.u)

    def step(self, s_int, x, u=None):
        """z_t -> z_s for the step index ``s_int`` (> 0); returns the graph's output buffer (overwritten by the next call)."""
        if (u is not None) != (self.u is not None):
            raise ValueError("this graph was captured %s injected uniforms" % ("with" if self.u is not None else "without"))
        self.s.fill_(float(s_int))
        if x is not self.x:
            self.x.copy_(x)
        if u is not None:
            self.u.copy_(u)
        self.graph.replay()
        return self.out


GRAPH_MAX_ROWS = 512      # token rows (B x L) up to which a chain replays a captured graph by default


def _use_graph(x):
    env = os.environ.get("E3D_SAMPLE_GRAPH")
    if env in ("0", "1"):
        return env == "1"
    return x.shape[0] * x.shape[1] <= GRAPH_MAX_ROWS


@torch.no_grad()
def denoise(batch, model: PeptideDiff, noise_schedule, transition, diverse, x_T=None, us=None,
            generated_angles=None, timesteps=None, trim_padding=False, use_graph=None):
    """Full reverse chain over CONFIG["timesteps"] steps + recovery metrics (reference
    sample.py:181-229).  ``x_T`` / ``us`` inject the initial one-hot noise and the per-step
    uniforms (parity tests); ``generated_angles`` replaces the dataset's ligand angles
    (sample_by_generated_angles.py:202)."""
    T = CONFIG["timesteps"] if timesteps is None else timesteps
    B, max_len, C = batch["ligand_seq"].shape
    dev = next(model.parameters()).device
    x = generate_discrete_noise(B, max_len, C, dev) if x_T is None else x_T.to(dev)
    ligand_seq = batch["ligand_seq"].to(dev)
    ligand_mask = batch["ligand_attn_mask"].to(dev)
    ligand_angles = (batch["ligand_angles"] if generated_angles is None else generated_angles).to(dev)
    receptor_seq = batch["receptor_seq"].to(dev)
    receptor_angles = batch["receptor_angles"].to(dev)
    receptor_mask = batch["receptor_attn_mask"].to(dev)
    if trim_padding:
        # as structure_model/sample.py::p_sample_loop(trim_padding=True): padding cannot influence valid positions,
        # and only valid positions are read below, so the chain runs on the frame of the longest ligand / pocket
        from ..structure_model.sample import trimmed_length
        Ll, Lr = trimmed_length(ligand_mask), trimmed_length(receptor_mask)
        x, ligand_seq, ligand_mask = x[:, :Ll].contiguous(), ligand_seq[:, :Ll], ligand_mask[:, :Ll].contiguous()
        ligand_angles = ligand_angles[:, :Ll].contiguous()
        receptor_seq, receptor_angles = receptor_seq[:, :Lr].contiguous(), receptor_angles[:, :Lr].contiguous()
        receptor_mask = receptor_mask[:, :Lr].contiguous()
        if us is not None:
            us = [u[:, :Ll] if u is not None and u.dim() >= 2 else u for u in us]
    graphed = None
    if (_use_graph(x) if use_graph is None else use_graph) and T > 4:
        try:
            graphed = GraphedDenoiseStep(model, x, ligand_angles.contiguous(), ligand_mask, receptor_seq, receptor_angles, receptor_mask,
                                         noise_schedule, transition, diverse, T, inject_u=us is not None)
        except Exception as e:   # noqa: BLE001 -- any capture problem: eager launches are always correct
            import warnings
            warnings.warn(f"HIP-graph capture of the sequence reverse step failed ({type(e).__name__}: {e}); using eager launches")
    for n, s_int in enumerate(reversed(range(T))):
        if graphed is not None and s_int > 0:
            u_n = None if us is None else us[n]
            x = graphed.step(s_int, x, None if u_n is None else u_n.to(dev).float())
            continue
        s_array = s_int * torch.ones((B, 1), device=dev)
        t_array = s_array + 1
        logits = model.forward(s_array, x, ligand_angles, ligand_mask, receptor_seq, receptor_angles, receptor_mask)
        x = sample_p_zs_given_zt_discrete(t_array / T, s_array / T, x, logits, noise_schedule, transition,
                                          diverse, is_last_step=s_int == 0, u=None if us is None else us[n])
    pred_idx, true_idx = x.argmax(dim=-1).cpu(), ligand_seq.argmax(dim=-1).cpu()
    mask = ligand_mask.bool().cpu()
    ids, true_sequences, pred_sequences, recovery_rates = [], [], [], []
    for i in range(B):
        m = mask[i]
        recovery_rates.append(((pred_idx[i][m] == true_idx[i][m]).sum() / m.sum()).item())
        pred_sequences.append("".join(AA_VOCAB[j] for j in pred_idx[i][m]))
        true_sequences.append("".join(AA_VOCAB[j] for j in true_idx[i][m]))
        sid = batch.get("structure_ids")
        ids.append(f'{sid["pdb_id"][i]}_{sid["ligand_chain"][i]}' if sid is not None else str(i))
    print(sum(recovery_rates) / len(recovery_rates))
    return ids, true_sequences, pred_sequences, recovery_rates


def run(transition, diverse=True):
    import pandas as pd
    loader = get_dataloader(DATA_PATH)
    model = get_model(len(loader))
    schedule = PredefinedNoiseScheduleDiscrete(CONFIG["noise_schedule"], CONFIG["timesteps"]).to(DEVICE)
    cols = ([], [], [], [])
    for idx, batch in enumerate(loader):
        print(f"Generating Batch {idx}")
        for acc, part in zip(cols, denoise(batch, model, schedule, transition, diverse)):
            acc.extend(part)
    res = pd.DataFrame(zip(*cols), columns=["structure_ids", "true_sequence", "predict_sequence", "recovery_rate"])
    res.to_pickle(OUTPUT_PATH)
    print(res)
    return res


if __name__ == "__main__":
    torch.cuda.set_device(GPU_ID)
    torch.set_num_threads(THREAD_NUM)
    run(BlosumTransition(x_classes=20), diverse=True)
