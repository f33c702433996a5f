"""Reverse (ancestral) sampler of the structure model -- entry point and function names of the
reference's structure_model/sample.py, restructured for the device:

  * the pocket encoder and the decoder's cross K/V run ONCE per batch (they do not depend on the
    timestep; the reference recomputes them every step, sample.py:86-89);
  * the schedule tables are built once (the reference recomputes them every step, sample.py:74);
  * update + wrap is one HIP kernel (``e3d_ddpm_step_wrap``) and the trajectory stays in HBM
    until the loop ends (the reference does a blocking D2H copy per step, sample.py:143).

Run as ``python sample.py`` from this directory after editing the constants, like the reference.
"""
if __package__ in (None, ""):  # executed as a script from inside this directory
    import os as _os, sys as _sys
    _sys.path.insert(0, _os.path.dirname(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__)))))
    import __graft_entry__ as _g
    _g.load_package()
    __package__ = "e3diff_amd.structure_model"

import os
import pickle
import warnings

import torch
from torch import nn

from .. import ops
from ..bert import BertConfig
from .dataset import LigandBindingSiteDataset, NoisedAnglesDataset
from .model import ConditionalBertForDiffusion
from .utils import CosineTables

MODEL_PATH = ""  # trained state_dict (same key names as the reference's checkpoints)
OUTPUT = "./data/output.pkl"
DATA_FILE = "./data/biolip.pt"
GPU_ID = 0
STEP = 1  # stride over timesteps; >1 trades quality for speed (reference sample.py:16)
# Arithmetic of this entry point.  The reference's structure_model/sample.py never calls
# torch.set_float32_matmul_precision (only the train scripts and the sequence sampler set "medium"), i.e. it
# samples at full fp32: ``sample()`` runs the fp32-grade f16x3 kernels (two fp16 terms per operand, see ops.py; 4.9e-6
# from the CPU oracle end to end vs 3.2e-6 for the exact fp32 MFMA path) unless E3D_GEMM_MODE says otherwise.
ARITHMETIC = "f16x3"

CONFIG = {
    "pocket_ext": 0,
    "timesteps": 1000,
    "max_seq_len": 64,

    "num_heads": 12,
    "dropout_p": 0.1,
    "hidden_size": 768,
    "num_hidden_layers": 12,
    "intermediate_size": 1024,
    "position_embedding_type": "relative_key",

    "lr": 5e-5,
    "l2_norm": 0.1,
    "loss": "smooth_l1",
    "gradient_clip": 1.0,
    "lr_scheduler": "LinearWarmup",

    "min_epochs": 500,
    "max_epochs": 1000,
    "batch_size": 64,
}

DEVICE = f"cuda:{GPU_ID}"


def _tables(betas):
    return betas if isinstance(betas, CosineTables) else CosineTables.from_betas(betas.detach().cpu().float())


@torch.no_grad()
def p_sample(model, ligand_mask, ligand_angle_noise, receptor_seq, receptor_mask, receptor_angle,
             timestep, betas, noise=None, receptor_cache=None, out=None, wrap=False) -> torch.Tensor:
    """One reverse step x_t -> x_{t-1} (reference sample.py:55-99).  Like the reference's
    p_sample the result is NOT wrapped unless ``wrap=True`` (p_sample_loop's sample.py:140-142
    fused into the same kernel).

    ``timestep``: int64 [B] with one distinct value (asserted, as in the reference) or an int.
    ``betas``: the schedule betas [T] (any device) or a prebuilt CosineTables.
    ``noise``: optional injected N(0,1) draw (parity tests); default torch.randn_like on device.
    """
    return _reverse_step(model, ligand_mask, ligand_angle_noise, receptor_seq, receptor_mask,
                         receptor_angle, timestep, betas, noise, receptor_cache, out, wrap=wrap)


def _reverse_step(model, ligand_mask, x_t, receptor_seq, receptor_mask, receptor_angle, timestep,
                  betas, noise, receptor_cache, out, wrap, mod=None):
    tab = _tables(betas)
    if isinstance(timestep, int):
        t_index = timestep
        timestep = torch.full((x_t.shape[0],), t_index, device=x_t.device, dtype=torch.long)
    else:
        t_unique = torch.unique(timestep)
        assert len(t_unique) == 1, f"Got multiple values for t: {t_unique}"
        t_index = int(t_unique.item())
    if receptor_cache is None:
        receptor_cache = model.encode_receptor(receptor_seq, receptor_angle, receptor_mask)
    eps_hat = model.decode(timestep, x_t, ligand_mask, receptor_cache, mod=mod)
    sra = float(tab.sqrt_recip_alphas[t_index])
    beta = float(tab.betas[t_index])
    s1m = float(tab.sqrt_one_minus_alphas_cumprod[t_index])
    x_c = x_t.contiguous().float()
    if t_index == 0:
        noise, sigma = None, 0.0
    else:
        sigma = float(tab.sigma[t_index])
        noise = torch.randn_like(x_c) if noise is None else noise.contiguous()
    return ops.ddpm_step_wrap(x_c, eps_hat.contiguous(), noise, sra, beta, s1m, sigma, wrap=wrap, out=out)




class GraphedReverseStep:
    """One reverse step (decoder forward + DDPM update + wrap) captured once into a HIP graph and replayed
    per step.  Everything that varies between steps lives on the device: the step index (``self.t``,
    also the row of the [T,4] coefficient table read by ``e3d_ddpm_step_wrap_table``), the state
    ``self.x`` and the noise draw.  Results are bit-identical to the eager path for the same noise.

    Default for chains of at most GRAPH_MAX_ROWS token rows (up to ~16 pockets of 64 residues), ``use_graph=True`` /
    E3D_SAMPLE_GRAPH=1 forces it, =0 turns it off.  Measured on MI355X, one 64-residue pocket (tools/bench_single.py):
    round 1, 6-workgroup tiled GEMMs of ~29 us each: 3.9 ms per replayed step against 3.8 ms eager -- the GPU was 100 %
    busy with dependent kernels, the graph had nothing to remove.  Round 2, K-sliced small-M GEMMs of ~7 us per product
    (csrc/gemm_skinny.hip): eager launches are now host-bound at 2.4 ms per step, a replay takes 1.5 ms."""

    def __init__(self, model, ligand_mask, cache, tab, x_like, wrap=True, draw=True, mod_table=None):
        """``draw``: the graph draws its own N(0,1) noise each replay; False: ``step`` takes the draw (parity tests).
        ``mod_table`` [T,6H]: row t = model.timestep_modulation(t), read on the device by the step index."""
        dev = x_like.device
        self.model, self.mask, self.cache, self.wrap, self.mod_table = model, ligand_mask, cache, wrap, mod_table
        self.x = torch.empty_like(x_like)
        self.out = torch.empty_like(x_like)
        self.noise = torch.zeros_like(x_like)
        self.t = torch.zeros((x_like.shape[0],), device=dev, dtype=torch.long)
        self.coef = torch.stack([tab.sqrt_recip_alphas, tab.betas, tab.sqrt_one_minus_alphas_cumprod, tab.sigma],
                                dim=1).float().contiguous().to(dev)
        self.draw = draw
        self.x.copy_(x_like)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):      # warm-up off the capture: first-launch attribute calls, allocator
            self._body()                   # (one eager step: the chain pays for it once, ~2.4 ms at B=1, L=64)
        torch.cuda.current_stream(dev).wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._body()

    def _body(self):
        mod = None if self.mod_table is None else self.mod_table.index_select(0, self.t[:1])
        eps_hat = self.model.decode(self.t, self.x, self.mask, self.cache, mod=mod)
        if self.draw:
            self.noise.normal_()
        ops.ddpm_step_wrap_table(self.x, eps_hat.contiguous(), self.noise, self.coef, self.t, wrap=self.wrap, out=self.out)

    def step(self, i, x, noise=None):
        """x_t -> x_{t-1} for step index i; returns the graph's output buffer (overwritten by the next call)."""
        if (noise is not None) == self.draw:
            raise ValueError("this graph was captured %s injected noise" % ("without" if self.draw else "with"))
        self.t.fill_(i)
        if x is not self.x:
            self.x.copy_(x)
        if noise is not None:
            self.noise.copy_(noise)
        self.graph.replay()
        return self.out


GRAPH_MAX_ROWS = 512      # token rows (B x L) up to which a chain replays a captured graph by default (B=8 x L=64: 2.05 vs 2.16 ms eager; B=16: GPU-bound, eager)


def _use_graph(x):
    env = os.environ.get("E3D_SAMPLE_GRAPH")
    if env in ("0", "1"):
        return env == "1"
    return x.shape[0] * x.shape[1] <= GRAPH_MAX_ROWS


def trimmed_length(mask, multiple=32):
    """Smallest multiple of ``multiple`` that covers every valid position of a [B,L] 0/1 padding mask whose valid
    positions are a prefix (the dataset layout, dataset.py:119-132); L itself if any row is not a prefix mask."""
    L = mask.shape[1]
    lengths = mask.sum(dim=1)
    prefix = (mask[:, :-1] >= mask[:, 1:]).all() if L > 1 else torch.tensor(True)
    longest = int(lengths.max().item())
    if not bool(prefix):
        return L
    return max(multiple, min(L, -(-longest // multiple) * multiple))


@torch.no_grad()
def p_sample_loop(model: nn.Module, ligand_mask, ligand_angle_noise, receptor_seq, receptor_mask,
                  receptor_angle, total_timesteps: int, betas, disable_pbar: bool = False,
                  noises=None, return_device: bool = False, step: int = None, use_graph: bool = None,
                  trim_padding: bool = False) -> torch.Tensor:
    """Full reverse chain; returns [T/STEP, B, L, n_ft] (on the host like the reference,
    sample.py:101-144, unless ``return_device``).  ``noises`` [T/STEP,B,L,n_ft] injects the draws.
    ``use_graph``: replay one captured HIP graph per step (None: by size, E3D_SAMPLE_GRAPH=0/1 overrides -- see
    GraphedReverseStep); falls back to eager launches if the capture fails."""
    step = STEP if step is None else step
    tab = _tables(betas)
    order = list(reversed(range(0, total_timesteps, step)))
    x = ligand_angle_noise.contiguous().float()
    full_traj = None
    if trim_padding:
        # Padding positions cannot influence valid ones (their keys carry the -10000 bias, whose softmax weight
        # underflows to exactly 0.0f; every other op is row-wise), so the chain only needs the rows up to the longest
        # ligand / pocket of the batch, rounded up to the 32-row attention tile.  BioLiP ligands are 5-30 residues
        # in a 64-256 row frame: the decoder then runs on 1/8 of the rows.  Valid positions are unchanged; trimmed
        # positions come back as 0 (the reference's values there are never used: sample.py:243 slices them off).
        Ll, Lr = trimmed_length(ligand_mask), trimmed_length(receptor_mask)
        if Ll < x.shape[1] or Lr < receptor_mask.shape[1]:
            full_traj = torch.zeros((len(order),) + tuple(x.shape), device=x.device, dtype=torch.float32)
            x, ligand_mask = x[:, :Ll].contiguous(), ligand_mask[:, :Ll].contiguous()
            receptor_seq, receptor_mask = receptor_seq[:, :Lr].contiguous(), receptor_mask[:, :Lr].contiguous()
            receptor_angle = receptor_angle[:, :Lr].contiguous()
            if noises is not None:
                noises = noises[:, :, :Ll]
    cache = model.encode_receptor(receptor_seq, receptor_angle, receptor_mask)
    traj = torch.empty((len(order),) + tuple(x.shape), device=x.device, dtype=torch.float32)
    # what depends on the timestep alone, for the whole chain at once: row t of the table = timestep_modulation(t)
    mod_rows = model.timestep_modulation(torch.tensor(order, device=x.device, dtype=torch.long))
    mod_table = torch.zeros((total_timesteps, mod_rows.shape[1]), device=x.device, dtype=torch.float32)
    mod_table[order] = mod_rows
    graphed = None
    if (_use_graph(x) if use_graph is None else use_graph) and len(order) > 4:
        try:
            graphed = GraphedReverseStep(model, ligand_mask.contiguous().float(), cache, tab, x, draw=noises is None,
                                         mod_table=mod_table)
        except Exception as e:   # noqa: BLE001 -- any capture problem: eager launches are always correct
            import warnings
            warnings.warn(f"HIP-graph capture of the reverse step failed ({type(e).__name__}: {e}); using eager launches")
            graphed = None
    for n, i in enumerate(order):
        if graphed is not None:
            x = graphed.step(i, graphed.out if n else x, None if noises is None else noises[n].contiguous())
            traj[n].copy_(x)
        else:
            x = _reverse_step(model, ligand_mask, x, None, None, None, i, tab,
                              None if noises is None else noises[n], cache, traj[n], wrap=True, mod=mod_table[i:i + 1])
    if full_traj is not None:
        full_traj[:, :, :traj.shape[2]] = traj
        traj = full_traj
    return traj if return_device else traj.cpu()


def get_dataset(file_path):
    ds = LigandBindingSiteDataset(file_path, "test", CONFIG["max_seq_len"], CONFIG["pocket_ext"])
    return NoisedAnglesDataset(ds, timesteps=CONFIG["timesteps"])


def build_configs(cfg=None):
    cfg = cfg or CONFIG
    common = dict(max_position_embeddings=cfg["max_seq_len"], num_attention_heads=cfg["num_heads"],
                  hidden_size=cfg["hidden_size"], intermediate_size=cfg["intermediate_size"],
                  num_hidden_layers=cfg["num_hidden_layers"],
                  position_embedding_type=cfg["position_embedding_type"],
                  hidden_dropout_prob=cfg["dropout_p"], attention_probs_dropout_prob=cfg["dropout_p"],
                  use_cache=False)
    return BertConfig(**common), BertConfig(**common, is_decoder=True, add_cross_attention=True)


def load_model(dataset, model_path=None):
    encoder_config, decoder_config = build_configs()
    model = ConditionalBertForDiffusion(
        encoder_config=encoder_config, decoder_config=decoder_config,
        feature_names=dataset.feature_names, epochs=CONFIG["max_epochs"],
        lr_scheduler=CONFIG["lr_scheduler"], l2_lambda=CONFIG["l2_norm"],
        steps_per_epoch=len(dataset), learning_rate=CONFIG["lr"],
        loss_func=[ConditionalBertForDiffusion.diheral_loss_func] * 4
        + [ConditionalBertForDiffusion.angle_loss_func] * 4)
    path = MODEL_PATH if model_path is None else model_path
    if path:
        model.load_state_dict(torch.load(path, map_location="cpu", weights_only=True))
    return model.eval().to(DEVICE)


def sample(model, test_angle_ds, all_batches: bool = False):
    """Sample the test pockets in batches of CONFIG["batch_size"]; returns a list of
    [T, l_i, 8] arrays trimmed to each ligand's length.  Like the reference (sample.py:237) only
    the first batch is generated unless ``all_batches``."""
    bs = CONFIG["batch_size"]
    items = [test_angle_ds[i] for i in range(len(test_angle_ds))]

    def chunk(name):
        return [torch.stack([it[name] for it in items[i:i + bs]]) for i in range(0, len(items), bs)]

    ligand_mask, receptor_angle = chunk("ligand_attn_mask"), chunk("receptor_angles")
    receptor_seq, receptor_mask = chunk("receptor_seq"), chunk("receptor_attn_mask")
    pad, feature_size = items[0]["ligand_angles"].shape
    retval = []
    for idx, lm in enumerate(ligand_mask):
        print(f"Generating Batch {idx}/{len(ligand_mask)}")
        lengths = lm.sum(dim=1).int()
        x_T = test_angle_ds.sample_noise(torch.zeros((len(lengths), pad, feature_size)))
        def chain(arithmetic):
            with ops.arithmetic(arithmetic):
                return p_sample_loop(
                    model=model, ligand_mask=lm.to(DEVICE), ligand_angle_noise=x_T.to(DEVICE),
                    receptor_seq=receptor_seq[idx].to(DEVICE), receptor_mask=receptor_mask[idx].to(DEVICE),
                    receptor_angle=receptor_angle[idx].to(DEVICE), total_timesteps=test_angle_ds.timesteps,
                    betas=test_angle_ds.alpha_beta_terms["betas"], trim_padding=True)   # sliced to l_i right below

        sampled = chain(ARITHMETIC)
        if ops.GEMM_MODES.get(ARITHMETIC) == 19 and "E3D_GEMM_MODE" not in os.environ and not bool(torch.isfinite(sampled).all()):
            # f16x3's one remaining range limit: an ACTIVATION beyond 65504 (weights are pre-scaled) turns into inf / NaN.
            # The chain is then run again in bf16x6 -- same fp32 grade, fp32 exponent range, half the speed -- and says so
            warnings.warn("structure sampling: non-finite angles in f16x3 arithmetic (an activation left the fp16 range); "
                          "re-running this batch in bf16x6")
            sampled = chain("bf16x6")
        retval.extend(sampled[:, i, :l, :].numpy() for i, l in enumerate(lengths))
        if not all_batches:
            break
    return retval


if __name__ == "__main__":
    torch.cuda.set_device(GPU_ID)
    test_angle_dataset = get_dataset(DATA_FILE)
    model = load_model(test_angle_dataset)
    sample_result = sample(model, test_angle_dataset)
    with open(OUTPUT, "+wb") as f:
        pickle.dump(sample_result, f)
