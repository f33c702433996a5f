"""Plain training loop that stands in for ``pl.Trainer.fit`` as the reference configures it
(structure_model/train_model.py:99-116): gradient-norm clipping, AdamW from
``model.configure_optimizers()``, scheduler stepped per EPOCH (or per step for OneCycleLR),
validation every epoch, best-checkpoint bookkeeping with the reference's ``mode='max'`` quirk,
and -- new relative to the single-GPU reference -- data-parallel gradient averaging over RCCL when
launched with one process per GPU (``torchrun``).
"""
import math
import os
import time

import torch


def adamw(params, lr, weight_decay):
    """torch.optim.AdamW as the reference builds it (structure_model/model.py:361-366).  With every parameter on the GPU
    this is ``optim.ClipAdamW``: the same optimizer (state, state_dict, hooks, schedulers) whose step -- together with the
    gradient-norm clip around it, see ``clip_and_step`` -- runs as three HIP launches over all parameters (csrc/optim.hip)
    instead of torch's multi-tensor passes (norm 0.2 ms + clip multiply 0.27 ms + fused update 0.98 ms for the 146 M
    parameters of the structure model).  E3D_FUSED_ADAMW=torch: torch's own ``fused`` kernel; =0: torch's default."""
    params = list(params)
    choice = os.environ.get("E3D_FUSED_ADAMW", "1")
    on_gpu = bool(params) and all(p.is_cuda for p in params)
    if choice == "1" and on_gpu:
        from .optim import ClipAdamW
        return ClipAdamW(params, lr=lr, weight_decay=weight_decay)
    try:
        return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay, fused=True) if (choice == "torch" and on_gpu) else \
            torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)
    except (TypeError, RuntimeError):
        return torch.optim.AdamW(params, lr=lr, weight_decay=weight_decay)


from . import autograd, ops, sharding


def clip_and_step(params, optim, max_norm, fold=None):
    """``clip_grad_norm_(params, max_norm)`` + ``optim.step()`` -- what Lightning's ``gradient_clip_val`` does around the
    reference's AdamW (structure_model/train_model.py:99-110).  Returns the total gradient norm (device tensor).
    ``fold`` (E3D_FOLD_CLIP=1): hand torch's fused AdamW 1 / clip_coef as its ``grad_scale`` (the GradScaler hook) so that
    the clip happens inside the update kernel instead of a separate multi-tensor pass.  Measured on MI355X (146 M
    parameters, rocprofv3): the multiply pass it removes costs 0.27 ms, but the fused kernel with a grad_scale also writes
    every unscaled gradient back and goes from 59 to 93 us per launch x 17 = +0.59 ms -- a net LOSS of 0.3 ms, so the
    two-call form stays the default; the folded form is kept for the comparison (tests/test_training_gpu.py pins that both
    give the same parameters)."""
    from .optim import ClipAdamW
    if isinstance(optim, ClipAdamW):                  # norm, clip and update in three launches (csrc/optim.hip)
        return optim.step_clipped(max_norm)
    if fold is None:
        fold = os.environ.get("E3D_FOLD_CLIP", "0") == "1"
    grads = [p.grad for p in params if p.grad is not None]
    fused = bool(optim.defaults.get("fused")) and isinstance(optim, (torch.optim.AdamW, torch.optim.Adam))
    if not fold or not max_norm or not grads or not fused or not hasattr(torch.nn.utils, "get_total_norm"):
        norm = torch.nn.utils.clip_grad_norm_(params, max_norm) if max_norm else None
        optim.step()
        return norm
    norm = torch.nn.utils.get_total_norm(grads, 2.0, error_if_nonfinite=False, foreach=True)
    coef = torch.clamp(max_norm / (norm + 1e-6), max=1.0)
    optim.grad_scale = (1.0 / coef).to(torch.float32)
    try:
        optim.step()
    finally:
        optim.grad_scale = None
    return norm


class GraphedStep:
    """One training step -- ``training_step`` + backward with deferred weight gradients + gradient-norm clip + AdamW -- as
    a HIP graph: captured once per batch signature (tensor names, shapes, dtypes) after ``warmup`` eager steps, then
    replayed.  The eager step of the structure model needs ~25 ms of Python and launch calls for ~1 350 kernels that keep
    the GPU busy for ~27 ms: replaying removes the host from the step.

    What makes the step replayable: no device-to-host synchronisation inside it (the losses are masked means, not index
    lists); the learning rate and the AdamW step counts live in device memory (``ClipAdamW.use_device_scalars``); dropout
    decisions take a device-side epoch word on top of their baked seeds (``ops.dropout_epoch``), advanced inside the graph;
    derived-weight caches (W^T, distance-table planes) are refreshed by launches inside the captured backward / forward.
    One graph per batch signature, up to ``MAX_GRAPHS`` of them (frames trimmed to the batch's longest ligand / pocket come
    in a handful of shapes: ``trim_batch``); a signature seen fewer than ``warmup`` times, or beyond that number, runs
    eagerly (a ragged last batch).  Every graph has its own memory pool, its own gradient tensors and its own staging
    buffer for their addresses (``ClipAdamW.new_capture_staging``); parameters, moments, learning rate and step counts are
    shared.  Single process; the data-parallel step is ``GraphedDDPStep`` (two segments around the collectives)."""

    MAX_GRAPHS = 8

    def __init__(self, model, optim, params, gradient_clip, warmup=2):
        from .optim import ClipAdamW
        if not isinstance(optim, ClipAdamW):
            raise TypeError("GraphedStep needs optim.ClipAdamW (device-side learning rate and step counts)")
        self.model, self.optim, self.params, self.clip = model, optim, params, gradient_clip
        self.warmup, self.seen = warmup, {}
        self.graphs = {}                             # signature -> the captured step (graph, static batch, loss tensor)
        self.graph = self.key = self.static = self.loss = None   # ... and the one used last
        self.failed = None
        optim.use_device_scalars(True)
        self.epoch = ops.dropout_epoch(params[0].device)
        # ONE side stream for the eager steps and for the capture: autograd binds an AccumulateGrad node to the stream
        # it was created on and keeps synchronising with that stream for as long as the node lives; eager steps on the
        # caller's stream followed by a capture on another one left the captured graph with unordered work (seen as
        # garbage gradients a few hundred replays in: tools/lab/train_soak.py)
        self.stream = torch.cuda.Stream(device=params[0].device)

    def _body(self, batch, batch_idx=0):
        loss = self.model.training_step(batch, batch_idx)
        self.optim.zero_grad(set_to_none=True)
        if DEFER_WEIGHT_GRADS:
            with autograd.deferred_weight_grads():
                loss.backward()
        else:
            loss.backward()
        clip_and_step(self.params, self.optim, self.clip)
        return loss

    @staticmethod
    def _signature(batch):
        return tuple((k, tuple(v.shape), v.dtype) for k, v in sorted(batch.items()) if torch.is_tensor(v))

    def _capture(self, batch):
        dev = self.params[0].device
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        import gc
        gc.collect()                                 # autograd graphs of earlier steps (and their AccumulateGrad nodes) gone
        autograd.forget_transposes({id(p) for p in self.model.parameters()})   # nothing of another model in this graph
        torch.cuda.synchronize(dev)
        self.optim.zero_grad(set_to_none=True)       # the gradients of the replayed step live in the graph's pool
        self.optim.sync_lr()
        self.optim.new_capture_staging()             # (an earlier graph keeps re-reading ITS gradient addresses from its own)
        graph = torch.cuda.CUDAGraph()
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:      # AccumulateGrad nodes of the eager steps meet the capture stream once: expected here
            quiet(False)
        try:
            with torch.cuda.graph(graph, stream=self.stream):
                loss = self._body(self.static)
                self.epoch.add_(1)
        finally:
            if quiet is not None:
                quiet(True)
        self.graph, self.loss = graph, loss
        self.optim.note_replayed_step(-1)            # capture ran step()'s host bookkeeping without executing anything
        self.tab = self.optim._e3d_tab               # the optimizer tables (parameter / moment pointers) the graph baked
        self.ptrs = [p.data_ptr() for p in self.params]

    def _stale(self):
        """The graph carries raw pointers: parameters that moved (``module.to()``) or optimizer state that was replaced
        (``load_state_dict``: ClipAdamW drops its tables) invalidate it -- warm up and capture again."""
        return self.optim._e3d_tab is not self.tab or any(p.data_ptr() != q for p, q in zip(self.params, self.ptrs))

    def _may_capture(self, key):
        return (self.failed is None and key not in self.graphs and len(self.graphs) < self.MAX_GRAPHS
                and self.seen.get(key, 0) >= self.warmup)

    def _remember(self, key):
        self.graphs[key] = dict(graph=self.graph, graph2=getattr(self, "graph2", None), static=self.static, loss=self.loss,
                                grads=[p.grad for p in self.params])
        self.key = key

    def _select(self, key):
        """Make the captured step of this signature the current one; False if there is none."""
        e = self.graphs.get(key)
        if e is None:
            return False
        if self.params[0].grad is not e["grads"][0]:   # ``p.grad`` shows the gradients of the step that ran last
            for p, g in zip(self.params, e["grads"]):
                p.grad = g
        self.graph, self.static, self.loss, self.key = e["graph"], e["static"], e["loss"], key
        if e["graph2"] is not None:
            self.graph2 = e["graph2"]
        return True

    def _drop_graphs(self):
        self.graphs, self.seen = {}, {}
        self.graph = self.key = self.static = self.loss = None
        if hasattr(self, "graph2"):
            self.graph2 = None

    def step(self, batch, batch_idx=0):
        """Returns the loss (a device tensor; for a replayed step it is overwritten by the next replay of its graph)."""
        key = self._signature(batch)
        if self.graphs and self._stale():
            self._drop_graphs()
        if self._may_capture(key):
            try:
                self._capture(batch)                # (records, does not execute: this batch runs as the first replay below)
                self._remember(key)
            except Exception as e:                  # noqa: BLE001 -- any capture failure: stay eager, say so once
                self.failed = e
                self._drop_graphs()
                import traceback
                import warnings
                warnings.warn(f"training step could not be captured in a HIP graph, staying eager: {e!r}\n"
                              + "".join(traceback.format_exc(limit=-6)))
        if self._select(key):
            for k, v in batch.items():
                if torch.is_tensor(v):
                    self.static[k].copy_(v, non_blocking=True)
            self.optim.sync_lr()
            self.graph.replay()
            self.optim.note_replayed_step()
            ops.invalidate_weight_caches()
            return self.loss
        self.seen[key] = self.seen.get(key, 0) + 1
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            loss = self._body(batch, batch_idx).detach()
            self.epoch.add_(1)
        cur.wait_stream(self.stream)
        loss.record_stream(cur)
        return loss


class GraphedDDPStep(GraphedStep):
    """The data-parallel step as TWO graph segments around the collectives: forward + backward (gradients written straight into
    the GradientAverager's flat buckets) | all-reduce of the buckets, eager: RCCL calls are not captured | mean, gradient-norm
    clip and AdamW.  ``world > 1`` then costs two replays and a handful of ``all_reduce`` calls per step instead of ~1 350
    eager launches (the eager data-parallel step was host-bound: as much Python as kernel time).  Captured after ``warmup``
    eager steps through the ordinary overlapped path (``prepare`` / hooks / ``average``), so every rank reaches the capture at
    the same step; ``E3D_TRAIN_GRAPH=0`` keeps that eager path for every step.  The buckets travel AFTER the backward segment
    (the eager path overlaps them with it): ~1-2 ms for the sequence model's 289 MB on eight GPUs against a ~20-ms step."""

    def __init__(self, model, optim, params, gradient_clip, averager, warmup=2):
        super().__init__(model, optim, params, gradient_clip, warmup)
        self.avg = averager
        self.graph2 = None
        self.world = torch.distributed.get_world_size()

    def _eager(self, batch, batch_idx=0):
        loss = self.model.training_step(batch, batch_idx)
        self.optim.zero_grad(set_to_none=True)
        self.avg.prepare()
        if DEFER_WEIGHT_GRADS:
            with autograd.deferred_weight_grads(on_param=self.avg.mark_ready):
                loss.backward()
        else:
            loss.backward()
        self.avg.average()
        clip_and_step(self.params, self.optim, self.clip)
        return loss

    def _body(self, batch, batch_idx=0):
        return self._eager(batch, batch_idx)

    def _capture(self, batch):
        dev = self.params[0].device
        self.static = {k: (v.clone() if torch.is_tensor(v) else v) for k, v in batch.items()}
        import gc
        gc.collect()
        autograd.forget_transposes({id(p) for p in self.model.parameters()})
        torch.cuda.synchronize(dev)
        self.optim.zero_grad(set_to_none=True)
        self.optim.sync_lr()
        self.optim.new_capture_staging()
        avg = self.avg
        g1, g2 = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
        quiet = getattr(torch.autograd.graph, "set_warn_on_accumulate_grad_stream_mismatch", None)
        if quiet is not None:
            quiet(False)
        try:
            with torch.cuda.graph(g1, stream=self.stream):
                loss = self.model.training_step(self.static, 0)
                avg.bind(collect_only=True)          # gradients = zeroed views of the flat buckets; hooks only take notes
                if DEFER_WEIGHT_GRADS:
                    with autograd.deferred_weight_grads(on_param=avg.mark_ready):
                        loss.backward()
                else:
                    loss.backward()
                self.epoch.add_(1)
            avg.finish_collect()                      # never-used parameters: grad None, as in every eager step
            with torch.cuda.graph(g2, stream=self.stream, pool=g1.pool()):
                for flat in avg.flats():
                    flat.div_(self.world)
                clip_and_step(self.params, self.optim, self.clip)
        finally:
            if quiet is not None:
                quiet(True)
        self.graph, self.graph2, self.loss = g1, g2, loss
        self.optim.note_replayed_step(-1)
        self.tab = self.optim._e3d_tab
        self.ptrs = [p.data_ptr() for p in self.params]

    def step(self, batch, batch_idx=0):
        key = self._signature(batch)
        if self.graphs and self._stale():
            self._drop_graphs()
        if self._may_capture(key):
            try:
                self._capture(batch)
                self._remember(key)
            except Exception as e:                  # noqa: BLE001 -- any capture failure: stay eager, say so once
                self.failed = e
                self._drop_graphs()
                import traceback
                import warnings
                warnings.warn(f"data-parallel training step could not be captured in HIP graphs, staying eager: {e!r}\n"
                              + "".join(traceback.format_exc(limit=-6)))
        cur = torch.cuda.current_stream(self.stream.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            if self._select(key):
                for k, v in batch.items():
                    if torch.is_tensor(v):
                        self.static[k].copy_(v, non_blocking=True)
                self.optim.sync_lr()
                self.graph.replay()
                self.avg.all_reduce_flats()          # the only eager launches of the step
                self.graph2.replay()
                self.optim.note_replayed_step()
                ops.invalidate_weight_caches()
                loss = self.loss
            else:
                self.seen[key] = self.seen.get(key, 0) + 1
                loss = self._eager(batch, batch_idx).detach()
                self.epoch.add_(1)
        cur.wait_stream(self.stream)
        loss.record_stream(cur)
        return loss


# the tensors of a dataset.py batch that are laid out [B, L, ...] over the ligand / the pocket frame
# (structure_model/dataset.py:119-162, sequence_model/dataset.py: the same names)
LIGAND_FRAME_KEYS = ("ligand_angles", "ligand_attn_mask", "ligand_seq", "known_noise", "noised_ligand_angle")
RECEPTOR_FRAME_KEYS = ("receptor_angles", "receptor_attn_mask", "receptor_seq")
TRIM_TRAIN = os.environ.get("E3D_TRAIN_TRIM", "0") == "1"            # default of fit(trim_padding=None)


def trimmed_frame(batch, multiple=32):
    """(ligand rows, pocket rows) that cover every valid position of the batch, rounded up to the attention tile."""
    from .structure_model.sample import trimmed_length
    return (trimmed_length(batch["ligand_attn_mask"], multiple), trimmed_length(batch["receptor_attn_mask"], multiple))


def trim_batch(batch, frame=None, multiple=32):
    """The batch on the frame of its longest ligand / pocket (``frame``: rows agreed on elsewhere, e.g. across ranks).

    dataset.py pads every item to ``max_seq_len`` = 128 rows; BioLiP ligands are 5-30 residues, so 3 of the decoder's 4
    attention tiles -- and 3/4 of the rows of every decoder GEMM, LayerNorm and activation -- are padding.  Padding cannot
    reach a valid position in the forward pass (its keys carry the -10000 bias, whose softmax weight underflows to exactly
    0.0f; every other op is row-wise) and receives exactly zero gradient in the backward pass (the losses are means over
    valid positions), so loss and parameter gradients of the trimmed batch are those of the padded one up to the order of
    the fp32 sums (tests/test_training_gpu.py::test_trimmed_*).  What changes: draws made per frame position inside the
    step (dropout, PeptideDiff.apply_aa_noise) come from a different place of the random stream."""
    Ll, Lr = frame if frame is not None else trimmed_frame(batch, multiple)
    out = dict(batch)
    for keys, n in ((LIGAND_FRAME_KEYS, Ll), (RECEPTOR_FRAME_KEYS, Lr)):
        for k in keys:
            v = out.get(k)
            if torch.is_tensor(v) and v.dim() >= 2 and v.shape[1] > n:
                out[k] = v[:, :n].contiguous()
    return out


def move_batch(batch, device):
    return {k: (v.to(device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}


class BestCheckpoint:
    """ModelCheckpoint(monitor='val_loss', save_top_k=1, mode=...) semantics; the reference passes
    mode='max', i.e. it keeps the HIGHEST validation loss -- reproduced by default."""

    def __init__(self, path, mode="max"):
        self.path, self.mode, self.best = path, mode, None

    def update(self, model, val_loss, rank=0):
        better = self.best is None or (val_loss > self.best if self.mode == "max" else val_loss < self.best)
        if better:
            self.best = val_loss
            if rank == 0 and self.path:
                torch.save(model.state_dict(), self.path)
        return better


# The reference trains with torch.set_float32_matmul_precision("medium") (structure_model/train_model.py:120,
# sequence_model/train_model.py:114): bf16 products.  bf16x3 is 500x finer per product and is the arithmetic every
# backward kernel of this package exists in; the inference default (f16x3) has forward kernels only.
# E3D_TRAIN_ARITHMETIC=bf16 (opt-in): the reference's own training precision -- plain bf16 products in every GEMM (forward,
# input and weight gradients), bf16x3 in the attention kernels; ~1e-2-grade gradients instead of ~1e-4-grade.
TRAIN_ARITHMETIC = os.environ.get("E3D_TRAIN_ARITHMETIC", "bf16x3")
# single-process training replays the step from a HIP graph (GraphedStep); 0: eager steps
GRAPH_TRAIN = os.environ.get("E3D_TRAIN_GRAPH", "1") == "1"
DEFER_WEIGHT_GRADS = os.environ.get("E3D_DEFER_WGRAD", "1") == "1"   # autograd.deferred_weight_grads in the step


def fit(model, train_loader, val_loader=None, *, max_epochs, min_epochs=0, gradient_clip=1.0, device="cuda:0",
        log_every_n_steps=30, checkpoint_path="./best_val_model.pt", checkpoint_mode="max", max_steps=None,
        log=print, trim_padding=None):
    """Returns a history dict.  ``model`` provides training_step / validation_step /
    configure_optimizers (the reference's LightningModule surface).
    ``trim_padding`` (None: E3D_TRAIN_TRIM, default off = the reference's padded frames): run every training and
    validation step on the frame of the batch's longest ligand / pocket (``trim_batch``; under a process group the frame
    is the maximum over the ranks, agreed on the host, so that every rank replays the same kind of step)."""
    with ops.arithmetic(TRAIN_ARITHMETIC):
        return _fit(model, train_loader, val_loader, max_epochs=max_epochs, min_epochs=min_epochs, gradient_clip=gradient_clip,
                    device=device, log_every_n_steps=log_every_n_steps, checkpoint_path=checkpoint_path,
                    checkpoint_mode=checkpoint_mode, max_steps=max_steps, log=log,
                    trim_padding=TRIM_TRAIN if trim_padding is None else bool(trim_padding))


def _fit(model, train_loader, val_loader, *, max_epochs, min_epochs, gradient_clip, device, log_every_n_steps, checkpoint_path,
         checkpoint_mode, max_steps, log, trim_padding=False):
    rank, world, _ = sharding.init_distributed()
    model.to(device)
    sharding.broadcast_parameters(model, src=0)
    conf = model.configure_optimizers()
    optim = conf["optimizer"]
    sched = conf.get("lr_scheduler")
    averager = sharding.GradientAverager(model.parameters())
    ckpt = BestCheckpoint(checkpoint_path, checkpoint_mode)
    history = {"train_loss": [], "val_loss": [], "steps": 0, "seconds": 0.0}
    params = [p for p in model.parameters() if p.requires_grad]
    stepper = None
    if GRAPH_TRAIN and params and params[0].is_cuda:
        from .optim import ClipAdamW
        if isinstance(optim, ClipAdamW):
            if world == 1 and not averager._active():
                stepper = GraphedStep(model, optim, params, gradient_clip)
            elif averager._active() and averager._hooked:
                stepper = GraphedDDPStep(model, optim, params, gradient_clip, averager)
    t0 = time.perf_counter()
    step = 0
    for epoch in range(max_epochs):
        model.train()
        if hasattr(getattr(train_loader, "sampler", None), "set_epoch"):
            train_loader.sampler.set_epoch(epoch)
        losses = []
        for batch_idx, batch in enumerate(train_loader):
            if trim_padding:                         # (on the loader's host tensors: no device round trip)
                batch = trim_batch(batch, sharding.max_over_ranks_host(trimmed_frame(batch)) if world > 1 else None)
            batch = move_batch(batch, device)
            if stepper is not None:
                loss = stepper.step(batch, batch_idx)
                if sched is not None and sched.get("interval") == "step":
                    sched["scheduler"].step()
                losses.append(loss.detach().clone())     # (a replayed step's loss tensor is overwritten by the next replay)
                step += 1
                if rank == 0 and log_every_n_steps and step % log_every_n_steps == 0:
                    log(f"epoch {epoch} step {step} train_loss {float(losses[-1]):.5f}")
                if max_steps is not None and step >= max_steps:
                    break
                continue
            loss = model.training_step(batch, batch_idx)
            optim.zero_grad(set_to_none=True)
            averager.prepare()                       # grads as views of the all-reduce buckets (no-op for one process)
            # the weight gradients of all linear layers are computed together when the block ends (grouped launches,
            # written into .grad, i.e. into the all-reduce buckets); E3D_DEFER_WGRAD=0: layer by layer inside backward
            if DEFER_WEIGHT_GRADS:
                with autograd.deferred_weight_grads(on_param=averager.mark_ready if averager._active() else None):
                    loss.backward()
            else:
                loss.backward()
            averager.average()                       # RCCL all-reduce (no-op for one process)
            clip_and_step(params, optim, gradient_clip)   # global-norm clip of the averaged grads, then AdamW
            ops.invalidate_weight_caches()           # belt and braces beside the global optimizer hook (ops.py)
            if sched is not None and sched.get("interval") == "step":
                sched["scheduler"].step()
            losses.append(loss.detach())
            step += 1
            if rank == 0 and log_every_n_steps and step % log_every_n_steps == 0:
                log(f"epoch {epoch} step {step} train_loss {float(losses[-1]):.5f}")
            if max_steps is not None and step >= max_steps:
                break
        if sched is not None and sched.get("interval") == "epoch":
            sched["scheduler"].step()
        # the per-step losses stay on the device until here: a float() per step would make the host wait for every step
        # (Lightning reads the loss for its progress bar every step; the values of the logged steps are the same)
        losses = torch.stack(losses).double().cpu().tolist() if losses else []
        mean_train = sum(losses) / max(1, len(losses))
        history["train_loss"].append(mean_train)
        if rank == 0:
            log(f"Traning Loss:{mean_train}")
        if val_loader is not None:
            model.eval()
            vals = []
            with torch.no_grad():
                for batch_idx, batch in enumerate(val_loader):
                    if trim_padding:
                        batch = trim_batch(batch)
                    out = model.validation_step(move_batch(batch, device), batch_idx)
                    vals.append(float(out["val_loss"] if isinstance(out, dict) else out))
            val = sum(vals) / max(1, len(vals)) if vals else math.nan
            if world > 1:
                val = sharding.mean_over_ranks(val)
            history["val_loss"].append(val)
            if rank == 0:
                log(f"Validation Loss:{val}")
            if not math.isnan(val):
                ckpt.update(model, val, rank)
        if max_steps is not None and step >= max_steps:
            break
    history["steps"] = step
    history["seconds"] = time.perf_counter() - t0
    return history
