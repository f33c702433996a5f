"""One process per GPU: pocket sharding for sampling, gradient averaging for training.

Pockets are independent (attention never crosses batch items; schedules are small replicated
tables), so sampling needs NO collective on the data path -- only a gather of the finished
samples.  Training is data parallel: an all-reduce of the fp32 gradients (RCCL over xGMI via
torch.distributed backend "nccl"; "gloo" on CPU for the plumbing tests), bucketed so that a
handful of large messages cross the point-to-point xGMI links instead of ~500 small ones.
The reference is single-GPU; this file is new work for BASELINE configs 4-5.
"""
import os

import torch
import torch.distributed as dist


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (torchrun).  Returns
    (rank, world, local_rank); a single process needs no group and gets (0, 1, 0)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this driver
        if backend is None:   # E3D_DIST_BACKEND=gloo: rehearse the multi-rank paths on a box with one GPU
            backend = os.environ.get("E3D_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if backend == "nccl":
            torch.cuda.set_device(local_rank)
            dist.init_process_group(backend, device_id=torch.device(f"cuda:{local_rank}"))
        else:
            dist.init_process_group(backend)
    return rank, world, local_rank


def shard_slice(n_items, rank, world):
    """Contiguous, balanced [start, stop) of ``n_items`` for ``rank`` (first n % world ranks get
    one extra item), so concatenating the ranks' outputs in rank order restores dataset order."""
    base, extra = divmod(n_items, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def gather_in_rank_order(local_items):
    """list (picklable, e.g. numpy trajectories trimmed to ligand length) -> the concatenation
    over ranks on every rank.  One collective at the END of sampling (a few MB)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return list(local_items)
    parts = [None] * dist.get_world_size()
    dist.all_gather_object(parts, list(local_items))
    return [x for part in parts for x in part]


def max_over_ranks(value):
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def mean_over_ranks(value):
    """Mean of a scalar over ranks (the reference's only collective: ``self.all_gather(val).mean()``
    on the per-feature validation losses, structure_model/model.py:344)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return float(value)
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([float(value)], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item()) / dist.get_world_size()


_HOST_GROUP = None


def max_over_ranks_host(values):
    """Element-wise maximum of a few host integers over the ranks, on the host (a gloo group beside the RCCL one, made on
    first use): nothing is enqueued on the GPU and nothing of the GPU is waited for, so the step before keeps running."""
    values = [int(v) for v in values]
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return tuple(values)
    global _HOST_GROUP
    if dist.get_backend() == "gloo":
        group = None
    else:
        if _HOST_GROUP is None:
            _HOST_GROUP = dist.new_group(backend="gloo")
        group = _HOST_GROUP
    t = torch.tensor(values, dtype=torch.int64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    return tuple(int(v) for v in t.tolist())


class GradientAverager:
    """Bucketed all-reduce(sum)/world of parameter gradients, overlapped with the backward pass
    (``prepare`` before ``backward``, ``average`` after it).

    The bucket plan is STATIC (built from the parameter list, identical on every rank), and a
    parameter whose grad is None contributes zeros -- the sequence model owns parameters that its
    forward never touches (``receptor_feature_emb``, reference sequence_model/model.py:176 vs 221),
    which would desynchronise a plan built from the non-None grads."""

    def __init__(self, params, bucket_bytes=64 << 20, overlap=True):
        self.params = [p for p in params if p.requires_grad]
        # buckets follow the REVERSE parameter order: gradients become ready roughly back to front, so the
        # first buckets to complete are the first to go on the wire while the backward pass continues
        self.buckets, cur, size = [], [], 0
        for p in reversed(self.params):
            cur.append(p)
            size += p.numel() * 4
            if size >= bucket_bytes:
                self.buckets.append(cur)
                cur, size = [], 0
        if cur:
            self.buckets.append(cur)
        self._bucket_of = {id(p): i for i, b in enumerate(self.buckets) for p in b}
        self._flat, self._handles, self._pending, self._bound, self._touched = {}, {}, None, False, set()
        # the view / overlap path of ``prepare`` needs to hear about every gradient (hooks, or ``mark_ready``):
        # without hooks it would not know which parameters backward touched, so ``prepare`` then leaves the
        # gradients alone and ``average`` takes the copying path
        self._hooked = bool(overlap) and hasattr(torch.Tensor, "register_post_accumulate_grad_hook")
        if self._hooked:
            for p in self.params:
                p.register_post_accumulate_grad_hook(self._on_grad)

    @staticmethod
    def _active():
        # E3D_DDP_SINGLE_RANK=1: a one-rank process group still takes the whole data-parallel path (buckets, hooks, collectives
        # over a group of one) -- how a one-GPU box exercises the RCCL calls (tools/lab/rccl_single_rank_step.py)
        return dist.is_initialized() and (dist.get_world_size() > 1 or os.environ.get("E3D_DDP_SINGLE_RANK") == "1")

    def _buffer(self, i):
        bucket = self.buckets[i]
        flat = self._flat.get(i)
        if flat is None or flat.device != bucket[0].device:
            flat = self._flat[i] = torch.empty(sum(p.numel() for p in bucket), dtype=torch.float32, device=bucket[0].device)
        return flat

    def prepare(self):
        """Call after ``zero_grad`` and before ``backward``: every ``.grad`` becomes a zeroed VIEW of its
        bucket's flat buffer (autograd then accumulates in place: no flatten / unflatten copies), and a bucket
        goes on the wire -- asynchronously, overlapping the rest of the backward pass -- the moment its last
        gradient has been accumulated.  Without this call ``average`` still works (copying path)."""
        if not self._active() or not self._hooked:
            return
        for i, bucket in enumerate(self.buckets):
            flat = self._buffer(i)
            flat.zero_()
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self._pending = [len(b) for b in self.buckets]
        self._handles, self._bound, self._touched = {}, True, set()

    # ---- graph-segment form (training.GraphedDDPStep): the same buckets, no collective launched from a hook
    def bind(self, collect_only=False):
        """``prepare`` for a backward pass that is being CAPTURED: every ``.grad`` becomes a zeroed view of its bucket (the
        zero fills are captured with the pass), the hooks only note which parameters the pass touches."""
        for i, bucket in enumerate(self.buckets):
            flat = self._buffer(i)
            flat.zero_()
            off = 0
            for p in bucket:
                p.grad = flat[off:off + p.numel()].view_as(p)
                off += p.numel()
        self._pending = [1 << 30] * len(self.buckets) if collect_only else [len(b) for b in self.buckets]
        self._handles, self._bound, self._touched = {}, True, set()

    def finish_collect(self):
        self._bound = False
        self._drop_unused()

    def flats(self):
        return [self._flat[i] for i in range(len(self.buckets))]

    def all_reduce_flats(self):
        """Sum every bucket over the ranks (the division by the world size belongs to the caller's captured segment);
        asynchronous on the wire, in bucket order, joined before returning to the caller's stream."""
        self._order_for_host_backend(self._flat[0])
        works = [dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, async_op=True) for i in range(len(self.buckets))]
        for w in works:
            w.wait()

    def _drop_unused(self):
        """A parameter no rank's backward touched (the hook never fired anywhere: the autograd graph is the same
        on every rank) gets ``grad = None`` back, as in a single-process run: the optimizer then skips it
        (no weight decay on e.g. the sequence model's never-used ``receptor_feature_emb``), so single- and
        multi-rank training produce the same checkpoint entries."""
        for p in self.params:
            if id(p) not in self._touched:
                p.grad = None

    def _on_grad(self, p):
        if not self._bound or id(p) in self._touched:    # once per parameter and step: a weight may be reported both by
            return                                       # its hook and by ``mark_ready`` (mixed deferred / direct uses)
        self._touched.add(id(p))
        i = self._bucket_of[id(p)]
        self._pending[i] -= 1
        if self._pending[i] == 0:
            self._launch(i)

    def mark_ready(self, p):
        """For gradients written outside autograd's accumulation (autograd.deferred_weight_grads(on_param=...)): the
        post-accumulate-grad hook never fires for them."""
        self._on_grad(p)

    @staticmethod
    def _order_for_host_backend(flat):
        """gloo on GPU tensors (the one-card rehearsals; production is RCCL): its device-to-host copy runs on a pool stream that
        waits for an event recorded on the caller's stream -- and was seen reading a bucket before the gradient kernels queued
        ahead of that event had finished (ROCm 7.2, torch 2.10: ranks 1e-5 apart after a few steps, tools/lab/ddp_second_run_diff.py;
        gone with this wait).  So the host waits for the caller's stream first.  Costs the rehearsal its overlap, nothing else."""
        if flat.is_cuda and dist.get_backend() == "gloo":
            torch.cuda.current_stream(flat.device).synchronize()

    def _launch(self, i):
        self._order_for_host_backend(self._flat[i])
        self._handles[i] = dist.all_reduce(self._flat[i], op=dist.ReduceOp.SUM, async_op=True)

    def average(self):
        if not self._active():
            return
        world = dist.get_world_size()
        if self._bound:
            # buckets that hold a never-used parameter (its hook never fires) go last, in index order: the
            # set and the order are the same on every rank because the autograd graph is
            for i in range(len(self.buckets)):
                if i not in self._handles:
                    self._launch(i)
            for i in range(len(self.buckets)):
                self._handles[i].wait()
                self._flat[i].div_(world)
            self._bound = False
            self._drop_unused()
            return
        handles = []
        for i, bucket in enumerate(self.buckets):
            flat = self._buffer(i)
            off = 0
            for p in bucket:
                dst = flat[off:off + p.numel()]
                if p.grad is None:
                    dst.zero_()
                else:
                    dst.copy_(p.grad.reshape(-1))
                off += p.numel()
            self._order_for_host_backend(flat)
            handles.append((dist.all_reduce(flat, op=dist.ReduceOp.SUM, async_op=True), flat, bucket))
        for work, flat, bucket in handles:     # buckets overlap each other on the wire
            work.wait()
            off = 0
            for p in bucket:
                g = flat[off:off + p.numel()].view_as(p) / world
                if p.grad is not None:     # never-used parameters keep grad None (same graph on every rank)
                    p.grad.copy_(g)
                off += p.numel()


def broadcast_parameters(module, src=0):
    """Make every rank start from rank ``src``'s weights (DDP construction semantics)."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return
    # the collective writes behind autograd's back (no version bump, and ``.data`` has a counter of its own), so
    # the received values are copied in with an ordinary in-place op: ``_version`` moves and the inference caches
    # keyed on it (packed QKV, W^T, distance-table planes) are rebuilt on ranks that ran a forward before fit()
    with torch.no_grad():
        for t in list(module.parameters()) + list(module.buffers()):
            recv = t.detach().clone()
            dist.broadcast(recv, src=src)
            t.copy_(recv)
    for m in module.modules():
        m.__dict__.pop("_e3d_pack", None)
