"""Differentiable versions of the fused ops: torch.autograd.Function shells whose forward AND
backward are the HIP kernels (the reference trains through torch.autograd on stock ops; here every
fused forward kernel has a hand-written backward kernel behind the same C-ABI).

``functional.*`` entry points pick the plain inference op when no gradient is required, so the
sampling path pays nothing for the training support.
"""
import os

import torch

from . import hip, ops
from .ops import _chk, _p, _stream


# ----------------------------------------------------------------------------- raw backward wrappers
def gemm_general(a, a_kmajor, b, b_kmajor, M, N, K, bias=None, act=ops.ACT_NONE, mode=None):
    """out[M,N] = act(A . B^T + bias) with per-operand storage order (include/e3d_hip.h)."""
    _chk(a, "gemm_general.a"); _chk(b, "gemm_general.b"); _chk(bias, "gemm_general.bias")
    assert a.dim() == 2 and b.dim() == 2 and a.stride(1) == 1 and b.stride(1) == 1
    out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    terms = ops.GEMM_MODES[ops.GEMM_MODE if mode is None else mode] or 6   # exact-f32 mode: fp32-grade split
    hip.check(hip.lib().e3d_gemm_f32_split_general(_p(a), a.stride(0), int(a_kmajor), _p(b), b.stride(0),
                                                   int(b_kmajor), _p(bias), _p(out), out.stride(0), M, N, K, act,
                                                   terms, _stream()), "e3d_gemm_f32_split_general")
    return out


# ----------------------------------------------------------------------------- deferred weight gradients
class deferred_weight_grads:
    """``with deferred_weight_grads() as q: loss.backward()`` -- inside the block the backward of every linear layer
    computes only its input gradient and QUEUES its weight / bias gradient (the (dz, x) pair stays alive); leaving the
    block (or ``q.flush()``) computes all queued gradients, the layers of one shape together in one launch
    (``e3d_gemm_wgrad_grouped_f32_split``: whole reductions over the tokens instead of split-K slices meeting through
    atomics -- 82 -> 250 TFLOP/s on a 768x768 weight at 4096 tokens -- with the bias column sums riding along), and
    writes them straight into ``param.grad`` (added to an existing gradient, like autograd's accumulation).
    Nothing reaches autograd's own accumulation for these parameters, so post-accumulate-grad hooks do not fire for
    them: ``on_param`` (called per parameter at flush time, after its gradient is enqueued) is where a gradient
    averager learns that the gradient is there (sharding.GradientAverager.mark_ready); the queue is then worked off in
    ``stages`` slices, last layers first, so that the first buckets travel while the later slices compute."""
    active = None

    # The queued (dz, x) pairs stay alive until the flush -- without the queue autograd frees each pair as backward
    # proceeds.  At BASELINE config 2 (4096 tokens, 12 + 12 layers) that is ~2.6 GB of extra peak memory, growing linearly
    # with B x L; when the queued bytes pass this budget the queue is flushed early (what is queued so far is grouped
    # and computed, backward continues with an empty queue).  E3D_DEFER_WGRAD_MAX_GB, default 16 (of 288 GB of HBM).
    MAX_BYTES = int(float(os.environ.get("E3D_DEFER_WGRAD_MAX_GB", "16")) * 2 ** 30)

    # OPT-IN (E3D_WGRAD_OVERLAP=1), measured and not kept as the default: in a single process every
    # OVERLAP_CHUNK queued layers can be computed at once on a SIDE stream while backward carries on with the input-gradient
    # chain of the earlier layers on the caller's stream (a slice needs nothing but its saved (dz, x); the streams join before
    # the last slice, i.e. before the optimizer; a captured step keeps the fork / join as graph dependencies; every problem
    # of a grouped launch is a whole reduction, so the gradients are bit-identical to the serial order -- tested).  The
    # idea was to fill the quarter of the chip the M = 4096 input-gradient launches leave idle.  Measured on MI355X, graph
    # replay, same box: structure 24.9 -> 25.3 ms, sequence 19.7 -> 19.6 ms (profiles/r04_train_wgrad_overlap_ab.log): a
    # 512-thread weight-gradient workgroup (96 KB of LDS) and an input-gradient workgroup (84 KB) cannot share a CU, so the
    # two streams take turns on the CUs instead of overlapping, and the chain on the caller's stream is what gets delayed.
    OVERLAP = os.environ.get("E3D_WGRAD_OVERLAP", "0") == "1"
    OVERLAP_CHUNK = int(os.environ.get("E3D_WGRAD_CHUNK", "24"))
    _SIDE = {}

    def __init__(self, on_param=None, stages=3, overlap=None):
        self.pending, self.on_param, self.stages = [], on_param, max(1, int(stages))
        self.queued_bytes, self.early_flushes, self._seen_ptrs, self._early_touched = 0, 0, set(), {}
        self.overlap = (self.OVERLAP if overlap is None else bool(overlap)) and on_param is None
        self.side_flushes, self._side, self.queued_total = 0, None, 0

    def _flush_on_side_stream(self, items):
        dev = items[0][0].device
        main = torch.cuda.current_stream(dev)
        side = self._SIDE.get(dev.index)
        if side is None:
            side = self._SIDE[dev.index] = torch.cuda.Stream(device=dev)
        side.wait_stream(main)                     # the slice's dz / x were written on the caller's stream
        with torch.cuda.stream(side):
            touched = self._flush_slice(items)
        for dz, x, _, _ in items:                  # freed by the caller's stream while the side stream may still read them
            dz.record_stream(side)
            if x is not None:
                x.record_stream(side)
        for p_ in touched.values():                # allocated on the side stream, consumed by the optimizer on the caller's
            if p_.grad is not None:
                p_.grad.record_stream(main)
        self._side, self.side_flushes = side, self.side_flushes + 1
        return touched

    def __enter__(self):
        assert deferred_weight_grads.active is None, "deferred_weight_grads blocks do not nest"
        deferred_weight_grads.active = self
        return self

    def __exit__(self, exc_type, *_):
        deferred_weight_grads.active = None
        if exc_type is None:
            self.flush()
        self.pending = []

    def add(self, dz, x, weight, bias):
        self.pending.append((dz, x, weight, bias))
        self.queued_total += 1
        for t in (dz, x):      # (the row blocks of a packed weight share one dz / x: count a storage once)
            if t is None:
                continue
            key = t.untyped_storage().data_ptr()
            if key not in self._seen_ptrs:
                self._seen_ptrs.add(key)
                self.queued_bytes += t.untyped_storage().nbytes()
        if self.overlap and dz.is_cuda and len(self.pending) >= self.OVERLAP_CHUNK:
            pending, self.pending = self.pending, []
            self.queued_bytes, self._seen_ptrs = 0, set()
            self._flush_on_side_stream(pending)
            return
        if self.queued_bytes > self.MAX_BYTES:
            # early flush: compute what is queued, but report nothing to a listening averager yet -- a weight used
            # twice in the forward pass (sequence model: ligand_feature_emb) may still receive its second contribution
            self.early_flushes += 1
            pending, self.pending = self.pending, []
            self.queued_bytes, self._seen_ptrs = 0, set()
            self._early_touched.update(self._flush_slice(pending))

    MIN_TILES = int(os.environ.get("E3D_WGRAD_MIN_TILES", "96"))   # 256x128 output tiles a grouped launch needs (of 256 CUs; sweep in DESIGN.md)

    @staticmethod
    def _single(dz, x, w, b, N, K, M):
        """One layer on its own: the split-K weight-gradient launch + column sums, accumulated like autograd would."""
        for p, g in ((w, gemm_general(dz, True, x, True, N, K, M)), (b, colsum(dz) if b is not None else None)):
            if p is None:
                continue
            if p.grad is None:
                p.grad = g
            else:
                p.grad.add_(g)

    @staticmethod
    def _grad_buffer(p):
        """(tensor to write, accumulate?) -- allocates ``p.grad`` if the parameter has none yet."""
        if p.grad is None:
            p.grad = torch.empty_like(p, memory_format=torch.contiguous_format)
            return p.grad, False
        assert p.grad.is_contiguous() and p.grad.dtype == torch.float32
        return p.grad, True

    def flush(self):
        pending, self.pending = self.pending, []
        self.queued_bytes, self._seen_ptrs = 0, set()
        if self._side is not None:                 # join: the last slice (and the optimizer) follow every earlier one
            torch.cuda.current_stream(self._side.device).wait_stream(self._side)
            self._side = None
        early, self._early_touched = self._early_touched, {}
        if self.on_param is not None:     # parameters finished by an early flush and not queued again: complete now
            still = {id(t) for item in pending for t in item[2:] if t is not None}
            for pid, p_ in early.items():
                if pid not in still:
                    self.on_param(p_)
        if not pending:
            return
        # With a gradient averager listening (on_param), the queue is worked off in ``stages`` slices, last layers
        # first (the order backward queued them = the order of the averager's buckets): a slice's parameters are
        # reported as soon as its launches are enqueued, so their buckets go on the wire while the next slice computes.
        # Alone (one process) one slice: the largest groups.
        stages = self.stages if self.on_param is not None else 1
        per = -(-len(pending) // stages)
        slices = [pending[i:i + per] for i in range(0, len(pending), per)]
        last_slice = {}
        for si, sl in enumerate(slices):
            for _, _, w, b in sl:
                last_slice[id(w)] = si
                if b is not None:
                    last_slice[id(b)] = si
        for si, sl in enumerate(slices):
            touched = self._flush_slice(sl)
            if self.on_param is not None:
                for p in touched.values():       # once per parameter, after its last use
                    if last_slice[id(p)] == si:
                        self.on_param(p)

    # One launch for layers of different shapes / strides (e3d_gemm_wgrad_ragged_f32_split); 0: one launch per
    # (shape, stride) group as in round 2 (A/B runs)
    RAGGED = os.environ.get("E3D_WGRAD_RAGGED", "1") == "1"

    def _flush_ragged(self, items, M, terms, touched):
        """All queued layers of one token count, 64 per launch, whatever their shapes and row strides."""
        import ctypes
        lib = hip.lib()
        tiles = sum((-(-it[0].shape[1] // 256)) * (-(-it[1].shape[1] // 128)) for it in items)
        if tiles < self.MIN_TILES:
            for dz, x, w, b in items:
                self._single(dz, x, w, b, dz.shape[1], x.shape[1], M)
                touched[id(w)] = w
                if b is not None:
                    touched[id(b)] = b
            return
        mixed = [it for it in items if it[3] is not None and (it[2].grad is None) != (it[3].grad is None)]
        for dz, x, w, b in mixed:     # one accumulate bit per problem covers weight and bias: these take the per-layer path
            self._single(dz, x, w, b, dz.shape[1], x.shape[1], M)
            touched[id(w)] = w
            touched[id(b)] = b
        if mixed:
            items = [it for it in items if not any(it is m for m in mixed)]
        for lo in range(0, len(items), 64):
            chunk = items[lo:lo + 64]
            n = len(chunk)
            a_dz, a_x, a_dw, a_db = ((ctypes.c_void_p * n)() for _ in range(4))
            a_n, a_k = (ctypes.c_int * n)(), (ctypes.c_int * n)()
            a_ldz, a_ldx = (ctypes.c_int64 * n)(), (ctypes.c_int64 * n)()
            bits = 0
            for i, (dz, x, w, b) in enumerate(chunk):
                gw, acc_w = self._grad_buffer(w)
                a_dz[i], a_x[i], a_dw[i] = dz.data_ptr(), x.data_ptr(), gw.data_ptr()
                if b is not None:
                    gb, acc_b = self._grad_buffer(b)
                    assert acc_b == acc_w
                    a_db[i] = gb.data_ptr()
                a_n[i], a_k[i], a_ldz[i], a_ldx[i] = dz.shape[1], x.shape[1], dz.stride(0), x.stride(0)
                bits |= int(acc_w) << i
                touched[id(w)] = w
                if b is not None:
                    touched[id(b)] = b
            hip.check(lib.e3d_gemm_wgrad_ragged_f32_split(a_dz, a_x, a_dw, a_db, a_n, a_k, a_ldz, a_ldx, bits, n, M, terms,
                                                          _stream()), "e3d_gemm_wgrad_ragged_f32_split")

    def _flush_slice(self, pending):
        import ctypes
        terms = ops.GEMM_MODES[ops.GEMM_MODE] or 6
        lib = hip.lib()
        if self.RAGGED:
            # a weight used more than once in the forward pass: its later uses go to later rounds (a launch must not hold
            # two problems with the same output), each adding to what the earlier rounds wrote
            rounds, seen, touched = [], {}, {}
            for item in pending:
                r = seen.get(id(item[2]), 0)
                seen[id(item[2])] = r + 1
                while len(rounds) <= r:
                    rounds.append({})
                rounds[r].setdefault(item[0].shape[0], []).append(item)       # by token count
            for by_m in rounds:
                for M, items in by_m.items():
                    self._flush_ragged(items, M, terms, touched)
            return touched
        # a weight used more than once in the forward pass: its later uses go to later rounds (a launch must not
        # hold two problems with the same output), each adding to what the earlier rounds wrote
        rounds, seen = [], {}
        for item in pending:
            r = seen.get(id(item[2]), 0)
            seen[id(item[2])] = r + 1
            while len(rounds) <= r:
                rounds.append({})
            dz, x = item[0], item[1]
            key = (dz.shape[1], x.shape[1], dz.shape[0], dz.stride(0), x.stride(0), item[3] is not None)
            rounds[r].setdefault(key, []).append(item)
        touched = {}
        for groups in rounds:
            for (N, K, M, ldz, ldx, has_bias), items in groups.items():
                if len(items) * (-(-N // 256)) * (-(-K // 128)) < self.MIN_TILES:
                    # too few output tiles even together (a shape only one or two layers have): whole reductions would
                    # leave most of the chip idle -- the per-layer split-K launch is the better kernel for these
                    for dz, x, w, b in items:
                        self._single(dz, x, w, b, N, K, M)
                        touched[id(w)] = w
                        if has_bias:
                            touched[id(b)] = b
                    continue
                # a layer whose weight already holds a gradient while its bias does not (or the reverse: a bias shared
                # with a layer outside the queue, a gradient set by hand) cannot ride in a grouped launch -- one
                # accumulate bit per problem covers both outputs -- so it takes the per-layer path
                mixed = [it for it in items if has_bias and (it[2].grad is None) != (it[3].grad is None)]
                for dz, x, w, b in mixed:
                    self._single(dz, x, w, b, N, K, M)
                    touched[id(w)] = w
                    touched[id(b)] = b
                if mixed:
                    items = [it for it in items if not any(it is m for m in mixed)]
                for lo in range(0, len(items), 64):
                    chunk = items[lo:lo + 64]
                    n = len(chunk)
                    arr = lambda: (ctypes.c_void_p * n)()   # noqa: E731
                    a_dz, a_x, a_dw, a_db, bits = arr(), arr(), arr(), arr(), 0
                    for i, (dz, x, w, b) in enumerate(chunk):
                        gw, acc_w = self._grad_buffer(w)
                        a_dz[i], a_x[i], a_dw[i] = dz.data_ptr(), x.data_ptr(), gw.data_ptr()
                        if has_bias:
                            gb, acc_b = self._grad_buffer(b)
                            assert acc_b == acc_w      # (mixed layers were taken out above)
                            a_db[i] = gb.data_ptr()
                        bits |= int(acc_w) << i
                    hip.check(lib.e3d_gemm_wgrad_grouped_f32_split(a_dz, a_x, a_dw, a_db if has_bias else None, bits, n, ldz,
                                                                   ldx, N, K, M, terms, _stream()),
                              "e3d_gemm_wgrad_grouped_f32_split")
                    for _, _, w, b in chunk:
                        touched[id(w)] = w
                        if has_bias:
                            touched[id(b)] = b
        return touched


def colsum(x):
    out = torch.empty((x.shape[1],), device=x.device, dtype=torch.float32)
    hip.check(hip.lib().e3d_colsum(_p(x), x.stride(0), _p(out), x.shape[0], x.shape[1], _stream()), "e3d_colsum")
    return out


def group_sum(x, rows_per_group):
    M, H = x.shape
    out = torch.empty((M // rows_per_group, H), device=x.device, dtype=torch.float32)
    hip.check(hip.lib().e3d_group_sum(_p(x), rows_per_group, _p(out), M, H, _stream()), "e3d_group_sum")
    return out


def act_fwd(z, act):
    out = torch.empty_like(z)
    hip.check(hip.lib().e3d_act_fwd(_p(z), act, _p(out), z.numel(), _stream()), "e3d_act_fwd")
    return out


def act_bwd(dh, z, act):
    dz = torch.empty_like(z)
    hip.check(hip.lib().e3d_act_bwd(_p(dh), _p(z), act, _p(dz), z.numel(), _stream()), "e3d_act_bwd")
    return dz


def layernorm_bwd(dy, s, gamma, eps, want_affine_grads=True, drop=None):
    """``drop`` = (p, seed) of a dropout folded into the forward (ops.residual_layernorm(..., drop=)): returns
    (ds, dg, db, ds_dropped) -- ds_dropped = dropout(ds, p, seed), the gradient of the dropped-out input."""
    M, H = s.shape
    ds = torch.empty_like(s)
    dg = db = None
    ws, n_ws = None, 0
    if want_affine_grads:      # dgamma | dbeta in one allocation; per-block partial sums meet in ``ws`` (no atomics, no zero-fill)
        both = torch.empty((2, H), device=s.device, dtype=torch.float32)
        dg, db = both[0], both[1]
        n_ws = hip.lib().e3d_layernorm_bwd_workspace_floats(M, H)
        ws = torch.empty((n_ws,), device=s.device, dtype=torch.float32)
    dsd = torch.empty_like(s) if drop is not None else None
    p, seed = (float(drop[0]), int(drop[1])) if drop is not None else (0.0, 0)
    hip.check(hip.lib().e3d_layernorm_bwd_ws(_p(dy), _p(s), _p(gamma), eps, _p(ds), _p(dsd), _p(dg), _p(db), M, H, p, seed,
                                             _p(ws), n_ws, _stream()), "e3d_layernorm_bwd_ws")
    if drop is not None:
        return ds, dg, db, dsd
    return ds, dg, db


def small_k_wgrad(g, x, transpose_out=False, want_bias=True):
    M, H = g.shape
    Fk = x.shape[1]
    dW = torch.empty((Fk, H) if transpose_out else (H, Fk), device=g.device, dtype=torch.float32)
    db = torch.empty((H,), device=g.device, dtype=torch.float32) if want_bias else None
    hip.check(hip.lib().e3d_small_k_wgrad(_p(g), _p(x), _p(dW), _p(db), M, H, Fk, int(transpose_out), _stream()),
              "e3d_small_k_wgrad")
    return dW, db


# ----------------------------------------------------------------------------- autograd Functions
class _WtEntry:
    __slots__ = ("refs", "wt", "key")

    def __init__(self, parts):
        import weakref
        self.refs, self.wt, self.key = [weakref.ref(p) for p in parts], None, None

    def parts(self):
        ps = [r() for r in self.refs]
        return None if any(p is None for p in ps) else ps


_WT_REGISTRY = {}     # ids of the parameters behind a weight -> _WtEntry
_WT_RETIRED = []     # entries dropped from the registry while a captured graph may still use their buffers


def _refresh_transposes(entries):
    """W^T of every entry in grouped launches (e3d_transpose_grouped_f32: one per weight shape); a packed weight's
    parts land side by side in its [K, N_total] buffer."""
    import ctypes
    groups = {}
    for ent, parts in entries:
        n_total, k = sum(p.shape[0] for p in parts), parts[0].shape[1]
        if ent.wt is None or ent.wt.shape != (k, n_total) or ent.wt.device != parts[0].device:
            ent.wt = torch.empty((k, n_total), device=parts[0].device, dtype=torch.float32)
        r0 = 0
        for p in parts:
            assert p.is_contiguous() and p.shape[1] == k
            if p.is_cuda:
                groups.setdefault((p.shape[0], k, n_total), []).append((p.data_ptr(), ent.wt.data_ptr() + 4 * r0))
            else:   # host tensors (the cache-logic test): nothing here can launch, and ops.gemm refuses them anyway
                ent.wt[:, r0:r0 + p.shape[0]] = p.detach().t()
            r0 += p.shape[0]
        ent.key = ops.weight_key(*parts)
    for (rows, cols, ld_dst), items in groups.items():
        for lo in range(0, len(items), 64):
            chunk = items[lo:lo + 64]
            src = (ctypes.c_void_p * len(chunk))(*[c[0] for c in chunk])
            dst = (ctypes.c_void_p * len(chunk))(*[c[1] for c in chunk])
            hip.check(hip.lib().e3d_transpose_grouped_f32(src, dst, len(chunk), rows, cols, cols, ld_dst, _stream()),
                      "e3d_transpose_grouped_f32")


def forget_transposes(keep):
    """Drop every registered W^T whose parameters are not all in ``keep`` (a set of ``id(parameter)``).  A training step
    about to be captured into a graph calls this (training.GraphedStep): the refresh that the first stale use triggers
    covers EVERY stale entry of the registry, and a captured step must not bake launches that read another model's
    weights and write buffers that die with that model."""
    for key in list(_WT_REGISTRY):
        ps = _WT_REGISTRY[key].parts()
        if ps is None or not all(id(p) in keep for p in ps):
            # parked, not freed: another model's GraphedStep may still be alive, and its graph baked both the
            # refresh launches that WRITE this buffer and the input-gradient GEMMs that READ it -- as long as the buffer
            # stays allocated that graph remains self-consistent; returned to the allocator it would be written into memory
            # someone else owns.  (As ops._SKINNY_RETIRED; release_retired_transposes() frees them when no graph is left.)
            _WT_RETIRED.append(_WT_REGISTRY.pop(key))


def release_retired_transposes():
    """Free the W^T buffers forget_transposes parked (call when no captured training step of another model is alive)."""
    _WT_RETIRED.clear()


def _transposed_weight(weight):
    """W^T [K,N] contiguous for the input-gradient GEMM dz . W, which then runs in the forward (both operands
    K-contiguous) layout (~25 % faster than reading W "K-major", one dword per lane).  Weights are registered on first
    use; when one is found stale (ops.weight_key: an optimizer step changes the generation) EVERY registered weight is
    re-transposed in the same few grouped launches -- one per weight shape and step instead of one copy kernel per layer
    (72 x 8.7 us in a structure training step).  A packed weight (bert._packed: cat of query / key / value, a new tensor
    every step) is registered through the parameters behind it.  Gradient accumulation over several backward passes
    re-uses the transposes."""
    parts = getattr(weight, "_e3d_parts", None) or [weight]
    rid = tuple(id(p) for p in parts)
    ent = _WT_REGISTRY.get(rid)
    if ent is None or ent.parts() is None or any(a is not b for a, b in zip(ent.parts(), parts)):
        ent = _WT_REGISTRY[rid] = _WtEntry(parts)
    if ent.key == ops.weight_key(*parts):
        return ent.wt
    stale = [(ent, parts)]
    for key in list(_WT_REGISTRY):
        other = _WT_REGISTRY[key]
        if other is ent:
            continue
        ps = other.parts()
        if ps is None:
            del _WT_REGISTRY[key]          # its parameters are gone
        elif other.key != ops.weight_key(*ps) and ps[0].device == parts[0].device:
            stale.append((other, ps))
    with torch.no_grad():
        _refresh_transposes(stale)
    return ent.wt



class _Linear(torch.autograd.Function):
    """y = act(x W^T + b); training keeps the pre-activation z (the inference path fuses act into
    the GEMM epilogue instead)."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, absmax=None):
        z = ops.gemm(x, weight, bias, ops.ACT_NONE, absmax=absmax, prescale=False)   # the weight changes every step
        ctx.act = act
        ctx.save_for_backward(x, weight, z if act != ops.ACT_NONE else None)
        ctx.has_bias = bias is not None
        # deferred weight gradients write into the PARAMETERS: the bias itself, and for a packed weight
        # (bert._packed: cat of query / key / value) the parameters behind its row blocks
        ctx.bias = bias if (bias is not None and bias.requires_grad) else None
        ctx.w_parts, ctx.b_parts = getattr(weight, "_e3d_parts", None), getattr(bias, "_e3d_parts", None)
        return act_fwd(z, act) if act != ops.ACT_NONE else z

    @staticmethod
    def backward(ctx, dy):
        x, weight, z = ctx.saved_tensors
        dy = dy.contiguous()
        dz = act_bwd(dy, z, ctx.act) if ctx.act != ops.ACT_NONE else dy
        M, K = x.shape
        N = weight.shape[0]
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            # (few rows against a large weight -- the adaLN modulation of the per-item conditioning, M = batch size, W [6 H, H] --
            #  also go through W^T: the K-major read of W took 141 us per step there, the skinny kernel on W^T ~10)
            if K % 128 == 0 and N % 32 == 0 and (M >= 1024 or N * K >= (1 << 21)):
                dx = ops.gemm(dz, _transposed_weight(weight), None)      # dz [M,N] . (W^T [K,N])^T
            else:
                dx = gemm_general(dz, False, weight, True, M, K, N)      # dz [M,N] . W[N,K]
        q = deferred_weight_grads.active
        if q is not None and ctx.needs_input_grad[1] and dz.stride(1) == 1 and x.stride(1) == 1 and K >= 32:
            # (weight parameter, bias parameter or None, first row) of each row block of the weight
            if ctx.w_parts is not None:
                bs = ctx.b_parts if ctx.has_bias else [None] * len(ctx.w_parts)
                blocks = list(zip(ctx.w_parts, bs)) if bs is not None and len(bs) == len(ctx.w_parts) else None
            else:
                blocks = [(weight, ctx.bias if ctx.has_bias else None)]
            ok = blocks is not None and all(w.is_leaf and w.requires_grad and (b is None or (b.is_leaf and b.requires_grad))
                                            for w, b in blocks) and (not ctx.has_bias or blocks[0][1] is not None)
            if ok:
                r0 = 0
                for w, b in blocks:                  # computed with their peers at flush time
                    q.add(dz[:, r0:r0 + w.shape[0]], x, w, b)
                    r0 += w.shape[0]
                assert r0 == N
                return dx, None, None, None, None
        if ctx.needs_input_grad[1]:
            dw = gemm_general(dz, True, x, True, N, K, M)                # dz^T [N,M] . x [M,K]
        if ctx.has_bias and ctx.needs_input_grad[2]:
            db = colsum(dz)
        return dx, dw, db, None, None


class _Attention(torch.autograd.Function):
    """Fused attention on PACKED projections so that the gradients come back packed too:
    self-attention: ``q_src`` = qkv [B*L, 3H], ``kv_src`` = None -> grad dqkv [B*L, 3H];
    cross-attention: ``q_src`` = q [B*Lq, H], ``kv_src`` = kv [B*Lk, 2H] -> grads dq, dkv."""

    @staticmethod
    def _views(q_src, kv_src, H):
        if kv_src is None:
            return q_src[:, :H], q_src[:, H:2 * H], q_src[:, 2 * H:]
        return q_src, kv_src[:, :H], kv_src[:, H:]

    @staticmethod
    def forward(ctx, q_src, kv_src, dist_emb, key_mask, B, nh, Lq, Lk, max_pos, drop_p=0.0, bounds=None):
        H = nh * 64
        q, k, v = _Attention._views(q_src, kv_src, H)
        drop = (float(drop_p), ops.next_dropout_seed()) if drop_p > 0 else (0.0, 0)
        out, lse = ops.attention(q, k, v, B, nh, Lq, Lk, key_mask=key_mask, dist_emb=dist_emb, max_pos=max_pos,
                                 want_lse=True, drop=drop, bounds=bounds)
        ctx.save_for_backward(q_src, kv_src, dist_emb, key_mask, out, lse)
        ctx.dims = (B, nh, Lq, Lk, max_pos)
        ctx.drop = drop
        ctx.terms = ops.GEMM_MODES[ops.ATTN_MODE]   # the backward products run in the forward's arithmetic
        if ctx.terms == 19:                          # f16x3 is a forward-only arithmetic: exact fp32 backward kernels
            ctx.terms = 6
        return out

    @staticmethod
    def backward(ctx, dout):
        q_src, kv_src, dist_emb, key_mask, out, lse = ctx.saved_tensors
        B, nh, Lq, Lk, max_pos = ctx.dims
        H = nh * 64
        dout = dout.contiguous()
        q, k, v = _Attention._views(q_src, kv_src, H)
        dq_src = torch.empty_like(q_src)
        dkv_src = torch.empty_like(kv_src) if kv_src is not None else None
        dq, dk, dv = _Attention._views(dq_src, dkv_src, H)
        dE = torch.empty_like(dist_emb) if dist_emb is not None else None
        lib = hip.lib()
        ws = torch.empty((lib.e3d_relkey_attn_bwd_workspace_floats(B, nh, Lq, Lk, int(dist_emb is not None)),),
                         device=q.device, dtype=torch.float32)
        with ops._timed("attn_bwd", (B, nh, Lq, Lk)):
            hip.check(lib.e3d_relkey_attn_bwd_ex(
                _p(q), Lq * q.stride(0), q.stride(0), _p(k), Lk * k.stride(0), k.stride(0), _p(v), Lk * v.stride(0),
                v.stride(0), _p(dist_emb), max_pos, _p(key_mask), _p(out), _p(lse), _p(dout),
                _p(dq), Lq * dq.stride(0), dq.stride(0), _p(dk), Lk * dk.stride(0), dk.stride(0),
                _p(dv), Lk * dv.stride(0), dv.stride(0), _p(dE), _p(ws), B, nh, Lq, Lk, ctx.terms, ctx.drop[0], ctx.drop[1],
                _stream()), "e3d_relkey_attn_bwd_ex")
        return dq_src, dkv_src, dE, None, None, None, None, None, None, None, None


class _Dropout(torch.autograd.Function):
    """y = x * keep / (1 - p'); the gradient passes through the same decisions (regenerated from the seed)."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p, ctx.seed = float(p), ops.next_dropout_seed()
        return ops.dropout(x.contiguous(), ctx.p, ctx.seed)

    @staticmethod
    def backward(ctx, dy):
        return ops.dropout(dy.contiguous(), ctx.p, ctx.seed), None


class _ResidualLayerNorm(torch.autograd.Function):
    """LayerNorm(dropout(x) + residual): ``p_drop`` > 0 folds the nn.Dropout that follows the dense layer of
    BertSelfOutput / BertOutput into the LayerNorm kernels (no [M, H] dropout pass in either direction)."""

    @staticmethod
    def forward(ctx, x, residual, gamma, beta, eps, p_drop=0.0):
        ctx.drop = (float(p_drop), ops.next_dropout_seed()) if p_drop and p_drop > 0 else None
        out, s = ops.residual_layernorm(x, residual, gamma, beta, eps, want_s=True, drop=ctx.drop)
        ctx.save_for_backward(s, gamma)
        ctx.eps, ctx.has_res = eps, residual is not None
        return out

    @staticmethod
    def backward(ctx, dy):
        s, gamma = ctx.saved_tensors
        if ctx.drop is not None:
            ds, dg, db, dsd = layernorm_bwd(dy.contiguous(), s, gamma, ctx.eps, drop=ctx.drop)
            return dsd, (ds if ctx.has_res else None), dg, db, None, None
        ds, dg, db = layernorm_bwd(dy.contiguous(), s, gamma, ctx.eps)
        return ds, (ds if ctx.has_res else None), dg, db, None, None


class _AdaLNGate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, y, mod, branch, rows_per_cond):
        ctx.save_for_backward(y, mod)
        ctx.branch, ctx.rpc = branch, rows_per_cond
        return ops.adaln_gate(x, y, mod, branch, rows_per_cond)

    @staticmethod
    def backward(ctx, dout):
        y, mod = ctx.saved_tensors
        dout = dout.contiguous()
        M, H = y.shape
        dy = torch.empty_like(y)
        dmod = torch.zeros_like(mod)      # this branch's three chunks; autograd sums the two branches
        hip.check(hip.lib().e3d_adaln_gate_bwd(_p(dout), _p(y), _p(mod), ctx.branch, ctx.rpc, _p(dy), _p(dmod), M, H,
                                               _stream()), "e3d_adaln_gate_bwd")
        return dout, dy, dmod, None, None


class _EmbedLayerNorm(torch.autograd.Function):
    """LN(x W^T + b) * gamma + beta (+ post_add): x is data (no gradient)."""

    @staticmethod
    def forward(ctx, x, weight, bias, gamma, beta, eps, post_add, rows_per_add):
        out, z = ops.embed_layernorm(x, weight, bias, gamma, beta, eps, post_add, rows_per_add, want_z=True)
        ctx.save_for_backward(x, z, gamma)
        ctx.eps, ctx.rpa, ctx.has_add = eps, rows_per_add, post_add is not None
        return out

    @staticmethod
    def backward(ctx, dout):
        x, z, gamma = ctx.saved_tensors
        dout = dout.contiguous()
        dz, dg, db_ln = layernorm_bwd(dout, z, gamma, ctx.eps)
        dW, db = small_k_wgrad(dz, x)
        dadd = group_sum(dout, ctx.rpa) if (ctx.has_add and ctx.needs_input_grad[6]) else None
        return None, dW, db, dg, db_ln, None, dadd, None


class _HeadLinear(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias):
        ctx.save_for_backward(x, weight)
        return ops.head_linear(x, weight, bias)

    @staticmethod
    def backward(ctx, dout):
        x, weight = ctx.saved_tensors
        dout = dout.contiguous()
        M, H = x.shape
        dx = torch.empty_like(x)
        hip.check(hip.lib().e3d_head_linear_bwd_dx(_p(dout), _p(weight), _p(dx), M, H, weight.shape[0], _stream()),
                  "e3d_head_linear_bwd_dx")
        dW, _ = small_k_wgrad(x, dout, transpose_out=True, want_bias=False)   # [n_out, H]
        return dx, dW, colsum(dout)


# nn.Dropout after BertSelfOutput.dense / BertOutput.dense inside the residual-LayerNorm kernels (0: its own launches, A/B runs)
FUSE_HIDDEN_DROPOUT = os.environ.get("E3D_FUSE_HIDDEN_DROPOUT", "1") == "1"


# ----------------------------------------------------------------------------- functional front-end
def _needs_grad(*tensors):
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)


class functional:
    @staticmethod
    def linear(x, weight, bias=None, act=ops.ACT_NONE, absmax=None):
        """``absmax`` (ops.absmax_slot): raised to the largest |output| (act = none); the result then carries it as
        ``_e3d_absmax`` for ``attention`` (q / k projections)."""
        if _needs_grad(x, weight, bias):
            out = _Linear.apply(x, weight, bias, act, absmax)
        else:
            out = ops.gemm(x, weight, bias, act, absmax=absmax)
        if absmax is not None:
            out._e3d_absmax = absmax
        return out

    @staticmethod
    def attention(q_src, kv_src, B, nh, Lq, Lk, key_mask=None, dist_emb=None, max_pos=0, drop_p=0.0):
        """q_src = packed qkv [B*L,3H] (kv_src None, self-attention) or q [B*Lq,H] with packed kv [B*Lk,2H].
        ``drop_p`` > 0 (training): dropout on the attention probabilities."""
        # element bounds left by the projections (functional.linear(..., absmax=)): they let the kernels skip all-padding
        # key tiles when that is provably exact; absent (direct callers) = full sweep
        q_abs = getattr(q_src, "_e3d_absmax", None)
        k_abs = q_abs if kv_src is None else getattr(kv_src, "_e3d_absmax", None)
        bounds = (q_abs, k_abs) if q_abs is not None and k_abs is not None else None
        if _needs_grad(q_src, kv_src, dist_emb):
            return _Attention.apply(q_src, kv_src, dist_emb, key_mask, B, nh, Lq, Lk, max_pos, drop_p, bounds)
        q, k, v = _Attention._views(q_src, kv_src, nh * 64)
        drop = (float(drop_p), ops.next_dropout_seed()) if drop_p > 0 else None
        return ops.attention(q, k, v, B, nh, Lq, Lk, key_mask=key_mask, dist_emb=dist_emb, max_pos=max_pos, drop=drop,
                             bounds=bounds)

    @staticmethod
    def dropout(x, p, training=True):
        """nn.Dropout(p) of the reference: identity unless ``training`` and p > 0."""
        if not training or not p:
            return x
        if _needs_grad(x):
            return _Dropout.apply(x, p)
        return ops.dropout(x.contiguous(), p, ops.next_dropout_seed())

    @staticmethod
    def residual_layernorm(x, residual, gamma, beta, eps, p_drop=0.0):
        """LayerNorm(dropout(x, p_drop) + residual)."""
        if _needs_grad(x, residual, gamma, beta):
            return _ResidualLayerNorm.apply(x, residual, gamma, beta, eps, p_drop)
        drop = (float(p_drop), ops.next_dropout_seed()) if p_drop and p_drop > 0 else None
        return ops.residual_layernorm(x, residual, gamma, beta, eps, drop=drop)

    @staticmethod
    def linear_residual_layernorm(x, weight, bias, residual, gamma, beta, eps, p_drop=0.0):
        """BertSelfOutput / BertOutput: LayerNorm(dropout(x W^T + b) + residual).  Inference at small M takes the
        fused skinny-GEMM finish (ops.linear_residual_layernorm); training and every other shape the three ops."""
        if not p_drop and not _needs_grad(x, weight, bias, residual, gamma, beta):
            return ops.linear_residual_layernorm(x, weight, bias, residual, gamma, beta, eps)
        fn = functional
        if FUSE_HIDDEN_DROPOUT:
            return fn.residual_layernorm(fn.linear(x, weight, bias), residual, gamma, beta, eps, p_drop)
        return fn.residual_layernorm(fn.dropout(fn.linear(x, weight, bias), p_drop), residual, gamma, beta, eps)

    @staticmethod
    def adaln_gate(x, y, mod, branch, rows_per_cond):
        if _needs_grad(x, y, mod):
            return _AdaLNGate.apply(x, y, mod, branch, rows_per_cond)
        return ops.adaln_gate(x, y, mod, branch, rows_per_cond)

    @staticmethod
    def embed_layernorm(x, weight, bias, gamma, beta, eps, post_add=None, rows_per_add=1):
        if _needs_grad(weight, bias, gamma, beta, post_add):
            return _EmbedLayerNorm.apply(x, weight, bias, gamma, beta, eps, post_add, rows_per_add)
        return ops.embed_layernorm(x, weight, bias, gamma, beta, eps, post_add, rows_per_add)

    @staticmethod
    def head_linear(x, weight, bias):
        if _needs_grad(x, weight, bias):
            return _HeadLinear.apply(x, weight, bias)
        return ops.head_linear(x, weight, bias)
