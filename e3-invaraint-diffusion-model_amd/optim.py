"""``ClipAdamW``: torch.optim.AdamW (the reference's optimizer: structure_model/model.py:361-366, sequence_model/model.py
``configure_optimizers``) whose ``step`` runs the global-norm gradient clip Lightning applies around it
(``gradient_clip_val=1.0``, structure_model/train_model.py:99-110) and the update itself as THREE HIP launches over all
parameters (csrc/optim.hip: squared-norm partials, their ordered sum, the update) instead of torch's ~100 multi-tensor
launches: the gradients are read twice and never rewritten, the clip coefficient stays on the device.

State layout and ``state_dict`` are torch.optim.AdamW's (``step`` / ``exp_avg`` / ``exp_avg_sq`` per parameter, the same
param_group keys), so optimizer checkpoints move between the two; LR schedulers and optimizer hooks see an ordinary
``Optimizer``.  Anything the kernels do not cover (amsgrad, maximize, non-fp32 or non-contiguous tensors, CPU
parameters, a closure-free ``step`` on a box without the HIP library) falls back to ``torch.optim.AdamW.step`` after a
``clip_grad_norm_``.
"""
import torch

from . import hip


DYN_STRIDE = 8       # floats per parameter range in the device-side scalar table


class ClipAdamW(torch.optim.AdamW):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, lr=lr, betas=betas, eps=eps, weight_decay=weight_decay)
        self._e3d_clip = None        # max_norm of the NEXT step (step_clipped sets it; None: no clip)
        self.device_scalars = False  # learning rate and step counts read from device memory (graph replay): use_device_scalars
        self._e3d_tab = None         # device tables of the current parameter partition
        self.last_norm = None        # total gradient norm of the last clipped step (0-dim device tensor)

    # ------------------------------------------------------------------------------------------------ public
    def step_clipped(self, max_norm, closure=None):
        """clip_grad_norm_(all parameters of this optimizer, max_norm) + step(); returns the total norm (device tensor)."""
        self._e3d_clip = float(max_norm) if max_norm else None
        try:
            self.step(closure)
        finally:
            self._e3d_clip = None
        return self.last_norm

    def use_device_scalars(self, on=True):
        """Keep every scalar of the update -- learning rate, step count, betas, eps, weight decay -- of every parameter range
        in device memory (``e3d_adamw_step_dev``) so that a captured HIP graph of the step can be replayed: ``sync_lr()``
        before a replay pushes whatever a scheduler changed (OneCycleLR moves beta1 with the learning rate),
        ``note_replayed_step()`` after it does the host-side bookkeeping ``step()`` would have done."""
        if bool(on) != self.device_scalars:
            self.device_scalars = bool(on)
            self._e3d_tab = None

    def sync_lr(self):
        tab = self._e3d_tab
        if tab is None or "dyn" not in tab:
            return
        for r, (gi, _, _) in enumerate(tab["dyn_ranges"]):
            hyper = self._hyper(self.param_groups[gi])
            if tab["dyn_hyper"][r] != hyper:
                if tab["dyn_hyper"][r][0] != hyper[0]:
                    tab["dyn"][r, 0:1].fill_(hyper[0])
                if tab["dyn_hyper"][r][1:] != hyper[1:]:
                    tab["dyn"][r, 2:6].copy_(torch.tensor(hyper[1:], dtype=torch.float32), non_blocking=False)
                tab["dyn_hyper"][r] = hyper

    @staticmethod
    def _hyper(group):
        """(lr, beta1, beta2, eps, weight_decay) of a parameter group as the kernels see them (fp32 values)."""
        b1, b2 = group["betas"]
        return tuple(float(torch.tensor(float(x), dtype=torch.float32)) for x in (group["lr"], b1, b2, group["eps"], group["weight_decay"]))

    def note_replayed_step(self, delta=1):
        """Host-side bookkeeping of one replayed step (``delta=-1``: undo that of a capture pass, which runs ``step()`` on
        the host without executing anything on the device)."""
        tab = self._e3d_tab
        if tab is None:
            return
        tab["ranges"] = [[(c0, c1, step + delta) for (c0, c1, step) in rs] for rs in tab["ranges"]]
        torch._foreach_add_(tab["step_tensors"], delta)
        tab["gptr_host"] = None      # the graph re-copies ITS gradient pointers: an eager step must upload its own again
        self._opt_called = True      # what LR schedulers look at to order scheduler.step() after optimizer.step()

    def new_capture_staging(self):
        """Call before capturing ANOTHER graph of the step: a captured ``step()`` uploads the addresses of its gradient
        tensors from a pinned staging buffer, and every replay reads that buffer again -- so each graph needs its own (the
        earlier ones stay alive here; a few KB each)."""
        tab = self._e3d_tab
        if tab is None:
            return
        tab.setdefault("gptr_pin_kept", []).append(tab["gptr_pin"])
        tab["gptr_pin"] = torch.zeros_like(tab["gptr_pin"]).pin_memory()
        tab["gptr_host"] = None

    def load_state_dict(self, state_dict):
        super().load_state_dict(state_dict)
        self._e3d_tab = None

    def add_param_group(self, group):
        super().add_param_group(group)
        self._e3d_tab = None

    # ------------------------------------------------------------------------------------------------ step
    def _covered(self, group, p):
        g = p.grad
        return (p.is_cuda and p.dtype == torch.float32 and g.dtype == torch.float32 and p.is_contiguous() and g.is_contiguous()
                and not g.is_sparse and not group.get("amsgrad") and not group.get("maximize") and not group.get("capturable"))

    def _fallback(self, closure):
        if self._e3d_clip is not None:
            params = [p for g in self.param_groups for p in g["params"] if p.grad is not None]
            self.last_norm = torch.nn.utils.clip_grad_norm_(params, self._e3d_clip) if params else None
        self._e3d_tab = None      # torch advances state["step"] itself: the cached ranges / device counts would be stale
        # (the undecorated implementation: this call already runs inside the hooked ClipAdamW.step, so going through a
        #  hooked torch.optim.AdamW.step -- patched as soon as a plain AdamW exists in the process -- would fire every
        #  optimizer pre / post hook twice)
        plain = getattr(torch.optim.AdamW.step, "__wrapped__", torch.optim.AdamW.step)
        return plain(self, closure)

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        parts = []       # (group index, [params with a gradient])
        for gi, group in enumerate(self.param_groups):
            ps = [p for p in group["params"] if p.grad is not None]
            if ps:
                if not all(self._covered(group, p) for p in ps):
                    self._fallback(None)
                    return loss
                parts.append((gi, ps))
        if not parts:
            self.last_norm = None
            return loss
        if len({p.device for _, ps in parts for p in ps}) != 1:      # one pointer table = one device
            self._fallback(None)
            return loss
        tab = self._tables(parts)
        lib = hip.lib()
        dev = tab["device"]
        stream = torch.cuda.current_stream(dev).cuda_stream
        # gradient pointers: new tensors every step (zero_grad(set_to_none=True)), normally at last step's addresses
        capturing = torch.cuda.is_current_stream_capturing()
        gptr = [p.grad.data_ptr() for _, ps in parts for p in ps]
        if gptr != tab["gptr_host"]:
            tab["gptr_host"] = gptr
            if capturing:      # a persistent staging buffer (no host allocation while a stream is capturing; replays re-copy it)
                tab["gptr_pin"].copy_(torch.tensor(gptr, dtype=torch.int64))
                tab["gptr"].copy_(tab["gptr_pin"], non_blocking=True)
            else:              # eager: a fresh pinned tensor per change (an earlier asynchronous copy may still be reading the last one)
                tab["gptr"].copy_(torch.tensor(gptr, dtype=torch.int64).pin_memory(), non_blocking=True)
        if self.device_scalars and not capturing:
            self.sync_lr()
        nc = None
        if self._e3d_clip is not None:
            nc = tab["norm_and_clip"]
            hip.check(lib.e3d_grad_global_norm(tab["gptr"].data_ptr(), tab["numel"].data_ptr(), tab["chunk_tensor"].data_ptr(),
                                               tab["chunk_first"].data_ptr(), tab["n_chunks"], self._e3d_clip,
                                               tab["partial"].data_ptr(), nc.data_ptr(), stream), "e3d_grad_global_norm")
            self.last_norm = nc[0]
        r = -1
        for k, (gi, ps) in enumerate(parts):
            group = self.param_groups[gi]
            for (c0, c1, step) in tab["ranges"][k]:
                b1, b2 = group["betas"]
                lr = group["lr"]
                r += 1
                if self.device_scalars:
                    hip.check(lib.e3d_adamw_step_dev(
                        tab["pptr"].data_ptr(), tab["gptr"].data_ptr(), tab["mptr"].data_ptr(), tab["vptr"].data_ptr(),
                        tab["numel"].data_ptr(), tab["chunk_tensor"].data_ptr() + 4 * c0, tab["chunk_first"].data_ptr() + 8 * c0,
                        c1 - c0, nc.data_ptr() if nc is not None else None, tab["dyn"].data_ptr() + 4 * DYN_STRIDE * r, stream),
                        "e3d_adamw_step_dev")
                    continue
                hip.check(lib.e3d_adamw_step(
                    tab["pptr"].data_ptr(), tab["gptr"].data_ptr(), tab["mptr"].data_ptr(), tab["vptr"].data_ptr(),
                    tab["numel"].data_ptr(), tab["chunk_tensor"].data_ptr() + 4 * c0, tab["chunk_first"].data_ptr() + 8 * c0,
                    c1 - c0, nc.data_ptr() if nc is not None else None, float(lr), float(b1), float(b2), float(group["eps"]),
                    float(group["weight_decay"]), step + 1, stream), "e3d_adamw_step")
            tab["ranges"][k] = [(c0, c1, step + 1) for (c0, c1, step) in tab["ranges"][k]]
        if self.device_scalars:
            tab["dyn"][:, 1] += 1.0                      # the device-side step counts (captured with the step)
        torch._foreach_add_(tab["step_tensors"], 1)
        return loss

    # ------------------------------------------------------------------------------------------------ tables
    def _tables(self, parts):
        sig = tuple((gi, tuple(id(p) for p in ps)) for gi, ps in parts)
        tab = self._e3d_tab
        if tab is not None and tab["sig"] == sig and all(p.data_ptr() == q for p, q in zip(tab["params"], tab["pptr_host"])):
            return tab
        chunk = hip.lib().e3d_optim_chunk_elems()
        dev = parts[0][1][0].device
        params, steps = [], []
        for gi, ps in parts:
            for p in ps:
                st = self.state[p]
                if len(st) == 0:                       # torch.optim.AdamW's lazy state initialisation
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                if st["step"].is_cuda:                 # a state saved by torch's fused implementation: keep the counter on the host
                    st["step"] = st["step"].detach().cpu()
                if not (st["exp_avg"].is_contiguous() and st["exp_avg_sq"].is_contiguous() and st["exp_avg"].dtype == torch.float32):
                    st["exp_avg"], st["exp_avg_sq"] = st["exp_avg"].float().contiguous(), st["exp_avg_sq"].float().contiguous()
                params.append(p)
                steps.append(int(st["step"]))
        # within a group, parameters are ordered by their step count so that every count is one contiguous chunk range
        # (normally a single count: a parameter that received its first gradient later than the others gets its own range)
        order, k0 = [], 0
        for gi, ps in parts:
            idx = sorted(range(k0, k0 + len(ps)), key=lambda i: steps[i])
            order.extend(idx)
            k0 += len(ps)
        ranges, chunk_tensor, chunk_first = [], [], []
        k0 = 0
        for gi, ps in parts:
            rs, cur = [], None
            for i in order[k0:k0 + len(ps)]:
                c0 = len(chunk_tensor)
                n = params[i].numel()
                for f in range(0, n, chunk):
                    chunk_tensor.append(i)
                    chunk_first.append(f)
                if cur is not None and cur[2] == steps[i]:
                    cur[1] = len(chunk_tensor)
                else:
                    cur = [c0, len(chunk_tensor), steps[i]]
                    rs.append(cur)
            ranges.append([tuple(r) for r in rs])
            k0 += len(ps)

        def dev_i64(vals):
            return torch.tensor(vals, dtype=torch.int64).to(dev)

        n_chunks = len(chunk_tensor)
        tab = {
            "sig": sig, "device": dev, "params": params, "pptr_host": [p.data_ptr() for p in params],
            "pptr": dev_i64([p.data_ptr() for p in params]),
            "mptr": dev_i64([self.state[p]["exp_avg"].data_ptr() for p in params]),
            "vptr": dev_i64([self.state[p]["exp_avg_sq"].data_ptr() for p in params]),
            "numel": dev_i64([p.numel() for p in params]),
            "chunk_tensor": torch.tensor(chunk_tensor, dtype=torch.int32).to(dev), "chunk_first": dev_i64(chunk_first),
            "n_chunks": n_chunks, "ranges": ranges, "gptr_host": None,
            "gptr": torch.zeros(len(params), dtype=torch.int64, device=dev),
            "gptr_pin": torch.zeros(len(params), dtype=torch.int64).pin_memory(),
            "partial": torch.empty(n_chunks, dtype=torch.float32, device=dev),
            "norm_and_clip": torch.zeros(2, dtype=torch.float32, device=dev),
            "step_tensors": [self.state[p]["step"] for p in params],
            # (the state tensors the device tables point into: kept alive with the tables)
            "keep": [(self.state[p]["exp_avg"], self.state[p]["exp_avg_sq"]) for p in params],
        }
        if self.device_scalars:
            flat = [(gi, c0, st) for (gi, _), rs in zip(parts, ranges) for (c0, c1, st) in rs]
            tab["dyn_ranges"] = flat
            tab["dyn_hyper"] = [self._hyper(self.param_groups[gi]) for gi, _, _ in flat]
            # per range: lr, steps taken, beta1, beta2, eps, weight_decay, (2 pad) -- e3d_adamw_step_dev's ``hyper`` block
            tab["dyn"] = torch.tensor([[h[0], float(st), h[1], h[2], h[3], h[4], 0.0, 0.0] for h, (_, _, st) in zip(tab["dyn_hyper"], flat)],
                                      dtype=torch.float32).to(dev)
        self._e3d_tab = tab
        return tab
