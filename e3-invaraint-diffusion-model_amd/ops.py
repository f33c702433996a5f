"""Tensor-level wrappers over the C-ABI: allocate outputs with torch (device memory + stream
plumbing only) and enqueue the HIP kernels on torch's current stream."""
import math
import os
import warnings

import torch

from . import hip

ACT_NONE, ACT_GELU, ACT_SILU = 0, 1, 2


def _stream():
    return torch.cuda.current_stream().cuda_stream


# Per-launch timing hook for bench.py: when TRACE is a list, timed launches append
# (kernel name, start event, end event, meta) -- HIP events on the launch stream.
TRACE = None


# roctx ranges around the hot-path kernels (K1 rel-key attention, K2 cross attention, K3 their
# backward, K4 adaLN gate, K5 LayerNorm / GEMM epilogues, K6 DDPM update + wrap, K7 discrete posterior; plus the
# GEMMs).  E3D_ROCTX=1 (tools/profile_round.sh sets it) loads libroctx64 and brackets every launch of those ops;
# `rocprofv3 --marker-trace` then shows the ranges next to the kernel trace.  Off by default: two ctypes calls
# per launch are measurable in the single-pocket loop.
_ROCTX = None


def _roctx():
    global _ROCTX
    if _ROCTX is None:
        _ROCTX = False
        if os.environ.get("E3D_ROCTX") == "1":
            import ctypes
            for name in ("libroctx64.so", "libroctx64.so.4", "/opt/rocm/lib/libroctx64.so"):
                try:
                    lib = ctypes.CDLL(name)
                    lib.roctxRangePushA.argtypes = [ctypes.c_char_p]
                    _ROCTX = lib
                    break
                except OSError:
                    continue
    return _ROCTX


ROCTX_NAMES = {"attn_relkey": b"K1 relkey_attention_fwd", "attn_cross": b"K2 cross_attention_fwd",
               "attn_bwd": b"K3 attention_bwd", "adaln_gate": b"K4 adaln_gate",
               "residual_layernorm": b"K5 residual_layernorm", "embed_layernorm": b"K5 embed_layernorm",
               "gemm_layernorm": b"K5 gemm_residual_layernorm",
               "gemm": b"K5 gemm_bias_act", "ddpm_step_wrap": b"K6 ddpm_step_wrap",
               "discrete_posterior": b"K7 discrete_posterior_sample", "discrete_q_sample": b"K7 discrete_q_sample"}


class _timed:
    """Brackets one launch: HIP events on the launch stream when bench.py collects a TRACE, a roctx range when
    E3D_ROCTX=1."""

    def __init__(self, name, meta=None):
        self.on = TRACE is not None
        self.rx = _roctx()
        self.name = name
        if self.on:
            self.rec = (name, torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True), meta)

    def __enter__(self):
        if self.rx:
            self.rx.roctxRangePushA(ROCTX_NAMES.get(self.name, self.name.encode()))
        if self.on:
            self.rec[1].record()

    def __exit__(self, *exc):
        if self.on:
            self.rec[2].record()
            TRACE.append(self.rec)
        if self.rx:
            self.rx.roctxRangePop()


def _chk(t, name, dtype=torch.float32):
    if t is None:
        return
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (this package has no CPU path), got {t.device}")
    if t.dtype != dtype:
        raise TypeError(f"{name}: expected {dtype}, got {t.dtype}")


def _p(t):
    return None if t is None else t.data_ptr()


# ---------------------------------------------------------------------------------------------- derived-weight caches
# Tensors derived from parameters (packed QKV weights, W^T for the input-gradient GEMM, the distance-table planes) are
# cached on their source and keyed on (tensor._version, data_ptr, PARAM_GENERATION).  ``_version`` alone is NOT enough:
# torch's fused optimizer kernels (torch.optim.AdamW(fused=True), used by training.adamw) update parameters without
# bumping it, so a cache keyed only on the version would keep serving the weights of the first step.  Every
# optimizer step therefore bumps PARAM_GENERATION through a global post-step hook (registered at import); code that
# writes parameter storage by other means behind autograd's back calls ``invalidate_weight_caches()`` itself.
PARAM_GENERATION = 0


def invalidate_weight_caches(*_args, **_kwargs):
    global PARAM_GENERATION
    PARAM_GENERATION += 1


def weight_key(*tensors):
    return (PARAM_GENERATION,) + tuple((t.data_ptr(), t._version) for t in tensors)


try:
    from torch.optim.optimizer import register_optimizer_step_post_hook as _reg_post_hook
    _reg_post_hook(invalidate_weight_caches)
except ImportError:      # very old torch: training.fit() bumps the generation after every step instead
    pass


# ---------------------------------------------------------------------------------------------- |out| bounds
# The attention kernels may skip key tiles that are padding in every position only when that is exact: the reference
# masks additively with -10000 (structure_model/model.py:226-231), so a padded key stays out of the softmax only while
# the scores of a row spread over less than ~9900 (include/e3d_hip.h, e3d_attn_skip_padded_tiles).  The proof comes from
# the GEMMs that produce Q and K: launched with ``absmax=<slot>`` they raise a device scalar to the largest |output|
# (atomic max in the epilogue), the attention call reads the scalars on the device.  Slots are one float each in a
# per-device pool, handed out once per owner (a BertSelfAttention module) and only ever raised by the kernels:
# ``reset_absmax()`` (one fill launch) zeroes them all -- models call it once per batch / sampling chain, so a slot holds
# the running maximum of that chain.  A stale (too large) value is still a valid bound: it can only cost the skip.
_ABSMAX_POOLS = {}      # device index -> [current pool tensor, slots handed out of it, earlier (full) pool tensors]
ABSMAX_POOL_SLOTS = 4096


def absmax_slot(owner, name, device):
    """The device scalar (1-element view) that bounds |outputs| of ``owner``'s projection ``name``."""
    slots = owner.__dict__.setdefault("_e3d_absmax", {})
    key = (name, device.index)
    t = slots.get(key)
    if t is None:
        ent = _ABSMAX_POOLS.get(device.index)
        if ent is None:
            ent = _ABSMAX_POOLS[device.index] = [torch.zeros(ABSMAX_POOL_SLOTS, device=device, dtype=torch.float32), 0, []]
        if ent[1] >= ABSMAX_POOL_SLOTS:     # slots are never handed back (a process that builds model after model): grow
            ent[2].append(ent[0])
            ent[0], ent[1] = torch.zeros(ABSMAX_POOL_SLOTS, device=device, dtype=torch.float32), 0
        t = slots[key] = ent[0][ent[1]:ent[1] + 1]
        ent[1] += 1
    return t


def reset_absmax(device=None):
    """Zero every slot of the device's pool (one launch): the start of a batch or of a sampling chain."""
    idx = torch.cuda.current_device() if device is None else torch.device(device).index
    ent = _ABSMAX_POOLS.get(idx)
    if ent is not None:
        ent[0].zero_()
        for old_pool in ent[2]:
            old_pool.zero_()


def absmax(x, target=None):
    """max(target, max |x|) as a 1-element device tensor (e3d_absmax_f32); NaN / inf in x give NaN / inf."""
    _chk(x, "absmax.x")
    assert x.is_contiguous()
    if target is None:
        target = torch.zeros(1, device=x.device, dtype=torch.float32)
    hip.check(hip.lib().e3d_absmax_f32(_p(x), x.numel(), _p(target), _stream()), "e3d_absmax_f32")
    return target


# GEMM arithmetic: "f32" = exact fp32 MFMA; "bf16x3" / "bf16x6" = fp32 operands split into 2 / 3
# bf16 terms on the bf16 matrix cores with fp32 accumulation (gemm_split.hip).  bf16x6 is fp32-grade
# (4e-7 end-to-end vs 1.8e-6 for fp32 itself), bf16x3 ~2.6e-5 end-to-end (tolerance 1e-4).
# "f16x3" = 2 fp16 terms (11 + 11 bits), 3 cross products: fp32-grade accuracy at the cost of bf16x3 for operands
# inside the fp16 range (|x| < 65504 -- larger values give inf/NaN, loudly; include/e3d_hip.h E3D_TERMS_F16X3);
# forward GEMMs and the cooperative attention kernel, every other path of that mode runs bf16x6 / fp32 kernels.
# "bf16" = plain bf16 products (operands rounded to bf16, ONE MFMA per product, fp32 accumulate: ~2^-8 per product) -- the
# precision the reference's TRAIN scripts select (torch.set_float32_matmul_precision("medium"),
# structure_model/train_model.py:120).  GEMMs only (the attention kernels then run bf16x3), opt-in through
# E3D_TRAIN_ARITHMETIC=bf16 / ops.arithmetic("bf16"); never an inference default: it is 1e-2-grade, not 1e-4.
GEMM_MODES = {"f32": 0, "bf16x3": 3, "bf16x6": 6, "f16x3": 19, "bf16": 1}
GEMM_MODE = os.environ.get("E3D_GEMM_MODE", "f16x3")   # inference default: fp32 grade at bf16x3 speed
if GEMM_MODE not in GEMM_MODES:
    raise ValueError(f"E3D_GEMM_MODE must be one of {sorted(GEMM_MODES)}, got {GEMM_MODE!r}")


def set_gemm_mode(mode):
    global GEMM_MODE
    if mode not in GEMM_MODES:
        raise ValueError(f"gemm mode must be one of {sorted(GEMM_MODES)}, got {mode!r}")
    prev, GEMM_MODE = GEMM_MODE, mode
    return prev


# f16x3 weights: two fp16 terms carry 22 bits only while BOTH are normal numbers -- an element below 2^-3 = 0.125
# already has its low term in the fp16 subnormals (absolute floor 2^-25), and nn.Linear weights are ~0.02-0.05.  Each
# weight is therefore multiplied by a power of two (exact) that puts its largest |element| in [2^11, 2^12) -- elements
# down to 2^-15 of the largest keep 22 bits, the largest is 16x below the fp16 maximum, a weight of ANY magnitude is safe
# -- and the kernels multiply the accumulator by the inverse power (exact) before the bias (``out_scale`` of the *_ex
# entry points).  Scaled copies are cached on the weight tensor (keyed like every derived-weight cache); building one
# reads max |w| back to the host, once per weight and version.  Activations are not scaled: the path's GEMM inputs are
# LayerNorm / GELU / softmax outputs, and an activation beyond 65504 gives inf / NaN -- loudly.
F16_WEIGHT_EXP = 12
_f16_warned = False


def f16_weight(weight):
    """(weight * 2^k, 2^-k): the f16x3 operand of ``weight`` and the accumulator scale that undoes it."""
    global _f16_warned
    ent = getattr(weight, "_e3d_f16w", None)
    key = weight_key(weight)
    if ent is not None and ent[0] == key:
        return ent[1], ent[2]
    if torch.cuda.is_current_stream_capturing():
        # no host read-back inside a capture, and no silent fallback either: a graph captured with the raw
        # weight would run f16x3 on un-pre-scaled weights at every replay.  The capturers (GraphedReverseStep,
        # GraphedDenoiseStep) run their body eagerly first, which fills this cache; their except-paths fall back to eager
        # launches if this is ever raised.
        raise RuntimeError("f16x3: the pre-scaled copy of a weight is missing inside a stream capture -- run the step eagerly once "
                           "before capturing it (building the copy reads max |w| back to the host)")
    with torch.no_grad():
        amax = float(weight.detach().abs().max())
        if not math.isfinite(amax) or amax == 0.0:
            if not math.isfinite(amax) and not _f16_warned:
                _f16_warned = True
                warnings.warn("f16x3: a weight holds inf / NaN -- used as it is (the outputs will say so)")
            scaled, inv = weight.detach(), 1.0
        else:
            k = max(-120, min(120, F16_WEIGHT_EXP - math.frexp(amax)[1]))
            scaled, inv = (weight.detach() * (2.0 ** k)).contiguous(), 2.0 ** -k
    try:
        weight._e3d_f16w = (key, scaled, inv)
    except AttributeError:
        pass
    return scaled, inv


# Small launches (one to ~16 pockets per step) go to the "skinny" kernels (csrc/gemm_skinny.hip: K cut across
# workgroups so the whole chip streams the weight, partial tiles in a workspace, a second launch finishes the rows).  The
# workspace must not be shared by launches that may run concurrently: one per (device, stream), grown on demand and kept.
# Where they win (tools/lab/skinny_ab.py, graph-timed, K = 768): up to ~768 output tiles of 32x32 -- M = 64: 7.4 us
# against 21-28 us tiled, M = 256: 11 against 23.5, M = 512 x N = 768: 15.5 against 23.8, M = 1024 x N = 768: 21.5
# against 24.2; beyond that the tiled kernels (a 24-step k chain, ~24 us) are faster (M = 512 x N = 2304: 33 vs 24.5).
# E3D_GEMM_SKINNY=0 disables the path (A/B timing).
SKINNY_MAX_M = 1024 if os.environ.get("E3D_GEMM_SKINNY", "1") == "1" else 0
# plain GEMMs (no fused LayerNorm finish) above this many rows take the general kernel's 128x64 four-wave form instead (round 4:
# 18.8 / 23.0 / 19.4 us against 21.6 / 27.1 / 27.5 skinny at M = 1024 for (N, K) = (768, 768) / (768, 1024) / (1024, 768);
# at M = 512 the skinny kernels win two of the three: profiles/r04_gemm_mid_m_ab.log)
SKINNY_GEMM_MAX_M = int(os.environ.get("E3D_GEMM_SKINNY_MAX_M", "512"))
SKINNY_MAX_TILES = 768
_SKINNY_WS = {}
_SKINNY_RETIRED = []


def _skinny_workspace(device, M, N, K):
    nbytes = hip.lib().e3d_gemm_skinny_workspace_bytes(M, N, K)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    ws = _SKINNY_WS.get(key)
    if ws is None or ws.numel() < nbytes:
        # (under a graph capture this comes from the graph's private pool and is kept alive here: replay-safe; a workspace
        #  that is outgrown is parked, not freed -- an earlier capture on this stream may have baked its address)
        if ws is not None:
            _SKINNY_RETIRED.append(ws)
        ws = _SKINNY_WS[key] = torch.empty(max(nbytes, 8 << 20), dtype=torch.uint8, device=device)
    return ws


def _skinny_ok(terms, M, N, K, a):
    return (terms in (3, 19) and 0 < M <= SKINNY_MAX_M and N % 32 == 0 and K % 16 == 0 and a.stride(0) % 4 == 0
            and -(-M // 32) * (N // 32) <= SKINNY_MAX_TILES)


def gemm(a, weight, bias=None, act=ACT_NONE, out=None, mode=None, absmax=None, prescale=True):
    """out[M,N] = act(a[M,K] @ weight[N,K]^T + bias).  ``a`` may be a row-strided 2-D view.  ``absmax`` (1-element
    device tensor, act = none): raised to the largest |out| by the kernel's epilogue (split arithmetics; the exact-f32
    kernels have no use for it -- their attention twin never skips key tiles).  ``prescale`` (f16x3 only): run the
    product on the cached power-of-two-scaled copy of the weight (``f16_weight``); False for weights that change every
    step (the training forward)."""
    _chk(a, "gemm.a"); _chk(weight, "gemm.weight"); _chk(bias, "gemm.bias")
    assert a.dim() == 2 and a.stride(1) == 1 and weight.is_contiguous()
    M, K = a.shape
    N = weight.shape[0]
    assert weight.shape[1] == K, (a.shape, weight.shape)
    if out is None:
        out = torch.empty((M, N), device=a.device, dtype=torch.float32)
    assert out.dim() == 2 and out.stride(1) == 1 and out.shape == (M, N)
    terms = GEMM_MODES[GEMM_MODE if mode is None else mode]
    if absmax is not None:
        _chk(absmax, "gemm.absmax")
        assert act == ACT_NONE and absmax.numel() == 1
    scale = 1.0
    if terms == 19 and prescale:
        weight, scale = f16_weight(weight)
    with _timed("gemm", (M, N, K, act)):
        if _skinny_ok(terms, M, N, K, a) and out.stride(0) % 4 == 0 and M <= min(SKINNY_MAX_M, SKINNY_GEMM_MAX_M):
            ws = _skinny_workspace(a.device, M, N, K)
            hip.check(hip.lib().e3d_gemm_skinny_f32_split_ex(_p(a), a.stride(0), _p(weight), _p(bias), _p(out), out.stride(0),
                                                             M, N, K, act, terms, _p(ws), ws.numel(), _p(absmax), scale, _stream()),
                      "e3d_gemm_skinny_f32_split_ex")
        elif terms == 0:
            hip.check(hip.lib().e3d_gemm_bias_act_f32(_p(a), a.stride(0), _p(weight), _p(bias), _p(out),
                                                      out.stride(0), M, N, K, act, _stream()), "e3d_gemm_bias_act_f32")
        else:
            hip.check(hip.lib().e3d_gemm_bias_act_f32_split_ex(
                _p(a), a.stride(0), _p(weight), _p(bias), _p(out), out.stride(0), M, N, K, act, terms, _p(absmax), scale,
                _stream()), "e3d_gemm_bias_act_f32_split_ex")
    return out


class arithmetic:
    """``with ops.arithmetic("bf16x6"):`` -- GEMM and attention arithmetic for the enclosed calls (entry points use it
    to mirror the precision the reference runs that script at; E3D_GEMM_MODE / E3D_ATTN_MODE in the environment
    win over an entry point's default)."""

    def __init__(self, mode, respect_env=True):
        self.gemm = os.environ.get("E3D_GEMM_MODE", mode) if respect_env else mode
        self.attn = os.environ.get("E3D_ATTN_MODE", self.gemm) if respect_env else mode

    def __enter__(self):
        self.prev = (set_gemm_mode(self.gemm), set_attn_mode(self.attn))
        return self

    def __exit__(self, *exc):
        set_gemm_mode(self.prev[0])
        set_attn_mode(self.prev[1])


ATTN_MODE = os.environ.get("E3D_ATTN_MODE", GEMM_MODE)   # same choices and meaning as GEMM_MODE
if ATTN_MODE == "bf16":
    ATTN_MODE = "bf16x3"
if ATTN_MODE not in GEMM_MODES:
    raise ValueError(f"E3D_ATTN_MODE must be one of {sorted(GEMM_MODES)}, got {ATTN_MODE!r}")


def set_attn_mode(mode):
    global ATTN_MODE
    if mode not in GEMM_MODES:
        raise ValueError(f"attention mode must be one of {sorted(GEMM_MODES)}, got {mode!r}")
    if mode == "bf16":      # the single-product form exists for the GEMMs only
        mode = "bf16x3"
    prev, ATTN_MODE = ATTN_MODE, mode
    return prev


_DROPOUT_EPOCH = {}


def dropout_epoch(device):
    """The device word every dropout launch on ``device`` adds to its seed INSIDE the kernel (created and registered with
    the library on first use: include/e3d_hip.h, e3d_dropout_set_epoch_ptr).  A training step replayed from a HIP graph
    carries the seeds of its capture as baked arguments; it advances this word instead (``training.GraphedStep``), so every
    replay draws fresh decisions while the forward and backward launches of one step still agree."""
    dev = torch.device(device)
    idx = dev.index if dev.index is not None else torch.cuda.current_device()
    t = _DROPOUT_EPOCH.get(idx)
    if t is None:
        t = torch.zeros(1, dtype=torch.int64, device=f"cuda:{idx}")
        with torch.cuda.device(idx):
            hip.check(hip.lib().e3d_dropout_set_epoch_ptr(t.data_ptr()), "e3d_dropout_set_epoch_ptr")
        _DROPOUT_EPOCH[idx] = t
    return t


def next_dropout_seed():
    """A fresh 63-bit seed for one dropout site call, drawn from torch's global CPU generator
    (so ``torch.manual_seed`` makes training runs repeatable)."""
    return int(torch.randint(0, 2 ** 62, (), dtype=torch.int64))


def dropout(x, p, seed, out=None):
    """out = x * keep / (1 - p'): the counter-based dropout of include/e3d_hip.h.  The backward pass is the
    same call on the gradient with the same (p, seed)."""
    _chk(x, "dropout.x")
    assert x.is_contiguous()
    out = torch.empty_like(x) if out is None else out
    hip.check(hip.lib().e3d_dropout_f32(_p(x), float(p), int(seed), _p(out), x.numel(), _stream()), "e3d_dropout_f32")
    return out


def attn_dropout_mask(B, nh, Lq, Lk, p, seed, device="cuda"):
    """Test aid: multipliers applied to the attention probabilities by ``attention(..., drop=(p, seed))``."""
    out = torch.empty((B, nh, Lq, Lk), device=device, dtype=torch.float32)
    hip.check(hip.lib().e3d_attn_dropout_mask(B, nh, Lq, Lk, float(p), int(seed), _p(out), _stream()),
              "e3d_attn_dropout_mask")
    return out


def attention(q, k, v, B, nh, Lq, Lk, key_mask=None, dist_emb=None, max_pos=0, want_lse=False, mode=None, drop=None,
              bounds=None):
    """q [B*Lq, >=nh*64] / k, v [B*Lk, ...] row-strided 2-D views (e.g. slices of a fused QKV
    buffer).  Returns ctx [B*Lq, nh*64] (and lse [B,nh,Lq]).  ``drop`` = (p, seed): dropout on the
    normalised probabilities (training); the exact-fp32 mode then runs its fp32-grade bf16x6 twin.
    ``bounds`` = (q_absmax, k_absmax): 1-element device tensors bounding |element| of the Q rows and of ALL K rows
    (``gemm(..., absmax=)``); with them the split kernels skip all-padding key tiles when that is provably exact,
    without them every call sweeps all keys."""
    for n, t in (("q", q), ("k", k), ("v", v), ("key_mask", key_mask), ("dist_emb", dist_emb)):
        _chk(t, "attention." + n)
    assert q.stride(1) == 1 and k.stride(1) == 1 and v.stride(1) == 1
    assert q.shape[0] == B * Lq and k.shape[0] == B * Lk and v.shape[0] == B * Lk
    if key_mask is not None:
        assert key_mask.is_contiguous() and key_mask.shape == (B, Lk)
    if dist_emb is not None:
        assert dist_emb.is_contiguous() and dist_emb.shape == (2 * max_pos - 1, 64), dist_emb.shape
    out = torch.empty((B * Lq, nh * 64), device=q.device, dtype=torch.float32)
    lse = torch.empty((B, nh, Lq), device=q.device, dtype=torch.float32) if want_lse else None
    terms = GEMM_MODES[ATTN_MODE if mode is None else mode]
    with _timed("attn_relkey" if dist_emb is not None else "attn_cross", (B, nh, Lq, Lk)):
        args = (_p(q), Lq * q.stride(0), q.stride(0), _p(k), Lk * k.stride(0), k.stride(0),
                _p(v), Lk * v.stride(0), v.stride(0), _p(dist_emb), max_pos, _p(key_mask), _p(out), _p(lse),
                B, nh, Lq, Lk)
        dropping = drop is not None and drop[0] > 0
        if terms == 0 and not dropping:
            hip.check(hip.lib().e3d_relkey_attn_fwd(*args, _stream()), "e3d_relkey_attn_fwd")
        else:
            # scratch for the bf16 planes of dist_emb (cooperative kernel): a torch allocation keeps the call
            # free of stream-ordered hipMallocAsync, so the launch sequence can be captured into a HIP graph
            scratch, ready, e_abs = None, 0, None
            q_abs, k_abs = bounds if bounds is not None else (None, None)
            if dist_emb is not None:
                # inference (grad disabled): the planes are kept across calls, keyed by the tensor's version / storage
                infer = not torch.is_grad_enabled()
                ent = getattr(dist_emb, "_e3d_planes", None) if infer else None
                if ent is not None and ent[0] == weight_key(dist_emb) + (Lk, terms):
                    scratch, ready, e_abs = ent[1], 1, ent[2]
                else:
                    scratch = torch.empty((hip.lib().e3d_attn_scratch_bytes(Lk),), device=q.device, dtype=torch.uint8)
                    # training (weights change every step): a pool slot of the table, raised by this call -- never lowered,
                    # so a value from an earlier step is still a bound; inference keeps a scalar of its own with the planes
                    # (a pool slot would be zeroed by the next chain's reset while the cached planes live on)
                    # (the call that writes the planes raises the slot it is given to max |table| itself -- no launch for it -- but
                    #  only a call that carries all three bounds hands the slot over: without them the scalar kept with the cached
                    #  planes is computed here, so that it is a bound for whichever later call re-uses the planes)
                    e_abs = None
                    have_bounds = q_abs is not None and k_abs is not None
                    if have_bounds and not infer:
                        try:
                            e_abs = absmax_slot(dist_emb, "e", q.device)
                        except AttributeError:
                            e_abs = torch.zeros(1, device=q.device, dtype=torch.float32)
                    elif have_bounds:
                        e_abs = torch.zeros(1, device=q.device, dtype=torch.float32)
                    if infer and not torch.cuda.is_current_stream_capturing():
                        try:
                            if e_abs is None:
                                e_abs = absmax(dist_emb.detach())
                            dist_emb._e3d_planes = (weight_key(dist_emb) + (Lk, terms), scratch, e_abs)
                        except AttributeError:
                            pass
            p, seed = (float(drop[0]), int(drop[1])) if dropping else (0.0, 0)
            if q_abs is None or k_abs is None or (dist_emb is not None and e_abs is None):
                q_abs = k_abs = e_abs = None
            hip.check(hip.lib().e3d_relkey_attn_fwd_split_ex(*args, terms or 6, p, seed, _p(scratch), ready, _p(q_abs),
                                                             _p(k_abs), _p(e_abs), _stream()),
                      "e3d_relkey_attn_fwd_split_ex")
    return (out, lse) if want_lse else out


def residual_layernorm(x, residual, gamma, beta, eps, want_s=False, drop=None):
    """LayerNorm(x + residual); ``drop`` = (p, seed): LayerNorm(dropout(x) + residual) with the multipliers of
    ``dropout(x, p, seed)`` applied inside the kernel."""
    for n, t in (("x", x), ("residual", residual), ("gamma", gamma), ("beta", beta)):
        _chk(t, "residual_layernorm." + n)
    assert x.is_contiguous() and (residual is None or residual.is_contiguous())
    M, H = x.shape
    out = torch.empty_like(x)
    s = torch.empty_like(x) if want_s else None
    with _timed("residual_layernorm", (M, H)):
        if drop is not None and drop[0] > 0:
            hip.check(hip.lib().e3d_residual_layernorm_drop_fwd(_p(x), _p(residual), _p(gamma), _p(beta), eps, _p(s), _p(out), M, H,
                                                                float(drop[0]), int(drop[1]), _stream()),
                      "e3d_residual_layernorm_drop_fwd")
        else:
            hip.check(hip.lib().e3d_residual_layernorm_fwd(_p(x), _p(residual), _p(gamma), _p(beta), eps, _p(s),
                                                           _p(out), M, H, _stream()), "e3d_residual_layernorm_fwd")
    return (out, s) if want_s else out


# Row-complete GEMM + bias + residual + LayerNorm (csrc/gemm_rowln.hip): BertSelfOutput / BertOutput in ONE launch at large
# M -- a workgroup owns whole 768-wide rows, the pre-norm sum never leaves its registers.  The weight goes in PRE-SPLIT
# into its two 16-bit terms (fragment-order planes, built once per weight version by a small kernel and cached on the
# weight like every derived-weight cache).  From ROWLN_MIN_M rows upwards: measured against the pair in one process
# (tools/lab/rowln_ab.py, f16x3, K = 768 / 1024): M = 65536 257 / 319 us against 323 / 376, M = 32768 137 / 172 against
# 168 / 208, M = 16384 66 / 83 against 84 / 98, M = 8192 41 / 51 against 53 / 61 (one 32-row tile per CU); smaller launches stay
# on the skinny / 128 x 128 forms.  E3D_GEMM_ROWLN=0 switches the path off (A/B runs).
ROWLN_MIN_M = int(os.environ.get("E3D_GEMM_ROWLN_MIN_M", "8192")) if os.environ.get("E3D_GEMM_ROWLN", "1") == "1" else 1 << 62


def weight_planes(weight, terms):
    """(planes, out_scale): ``weight`` [N, K] split into its hi / lo 16-bit terms in MFMA-fragment order
    (e3d_weight_planes_f32_split); for f16x3 the power-of-two-scaled copy is split and its inverse power returned."""
    ent = getattr(weight, "_e3d_planes_w", None)
    key = weight_key(weight) + (terms,)
    if ent is not None and ent[0] == key:
        return ent[1], ent[2]
    if torch.cuda.is_current_stream_capturing():
        raise RuntimeError("weight_planes: cache miss inside a stream capture (warm the op up before capturing)")
    src, scale = f16_weight(weight) if terms == 19 else (weight.detach(), 1.0)
    N, K = src.shape
    nbytes = hip.lib().e3d_weight_planes_bytes(N, K)
    if nbytes < 0:
        raise ValueError(f"weight_planes: unsupported shape {tuple(src.shape)}")
    planes = torch.empty((nbytes,), dtype=torch.uint8, device=src.device)
    hip.check(hip.lib().e3d_weight_planes_f32_split(_p(src), N, K, terms, _p(planes), _stream()), "e3d_weight_planes_f32_split")
    try:
        weight._e3d_planes_w = (key, planes, scale)
    except AttributeError:
        pass
    return planes, scale


def rowln_ok(terms, M, H, K, a):
    return (terms in (3, 19) and M >= ROWLN_MIN_M and a.stride(1) == 1
            and bool(hip.lib().e3d_gemm_residual_layernorm_supported(M, H, K, a.stride(0))))


def linear_residual_layernorm(a, weight, bias, residual, gamma, beta, eps, mode=None):
    """LayerNorm(a @ weight^T + bias + residual) * gamma + beta  (BertSelfOutput / BertOutput in eval mode).  Large M: the
    row-complete kernel (one launch, no pre-norm tensor); small M: the second launch of the skinny GEMM does the row
    finish (bit-identical to gemm + residual_layernorm); otherwise the pair."""
    M, K = a.shape
    H = weight.shape[0]
    terms = GEMM_MODES[GEMM_MODE if mode is None else mode]
    if rowln_ok(terms, M, H, K, a) and (residual is None or (residual.stride(1) == 1 and residual.stride(0) % 4 == 0)):
        for t, n in ((a, "a"), (weight, "weight"), (bias, "bias"), (residual, "residual"), (gamma, "gamma"), (beta, "beta")):
            _chk(t, "linear_residual_layernorm." + n)
        assert weight.is_contiguous() and weight.shape[1] == K and (residual is None or residual.shape == (M, H))
        planes, scale = weight_planes(weight, terms)
        out = torch.empty((M, H), device=a.device, dtype=torch.float32)
        with _timed("gemm_layernorm", (M, H, K)):
            hip.check(hip.lib().e3d_gemm_residual_layernorm_f32_split(
                _p(a), a.stride(0), _p(planes), _p(bias), _p(residual), residual.stride(0) if residual is not None else 0,
                _p(gamma), _p(beta), eps, _p(out), out.stride(0), M, H, K, terms, scale, _stream()),
                "e3d_gemm_residual_layernorm_f32_split")
        return out
    if not (_skinny_ok(terms, M, H, K, a) and H in (256, 512, 768, 1024) and a.stride(1) == 1):
        return residual_layernorm(gemm(a, weight, bias, mode=mode), residual, gamma, beta, eps)
    for t, n in ((a, "a"), (weight, "weight"), (bias, "bias"), (residual, "residual"), (gamma, "gamma"), (beta, "beta")):
        _chk(t, "linear_residual_layernorm." + n)
    assert weight.is_contiguous() and weight.shape[1] == K and (residual is None or (residual.is_contiguous() and residual.shape == (M, H)))
    out = torch.empty((M, H), device=a.device, dtype=torch.float32)
    ws = _skinny_workspace(a.device, M, H, K)
    scale = 1.0
    if terms == 19:
        weight, scale = f16_weight(weight)
    with _timed("gemm_layernorm", (M, H, K)):
        hip.check(hip.lib().e3d_gemm_skinny_residual_layernorm_f32_split_ex(
            _p(a), a.stride(0), _p(weight), _p(bias), _p(residual), _p(gamma), _p(beta), eps, _p(out), M, H, K, terms,
            _p(ws), ws.numel(), scale, _stream()), "e3d_gemm_skinny_residual_layernorm_f32_split_ex")
    return out

def adaln_gate(x, y, mod, branch, rows_per_cond):
    for n, t in (("x", x), ("y", y), ("mod", mod)):
        _chk(t, "adaln_gate." + n)
    assert x.is_contiguous() and y.is_contiguous() and mod.is_contiguous()
    M, H = x.shape
    assert mod.shape[1] == 6 * H and mod.shape[0] * rows_per_cond == M, (mod.shape, M, rows_per_cond)
    out = torch.empty_like(x)
    with _timed("adaln_gate", (M, H)):
        hip.check(hip.lib().e3d_adaln_gate_fwd(_p(x), _p(y), _p(mod), branch, rows_per_cond, _p(out),
                                               M, H, _stream()), "e3d_adaln_gate_fwd")
    return out


def embed_layernorm(x, weight, bias, gamma, beta, eps, post_add=None, rows_per_add=1, want_z=False):
    for n, t in (("x", x), ("weight", weight), ("bias", bias), ("gamma", gamma), ("beta", beta),
                 ("post_add", post_add)):
        _chk(t, "embed_layernorm." + n)
    assert x.is_contiguous() and weight.is_contiguous()
    M, F = x.shape
    H = weight.shape[0]
    assert weight.shape[1] == F
    if post_add is not None:
        assert post_add.is_contiguous() and post_add.shape == (M // rows_per_add, H)
    out = torch.empty((M, H), device=x.device, dtype=torch.float32)
    z = torch.empty_like(out) if want_z else None
    with _timed("embed_layernorm", (M, H)):
        hip.check(hip.lib().e3d_embed_layernorm_fwd(_p(x), F, _p(weight), _p(bias), _p(gamma), _p(beta), eps,
                                                    _p(post_add), rows_per_add, _p(z), _p(out), M, H, _stream()),
                  "e3d_embed_layernorm_fwd")
    return (out, z) if want_z else out


def head_linear(x, weight, bias):
    for n, t in (("x", x), ("weight", weight), ("bias", bias)):
        _chk(t, "head_linear." + n)
    assert x.is_contiguous() and weight.is_contiguous()
    M, H = x.shape
    n_out = weight.shape[0]
    out = torch.empty((M, n_out), device=x.device, dtype=torch.float32)
    hip.check(hip.lib().e3d_head_linear_fwd(_p(x), _p(weight), _p(bias), _p(out), M, H, n_out, _stream()),
              "e3d_head_linear_fwd")
    return out


def ddpm_step_wrap(x, eps_hat, noise, sqrt_recip_alpha, beta, sqrt_one_minus_ab, sigma, wrap=True, out=None):
    for n, t in (("x", x), ("eps_hat", eps_hat), ("noise", noise)):
        _chk(t, "ddpm_step_wrap." + n)
    assert x.is_contiguous() and eps_hat.is_contiguous() and (noise is None or noise.is_contiguous())
    if out is None:
        out = torch.empty_like(x)
    with _timed("ddpm_step_wrap"):
        hip.check(hip.lib().e3d_ddpm_step_wrap(_p(x), _p(eps_hat), _p(noise), sqrt_recip_alpha, beta,
                                               sqrt_one_minus_ab, sigma, int(wrap), _p(out), x.numel(), _stream()),
                  "e3d_ddpm_step_wrap")
    return out


def ddpm_step_wrap_table(x, eps_hat, noise, coef_table, t_dev, wrap=True, out=None):
    """ddpm_step_wrap with the step index on the device (``t_dev`` int64, first element) and the four
    coefficients in ``coef_table`` [T,4]: nothing host-side changes between steps (HIP-graph replay)."""
    for n, t in (("x", x), ("eps_hat", eps_hat), ("noise", noise), ("coef_table", coef_table)):
        _chk(t, "ddpm_step_wrap_table." + n)
    _chk(t_dev, "ddpm_step_wrap_table.t_dev", torch.int64)
    assert x.is_contiguous() and eps_hat.is_contiguous() and noise.is_contiguous() and coef_table.is_contiguous()
    assert coef_table.dim() == 2 and coef_table.shape[1] == 4
    if out is None:
        out = torch.empty_like(x)
    with _timed("ddpm_step_wrap"):
        hip.check(hip.lib().e3d_ddpm_step_wrap_table(_p(x), _p(eps_hat), _p(noise), _p(coef_table), _p(t_dev), int(wrap),
                                                     _p(out), x.numel(), _stream()), "e3d_ddpm_step_wrap_table")
    return out


def q_sample_wrap(x0, noise, t, sqrt_ab, sqrt_1mab):
    for n, tt in (("x0", x0), ("noise", noise), ("sqrt_ab", sqrt_ab), ("sqrt_1mab", sqrt_1mab)):
        _chk(tt, "q_sample_wrap." + n)
    _chk(t, "q_sample_wrap.t", torch.int64)
    assert x0.is_contiguous() and noise.is_contiguous() and t.is_contiguous()
    B = x0.shape[0]
    out = torch.empty_like(x0)
    hip.check(hip.lib().e3d_q_sample_wrap(_p(x0), _p(noise), _p(t), _p(sqrt_ab), _p(sqrt_1mab), _p(out),
                                          B, x0.numel() // B, _stream()), "e3d_q_sample_wrap")
    return out


def discrete_posterior_sample(xt_idx, logits, qsb, qtb, u=None, want_prob=False):
    """xt_idx int32 [B,L]; logits [B,L,C]; qsb/qtb [B,C,C]; u [B,L] uniforms or None (argmax)."""
    _chk(xt_idx, "xt_idx", torch.int32); _chk(logits, "logits"); _chk(qsb, "qsb"); _chk(qtb, "qtb"); _chk(u, "u")
    B, L, C = logits.shape
    assert xt_idx.is_contiguous() and logits.is_contiguous() and qsb.is_contiguous() and qtb.is_contiguous()
    assert qsb.shape == (B, C, C) and qtb.shape == (B, C, C) and xt_idx.shape == (B, L)
    out = torch.empty((B, L), device=logits.device, dtype=torch.int32)
    prob = torch.empty((B, L, C), device=logits.device, dtype=torch.float32) if want_prob else None
    with _timed("discrete_posterior"):
        hip.check(hip.lib().e3d_discrete_posterior_sample(
            _p(xt_idx), _p(logits), _p(qsb), _p(qtb), _p(u), 0 if u is None else 1, _p(out), _p(prob),
            B, L, C, _stream()), "e3d_discrete_posterior_sample")
    return (out, prob) if want_prob else out


def discrete_q_sample(x0_idx, qtb, u=None):
    """x0_idx int32 [B,L] (-1 = padding row); qtb [B,C,C]; u [B,L] uniforms or None (argmax)."""
    _chk(x0_idx, "x0_idx", torch.int32); _chk(qtb, "qtb"); _chk(u, "u")
    B, L = x0_idx.shape
    C = qtb.shape[-1]
    assert x0_idx.is_contiguous() and qtb.is_contiguous() and qtb.shape == (B, C, C)
    out = torch.empty((B, L), device=qtb.device, dtype=torch.int32)
    with _timed("discrete_q_sample"):
        hip.check(hip.lib().e3d_discrete_q_sample(_p(x0_idx), _p(qtb), _p(u), 0 if u is None else 1, _p(out),
                                                  B, L, C, _stream()), "e3d_discrete_q_sample")
    return out
