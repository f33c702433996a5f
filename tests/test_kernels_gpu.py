"""GPU parity of each HIP kernel (through the C-ABI via ops.py) against the CPU oracle / a plain
fp32 torch CPU statement of the same op.  Tolerance: the north star's 1e-4 relative fp32 (most
kernels are far tighter; the bound asserted is written per test)."""
import math

import pytest
import torch
import torch.nn.functional as F

from helpers import rel_err
from oracle import bert as obert, sequence as oseq, structure as ostr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


# ------------------------------------------------------------------------------------- GEMM
@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (130, 256, 64), (257, 768, 768), (1000, 1024, 3072),
                                   (64, 4608, 768), (4096, 2304, 768)])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("mode,tol", [("f32", 5e-6), ("bf16x6", 5e-6), ("bf16x3", 3e-5), ("f16x3", 5e-6)])
def test_gemm_bias_act(pkg, hip, M, N, K, act, mode, tol):
    """All three arithmetic modes against an fp64 statement: exact fp32 MFMA, and fp32 operands
    split into 3 / 2 bf16 terms on the bf16 matrix cores (fp32 accumulate)."""
    a = torch.randn(M, K, generator=g(M + N))
    w = torch.randn(N, K, generator=g(K)) / math.sqrt(K)
    b = torch.randn(N, generator=g(7))
    ref = F.linear(a.double(), w.double(), b.double())
    ref = {0: lambda x: x, 1: F.gelu, 2: F.silu}[act](ref).float()
    got = pkg.ops.gemm(a.to(DEV), w.to(DEV), b.to(DEV), act, mode=mode)
    assert rel_err(got, ref) < tol


@pytest.mark.parametrize("M,N,K,act", [(16384, 1024, 768, 1),     # >= 256 tiles of 256x256, M % 256 == 0: persistent kernel
                                       (32768, 768, 1024, 0),     # 3 tiles per CU walk, K = 1024
                                       (16400, 1024, 768, 2),     # ragged M: interleaved-staging kernel with row clamps
                                       (16384, 1024, 32, 0),      # a single k-tile: the stream wraps every tile
                                       (16384, 768, 768, 0)])     # 192 tiles: fewer persistent workgroups than CUs
def test_gemm_large_m_kernels(pkg, hip, M, N, K, act):
    """The large-M forward kernels (256x256 tiles; bench.py shapes are of this kind) against fp64, with a
    strided activation view (lda > K) as the packed QKV / KV consumers pass."""
    wide = torch.randn(M, K + 64, generator=g(M))
    a = wide[:, 32:32 + K]
    w = torch.randn(N, K, generator=g(K)) / math.sqrt(K)
    b = torch.randn(N, generator=g(7))
    ref = F.linear(a.double(), w.double(), b.double())
    ref = {0: lambda x: x, 1: F.gelu, 2: F.silu}[act](ref).float()
    a_dev = wide.to(DEV)[:, 32:32 + K]
    for mode, tol in (("bf16x3", 3e-5), ("bf16x6", 5e-6), ("f16x3", 5e-6)):
        got = pkg.ops.gemm(a_dev, w.to(DEV), b.to(DEV), act, mode=mode)
        assert rel_err(got, ref) < tol, mode
        # every row and column block is right, not only the global max-norm
        blk = (got.cpu() - ref).abs().view(-1, 16, N).amax(1).amax(1) if M % 16 == 0 else None
        if blk is not None:
            assert blk.max() < 3 * tol * ref.abs().max(), mode


def test_gemm_split_handles_wide_dynamic_range(pkg, hip):
    """Split terms must reconstruct operands spanning many binades (outliers next to tiny values)."""
    a = torch.randn(300, 256, generator=g(1)) * torch.logspace(-6, 4, 256)[None, :]
    w = torch.randn(128, 256, generator=g(2)) * torch.logspace(3, -5, 256)[None, :]
    ref = (a.double() @ w.double().t()).float()
    assert rel_err(pkg.ops.gemm(a.to(DEV), w.to(DEV), None, mode="bf16x6"), ref) < 5e-6
    assert rel_err(pkg.ops.gemm(a.to(DEV), w.to(DEV), None, mode="bf16x3"), ref) < 3e-5


def test_gemm_strided_input_and_no_bias(pkg, hip):
    a = torch.randn(300, 3 * 256, generator=g(1))
    w = torch.randn(128, 256, generator=g(2)) / 16
    got = pkg.ops.gemm(a.to(DEV)[:, 256:512], w.to(DEV), None)
    assert rel_err(got, a[:, 256:512] @ w.t()) < 5e-6


def test_gemm_rejects_bad_shapes(pkg, hip):
    a, w = torch.zeros(4, 48, device=DEV), torch.zeros(100, 48, device=DEV)
    with pytest.raises(RuntimeError, match="N%128"):
        pkg.ops.gemm(a, w)
    with pytest.raises(RuntimeError, match="GPU tensor"):
        pkg.ops.gemm(a.cpu(), w)


# ------------------------------------------------------------------------------------- attention
def ref_attention(q, k, v, mask, E, P):
    """q,k,v [B,nh,L,64] fp64 statement of SURVEY App. A eq. (1)-(4)."""
    s = q @ k.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(q, E, P)
    s = s / 8.0
    if mask is not None:
        s = s + ((1.0 - mask) * -10000.0)[:, None, None, :]
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,nh,L,P", [(2, 4, 16, 16), (1, 12, 64, 64), (3, 2, 50, 64), (2, 12, 128, 128),
                                      (1, 3, 256, 256), (2, 1, 33, 40)])
@pytest.mark.parametrize("relkey", [True, False])
@pytest.mark.parametrize("mode,tol", [("f32", 1e-5), ("bf16x6", 1e-5), ("bf16x3", 1e-4), ("f16x3", 1e-5)])
def test_attention_self(pkg, hip, B, nh, L, P, relkey, mode, tol):
    H = nh * 64
    qkv = torch.randn(B * L, 3 * H, generator=g(L))
    E = torch.randn(2 * P - 1, 64, generator=g(P + 1)) if relkey else None
    lens = torch.randint(1, L + 1, (B,), generator=g(3))
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    dq = qkv.to(DEV)
    got, lse = pkg.ops.attention(dq[:, :H], dq[:, H:2 * H], dq[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV),
                                 dist_emb=None if E is None else E.to(DEV), max_pos=P, want_lse=True, mode=mode)
    split = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()  # noqa: E731
    q, k, v = split(qkv[:, :H]), split(qkv[:, H:2 * H]), split(qkv[:, 2 * H:])
    ref = ref_attention(q, k, v, mask.double(), None if E is None else E.double(), P)
    ref = ref.permute(0, 2, 1, 3).reshape(B * L, H).float()
    assert rel_err(got, ref) < tol
    s = q @ k.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(q, E.double(), P)
    s = s / 8.0 + ((1.0 - mask.double()) * -10000.0)[:, None, None, :]
    assert rel_err(lse, torch.logsumexp(s, -1).float()) < tol


def test_attention_cross_rectangular_and_nomask(pkg, hip):
    B, nh, Lq, Lk = 2, 3, 40, 70
    H = nh * 64
    qb = torch.randn(B * Lq, H, generator=g(1))
    kv = torch.randn(B * Lk, 2 * H, generator=g(2))
    dkv = kv.to(DEV)
    sp = lambda x, L: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()  # noqa: E731
    ref = ref_attention(sp(qb, Lq), sp(kv[:, :H], Lk), sp(kv[:, H:], Lk), None, None, 0)
    for mode, tol in (("f32", 1e-5), ("bf16x6", 1e-5), ("bf16x3", 1e-4), ("f16x3", 1e-5)):
        got = pkg.ops.attention(qb.to(DEV), dkv[:, :H], dkv[:, H:], B, nh, Lq, Lk, mode=mode)
        assert rel_err(got, ref.permute(0, 2, 1, 3).reshape(B * Lq, H).float()) < tol, mode


def test_attention_fully_padded_item_matches_reference_semantics(pkg, hip):
    """An all-zero key mask adds -10000 to every key in fp32, as the reference does: the softmax
    is that of scores quantised to ulp(10000) ~ 1e-3, NOT -inf/NaN.  Checked against the same fp32
    statement on the CPU; a one-ulp difference of a score moves a probability by ~1e-3, hence the
    looser bound of this degenerate case."""
    B, nh, L = 1, 1, 32
    qkv = torch.randn(L, 192, generator=g(5))
    d = qkv.to(DEV)
    a = pkg.ops.attention(d[:, :64], d[:, 64:128], d[:, 128:], B, nh, L, L, key_mask=torch.zeros(1, L, device=DEV))
    q, k, v = qkv[:, :64], qkv[:, 64:128], qkv[:, 128:]
    s = (q @ k.t()) / 8.0 + (1.0 - torch.zeros(1, L)) * -10000.0
    assert torch.isfinite(a).all()
    assert rel_err(a, torch.softmax(s, -1) @ v) < 5e-3


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6", "f16x3"])
def test_attention_padded_tile_skipping_is_bit_exact(pkg, hip, mode):
    """Stopping the key sweep after the last valid key's tile must not change a single bit
    (trailing padding tiles contribute exp(s - 10000 - m) == 0 while the element bounds prove the score spread small),
    incl. masks with holes and an all-padding item (which must keep the dense sweep)."""
    B, nh, L, P = 4, 2, 256, 256
    H = nh * 64
    qkv = (torch.randn(B * L, 3 * H, generator=g(1)) * 1.5).to(DEV)
    E = torch.randn(2 * P - 1, 64, generator=g(2)).to(DEV)
    mask = torch.zeros(B, L)
    mask[0, :37] = 1
    mask[1, :200] = 1
    mask[1, 50:90] = 0           # hole
    mask[2, :] = 1
    # item 3: no valid key at all
    mask = mask.to(DEV)
    bound = pkg.ops.absmax(qkv)
    assert float(bound) == float(qkv.abs().max())
    outs = []
    for enable in (1, 0):
        prev = hip.e3d_attn_skip_padded_tiles(enable)
        try:
            outs.append(pkg.ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask,
                                          dist_emb=E, max_pos=P, want_lse=True, mode=mode, bounds=(bound, bound)))
        finally:
            hip.e3d_attn_skip_padded_tiles(prev)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
    assert torch.isfinite(outs[0][0]).all()


@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6", "f16x3"])
@pytest.mark.parametrize("L", [64, 256])     # per-wave kernel / cooperative kernel
def test_attention_skips_padded_tiles_only_when_the_bounds_prove_it_exact(pkg, hip, mode, L):
    """The skip is engaged exactly when 16 qa (ka + ea) < 9890 (include/e3d_hip.h).  Probe: V rows of the all-padding key
    tiles are NaN -- the dense sweep multiplies them by P == 0 and returns NaN, the skipping sweep never reads them."""
    B, nh, P = 2, 2, L
    H = nh * 64
    qkv = torch.randn(B * L, 3 * H, generator=g(3))
    E = torch.randn(2 * P - 1, 64, generator=g(4)).to(DEV)
    mask = torch.zeros(B, L)
    mask[0, :20] = 1
    mask[1, :L] = 1
    small = pkg.ops.absmax(qkv.to(DEV))
    qkv = qkv.view(B, L, 3 * H)
    qkv[0, 32:, 2 * H:] = float("nan")
    qkv = qkv.view(B * L, 3 * H).to(DEV)
    big = torch.full((1,), 1.0e3, device=DEV)
    nan = torch.full((1,), float("nan"), device=DEV)
    run = lambda b: pkg.ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV),  # noqa: E731
                                      dist_emb=E, max_pos=P, mode=mode, bounds=b)
    assert torch.isfinite(run((small, small))).all()                 # proven small: padded tiles never read
    for b in (None, (big, small), (small, big), (nan, small), (small, nan)):
        out = run(b).view(B, L, H)
        assert torch.isnan(out[0]).all() and torch.isfinite(out[1]).all(), b   # dense sweep of item 0
    # the distance table's bound is raised by the call that writes its planes (no launch of its own since round 4): a table
    # with large elements must switch the skip off although q and k are proven small ...
    run_e = lambda e, p_: pkg.ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV),  # noqa: E731
                                            dist_emb=e, max_pos=p_, mode=mode, bounds=(small, small))
    out = run_e(E * 1.0e3, P).view(B, L, H)
    assert torch.isnan(out[0]).all() and torch.isfinite(out[1]).all()
    # ... and only the rows a valid (query, key) pair can reach count: max_pos = 2 L, a huge element in a row no pair of an
    # L-token frame reaches leaves the skip on, the same element inside the window switches it off
    E2 = torch.randn(4 * L - 1, 64, generator=g(5)).to(DEV)
    far, near = E2.clone(), E2.clone()
    far[0, 3] = 1.0e6
    near[2 * L - 1, 3] = 1.0e6
    assert torch.isfinite(run_e(far, 2 * L)).all()
    assert torch.isnan(run_e(near, 2 * L).view(B, L, H)[0]).all()
    # inference keeps the planes AND the table's bound across calls: a first call without bounds must still leave a valid bound
    # for a later call that brings them (the planes are then ready: nothing would raise the scalar any more)
    with torch.no_grad():
        E3 = E.clone() * 1.0e3
        first = pkg.ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV), dist_emb=E3,
                                  max_pos=P, mode=mode)
        assert torch.isnan(first.view(B, L, H)[0]).all()
        assert torch.isnan(run_e(E3, P).view(B, L, H)[0]).all()       # cached planes, cached bound: still the dense sweep
        E4 = E.clone()
        assert torch.isfinite(run_e(E4, P)).all() and torch.isfinite(run_e(E4, P)).all()   # small table: skip on, twice


def _attention_case(pkg, B, nh, L, scale, seed, lens):
    """qkv (Q, K x ``scale``), distance table, mask and the fp64 statement of the reference semantics."""
    H, P = nh * 64, L
    qkv = torch.randn(B * L, 3 * H, generator=g(seed))
    qkv[:, :2 * H] *= scale
    E = torch.randn(2 * P - 1, 64, generator=g(seed + 1))
    mask = (torch.arange(L)[None] < torch.tensor(lens)[:, None]).float()
    split = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()  # noqa: E731
    q, k, v = split(qkv[:, :H]), split(qkv[:, H:2 * H]), split(qkv[:, 2 * H:])
    ref = ref_attention(q, k, v, mask.double(), E.double(), P).permute(0, 2, 1, 3).reshape(B * L, H).float()
    return qkv, E, mask, ref, (q, k, v)


@pytest.mark.parametrize("L", [64, 256])
@pytest.mark.parametrize("mode,tol", [("f32", 2e-4), ("bf16x6", 2e-4), ("f16x3", 2e-4), ("bf16x3", 4e-3)])
def test_attention_sharp_softmax_rows_match_fp64(pkg, hip, mode, tol, L):
    """logits far beyond std 1 -- Q, K x16, scores of std ~256 (rows are all but one-hot), the regime
    trained checkpoints move towards -- in every arithmetic against the fp64 statement, through the bounded call
    (the skip decision is live: 16 qa (ka + ea) is ~1e5 here, so the kernels must run the dense sweep).  A score carries an
    absolute error of ~|s| 2^-24 (fp32 grade) or ~|s| 2^-17 (bf16x3), which is the relative error of its probability."""
    B, nh = 2, 3
    H = nh * 64
    qkv, E, mask, ref, _ = _attention_case(pkg, B, nh, L, 16.0, 11, [L, max(1, L // 5)])
    d = qkv.to(DEV)
    bound = pkg.ops.absmax(d[:, :2 * H].contiguous())
    got = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV), dist_emb=E.to(DEV),
                            max_pos=L, mode=mode, bounds=(bound, bound))
    assert rel_err(got, ref) < tol, rel_err(got, ref)


@pytest.mark.parametrize("L", [64, 256])
@pytest.mark.parametrize("mode", ["f32", "bf16x6", "f16x3", "bf16x3"])
def test_attention_padded_keys_reach_the_softmax_when_scores_exceed_the_mask(pkg, hip, mode, L):
    """The reference's mask is ADDITIVE (-10000, structure_model/model.py:226-231): once scores spread over more than
    that, padded keys win softmax rows.  Round 2's kernels skipped all-padding key tiles unconditionally and were 0.47
    off the oracle in the weights-x4 regime of the margin test (first encoder layer: |q|, |k| ~ 7000); the skip is now
    taken only when the element bounds prove it exact.  Construction: queries and PADDED keys share a direction u
    (q.k / 8 ~ +8192), valid keys point the other way (~ -8192): after the mask the padded keys lead by ~6000, so every
    row of item 0 attends to padding, as the reference would.  Small-integer operands: the products are exact in every
    arithmetic, what is left is the scale / exponent / PV rounding."""
    B, nh = 2, 2
    H = nh * 64
    lens = [L // 4, L]
    u = torch.where(torch.rand(64, generator=g(31)) < 0.5, -1.0, 1.0)
    noise = lambda seed, *shape, sd=8.0: torch.round(torch.randn(*shape, generator=g(seed)) * sd).clamp(-60, 60)  # noqa: E731
    valid = (torch.arange(L)[None] < torch.tensor(lens)[:, None])                      # [B,L]
    q = 32.0 * u + noise(34, B, L, nh, 64)
    k = torch.where(valid[:, :, None, None], -32.0 * u, 32.0 * u) + noise(35, B, L, nh, 64)
    v = torch.randn(B, L, nh, 64, generator=g(33))
    qkv = torch.cat([q.reshape(B * L, H), k.reshape(B * L, H), v.reshape(B * L, H)], 1).contiguous()
    E = noise(36, 2 * L - 1, 64, sd=2.0)
    mask = valid.float()
    sp = lambda x: x.permute(0, 2, 1, 3).double()  # noqa: E731
    qd, kd, vd = sp(q), sp(k), sp(v)
    ref = ref_attention(qd, kd, vd, mask.double(), E.double(), L).permute(0, 2, 1, 3).reshape(B * L, H).float()
    hard = torch.softmax((qd @ kd.transpose(-1, -2) + obert.relkey_scores_literal(qd, E.double(), L)) / 8.0
                         + torch.where(valid, 0.0, -float("inf"))[:, None, None, :].double(), -1) @ vd
    hard = hard.permute(0, 2, 1, 3).reshape(B * L, H).float()
    assert rel_err(hard, ref) > 0.1          # the case does exercise the additive-mask semantics
    d = qkv.to(DEV)
    bound = pkg.ops.absmax(d[:, :2 * H].contiguous())
    got = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV), dist_emb=E.to(DEV),
                            max_pos=L, mode=mode, bounds=(bound, bound))
    assert rel_err(got, ref) < (5e-2 if mode == "bf16x3" else 2e-3), rel_err(got, ref)


@pytest.mark.parametrize("M,N,K", [(64, 768, 768), (512, 2304, 768), (2000, 768, 1024), (4096, 2304, 768), (5120, 2304, 768)])
@pytest.mark.parametrize("mode", ["bf16x3", "bf16x6", "f16x3"])
def test_gemm_reports_the_largest_output_magnitude(pkg, hip, M, N, K, mode):
    """``absmax=``: every forward kernel form (skinny, 128x128 / 256x128 general, 256x256, persistent) raises the device
    scalar to exactly max |out| -- and never lowers it."""
    a = torch.randn(M, K, generator=g(M)).to(DEV)
    w = (torch.randn(N, K, generator=g(N + 1)) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g(3)).to(DEV)
    slot = torch.zeros(1, device=DEV)
    out = pkg.ops.gemm(a, w, b, mode=mode, absmax=slot)
    assert float(slot) == float(out.abs().max())
    assert torch.equal(out, pkg.ops.gemm(a, w, b, mode=mode))        # same results with and without the report
    slot.fill_(1.0e9)
    pkg.ops.gemm(a, w, b, mode=mode, absmax=slot)
    assert float(slot) == 1.0e9
    a[M // 2, 3] = float("nan")
    slot.zero_()
    pkg.ops.gemm(a, w, b, mode=mode, absmax=slot)
    assert math.isnan(float(slot))                                     # a NaN output reads as "no bound"


@pytest.mark.parametrize("mode,tol", [("f32", 2e-6), ("bf16x6", 2e-6), ("f16x3", 3e-6), ("bf16x3", 2e-5)])
def test_gemm_large_activations(pkg, hip, mode, tol):
    """activations x16 (and x256, the first encoder layer of the weights-x4 regime) -- the split
    arithmetics are scale-free inside their range: same relative accuracy as at unit scale."""
    a = torch.randn(777, 768, generator=g(1))
    w = torch.randn(256, 768, generator=g(2)) / math.sqrt(768)
    for scale in (16.0, 256.0):
        ref = ((a * scale).double() @ w.double().t()).float()
        got = pkg.ops.gemm((a * scale).to(DEV), w.to(DEV), None, mode=mode)
        assert rel_err(got, ref) < tol, (scale, rel_err(got, ref))


def test_attention_rejects_relkey_longer_than_table(pkg, hip):
    d = torch.zeros(80, 192, device=DEV)
    with pytest.raises(RuntimeError, match="relative_key"):
        pkg.ops.attention(d[:, :64], d[:, 64:128], d[:, 128:], 1, 1, 80, 80,
                          dist_emb=torch.zeros(127, 64, device=DEV), max_pos=64)


# ------------------------------------------------------------------------------------- row ops
@pytest.mark.parametrize("H", [256, 512, 768, 1024])
@pytest.mark.parametrize("M", [1, 5, 1030])
def test_residual_layernorm(pkg, hip, H, M):
    x, r = torch.randn(M, H, generator=g(H)) * 3, torch.randn(M, H, generator=g(M))
    ga, be = 1 + 0.1 * torch.randn(H, generator=g(1)), torch.randn(H, generator=g(2))
    ref = F.layer_norm((x + r).double(), (H,), ga.double(), be.double(), 1e-12).float()
    got = pkg.ops.residual_layernorm(x.to(DEV), r.to(DEV), ga.to(DEV), be.to(DEV), 1e-12)
    assert rel_err(got, ref) < 2e-6
    got = pkg.ops.residual_layernorm(x.to(DEV), None, ga.to(DEV), be.to(DEV), 1e-12)
    assert rel_err(got, F.layer_norm(x.double(), (H,), ga.double(), be.double(), 1e-12).float()) < 2e-6


@pytest.mark.parametrize("rows_per_cond", [1, 16])
@pytest.mark.parametrize("branch", [0, 1])
def test_adaln_gate(pkg, hip, rows_per_cond, branch):
    B, L, H = 3, 16, 768
    M = B * L
    x, y = torch.randn(M, H, generator=g(1)), torch.randn(M, H, generator=g(2)) * 2
    mod = torch.randn(M // rows_per_cond, 6 * H, generator=g(3))
    sh, sc, ga = [mod[:, (3 * branch + i) * H:(3 * branch + i + 1) * H].repeat_interleave(rows_per_cond, 0)
                  for i in range(3)]
    ref = x + ga * (F.layer_norm(y, (H,)) * (1 + sc) + sh)
    got = pkg.ops.adaln_gate(x.to(DEV), y.to(DEV), mod.to(DEV), branch, rows_per_cond)
    assert rel_err(got, ref) < 2e-6


@pytest.mark.parametrize("M", [48, 1040])     # one wave per row (M <= 512) / LDS-staged W^T
@pytest.mark.parametrize("Fin,H", [(8, 768), (20, 768), (20, 256)])
def test_embed_layernorm(pkg, hip, Fin, H, M):
    L = 16
    x = torch.randn(M, Fin, generator=g(1))
    w, b = torch.randn(H, Fin, generator=g(2)), torch.randn(H, generator=g(3))
    ga, be = 1 + 0.1 * torch.randn(H, generator=g(4)), torch.randn(H, generator=g(5))
    add = torch.randn(M // L, H, generator=g(6))
    ref = F.layer_norm(F.linear(x, w, b), (H,), ga, be, 1e-12)
    d = lambda t: t.to(DEV)  # noqa: E731
    assert rel_err(pkg.ops.embed_layernorm(d(x), d(w), d(b), d(ga), d(be), 1e-12), ref) < 2e-6
    got = pkg.ops.embed_layernorm(d(x), d(w), d(b), d(ga), d(be), 1e-12, d(add), L)
    assert rel_err(got, ref + add.repeat_interleave(L, 0)) < 2e-6
    if M > 512:   # the two forms do the same arithmetic in the same order
        few = pkg.ops.embed_layernorm(d(x[:48]), d(w), d(b), d(ga), d(be), 1e-12, d(add[:3]), L)
        assert torch.equal(few, got[:48])


@pytest.mark.parametrize("n_out", [8, 20])
def test_head_linear(pkg, hip, n_out):
    x = torch.randn(77, 768, generator=g(1))
    w, b = torch.randn(n_out, 768, generator=g(2)) / 27, torch.randn(n_out, generator=g(3))
    got = pkg.ops.head_linear(x.to(DEV), w.to(DEV), b.to(DEV))
    assert rel_err(got, F.linear(x.double(), w.double(), b.double()).float()) < 2e-6


# ------------------------------------------------------------------------------------- samplers
def test_ddpm_step_wrap_bit_exact_vs_oracle(pkg, hip):
    T = 1000
    tab = ostr.compute_alphas(ostr.cosine_beta_schedule(T))
    x = ostr.modulo_with_wrapped_range(torch.randn(5, 33, 8, generator=g(1)) * 2)
    eps, noise = torch.randn(5, 33, 8, generator=g(2)), torch.randn(5, 33, 8, generator=g(3))
    for t in (0, 1, 500, 999):
        const = lambda t, x, *a: eps  # noqa: E731
        want = ostr.p_sample(const, None, x, None, None, None, torch.full((5,), t), tab["betas"], noise)
        args = (float(1.0 / torch.sqrt(tab["alphas"][t])), float(tab["betas"][t]),
                float(tab["sqrt_one_minus_alphas_cumprod"][t]),
                0.0 if t == 0 else float(torch.sqrt(tab["posterior_variance"][t])))
        got = pkg.ops.ddpm_step_wrap(x.to(DEV), eps.to(DEV), noise.to(DEV), *args, wrap=False).cpu()
        assert torch.allclose(got, want, rtol=1e-6, atol=1e-6), t
        got_w = pkg.ops.ddpm_step_wrap(x.to(DEV), eps.to(DEV), noise.to(DEV), *args, wrap=True).cpu()
        d = ostr.modulo_with_wrapped_range(got_w - ostr.modulo_with_wrapped_range(want)).abs().max()
        assert d < 1e-5 and got_w.min() >= -math.pi - 1e-6 and got_w.max() < math.pi + 1e-6


def test_wrap_kernel_edge_values(pkg, hip):
    v = torch.tensor([3.0, -3.0, 0.0, math.pi, -math.pi, 7.5, -7.5, 100.0, -100.0, 3.1415927410125732,
                      -3.1415927410125732, 1e-8])
    z = torch.zeros_like(v)
    got = pkg.ops.ddpm_step_wrap(v.to(DEV), z.to(DEV), None, 1.0, 0.0, 1.0, 0.0, wrap=True).cpu()
    assert torch.equal(got, ostr.modulo_with_wrapped_range(v))     # bit-exact incl. the +-pi edges


def test_q_sample_wrap(pkg, hip):
    T = 100
    tab = ostr.compute_alphas(ostr.cosine_beta_schedule(T))
    x0 = ostr.modulo_with_wrapped_range(torch.randn(6, 16, 8, generator=g(1)) * 2)
    noise = ostr.modulo_with_wrapped_range(torch.randn(6, 16, 8, generator=g(2)))
    t = torch.tensor([0, 5, 37, 50, 98, 99])
    want = torch.stack([ostr.add_noise_by_timestep(x0[i], int(t[i]), tab, noise[i]) for i in range(6)])
    got = pkg.ops.q_sample_wrap(x0.to(DEV), noise.to(DEV), t.to(DEV), tab["sqrt_alphas_cumprod"].to(DEV),
                                tab["sqrt_one_minus_alphas_cumprod"].to(DEV)).cpu()
    assert ostr.modulo_with_wrapped_range(got - want).abs().max() < 2e-6


def _blosum():
    import os
    from helpers import GOLDEN
    return torch.load(os.path.join(GOLDEN, "blosum_substitute.pt"), weights_only=True)


@pytest.mark.parametrize("trans_name", ["blosum", "uniform"])
def test_discrete_posterior_sample(pkg, hip, trans_name):
    B, L, C, T = 4, 37, 20, 50
    sched = oseq.NoiseScheduleDiscrete(T)
    trans = oseq.BlosumTransition(_blosum()) if trans_name == "blosum" else oseq.UniformTransition(C)
    xt_idx = torch.randint(0, C, (B, L), generator=g(1))
    x_t = F.one_hot(xt_idx, C).float()
    logits = torch.randn(B, L, C, generator=g(2)) * 2
    u = torch.rand(B, L, generator=g(3))
    for s_int in (1, 24, 48):
        s = s_int * torch.ones(B, 1) / T
        t = (s_int + 1) * torch.ones(B, 1) / T
        prob = oseq.reverse_prob(t, s, x_t, logits, sched, trans)
        qtb = trans.get_Qt_bar(sched.get_alpha_bar(t)).to(DEV)
        qsb = trans.get_Qt_bar(sched.get_alpha_bar(s)).to(DEV)
        idx, p = pkg.ops.discrete_posterior_sample(xt_idx.int().to(DEV), logits.to(DEV), qsb, qtb, None, True)
        assert rel_err(p.reshape(-1, C), prob) < 1e-5
        # argmax: identical wherever the oracle's top-2 gap exceeds the fp tolerance
        top2 = prob.topk(2, -1).values
        clear = (top2[:, 0] - top2[:, 1]) > 1e-5
        assert torch.equal(idx.cpu().reshape(-1).long()[clear], prob.argmax(-1)[clear])
        idx_u = pkg.ops.discrete_posterior_sample(xt_idx.int().to(DEV), logits.to(DEV), qsb, qtb, u.to(DEV))
        want_u = oseq.categorical_from_uniform(prob, u.reshape(-1))
        cdf = prob.cumsum(-1)
        clear = ((cdf - (u.reshape(-1, 1) * cdf[:, -1:])).abs().min(-1).values) > 1e-5
        assert torch.equal(idx_u.cpu().reshape(-1).long()[clear], want_u[clear])
        assert clear.float().mean() > 0.99


def test_discrete_q_sample(pkg, hip):
    B, L, C, T = 3, 16, 20, 50
    sched, trans = oseq.NoiseScheduleDiscrete(T), oseq.BlosumTransition(_blosum())
    x0 = torch.randint(0, C, (B, L), generator=g(1))
    x0[:, 10:] = -1                                        # padding rows
    onehot = F.one_hot(x0.clamp_min(0), C).float() * (x0 >= 0)[..., None]
    t_int = torch.tensor([[3.0], [25.0], [50.0]])
    u = torch.rand(B, L, generator=g(2))
    qtb = trans.get_Qt_bar(sched.get_alpha_bar(t_int / T)).to(DEV)
    want = oseq.apply_aa_noise(onehot, t_int, sched, trans, u=u).argmax(-1)
    got = pkg.ops.discrete_q_sample(x0.int().to(DEV), qtb, u.to(DEV)).cpu().long()
    prob = oseq.aa_noise_prob(onehot, t_int, sched, trans)
    cdf = prob.cumsum(-1)
    clear = ((cdf - (u.reshape(-1, 1) * cdf[:, -1:])).abs().min(-1).values > 1e-6) | (prob.sum(-1) == 0)
    assert torch.equal(got.reshape(-1)[clear], want.reshape(-1)[clear])
    assert bool((got[:, 10:] == 0).all())


def test_distance_table_planes_cache_follows_weight_updates(pkg, hip):
    """ops.attention keeps the bf16 planes of dist_emb across inference calls (keyed by the tensor's version and
    storage): an in-place update of the table must be seen by the next call."""
    ops = pkg.ops
    B, nh, L, H = 2, 2, 128, 128
    qkv = torch.randn(B * L, 3 * H, generator=g(1)).to(DEV)
    E = torch.randn(2 * L - 1, 64, generator=g(2)).to(DEV)
    mask = torch.ones(B, L, device=DEV)
    call = lambda e: ops.attention(qkv[:, :H], qkv[:, H:2 * H], qkv[:, 2 * H:], B, nh, L, L, key_mask=mask,  # noqa: E731
                                   dist_emb=e, max_pos=L, mode="bf16x3")
    with torch.no_grad():
        out1 = call(E)
        assert getattr(E, "_e3d_planes", None) is not None
        assert torch.equal(call(E), out1)            # served from the cached planes
        E.mul_(-1.5)                                  # version bump
        out2 = call(E)
        fresh = call(E.clone())                       # a tensor without cache entry
    assert torch.equal(out2, fresh) and not torch.equal(out2, out1)


@pytest.mark.parametrize("mode", ["bf16x3", "f16x3"])
@pytest.mark.parametrize("relkey", [True, False])
def test_attention_deferred_rescale_branch_is_forced_and_exact(pkg, hip, relkey, mode):
    """The cooperative kernel only raises its running softmax maximum when a key tile exceeds it by more than 2^tau
    (tau = 8): bounded random data never takes that branch after the first tile, so force it -- spiked keys in late
    tiles (one query jumps at tile 6, another at tile 3 AND again at tile 7, a third by LESS than tau so that it keeps
    its stale maximum) -- and check (1) against the fp64 statement and (2) tau = 8 against tau = 0 (classic online
    softmax, rescale on every new maximum): the two must agree to rounding."""
    B, nh, L, P = 2, 2, 256, 256
    H = nh * 64
    qkv = torch.randn(B * L, 3 * H, generator=g(11)) * 0.5
    q, k = qkv[:, :H], qkv[:, H:2 * H]

    def spike(b, h, query, key, score):      # make (q . k) / 8 of that pair == score
        qv = q[b * L + query, 64 * h:64 * h + 64]
        k[b * L + key, 64 * h:64 * h + 64] = qv * (8.0 * score / float(qv @ qv))

    spike(0, 0, 5, 200, 40.0)
    spike(0, 1, 70, 100, 25.0)
    spike(0, 1, 70, 250, 45.0)
    spike(1, 0, 33, 130, 4.0)                # 4 / ln 2 = 5.8 log2 units above ~0: below tau, no rescale
    E = torch.randn(2 * P - 1, 64, generator=g(12)) * 0.2 if relkey else None
    mask = torch.ones(B, L)
    mask[1, 240:] = 0
    d = qkv.to(DEV)
    outs = {}
    for tau in (8.0, 0.0):
        prev = hip.e3d_attn_rescale_tau(tau)
        try:
            outs[tau] = pkg.ops.attention(d[:, :H], d[:, H:2 * H], d[:, 2 * H:], B, nh, L, L, key_mask=mask.to(DEV),
                                          dist_emb=None if E is None else E.to(DEV), max_pos=P, want_lse=True,
                                          mode=mode)
        finally:
            hip.e3d_attn_rescale_tau(prev)
    split = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3).double()  # noqa: E731
    qd, kd, vd = split(qkv[:, :H]), split(qkv[:, H:2 * H]), split(qkv[:, 2 * H:])
    ref = ref_attention(qd, kd, vd, mask.double(), None if E is None else E.double(), P)
    ref = ref.permute(0, 2, 1, 3).reshape(B * L, H).float()
    s = qd @ kd.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(qd, E.double(), P)
    s = s / 8.0 + ((1.0 - mask.double()) * -10000.0)[:, None, None, :]
    lse_ref = torch.logsumexp(s, -1).float()
    for tau, (got, lse) in outs.items():
        assert rel_err(got, ref) < 1e-4, tau
        assert rel_err(lse, lse_ref) < 1e-4, tau
        assert (got[5].cpu() - ref[5]).abs().max() < 1e-4 * ref.abs().max()      # the spiked query itself
    assert rel_err(outs[8.0][0], outs[0.0][0]) < 1e-5 and rel_err(outs[8.0][1], outs[0.0][1]) < 1e-5   # measured 4e-6


def test_f16x3_range_contract(pkg, hip):
    """f16x3 = two fp16 terms per operand (include/e3d_hip.h, E3D_TERMS_F16X3): fp32-grade for operands inside the
    fp16 range.  The contract at its edges, against fp64.  The KERNEL (``prescale=False``): (1) elements far below
    2^-14 keep an ABSOLUTE error of 2^-25 each -- norm-wise still fp32 grade when the operand's large elements are O(1);
    (2) a uniformly tiny operand (every |w| ~ 1e-6) degrades to that floor; (3) values beyond 65504 turn the affected
    outputs into inf/NaN -- loud, never silently wrong; (4) bf16x6 has none of these limits.  The OP (``ops.gemm``, what
    every model call runs): weights are pre-scaled by an exact power of two (ops.f16_weight) and the epilogue undoes it,
    so the weight side has no range limit left -- tiny (1e-4, 1e-30) and huge (1e6) weights are fp32 grade (VERDICT r02
    item 2: 2e-2 -> <= 1e-5) -- and typical nn.Linear weights (|w| ~ 0.03, low terms otherwise subnormal) gain too."""
    M, N, K = 512, 256, 768
    a = torch.randn(M, K, generator=g(1))
    w = torch.randn(N, K, generator=g(2)) / math.sqrt(K)
    ref = a.double() @ w.double().t()
    raw = rel_err(pkg.ops.gemm(a.to(DEV), w.to(DEV), None, mode="f16x3", prescale=False), ref.float())
    scaled = rel_err(pkg.ops.gemm(a.to(DEV), w.to(DEV), None, mode="f16x3"), ref.float())
    assert raw < 2e-6 and scaled < 1.2e-6, (raw, scaled)
    # (1) wide dynamic range inside one operand: columns spanning 1e-6 .. 1e2
    scale = torch.logspace(-6, 2, K)
    a2, w2 = a * scale[None, :], w / scale[None, :].clamp_min(1e-3)
    ref2 = a2.double() @ w2.double().t()
    err2 = rel_err(pkg.ops.gemm(a2.to(DEV), w2.to(DEV), None, mode="f16x3"), ref2.float())
    assert err2 < 2e-5, err2
    assert rel_err(pkg.ops.gemm(a2.to(DEV), w2.to(DEV), None, mode="bf16x6"), ref2.float()) < 5e-6
    # (2) uniformly tiny / huge weight matrices: the raw kernel shows the absolute floor 2^-25 per element (relative
    # error ~ 2^-25 / 1e-6) or overflows; the op does not
    for factor in (1e-2, 1e-4, 1e-30, 1e6):
        w3 = (w.double() * factor).float()
        ref3 = a.double() @ w3.double().t()
        err3 = rel_err(pkg.ops.gemm(a.to(DEV), w3.to(DEV), None, mode="f16x3"), ref3.float())
        assert err3 < 2e-6, (factor, err3)
        assert rel_err(pkg.ops.gemm(a.to(DEV), w3.to(DEV), None, mode="bf16x6"), ref3.float()) < 5e-6
    w3 = w * 1e-4
    err_raw = rel_err(pkg.ops.gemm(a.to(DEV), w3.to(DEV), None, mode="f16x3", prescale=False), (a.double() @ w3.double().t()).float())
    assert 1e-4 < err_raw < 2e-2, err_raw
    assert not torch.isfinite(pkg.ops.gemm(a.to(DEV), (w * 1e7).to(DEV), None, mode="f16x3", prescale=False)).any()
    # the fused GEMM + LayerNorm finish takes the same scaled weight
    res, ga, be = torch.randn(64, N, generator=g(5)), 1 + 0.1 * torch.randn(N, generator=g(6)), torch.randn(N, generator=g(7))
    w5 = w * 1e-4
    want5 = F.layer_norm(a[:64].double() @ w5.double().t() + res.double(), (N,), ga.double(), be.double(), 1e-12).float()
    got5 = pkg.ops.linear_residual_layernorm(a[:64].to(DEV), w5.to(DEV), None, res.to(DEV), ga.to(DEV), be.to(DEV), 1e-12, mode="f16x3")
    assert rel_err(got5, want5) < 3e-6
    # (3) out-of-range ACTIVATIONS are loud
    a4 = a.clone()
    a4[3, 17] = 1.0e5
    got4 = pkg.ops.gemm(a4.to(DEV), w.to(DEV), None, mode="f16x3")
    assert not torch.isfinite(got4[3]).all() and torch.isfinite(got4[4:]).all()
    assert torch.isfinite(pkg.ops.gemm(a4.to(DEV), w.to(DEV), None, mode="bf16x6")).all()


def test_f16_weight_cache_follows_weight_updates(pkg, hip):
    """The scaled copy is keyed like every derived-weight cache: an in-place update (load_state_dict, optimizer step)
    rebuilds it."""
    w = (torch.randn(128, 64, generator=g(1)) * 0.03).to(DEV)
    a = torch.randn(40, 64, generator=g(2)).to(DEV)
    s1, inv1 = pkg.ops.f16_weight(w)
    assert 2 ** 11 <= float(s1.abs().max()) < 2 ** 12 and torch.equal(s1 * inv1, w)
    assert pkg.ops.f16_weight(w)[0] is s1
    y1 = pkg.ops.gemm(a, w, None, mode="f16x3")
    w.mul_(1024.0)
    s2, inv2 = pkg.ops.f16_weight(w)
    assert s2 is not s1 and inv2 == inv1 * 1024 and torch.equal(s2 * inv2, w)
    assert torch.equal(pkg.ops.gemm(a, w, None, mode="f16x3"), y1 * 1024.0)     # same scaled operand, exact epilogue scale


@pytest.mark.parametrize("M,N,K", [(1, 128, 32), (64, 768, 768), (64, 2304, 768), (33, 768, 1024), (128, 4608, 768),
                                   (100, 1024, 3072), (64, 768, 48), (512, 768, 768), (1000, 768, 1024), (256, 2304, 768)])
@pytest.mark.parametrize("act", [0, 1, 2])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 3e-5), ("f16x3", 5e-6)])
def test_gemm_skinny_split_k(pkg, hip, M, N, K, act, mode, tol):
    """The small-M kernels (one wave per 32x32 tile and K slice, slices summed in slice order by a second launch):
    against fp64; run-to-run bit-identical; a strided activation view; rows past M; launches of different shapes share
    one workspace back to back (nothing of an earlier shape's slabs may show)."""
    assert pkg.ops._skinny_ok(pkg.ops.GEMM_MODES[mode], M, N, K, torch.empty(1, K))   # the shape takes the skinny path
    wide = torch.randn(M, K + 32, generator=g(M + N))
    a = wide[:, 16:16 + K]
    w = torch.randn(N, K, generator=g(K)) / math.sqrt(K)
    b = torch.randn(N, generator=g(7))
    ref = F.linear(a.double(), w.double(), b.double())
    ref = {0: lambda x: x, 1: F.gelu, 2: F.silu}[act](ref).float()
    ad, wd, bd = wide.to(DEV)[:, 16:16 + K], w.to(DEV), b.to(DEV)
    outs = [pkg.ops.gemm(ad, wd, bd, act, mode=mode).clone() for _ in range(4)]
    assert rel_err(outs[0], ref) < tol
    assert all(torch.equal(o, outs[0]) for o in outs[1:])
    # same maths as the tiled kernel (other summation order): fp32 rounding apart
    prev = pkg.ops.SKINNY_MAX_M
    pkg.ops.SKINNY_MAX_M = 0
    try:
        tiled = pkg.ops.gemm(ad, wd, bd, act, mode=mode) if N % 128 == 0 and K % 32 == 0 else None
    finally:
        pkg.ops.SKINNY_MAX_M = prev
    if tiled is not None:
        assert rel_err(outs[0], tiled) < 2e-6


@pytest.mark.parametrize("M,H,K", [(64, 768, 768), (64, 768, 1024), (33, 256, 512), (128, 1024, 3072), (5, 512, 64),
                                   (512, 768, 1024), (1024, 768, 768)])
@pytest.mark.parametrize("mode,tol", [("bf16x3", 3e-5), ("f16x3", 5e-6)])
def test_gemm_skinny_residual_layernorm(pkg, hip, M, H, K, mode, tol):
    """BertSelfOutput / BertOutput in one call (dense -> + residual -> LayerNorm): bit-identical to the unfused pair
    (skinny GEMM, then the LayerNorm kernel), against fp64, with and without a residual."""
    a = torch.randn(M, K, generator=g(M + K))
    w = torch.randn(H, K, generator=g(H)) / math.sqrt(K)
    b, res = torch.randn(H, generator=g(1)), torch.randn(M, H, generator=g(2))
    gamma, beta = torch.rand(H, generator=g(3)) + 0.5, torch.randn(H, generator=g(4))
    ad, wd, bd, rd, gd, be = (t.to(DEV) for t in (a, w, b, res, gamma, beta))
    for r_cpu, r_dev in ((res, rd), (None, None)):
        pre = F.linear(a.double(), w.double(), b.double()) + (0 if r_cpu is None else r_cpu.double())
        ref = F.layer_norm(pre, (H,), gamma.double(), beta.double(), 1e-12).float()
        fused = pkg.ops.linear_residual_layernorm(ad, wd, bd, r_dev, gd, be, 1e-12, mode=mode)
        pair = pkg.ops.residual_layernorm(pkg.ops.gemm(ad, wd, bd, mode=mode), r_dev, gd, be, 1e-12)
        assert torch.equal(fused, pair)
        assert rel_err(fused, ref) < 4 * tol
    # larger M: the same entry point falls back to the two ops
    big = torch.randn(2048, K, device=DEV)
    assert torch.equal(pkg.ops.linear_residual_layernorm(big, wd, bd, None, gd, be, 1e-12, mode=mode),
                       pkg.ops.residual_layernorm(pkg.ops.gemm(big, wd, bd, mode=mode), None, gd, be, 1e-12))


def test_gemm_skinny_under_load_every_word(pkg, hip):
    """The small-M kernels while a large GEMM on a second stream keeps the CUs busy: many launches back to back on
    fresh data, every output word compared."""
    M, N, K = 64, 2304, 768
    big_a = torch.randn(16384, 768, device=DEV)
    big_w = torch.randn(768, 768, device=DEV) / 27.7
    side = torch.cuda.Stream()
    w = (torch.randn(N, K, generator=g(3)) / math.sqrt(K)).to(DEV)
    b = torch.randn(N, generator=g(4)).to(DEV)
    bad = 0
    for it in range(60):
        a = torch.randn(M, K, device=DEV)
        with torch.cuda.stream(side):
            pkg.ops.gemm(big_a, big_w, None, mode="bf16x3")
        got = pkg.ops.gemm(a, w, b, mode="f16x3")
        ref = torch.addmm(b.double(), a.double(), w.double().t()).float()
        bad += int(((got - ref).abs() > 2e-5 * ref.abs().max()).sum())
    torch.cuda.synchronize()
    assert bad == 0, bad


# ------------------------------------------------------------------------------------- row-complete GEMM + LayerNorm
@pytest.mark.parametrize("M,K", [(32, 768), (96, 64), (224, 1024), (4096, 768), (8192 + 160, 1024), (65536, 768)])
@pytest.mark.parametrize("mode,tol", [("f16x3", 5e-6), ("bf16x3", 3e-5)])
def test_gemm_rowln_matches_the_unfused_pair(pkg, hip, monkeypatch, M, K, mode, tol):
    """BertSelfOutput / BertOutput as one row-complete launch (e3d_gemm_residual_layernorm_f32_split: LDS-DMA staging,
    pre-split weight planes, residual through an LDS ring, LayerNorm in the accumulators) against the unfused pair it
    replaces (<= 1e-6 of the output scale: same products in the same order, the row statistics summed in another order)
    and against fp64; every tile form (96 / 64 / 32 rows, several tiles per workgroup), with and without a residual,
    a row-strided A."""
    monkeypatch.setattr(pkg.ops, "ROWLN_MIN_M", 1)
    H = 768
    a_full = torch.randn(M, K + 32, generator=g(M + K))
    a = a_full[:, :K]                                      # row stride K + 32
    w = torch.randn(H, K, generator=g(H)) / math.sqrt(K)
    b, res = torch.randn(H, generator=g(1)), torch.randn(M, H, generator=g(2)) * 3 + 0.5
    gamma, beta = torch.rand(H, generator=g(3)) + 0.5, torch.randn(H, generator=g(4))
    ad, wd, bd, rd, gd, be = (t.to(DEV) for t in (a_full, w, b, res, gamma, beta))
    ad = ad[:, :K]
    rows = torch.randperm(M, generator=g(5))[:2048]        # fp64 check on a sample of rows (the pair check covers all)
    for r_cpu, r_dev in ((res, rd), (None, None)):
        fused = pkg.ops.linear_residual_layernorm(ad, wd, bd, r_dev, gd, be, 1e-12, mode=mode)
        pair = pkg.ops.residual_layernorm(pkg.ops.gemm(ad, wd, bd, mode=mode), r_dev, gd, be, 1e-12)
        assert torch.isfinite(fused).all()
        assert float((fused - pair).abs().max()) <= 1e-6 * float(pair.abs().max()), float((fused - pair).abs().max())
        pre = F.linear(a[rows].double(), w.double(), b.double()) + (0 if r_cpu is None else r_cpu[rows].double())
        ref = F.layer_norm(pre, (H,), gamma.double(), beta.double(), 1e-12).float()
        assert rel_err(fused[rows.to(DEV)], ref) < 4 * tol


def test_gemm_rowln_dispatch_and_weight_updates(pkg, hip, monkeypatch):
    """The fused path is taken from ROWLN_MIN_M rows upwards only, follows in-place weight updates (planes are keyed like
    every derived-weight cache) and leaves the other arithmetic modes on the unfused pair."""
    H, K, M = 768, 768, 512
    a = torch.randn(M, K, device=DEV)
    w = (torch.randn(H, K, device=DEV) / math.sqrt(K)).contiguous()
    b, res = torch.randn(H, device=DEV), torch.randn(M, H, device=DEV)
    gamma, beta = torch.rand(H, device=DEV) + 0.5, torch.randn(H, device=DEV)
    small = pkg.ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode="f16x3")
    assert getattr(w, "_e3d_planes_w", None) is None           # default threshold: M = 512 stays on the skinny / pair path
    monkeypatch.setattr(pkg.ops, "ROWLN_MIN_M", 1)
    fused = pkg.ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode="f16x3")
    assert w._e3d_planes_w is not None and float((fused - small).abs().max()) < 2e-6 * float(small.abs().max())
    w.mul_(0.5)
    pkg.ops.invalidate_weight_caches()
    again = pkg.ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode="f16x3")
    pair = pkg.ops.residual_layernorm(pkg.ops.gemm(a, w, b, mode="f16x3"), res, gamma, beta, 1e-12)
    assert float((again - pair).abs().max()) < 2e-6 * float(pair.abs().max())
    for mode in ("f32", "bf16x6"):
        got = pkg.ops.linear_residual_layernorm(a, w, b, res, gamma, beta, 1e-12, mode=mode)
        assert torch.equal(got, pkg.ops.residual_layernorm(pkg.ops.gemm(a, w, b, mode=mode), res, gamma, beta, 1e-12))
