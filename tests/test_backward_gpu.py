"""GPU: hand-written backward kernels (through the C-ABI and the autograd shells) against CPU
torch.autograd of the oracle / of plain fp64 statements.  Gradients: 1e-4 relative (max-norm) in
the fp32-grade GEMM mode, looser where the default bf16x3 GEMM enters (stated per test)."""
import math

import pytest
import torch
import torch.nn.functional as TF

from helpers import rel_err, seeded_state_dict, synthetic_pockets
from oracle import bert as obert, sequence as oseq, structure as ostr

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def g(seed):
    return torch.Generator().manual_seed(seed)


def leaf(t, dev=None, dtype=None):
    t = t.detach().clone()
    if dtype is not None:
        t = t.to(dtype)
    if dev is not None:
        t = t.to(dev)
    return t.requires_grad_(True)


@pytest.fixture()
def Fm(pkg):
    from e3diff_amd.autograd import functional
    return functional


@pytest.mark.parametrize("mode,tol", [("bf16x6", 1e-5), ("bf16x3", 1e-4), ("f32", 1e-5), ("bf16", 2e-2)])   # bf16: plain bf16 products, the reference's training precision (opt-in)
@pytest.mark.parametrize("M,N,K,act", [(96, 256, 128, 0), (300, 768, 1024, 1), (64, 1536, 256, 2), (4096, 768, 768, 0)])
def test_linear_backward(pkg, hip, Fm, M, N, K, act, mode, tol):
    x, w, b = torch.randn(M, K, generator=g(1)), torch.randn(N, K, generator=g(2)) / math.sqrt(K), torch.randn(N, generator=g(3))
    go = torch.randn(M, N, generator=g(4))
    xr, wr, br = leaf(x, dtype=torch.double), leaf(w, dtype=torch.double), leaf(b, dtype=torch.double)
    y = TF.linear(xr, wr, br)
    y = {0: lambda t: t, 1: TF.gelu, 2: TF.silu}[act](y)
    y.backward(go.double())
    prev = pkg.ops.set_gemm_mode(mode)
    try:
        xd, wd, bd = leaf(x, DEV), leaf(w, DEV), leaf(b, DEV)
        out = Fm.linear(xd, wd, bd, act)
        out.backward(go.to(DEV))
    finally:
        pkg.ops.set_gemm_mode(prev)
    assert rel_err(out, y.float()) < tol
    assert rel_err(xd.grad, xr.grad.float()) < tol
    assert rel_err(wd.grad, wr.grad.float()) < tol
    assert rel_err(bd.grad, br.grad.float()) < (1e-5 if mode != "bf16" else tol)   # (through act'(z): z carries the products' error)


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("bf16x6", 1e-5), ("bf16", 2e-2)])
@pytest.mark.parametrize("N,K,M,count", [(768, 768, 4096, 5), (300, 160, 1000, 3), (2304, 768, 515, 2), (64, 1024, 96, 64),
                                         (302, 162, 1000, 3)])
def test_grouped_weight_gradients(pkg, hip, N, K, M, count, mode, tol):
    """e3d_gemm_wgrad_grouped_f32_split: dW_p = dz_p^T x_p and db_p = column sums of dz_p for ``count`` layers of one
    shape in one launch, against fp64: ragged tiles (N, K not multiples of the 256x128 tile), a token count that is
    not a multiple of the k-step, dz given as a column block of a wider matrix (packed QKV), the accumulate bit,
    problems without a bias, and run-to-run bit-identity (no atomics).  Shapes in whole quads (N, K, both row strides
    multiples of 4: every layer of the models) take the transposing staging (float4 loads, ds_read_b64_tr_b16 fragments);
    the last case (N = 302) the dword staging."""
    import ctypes
    lib = pkg.hip.lib()
    terms = pkg.ops.GEMM_MODES[mode]
    wide = [torch.randn(M, N + 64, generator=g(10 + p)).to(DEV) for p in range(count)]
    dz = [w[:, 32:32 + N] for w in wide]
    x = [torch.randn(M, K, generator=g(100 + p)).to(DEV) for p in range(count)]
    dw = [torch.full((N, K), float(p), device=DEV) for p in range(count)]     # old contents: kept only where the bit is set
    db = [torch.full((N,), -float(p), device=DEV) for p in range(count)]
    bits = sum(1 << p for p in range(count) if p % 3 == 1)

    def run(with_bias=True):
        arr = lambda ts: (ctypes.c_void_p * count)(*[t.data_ptr() for t in ts])   # noqa: E731
        pkg.hip.check(lib.e3d_gemm_wgrad_grouped_f32_split(arr(dz), arr(x), arr(dw), arr(db) if with_bias else None, bits,
                                                           count, dz[0].stride(0), x[0].stride(0), N, K, M, terms,
                                                           torch.cuda.current_stream().cuda_stream), "grouped wgrad")
    run()
    first = [t.clone() for t in dw + db]
    for p in range(count):
        want_w = dz[p].double().t() @ x[p].double()
        want_b = dz[p].double().sum(0)
        if bits >> p & 1:
            want_w, want_b = want_w + p, want_b - p
        assert rel_err(dw[p], want_w.float()) < tol, p
        assert rel_err(db[p], want_b.float()) < 1e-5, p
    for t, p in zip(dw + db, list(range(count)) * 2):       # same launch again on the same inputs: bit-identical
        t.fill_(float(p) if t.dim() == 2 else -float(p))
    run()
    assert all(torch.equal(a, b) for a, b in zip(first, dw + db))
    keep = [t.clone() for t in db]
    run(with_bias=False)
    assert all(torch.equal(a, b) for a, b in zip(keep, db))


@pytest.mark.parametrize("mode,tol", [("bf16x3", 1e-4), ("bf16x6", 1e-5)])
def test_ragged_weight_gradients_of_mixed_shapes_in_one_launch(pkg, hip, mode, tol):
    """e3d_gemm_wgrad_ragged_f32_split: layers of different shapes and row strides (one token count) in ONE launch -- a
    packed-QKV column block (stride 2304), square and rectangular weights, a ragged one (N, K not multiples of the tile
    nor of 4: the launch then takes the dword staging), problems without a bias, the accumulate bit -- against fp64, and
    bit-identical between two launches."""
    import ctypes
    lib = pkg.hip.lib()
    terms = pkg.ops.GEMM_MODES[mode]
    for M, shapes in ((1000, [(768, 768, 2304), (768, 768, 768), (1024, 768, 1024), (768, 1024, 768), (1536, 768, 1536), (64, 96, 64)]),
                      (515, [(300, 160, 364), (768, 768, 768), (302, 162, 302)])):
        count = len(shapes)
        wide = [torch.randn(M, ld, generator=g(10 + p)).to(DEV) for p, (N, K, ld) in enumerate(shapes)]
        dz = [w[:, :N] for w, (N, K, ld) in zip(wide, shapes)]
        x = [torch.randn(M, K, generator=g(100 + p)).to(DEV) for p, (N, K, ld) in enumerate(shapes)]
        bits = sum(1 << p for p in range(count) if p % 3 == 1)
        no_bias = {2}
        outs = []
        for rep in range(2):
            dw = [torch.full((N, K), float(p), device=DEV) for p, (N, K, ld) in enumerate(shapes)]
            db = [torch.full((N,), -float(p), device=DEV) for p, (N, K, ld) in enumerate(shapes)]
            arr = lambda ts: (ctypes.c_void_p * count)(*[t.data_ptr() for t in ts])   # noqa: E731
            a_db = (ctypes.c_void_p * count)(*[None if p in no_bias else db[p].data_ptr() for p in range(count)])
            ints = lambda vals, ty: (ty * count)(*vals)   # noqa: E731
            pkg.hip.check(lib.e3d_gemm_wgrad_ragged_f32_split(
                arr(dz), arr(x), arr(dw), a_db, ints([s_[0] for s_ in shapes], ctypes.c_int), ints([s_[1] for s_ in shapes], ctypes.c_int),
                ints([t.stride(0) for t in dz], ctypes.c_int64), ints([t.stride(0) for t in x], ctypes.c_int64), bits, count, M, terms,
                torch.cuda.current_stream().cuda_stream), "ragged wgrad")
            outs.append(dw + db)
        assert all(torch.equal(a, b) for a, b in zip(*outs))
        for p, (N, K, ld) in enumerate(shapes):
            want_w = dz[p].double().t() @ x[p].double()
            want_b = dz[p].double().sum(0)
            if bits >> p & 1:
                want_w, want_b = want_w + p, want_b - p
            assert rel_err(outs[0][p], want_w.float()) < tol, (M, p)
            if p in no_bias:
                assert torch.equal(outs[0][count + p], torch.full((N,), -float(p), device=DEV))
            else:
                assert rel_err(outs[0][count + p], want_b.float()) < 1e-5, (M, p)


@pytest.mark.parametrize("pattern", ["ligand", "pocket", "all_dead", "first_dead", "columns"])
def test_weight_gradients_of_batches_whose_padded_gradient_rows_are_zero(pkg, hip, pattern):
    """The weight-gradient launches on dz as a training step produces it -- rows of padded positions exactly zero, whole
    32-token k-tiles of them (3 of 4 in a ligand frame), leading ones, all of them, or zero only in one tile row of output
    workgroups -- against fp64, the grouped (ragged) launch and the per-layer split-K launch.  (A form of the kernel body
    that skips the MFMAs of such k-tiles was built and measured in round 4: -11 % at 3/4 dead tiles, +6..15 % on live ones,
    nothing on the step -- the k-step of this layout is bound by its staging, profiles/r04_wgrad_dead_tile_skip_ab.log.)"""
    import ctypes
    lib = pkg.hip.lib()
    terms = pkg.ops.GEMM_MODES["bf16x3"]
    B, L = 6, 128
    M = B * L
    shapes = [(768, 768, 2304), (1024, 768, 1024), (300, 160, 304)]
    count = len(shapes)
    tok = torch.arange(M)
    if pattern == "ligand":
        lens = torch.tensor([5, 30, 17, 31, 1, 32])
    elif pattern == "pocket":
        lens = torch.tensor([20, 128, 64, 65, 97, 33])
    elif pattern == "all_dead":
        lens = torch.zeros(B, dtype=torch.long)
    elif pattern == "first_dead":
        lens = torch.tensor([0, 0, 128, 0, 40, 0])
    else:
        lens = torch.full((B,), L)
    valid = ((tok % L) < lens[tok // L]).float()[:, None]
    wide = [torch.randn(M, ld, generator=g(10 + p)) * valid for p, (N, K, ld) in enumerate(shapes)]
    if pattern == "columns":
        wide[0][:, :256] *= ((tok % L) < 20).float()[:, None]
    x = [torch.randn(M, K, generator=g(100 + p)).to(DEV) for p, (N, K, ld) in enumerate(shapes)]
    dz = [w.to(DEV)[:, :N] for w, (N, K, ld) in zip(wide, shapes)]
    dw = [torch.full((N, K), 7.0, device=DEV) for (N, K, ld) in shapes]
    db = [torch.full((N,), 7.0, device=DEV) for (N, K, ld) in shapes]
    arr = lambda ts: (ctypes.c_void_p * count)(*[t.data_ptr() for t in ts])   # noqa: E731
    ints = lambda vals, ty: (ty * count)(*vals)   # noqa: E731
    pkg.hip.check(lib.e3d_gemm_wgrad_ragged_f32_split(
        arr(dz), arr(x), arr(dw), arr(db), ints([s_[0] for s_ in shapes], ctypes.c_int), ints([s_[1] for s_ in shapes], ctypes.c_int),
        ints([t.stride(0) for t in dz], ctypes.c_int64), ints([t.stride(0) for t in x], ctypes.c_int64), 0, count, M, terms,
        torch.cuda.current_stream().cuda_stream), "ragged wgrad")
    from e3diff_amd.autograd import gemm_general
    for p, (N, K, ld) in enumerate(shapes):
        want = wide[p][:, :N].double().t() @ x[p].cpu().double()
        bound = 1e-4 * float(want.abs().max())          # (all_dead: exact zeros asked for)
        assert float((dw[p].cpu().double() - want).abs().max()) <= bound, (pattern, p)
        want_b = wide[p][:, :N].double().sum(0)
        assert float((db[p].cpu().double() - want_b).abs().max()) <= 1e-5 * float(want_b.abs().max()), (pattern, p)
        if ld % 4 == 0 and N % 4 == 0:
            got = gemm_general(dz[p], True, x[p], True, N, K, M, mode="bf16x3")
            assert float((got.cpu().double() - want).abs().max()) <= bound, (pattern, p)


def test_gemm_general_odd_reduction_and_strided(pkg, hip):
    from e3diff_amd.autograd import gemm_general
    M, N, K = 70, 200, 45          # K-major operands: any K, any N
    a, b = torch.randn(K, M, generator=g(1)), torch.randn(K, 300, generator=g(2))
    got = gemm_general(a.to(DEV), True, b.to(DEV)[:, 50:250], True, M, N, K, mode="bf16x6")
    assert rel_err(got, (a.t().double() @ b[:, 50:250].double()).float()) < 1e-5


@pytest.mark.parametrize("H,M", [(768, 70), (256, 9), (1024, 33), (768, 96), (512, 160), (768, 4096), (768, 8200)])
def test_layernorm_and_adaln_backward(pkg, hip, Fm, H, M):
    x, r = torch.randn(M, H, generator=g(1)) * 2, torch.randn(M, H, generator=g(2))
    ga, be = 1 + 0.1 * torch.randn(H, generator=g(3)), torch.randn(H, generator=g(4))
    go = torch.randn(M, H, generator=g(5))
    xr, rr, gr, br = (leaf(t, dtype=torch.double) for t in (x, r, ga, be))
    TF.layer_norm(xr + rr, (H,), gr, br, 1e-12).backward(go.double())
    xd, rd, gd, bd = (leaf(t, DEV) for t in (x, r, ga, be))
    Fm.residual_layernorm(xd, rd, gd, bd, 1e-12).backward(go.to(DEV))
    for a, b in ((xd, xr), (rd, rr), (gd, gr), (bd, br)):
        assert rel_err(a.grad, b.grad.float()) < 1e-5
    # the parameter gradients are per-block partial sums added up in a fixed order (e3d_layernorm_bwd_ws): run-to-run identical
    xd1, rd1, gd1, bd1 = (leaf(t, DEV) for t in (x, r, ga, be))
    Fm.residual_layernorm(xd1, rd1, gd1, bd1, 1e-12).backward(go.to(DEV))
    assert torch.equal(gd1.grad, gd.grad) and torch.equal(bd1.grad, bd.grad)
    # no residual
    xd2, gd2, bd2 = leaf(x, DEV), leaf(ga, DEV), leaf(be, DEV)
    Fm.residual_layernorm(xd2, None, gd2, bd2, 1e-12).backward(go.to(DEV))
    xr2 = leaf(x, dtype=torch.double)
    TF.layer_norm(xr2, (H,), ga.double(), be.double(), 1e-12).backward(go.double())
    assert rel_err(xd2.grad, xr2.grad.float()) < 1e-5
    # adaLN gate, both broadcast modes and both branches (M % 32 == 0: also two conditioning rows of M/2 rows each --
    # multiples of 16 take the kernel that sums the modulation gradients per workgroup before its atomics)
    for rpc in (1, M) + ((M // 2,) if M % 32 == 0 else ()):
        for branch in (0, 1):
            y = torch.randn(M, H, generator=g(6)) * 2
            mod = torch.randn(M // rpc, 6 * H, generator=g(7))
            xr3, yr3, mr3 = (leaf(t, dtype=torch.double) for t in (x, y, mod))
            sh, sc, gt = [mr3[:, (3 * branch + i) * H:(3 * branch + i + 1) * H].repeat_interleave(rpc, 0) for i in range(3)]
            (xr3 + gt * (TF.layer_norm(yr3, (H,)) * (1 + sc) + sh)).backward(go.double())
            xd3, yd3, md3 = (leaf(t, DEV) for t in (x, y, mod))
            Fm.adaln_gate(xd3, yd3, md3, branch, rpc).backward(go.to(DEV))
            for a, b in ((xd3, xr3), (yd3, yr3), (md3, mr3)):
                assert rel_err(a.grad, b.grad.float()) < 1e-5, (rpc, branch)


def test_embed_and_head_backward(pkg, hip, Fm):
    M, L, H = 48, 16, 768
    for Fin in (8, 20):
        x = torch.randn(M, Fin, generator=g(1))
        w, b = torch.randn(H, Fin, generator=g(2)), torch.randn(H, generator=g(3))
        ga, be = 1 + 0.1 * torch.randn(H, generator=g(4)), torch.randn(H, generator=g(5))
        add = torch.randn(M // L, H, generator=g(6))
        go = torch.randn(M, H, generator=g(7))
        ref = [leaf(t, dtype=torch.double) for t in (w, b, ga, be, add)]
        (TF.layer_norm(TF.linear(x.double(), ref[0], ref[1]), (H,), ref[2], ref[3], 1e-12)
         + ref[4].repeat_interleave(L, 0)).backward(go.double())
        dev = [leaf(t, DEV) for t in (w, b, ga, be, add)]
        Fm.embed_layernorm(x.to(DEV), dev[0], dev[1], dev[2], dev[3], 1e-12, dev[4], L).backward(go.to(DEV))
        for a, r in zip(dev, ref):
            assert rel_err(a.grad, r.grad.float()) < 1e-5
    for n_out in (8, 20):
        x, w, b = torch.randn(77, H, generator=g(1)), torch.randn(n_out, H, generator=g(2)) / 27, torch.randn(n_out, generator=g(3))
        go = torch.randn(77, n_out, generator=g(4))
        ref = [leaf(t, dtype=torch.double) for t in (x, w, b)]
        TF.linear(*ref).backward(go.double())
        dev = [leaf(t, DEV) for t in (x, w, b)]
        Fm.head_linear(*dev).backward(go.to(DEV))
        for a, r in zip(dev, ref):
            assert rel_err(a.grad, r.grad.float()) < 1e-5


def ref_attention(q, k, v, mask, E, P):
    s = q @ k.transpose(-1, -2)
    if E is not None:
        s = s + obert.relkey_scores_literal(q, E, P)
    s = s / 8.0
    if mask is not None:
        s = s + ((1.0 - mask) * -10000.0)[:, None, None, :]
    return torch.softmax(s, -1) @ v


@pytest.mark.parametrize("B,nh,L,P", [(2, 2, 16, 16), (1, 3, 64, 64), (2, 2, 50, 64), (1, 2, 128, 128), (1, 1, 256, 256)])
@pytest.mark.parametrize("relkey", [True, False])
def test_self_attention_backward(pkg, hip, Fm, B, nh, L, P, relkey):
    H = nh * 64
    qkv = torch.randn(B * L, 3 * H, generator=g(L))
    E = torch.randn(2 * P - 1, 64, generator=g(P)) if relkey else None
    lens = torch.randint(1, L + 1, (B,), generator=g(3))
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    go = torch.randn(B * L, H, generator=g(9))
    qr = leaf(qkv, dtype=torch.double)
    Er = leaf(E, dtype=torch.double) if relkey else None
    sp = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attention(sp(qr[:, :H]), sp(qr[:, H:2 * H]), sp(qr[:, 2 * H:]), mask.double(), Er, P)
    ref.permute(0, 2, 1, 3).reshape(B * L, H).backward(go.double())
    # forward mode feeds out/lse into the (fp32 MFMA) backward: fp32-grade forward -> tight bound,
    # default bf16x3 forward -> the 1e-4 budget
    for mode, tol in (("bf16x6", 2e-5), ("bf16x3", 1e-4)):
        prev = pkg.ops.set_attn_mode(mode)
        try:
            qd = leaf(qkv, DEV)
            Ed = leaf(E, DEV) if relkey else None
            out = Fm.attention(qd, None, B, nh, L, L, key_mask=mask.to(DEV), dist_emb=Ed, max_pos=P)
            out.backward(go.to(DEV))
        finally:
            pkg.ops.set_attn_mode(prev)
        assert rel_err(qd.grad, qr.grad.float()) < tol, mode
        if relkey:
            assert rel_err(Ed.grad, Er.grad.float()) < tol, mode


def test_cross_attention_backward_rectangular(pkg, hip, Fm):
    B, nh, Lq, Lk = 2, 3, 40, 70
    H = nh * 64
    q, kv = torch.randn(B * Lq, H, generator=g(1)), torch.randn(B * Lk, 2 * H, generator=g(2))
    mask = (torch.arange(Lk)[None] < torch.tensor([[70], [33]])).float()
    go = torch.randn(B * Lq, H, generator=g(3))
    qr, kr = leaf(q, dtype=torch.double), leaf(kv, dtype=torch.double)
    sp = lambda x, L: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attention(sp(qr, Lq), sp(kr[:, :H], Lk), sp(kr[:, H:], Lk), mask.double(), None, 0)
    ref.permute(0, 2, 1, 3).reshape(B * Lq, H).backward(go.double())
    for mode, tol in (("bf16x6", 2e-5), ("bf16x3", 1e-4)):    # bf16x3: the fused recomputing kernel (attn_bwd_coop.hip)
        prev = pkg.ops.set_attn_mode(mode)
        try:
            qd, kd = leaf(q, DEV), leaf(kv, DEV)
            Fm.attention(qd, kd, B, nh, Lq, Lk, key_mask=mask.to(DEV)).backward(go.to(DEV))
        finally:
            pkg.ops.set_attn_mode(prev)
        assert rel_err(qd.grad, qr.grad.float()) < tol, mode
        assert rel_err(kd.grad[:, :H], kr.grad[:, :H].float()) < tol, mode
        assert rel_err(kd.grad[:, H:], kr.grad[:, H:].float()) < tol, mode


@pytest.mark.parametrize("B,nh,L", [(3, 2, 128), (2, 3, 96), (2, 1, 33)])
def test_fused_attention_backward_matches_the_two_launch_kernels(pkg, hip, Fm, B, nh, L, monkeypatch):
    """attn_bwd_coop.hip (one launch, P / dS recomputed on chip, operands split once per (item, head)) against the
    two-launch bf16x3 kernels it replaces (P / dS through HBM): same arithmetic class, so the two agree far inside the
    1e-4 budget -- every gradient incl. the distance table, masks with padded tails, L not a multiple of 32."""
    import subprocess, sys, json   # noqa: E401 (the switch is read once per process: the old path runs in a child)
    H, P = nh * 64, L
    qkv = torch.randn(B * L, 3 * H, generator=g(L))
    E = torch.randn(2 * P - 1, 64, generator=g(P + 1))
    lens = torch.randint(1, L + 1, (B,), generator=g(5))
    lens[0] = L
    mask = (torch.arange(L)[None] < lens[:, None]).float()
    go = torch.randn(B * L, H, generator=g(9))
    prev = pkg.ops.set_attn_mode("bf16x3")
    try:
        qd, Ed = leaf(qkv, DEV), leaf(E, DEV)
        Fm.attention(qd, None, B, nh, L, L, key_mask=mask.to(DEV), dist_emb=Ed, max_pos=P).backward(go.to(DEV))
    finally:
        pkg.ops.set_attn_mode(prev)
    qr, Er = leaf(qkv, dtype=torch.double), leaf(E, dtype=torch.double)
    sp = lambda x: x.reshape(B, L, nh, 64).permute(0, 2, 1, 3)  # noqa: E731
    ref = ref_attention(sp(qr[:, :H]), sp(qr[:, H:2 * H]), sp(qr[:, 2 * H:]), mask.double(), Er, P)
    ref.permute(0, 2, 1, 3).reshape(B * L, H).backward(go.double())
    for part, name in ((slice(0, H), "dq"), (slice(H, 2 * H), "dk"), (slice(2 * H, 3 * H), "dv")):
        assert rel_err(qd.grad[:, part], qr.grad[:, part].float()) < 5e-5, name
    assert rel_err(Ed.grad, Er.grad.float()) < 5e-5
    # rows of padded queries / keys get exact zeros where the reference does
    valid = mask.bool().reshape(-1)
    assert float(qd.grad[~valid][:, H:].abs().max() if (~valid).any() else 0.0) < 1e-30


def _grad_compare(model, sd, loss_dev, loss_ref_fn, tol, defer=False):
    """``defer``: the weight / bias gradients of the linear layers computed at the end of the backward pass, grouped by
    shape (autograd.deferred_weight_grads) instead of layer by layer inside it."""
    model.zero_grad(set_to_none=True)
    if defer:
        from e3diff_amd.autograd import deferred_weight_grads
        reported = []
        staged = defer == "staged"         # as under a gradient averager: flushed in 3 slices, parameters reported
        with deferred_weight_grads(on_param=reported.append if staged else None, stages=3) as q:
            loss_dev.backward()
            n_queued = q.queued_total      # (single process: slices of 24 layers already left on the side stream)
            assert n_queued > 10           # the linear layers really took the deferred path
        if staged:
            ids = [id(p) for p in reported]
            assert len(ids) == len(set(ids)) and len(ids) >= n_queued   # weights + biases, each exactly once
            assert all(p.grad is not None for p in reported)
    else:
        loss_dev.backward()
    ref_sd = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    loss_ref = loss_ref_fn(ref_sd)
    loss_ref.backward()
    assert loss_dev.item() == pytest.approx(loss_ref.item(), rel=2e-4 if tol < 1e-2 else 2e-2)   # (plain-bf16 mode: 1e-2 grade)
    worst = {}
    # key biases have an exactly-zero gradient (softmax is invariant to a per-query shift of all
    # scores), so their "reference" gradient is rounding noise: floor every denominator at 1e-3 of
    # the typical gradient magnitude
    scale = torch.stack([v.grad.abs().max() for v in ref_sd.values() if v.grad is not None]).median().item()
    for k, p in model.named_parameters():
        want = ref_sd[k].grad
        if want is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, k
            continue
        assert p.grad is not None, k
        if k.endswith("self.key.bias"):      # mathematically zero: only check that it is noise-sized
            assert float(p.grad.abs().max()) < 1e-2 * scale, k
            continue
        worst[k] =((p.grad.cpu() - want).abs().max() / max(want.abs().max().item(), 1e-3 * scale)).item()
    bad = {k: v for k, v in worst.items() if v > tol}
    assert not bad, sorted(bad.items(), key=lambda kv: -kv[1])[:8]
    return max(worst.values())


@pytest.mark.parametrize("defer", [False, True, "staged"])
@pytest.mark.parametrize("mode,tol", [("bf16x6", 2e-5), ("bf16x3", 2e-4), ("bf16", 1e-1)])   # measured on MI355X: 4.0e-6 / 4.1e-5 / (printed)
def test_structure_training_step_gradients_match_oracle(pkg, hip, mode, tol, defer, capsys):
    """Whole structure model, loss of the reference (wrapped L1 x4 + smooth-L1 x4), every parameter
    gradient against CPU autograd of the oracle."""
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.structure_model.model import ConditionalBertForDiffusion
    L, B = 64, 3
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2,
             max_position_embeddings=L, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = ConditionalBertForDiffusion(
        BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
        loss_func=[ConditionalBertForDiffusion.diheral_loss_func] * 4 + [ConditionalBertForDiffusion.angle_loss_func] * 4)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=12)
    model.load_state_dict(sd)
    model = model.train().to(DEV)
    pk = synthetic_pockets(B, L, seed=5)
    gen = g(1)
    batch = dict(pk, timestep=torch.randint(0, 1000, (B, 1), generator=gen),
                 noised_ligand_angle=ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=gen)),
                 known_noise=ostr.modulo_with_wrapped_range(torch.randn(B, L, 8, generator=gen)))
    dbatch = {k: v.to(DEV) for k, v in batch.items() if torch.is_tensor(v)}
    prev, prev_a = pkg.ops.set_gemm_mode(mode), pkg.ops.set_attn_mode(mode)
    try:
        loss = model.training_step(dbatch)

        def ref_loss(rsd):
            pred = ostr.forward(rsd, {"num_heads": 12, "max_pos": L}, batch["timestep"], batch["noised_ligand_angle"],
                                pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
            return ostr.loss_terms(pred, batch["known_noise"], pk["ligand_attn_mask"]).mean()

        worst = _grad_compare(model, sd, loss, ref_loss, tol, defer)
    finally:
        pkg.ops.set_gemm_mode(prev); pkg.ops.set_attn_mode(prev_a)
    with capsys.disabled():
        print(f"\n[structure grads, {mode}{', grouped weight gradients' if defer else ''}] worst relative gradient error {worst:.2e}")


@pytest.mark.parametrize("defer", [False, True, "staged"])    # deferred: incl. ligand_feature_emb, whose weights are used twice
def test_sequence_training_step_gradients_match_oracle(pkg, hip, defer, capsys):
    from e3diff_amd.bert import BertConfig
    from e3diff_amd.sequence_model.model import PeptideDiff
    L, B = 64, 3
    c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=2,
             max_position_embeddings=L, hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
    model = PeptideDiff(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True),
                        feature_names=list("ACDEFGHIKLMNPQRSTVWY"), loss_func=torch.nn.CrossEntropyLoss(),
                        noise_schedule="cosine", timesteps=50)
    sd = seeded_state_dict({k: tuple(v.shape) for k, v in model.state_dict().items()}, seed=13)
    model.load_state_dict(sd)
    model = model.train().to(DEV)
    pk = synthetic_pockets(B, L, seed=6, with_ligand_seq=True)
    t_int = torch.tensor([[5.0], [30.0], [48.0]])
    noised = oseq.apply_aa_noise(pk["ligand_seq"], t_int, oseq.NoiseScheduleDiscrete(50),
                                 oseq.BlosumTransition(torch.load(__import__("os").path.join(
                                     __import__("helpers").GOLDEN, "blosum_substitute.pt"), weights_only=True)),
                                 u=torch.rand(B * L, generator=g(2)))
    dpk = {k: v.to(DEV) for k, v in pk.items() if torch.is_tensor(v)}
    prev, prev_a = pkg.ops.set_gemm_mode("bf16x6"), pkg.ops.set_attn_mode("bf16x6")
    try:
        loss = model.get_loss(dpk, (t_int / 50).to(DEV), noised.to(DEV))[0]

        def ref_loss(rsd):
            pred = oseq.forward(rsd, {"num_heads": 12, "max_pos": L}, t_int / 50, noised, pk["ligand_angles"],
                                pk["ligand_attn_mask"], pk["receptor_seq"], pk["receptor_angles"], pk["receptor_attn_mask"])
            return oseq.get_loss(pred, pk, noised)[0]

        worst = _grad_compare(model, sd, loss, ref_loss, 2e-4, defer)
    finally:
        pkg.ops.set_gemm_mode(prev); pkg.ops.set_attn_mode(prev_a)
    # parameters the forward never touches keep no gradient (reference quirk)
    assert all(p.grad is None for n, p in model.named_parameters() if n.startswith("receptor_feature_emb."))
    with capsys.disabled():
        print(f"\n[sequence grads, bf16x6] worst relative gradient error {worst:.2e}")


def test_transposed_weight_registry_refreshes_all_stale_weights_together(pkg, hip):
    """autograd._transposed_weight: weights (and packed query / key / value weights through their parts) are registered
    on first use; after an optimizer step the first request re-transposes EVERY registered weight (grouped launches),
    the others then hit the cache; values always equal W^T."""
    from e3diff_amd import autograd as AG, ops
    ws = [torch.nn.Parameter(torch.randn(n, k, generator=g(i)).to(DEV)) for i, (n, k) in
          enumerate([(768, 256), (768, 256), (100, 70), (768, 256), (96, 1024)])]
    q, k_, v = (torch.nn.Parameter(torch.randn(128, 64, generator=g(20 + i)).to(DEV)) for i in range(3))
    packed = torch.cat([q, k_, v], 0)
    packed._e3d_parts = [q, k_, v]
    got = [AG._transposed_weight(w) for w in ws] + [AG._transposed_weight(packed)]
    for w, t in zip(ws + [packed], got):
        assert torch.equal(t, w.detach().t())
    assert all(AG._transposed_weight(w) is t for w, t in zip(ws, got))          # cached
    with torch.no_grad():
        for p in ws + [q, k_, v]:
            p.mul_(1.5)                                                         # in-place update: version bump
    ops.invalidate_weight_caches()
    first = AG._transposed_weight(ws[2])                                        # one request ...
    assert first is got[2] and torch.equal(first, ws[2].detach().t())
    keys_now = {id(w): AG._WT_REGISTRY[(id(w),)].key for w in ws}
    assert all(keys_now[id(w)] == ops.weight_key(w) for w in ws)                # ... refreshed them all
    packed2 = torch.cat([q, k_, v], 0)                                          # a new packed tensor every step
    packed2._e3d_parts = [q, k_, v]
    t2 = AG._transposed_weight(packed2)
    assert t2 is got[-1] and torch.equal(t2, packed2.detach().t())
    for w, t in zip(ws, got):
        assert torch.equal(AG._transposed_weight(w), w.detach().t())


def test_purged_transposes_are_parked_while_another_captured_step_may_use_them(pkg, hip):
    """a training step about to be captured drops the W^T entries of every OTHER model from the registry
    (autograd.forget_transposes) -- but another model's captured step may still be alive, and its graph baked both the refresh
    launches that write those buffers and the input-gradient GEMMs that read them.  The dropped entries are therefore parked
    (their buffers stay allocated, at the same addresses, with the same contents) until release_retired_transposes()."""
    from e3diff_amd import autograd as AG
    mine = torch.nn.Parameter(torch.randn(256, 128, generator=g(40)).to(DEV))
    other = torch.nn.Parameter(torch.randn(384, 128, generator=g(41)).to(DEV))
    t_mine, t_other = AG._transposed_weight(mine), AG._transposed_weight(other)
    ptr_other, parked_before = t_other.data_ptr(), len(AG._WT_RETIRED)
    AG.forget_transposes({id(mine)})
    assert (id(mine),) in AG._WT_REGISTRY and (id(other),) not in AG._WT_REGISTRY
    assert len(AG._WT_RETIRED) == parked_before + 1 and AG._WT_RETIRED[-1].wt.data_ptr() == ptr_other
    junk = [torch.empty_like(t_other) for _ in range(4)]                 # the allocator must not hand the parked buffer out
    assert all(j.data_ptr() != ptr_other for j in junk)
    assert torch.equal(AG._WT_RETIRED[-1].wt, other.detach().t())
    assert AG._transposed_weight(mine) is t_mine                          # the keeper is untouched
    AG.release_retired_transposes()
    assert AG._WT_RETIRED == []


def test_f16x3_weight_cache_miss_inside_a_capture_is_loud(pkg, hip):
    """building the pre-scaled f16x3 copy of a weight reads max |w| back to the host, which a stream capture
    cannot do; it used to fall back silently to the RAW weight (a graph captured that way would run f16x3 on un-pre-scaled
    weights at every replay).  Now it raises; warmed up, the same call inside a capture hits the cache."""
    from e3diff_amd import ops
    w = torch.randn(64, 64, generator=g(42)).to(DEV) * 1e-4
    graph = torch.cuda.CUDAGraph()
    with pytest.raises(RuntimeError, match="pre-scaled copy"):
        with torch.cuda.graph(graph):
            ops.f16_weight(w)
    torch.cuda.synchronize()
    scaled, inv = ops.f16_weight(w)                                       # eager: fills the cache
    assert inv != 1.0 and torch.equal(scaled * inv, w)
    graph2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph2):
        again, inv2 = ops.f16_weight(w)
    assert again is scaled and inv2 == inv


def test_deferred_weight_gradient_queue_flushes_early_past_its_memory_budget(pkg, hip, Fm, monkeypatch):
    """the queue keeps every (dz, x) pair alive until the end of backward.  Past E3D_DEFER_WGRAD_MAX_GB it
    computes what is queued and goes on -- same gradients, every parameter reported to a listener exactly once and only
    after its last contribution (a weight used twice must not be reported by the early flush)."""
    from e3diff_amd.autograd import deferred_weight_grads
    torch.manual_seed(0)
    lin = [torch.nn.Linear(256, 256).to(DEV) for _ in range(3)]
    x = torch.randn(512, 256, device=DEV)

    def loss():
        h = Fm.linear(x, lin[0].weight, lin[0].bias)
        h = Fm.linear(h, lin[1].weight, lin[1].bias)
        h = Fm.linear(h, lin[0].weight, lin[0].bias)      # lin[0] twice
        return Fm.linear(h, lin[2].weight, lin[2].bias).square().mean()

    def grads(budget_bytes):
        for m in lin:
            m.zero_grad(set_to_none=True)
        monkeypatch.setattr(deferred_weight_grads, "MAX_BYTES", budget_bytes)
        reported = []
        with pkg.ops.arithmetic("bf16x6", respect_env=False), deferred_weight_grads(on_param=reported.append) as q:
            loss().backward()
        return [p.grad.clone() for m in lin for p in m.parameters()], reported, q.early_flushes

    ref, rep0, n0 = grads(1 << 40)
    got, rep1, n1 = grads(1 << 20)          # 1 MiB: every second queued layer trips the budget
    assert n0 == 0 and n1 >= 1
    for a, b in zip(got, ref):
        assert rel_err(a, b) < 1e-6
    params = [p for m in lin for p in m.parameters()]
    for rep in (rep0, rep1):
        assert sorted(id(p) for p in rep) == sorted(id(p) for p in params)     # each exactly once
