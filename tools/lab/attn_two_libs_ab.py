#!/usr/bin/env python3
"""Lab: the same attention launches (e3d_relkey_attn_fwd_split_ex, f16x3 = the sampling default) through TWO OR MORE builds of
the library in one process, interleaved (same box, same clocks); outputs compared against the first library's.
    python tools/lab/attn_two_libs_ab.py lab_build/libe3d_prev.so e3-invaraint-diffusion-model_amd/libe3d_hip.so [...]
ATTN_SHAPES="256x256,512x128" restricts the (B x L) list, ATTN_TERMS=3|19 the arithmetic (default 19)."""
import ctypes, os, sys
import torch
from ctypes import c_float, c_int, c_int64, c_uint64, c_void_p as P
libs = []
for path in sys.argv[1:]:
    h = ctypes.CDLL(path)
    h.e3d_relkey_attn_fwd_split_ex.restype = c_int
    h.e3d_relkey_attn_fwd_split_ex.argtypes = [P, c_int64, c_int64, P, c_int64, c_int64, P, c_int64, c_int64, P, c_int, P, P, P, c_int,
                                               c_int, c_int, c_int, c_int, c_float, c_uint64, P, c_int, P, P, P, P]
    h.e3d_attn_scratch_bytes.restype = c_int64
    h.e3d_attn_scratch_bytes.argtypes = [c_int]
    libs.append((os.path.basename(path), h))
DEV = "cuda:0"
nh, H = 12, 768
terms = int(os.environ.get("ATTN_TERMS", "19"))
shapes = [tuple(int(x) for x in s.split("x")) for s in os.environ.get("ATTN_SHAPES", "256x256,512x128,1024x64").split(",")]
rounds = int(os.environ.get("ATTN_ROUNDS", "9"))
for B, L in shapes:
    qkv = torch.randn(B * L, 3 * H, device=DEV)
    E = torch.randn(2 * L - 1, 64, device=DEV)
    mask = torch.ones(B, L, device=DEV)
    mask[:, L - L // 5:] = 0           # a fifth of every row is padding (no bounds handed in: dense sweep)
    scratch = torch.empty(int(libs[0][1].e3d_attn_scratch_bytes(L)), dtype=torch.uint8, device=DEV)
    s = torch.cuda.current_stream().cuda_stream
    for name, e in (("rel-key", E), ("cross", None)):
        outs = {}

        def call(h, out, ready):
            rc = h.e3d_relkey_attn_fwd_split_ex(qkv.data_ptr(), L * 3 * H, 3 * H, qkv.data_ptr() + 4 * H, L * 3 * H, 3 * H,
                                                qkv.data_ptr() + 8 * H, L * 3 * H, 3 * H, e.data_ptr() if e is not None else None, L,
                                                mask.data_ptr(), out.data_ptr(), None, B, nh, L, L, terms, 0.0, 0,
                                                scratch.data_ptr(), ready, None, None, None, s)
            assert rc == 0, rc
        times = {n: [] for n, _ in libs}
        for rnd in range(rounds):
            for n, h in libs:
                out = outs.setdefault(n, torch.empty(B * L, H, device=DEV))
                call(h, out, 0)
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10):
                    call(h, out, 1)
                e1.record(); torch.cuda.synchronize()
                times[n].append(e0.elapsed_time(e1) / 10)
        ref = outs[libs[0][0]]
        diffs = "  ".join(f"{n}: max|d|={float((outs[n] - ref).abs().max()):.2e}" for n, _ in libs[1:])
        print(f"B={B} L={L} {name} terms={terms}: " + "   ".join(f"{n}: {sorted(t)[len(t) // 2] * 1e3:7.1f} us (min {min(t) * 1e3:.1f})" for n, t in times.items())
              + "   " + diffs, flush=True)
