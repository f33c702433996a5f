"""CPU, world_size 2, gloo: the N>1 plumbing -- pocket sharding + final gather (sampling) and
bucketed gradient averaging incl. never-used parameters (training)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import __graft_entry__
    pkg = __graft_entry__.load_package()
    from e3diff_amd import sharding as S
    r, w, _ = S.init_distributed("gloo")
    assert (r, w) == (rank, world)

    # ---- sampling: shard 7 pockets, "sample" locally, gather in dataset order
    lo, hi = S.shard_slice(7, rank, world)
    local = [f"pocket{i}" for i in range(lo, hi)]
    gathered = S.gather_in_rank_order(local)
    assert gathered == [f"pocket{i}" for i in range(7)], gathered
    assert S.max_over_ranks(1.0 + rank) == float(world)
    # the frame of a trimmed training batch, agreed on the host (training.fit(trim_padding=True) under a process group)
    assert S.max_over_ranks_host((32 + 32 * rank, 96 - 32 * rank)) == (64, 96)
    from e3diff_amd import training as T
    m = (torch.arange(128)[None, :] < torch.tensor([[5 + 40 * rank], [17]])).float()
    b = {"ligand_attn_mask": m, "ligand_angles": torch.zeros(2, 128, 8), "receptor_attn_mask": torch.ones(2, 128),
         "receptor_seq": torch.zeros(2, 128, 20), "timestep": torch.zeros(2, 1)}
    t = T.trim_batch(b, S.max_over_ranks_host(T.trimmed_frame(b)))
    assert t["ligand_angles"].shape == (2, 64, 8) and t["receptor_seq"].shape == (2, 128, 20) and t["timestep"].shape == (2, 1)

    # ---- training: averaged grads == mean over ranks; a never-used parameter stays in sync
    torch.manual_seed(0)
    model = torch.nn.ModuleDict({"a": torch.nn.Linear(5, 3), "unused": torch.nn.Linear(4, 4),
                                 "b": torch.nn.Linear(3, 2)})
    if rank == 1:
        with torch.no_grad():
            for p in model.parameters():
                p.add_(1.0)
    v_before = model["a"].weight._version
    S.broadcast_parameters(model, src=0)
    assert model["a"].weight._version > v_before      # version-keyed inference caches see the broadcast
    x = torch.full((2, 5), float(rank + 1))
    model["b"](model["a"](x)).sum().backward()
    local_grad = model["a"].weight.grad.clone()
    avg = S.GradientAverager(model.parameters(), bucket_bytes=32)   # tiny buckets: several collectives
    assert len(avg.buckets) > 2
    avg.average()
    both = [torch.zeros_like(local_grad) for _ in range(world)]
    torch.distributed.all_gather(both, local_grad)
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    assert model["unused"].weight.grad is None        # never used on any rank: skipped by the optimizer, as single-rank
    # ---- the overlapped path: grads are views of the buckets, collectives start from the autograd hooks
    for p in model.parameters():
        p.grad = None
    avg.prepare()
    model["b"](model["a"](x)).sum().backward()
    launched_early = len(avg._handles)
    assert launched_early >= 1                       # at least one bucket went out before average()
    avg.average()
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    assert model["unused"].weight.grad is None
    views = [p.grad.untyped_storage().data_ptr() for p in model.parameters() if p.grad is not None]
    assert len(set(views)) <= len(avg.buckets)       # every grad lives inside a bucket buffer
    # ---- gradients written OUTSIDE autograd's accumulation (autograd.deferred_weight_grads computes the weight gradients
    # of the linear layers after backward and adds them into the bucket views): no hook fires, the writer reports them
    for p in model.parameters():
        p.grad = None
    avg.prepare()
    used = [p for n, p in model.named_parameters() if not n.startswith("unused.")]
    for p, g in zip(used, [both[rank]] + [torch.full_like(p, float(rank + 1)) for p in used[1:]]):
        assert float(p.grad.abs().max()) == 0.0           # zeroed view of its bucket
        p.grad.add_(g)                                    # what the grouped launch does (accumulate bit set)
        avg.mark_ready(p)
    avg.average()
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    assert torch.allclose(model["b"].bias.grad, torch.full_like(model["b"].bias, 1.5))
    assert model["unused"].weight.grad is None
    # a parameter reported twice in one step (hook AND mark_ready: a weight with a deferred and a direct use) counts once
    for p in model.parameters():
        p.grad = None
    avg.prepare()
    model["b"](model["a"](x)).sum().backward()
    for p in used:
        avg.mark_ready(p)
    assert all(n >= 0 for n in avg._pending)
    avg.average()
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    # ---- overlap=False installs no hooks, so prepare() must not bind the view path -- average()
    # then takes the copying path and EVERY used parameter keeps its (averaged) gradient
    for p in model.parameters():
        p.grad = None
    plain = S.GradientAverager(model.parameters(), bucket_bytes=32, overlap=False)
    plain.prepare()
    assert not plain._bound
    model["b"](model["a"](x)).sum().backward()
    plain.average()
    assert all(p.grad is not None for p in used)
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    assert model["unused"].weight.grad is None
    # ---- the graph-segment form (training.GraphedDDPStep): backward into the bucket views with NO collective from the
    # hooks (that pass is captured), the buckets summed afterwards in one place, the division left to the caller's segment
    for p in model.parameters():
        p.grad = None
    avg.bind(collect_only=True)
    model["b"](model["a"](x)).sum().backward()
    assert len(avg._handles) == 0                     # nothing on the wire yet
    avg.finish_collect()
    assert model["unused"].weight.grad is None and model["a"].weight.grad is not None
    avg.all_reduce_flats()
    for flat in avg.flats():
        flat.div_(world)
    assert torch.allclose(model["a"].weight.grad, sum(both) / world)
    w0 = [torch.zeros_like(model["a"].weight) for _ in range(world)]
    torch.distributed.all_gather(w0, model["a"].weight.data)
    assert torch.equal(w0[0], w0[1])
    q.put((rank, "ok"))
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_sharding_and_gradient_average():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=180)
    assert all(p.exitcode == 0 for p in procs), [p.exitcode for p in procs]
    assert sorted(q.get(timeout=5) for _ in range(2)) == [(0, "ok"), (1, "ok")]


@pytest.mark.parametrize("n,world", [(7, 2), (1024, 8), (5, 8), (0, 3)])
def test_shard_slice_partitions(n, world):
    import __graft_entry__
    __graft_entry__.load_package()
    from e3diff_amd.sharding import shard_slice
    spans = [shard_slice(n, r, world) for r in range(world)]
    assert spans[0][0] == 0 and spans[-1][1] == n
    assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
    sizes = [b - a for a, b in spans]
    assert max(sizes) - min(sizes) <= 1


def test_bench_self_launches_two_ranks_without_external_launcher():
    """`python bench.py --gpus 2` as the driver invokes it (no torchrun, WORLD_SIZE unset): the parent spawns one
    fresh worker per rank before touching any GPU, the ranks rendezvous (gloo here), reduce max-over-ranks, and
    rank 0 prints exactly one JSON line.  E3D_BENCH_REHEARSAL=cpu replaces the GPU work by a stub."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["E3D_BENCH_REHEARSAL"] = "cpu"
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1"],
                       env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["steps"] == 4 and out["warmup"] == 1
    assert abs(out["max_elapsed_s"] - 0.002) < 1e-12      # the slowest rank's time, not rank 0's
    # a mismatch between --gpus and an inherited WORLD_SIZE is an error, not an assert deep inside
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2"], env=dict(env, WORLD_SIZE="1"),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr
