#!/usr/bin/env python3
"""Lab: does a captured training step touch memory it does not own?  After the capture every FREE block of the default
caching-allocator pool is claimed and filled with a poison value; one replay later, any claimed block whose contents changed
was written by the graph -- its address was baked into the graph although the tensor that owned it has been freed.  The
allocation history (recorded from the start) then names the code that allocated that address last."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__  # noqa: E402
pkg = __graft_entry__.load_package()
from helpers import synthetic_pockets  # noqa: E402
from e3diff_amd import ops, training  # noqa: E402
from e3diff_amd.bert import BertConfig  # noqa: E402
from e3diff_amd.structure_model.model import ConditionalBertForDiffusion as M  # noqa: E402
from e3diff_amd.structure_model.dataset import noise_batch_on_device  # noqa: E402
from e3diff_amd.structure_model.utils import CosineTables  # noqa: E402

DEV, L, B = "cuda:0", 128, 32
layers = int(sys.argv[1]) if len(sys.argv) > 1 else 2
torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python", max_entries=400000)
c = dict(hidden_size=768, num_attention_heads=12, intermediate_size=1024, num_hidden_layers=layers, max_position_embeddings=L,
         hidden_dropout_prob=0.0, attention_probs_dropout_prob=0.0)
torch.manual_seed(0)
model = M(BertConfig(**c), BertConfig(**c, is_decoder=True, add_cross_attention=True), feature_names=list("abcdefgh"),
          loss_func=[M.diheral_loss_func] * 4 + [M.angle_loss_func] * 4, l2_lambda=0.1, learning_rate=1e-4).train().to(DEV)
tab = CosineTables(1000)
optim = model.configure_optimizers()["optimizer"]
params = [p for p in model.parameters() if p.requires_grad]
stepper = training.GraphedStep(model, optim, params, 1.0)
pk = {k: v.to(DEV) for k, v in synthetic_pockets(B, L, seed=0).items() if torch.is_tensor(v)}
with ops.arithmetic(training.TRAIN_ARITHMETIC):
    for k in range(4):
        stepper.step(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)))
    torch.cuda.synchronize()
    assert stepper.graph is not None
    batch = dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab))
    torch.cuda.synchronize()

    def free_default_blocks():
        out = []
        for seg in torch.cuda.memory_snapshot():
            if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
                continue
            addr = seg["address"]
            for b in seg["blocks"]:
                if b["state"] == "inactive":
                    out.append((addr, b["size"], seg.get("stream", 0)))
                addr += b["size"]
        return out

    import collections
    where = collections.Counter()
    detail = {}
    snap0 = torch.cuda.memory_snapshot()
    names0 = {id(p): n for n, p in model.named_parameters()}
    for p in params:
        a = p.grad.data_ptr()
        for seg in snap0:
            if seg["address"] <= a < seg["address"] + seg["total_size"]:
                addr = seg["address"]
                for b in seg["blocks"]:
                    if addr <= a < addr + b["size"]:
                        key = (tuple(seg.get("segment_pool_id", (0, 0))), b["state"], seg.get("stream", 0))
                        where[key] += 1
                        detail.setdefault(key, []).append(names0[id(p)])
                    addr += b["size"]
    print("where the parameters' .grad tensors live after the capture, (pool id, block state, segment stream) -> count:")
    for key, n in where.items():
        print("   ", key, n, detail[key][:3])
    held = []
    for rnd in range(6):                      # claim free blocks, largest first, until nothing (much) is left
        fb = sorted(free_default_blocks(), key=lambda t: -t[1])
        if not fb or sum(s for _, s, _ in fb) < (1 << 16):
            break
        for addr, size, stream in fb:
            try:
                with torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()):
                    t = torch.empty(max(1, (size - 256) // 4), dtype=torch.float32, device=DEV)
                held.append(t)
            except Exception as e:   # noqa: BLE001
                print("claim failed", size, e)
    torch.cuda.synchronize()
    POISON = 1.2345e30
    for t in held:
        t.fill_(POISON)
    torch.cuda.synchronize()
    print(f"claimed {len(held)} blocks, {sum(t.numel() * 4 for t in held) / 2 ** 20:.1f} MiB; free bytes left in the default pool: {sum(s for _, s, _ in free_default_blocks())}")
    stepper.step(batch)
    torch.cuda.synchronize()
    norm = float(optim.last_norm)
    print("gradient norm after one replay with every free default-pool block poisoned:", norm)
    names = {id(p): n for n, p in model.named_parameters()}
    order = {n: i for i, (n, _) in enumerate(model.named_parameters())}
    garbage = sorted(((order[names[id(p)]], names[id(p)], float(p.grad.abs().max())) for p in params if p.grad is not None and not (float(p.grad.abs().max()) < 1e3)))
    print(f"parameters with garbage gradients: {len(garbage)} of {len(params)}; last in forward order (first in backward):")
    for _, n, v in garbage[-8:]:
        print(f"      {n}: max |g| {v:.3e}")
    print("   first in forward order:", [n for _, n, _ in garbage[:4]])
    print("   loss:", float(stepper.loss))
    touched = [t for t in held if not bool((t == POISON).all())]
    print(f"claimed blocks the replay WROTE into: {len(touched)}")
    snap = torch.cuda.memory._snapshot()
    for t in touched[:6]:
        a = t.data_ptr()
        changed = int((t != POISON).sum())
        last = None
        for tr in snap["device_traces"]:
            for ev in tr:
                if ev["action"] == "alloc" and ev["addr"] <= a < ev["addr"] + ev["size"]:
                    last = ev if (last is None or True) else last
        hist = [ev for tr in snap["device_traces"] for ev in tr if ev["action"] == "alloc" and ev["addr"] <= a < ev["addr"] + ev["size"]]
        print(f"   block at {a:#x} ({t.numel() * 4} B): {changed} elements changed; {len(hist)} earlier allocations covered this address; the last ones:")
        for ev in hist[-3:]:
            fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "e3-invaraint" in f["filename"] or "tools/" in f["filename"]][:5]
            print(f"        size {ev['size']} stream {ev.get('stream')}: {' < '.join(fr)}")
