#!/usr/bin/env python3
"""Lab: which allocations made WHILE a training step is being captured bypass the graph's private memory pool?  Those
blocks are ordinary caching-allocator memory: freed after the capture and handed to later eager allocations while the
graph keeps reading / writing them."""
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


def default_pool_blocks():
    out = {}
    for seg in torch.cuda.memory_snapshot():
        if tuple(seg.get("segment_pool_id", (0, 0))) != (0, 0):
            continue
        addr = seg["address"]
        for b in seg["blocks"]:
            if b["state"] == "active_allocated":
                out[addr] = b["size"]
            addr += b["size"]
    return out


orig_capture = stepper._capture


def audited(batch):
    torch.cuda.synchronize()
    before = default_pool_blocks()
    torch.cuda.memory._record_memory_history(enabled="all", context="all", stacks="python", max_entries=200000)
    orig_capture(batch)
    snap = torch.cuda.memory._snapshot()
    torch.cuda.memory._record_memory_history(enabled=None)
    default_segs = [(sg["address"], sg["address"] + sg["total_size"]) for sg in snap["segments"] if tuple(sg.get("segment_pool_id", (0, 0))) == (0, 0)]
    import collections
    where = collections.Counter()
    nbytes = collections.Counter()
    for tr in snap["device_traces"]:
        for ev in tr:
            if ev["action"] == "alloc" and any(a <= ev["addr"] < b for a, b in default_segs):
                fr = [f"{os.path.basename(f['filename'])}:{f['line']}:{f['name']}" for f in ev.get("frames", []) if "e3-invaraint" in f["filename"] or "tools/" in f["filename"]][:3]
                key = (ev.get("stream"), " < ".join(fr))
                where[key] += 1
                nbytes[key] += ev["size"]
    print("allocations served from DEFAULT-pool segments while capturing, by (stream, nearest repo frames):")
    for key, n in where.most_common(25):
        print(f"   {n:5d} x  {nbytes[key] / 2 ** 20:9.2f} MiB  stream {key[0]}  {key[1]}")
    after = default_pool_blocks()
    new = {a: s for a, s in after.items() if a not in before}
    print(f"capture: {len(new)} blocks ({sum(new.values()) / 2 ** 20:.2f} MiB) were allocated in the DEFAULT pool while capturing")
    # which live tensors own them?
    import gc
    owners = {}
    for obj in gc.get_objects():
        try:
            if torch.is_tensor(obj) and obj.is_cuda and obj.untyped_storage().data_ptr() in new:
                owners.setdefault(obj.untyped_storage().data_ptr(), (tuple(obj.shape), obj.dtype))
        except Exception:   # noqa: BLE001
            pass
    print("   still referenced from Python:", len(owners), "of", len(new), list(owners.values())[:12])
    sizes = sorted(new.values(), reverse=True)[:12]
    print("   largest:", sizes)


stepper._capture = audited
with ops.arithmetic(training.TRAIN_ARITHMETIC):
    for k in range(5):
        stepper.step(dict(pk, **noise_batch_on_device(pk["ligand_angles"], tab)))
torch.cuda.synchronize()
print("replaying:", stepper.graph is not None)
