#!/usr/bin/env python3
"""Host-time breakdown of one sharded step at world size 1 (development aid)."""
import os, sys, time, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import numpy as np, torch, torch.distributed as dist
from herald_amd import synth, sharded
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
rows, width, n = 4000000, 512, 6656
emb = sharded.ShardedEmbedding(rows, width, dev)
emb.table.normal_(0, 0.01)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(64)]
grads = torch.randn((n, width), device=dev)
acc = collections.defaultdict(float)
def wrap(obj, name):
    f = getattr(obj, name)
    def g(*a, **k):
        t = time.perf_counter(); r = f(*a, **k); acc[name] += time.perf_counter() - t; return r
    setattr(obj, name, g)
for nm in ("route_issue", "gather_keys", "expand", "reduce_scaled", "acc_apply", "to_host", "host_sync"):
    wrap(emb.engine, nm)
wrap(emb, "_a2a")
state = {"route": emb.prefetch(ids[0], after_current=False)}
def step(k):
    cur = state["route"]
    nxt = emb.prefetch(ids[(k + 1) % 64], after_current=False)
    emb.pull(route=cur)
    emb.push(None, grads, 1e-6, route=cur)
    emb.complete(nxt)
    state["route"] = nxt
for k in range(20): step(k)
torch.cuda.synchronize(); acc.clear()
t0 = time.perf_counter()
for k in range(200): step(k)
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("host %.1f us/step, host+drain %.1f us/step" % ((t1 - t0) / 200 * 1e6, (t2 - t0) / 200 * 1e6))
for k, v in sorted(acc.items(), key=lambda kv: -kv[1]):
    print("  %-14s %.1f us/step" % (k, v / 200 * 1e6))
dist.destroy_process_group()
