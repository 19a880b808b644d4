"""Host-time profile of the sharded step at world size 1 (nccl backend), development aid."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29655")
os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
import numpy as np, torch, torch.distributed as dist
from herald_amd import synth
from herald_amd.sharded import ShardedEmbedding
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
rows, width, n = 4_000_000, 512, 6656
emb = ShardedEmbedding(rows, width, dev)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(256, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(64)]
g = torch.randn((n, width), device=dev)
state = {"route": emb.prefetch(ids[0], after_current=False)}
def step(k):
    cur = state["route"]
    nxt = emb.prefetch(ids[(k + 1) % 64], after_current=False)
    emb.pull(route=cur); emb.push(None, g, 1e-6, route=cur); emb.complete(nxt)
    state["route"] = nxt
for k in range(200): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(500): step(k)
torch.cuda.synchronize()
print("us/step", (time.perf_counter() - t0) / 500 * 1e6)
pr = cProfile.Profile(); pr.enable()
for k in range(300): step(k)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(18)
dist.destroy_process_group()
