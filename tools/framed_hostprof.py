"""Host-time profile of FramedStep (sized exchanges) at world size 1 (development aid): wall time per step, the host's
enqueue time per step, and where the host spends it."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import synth
from herald_amd.sharded import FramedStep, ShardedEmbedding
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
rows, width, bs = int(os.environ.get("ROWS", "33762577")), int(os.environ.get("WIDTH", "512")), int(os.environ.get("BATCH", "256"))
n = bs * 26
emb = ShardedEmbedding(rows, width, dev)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(bs, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(64)]
g = [torch.randn((n, width), device=dev) for _ in range(2)]
outs = [torch.empty((n, width), device=dev) for _ in range(2)]
fs = FramedStep(emb, n, block=int(os.environ.get("HA_SHARD_BLOCK", "16")), graphs=False)
LA = fs.LOOKAHEAD
fs.start([ids[j % 64] for j in range(LA)])
NATIVE = os.environ.get("NATIVE", "1") == "1" and fs.native_ok()      # runs of steps by one library call (ha_shard_steps)
def step(k):
    fs.pull(ids[(k + LA) % 64], out=outs[k % 2]); fs.push(g[k % 2], 1e-6)
if NATIVE:
    B = fs.block
    def step(k):        # (k = a block's first step: the whole block)
        if k % B == 0:
            fs.steps([ids[(k + i + LA) % 64] for i in range(B)], [g[(k + i) % 2] for i in range(B)], 1e-6,
                     outs=[outs[(k + i) % 2] for i in range(B)])
for k in range(200 if not NATIVE else 208): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(208 if NATIVE else 200, 1208 if NATIVE else 1200): step(k)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print("us/step wall %.2f   host enqueue %.2f" % ((t2 - t0) / 1000 * 1e6, (t1 - t0) / 1000 * 1e6))
pr = cProfile.Profile(); pr.enable()
for k in range(1208 if NATIVE else 1200, 1608 if NATIVE else 1600): step(k)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
