"""FramedStep at world size 1 (no process group): us per step, host-time profile (development aid).
ROWS / WIDTH / BATCH / STEPS / EAGER=1 from the environment."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from herald_amd import synth
from herald_amd.sharded import FramedStep, ShardedEmbedding
dev = torch.device("cuda:0"); torch.cuda.set_device(0)
rows, width, batch = int(os.environ.get("ROWS", 4_000_000)), int(os.environ.get("WIDTH", 512)), int(os.environ.get("BATCH", 256))
steps = int(os.environ.get("STEPS", 600))
n = batch * 26
emb = ShardedEmbedding(rows, width, dev)
emb.table.normal_(0, 0.01)
ids = [torch.from_numpy(np.minimum(synth.as_f32_ids(synth.criteo_batch(batch, b, rows=rows)).reshape(-1), rows - 1)).to(dev) for b in range(66)]
g = [torch.randn((n, width), device=dev) for _ in range(2)]
outs = [torch.empty((n, width), device=dev) for _ in range(2)]
fs = FramedStep(emb, n, graphs=os.environ.get("EAGER") != "1", block=int(os.environ.get("BLOCK", "8")))
LA = fs.LOOKAHEAD
fs.start([ids[j % 66] for j in range(LA)])
def step(k):
    fs.pull(ids[(k + LA) % 66], out=outs[k % 2]); fs.push(g[k % 2], 1e-6)
for k in range(6 * fs.block + 12): step(k)
torch.cuda.synchronize()
t0 = time.perf_counter()
for k in range(6 * fs.block + 12, 6 * fs.block + 12 + steps): step(k)
t_host = time.perf_counter() - t0
torch.cuda.synchronize()
print("host enqueue us/step %.2f" % (t_host / steps * 1e6))
print("us/step %.2f  (graphs %s, %d captured, fallbacks %d, row_cap %d)" % ((time.perf_counter() - t0) / steps * 1e6, fs.graphs, sum(1 for x in fs._graphs.values() if x), fs.fallbacks, fs.rcap))
if os.environ.get("HOSTPROF") == "1":
    pr = cProfile.Profile(); pr.enable()
    for k in range(6 * fs.block + 12 + steps, 6 * fs.block + 312 + steps): step(k)
    torch.cuda.synchronize(); pr.disable()
    pstats.Stats(pr).sort_stats("tottime").print_stats(14)
