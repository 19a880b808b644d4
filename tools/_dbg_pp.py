import sys, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
from herald_amd import cache as hcache
from oracle import cache_model
dev = torch.device("cuda:0")
rng = np.random.default_rng(21)
rows, width, n, limit = 300, 8, 48, 30
table0 = rng.standard_normal((rows, width), dtype=np.float32)
server = cache_model.Server(table0)
model = cache_model.CacheModel("lru", limit, width, server, 2, 2)
table = torch.from_numpy(table0.copy()).to(dev)
versions = torch.zeros(rows, dtype=torch.int64, device=dev)
gpu = hcache.LRUCache(limit, rows, width, node_id=0, max_batch=64, device=dev)
gpu.bind_store(table, versions); gpu.pull_bound = gpu.push_bound = 2
batches = [((np.minimum(rng.zipf(1.3, size=n) - 1, rows - 1) * 31) % rows).astype(np.float32) for _ in range(40)]
want = model.lookup(batches[0].astype(np.uint64))
dest = torch.empty((n, width), dtype=torch.float32, device=dev)
gpu.embedding_lookup(torch.from_numpy(batches[0]).to(dev), dest).wait()
assert np.array_equal(dest.cpu().numpy(), want)
for k in range(5):
    grads = rng.standard_normal((n, width), dtype=np.float32) * np.float32(0.01)
    res_before = set(model.resident().keys())
    want = model.push_pull(batches[k + 1].astype(np.uint64), batches[k].astype(np.uint64), grads)
    gpu.embedding_push_pull(torch.from_numpy(batches[k + 1]).to(dev), dest, torch.from_numpy(batches[k]).to(dev), torch.from_numpy(grads).to(dev)).wait()
    got = dest.cpu().numpy()
    bad = np.nonzero((got != want).any(axis=1))[0]
    print("step", k, "bad rows", len(bad))
    for i in bad[:8]:
        key = int(batches[k+1][i])
        print("  pos", i, "key", key, "was_resident", key in res_before, "in_push", key in set(batches[k].astype(int).tolist()),
              "diff", (got[i]-want[i])[:3], "srvver", int(server.ver[key]), int(versions[key].item()))
    print("  table equal", np.array_equal(table.cpu().numpy(), server.table), "ver equal", np.array_equal(versions.cpu().numpy(), server.ver))
    if len(bad): break
