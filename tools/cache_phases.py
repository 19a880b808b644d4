"""Where the single-workgroup bookkeeping of a cache lookup spends its time (development aid; run on the GPU box):
the criteo batch against a cache filled to its limit, phase boundaries stamped with the 100 MHz clock."""
import ctypes
import sys

import numpy as np
import torch

sys.path.insert(0, ".")
from herald_amd import cache as hcache  # noqa: E402

dev = torch.device("cuda:0")
rows, width, n = 33762577, 512, 6656
table = torch.zeros((rows, width), dtype=torch.float32, device=dev)
versions = torch.zeros(rows, dtype=torch.int64, device=dev)
hcache.register_table(0, table, versions)
limit = int(0.1 * rows)
c = hcache.CacheSparseTable(limit, rows, width, 0, "LRU", bound=100, max_batch=n, device=dev)
out = torch.empty((n, width), dtype=torch.float32, device=dev)
grad = torch.full((n, width), 1e-3, dtype=torch.float32, device=dev)
base = torch.arange(n, device=dev)
for lo in range(0, limit + n, n):
    kk = ((base + lo) % rows).to(torch.int64)
    c.embedding_lookup(kk, out)
    c.embedding_update(kk, grad, same_as_lookup=True)
rng = np.random.default_rng(0)
card = np.maximum((rows * np.array([0.3 ** (i % 7 + 1) for i in range(26)]) / 5).astype(np.int64), 4)
acc = []
for step in range(64):
    ids = np.concatenate([(rng.zipf(1.2, 256) - 1) % card[f] + card[:f].sum() for f in range(26)]).astype(np.int64) % rows
    kk = torch.from_numpy(ids).to(dev)
    c.embedding_lookup(kk, out)
    ph = (ctypes.c_uint64 * 16)()
    c.cache._L.ha_cache_phase_times(c.cache._h, ph, ctypes.c_void_p(c.cache._stream().cuda_stream))
    c.embedding_update(kk, grad, same_as_lookup=True)
    if step >= 16:
        acc.append([int(x) for x in ph])
raw = np.array(acc, dtype=np.float64)
a = raw / 100.0      # us
names = {1: "probe / U", 2: "miss scan", 3: "assign + touch", 4: "commit + pull count"}
print("bookkeeping (us, mean over %d lookups):" % len(acc))
if (a[:, 3] >= a[:, 1]).all() and (a[:, 3] <= a[:, 4]).all():      # the phase-by-phase path (> 8192 unique keys)
    for i in range(1, 5):
        print("  %-22s %6.2f" % (names[i], (a[:, i] - a[:, i - 1]).mean()))
else:           # cache_finish_book_kernel (the last chunk's workgroup) or the register-resident book
    print("  %-34s %6.2f" % ("finish + probe + count exchange", (a[:, 1] - a[:, 0]).mean()))
    print("  %-34s %6.2f" % ("assignment / touch / commit", (a[:, 4] - a[:, 1]).mean()))
print("  total                  %6.2f" % (a[:, 4] - a[:, 0]).mean())
print("insert / eviction workgroup:")
print("  header + insert        %6.2f" % (a[:, 9] - a[:, 8]).mean())
print("  eviction               %6.2f" % (a[:, 10] - a[:, 9]).mean())
print("  compaction + report    %6.2f" % (a[:, 12] - a[:, 10]).mean())
print("  total                  %6.2f" % (a[:, 12] - a[:, 8]).mean())
print("  starts after the bookkeeping's end by %6.2f" % (a[:, 8] - a[:, 4]).mean())
print("  eviction walk: %.1f rounds, %.0f lines to evict, %.0f log entries consumed" % (
    raw[:, 13].mean(), raw[:, 14].mean(), raw[:, 15].mean()))
