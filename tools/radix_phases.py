"""Where a scatter launch of the radix sort spends its time (development aid; run on the GPU box with HA_RADIX_STAMPS=1):
106,496 criteo-shaped ids, the last workgroup's clock at its phase boundaries, averaged over 32 sorts."""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["HA_RADIX_STAMPS"] = "1"
sys.path.insert(0, ".")
from herald_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
n, rows = int(os.environ.get("N", 106496)), 33762577
rng = np.random.default_rng(0)
card = np.maximum((rows * np.array([0.3 ** (i % 7 + 1) for i in range(26)]) / 5).astype(np.int64), 4)
per = n // 26
ids = np.concatenate([(rng.zipf(1.2, per) - 1) % card[f] + card[:f].sum() for f in range(26)]).astype(np.float32)
t = torch.from_numpy(ids).to(dev)
plan = ops.IndexPlan(t.numel(), dev)
L = _lib.load()
acc = []
for i in range(40):
    plan.sort(t, key_limit=rows)
    torch.cuda.synchronize()
    out = (ctypes.c_uint64 * 24)()
    L.ha_plan_radix_stamps(ctypes.c_void_p(plan.ws.data_ptr()), t.numel(), out, None)
    if i >= 8:
        acc.append([int(x) for x in out])
a = np.array(acc, dtype=np.float64) / 100.0
names = ["LDS zeroed + keys loaded", "ranked", "histograms exchanged", "sums read", "digit bases", "scattered (stores landed)"]
for p in range(3):
    b = a[:, 8 * p:8 * p + 8]
    print("pass %d (%s): total %.2f us" % (p, "scatter behind the histogram launch" if p == 0 else "one launch", (b[:, 6] - b[:, 0]).mean()))
    for i, nm in enumerate(names):
        lo = b[:, i] if i < 5 else b[:, 5]
        hi = b[:, i + 1]
        print("    %-28s %6.2f" % (nm, (hi - lo).mean()))
    if p < 2:
        print("  gap to the next pass's start %.2f us" % (a[:, 8 * (p + 1)] - b[:, 6]).mean())
