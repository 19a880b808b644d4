"""laia LaiaScheduler at BASELINE configs[3]'s shape (4 workers x 1024 samples, 26 tables, cache 0.1 x rows): us per
global batch (development aid; BATCHES from the environment, HA_LAIA_HOST=1 for the host-snapshot mode)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from herald_amd import laia as hlaia, synth
rows, W, mini_bs, T = 33762577, 4, 1024, 26
batch_num = int(os.environ.get("BATCHES", "40"))
per = 256
need = W * mini_bs * batch_num + 1000
parts = [synth.criteo_batch(per, step=5000 + s, rows=rows, nfields=T) for s in range((need + per - 1) // per)]
samples = np.concatenate(parts, axis=0)[:need].astype(np.uint64)
s = hlaia.LaiaScheduler()
t0 = time.perf_counter()
s.start(samples, samples.shape[0], T, 1, mini_bs, batch_num, W, 0, int(0.1 * rows), 16, 24, key_limit=rows)
n = 0
while s.pop() != [0]:
    n += 1
el = time.perf_counter() - t0
tm = s.timing()
s.close()
print("batches %d: %.0f us per global batch inside ha_laia_next (host assign %.0f, host snapshot %.0f, rest %.0f); wall %.0f us per batch incl. queue + init"
      % (tm["batches"], tm["us_per_batch"], tm["host_assign_us"], tm["host_snapshot_us"], tm["gpu_and_transfer_us"], el / max(tm["batches"], 1) * 1e6))
