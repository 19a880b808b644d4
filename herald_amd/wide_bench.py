"""Secondary measurement of bench.py (never part of `value`): the work-queue step at the per-GPU shapes of BASELINE
configs[2] / configs[3] -- bs=4096 d=128 (106,496 ids per step) and bs=1024 d=512 (26,624) -- on one GPU, through the
WIDE path of ops.QueueStepPipeline (hash buckets planned and joined side by side, no sort; csrc/qstep.hip).  Blocks of
steps are prepared on a side stream inside the timed region, as in the headline."""
import numpy as np
import torch

HBM_PEAK_GBS = 8000.0


def pmc_traffic(batch, width):
    """HBM bytes per ha::qapply_kernel launch at this shape from the newest committed PMC summary (profiles/rNN/
    pmc_traffic_wide_bs<batch>_d<width>.json: separate rocprofv3 --pmc passes over tools/shape_bench.py), or None."""
    import glob
    import json
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*", "pmc_traffic_wide_bs%d_d%d.json" % (batch, width))),
                    reverse=True):
        try:
            with open(f) as fh:
                ks = json.load(fh)["kernels"]
        except (OSError, ValueError, KeyError):
            continue
        for name, v in ks.items():
            if name.startswith("ha::qapply_kernel"):
                return v["hbm_bytes_per_launch"], os.path.relpath(f, root)
    return None, None


def measure(table, rows, batch, width, fields=26, block=16, steps=96, sync="flags", distinct=32, lr=1e-6, alone=False):
    """block: steps per preparation block (16 as the headline: 47.4–47.5 / 35.3–36.0 µs per step at the two shapes against
    49.1 / 36.6 with blocks of 4, docs/EXPERIMENTS.md round 6 §9)."""
    from . import ops, synth
    dev = table.device
    n = batch * fields
    host = [np.minimum(synth.as_f32_ids(synth.criteo_batch(batch, b, rows=rows, nfields=fields)).reshape(-1),
                       np.float32(rows - 1)) for b in range(distinct)]
    ids = [torch.from_numpy(h).to(dev) for h in host]
    u_mean = float(np.mean([np.unique(h).size for h in host]))
    # gradient / output buffers: more than the 256 MiB Infinity Cache in rotation, so both streams are HBM traffic
    nbuf = max(2, min(24, (400 << 20) // (n * width * 4)))
    gen = torch.Generator(device=dev)
    gen.manual_seed(456)
    grads = [torch.randn((n, width), dtype=torch.float32, device=dev, generator=gen) for _ in range(nbuf)]
    outs = [torch.empty((n, width), dtype=torch.float32, device=dev) for _ in range(nbuf)]
    pipe = ops.QueueStepPipeline(table, n, lr, block=block, sync=sync)
    LA = pipe.LOOKAHEAD
    s = torch.cuda.Stream(device=dev)
    ids_of = lambda j: ids[j % distinct] if j >= 0 else None
    warm = max(24, 3 * block)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    with torch.cuda.stream(s):
        for c in range(-LA, warm + steps):
            if c % block == 0:
                pipe.prepare_block(c // block, ids_of, stream=s)
            if c < -1:
                continue
            if c == warm:
                e0.record(s)
            pipe.apply(c, grads[c % nbuf] if c >= 0 else None, outs[(c + 1) % nbuf], stream=s, n_cur=n if c >= 0 else 0,
                       n_next=n)
        e1.record(s)
    torch.cuda.synchronize()
    if pipe.overflowed():
        raise RuntimeError("the wide path raised its sticky error word")
    us = e0.elapsed_time(e1) * 1e3 / steps
    alone_us = None
    if alone:
        # development aid: the apply launch of ONE step again and again with nothing beside it (its queue stays built;
        # the table takes the same gradient positions repeatedly -- a timing, not a training step)
        c = warm + steps - 2 if (warm + steps - 1) % block == block - 1 else warm + steps - 1
        reps = 40
        with torch.cuda.stream(s):
            for i in range(reps + 4):
                if i == 4:
                    e0.record(s)
                pipe.apply(c, grads[i % nbuf], outs[(i + 1) % nbuf], stream=s, n_cur=n, n_next=n)
            e1.record(s)
        torch.cuda.synchronize()
        alone_us = e0.elapsed_time(e1) * 1e3 / reps
    alg = n * (12 * width + 8) + u_mean * 8 * width
    hdr = pipe.queue_header(warm + steps - 1)
    traffic, traffic_src = pmc_traffic(batch, width)
    return {"workload": "wdl_criteo bs=%d d=%d on ONE GPU, %d ids per step, full %d-row table; work-queue step, wide path "
                        "(%d hash buckets per batch, no sort), blocks of %d steps prepared on a side stream inside the timed "
                        "region" % (batch, width, n, rows, pipe.plans[0].buckets if pipe.wide else 1, block),
            "us_per_step": us, "rows_per_s": n / (us * 1e-6), "steps": steps, "ids_per_step": n, "unique_per_step": u_mean,
            "numerics": "tolerance>=16", "stream_sync": pipe.sync,
            "roofline": {"bound": "hbm", "kernel": "ha::qapply_kernel (launch-to-launch period, preparation beside it)",
                         "achieved": alg / us / 1e3, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": alg / us / 1e3 / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # the same period priced on the bytes the counters saw (2 x FETCH_SIZE + WRITE_SIZE per launch)
                         "frac_by_traffic": (traffic / us / 1e3 / HBM_PEAK_GBS) if traffic else None,
                         "algorithmic_bytes_per_launch": alg},
            "queue_items": hdr, "apply_alone_us": alone_us}
