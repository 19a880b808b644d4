"""bench.py's JSON contract on a small table: the N=1 line (roofline + cpu_baseline + cache_tier) and the
sharded N>1 leg forced at world size 1 through the nccl backend (HA_FORCE_SHARDED=1)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra_env, *args):
    env = dict(os.environ)
    env.update(extra_env)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(args), env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_n1_line_has_the_contract_fields(dev):
    d = _run({}, "--rows", "1000000", "--steps", "256", "--warmup", "64", "--distinct-batches", "64",
             "--cold-rows", "2000000")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 256 and d["unit"] == "rows/s" and d["vs_baseline"] is None
    assert d["dtype"] == "f32" and d["scaling"] == "weak" and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["ids_per_step"] * 1e3 / d["ms_per_step"]) / d["value"] < 1e-6
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0 < rf["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "rows/s"
    assert d["value"] > 20 * cb["value"]
    assert d["cache_tier"]["value"] > 0
    ct = d["cold_tier"]
    assert ct["value"] > 0 and 0 <= ct["hot_tier_hit_rate"] <= 1 and ct["pcie_GBps"] >= 0
    assert d["config"]["launches_per_step"] == 1 and d["config"]["engine"] == "queue"
    assert "qapply_kernel" in rf["kernel"]
    assert abs(rf["avg_launch_us"] - d["device_ms"] / d["steps"] * 1e3) < 1e-6


def test_bench_handoff_engine_line(dev):
    """--engine handoff: the bit-exact one-launch step with the in-launch hand-off (round 2's default)."""
    d = _run({}, "--rows", "1000000", "--steps", "128", "--warmup", "32", "--distinct-batches", "64", "--engine", "handoff",
             "--no-cache-tier", "--no-cpu-baseline", "--no-cold-tier", "--no-laia")
    assert d["config"]["engine"] == "handoff" and d["handoff_timeouts"] == 0
    assert "step_kernel" in d["roofline"]["kernel"] and d["value"] > 0


def test_bench_short_driver_run(dev):
    """The driver's `--steps 20 --warmup 5`: the line reports the warm-up it was asked for and the launch mode it really
    used -- plain launches for the queue engine by default, hipGraph replays on request."""
    common = ("--rows", "1000000", "--steps", "20", "--warmup", "5", "--distinct-batches", "64",
              "--no-cache-tier", "--no-cpu-baseline", "--no-cold-tier", "--no-laia")
    d = _run({}, *common)
    assert d["steps"] == 20 and d["warmup"] == 5
    assert "plain launches" in d["config"]["launch"] and "gate" in d["timed_region"]
    assert d["config"]["grad_and_out_buffers"] >= 24
    # behind the gate the steps are enqueued before the device starts on them: the region is the device's time, the
    # enqueue time is reported beside it
    assert abs(d["ms_per_step"] * 20 - d["device_ms"]) < 1e-6 and d["enqueue_ms"] > 0 and d["host_bound"] is False
    u = _run({}, *common, "--no-gate")
    assert abs(u["ms_per_step"] * 20 - max(u["device_ms"], u["enqueue_ms"])) < 1e-6 and "gate" not in u["timed_region"]
    g = _run({}, *common, "--graph-steps", "16")
    assert "hipGraph" in g["config"]["launch"]        # steps 5..24: the blocks [0, 16) and [16, 32) -> two replays


def test_bench_two_launch_mode_still_reports_measured_kernel_times(dev):
    d = _run({}, "--rows", "1000000", "--steps", "64", "--warmup", "32", "--distinct-batches", "64", "--launches", "2",
             "--no-cache-tier", "--no-cpu-baseline", "--no-cold-tier", "--no-laia")
    assert d["config"]["launches_per_step"] == 2
    k = d["kernels"]
    assert all("measured_us" in v and v["measured_us"] > 0 for v in k.values()) and len(k) == 2
    assert d["roofline"]["avg_launch_us"] == max(v["measured_us"] for v in k.values())


def test_bench_sharded_leg_runs_on_nccl_at_world_size_1(dev):
    d = _run({"HA_FORCE_SHARDED": "1", "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": "29641", "RANK": "0",
              "WORLD_SIZE": "1", "LOCAL_RANK": "0"},
             "--rows", "1000000", "--steps", "64", "--warmup", "16")
    assert d["n_gpus"] == 1 and d["steps"] == 64 and d["value"] > 0 and d["scaling"] == "weak"
    assert "sharded" in d["config"]["workload"] and "xgmi" in d
    assert d["ranks_seen"] == 1 and d["roofline"]["frac"] > 0 and d["cpu_baseline"]["value"] > 0
    # the framed step with sized exchanges (no host read-back in the step), per-launch roofline, and BASELINE configs[2]'s
    # shape in the same line; at world size 1 a step is the lookup + ONE reduce-and-add launch
    assert "sized by the real per-owner counts" in d["config"]["exchange"] and "0 of 64 timed steps overflowed" in d["config"]["exchange"]
    ks = d["roofline"]["kernels"]
    assert len(ks) == 2 and all(v["us"] > 0 and 0 < v["frac_of_hbm_peak"] < 1.3 for v in ks.values())
    assert d["same_path_world1_ms_per_step"] is None and d["xgmi"]["bytes_carried_per_step"] == 0
    c = d["config_c"]
    assert "error" not in c, c
    assert "bs=4096 d=128" in c["config"]["workload"] and c["value"] > 0 and c["config"]["ids_per_step_per_gpu"] == 106496


def test_same_path_world1_measurement_runs_on_a_shard(dev):
    """The diagnostic of bench.py's N>1 line (`same_path_world1_ms_per_step`: the N>1 engine at world size 1 on the rank's
    own shard) needs no process group: run it on a store of its own here."""
    import numpy as np
    import torch
    from herald_amd import sharded_bench, synth
    from herald_amd.sharded import ShardedEmbedding
    rows, width, bs = 300_000, 64, 64
    emb = ShardedEmbedding(rows, width, dev)
    emb.table.normal_(0, 0.01)
    n = bs * 26
    ids = torch.from_numpy(np.stack([synth.as_f32_ids(synth.criteo_batch(bs, b, rows=4 * rows)).reshape(-1)
                                     for b in range(64)])).to(dev)          # ids beyond the shard: folded into it
    grads = [torch.randn((n, width), device=dev) for _ in range(2)]
    before = emb.table.clone()
    ms = sharded_bench._world1_same_path(None, 0, dev, emb, ids, grads, n, 1e-3, steps=40, warmup=20)
    assert 0 < ms < 5.0
    assert not torch.equal(before, emb.table)           # the steps did push into the shard


def _n_gt_1_worker(rank, world, port, out_path):
    """One rank of bench.py's N>1 leg (herald_amd.sharded_bench.run) with `world` processes on ONE GPU: a gloo group, the
    row / key exchanges staged through the host (RCCL refuses two ranks on one device)."""
    import argparse
    import contextlib
    import io
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from herald_amd import sharded_bench

    def staged(out, inp, out_splits, in_splits, group):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu(), out_splits, in_splits, group=group)
        out.copy_(o)

    sharded_bench.A2A_HOOK = staged
    args = argparse.Namespace(rows=400_000, width=64, batch=64, fields=26, steps=24, warmup=6, distinct_batches=32,
                              no_cpu_baseline=True, no_config_c=False)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        sharded_bench.run(args, rank, world, torch.device("cuda:0"))
    if rank == 0:
        with open(out_path, "w") as f:
            f.write(buf.getvalue())


def test_bench_n_gt_1_leg_end_to_end_with_two_ranks_on_one_gpu(dev, tmp_path):
    """Everything bench.py --gpus N runs on a rank -- the sized FramedStep over a real (gloo) group, the same-path world-1
    diagnostic, the reductions of the xGMI statistics, configs[2]'s shape as the second measurement -- at world size 2, and
    the line rank 0 prints: whole-job rows/s, bytes carried = bytes useful for the rows, owner skew, both shapes."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    out = str(tmp_path / "line.json")
    mp.spawn(_n_gt_1_worker, args=(2, port, out), nprocs=2, join=True)
    lines = [l for l in open(out).read().splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["ranks_seen"] == 2 and d["scaling"] == "weak" and d["steps"] == 24 and d["value"] > 0
    assert abs(d["value"] - 2 * 64 * 26 / (d["ms_per_step"] * 1e-3)) / d["value"] < 1e-6       # whole-job rows/s
    x = d["xgmi"]
    assert x["bytes_carried_per_step"] > 0 and x["owner_rows_max"] >= x["owner_rows_mean"] > 0
    assert d["same_path_world1_ms_per_step"] is not None and d["same_path_world1_ms_per_step"] > 0
    assert "sized" in d["config"]["exchange"]
    c = d["config_c"]
    assert "error" not in c and c["value"] > 0 and c["config"]["ids_per_step_per_gpu"] == 4096 * 26
    assert c["xgmi"]["bytes_carried_per_step"] > 0
