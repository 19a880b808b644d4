"""End-to-end drop-in check: examples/ctr/run_wdl.py (the reference's wdl_criteo model with the
embedding path on herald_amd) against the same training loop written with plain PyTorch ops.

PyTorch sums the gradients of duplicate ids before the update while the reference (and herald_amd)
applies them occurrence by occurrence, so tables agree within rounding: 1e-5 relative, the tolerance
BASELINE.json's north_star states for accumulated fp32 gradients."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "examples", "ctr"))

pytestmark = pytest.mark.gpu

ROWS, WIDTH, BATCH, STEPS, LR = 30000, 16, 128, 12, 0.05
RTOL = 1e-5     # north_star tolerance for accumulated fp32 gradients
ATOL = 5e-7     # table values are ~1e-2: entries that cancel to ~0 have no meaningful relative error


def _torch_reference(dev, table_init):
    import run_wdl
    tower = run_wdl.Tower(WIDTH, 0).to(dev)
    table = table_init.clone()
    opt = torch.optim.SGD(tower.parameters(), lr=LR)
    batches = run_wdl.make_batches(min(STEPS + 1, 64), BATCH, ROWS, 0)
    losses = []
    for k in range(STEPS):
        ids, dense, label = (torch.from_numpy(a).to(dev) for a in batches[k % len(batches)])
        idx = ids.long()
        emb = table[idx].clone().requires_grad_(True)
        pred = tower(dense, emb.reshape(BATCH, run_wdl.NFIELD * WIDTH))
        loss = torch.nn.functional.binary_cross_entropy(pred, label)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        table.index_add_(0, idx.reshape(-1), emb.grad.reshape(-1, WIDTH), alpha=-LR)
        losses.append(float(loss.detach()))
    return losses, table, tower


@pytest.fixture(scope="module")
def reference(dev):
    g = torch.Generator(device=dev).manual_seed(1)
    table_init = torch.randn((ROWS, WIDTH), generator=g, device=dev) * 0.01
    return table_init, _torch_reference(dev, table_init)


def _global_table(param):
    return param.table if param.table is not None else param.store.table


@pytest.mark.parametrize("embedding,kw", [("hbm", {}), ("step", {}), ("step3", {}), ("queue", {}), ("ps", {}), ("cache", {"cache": "LRU", "bound": 0}),
                                          ("cache", {"cache": "LFUOpt", "bound": 0, "cache_limit": 2000}),
                                          ("cache", {"cache": "LRU", "bound": 0, "cache_planned": True}),
                                          ("cache", {"cache": "LFU", "bound": 2, "cache_planned": True})])
def test_wdl_training_matches_pytorch(dev, reference, embedding, kw):
    import run_wdl
    table_init, (ref_losses, ref_table, ref_tower) = reference
    losses, param, tower = run_wdl.train(embedding, ROWS, WIDTH, BATCH, STEPS, LR, table_init=table_init,
                                         device=str(dev), **kw)
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4)
    if kw.get("cache_planned"):
        # (--cache-planned: the cache's planned flow.  Planned batches are outstanding when the loop stops -- the next batch is
        # pulled, the one after it booked --, so the cache takes no call-by-call lookup here; what the lookups returned step by
        # step is in the losses, what the tower learned from them below)
        assert param.cache.cache.plan_pending() > 0
        for p, q in zip(tower.parameters(), ref_tower.parameters()):
            torch.testing.assert_close(p, q, rtol=1e-4, atol=1e-7)
        return
    if embedding == "cache":
        # rows still held by the cache with unpushed updates are not in the store yet: flush by
        # comparing through a lookup of every touched row instead
        touched = torch.unique(torch.cat([torch.from_numpy(b[0]).reshape(-1) for b in
                                          run_wdl.make_batches(STEPS, BATCH, ROWS, 0)])).to(dev)
        dest = torch.empty((touched.numel(), WIDTH), dtype=torch.float32, device=dev)
        param.cache.embedding_lookup(touched, dest).wait()
        got, want = dest, ref_table[touched.long()]
    else:
        got, want = _global_table(param), ref_table
    torch.testing.assert_close(got, want, rtol=RTOL, atol=ATOL)
    for p, q in zip(tower.parameters(), ref_tower.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-4, atol=1e-7)
    assert not torch.equal(want, table_init[touched.long()] if embedding == "cache" else table_init)


def test_wdl_one_launch_step_equals_the_two_call_path_bit_for_bit(dev):
    """--embedding step (ha_sgd_push_pull, width 32: the single-launch path with the in-launch hand-off) and
    --embedding hbm (DLGpuEmbeddingLookUp + SGDOptimizerSparseUpdate) train the same model to the same bits."""
    import run_wdl
    g = torch.Generator(device=dev).manual_seed(3)
    table_init = torch.randn((20000, 32), generator=g, device=dev) * 0.01
    la, pa, ta = run_wdl.train("hbm", 20000, 32, 128, 10, 0.05, table_init=table_init, device=str(dev))
    for mode in ("step", "step3"):
        lb, pb, tb = run_wdl.train(mode, 20000, 32, 128, 10, 0.05, table_init=table_init, device=str(dev))
        assert la == lb, mode
        assert torch.equal(pa.table, pb.table), mode
        for p, q in zip(ta.parameters(), tb.parameters()):
            assert torch.equal(p, q), mode


# ---- Deep & Cross (examples/ctr/models/dcn_criteo.py) through the same embedding placements -----------------------
def test_dcn_training_matches_pytorch(dev):
    """--model dcn: the cross network + DNN tower over [embeddings | dense features] with the table in HBM, against the
    same loop in plain PyTorch ops."""
    import run_wdl
    g = torch.Generator(device=dev).manual_seed(5)
    table_init = torch.randn((ROWS, WIDTH), generator=g, device=dev) * 0.01
    tower = run_wdl.make_tower("dcn", WIDTH, 0).to(dev)
    table = table_init.clone()
    opt = torch.optim.SGD(tower.parameters(), lr=LR)
    batches = run_wdl.make_batches(min(STEPS + 1, 64), BATCH, ROWS, 0)
    ref = []
    for k in range(STEPS):
        ids, dense, label = (torch.from_numpy(a).to(dev) for a in batches[k % len(batches)])
        idx = ids.long()
        emb = table[idx].clone().requires_grad_(True)
        loss = torch.nn.functional.binary_cross_entropy(tower(dense, emb.reshape(BATCH, run_wdl.NFIELD * WIDTH)), label)
        opt.zero_grad(set_to_none=True)
        loss.backward()
        opt.step()
        table.index_add_(0, idx.reshape(-1), emb.grad.reshape(-1, WIDTH), alpha=-LR)
        ref.append(float(loss.detach()))
    for mode in ("hbm", "queue"):
        losses, param, tw = run_wdl.train(mode, ROWS, WIDTH, BATCH, STEPS, LR, table_init=table_init, device=str(dev),
                                          model="dcn")
        np.testing.assert_allclose(losses, ref, rtol=1e-4)
        torch.testing.assert_close(param.table, table, rtol=RTOL, atol=ATOL)
        for p, q in zip(tw.parameters(), tower.parameters()):
            torch.testing.assert_close(p, q, rtol=1e-4, atol=1e-7)


# ---- the laia-driven loop at world size 2 (run_laia.py:214-236) ---------------------------------------------------------
def _free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _laia_example_worker(rank, world, port, model, local_shared):
    """One rank of `run_wdl.py --laia` on a GPU shared by all ranks (collectives staged through the host under gloo):
    LAIAScheduler -> LAIADataloader tuples (ids, plan) -> HET cache over the row-sharded table ->
    embedding_update_with_push_keys, dense tower all-reduced.  Held to
      * a single-process PyTorch run of the same global step (every rank's samples as the laia MODEL's stream assigns
        them, dense gradients averaged, embedding gradients applied to one table): this rank's losses at rtol 1e-4;
      * oracle/cache_model.py fed with the same key / plan streams (one model per rank, one server): the perf
        counters of this rank's cache, every lookup and every update."""
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "examples", "ctr"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    dev = torch.device("cuda:0")
    import run_wdl
    from herald_amd import laia as hlaia
    from oracle import cache_model, laia_model
    rows, width, mini_bs, steps, lr, limit = 40000, 16, 32, 8, 0.05, 3000

    def staged_a2a(out, inp, out_splits, in_splits, group):
        torch.cuda.current_stream().synchronize()
        o = torch.empty(out.shape, dtype=out.dtype)
        dist.all_to_all_single(o, inp.cpu().contiguous(), out_splits, in_splits, group=group)
        out.copy_(o)

    def staged_allreduce(t):
        c = t.detach().cpu()
        dist.all_reduce(c)
        t.copy_(c)

    g = torch.Generator(device=dev).manual_seed(9)
    table_init = torch.randn((rows, width), generator=g, device=dev) * 0.01
    losses, param, tower, comm = run_wdl.train_laia(model, rows, width, mini_bs, steps, lr, cache="LRU", bound=0,
                                                    cache_limit=limit, device=str(dev), table_init=table_init,
                                                    a2a=staged_a2a, allreduce=staged_allreduce,
                                                    local_shared=local_shared, perf=True)
    # ---- the same global steps in one process, plain PyTorch --------------------------------------------------------
    nsamples = world * mini_bs * max(steps + 2, hlaia.LAIAScheduler.WINDOW + 1)
    ids_all, dense_all, label_all = run_wdl.make_samples(nsamples, rows, 0)
    sparse = ids_all.astype(np.intc)
    batch_num = (nsamples // world) // mini_bs
    epochs = -(-steps * mini_bs * world // nsamples) + 1
    if local_shared:
        streams = [laia_model.TopkSchedulerModel(sparse.astype(np.uint64), epochs, mini_bs, batch_num, world, r, limit,
                                                 hlaia.topk_num_threads(mini_bs), "criteo", hlaia.top_k_table["criteo"]).emit()[r]
                   for r in range(world)]
    else:
        streams = [laia_model.LaiaSchedulerModel(sparse.astype(np.uint64), epochs, mini_bs, batch_num, world, r, limit).emit()
                   for r in range(world)]
    ref_tower = run_wdl.make_tower(model, width, 0).to(dev)
    opt = torch.optim.SGD(ref_tower.parameters(), lr=lr)
    table = table_init.clone()
    ref_losses = []
    for k in range(steps):
        per_rank, pgrads = [], None
        for r in range(world):
            sel = np.asarray(streams[r][2 * k + 1], dtype=np.int64)
            idx = torch.from_numpy(ids_all[sel]).to(dev).long()
            dense = torch.from_numpy(dense_all[sel]).to(dev)
            label = torch.from_numpy(label_all[sel]).to(dev)
            emb = table[idx].clone().requires_grad_(True)
            loss = torch.nn.functional.binary_cross_entropy(ref_tower(dense, emb.reshape(mini_bs, run_wdl.NFIELD * width)), label)
            ref_tower.zero_grad(set_to_none=True)
            loss.backward()
            gs = [p.grad.clone() for p in ref_tower.parameters()]
            pgrads = gs if pgrads is None else [a + b for a, b in zip(pgrads, gs)]
            per_rank.append((idx, emb.grad.clone(), float(loss.detach())))
        for p, gsum in zip(ref_tower.parameters(), pgrads):
            p.grad = gsum / world
        opt.step()
        for idx, eg, _ in per_rank:                               # rank order
            table.index_add_(0, idx.reshape(-1), eg.reshape(-1, width), alpha=-lr)
        ref_losses.append(per_rank[rank][2])
    np.testing.assert_allclose(losses, ref_losses, rtol=1e-4, err_msg="losses of rank %d" % rank)
    for p, q in zip(tower.parameters(), ref_tower.parameters()):
        torch.testing.assert_close(p, q, rtol=1e-3, atol=1e-6)
    # ---- the cache's counters against the cache model, all ranks in lock step ---------------------------------------------
    server = cache_model.Server(table_init.cpu().numpy().copy())
    models = [cache_model.CacheModel("lru", limit, width, server, 0, 0) for _ in range(world)]
    keys = lambda r, k: ids_all[np.asarray(streams[r][2 * k + 1], dtype=np.int64)].reshape(-1).astype(np.uint64)
    zero = np.zeros((mini_bs * run_wdl.NFIELD, width), np.float32)
    for r in range(world):
        models[r].lookup(keys(r, 0))
    for k in range(steps):
        for r in range(world):
            models[r].update_with_push_keys(keys(r, k), np.asarray(streams[r][2 * k + 2], dtype=np.uint64), zero)
        for r in range(world):
            models[r].lookup(keys(r, k + 1))
    got, want = comm.cache.perf, models[rank].perf
    assert len(got) == len(want) == 2 * steps + 1
    for i, (a, b) in enumerate(zip(got, want)):
        for fld in ("type", "num_all", "num_unique", "num_miss", "num_transfered"):
            assert a[fld] == b[fld], (rank, i, fld, a, b)
    assert sum(p["num_transfered"] for p in got if p["type"] == "Push") > 0        # plans and evictions did push lines
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("model,local_shared", [("wdl", False), ("dcn", False), ("dcn", True)])
def test_laia_example_world2_on_one_gpu(dev, model, local_shared):
    import torch.multiprocessing as mp
    mp.spawn(_laia_example_worker, args=(2, _free_port(), model, local_shared), nprocs=2, join=True)


def test_reference_launch_line_and_cache_perf_csv(dev, tmp_path):
    """The reference's own launch vocabulary (/root/reference/examples/ctr/run_hetu.py:546-586: --model wdl_criteo --comm Hybrid
    --cache lru --bound --bsp -b -e -r --cache-perf --val --all) drives run_wdl.py as it is, and --cache-perf leaves the CSV
    run_hetu.py:508-515 writes: csv/hetu_cache<idx>_<rank>.csv, one row per cache call with the perf dict's columns."""
    import csv
    import subprocess
    script = os.path.join(ROOT, "examples", "ctr", "run_wdl.py")
    out = os.path.join(ROOT, "examples", "ctr", "csv", "hetu_cache0_0.csv")
    if os.path.exists(out):
        os.remove(out)
    env = dict(os.environ, PYTHONPATH=ROOT)
    r = subprocess.run([sys.executable, script, "--model", "wdl_criteo", "--comm", "Hybrid", "--cache", "lru", "--bound", "3",
                        "--bsp", "0", "-b", "64", "-e", "32", "-r", "0.1", "--cache-perf", "--val", "--all",
                        "--rows", "100000", "--steps", "12"], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "last 10 steps: loss" in r.stdout
    with open(out) as fh:
        rows = list(csv.DictReader(fh))
    assert len(rows) >= 12
    for col in ("num_all", "num_unique", "num_miss", "num_evict", "num_transfered", "type"):
        assert col in rows[0], (col, list(rows[0]))
    os.remove(out)
