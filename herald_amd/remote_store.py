"""Stores a HET cache can sit in front of without addressing them (herald_amd.cache `bind_remote`).

The reference's cache reaches its table only through two RPCs (src/hetu_cache/src/hetu_client.cc:6-39):
    syncEmbedding(keys, versions, bound) -> rows + versions of the lines that lag   (kSyncEmbedding)
    pushEmbedding(keys, gradient rows, update counts)                               (kPushEmbedding)
served by ps-lite/src/PSFhandle_embedding.cc:5-64 on the shard that owns each key (PSAgent.h:537-627 splits
the sorted keys by shard).  A store here implements the same two calls on device tensors:

    sync(keys, versions, bound, pull, idx, ver_out, rows_out)
        keys int32[u] (uint32 row ids, ascending), versions int64[u]; fills pull[u] (0/1), and for pulled
        keys ver_out[u] = server version and rows_out[idx[u], :] = the row.
    push(keys, updates, rows)
        keys int32[m] (0xFFFFFFFF = entry not pushed), updates int32[m], rows float32[m, width]:
        table[key] += rows[j]; version[key] += updates[j], entries of one caller in list order.

LocalStore    the table shard lives in this GPU's HBM (the owner side of every other store).
ShardedStore  row-range shards over the ranks of a process group: requests and pushes travel with the
              all-to-all of herald_amd.sharded (RCCL over xGMI), owners serve them with LocalStore.
HostStore     the table lives in pinned host DRAM (the cold tier of BASELINE configs[4]); versions stay in
              HBM; rows are staged by the GPU over PCIe on a copy stream.
All arithmetic is in libherald_amd.so (ha_store_serve_sync, ha_store_add_versions, ha_shard_serve_push).
"""
import ctypes

import torch
import torch.distributed as dist

from . import _lib
from ._lib import check
from .sharded import partition


def _p(t):
    return ctypes.c_void_p(t.data_ptr())


def _s():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


class LocalStore:
    """A table shard (rows [0, R) in shard-local numbering) and its per-row versions on this device."""

    host_counts = False   # served on this device: request / outbox are taken padded, no host read-back (cache.py)

    def __init__(self, table, versions=None):
        assert table.dtype == torch.float32 and table.is_contiguous() and table.dim() == 2
        self.table = table
        self.rows, self.width = int(table.shape[0]), int(table.shape[1])
        self.device = versions.device if versions is not None else table.device
        if versions is None:
            versions = torch.zeros(self.rows, dtype=torch.int64, device=self.device)
        assert versions.dtype == torch.int64 and versions.numel() == self.rows and versions.is_cuda
        self.versions = versions
        self._L = _lib.load()
        self._count = torch.zeros(1, dtype=torch.int64, device=self.device)
        torch.cuda.current_stream(self.device).synchronize()   # (zero fills: the store's calls come on streams of the caller's choice)
        self._plan_ws = None
        self._plan_cap = 0

    def sync(self, keys, versions, bound, pull, idx, ver_out, rows_out):
        m = keys.numel()
        check(self._L.ha_store_serve_sync(_p(self.table), _p(self.versions), self.rows, self.width, _p(keys),
                                          _p(versions), m, int(bound), _p(pull), _p(idx), _p(ver_out), _p(rows_out),
                                          _p(self._count), None, _s()), "ha_store_serve_sync")
        return self._count          # device int64[1]: number of rows written to rows_out

    def push(self, keys, updates, rows, distinct=False):
        """distinct=True: the caller vouches that the live keys of the list are pairwise distinct (the outbox of a
        two-launch cache update): one launch, no sort."""
        m = keys.numel()
        if m == 0:
            return
        if distinct:
            check(self._L.ha_store_push_distinct(_p(self.table), _p(self.versions), self.rows, self.width, _p(keys),
                                                 _p(updates), _p(rows), m, _s()), "ha_store_push_distinct")
            return
        if self._plan_cap < m:
            self._plan_cap = m * 5 // 4 + 64
            self._plan_ws = torch.empty(self._L.ha_plan_bytes(self._plan_cap), dtype=torch.uint8, device=self.device)
        # rows first (list order per key, kNoPush = 0xFFFFFFFF >= rows is skipped), then the versions
        check(self._L.ha_shard_serve_push(_p(self.table), self.rows, self.width, _p(keys), m, _p(rows),
                                          _p(self._plan_ws), _s()), "ha_shard_serve_push")
        check(self._L.ha_store_add_versions(_p(self.versions), self.rows, _p(keys), _p(updates), m, _s()),
              "ha_store_add_versions")


class HostStore(LocalStore):
    """Cold tier: the table in pinned (device-visible) host memory, the versions in HBM.  The owner-side
    kernels read / update the rows over PCIe; `stream` (default: the caller's stream) carries them -- pass a copy
    stream when the caller has work of its own to overlap the staging with."""

    def __init__(self, rows, width, device, table=None, stream=None):
        if table is None:
            table = torch.zeros((rows, width), dtype=torch.float32, pin_memory=True)
        assert table.is_pinned() and tuple(table.shape) == (rows, width)
        versions = torch.zeros(rows, dtype=torch.int64, device=device)
        super().__init__(table, versions)
        # stream=None: the owner-side kernels run on the caller's stream.  (A store-owned copy stream only pays when the
        # caller has other work to overlap: the cache needs the inbox at once, and the two cross-stream dependencies per
        # call cost more than the kernels between them.)
        self.copy_stream = stream
        # device-side traffic counters (no host sync on the path): keys synced, rows pulled, lines pushed
        self._acc = torch.zeros(3, dtype=torch.int64, device=device)
        torch.cuda.current_stream(torch.device(device)).synchronize()

    def _on_store_stream(self):
        import contextlib
        if self.copy_stream is None:
            return contextlib.nullcontext(), None
        cur = torch.cuda.current_stream()
        self.copy_stream.wait_stream(cur)
        return torch.cuda.stream(self.copy_stream), cur

    def sync(self, keys, versions, bound, pull, idx, ver_out, rows_out):
        ctx, cur = self._on_store_stream()
        with ctx:
            cnt = super().sync(keys, versions, bound, pull, idx, ver_out, rows_out)
            # keys actually asked for (the request is padded with 0xFFFFFFFF) and rows pulled, on the device
            check(self._L.ha_store_count_valid(_p(keys), keys.numel(), self.rows, _p(self._acc[0:1]), _s()),
                  "ha_store_count_valid")
            self._acc[1:2] += cnt
        if cur is not None:
            cur.wait_stream(self.copy_stream)
        return cnt

    def push(self, keys, updates, rows, distinct=False):
        ctx, cur = self._on_store_stream()
        with ctx:
            super().push(keys, updates, rows, distinct)
            check(self._L.ha_store_count_valid(_p(keys), keys.numel(), self.rows, _p(self._acc[2:3]), _s()),
                  "ha_store_count_valid")
        if cur is not None:
            cur.wait_stream(self.copy_stream)

    def traffic(self, reset=False):
        """{'keys_synced', 'rows_pulled', 'lines_pushed', 'pcie_bytes'} since creation / the last reset
        (host sync).  A pulled row crosses PCIe once, a pushed line is a read-modify-write of its row."""
        a = self._acc.cpu().tolist()
        if reset:
            self._acc.zero_()
        return {"keys_synced": a[0], "rows_pulled": a[1], "lines_pushed": a[2],
                "pcie_bytes": 4 * self.width * (a[1] + 2 * a[2])}


class ShardedStore:
    host_counts = True    # the exchanges are sized on the host: the cache reads n_unique / the outbox count back

    """Row-range shards (AveragePartitioner ranges, ps-lite/include/ps/partitioner.h:46-57) over the ranks of a
    process group; every rank owns `local` (a LocalStore or HostStore of its range) and is a client of all.
    One sync = counts all-to-all (host read-back), keys + versions out, decisions + versions back, pulled-row
    counts (host read-back), rows back.  One push = counts, keys + update counts + gradient rows out, the
    owner applies the W lists in rank order.  `a2a` as in ShardedEmbedding (host-staged exchange for tests)."""

    def __init__(self, rows, width, device, local, group=None, a2a=None):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.rows, self.width = int(rows), int(width)
        self.device = torch.device(device)
        self.starts = partition(rows, self.world)
        assert local.rows == self.starts[self.rank + 1] - self.starts[self.rank] and local.width == width
        self.local = local
        self._a2a_fn = a2a
        self._starts_dev = torch.tensor(self.starts, dtype=torch.int64, device=self.device)
        self.stats = {"xgmi_bytes_out": 0, "xgmi_bytes_in": 0, "rows_pulled": 0, "keys_synced": 0, "lines_pushed": 0}

    # -- exchange helpers ---------------------------------------------------------------------------------
    def _a2a(self, out, inp, out_splits, in_splits):
        if self.world == 1:
            out.copy_(inp)
        elif self._a2a_fn is not None:
            self._a2a_fn(out, inp, out_splits, in_splits, self.group)
        else:
            dist.all_to_all_single(out, inp, out_splits, in_splits, group=self.group)
        return out

    def _counts(self, sorted_keys):
        """Per-owner counts of an ascending key list (keys >= rows, e.g. 0xFFFFFFFF, belong to nobody),
        the counts the peers send here, and both as host lists (one read-back)."""
        k64 = sorted_keys.to(torch.int64) & 0xFFFFFFFF
        bounds = torch.searchsorted(k64, self._starts_dev)          # first key >= start[g], g = 0..W
        send = (bounds[1:] - bounds[:-1]).contiguous()
        recv = torch.empty_like(send)
        self._a2a(recv, send, None, None)
        both = torch.cat([send, recv]).cpu().tolist()               # the host needs the split sizes
        w = self.world
        return both[:w], both[w:], k64

    def _owner_local(self, k64, send_cnt):
        """shard-local row ids (int32 carrying uint32) of an ascending key list."""
        owner = torch.repeat_interleave(torch.arange(self.world, device=self.device),
                                        torch.tensor(send_cnt, device=self.device))
        return (k64[:owner.numel()] - self._starts_dev[owner]).to(torch.int32)

    # -- kSyncEmbedding ---------------------------------------------------------------------------------------
    def sync(self, keys, versions, bound, pull, idx, ver_out, rows_out):
        u, w, dev = keys.numel(), self.world, self.device
        send_cnt, recv_cnt, k64 = self._counts(keys)
        m = sum(recv_cnt)
        lk_send = self._owner_local(k64, send_cnt)
        keys_recv = torch.empty(m, dtype=torch.int32, device=dev)
        vers_recv = torch.empty(m, dtype=torch.int64, device=dev)
        self._a2a(keys_recv, lk_send, recv_cnt, send_cnt)
        self._a2a(vers_recv, versions.contiguous(), recv_cnt, send_cnt)
        # owner: decisions for the W request lists, rows packed in request order (grouped by requester)
        o_pull = torch.empty(m, dtype=torch.int32, device=dev)
        o_idx = torch.empty(m, dtype=torch.int32, device=dev)
        o_ver = torch.empty(m, dtype=torch.int64, device=dev)
        o_rows = torch.empty((max(m, 1), self.width), dtype=torch.float32, device=dev)
        self.local.sync(keys_recv, vers_recv, bound, o_pull, o_idx, o_ver, o_rows)
        # rows per requester = sum of its decisions; tell every requester how many rows come back
        seg = torch.repeat_interleave(torch.arange(w, device=dev), torch.tensor(recv_cnt, device=dev))
        rows_to = torch.zeros(w, dtype=torch.int64, device=dev).index_add_(0, seg, o_pull.to(torch.int64))
        rows_from = torch.empty_like(rows_to)
        self._a2a(rows_from, rows_to, None, None)
        both = torch.cat([rows_to, rows_from]).cpu().tolist()
        rows_to_l, rows_from_l = both[:w], both[w:]
        # answers: decisions and versions in request order, rows in the order of the pulled requests
        self._a2a(pull, o_pull, send_cnt, recv_cnt)
        self._a2a(ver_out, o_ver, send_cnt, recv_cnt)
        npulled = sum(rows_from_l)
        self._a2a(rows_out[:npulled], o_rows[:sum(rows_to_l)], rows_from_l, rows_to_l)
        # rows arrive owner by owner, each owner's in request order == ascending key order == the order of
        # the pulled keys of this request: idx = exclusive prefix of the decisions
        idx.copy_((torch.cumsum(pull, 0) - pull).to(torch.int32))
        self.stats["keys_synced"] += u
        self.stats["rows_pulled"] += npulled
        out_keys = sum(c for g, c in enumerate(send_cnt) if g != self.rank)
        in_rows = sum(c for g, c in enumerate(rows_from_l) if g != self.rank)
        self.stats["xgmi_bytes_out"] += out_keys * 12
        self.stats["xgmi_bytes_in"] += out_keys * 12 + in_rows * 4 * self.width

    # -- kPushEmbedding -----------------------------------------------------------------------------------------
    def push(self, keys, updates, rows, distinct=False):
        # (distinct is a single-store shortcut; the W lists an owner receives may repeat a key across senders)
        m, w, dev = keys.numel(), self.world, self.device
        # stable sort by key: groups by owner, keeps the list order of equal keys (the batch's line before
        # the older evicted ones), moves the not-pushed entries (0xFFFFFFFF) behind every owner's range
        k64 = keys.to(torch.int64) & 0xFFFFFFFF
        order = torch.sort(k64, stable=True).indices
        sk = k64[order]
        send_cnt, recv_cnt, _ = self._counts(sk.to(torch.int32))
        nsend, mrecv = sum(send_cnt), sum(recv_cnt)
        lk_send = self._owner_local(sk, send_cnt)
        upd_send = updates[order[:nsend]].contiguous()
        rows_send = rows[order[:nsend]].contiguous()
        keys_recv = torch.empty(mrecv, dtype=torch.int32, device=dev)
        upd_recv = torch.empty(mrecv, dtype=torch.int32, device=dev)
        rows_recv = torch.empty((max(mrecv, 1), self.width), dtype=torch.float32, device=dev)
        self._a2a(keys_recv, lk_send, recv_cnt, send_cnt)
        self._a2a(upd_recv, upd_send, recv_cnt, send_cnt)
        self._a2a(rows_recv[:mrecv], rows_send, recv_cnt, send_cnt)
        self.local.push(keys_recv, upd_recv, rows_recv[:mrecv])     # the W lists in rank order
        self.stats["lines_pushed"] += nsend
        out_lines = sum(c for g, c in enumerate(send_cnt) if g != self.rank)
        self.stats["xgmi_bytes_out"] += out_lines * (8 + 4 * self.width)
