"""ctypes binding of libherald_amd.so (the C-ABI of include/herald_amd.h).

The library is the product: if it is missing or a symbol is absent this module raises -- there is
no CPU or PyTorch fallback anywhere in herald_amd.
"""
import ctypes
import os
import re

_PKG = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_PKG, "libherald_amd.so")
HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "herald_amd.h")

_lib = None


class HeraldAmdError(RuntimeError):
    pass


class DLContext(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("device_type", ctypes.c_int)]


class DLArray(ctypes.Structure):
    """Mirror of the reference's DLArray (src/common/dlarray.h:40-55; python/hetu/ndarray.py:54-60)."""
    _fields_ = [("data", ctypes.c_void_p), ("ctx", DLContext), ("ndim", ctypes.c_int),
                ("shape", ctypes.POINTER(ctypes.c_int64)), ("stride", ctypes.POINTER(ctypes.c_int64))]


class DLStream(ctypes.Structure):
    _fields_ = [("device_id", ctypes.c_int), ("handle", ctypes.c_void_p)]


class CacheRemote(ctypes.Structure):
    """ha_cache_remote (include/herald_amd.h): device addresses of a remote-mode cache's inbox / outbox."""
    _fields_ = [("req_keys", ctypes.c_void_p), ("req_versions", ctypes.c_void_p), ("inbox_pull", ctypes.c_void_p),
                ("inbox_idx", ctypes.c_void_p), ("inbox_versions", ctypes.c_void_p), ("inbox_rows", ctypes.c_void_p),
                ("out_keys", ctypes.c_void_p), ("out_updates", ctypes.c_void_p), ("out_rows", ctypes.c_void_p),
                ("out_capacity", ctypes.c_int64), ("max_batch", ctypes.c_int64)]


class PlanView(ctypes.Structure):
    _fields_ = [("n", ctypes.c_int64), ("n_unique", ctypes.c_void_p), ("keys", ctypes.c_void_p),
                ("sorted", ctypes.c_void_p), ("perm", ctypes.c_void_p), ("inverse", ctypes.c_void_p),
                ("uniq", ctypes.c_void_p), ("counts", ctypes.c_void_p), ("seg", ctypes.c_void_p),
                ("upos", ctypes.c_void_p)]


def declared_symbols(header=None):
    """Names of every function a header of include/ declares (default: herald_amd.h)."""
    text = open(header or HEADER_PATH).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"\(\s*\*\s*\w+\s*\)\s*\([^;{}]*\)\s*;", ";", text)   # function-pointer members of structs
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{}]*\)\s*;", text)
    return sorted(set(n for n in names if n not in ("defined", "int", "void", "float")))


PS_LIB_PATH = os.path.join(_PKG, "libherald_ps.so")
PS_HEADER_PATH = os.path.join(os.path.dirname(_PKG), "include", "herald_ps.h")


def load(build_if_missing=True):
    """Load libherald_amd.so; torch is imported first so both share one HIP runtime."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        import torch  # noqa: F401  (loads torch's libamdhip64.so.7 before ours is resolved)
    except ImportError:
        pass
    if not os.path.exists(LIB_PATH):
        if not build_if_missing:
            raise HeraldAmdError("libherald_amd.so is not built (%s)" % LIB_PATH)
        from . import _build
        _build.build_lib()
    try:
        L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    except OSError as e:
        raise HeraldAmdError("cannot load %s: %s" % (LIB_PATH, e))
    _declare(L)
    _lib = L
    return L


class ShardSlot(ctypes.Structure):
    """ha_shard_slot (include/herald_amd.h): one routed batch of the sharded step."""
    _fields_ = [("world", ctypes.c_int32), ("rank", ctypes.c_int32), ("rcap", ctypes.c_int64), ("n", ctypes.c_int64),
                ("plan_ws", ctypes.c_void_p), ("keys_fixed", ctypes.c_void_p), ("meta_dev", ctypes.c_void_p),
                ("posmap", ctypes.c_void_p), ("rowmap", ctypes.c_void_p), ("counts_host", ctypes.c_void_p),
                ("owner_plan_ws", ctypes.c_void_p)]


def check(rc, what=""):
    if rc != 0:
        msg = load().ha_last_error().decode("utf-8", "replace")
        raise HeraldAmdError("%s failed: %s" % (what or "herald_amd call", msg))


def _declare(L):
    c = ctypes
    vp, i64, f32, sz = c.c_void_p, c.c_int64, c.c_float, c.c_size_t
    L.ha_version.restype = c.c_char_p
    L.ha_last_error.restype = c.c_char_p
    L.ha_device_count.restype = c.c_int
    L.ha_plan_bytes.restype = sz
    L.ha_plan_bytes.argtypes = [i64]
    L.ha_plan_view_of.argtypes = [vp, i64, c.POINTER(PlanView)]
    L.ha_plan_radix_stamps.argtypes = [vp, i64, vp, vp]
    L.ha_pend_bytes.restype = sz
    L.ha_pend_bytes.argtypes = []
    L.ha_plan_handoff_timeout.restype = vp
    L.ha_plan_handoff_timeout.argtypes = [vp]
    L.ha_step_tab_bytes.restype = sz
    L.ha_step_tab_bytes.argtypes = []
    L.ha_step_max_ids.restype = i64
    L.ha_step_max_ids.argtypes = []
    L.ha_qstep_max_ids.restype = i64
    L.ha_qstep_max_ids.argtypes = []
    L.ha_qstep_init.restype = c.c_int
    L.ha_qstep_init.argtypes = []
    L.ha_event_create.restype = vp
    L.ha_event_create.argtypes = []
    L.ha_host_unmap.restype = c.c_int
    L.ha_host_unmap.argtypes = [vp]
    L.ha_qbig_max_ids.restype = i64
    L.ha_qbig_max_ids.argtypes = []
    L.ha_xchg_create.restype = vp
    L.ha_xchg_create.argtypes = [vp, c.c_int, c.c_int]
    L.ha_qbig_plan_bytes.restype = sz
    L.ha_qbig_plan_bytes.argtypes = [i64]
    L.ha_qbig_buckets.restype = c.c_int
    L.ha_qbig_buckets.argtypes = [i64]
    L.ha_qstep_queue_bytes.restype = sz
    L.ha_qstep_queue_bytes.argtypes = [i64, i64]
    L.ha_qstep_queue_header.restype = vp
    L.ha_qstep_queue_header.argtypes = [vp]
    sigs = {
        "ha_gather_f32ids": [vp, i64, i64, vp, i64, vp, vp],
        "ha_gather_u64ids": [vp, i64, i64, vp, i64, vp, vp],
        "ha_gather_u32keys": [vp, i64, i64, vp, i64, vp, vp],
        "ha_scatter_rows_f32ids": [vp, vp, i64, i64, vp, i64, vp],
        "ha_scale_f32": [vp, i64, f32, vp],
        "ha_plan_build_f32ids": [vp, i64, vp, vp],
        "ha_plan_build_u64ids": [vp, i64, vp, vp],
        "ha_plan_build_u32keys": [vp, i64, vp, c.c_int, vp],
        "ha_plan_sort_f32ids": [vp, i64, vp, vp],
        "ha_plan_sort_u32keys": [vp, i64, vp, c.c_int, vp],
        "ha_plan_sort_u64ids": [vp, i64, vp, vp],
        "ha_plan_finish": [vp, i64, vp],
        "ha_plan_build_f32ids_lim": [vp, i64, vp, c.c_uint64, vp],
        "ha_plan_sort_f32ids_lim": [vp, i64, vp, c.c_uint64, vp],
        "ha_plan_build_u64ids_lim": [vp, i64, vp, c.c_uint64, vp],
        "ha_plan_sort_u64ids_lim": [vp, i64, vp, c.c_uint64, vp],
        "ha_plan_export_f32": [vp, i64, vp, vp, vp],
        "ha_dedup_reduce": [vp, i64, vp, i64, vp, vp],
        "ha_apply_mapped": [vp, i64, i64, vp, i64, vp, f32, vp, vp, vp, vp],
        "ha_apply_mapped2": [vp, i64, vp, i64, vp, i64, vp, f32, vp, vp, vp, vp],
        "ha_shard_bucket": [vp, i64, vp, c.c_int, vp, vp, vp],
        "ha_shard_route_f32ids": [vp, i64, vp, vp, c.c_int, vp, vp, vp],
        "ha_shard_route_u64ids": [vp, i64, vp, vp, c.c_int, vp, vp, vp],
        "ha_shard_route_pack_f32ids": [vp, i64, vp, vp, c.c_int, i64, vp, vp, vp],
        "ha_shard_route_pack_u64ids": [vp, i64, vp, vp, c.c_int, i64, vp, vp, vp],
        "ha_shard_route_unpack": [vp, c.c_int, i64, vp, vp, vp],
        "ha_shard_frames_route_f32ids": [vp, i64, vp, vp, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_frames_route_u64ids": [vp, i64, vp, vp, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_frames_pack": [vp, i64, vp, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_frames_unpack": [vp, c.c_int, i64, i64, vp, vp, vp],
        "ha_set_tolerance_mode": [c.c_int],
        "ha_get_tolerance_mode": [],
        "ha_shard_frames_serve_push": [vp, i64, i64, vp, c.c_int, i64, vp, vp, vp],
        "ha_shard_frames_pack_batch": [vp, vp, c.c_int, vp, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_frames_unpack_batch": [vp, c.c_int, c.c_int, i64, i64, vp, vp, vp],
        "ha_shard_frames_pack_batch_sized": [vp, vp, c.c_int, vp, c.c_int, c.c_int, i64, i64, vp, vp, vp, vp, vp, vp],
        "ha_shard_frames_unpack_batch_sized": [vp, c.c_int, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_sized_serve_pull": [vp, i64, i64, vp, c.c_int, c.c_int, i64, vp, vp, vp],
        "ha_gather2_u32map": [vp, i64, vp, i64, i64, vp, i64, vp, vp],
        "ha_shard_sized_serve_push": [vp, i64, i64, vp, c.c_int, c.c_int, i64, vp, i64, vp, vp, vp],
        "ha_push_apply_scaled_finished": [vp, i64, i64, vp, i64, vp, f32, vp],
        "ha_qbig_plan_batch_f32ids": [vp, vp, vp, i64, i64, vp],
        "ha_qbig_plan_batch_u64ids": [vp, vp, vp, i64, i64, vp],
        "ha_qbig_queue_batch": [i64, i64, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp],
        "ha_qbig_apply": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, i64, i64, c.c_uint32, vp, vp, vp],
        "ha_event_destroy": [vp],
        "ha_event_record": [vp, vp],
        "ha_stream_wait_event": [vp, vp],
        "ha_qqueue_batch_epochs": [i64, i64, vp, vp, vp, vp, vp, i64, i64, vp, vp, vp],
        "ha_shard_step_pull": [vp, i64, i64, vp, vp, vp, vp, i64, vp, vp],
        "ha_shard_step_push": [vp, i64, i64, vp, vp, vp, i64, vp, vp, f32, vp],
        "ha_shard_step": [vp, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, vp],
        "ha_shard_steps": [vp, i64, i64, i64, vp, vp, vp, vp, i64, vp, i64, vp, vp, vp, f32, vp],
        "ha_qapply_steps_sync": [vp, i64, i64, f32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "ha_qapply_steps_counts": [vp, i64, i64, f32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "ha_qapply_sync": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, i64, i64, c.c_uint32, vp, vp, vp],
        "ha_xchg_available": [],
        "ha_xchg_unique_id": [vp],
        "ha_xchg_destroy": [vp],
        "ha_xchg_bytes": [vp, vp, vp, vp, vp, vp],
        "ha_xchg_rows": [vp, vp, vp, vp, vp, i64, vp],
        "ha_qbig_plan_view": [vp, i64, vp, vp, vp, vp, vp, vp, vp],
        "ha_plan_build_batch_f32ids_lim": [vp, vp, vp, c.c_int, c.c_uint64, vp],
        "ha_plan_sort_batch_f32ids_lim": [vp, vp, vp, c.c_int, c.c_uint64, vp],
        "ha_plan_sort_batch_u64ids_lim": [vp, vp, vp, c.c_int, c.c_uint64, vp],
        "ha_plan_build_batch_u64ids_lim": [vp, vp, vp, c.c_int, c.c_uint64, vp],
        "ha_shard_frames_serve_pull": [vp, i64, i64, vp, c.c_int, i64, i64, vp, vp, vp, vp],
        "ha_shard_serve_push": [vp, i64, i64, vp, i64, vp, vp, vp],
        "ha_dedup_reduce_scaled": [vp, i64, vp, i64, f32, vp, vp],
        "ha_debug_apply_timeline": [vp, i64, i64, vp, i64, vp, f32, vp, vp],
        "ha_sgd_apply": [vp, i64, i64, vp, i64, vp, f32, vp],
        "ha_sgd_apply_finished": [vp, i64, i64, vp, i64, vp, f32, vp],
        "ha_push_apply": [vp, i64, i64, vp, i64, vp, vp],
        "ha_sgd_sparse_update_f32ids": [vp, i64, i64, vp, i64, vp, f32, vp],
        "ha_lookup_sort_f32ids": [vp, i64, i64, vp, i64, vp, vp, vp],
        "ha_lookup_sort_u64ids": [vp, i64, i64, vp, i64, vp, vp, vp],
        "ha_sgd_apply_finish": [vp, i64, i64, vp, i64, vp, f32, vp],
        "ha_push_apply_finish": [vp, i64, i64, vp, i64, vp, vp],
        "ha_sgd_apply_finish_prefetch_f32ids": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp],
        "ha_sparse_opt_fused_f32ids": [c.c_int, vp, i64, i64, vp, i64, vp, vp, vp, vp, vp, vp],
        "ha_pend_reset": [vp, vp],
        "ha_lookup_sort_pend_f32ids": [vp, i64, i64, vp, i64, vp, vp, vp, vp],
        "ha_lookup_sort_pend_u64ids": [vp, i64, i64, vp, i64, vp, vp, vp, vp],
        "ha_sgd_push_pull_f32ids": [vp, i64, i64, vp, i64, vp, f32, vp, vp, i64, vp, vp, vp, vp],
        "ha_sgd_push_pull_u64ids": [vp, i64, i64, vp, i64, vp, f32, vp, vp, i64, vp, vp, vp, vp],
        "ha_step_tab_reset": [vp, vp],
        "ha_step_f32ids": [vp, i64, i64, vp, i64, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp],
        "ha_step_u64ids": [vp, i64, i64, vp, i64, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp, vp, vp],
        "ha_qstep_f32ids": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp],
        "ha_qstep_u64ids": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, vp, i64, vp, i64, vp, i64, vp, vp],
        "ha_qprep_f32ids": [i64, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp],
        "ha_qprep_u64ids": [i64, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp],
        "ha_qplan_batch_f32ids": [vp, vp, vp, i64, vp],
        "ha_qplan_batch_u64ids": [vp, vp, vp, i64, vp],
        "ha_qqueue_batch": [i64, i64, vp, vp, vp, vp, vp, i64, i64, vp],
        "ha_stream_gate": [vp, vp],
        "ha_debug_occupy": [i64, i64, i64, i64, vp],
        "ha_qapply": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, i64, vp],
        "ha_qapply_sized": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, i64, i64, vp],
        "ha_qapply_steps": [vp, i64, i64, f32, i64, i64, vp, vp, vp, vp, vp, vp, vp, vp, vp],
        "ha_qqueue_batch_counts": [i64, i64, vp, vp, vp, vp, vp, i64, i64, vp, vp],
        "ha_debug_qprep_f32ids": [i64, i64, vp, i64, vp, vp, i64, vp, i64, vp, i64, vp, vp],
        "ha_debug_qapply": [vp, i64, i64, vp, i64, vp, f32, vp, i64, vp, vp, i64, vp, vp],
        "ha_debug_step_fwd_timeline": [vp, i64, i64, vp, i64, vp, f32, vp, vp, i64, vp, vp, vp, i64, vp, vp, i64, vp,
                                       vp, vp, vp],
    }
    for name, args in sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = c.c_int
    L.ha_cache_create.restype = vp
    L.ha_cache_create.argtypes = [c.c_int, i64, i64, i64, i64]
    L.ha_cache_destroy.restype = None
    L.ha_cache_destroy.argtypes = [vp]
    for name in ("ha_cache_data", "ha_cache_grad"):
        getattr(L, name).restype = vp
        getattr(L, name).argtypes = [vp]
    for name in ("ha_cache_limit", "ha_cache_width", "ha_cache_fused_updates"):
        getattr(L, name).restype = i64
        getattr(L, name).argtypes = [vp]
    cache_sigs = {
        "ha_cache_set_bounds": [vp, i64, i64],
        "ha_cache_set_bypass": [vp, c.c_int],
        "ha_cache_bind_store": [vp, vp, vp, i64, i64],
        "ha_cache_lookup": [vp, vp, c.c_int, i64, vp, vp],
        "ha_cache_sort_ahead": [vp, vp, c.c_int, i64, vp],
        "ha_cache_lookup_presorted": [vp, vp, c.c_int, i64, vp, vp],
        "ha_cache_sort_ahead_batch": [vp, vp, c.c_int, vp, c.c_int, vp],
        "ha_cache_plan_block": [vp, vp, c.c_int, vp, c.c_int, vp, vp],
        "ha_cache_lookup_planned": [vp, i64, vp, vp],
        "ha_cache_update_planned": [vp, i64, vp, vp],
        "ha_cache_plan_pending": [vp],
        "ha_cache_run_planned_pairs": [vp, c.c_int, vp, vp, vp, vp],
        "ha_cache_update": [vp, vp, c.c_int, i64, vp, vp],
        "ha_cache_update_same_keys": [vp, i64, vp, vp],
        "ha_cache_update_with_push_keys": [vp, vp, c.c_int, i64, vp, c.c_int, i64, vp, vp],
        "ha_cache_push_pull": [vp, vp, c.c_int, i64, vp, vp, c.c_int, i64, vp, vp],
        "ha_cache_perf": [vp, vp, vp],
        "ha_cache_state": [vp, vp, vp],
        "ha_cache_phase_times": [vp, vp, vp],
        "ha_cache_set_timing": [vp, c.c_int],
        "ha_cache_stage_times": [vp, vp],
        "ha_cache_snapshot": [vp, i64, vp, vp, vp, vp, vp, vp, vp],
        "ha_cache_set_line": [vp, i64, i64, vp, vp],
        "ha_cache_set_remote": [vp],
        "ha_cache_remote_buffers": [vp, vp],
        "ha_cache_lookup_begin": [vp, vp, c.c_int, i64, vp, vp],
        "ha_cache_lookup_finish": [vp, i64, vp, vp],
        "ha_cache_outbox_count": [vp, vp, vp],
        "ha_cache_outbox_pad": [vp, i64],
        "ha_store_count_valid": [vp, i64, i64, vp, vp],
        "ha_cache_push_pull_begin": [vp, vp, c.c_int, i64, vp, c.c_int, i64, vp, vp, vp],
        "ha_cache_push_pull_finish": [vp, vp, vp],
        "ha_store_serve_sync": [vp, vp, i64, i64, vp, vp, i64, i64, vp, vp, vp, vp, vp, vp, vp],
        "ha_store_add_versions": [vp, i64, vp, vp, i64, vp],
        "ha_store_push_distinct": [vp, vp, i64, i64, vp, vp, vp, i64, vp],
    }
    for name, args in cache_sigs.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = c.c_int
    L.ha_laia_create.restype = vp
    L.ha_laia_create.argtypes = [vp, i64, i64, i64, i64, i64, i64]
    L.ha_laia_destroy.restype = None
    L.ha_laia_destroy.argtypes = [vp]
    L.ha_laia_next.restype = c.c_int
    L.ha_laia_next.argtypes = [vp, i64, i64, vp, vp, i64, vp]
    L.ha_laia_hint_next.restype = c.c_int
    L.ha_laia_hint_next.argtypes = [vp, i64]
    L.ha_laia_next_for_rank.restype = c.c_int
    L.ha_laia_next_for_rank.argtypes = [vp, i64, i64, i64, vp, vp, i64, vp]
    L.ha_laia_snapshot_keys.restype = i64
    L.ha_laia_snapshot_keys.argtypes = [vp, i64, vp, i64]
    L.ha_laia_next_topk.restype = c.c_int
    L.ha_laia_next_topk.argtypes = [vp, i64, i64, vp, i64, i64, vp, vp, i64, vp]
    L.ha_laia_counters.restype = c.c_int
    L.ha_laia_counters.argtypes = [vp, vp]
    L.ha_laia_timing.restype = c.c_int
    L.ha_laia_timing.argtypes = [vp, vp]
    L.ha_laia_timing_device.restype = c.c_int
    L.ha_laia_timing_device.argtypes = [vp, vp]
    L.ha_laia_on_device.restype = c.c_int
    L.ha_laia_on_device.argtypes = [vp]
    L.ha_shm_ring_open.restype = vp
    L.ha_shm_ring_open.argtypes = [c.c_char_p, c.c_int, i64]
    L.ha_shm_ring_close.restype = None
    L.ha_shm_ring_close.argtypes = [vp]
    L.ha_shm_ring_send.restype = c.c_int
    L.ha_shm_ring_send.argtypes = [vp, vp, i64]
    L.ha_shm_ring_recv.restype = i64
    L.ha_shm_ring_recv.argtypes = [vp, vp, i64, vp]
    L.ha_shm_ring_pending_words.restype = i64
    L.ha_shm_ring_pending_words.argtypes = [vp]
    A, S = c.POINTER(DLArray), c.POINTER(DLStream)
    dl = {
        "DLGpuEmbeddingLookUp": [A, A, A, S],
        "DLGpuEmbeddingLookUp_Gradient": [A, A, A, S],
        "IndexedSlicesOneSideAdd": [A, A, A, S],
        "DeduplicateIndexedSlices": [A, A, A, S],
        "IndexedSlices2Dense": [A, A, A, S],
        "SGDOptimizerSparseUpdate": [A, A, A, f32, S],
        "AdaGradOptimizerSparseUpdate": [A, A, A, A, f32, f32, S],
        "AdamOptimizerSparseUpdate": [A, A, A, A, A, f32, f32, f32, f32, f32, f32, S],
        "AdamWOptimizerSparseUpdate": [A, A, A, A, A, f32, f32, f32, f32, f32, f32, f32, S],
        "AddL2RegularizationSparse": [A, A, A, f32, S],
        "MomentumOptimizerSparseUpdate": [A, A, A, A, f32, f32, c.c_bool, S],
        "LambOptimizerSparseUpdate": [A, A, A, A, A, f32, f32, f32, f32, f32, f32, f32, S],
        "cpu_EmbeddingLookup": [A, A, A],
        "cpu_SGDOptimizerSparseUpdate": [A, A, A, f32],
    }
    for name, args in dl.items():
        fn = getattr(L, name)
        fn.argtypes = args
        fn.restype = c.c_int
