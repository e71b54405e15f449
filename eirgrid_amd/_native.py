"""ctypes binding of libeirgrid_hip.so (the C ABI declared in include/eirgrid_hip.h).

The library is built in-tree (eirgrid_amd/libeirgrid_hip.so) by `eirgrid_amd.build.build()`; importing this module
fails loudly if it is missing — there is no Python or CPU implementation to fall back to.
"""
from __future__ import annotations

import ctypes as C
import os

YEARS, N_ACTIONS, N_DEFICIT, N_COUNTS, N_TYPES = 26, 61, 15, 21, 15
GRID, CELLS, YEARLY_FIELDS = 51, 2601, 21
MAX_GENS, MAX_OFFSETS, RUN_CAP, DEF_CAP, ACT_CAP, ONCHIP_GENS = 4096, 4096, 4096, 4096, 4096, 512
STATS_LEN = 8 + 2 * YEARS * N_ACTIONS + YEARS * N_DEFICIT
CANDIDATE_BYTES = 8 + 8 + 32 + 4 * YEARS + 4 * YEARS + RUN_CAP + DEF_CAP
PACKET_BYTES = 8 * STATS_LEN + CANDIDATE_BYTES

EG_OK, EG_ERR_NO_DEVICE, EG_ERR_BAD_ARG, EG_ERR_HIP, EG_ERR_UNSUPPORTED, EG_ERR_NOMEM, EG_ERR_INTERNAL = 0, -1, -2, -3, -4, -5, -6
EG_EP_OK, EG_EP_OVERFLOW, EG_EP_NO_LOCATION, EG_EP_INTERNAL = 0, -1, -2, -3

# EIRGRID_LIB selects another build of the same library (only used for the -DEG_STAMPS diagnostic build)
LIB_PATH = os.environ.get("EIRGRID_LIB") or os.path.join(os.path.dirname(os.path.abspath(__file__)), "libeirgrid_hip.so")

_dp = C.POINTER(C.c_double)
_u8p = C.POINTER(C.c_uint8)
_u16p = C.POINTER(C.c_uint16)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u64p = C.POINTER(C.c_uint64)


class EgWorld(C.Structure):
    _fields_ = [("n_settlements", C.c_int32), ("settlement_x", _dp), ("settlement_y", _dp), ("settlement_pop", _u32p),
                ("n_existing", C.c_int32), ("existing_x", _dp), ("existing_y", _dp), ("existing_type", _i32p),
                ("existing_capacity_mw", _dp), ("n_coast", C.c_int32), ("coast_x", _dp), ("coast_y", _dp),
                ("existing_operational_at_start", C.c_int32)]


class EgOpts(C.Structure):
    _fields_ = [("enable_energy_sales", C.c_int32), ("enable_construction_delays", C.c_int32), ("write_yearly", C.c_int32)]


class EgPolicySnapshot(C.Structure):
    _fields_ = [("weights", _dp), ("deficit_weights", _dp), ("count_weights", _dp),
                ("learning_rate", C.c_double), ("exploration_rate", C.c_double),
                ("iterations_without_improvement", C.c_uint32), ("has_best", C.c_int32),
                ("best_metrics", C.c_double * 4),
                ("best_count", _i32p), ("best_actions", _u8p), ("best_deficit_count", _i32p), ("best_deficit_actions", _u8p)]


class EgEpisodeOut(C.Structure):
    _fields_ = [("metrics", _dp), ("yearly", _dp), ("status", _i32p), ("n_run", _i32p), ("n_def", _i32p), ("n_act", _i32p),
                ("run_log", _u8p), ("def_log", _u8p), ("act_log", _u8p), ("n_gens", _i32p), ("gen_cell", _u16p),
                ("gen_pack", _u16p), ("n_offsets", _i32p), ("off_pack", _u16p), ("n_draws", _u64p), ("bytes_moved", _dp),
                ("n_chunks", _u32p)]


# every symbol include/eirgrid_hip.h declares
EXPORTS = [
    "eg_build_hash", "eg_last_error", "eg_device_count", "eg_create", "eg_destroy", "eg_rollout_batch", "eg_upload_snapshot",
    "eg_rollout_launch", "eg_rollout_launch_update", "eg_sync", "eg_last_batch_size", "eg_policy_hold", "eg_policy_rewind", "eg_replay_hoist", "eg_replay_hoist_stats", "eg_debug_hoist_stamps", "eg_fetch", "eg_timing_reset", "eg_timing_read", "eg_timing_read_grids", "eg_memory_report", "eg_update_stats", "eg_fetch_scores",
    "eg_fetch_episode_lists", "eg_fetch_record", "eg_fetch_best_run", "eg_best_result_track", "eg_fetch_best_result", "eg_evaluate_action_impact", "eg_place", "eg_find_suitable_location", "eg_debug_fill_lds", "eg_debug_occupy", "eg_policy_apply_reduced", "eg_policy_apply_packet", "eg_train_step",
    "eg_policy_push", "eg_device_rollout", "eg_device_apply", "eg_device_step", "eg_policy_pull",
    "eg_host_tables_create", "eg_host_tables_free", "eg_host_tables_f64", "eg_host_tables_i32",
    "eg_policy_new", "eg_policy_free", "eg_policy_snapshot_view", "eg_policy_get_tables", "eg_policy_set_tables",
    "eg_policy_get_scalar", "eg_policy_set_scalar", "eg_policy_get_list", "eg_policy_apply_episode", "eg_score_metrics",
    "eg_policy_save_json", "eg_policy_load_json", "eg_policy_append_weight_history", "eg_policy_export_improvement_csv", "eg_export_summary_csv", "eg_export_run_details",
]

_lib = None


def source_hash() -> str:
    """The hash csrc/Makefile stamps into the library (eg_build_hash): sha256 over the sources it is built from, in the
    Makefile's order (make's $(sort) = byte order of the relative paths)."""
    import glob
    import hashlib
    csrc = os.path.join(os.path.dirname(os.path.abspath(__file__)), "csrc")
    rel = [os.path.basename(f) for pat in ("*.hip", "*.cpp", "*.h") for f in glob.glob(os.path.join(csrc, pat))]
    rel += ["../../include/" + os.path.basename(f) for f in glob.glob(os.path.join(csrc, "..", "..", "include", "*.h"))]
    rel = sorted(set(rel + ["Makefile"]) - {"eg_build_hash.cpp"}, key=lambda r: r.encode())
    h = hashlib.sha256()
    for r in rel:
        h.update(open(os.path.join(csrc, r), "rb").read())
    return h.hexdigest()[:16]


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} is missing: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(hipcc --offload-arch=gfx950).  eirgrid_amd has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.eg_build_hash.restype = C.c_char_p
    if not os.environ.get("EIRGRID_LIB"):      # a stale or mixed binary (e.g. a prebuilt .so shipped next to edited sources) fails loudly
        built, tree = L.eg_build_hash().decode(), source_hash()
        if built != tree:
            raise ImportError(f"{LIB_PATH} was built from other sources (library {built}, tree {tree}): run `make -C eirgrid_amd/csrc`")
    L.eg_last_error.restype = C.c_char_p
    L.eg_device_count.restype = C.c_int32
    L.eg_create.restype = C.c_void_p
    L.eg_create.argtypes = [C.c_int32, C.POINTER(EgWorld)]
    L.eg_destroy.argtypes = [C.c_void_p]
    L.eg_rollout_batch.restype = C.c_int32
    L.eg_rollout_batch.argtypes = [C.c_void_p, C.POINTER(EgPolicySnapshot), C.POINTER(EgOpts), C.c_uint64, C.c_uint64,
                                   C.c_uint32, _u8p, C.POINTER(EgEpisodeOut)]
    L.eg_upload_snapshot.restype = C.c_int32
    L.eg_upload_snapshot.argtypes = [C.c_void_p, C.POINTER(EgPolicySnapshot), C.POINTER(EgOpts)]
    L.eg_rollout_launch.restype = C.c_int32
    L.eg_rollout_launch.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, _u8p]
    L.eg_rollout_launch_update.restype = C.c_int32
    L.eg_rollout_launch_update.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, _u8p, C.c_void_p]
    L.eg_sync.restype = C.c_int32
    L.eg_sync.argtypes = [C.c_void_p]
    L.eg_last_batch_size.restype = C.c_uint32
    for f in (L.eg_policy_hold, L.eg_policy_rewind):
        f.restype = C.c_int32; f.argtypes = [C.c_void_p]
    L.eg_last_batch_size.argtypes = [C.c_void_p]
    if hasattr(L, "eg_replay_hoist") or not os.environ.get("EIRGRID_LIB"):      # (EIRGRID_LIB: an A/B build may predate it)
        L.eg_replay_hoist.restype = C.c_int32
        L.eg_replay_hoist.argtypes = [C.c_void_p, C.c_int32]
        L.eg_replay_hoist_stats.restype = C.c_int32
        L.eg_replay_hoist_stats.argtypes = [C.c_void_p, _u64p, _i32p]
        L.eg_debug_hoist_stamps.restype = C.c_int32
        L.eg_debug_hoist_stamps.argtypes = [C.c_void_p, _u64p]
    L.eg_fetch.restype = C.c_int32
    L.eg_fetch.argtypes = [C.c_void_p, C.POINTER(EgEpisodeOut)]
    L.eg_timing_reset.restype = C.c_int32
    L.eg_timing_reset.argtypes = [C.c_void_p]
    L.eg_timing_read.restype = C.c_int32
    L.eg_timing_read.argtypes = [C.c_void_p, _dp, _i32p]
    if hasattr(L, "eg_memory_report") or not os.environ.get("EIRGRID_LIB"):      # (EIRGRID_LIB: an A/B build may predate it)
        L.eg_memory_report.restype = C.c_int32
        L.eg_memory_report.argtypes = [C.c_void_p, _u64p, _u64p, _u64p]
    L.eg_timing_read_grids.restype = C.c_int32
    L.eg_timing_read_grids.argtypes = [C.c_void_p, _dp, _dp, _i32p]
    L.eg_update_stats.restype = C.c_int32
    L.eg_update_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.eg_fetch_scores.restype = C.c_int32
    L.eg_fetch_scores.argtypes = [C.c_void_p, _dp]
    L.eg_fetch_episode_lists.restype = C.c_int32
    L.eg_fetch_episode_lists.argtypes = [C.c_void_p, C.c_uint32, _dp, _i32p, _u8p, _i32p, _u8p]
    L.eg_policy_apply_reduced.restype = C.c_int32
    L.eg_policy_apply_reduced.argtypes = [C.c_void_p, C.POINTER(C.c_int64), _dp, _i32p, _u8p, _i32p, _u8p, C.c_uint64]
    L.eg_policy_apply_packet.restype = C.c_int32
    L.eg_policy_apply_packet.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint64]
    L.eg_train_step.restype = C.c_int32
    L.eg_train_step.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, _u8p, C.c_uint64]
    L.eg_policy_push.restype = C.c_int32
    L.eg_policy_push.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.eg_device_rollout.restype = C.c_int32
    L.eg_device_rollout.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_void_p]
    L.eg_device_apply.restype = C.c_int32
    L.eg_device_apply.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_uint64]
    L.eg_device_step.restype = C.c_int32
    L.eg_device_step.argtypes = [C.c_void_p, C.c_uint64, C.c_uint64, C.c_uint32, C.c_uint32, C.c_uint64]
    L.eg_policy_pull.restype = C.c_int32
    L.eg_policy_pull.argtypes = [C.c_void_p, C.c_void_p]
    L.eg_policy_append_weight_history.restype = C.c_int32
    L.eg_policy_append_weight_history.argtypes = [C.c_void_p, C.c_char_p, C.c_uint64]
    L.eg_policy_export_improvement_csv.restype = C.c_int32
    L.eg_export_summary_csv.restype = C.c_int32
    L.eg_export_summary_csv.argtypes = [C.POINTER(EgEpisodeOut), C.c_char_p, C.c_char_p]
    L.eg_export_run_details.restype = C.c_int32
    L.eg_export_run_details.argtypes = [C.POINTER(EgWorld), C.POINTER(C.c_char_p), C.POINTER(EgEpisodeOut), C.c_char_p, C.c_uint64]
    L.eg_fetch_record.restype = C.c_int32
    L.eg_fetch_record.argtypes = [C.c_void_p, C.c_uint32, C.POINTER(EgEpisodeOut)]
    L.eg_best_result_track.restype = C.c_int32
    L.eg_best_result_track.argtypes = [C.c_void_p, C.c_int32]
    L.eg_fetch_best_result.restype = C.c_int32
    L.eg_fetch_best_result.argtypes = [C.c_void_p, C.POINTER(EgEpisodeOut), C.POINTER(C.c_int32), C.POINTER(C.c_int64)]
    L.eg_evaluate_action_impact.restype = C.c_double
    L.eg_evaluate_action_impact.argtypes = [_dp, _dp, C.c_int32]
    L.eg_fetch_best_run.restype = C.c_int32
    L.eg_fetch_best_run.argtypes = [C.c_void_p, C.POINTER(EgEpisodeOut), C.POINTER(C.c_int32)]
    L.eg_policy_export_improvement_csv.argtypes = [C.c_void_p, C.c_char_p]
    L.eg_find_suitable_location.restype = C.c_int32
    L.eg_find_suitable_location.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _dp, _dp, C.c_int32, C.c_float, _dp, _dp, _i32p, _dp]
    L.eg_place.restype = C.c_int32
    L.eg_debug_fill_lds.restype = C.c_int32
    L.eg_debug_fill_lds.argtypes = [C.c_void_p, C.c_uint32]
    if hasattr(L, "eg_debug_occupy") or not os.environ.get("EIRGRID_LIB"):
        L.eg_debug_occupy.restype = C.c_int32
        L.eg_debug_occupy.argtypes = [C.c_void_p, C.c_int32, C.c_uint64]
    L.eg_place.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _u16p, C.c_int32, _i32p, _dp]
    L.eg_host_tables_create.restype = C.c_void_p
    L.eg_host_tables_create.argtypes = [C.POINTER(EgWorld)]
    L.eg_host_tables_free.argtypes = [C.c_void_p]
    L.eg_host_tables_f64.restype = C.c_int32
    L.eg_host_tables_f64.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_dp), C.POINTER(C.c_int64)]
    L.eg_host_tables_i32.restype = C.c_int32
    L.eg_host_tables_i32.argtypes = [C.c_void_p, C.c_char_p, C.POINTER(_i32p), C.POINTER(C.c_int64)]
    L.eg_policy_new.restype = C.c_void_p
    L.eg_policy_free.argtypes = [C.c_void_p]
    L.eg_policy_snapshot_view.restype = C.c_int32
    L.eg_policy_snapshot_view.argtypes = [C.c_void_p, C.POINTER(EgPolicySnapshot)]
    L.eg_policy_get_tables.restype = C.c_int32
    L.eg_policy_get_tables.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.eg_policy_set_tables.restype = C.c_int32
    L.eg_policy_set_tables.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.eg_policy_get_scalar.restype = C.c_double
    L.eg_policy_get_scalar.argtypes = [C.c_void_p, C.c_int32]
    L.eg_policy_set_scalar.restype = C.c_int32
    L.eg_policy_set_scalar.argtypes = [C.c_void_p, C.c_int32, C.c_double]
    L.eg_policy_get_list.restype = C.c_int32
    L.eg_policy_get_list.argtypes = [C.c_void_p, C.c_int32, C.c_int32, _u8p, C.c_int32]
    L.eg_policy_apply_episode.restype = C.c_int32
    L.eg_policy_apply_episode.argtypes = [C.c_void_p, _dp, _i32p, _u8p, _i32p, _u8p, C.c_uint64]
    L.eg_policy_save_json.restype = C.c_int32
    L.eg_policy_save_json.argtypes = [C.c_void_p, C.c_char_p]
    L.eg_policy_load_json.restype = C.c_void_p
    L.eg_policy_load_json.argtypes = [C.c_char_p]
    L.eg_score_metrics.restype = C.c_double
    L.eg_score_metrics.argtypes = [_dp, C.c_int32]
    _lib = L
    return L


class EirgridError(RuntimeError):
    pass


def check(rc: int, what: str = "") -> None:
    if rc != EG_OK:
        msg = lib().eg_last_error()
        raise EirgridError(f"{what} failed with code {rc}: {msg.decode() if msg else ''}")
