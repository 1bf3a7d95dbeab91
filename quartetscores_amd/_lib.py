"""ctypes binding of quartetscores_amd/lib/libquartetscores_hip.so (include/quartetscores_hip.h).

Fails loudly when the library is missing: there is no CPU fallback in the product.
"""
import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libquartetscores_hip.so")
# A/B runs of the PYTHON harness against another build of the library (tools/Makefile `exp`): QS_PY_LIB=/path/to/libqs_expX.so.
# Like QS_PY_TUNING this is the harness' switch; the library itself reads no environment variables.
LIB_PATH = os.environ.get("QS_PY_LIB") or LIB_PATH

QS_OK = 0
QS_ERR_ARG, QS_ERR_HIP, QS_ERR_OOM, QS_ERR_STATE, QS_ERR_OVERFLOW, QS_ERR_NO_DEVICE, QS_ERR_UNSUPPORTED, QS_ERR_REFERENCE_THROWS = -1, -2, -3, -4, -5, -6, -7, -8
QS_ALGO_AUTO, QS_ALGO_GATHER, QS_ALGO_SCATTER = 0, 1, 2
QS_COUNT_OVERWRITE = 0x100
QS_COUNT_TIMED = 0x200
QS_COUNT_WIRE16X2 = 0x400
QS_SCORE_QP_WRAP32, QS_SCORE_QP_EXACT64, QS_SCORE_ROOT_AS_EDGE, QS_SCORE_SAVEMEM_LOOKUPS = 0, 1, 2, 4
QS_SCORE_CAND_SLOTS = 8
QS_BATCH_ALL_TAXA, QS_BATCH_BINARY = 1, 2
QS_TUNE_PANEL_SLICE_BYTES, QS_TUNE_GATHER_IMPL, QS_TUNE_PANEL_KERNEL, QS_TUNE_TILE_ORDER = 1, 2, 3, 4
QS_TUNE_SCORE_CAND_SLOTS, QS_TUNE_SCORE_TOL_EXP, QS_TUNE_SCORE_KERNEL, QS_TUNE_TABLE_TREES, QS_TUNE_COOP, QS_TUNE_SCORE_PASSES, QS_TUNE_SCORE_LOG_CAP, QS_TUNE_SCORE_SAMPLE, QS_TUNE_SCORE_DEDUPE, QS_TUNE_SCORE_LOAD, QS_TUNE_CLASS_PCT, QS_TUNE_CLASS_MIN_TREES = 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16
QS_TUNE_DEPTH_CLAMP = 17
QS_TUNE_FUSE_CLASSES = 18
QS_CLASS_PLAN_FUSED = 0x100
QS_IMPL_AUTO, QS_IMPL_SWAR, QS_IMPL_BITSLICE = 0, 1, 2
QS_SHARDS_BY_TUPLES, QS_SHARDS_BY_COST = 0, 1

# every symbol include/quartetscores_hip.h declares
EXPORTS = [
    "qs_create", "qs_destroy", "qs_last_error", "qs_version", "qs_table_tuples", "qs_table_bytes", "qs_table_alloc",
    "qs_table_attach", "qs_table_pack16", "qs_table_pack16x2", "qs_wire_attach", "qs_unpack16x2", "qs_table_device_ptr", "qs_table_clear", "qs_table_download", "qs_table_upload",
    "qs_batch_upload", "qs_batch_free", "qs_count_batch", "qs_count_trees", "qs_sync", "qs_trees_counted", "qs_lookup",
    "qs_score", "qs_score_pair_slots", "qs_score_set_view", "qs_score_pass1", "qs_score_pass2", "qs_score_finish", "qs_raw_qic", "qs_last_count_ms", "qs_last_count_variant",
    "qs_set_tuning", "qs_last_count_launches", "qs_batch_flags", "qs_score_overflow", "qs_free_host", "qs_raw_qic_lex",
    "qs_score_plan", "qs_last_score_ms", "qs_prepare", "qs_table_pack32x2", "qs_unpack32x2", "qs_last_score_log", "qs_last_score_estimate", "qs_score_prepare",
    "qs_sum_words", "qs_issue_probe", "qs_last_count_fix_ms", "qs_batch_clamp_info", "qs_depth_clamp_plan", "qs_score_check", "qs_last_count_events", "qs_class_plan", "qs_shard_bounds",
]


class TreeBatchC(C.Structure):
    _fields_ = [("n_trees", C.c_uint32), ("leaf_off", C.c_void_p), ("leaf_ids", C.c_void_p), ("adj_depth", C.c_void_p),
                ("node_off", C.c_void_p), ("rng_off", C.c_void_p), ("ranges", C.c_void_p)]


class RefTreeC(C.Structure):
    _fields_ = [("n_nodes", C.c_uint32), ("n_taxa", C.c_uint32), ("parent", C.c_void_p), ("leaf_node", C.c_void_p)]


_lib = None


class LibraryMissing(ImportError):
    pass


def load():
    """Load the HIP library; raises LibraryMissing if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise LibraryMissing(
            f"{LIB_PATH} not found: build it with `make -C quartetscores_amd/csrc` (or __graft_entry__.build()). "
            "quartetscores_amd has no CPU fallback.")
    # One HIP runtime per process: torch ships its own libamdhip64. If our library were loaded
    # first it would bind /opt/rocm's copy and a later `import torch` would bring a second
    # runtime that owns the devices (qs_create then sees none). Import torch first when present.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(LIB_PATH)
    vp, u32, u64, i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_int
    L.qs_create.restype = i32
    L.qs_create.argtypes = [C.POINTER(vp), u32, u32, u32, i32, vp, u32, u32]
    L.qs_destroy.restype = None
    L.qs_destroy.argtypes = [vp]
    L.qs_last_error.restype = C.c_char_p
    L.qs_last_error.argtypes = [vp]
    L.qs_version.restype = C.c_char_p
    L.qs_version.argtypes = []
    L.qs_table_tuples.restype = u64
    L.qs_table_tuples.argtypes = [vp]
    L.qs_table_bytes.restype = u64
    L.qs_table_bytes.argtypes = [vp]
    L.qs_table_alloc.restype = i32
    L.qs_table_alloc.argtypes = [vp]
    L.qs_table_attach.restype = i32
    L.qs_table_attach.argtypes = [vp, vp, u64]
    L.qs_table_pack16.restype = i32
    L.qs_table_pack16.argtypes = [vp, vp, u64]
    L.qs_table_pack16x2.restype = i32
    L.qs_table_pack16x2.argtypes = [vp, vp, u64]
    L.qs_wire_attach.restype = i32
    L.qs_wire_attach.argtypes = [vp, vp, u64]
    L.qs_unpack16x2.restype = i32
    L.qs_unpack16x2.argtypes = [vp, vp, u64, u32, vp]
    L.qs_table_pack32x2.restype = i32
    L.qs_table_pack32x2.argtypes = [vp, vp, u64]
    L.qs_unpack32x2.restype = i32
    L.qs_unpack32x2.argtypes = [vp, vp, u64, u64, vp]
    L.qs_issue_probe.restype = i32
    L.qs_issue_probe.argtypes = [vp, u32, vp]
    L.qs_sum_words.restype = i32
    L.qs_sum_words.argtypes = [vp, vp, vp, u32, u64]
    L.qs_table_device_ptr.restype = vp
    L.qs_table_device_ptr.argtypes = [vp]
    L.qs_table_clear.restype = i32
    L.qs_table_clear.argtypes = [vp]
    L.qs_table_download.restype = i32
    L.qs_table_download.argtypes = [vp, vp, u64]
    L.qs_table_upload.restype = i32
    L.qs_table_upload.argtypes = [vp, vp, u64]
    L.qs_batch_upload.restype = i32
    L.qs_batch_upload.argtypes = [vp, C.POINTER(TreeBatchC), C.POINTER(vp)]
    L.qs_batch_free.restype = None
    L.qs_batch_free.argtypes = [vp, vp]
    L.qs_batch_flags.restype = u32
    L.qs_batch_flags.argtypes = [vp]
    L.qs_count_batch.restype = i32
    L.qs_count_batch.argtypes = [vp, vp, u32]
    L.qs_count_trees.restype = i32
    L.qs_count_trees.argtypes = [vp, C.POINTER(TreeBatchC), u32]
    L.qs_sync.restype = i32
    L.qs_sync.argtypes = [vp]
    L.qs_trees_counted.restype = u64
    L.qs_trees_counted.argtypes = [vp]
    L.qs_lookup.restype = i32
    L.qs_lookup.argtypes = [vp, u64, vp, vp]
    L.qs_score.restype = i32
    L.qs_score.argtypes = [vp, C.POINTER(RefTreeC), u32, vp, vp, vp, C.POINTER(i32)]
    L.qs_score_pair_slots.restype = u64
    L.qs_score_pair_slots.argtypes = [C.POINTER(RefTreeC)]
    L.qs_score_set_view.restype = i32
    L.qs_score_set_view.argtypes = [vp, vp, u32, u64, u64]
    L.qs_score_prepare.restype = i32
    L.qs_score_prepare.argtypes = [vp, C.POINTER(RefTreeC), u64]
    L.qs_score_pass1.restype = i32
    L.qs_score_pass1.argtypes = [vp, C.POINTER(RefTreeC), vp, vp]
    L.qs_score_pass2.restype = i32
    L.qs_score_pass2.argtypes = [vp, C.POINTER(RefTreeC), vp, vp]
    L.qs_score_finish.restype = i32
    L.qs_score_finish.argtypes = [vp, C.POINTER(RefTreeC), u32, vp, vp, u32, vp, u64, vp, vp, vp, C.POINTER(i32)]
    L.qs_score_overflow.restype = i32
    L.qs_score_overflow.argtypes = [vp, C.POINTER(RefTreeC), vp, vp, C.POINTER(vp), C.POINTER(u64)]
    L.qs_free_host.restype = None
    L.qs_free_host.argtypes = [vp]
    L.qs_raw_qic.restype = i32
    L.qs_raw_qic.argtypes = [vp, C.POINTER(RefTreeC), u64, u64, vp, vp]
    L.qs_raw_qic_lex.restype = i32
    L.qs_raw_qic_lex.argtypes = [vp, C.POINTER(RefTreeC), u64, u64, vp, vp]
    L.qs_last_count_ms.restype = i32
    L.qs_last_count_ms.argtypes = [vp, C.POINTER(C.c_float * 3)]
    L.qs_last_score_log.restype = u64
    L.qs_last_score_log.argtypes = [vp]
    L.qs_last_score_estimate.restype = u64
    L.qs_last_score_estimate.argtypes = [vp]
    L.qs_prepare.restype = i32
    L.qs_prepare.argtypes = [vp, u64]
    L.qs_last_score_ms.restype = i32
    L.qs_last_score_ms.argtypes = [vp, C.POINTER(C.c_float * 6)]
    L.qs_set_tuning.restype = i32
    L.qs_set_tuning.argtypes = [vp, u32, u64]
    L.qs_last_count_launches.restype = i32
    L.qs_last_count_launches.argtypes = [vp]
    L.qs_score_plan.restype = i32
    L.qs_score_plan.argtypes = [u32, u64, u64, vp, vp, vp]
    L.qs_last_count_variant.restype = C.c_char_p
    L.qs_last_count_variant.argtypes = [vp]
    L.qs_last_count_fix_ms.restype = C.c_float
    L.qs_last_count_fix_ms.argtypes = [vp]
    L.qs_batch_clamp_info.restype = i32
    L.qs_batch_clamp_info.argtypes = [vp, C.POINTER(u64 * 3)]
    L.qs_last_count_events.restype = i32
    L.qs_last_count_events.argtypes = [vp, vp, vp, i32]
    L.qs_shard_bounds.restype = i32
    L.qs_shard_bounds.argtypes = [u32, u32, u32, vp]
    L.qs_class_plan.restype = i32
    L.qs_class_plan.argtypes = [u32, C.POINTER(TreeBatchC), u32, u32, u32, vp, vp, vp]
    L.qs_score_check.restype = i32
    L.qs_score_check.argtypes = [vp, C.POINTER(RefTreeC), u32]
    L.qs_depth_clamp_plan.restype = i32
    L.qs_depth_clamp_plan.argtypes = [u32, C.POINTER(TreeBatchC), u32, vp, vp, vp]
    _lib = L
    return L
