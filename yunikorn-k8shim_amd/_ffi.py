"""ctypes declarations for libykpred.so (include/ykpred.h) and libykhost.so (include/ykhost.h)."""
import ctypes as C
import os

from . import build as _build

c_i32p = C.POINTER(C.c_int32)


class YkpredConfig(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("device", C.c_int32), ("num_resources", C.c_int32), ("taint_words", C.c_int32),
                ("label_words", C.c_int32), ("topology_keys", C.c_int32), ("selector_classes", C.c_int32),
                ("port_words", C.c_int32), ("reserved", C.c_int32 * 8)]


class YkpredNodes(C.Structure):
    _fields_ = [("count", C.c_int32), ("allocatable", C.c_void_p), ("requested", C.c_void_p), ("allowed_pods", C.c_void_p),
                ("pod_count", C.c_void_p), ("flags", C.c_void_p), ("taint_bits", C.c_void_p), ("label_bits", C.c_void_p),
                ("domain_id", C.c_void_p), ("selector_count", C.c_void_p), ("domain_sizes", C.c_void_p),
                ("port_bits", C.c_void_p), ("name_rank", C.c_void_p)]


class YkpredSpecs(C.Structure):
    _fields_ = [("count", C.c_int32), ("requests", C.c_void_p), ("tolerated", C.c_void_p), ("flags", C.c_void_p),
                ("aff_term_off", C.c_void_p), ("aff_terms", C.c_void_p), ("pre_term_off", C.c_void_p), ("pre_terms", C.c_void_p),
                ("spread_off", C.c_void_p), ("spread", C.c_void_p), ("wanted_ports", C.c_void_p)]


class YkpredPods(C.Structure):
    _fields_ = [("count", C.c_int32), ("spec_index", C.c_void_p), ("node_name_index", C.c_void_p)]


class YkpredEvalArgs(C.Structure):
    _fields_ = [("prefilter_plugins", C.c_uint32), ("filter_plugins", C.c_uint32), ("options", C.c_uint32),
                ("bitmap_rows", C.c_uint32), ("bitmap", C.c_void_p), ("stream", C.c_void_p), ("counts", C.c_void_p),
                ("decisions", C.c_void_p), ("decision_keys", C.c_void_p)]


class YkpredLayout(C.Structure):
    _fields_ = [("num_nodes", C.c_int32), ("num_pods", C.c_int32), ("num_specs", C.c_int32), ("num_classes", C.c_int32),
                ("row_words", C.c_int32), ("row_stride", C.c_int32), ("num_chunks", C.c_int32), ("plane_rows", C.c_int32),
                ("bitmap_bytes", C.c_uint64), ("bitmap", C.c_void_p), ("counts", C.c_void_p), ("decisions", C.c_void_p),
                ("decision_keys", C.c_void_p), ("spread_counts", C.c_void_p), ("spread_present", C.c_void_p),
                ("spread_cells", C.c_int64), ("num_rows", C.c_int32), ("band_rows", C.c_int32), ("row_of_pod", C.c_void_p),
                ("index_rows", C.c_int32), ("band_steps", C.c_int32),
                ("sweep_rows", C.c_int32), ("index_rows_walked", C.c_int32), ("run_rows", C.c_int32), ("fused_rows", C.c_int32)]


MAX_TIMED = 24


class YkpredTiming(C.Structure):
    _fields_ = [("num_kernels", C.c_int32), ("total_ms", C.c_float), ("kernel_ms", C.c_float * MAX_TIMED),
                ("kernel_name", C.c_char_p * MAX_TIMED)]


class YkhostKwok(C.Structure):
    _fields_ = [("seed", C.c_uint64), ("num_nodes", C.c_int32), ("num_pods", C.c_int32), ("num_templates", C.c_int32),
                ("node_affinity", C.c_int32), ("tolerations", C.c_int32), ("unique_requests", C.c_int32),
                ("gang_size", C.c_int32), ("node_index_offset", C.c_int32), ("spread", C.c_int32), ("total_nodes", C.c_int32),
                ("reserved", C.c_int32 * 2)]


_pred = None
_host = None


def load_ykpred():
    """Loads libykpred.so (building it first if the sources are newer). Raises if it cannot be built/loaded."""
    global _pred
    if _pred is not None:
        return _pred
    path = _build.build_engine()
    L = C.CDLL(path, mode=C.RTLD_GLOBAL)
    L.ykpred_abi_version.restype = C.c_int32
    L.ykpred_last_error.restype = C.c_char_p
    L.ykpred_last_error.argtypes = [C.c_void_p]
    L.ykpred_create.argtypes = [C.POINTER(YkpredConfig), C.POINTER(C.c_void_p)]
    L.ykpred_destroy.argtypes = [C.c_void_p]
    L.ykpred_destroy.restype = None
    L.ykpred_set_nodes.argtypes = [C.c_void_p, C.POINTER(YkpredNodes)]
    L.ykpred_update_node.argtypes = [C.c_void_p, C.c_int32, C.POINTER(YkpredNodes)]
    L.ykpred_update_label_word.argtypes = [C.c_void_p, C.c_int32, C.c_void_p]
    L.ykpred_set_specs.argtypes = [C.c_void_p, C.POINTER(YkpredSpecs)]
    L.ykpred_set_pods.argtypes = [C.c_void_p, C.POINTER(YkpredPods)]
    L.ykpred_eval.argtypes = [C.c_void_p, C.POINTER(YkpredEvalArgs)]
    L.ykpred_eval_nodes.argtypes = [C.c_void_p, C.POINTER(YkpredEvalArgs), C.c_int32, C.c_void_p]
    L.ykpred_update_pods.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_eval_pods.argtypes = [C.c_void_p, C.POINTER(YkpredEvalArgs), C.c_int32, C.c_void_p]
    L.ykpred_synchronize.argtypes = [C.c_void_p]
    L.ykpred_get_layout.argtypes = [C.c_void_p, C.POINTER(YkpredLayout)]
    L.ykpred_last_timing.argtypes = [C.c_void_p, C.POINTER(YkpredTiming)]
    L.ykpred_get_counters.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_get_round_info.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_read_bitmap.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]
    L.ykpred_read_counts.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_read_decisions.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_read_scores.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_bitmap_checksum.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ykpred_read_rows.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]
    L.ykpred_read_pod_classes.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_read_row_map.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_check_class_rows.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ykpred_guard_selftest.argtypes = [C.c_void_p, C.c_int32]
    L.ykpred_allocate_round.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p, C.c_void_p]
    L.ykpred_set_spec_effects.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_comm_use_library.argtypes = [C.c_char_p]
    L.ykpred_comm_unique_id.argtypes = [C.c_void_p]
    L.ykpred_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.ykpred_comm_destroy.argtypes = [C.c_void_p]
    L.ykpred_comm_info.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ykpred_set_row_stride.argtypes = [C.c_void_p, C.c_int32]
    L.ykpred_set_row_capacity.argtypes = [C.c_void_p, C.c_int32]
    L.ykpred_gather_bitmap.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_gather_bitmap_compressed.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_layout_hash.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
    L.ykpred_collect_class_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_expand_class_rows.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_exchange_decisions.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_read_gathered.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.ykpred_query.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p,
                               C.c_void_p]
    L.ykpred_preemption_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                          C.c_void_p, C.c_uint32, C.c_uint32, C.c_void_p]
    L.ykpred_query_pod.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ykpred_preemption.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_uint32,
                                    C.c_uint32, C.POINTER(C.c_int32)]
    L.ykpred_preemption_ports.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                          C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]
    L.ykpred_peek_row.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
    L.ykpred_read_order.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_answer_state.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.POINTER(C.c_int32)]
    L.ykpred_pod_class.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int32)]
    L.ykpred_read_class_rows.argtypes = [C.c_void_p, C.c_void_p]
    L.ykpred_query_pod_packed.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.c_uint32, C.c_void_p]
    _pred = L
    return L


def load_ykhost():
    global _host
    if _host is not None:
        return _host
    load_ykpred()
    path = _build.build_host()
    L = C.CDLL(path)
    L.ykhost_create.restype = C.c_void_p
    L.ykhost_create.argtypes = [C.c_int32, C.c_char_p, C.c_int32]
    L.ykhost_destroy.argtypes = [C.c_void_p]
    L.ykhost_destroy.restype = None
    L.ykhost_last_error.restype = C.c_char_p
    L.ykhost_last_error.argtypes = [C.c_void_p]
    L.ykhost_set_plugins.argtypes = [C.c_void_p, C.c_uint32, C.c_uint32, C.c_uint32, C.c_uint32]
    for fn in ("ykhost_load_snapshot", "ykhost_update_node", "ykhost_remove_node", "ykhost_update_pod", "ykhost_remove_pod",
               "ykhost_forget_pod"):
        getattr(L, fn).argtypes = [C.c_void_p, C.c_char_p]
    L.ykhost_assume_pod.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.ykhost_validate_task_groups.argtypes = [C.c_void_p, C.c_char_p]
    L.ykhost_add_task_groups.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p]
    L.ykhost_pod_state.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32]
    L.ykhost_node_pod_count.argtypes = [C.c_void_p, C.c_char_p]
    L.ykhost_generate_kwok.argtypes = [C.c_void_p, C.POINTER(YkhostKwok)]
    L.ykhost_num_nodes.argtypes = [C.c_void_p]
    L.ykhost_num_pods.argtypes = [C.c_void_p]
    L.ykhost_pod_index.argtypes = [C.c_void_p, C.c_char_p]
    L.ykhost_node_index.argtypes = [C.c_void_p, C.c_char_p]
    L.ykhost_dump_snapshot.restype = C.c_int64
    L.ykhost_dump_snapshot.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]
    L.ykhost_set_dump_compact.argtypes = [C.c_void_p, C.c_int32]
    L.ykhost_set_row_stride.argtypes = [C.c_void_p, C.c_int32]
    L.ykhost_set_row_capacity.argtypes = [C.c_void_p, C.c_int32]
    L.ykhost_comm_init.argtypes = [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32]
    L.ykhost_comm_destroy.argtypes = [C.c_void_p]
    L.ykhost_sync.argtypes = [C.c_void_p]
    L.ykhost_encoded_tables_json.restype = C.c_int64
    L.ykhost_encoded_tables_json.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ykhost_engine.restype = C.c_void_p
    L.ykhost_engine.argtypes = [C.c_void_p]
    L.ykhost_evaluate.argtypes = [C.c_void_p, C.c_int32, C.c_uint32]
    L.ykhost_evaluate_dirty.argtypes = [C.c_void_p, C.c_int32, C.c_uint32, C.POINTER(C.c_int32)]
    L.ykhost_predicates.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
    L.ykhost_preemption_predicates.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.POINTER(C.c_char_p), C.c_int32, C.c_int32]
    L.ykhost_preemption_predicates_batch.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(C.c_char_p),
                                                     C.c_void_p, C.c_void_p]
    L.ykhost_is_pod_fit_node.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.c_int32, C.c_char_p, C.c_int32]
    L.ykhost_is_pod_fit_node_via_preemption.argtypes = [C.c_void_p, C.c_char_p, C.c_char_p, C.POINTER(C.c_char_p), C.c_int32, C.c_int32]
    L.ykhost_pod_request_json.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]
    L.ykhost_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_ask_supported.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int32]
    L.ykhost_routing_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_update_nodes_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ykhost_update_pods_batch.argtypes = [C.c_void_p, C.c_char_p, C.c_int64]
    L.ykhost_dump_documents.restype = C.c_int64
    L.ykhost_dump_documents.argtypes = [C.c_void_p, C.c_int32, C.c_char_p, C.c_int64]
    L.ykhost_ingest_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_ingest_timing.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_candidates.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]
    L.ykhost_resident_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_allocate_round.argtypes = [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p]
    L.ykhost_round_stats.argtypes = [C.c_void_p, C.c_void_p]
    L.ykhost_device_errors.restype = C.c_int64
    L.ykhost_device_errors.argtypes = [C.c_void_p]
    _host = L
    return L
