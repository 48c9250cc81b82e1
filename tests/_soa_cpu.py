"""ctypes binding of oracle/libsoacpu.so: the table-driven, multi-threaded CPU evaluator used as the second, stronger
baseline of bench.py. Test / measurement infrastructure only."""
import ctypes
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libsoacpu.so")


class Tables(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in ("N", "S", "P", "R", "KT", "W", "KP")] + [
        (n, ctypes.c_void_p) for n in ("allocatable", "requested", "allowed_pods", "pod_count", "node_flags", "taint_bits",
                                       "label_bits", "port_bits", "requests", "tolerated", "spec_flags", "aff_term_off", "aff_terms",
                                       "pre_term_off", "pre_terms", "wanted_ports", "pod_spec", "pod_node_name")]


DTYPES = {"allocatable": np.int64, "requested": np.int64, "allowed_pods": np.int32, "pod_count": np.int32, "node_flags": np.uint32,
          "taint_bits": np.uint64, "label_bits": np.uint64, "port_bits": np.uint64, "requests": np.int64, "tolerated": np.uint64,
          "spec_flags": np.uint32, "aff_term_off": np.int32, "aff_terms": np.uint64, "pre_term_off": np.int32, "pre_terms": np.uint64,
          "wanted_ports": np.uint64, "pod_spec": np.int32, "pod_node_name": np.int32}
_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH) or os.path.getmtime(os.path.join(ORACLE_DIR, "soa_cpu.c")) > os.path.getmtime(LIB_PATH):
            subprocess.check_call(["make", "-C", ORACLE_DIR, "-s", "libsoacpu.so"])
        _lib = ctypes.CDLL(LIB_PATH)
        _lib.soa_eval.restype = ctypes.c_int64
        _lib.soa_eval.argtypes = [ctypes.POINTER(Tables), ctypes.c_uint32, ctypes.c_uint32, ctypes.c_int32, ctypes.c_void_p]
    return _lib


def prepare(tables):
    """tables: dict from GpuPredicateManager.encoded_tables() (no topology constraints) → opaque prepared input."""
    if tables["KD"] or tables["spread_constraints"]:
        raise ValueError("the table-driven CPU evaluator does not cover topology constraints")
    keep = {k: np.ascontiguousarray(np.array(tables[k], dtype=dt)) if len(tables[k]) else np.zeros(1, dtype=dt) for k, dt in DTYPES.items()}
    t = Tables(**{k: int(tables[k]) for k in ("N", "S", "P", "R", "KT", "W", "KP")}, **{k: v.ctypes.data for k, v in keep.items()})
    out = np.zeros((max(tables["P"], 1), max((tables["N"] + 63) // 64, 1)), dtype=np.uint64)
    return t, keep, out, tables["P"]


def run(prepared, pre_mask, filt_mask, threads=1):
    """One pass over every (pod, node) pair of the prepared tables → bitmap uint64 [P][ceil(N/64)]."""
    t, _keep, out, P = prepared
    lib().soa_eval(ctypes.byref(t), pre_mask, filt_mask, threads, out.ctypes.data)
    return out[:P]


def evaluate(tables, pre_mask, filt_mask, threads=1):
    return run(prepare(tables), pre_mask, filt_mask, threads)
