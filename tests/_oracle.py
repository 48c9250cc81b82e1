"""ctypes binding for the CPU parity oracle (oracle/libykoracle.so). Test infrastructure only."""
import ctypes
import json
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB_PATH = os.path.join(ORACLE_DIR, "libykoracle.so")

PLUGIN_BITS = {
    "NodeUnschedulable": 1 << 0,
    "NodeName": 1 << 1,
    "TaintToleration": 1 << 2,
    "NodeAffinity": 1 << 3,
    "NodePorts": 1 << 4,
    "NodeResourcesFit": 1 << 5,
    "PodTopologySpread": 1 << 6,
    "InterPodAffinity": 1 << 7,
}
PLUGIN_NAMES = ["", "NodeUnschedulable", "NodeName", "TaintToleration", "NodeAffinity", "NodePorts", "NodeResourcesFit",
                "PodTopologySpread", "InterPodAffinity"]
ALL = sum(PLUGIN_BITS.values())
# predicate_manager.go:321-368 — reservation phase lists restricted to the plugins of this path
RESERVE_PRE = (PLUGIN_BITS["NodeAffinity"] | PLUGIN_BITS["NodePorts"] | PLUGIN_BITS["PodTopologySpread"]
               | PLUGIN_BITS["InterPodAffinity"])
RESERVE_FILT = (PLUGIN_BITS["NodeUnschedulable"] | PLUGIN_BITS["NodeName"] | PLUGIN_BITS["TaintToleration"]
                | PLUGIN_BITS["NodeAffinity"] | PLUGIN_BITS["NodePorts"] | PLUGIN_BITS["PodTopologySpread"]
                | PLUGIN_BITS["InterPodAffinity"])


def mask_of(names):
    m = 0
    for n in names:
        m |= PLUGIN_BITS.get(n, 0)
    return m


def build():
    src_newer = (not os.path.exists(LIB_PATH)) or any(
        os.path.getmtime(os.path.join(ORACLE_DIR, f)) > os.path.getmtime(LIB_PATH)
        for f in ("ykoracle.cpp", "orc_json.h", "orc_quantity.h"))
    if src_newer:
        subprocess.check_call(["make", "-C", ORACLE_DIR, "-s"])
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = ctypes.CDLL(build())
        L.orc_load.restype = ctypes.c_void_p
        L.orc_load.argtypes = [ctypes.c_char_p, ctypes.c_char_p, ctypes.c_int]
        L.orc_free.argtypes = [ctypes.c_void_p]
        L.orc_num_nodes.argtypes = [ctypes.c_void_p]
        L.orc_num_pods.argtypes = [ctypes.c_void_p]
        L.orc_predicates.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                     ctypes.POINTER(ctypes.c_int), ctypes.c_char_p, ctypes.c_int]
        L.orc_eval_grid.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int]
        L.orc_eval_rows.argtypes = L.orc_eval_grid.argtypes
        L.orc_preemption.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                     ctypes.c_int, ctypes.c_uint, ctypes.c_uint]
        L.orc_pod_request_json.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_char_p, ctypes.c_int]
        L.orc_node_info.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p]
        L.orc_binpack_scores.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        L.orc_decide.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint,
                                 ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]
        L.orc_decide_once.argtypes = L.orc_decide.argtypes
        L.orc_allocate_sequential.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_uint, ctypes.c_uint, ctypes.c_void_p,
                                              ctypes.c_int]
        L.orc_quantity_value.restype = ctypes.c_int64
        L.orc_quantity_value.argtypes = [ctypes.c_char_p]
        L.orc_quantity_milli.restype = ctypes.c_int64
        L.orc_quantity_milli.argtypes = [ctypes.c_char_p]
        _lib = L
    return _lib


class Oracle:
    """One loaded cluster snapshot: {"nodes": [...], "pods": [...]} in Kubernetes JSON field names."""

    def __init__(self, snapshot):
        text = snapshot if isinstance(snapshot, (str, bytes)) else json.dumps(snapshot)
        if isinstance(text, str):
            text = text.encode()
        err = ctypes.create_string_buffer(512)
        self._h = lib().orc_load(text, err, 512)
        if not self._h:
            raise ValueError("oracle: " + err.value.decode())
        self.num_nodes = lib().orc_num_nodes(self._h)
        self.num_pods = lib().orc_num_pods(self._h)

    def close(self):
        if self._h:
            lib().orc_free(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def predicates(self, pod, node, pre_mask=ALL, filt_mask=ALL):
        """Returns (fits, plugin_name, message) for one Predicates() call."""
        plugin = ctypes.c_int(0)
        msg = ctypes.create_string_buffer(512)
        r = lib().orc_predicates(self._h, pod, node, pre_mask, filt_mask, ctypes.byref(plugin), msg, 512)
        if r < 0:
            raise IndexError("pod/node index out of range")
        return bool(r), PLUGIN_NAMES[plugin.value], msg.value.decode()

    def eval_grid(self, pods=None, nodes=None, pre_mask=ALL, filt_mask=ALL, threads=1, want_plugin=False, prefilter_once=False):
        """[len(pods)][len(nodes)] verdicts (+ failing-plugin codes). prefilter_once: the PreFilter pass of a pod is run once for
        all its nodes instead of once per pair — same verdicts (held equal by tests/test_oracle_golden.py), affordable at 10^5
        nodes with hard spread constraints."""
        pods = np.arange(self.num_pods, dtype=np.int32) if pods is None else np.ascontiguousarray(pods, dtype=np.int32)
        nodes = np.arange(self.num_nodes, dtype=np.int32) if nodes is None else np.ascontiguousarray(nodes, dtype=np.int32)
        fit = np.zeros((len(pods), len(nodes)), dtype=np.uint8)
        plug = np.zeros((len(pods), len(nodes)), dtype=np.uint8) if want_plugin else None
        fn = lib().orc_eval_rows if prefilter_once else lib().orc_eval_grid
        fn(self._h, pods.ctypes.data, len(pods), nodes.ctypes.data, len(nodes), pre_mask, filt_mask,
           fit.ctypes.data, plug.ctypes.data if want_plugin else None, threads)
        return (fit, plug) if want_plugin else fit

    def preemption(self, pod, node, victims, start, pre_mask=ALL, filt_mask=ALL):
        v = np.ascontiguousarray(victims, dtype=np.int32)
        return lib().orc_preemption(self._h, pod, node, v.ctypes.data, len(v), start, pre_mask, filt_mask)

    def pod_request(self, pod):
        buf = ctypes.create_string_buffer(4096)
        lib().orc_pod_request_json(self._h, pod, buf, 4096)
        return json.loads(buf.value.decode())

    def node_info(self, node):
        out = np.zeros(9, dtype=np.int64)
        lib().orc_node_info(self._h, node, out.ctypes.data)
        return {"alloc": out[0:3].tolist(), "allowed_pods": int(out[3]), "pod_count": int(out[4]),
                "requested": out[6:9].tolist()}

    def binpack_scores(self):
        out = np.zeros(self.num_nodes, dtype=np.float64)
        lib().orc_binpack_scores(self._h, out.ctypes.data)
        return out

    def decide(self, pod, pre_mask=ALL, filt_mask=ALL, prefilter_once=False):
        c, b = ctypes.c_int(0), ctypes.c_int(0)
        (lib().orc_decide_once if prefilter_once else lib().orc_decide)(self._h, pod, pre_mask, filt_mask, ctypes.byref(c), ctypes.byref(b))
        return c.value, b.value


    def allocate_sequential(self, pods=None, pre_mask=ALL, filt_mask=ALL, early_exit=True, prefilter_once=False):
        """The loop yunikorn-core drives: decide pods[i] on the current state (first fit in bin-pack order), AssumePod it, go on.
        → node index per ask (-1: none fits). MUTATES the loaded snapshot (the assumed asks now sit on their nodes).
        prefilter_once: the ask's PreFilter pass once instead of once per candidate node (same answers — held equal by
        tests/test_oracle_sequential.py; the only form large clusters with topology constraints can afford)."""
        pods = np.arange(self.num_pods, dtype=np.int32) if pods is None else np.ascontiguousarray(pods, dtype=np.int32)
        out = np.full(len(pods), -1, dtype=np.int32)
        mode = 2 if (prefilter_once and early_exit) else (1 if early_exit else 0)
        lib().orc_allocate_sequential(self._h, pods.ctypes.data, len(pods), pre_mask, filt_mask, out.ctypes.data, mode)
        return out


def pack_bits(fit):
    """[P][N] 0/1 → [P][ceil(N/64)] uint64, bit j%64 of word j//64 = node j (the engine's bitmap layout)."""
    P, N = fit.shape
    W = (N + 63) // 64
    padded = np.zeros((P, W * 64), dtype=np.uint8)
    padded[:, :N] = fit
    return np.packbits(padded.reshape(P, W, 64), axis=2, bitorder="little").view(np.uint64).reshape(P, W)
