"""Python face of the host library: the reference's PredicateManager interface over the MI355X engine.

Method names and result conventions follow /root/reference/pkg/plugin/predicates/predicate_manager.go:47-55 so the
parity tests read like predicate_manager_test.go:

    pm = GpuPredicateManager()                       # NewPredicateManager(handle)
    pm = GpuPredicateManager.internal(ep, ep, ep, ep)  # newPredicateManagerInternal(handle, resPre, allocPre, resFilt, allocFilt)
    plugin, err = pm.predicates(pod, node, allocate)   # ("", None) when the pod fits
    index = pm.preemption_predicates(pod, node, victims, start_index)

There is no CPU evaluation path: constructing a manager without a visible GPU raises.
"""
import ctypes as C
import json

import numpy as np

from . import _ffi

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
ALL_PLUGINS = 0xFF
OUT_BITMAP, OUT_COUNTS, OUT_DECISIONS, OUT_DECISION_KEYS = 1, 2, 4, 8
EVAL_PROFILE, EVAL_DIRECT = 1 << 8, 1 << 9
EVAL_SPREAD_COUNT_ONLY, EVAL_SPREAD_COUNTS_READY = 1 << 10, 1 << 11


def plugin_mask(names):
    """enabledPlugins(...) of predicate_manager_test.go:2201-2207; names outside the engine's set are ignored like absent plugins."""
    if isinstance(names, int):
        return names
    m = 0
    for n in names:
        if n == "*":
            return ALL_PLUGINS
        m |= PLUGIN_BITS.get(n, 0)
    return m


class UnsupportedAsk(Exception):
    """Predicates() on an ask the engine does not evaluate (YKHOST_E_UNSUPPORTED): the CPU PredicateManager must answer it."""


class PredicateError(Exception):
    """The error value returned by Predicates(): carries the failing plugin like Context.IsPodFitNode's joined error."""

    def __init__(self, plugin, message):
        super().__init__(f"failed plugin: '{plugin}'\n{message}")
        self.plugin = plugin
        self.message = message


class GpuPredicateManager:
    def __init__(self, device=0):
        self._L = _ffi.load_ykhost()
        self._P = _ffi.load_ykpred()
        err = C.create_string_buffer(512)
        self._h = self._L.ykhost_create(device, err, 512)
        if not self._h:
            raise RuntimeError("ykhost_create failed (the engine has no CPU fallback): " + err.value.decode())
        # NewPredicateManager's phase lists (predicate_manager.go:321-373), restricted to the engine's plugins
        reserve_pre = (PLUGIN_BITS["NodeAffinity"] | PLUGIN_BITS["NodePorts"] | PLUGIN_BITS["PodTopologySpread"]
                       | PLUGIN_BITS["InterPodAffinity"])
        reserve_filt = reserve_pre | PLUGIN_BITS["NodeUnschedulable"] | PLUGIN_BITS["NodeName"] | PLUGIN_BITS["TaintToleration"]
        self._masks = (reserve_pre, ALL_PLUGINS, reserve_filt, ALL_PLUGINS)

    @classmethod
    def internal(cls, reservation_prefilters, allocation_prefilters, reservation_filters, allocation_filters, device=0):
        pm = cls(device)
        pm._masks = (plugin_mask(reservation_prefilters), plugin_mask(allocation_prefilters),
                     plugin_mask(reservation_filters), plugin_mask(allocation_filters))
        pm._L.ykhost_set_plugins(pm._h, *pm._masks)
        return pm

    def close(self):
        if getattr(self, "_h", None):
            self._L.ykhost_destroy(self._h)
            self._h = None

    def __del__(self):
        self.close()

    def _check(self, rc):
        if rc < 0:
            raise RuntimeError(self._L.ykhost_last_error(self._h).decode())
        return rc

    # ---- cluster state (SchedulerCache mirror) ------------------------------------------------------------
    def load_snapshot(self, snapshot):
        text = snapshot if isinstance(snapshot, (str, bytes)) else json.dumps(snapshot)
        self._check(self._L.ykhost_load_snapshot(self._h, text.encode() if isinstance(text, str) else text))

    def update_node(self, node):
        """SchedulerCache.UpdateNode → number of orphaned pods the node adopted."""
        return self._check(self._L.ykhost_update_node(self._h, json.dumps(node).encode()))

    def remove_node(self, name):
        """SchedulerCache.RemoveNode → number of pods orphaned (assumed pods are reverted to pending instead)."""
        return self._check(self._L.ykhost_remove_node(self._h, name.encode()))

    def update_pod(self, pod):
        """SchedulerCache.UpdatePod → False when the pod was stored as an orphan (its node is unknown)."""
        return bool(self._check(self._L.ykhost_update_pod(self._h, json.dumps(pod).encode())))

    def update_nodes_batch(self, nodes):
        """Many SchedulerCache.UpdateNode calls in one crossing: `nodes` = objects or JSON documents (bytes)."""
        text = b"\n".join(n if isinstance(n, bytes) else json.dumps(n).encode() for n in nodes)
        return self._check(self._L.ykhost_update_nodes_batch(self._h, text, len(text)))

    def update_pods_batch(self, pods):
        """Many SchedulerCache.UpdatePod calls in one crossing."""
        text = b"\n".join(p if isinstance(p, bytes) else json.dumps(p).encode() for p in pods)
        return self._check(self._L.ykhost_update_pods_batch(self._h, text, len(text)))

    def remove_pod(self, uid):
        return bool(self._check(self._L.ykhost_remove_pod(self._h, uid.encode())))

    def assume_pod(self, uid, node_name):
        self._check(self._L.ykhost_assume_pod(self._h, uid.encode(), node_name.encode()))

    def forget_pod(self, uid):
        """SchedulerCache.ForgetPod: drops the assumed mark; the pod stays accounted on its node."""
        return bool(self._check(self._L.ykhost_forget_pod(self._h, uid.encode())))

    def validate_task_groups(self, annotation):
        """GetTaskGroupsFromAnnotation: number of task groups; raises RuntimeError with the reason when rejected."""
        text = annotation if isinstance(annotation, str) else json.dumps(annotation)
        return self._check(self._L.ykhost_validate_task_groups(self._h, text.encode()))

    def add_task_groups(self, application_id, queue, namespace, annotation):
        """Creates the minMember placeholder asks of every task group (newPlaceholder); returns how many."""
        text = annotation if isinstance(annotation, str) else json.dumps(annotation)
        app = json.dumps({"applicationId": application_id, "queue": queue, "namespace": namespace})
        return self._check(self._L.ykhost_add_task_groups(self._h, app.encode(), text.encode()))

    def pod_state(self, uid):
        """Cache membership of a pod: None if not cached, else dict(node, assigned, assumed, orphan, ask)."""
        node = C.create_string_buffer(256)
        f = self._L.ykhost_pod_state(self._h, uid.encode(), node, 256)
        if not f & 1:
            return None
        return {"node": node.value.decode(), "assigned": bool(f & 2), "assumed": bool(f & 4), "orphan": bool(f & 8), "ask": bool(f & 16)}

    def node_pod_count(self, name):
        """len(NodeInfo.Pods) of a cached node; None if the node is not cached."""
        c = self._L.ykhost_node_pod_count(self._h, name.encode())
        return None if c < 0 else c

    def generate_kwok(self, seed, num_nodes, num_pods, num_templates=0, node_affinity=1, tolerations=1, unique_requests=0,
                      gang_size=0, node_index_offset=0, spread=0, total_nodes=0):
        cfg = _ffi.YkhostKwok(seed=seed, num_nodes=num_nodes, num_pods=num_pods, num_templates=num_templates,
                              node_affinity=node_affinity, tolerations=tolerations, unique_requests=unique_requests,
                              gang_size=gang_size, node_index_offset=node_index_offset, spread=spread, total_nodes=total_nodes)
        self._check(self._L.ykhost_generate_kwok(self._h, C.byref(cfg)))

    @property
    def num_nodes(self):
        return self._L.ykhost_num_nodes(self._h)

    @property
    def num_pods(self):
        return self._L.ykhost_num_pods(self._h)

    def pod_index(self, uid):
        return self._L.ykhost_pod_index(self._h, uid.encode())

    def node_index(self, name):
        return self._L.ykhost_node_index(self._h, name.encode())

    def dump_snapshot(self, pods=None, nodes=None, compact=False):
        """Snapshot JSON (text) of the selected pending pods / nodes — what the oracle is fed in parity tests. compact: runs
        of on-node pods sharing a template are written once with "replicas": k."""
        self._L.ykhost_set_dump_compact(self._h, 1 if compact else 0)
        pa = None if pods is None else np.ascontiguousarray(pods, dtype=np.int32)
        na = None if nodes is None else np.ascontiguousarray(nodes, dtype=np.int32)
        args = (pa.ctypes.data if pa is not None else None, 0 if pa is None else len(pa),
                na.ctypes.data if na is not None else None, 0 if na is None else len(na))
        need = self._L.ykhost_dump_snapshot(self._h, *args, None, 0)
        self._check(need)
        buf = C.create_string_buffer(need)
        self._check(self._L.ykhost_dump_snapshot(self._h, *args, buf, need))
        return buf.value.decode()

    def dump_documents(self, kind):
        """Newline-separated JSON documents (bytes) as the cache hooks would deliver them: 0 nodes, 1 pods on nodes, 2 asks."""
        need = self._L.ykhost_dump_documents(self._h, kind, None, 0)
        self._check(need)
        buf = C.create_string_buffer(need)
        self._check(self._L.ykhost_dump_documents(self._h, kind, buf, need))
        return buf.raw[:need - 1]

    def update_documents(self, kind, text):
        """ykhost_update_nodes_batch (kind 0) / ykhost_update_pods_batch on a buffer of documents."""
        fn = self._L.ykhost_update_nodes_batch if kind == 0 else self._L.ykhost_update_pods_batch
        return self._check(fn(self._h, text, len(text)))

    def ingest_stats(self):
        out = np.zeros(2, dtype=np.int64)
        self._L.ykhost_ingest_stats(self._h, out.ctypes.data)
        return {"template_reused": int(out[0]), "full_parse": int(out[1])}

    def ingest_timing(self):
        """Pod batches: scanning threads of the last parallel batch, ms of the parallel scan / of the ordered cache pass (summed)."""
        out = np.zeros(5, dtype=np.int64)
        self._L.ykhost_ingest_timing(self._h, out.ctypes.data)
        return {"threads": int(out[0]), "scan_ms": round(float(out[1]) / 1e3, 1), "apply_ms": round(float(out[2]) / 1e3, 1), "parallel_batches": int(out[3]), "bulk_batches": int(out[4])}

    def encoded_tables(self):
        """The structure-of-arrays tables the encoder produces (dict; masks as Python ints). Needs no device."""
        need = self._L.ykhost_encoded_tables_json(self._h, None, 0)
        self._check(need)
        buf = C.create_string_buffer(need)
        self._check(self._L.ykhost_encoded_tables_json(self._h, buf, need))
        t = json.loads(buf.value.decode())
        for k in ("taint_bits", "label_bits", "port_bits", "tolerated", "aff_terms", "pre_terms", "wanted_ports", "occupied_ports"):
            t[k] = [int(x, 16) for x in t[k]]
        return t

    def sync(self):
        self._check(self._L.ykhost_sync(self._h))

    def stats(self):
        out = np.zeros(8, dtype=np.int64)
        self._L.ykhost_stats(self._h, out.ctypes.data)
        keys = ["R", "KT", "W", "taints", "requirements", "templates", "specs", "encode_us"]
        return dict(zip(keys, out.tolist()))

    # ---- the PredicateManager interface ---------------------------------------------------------------------
    def _resolve(self, pod, node):
        p = pod if isinstance(pod, int) else self.pod_index(pod)
        n = node if isinstance(node, int) else self.node_index(node)
        if p < 0 or n < 0:
            raise KeyError("unknown pod or node")
        return p, n

    def predicates(self, pod, node, allocate):
        """Predicates(pod, node, allocate) → (plugin, err): ("", None) on fit, else (failing plugin, PredicateError)."""
        p, n = self._resolve(pod, node)
        plugin = C.create_string_buffer(64)
        msg = C.create_string_buffer(512)
        rc = self._L.ykhost_predicates(self._h, p, n, 1 if allocate else 0, plugin, 64, msg, 512)
        if rc == -13:
            raise UnsupportedAsk(msg.value.decode())
        self._check(rc)
        if rc == 1:
            return "", None
        return plugin.value.decode(), PredicateError(plugin.value.decode(), msg.value.decode())

    # ---- the scheduler-interface callbacks (by allocation key / node id) -------------------------------------
    def is_pod_fit_node(self, allocation_key, node_id, allocate):
        """Context.IsPodFitNode: None when the ask fits, else the error text the core receives (sentinel texts for an
        unknown pod / node, "failed plugin: '<name>'" + message otherwise)."""
        err = C.create_string_buffer(1200)
        rc = self._L.ykhost_is_pod_fit_node(self._h, allocation_key.encode(), node_id.encode(), 1 if allocate else 0, err, 1200)
        if rc == 1:
            return None
        if rc in (0, -10, -11):
            return err.value.decode()
        if rc == -13:
            raise UnsupportedAsk(err.value.decode())
        raise RuntimeError(err.value.decode() or self._L.ykhost_last_error(self._h).decode())

    def is_pod_fit_node_via_preemption(self, allocation_key, node_id, preempt_allocation_keys, start_index):
        """Context.IsPodFitNodeViaPreemption → (index, ok) as PreemptionPredicatesResponse{Index, Success}."""
        arr = (C.c_char_p * max(len(preempt_allocation_keys), 1))()
        for i, v in enumerate(preempt_allocation_keys):
            arr[i] = None if v is None else v.encode()
        r = self._L.ykhost_is_pod_fit_node_via_preemption(self._h, allocation_key.encode(), node_id.encode(), arr,
                                                          len(preempt_allocation_keys), start_index)
        if r < -1:
            raise RuntimeError(self._L.ykhost_last_error(self._h).decode())
        return r, r != -1

    def preemption_predicates(self, pod, node, victims, start_index):
        """PreemptionPredicates(pod, node, victims, startIndex) → index or -1. victims: UIDs (None = nil pod)."""
        p, n = self._resolve(pod, node)
        arr = (C.c_char_p * max(len(victims), 1))()
        for i, v in enumerate(victims):
            arr[i] = None if v is None else v.encode()
        r = self._L.ykhost_preemption_predicates(self._h, p, n, arr, len(victims), start_index)
        if r < -1:
            raise RuntimeError(self._L.ykhost_last_error(self._h).decode())
        return r

    def preemption_predicates_batch(self, queries):
        """queries: iterable of (pod, node, victim_uids, start_index) → list of indices; one device launch for all."""
        queries = list(queries)
        pods, nodes, off, starts, flat = [], [], [0], [], []
        for pod, node, victims, start in queries:
            p, n = self._resolve(pod, node)
            pods.append(p)
            nodes.append(n)
            flat.extend(victims)
            off.append(len(flat))
            starts.append(start)
        arr = (C.c_char_p * max(len(flat), 1))()
        for i, v in enumerate(flat):
            arr[i] = None if v is None else v.encode()
        pa, na = np.asarray(pods, dtype=np.int32), np.asarray(nodes, dtype=np.int32)
        oa, sa = np.asarray(off, dtype=np.int32), np.asarray(starts, dtype=np.int32)
        out = np.full(len(queries), -1, dtype=np.int32)
        self._check(self._L.ykhost_preemption_predicates_batch(self._h, len(queries), pa.ctypes.data, na.ctypes.data, oa.ctypes.data, arr,
                                                               sa.ctypes.data, out.ctypes.data))
        return out.tolist()

    def ask_supported(self, pod):
        """(True, "") when the engine evaluates pending pod #pod, else (False, reason): that ask goes to the CPU manager."""
        p = pod if isinstance(pod, int) else self.pod_index(pod)
        buf = C.create_string_buffer(600)
        rc = self._check(self._L.ykhost_ask_supported(self._h, p, buf, 600))
        return rc == 1, buf.value.decode()

    def candidates(self, pod, k, allocate=True):
        """The first k feasible nodes of the ask in bin-pack order, from the resident answer (needs a current evaluation
        with decisions; raises otherwise)."""
        p = pod if isinstance(pod, int) else self.pod_index(pod)
        out = np.full(max(k, 1), -1, dtype=np.int32)
        n = self._check(self._L.ykhost_candidates(self._h, p, 1 if allocate else 0, k, out.ctypes.data))
        return out[:n].copy()

    def allocate_round(self, asks=None, n=None, apply=True):
        """One scheduling round with conflict-resolved decisions (ykhost_allocate_round): ask i is decided with the earlier
        asks of the round assumed on their nodes. asks: ask indices in decision order (None: the first n asks, default all).
        → int32 array: node index, -1 = no node fits, -2 = routed to the CPU manager."""
        if asks is None:
            count = self.num_pods if n is None else int(n)
            ptr = None
        else:
            arr = np.ascontiguousarray(asks, dtype=np.int32)
            count, ptr = len(arr), arr.ctypes.data
        out = np.full(max(count, 1), -1, dtype=np.int32)
        self._check(self._L.ykhost_allocate_round(self._h, count, ptr, 1 if apply else 0, out.ctypes.data))
        return out[:count].copy()

    def device_errors(self):
        """Engine calls that failed on the device so far (each forces a full re-upload + full pass; see ykhost_device_errors)."""
        return int(self._L.ykhost_device_errors(self._h))

    def round_stats(self):
        out = np.zeros(4, dtype=np.int64)
        self._L.ykhost_round_stats(self._h, out.ctypes.data)
        return dict(zip(["rounds_on_device", "asks_on_device", "asks_one_by_one", "asks_routed"], out.tolist()))

    def resident_stats(self):
        out = np.zeros(5, dtype=np.int64)
        self._L.ykhost_resident_stats(self._h, out.ctypes.data)
        return dict(zip(["served_resident", "served_dirty_column", "served_query", "answer_fetches", "code_fetches"], out.tolist()))

    def peek_row(self, pod, allocate=True):
        """(row[row_words] uint64, count, decision) of one ask straight from the bitmap of the last evaluation (ykpred_peek_row)."""
        lay = self.layout()
        row = np.zeros(lay.row_words, dtype=np.uint64)
        cnt, dec = C.c_int32(0), C.c_int32(0)
        self._pcheck(self._P.ykpred_peek_row(self.engine, pod, self._masks[1] if allocate else self._masks[0],
                                             self._masks[3] if allocate else self._masks[2], row.ctypes.data, C.byref(cnt), C.byref(dec)))
        return row, cnt.value, dec.value

    def routing_stats(self):
        out = np.zeros(3, dtype=np.int64)
        self._L.ykhost_routing_stats(self._h, out.ctypes.data)
        return {"unsupported_asks": int(out[0]), "routed_to_cpu": int(out[1]), "dictionary_growths": int(out[2])}

    def pod_request(self, pod):
        buf = C.create_string_buffer(4096)
        self._check(self._L.ykhost_pod_request_json(self._h, pod, buf, 4096))
        return json.loads(buf.value.decode())

    # ---- batched evaluation ---------------------------------------------------------------------------------
    def evaluate(self, allocate=True, bitmap=True, counts=True, decisions=True, profile=False, direct=False):
        opts = (OUT_BITMAP if bitmap else 0) | (OUT_COUNTS if counts else 0) | (OUT_DECISIONS if decisions else 0)
        opts |= (EVAL_PROFILE if profile else 0) | (EVAL_DIRECT if direct else 0)
        self._check(self._L.ykhost_evaluate(self._h, 1 if allocate else 0, opts))

    def evaluate_dirty(self, allocate=True, counts=True, decisions=False, profile=False, spread_count_only=False, spread_counts_ready=False):
        """Patches only the node columns touched since the last evaluation (AssumePod / ForgetPod ...). Returns the
        number of columns re-evaluated, or -1 when a full evaluation had to be run instead. spread_count_only /
        spread_counts_ready: the two halves of the step on a node-sharded cluster whose host sums the topology histograms
        itself (with a communicator attached the engine does it inside one call)."""
        opts = OUT_BITMAP | (OUT_COUNTS if counts else 0) | (OUT_DECISIONS if decisions else 0) | (EVAL_PROFILE if profile else 0)
        opts |= (EVAL_SPREAD_COUNT_ONLY if spread_count_only else 0) | (EVAL_SPREAD_COUNTS_READY if spread_counts_ready else 0)
        n = C.c_int32(-1)
        self._check(self._L.ykhost_evaluate_dirty(self._h, 1 if allocate else 0, opts, C.byref(n)))
        return n.value

    def evaluate_into(self, bitmap=None, counts=None, decisions=None, keys=None, stream=None, allocate=True, profile=False,
                      direct=False, spread_count_only=False, spread_counts_ready=False):
        """ykpred_eval with caller-owned DEVICE outputs (objects exposing data_ptr(), e.g. torch tensors) on the
        caller's HIP stream — how a multi-GPU driver keeps the results where its collectives can reach them."""
        self.sync()
        a = _ffi.YkpredEvalArgs()
        a.prefilter_plugins = self._masks[1] if allocate else self._masks[0]
        a.filter_plugins = self._masks[3] if allocate else self._masks[2]
        a.options = OUT_BITMAP | OUT_COUNTS | OUT_DECISIONS | (OUT_DECISION_KEYS if keys is not None else 0)
        a.options |= (EVAL_PROFILE if profile else 0) | (EVAL_DIRECT if direct else 0)
        a.options |= (EVAL_SPREAD_COUNT_ONLY if spread_count_only else 0) | (EVAL_SPREAD_COUNTS_READY if spread_counts_ready else 0)
        a.bitmap = None if bitmap is None else bitmap.data_ptr()
        a.bitmap_rows = 0 if bitmap is None else int(bitmap.shape[0])  # a caller-owned bitmap states the rows it holds
        a.counts = None if counts is None else counts.data_ptr()
        a.decisions = None if decisions is None else decisions.data_ptr()
        a.decision_keys = None if keys is None else keys.data_ptr()
        a.stream = stream
        self._pcheck(self._P.ykpred_eval(self.engine, C.byref(a)))

    # ---- node-sharded clusters: RCCL exchanges behind the C ABI (include/ykpred.h) ---------------------------------
    @staticmethod
    def comm_unique_id():
        """ncclGetUniqueId through the C ABI: 128 bytes that rank 0 hands to the other shard processes out of band."""
        P = _ffi.load_ykpred()
        buf = (C.c_uint8 * 128)()
        if P.ykpred_comm_unique_id(buf) != 0:
            raise RuntimeError("ykpred_comm_unique_id: " + P.ykpred_last_error(None).decode())
        return bytes(buf)

    def set_row_stride(self, words):
        self._check(self._L.ykhost_set_row_stride(self._h, words))

    def set_row_capacity(self, rows):
        self._check(self._L.ykhost_set_row_capacity(self._h, rows))

    def row_map(self):
        """row_of_pod on the host: the physical bitmap row of every pending ask (rows are laid out for the writer)."""
        out = np.zeros(self.layout().num_pods, dtype=np.int32)
        self._pcheck(self._P.ykpred_read_row_map(self.engine, out.ctypes.data))
        return out

    def comm_init(self, unique_id, rank, world, node_offset):
        buf = (C.c_uint8 * 128).from_buffer_copy(unique_id)
        self._check(self._L.ykhost_comm_init(self._h, buf, rank, world, node_offset))

    def comm_destroy(self):
        self._check(self._L.ykhost_comm_destroy(self._h))

    def comm_info(self):
        """(rank, world, node_offset) of the communicator attached to the engine — (0, 1, 0) without one (ykpred_comm_info)."""
        r, w, o = C.c_int32(0), C.c_int32(1), C.c_int32(0)
        self._pcheck(self._P.ykpred_comm_info(self.engine, C.byref(r), C.byref(w), C.byref(o)))
        return int(r.value), int(w.value), int(o.value)

    def gather_bitmap(self, gathered=None, stream=None, compressed=False):
        """All-gather of the shard bitmaps of the last evaluation into [world][rows][row_stride] (device; engine-owned when
        `gathered` is None), on `stream`. compressed=True: the shards exchange their CLASS rows and every GPU expands the
        slabs locally (same result; raises when the shards' row layouts differ — then use the plain form)."""
        fn = self._P.ykpred_gather_bitmap_compressed if compressed else self._P.ykpred_gather_bitmap
        self._pcheck(fn(self.engine, None if gathered is None else gathered.data_ptr(), stream))

    def layout_hash(self):
        """Digest of the class / row layout: shards with equal digests can expand each other's class rows."""
        out = C.c_uint64(0)
        self._pcheck(self._P.ykpred_layout_hash(self.engine, C.byref(out)))
        return int(out.value)

    def collect_class_rows(self, out, stream=None):
        """out: device int64 tensor [num_classes][row_stride] <- the class rows of the last evaluation."""
        self._pcheck(self._P.ykpred_collect_class_rows(self.engine, out.data_ptr(), stream))

    def expand_class_rows(self, class_rows, bitmap_out, pod_class=None, stream=None):
        """bitmap_out (device [num_rows][row_stride]) <- the bitmap whose class rows are `class_rows`, in this engine's row
        layout. pod_class (device int32[P]): the pod -> class map `class_rows` is indexed by when it is a PEER's table."""
        self._pcheck(self._P.ykpred_expand_class_rows(self.engine, class_rows.data_ptr(), None if pod_class is None else pod_class.data_ptr(),
                                                      bitmap_out.data_ptr(), stream))

    def exchange_decisions(self, stream=None):
        """In place on the last evaluation's outputs: cluster-wide counts and GLOBAL best node per ask."""
        self._pcheck(self._P.ykpred_exchange_decisions(self.engine, stream))

    def read_gathered(self, shard, first=0, count=None):
        lay = self.layout()
        count = lay.num_pods - first if count is None else count
        out = np.zeros((count, lay.row_stride), dtype=np.uint64)
        self._pcheck(self._P.ykpred_read_gathered(self.engine, shard, first, count, out.ctypes.data))
        return out

    def spread_tensors(self):
        """(counts, present) int32 torch tensors VIEWING the engine's PodTopologySpread histograms on the device
        (zero-copy via __cuda_array_interface__) — what a node-sharded driver all-reduces between the two halves."""
        import torch
        lay = self.layout()

        class _View:
            def __init__(self, ptr, n):
                self.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i4", "data": (ptr, False), "version": 3}

        n = max(int(lay.spread_cells), 1)
        return (torch.as_tensor(_View(lay.spread_counts, n), device="cuda"),
                torch.as_tensor(_View(lay.spread_present, n), device="cuda"))

    @property
    def engine(self):
        return self._L.ykhost_engine(self._h)

    def _pcheck(self, rc):
        if rc != 0:
            raise RuntimeError("ykpred: " + self._P.ykpred_last_error(self.engine).decode())

    def layout(self):
        lay = _ffi.YkpredLayout()
        self._pcheck(self._P.ykpred_get_layout(self.engine, C.byref(lay)))
        return lay

    def synchronize(self):
        self._pcheck(self._P.ykpred_synchronize(self.engine))

    def read_bitmap(self, first=0, count=None):
        lay = self.layout()
        count = lay.num_pods - first if count is None else count
        out = np.zeros((count, lay.row_words), dtype=np.uint64)
        self._pcheck(self._P.ykpred_read_bitmap(self.engine, first, count, out.ctypes.data))
        return out

    def read_rows(self, pods):
        """Bitmap rows of the listed pods, [len(pods)][row_words] uint64 (one device gather + one copy)."""
        pa = np.ascontiguousarray(pods, dtype=np.int32)
        out = np.zeros((len(pa), self.layout().row_words), dtype=np.uint64)
        self._pcheck(self._P.ykpred_read_rows(self.engine, len(pa), pa.ctypes.data, out.ctypes.data))
        return out

    def pod_classes(self):
        """(pod_class[P], class_rep[C]): the engine's pod → class map and one representative pod per class (-1 = none)."""
        lay = self.layout()
        pc = np.zeros(lay.num_pods, dtype=np.int32)
        rep = np.zeros(lay.num_classes, dtype=np.int32)
        self._pcheck(self._P.ykpred_read_pod_classes(self.engine, pc.ctypes.data, rep.ctypes.data))
        return pc, rep

    def check_class_rows(self):
        """Number of bitmap words that differ from the row of the class representative, plus non-zero padding words."""
        v = C.c_uint64(0)
        self._pcheck(self._P.ykpred_check_class_rows(self.engine, C.byref(v)))
        return v.value

    def read_counts(self):
        out = np.zeros(self.layout().num_pods, dtype=np.int32)
        self._pcheck(self._P.ykpred_read_counts(self.engine, out.ctypes.data))
        return out

    def read_decisions(self):
        out = np.zeros(self.layout().num_pods, dtype=np.int32)
        self._pcheck(self._P.ykpred_read_decisions(self.engine, out.ctypes.data))
        return out

    def read_scores(self):
        out = np.zeros(self.layout().num_nodes, dtype=np.float64)
        self._pcheck(self._P.ykpred_read_scores(self.engine, out.ctypes.data))
        return out

    def checksum(self):
        v = C.c_uint64(0)
        self._pcheck(self._P.ykpred_bitmap_checksum(self.engine, C.byref(v)))
        return v.value

    def query(self, pods, nodes, allocate=True, pre_mask=None, filt_mask=None):
        """Batch of Predicates() answers evaluated on the device: returns (fit, plugin_code, reason) arrays."""
        pa = np.ascontiguousarray(pods, dtype=np.int32)
        na = np.ascontiguousarray(nodes, dtype=np.int32)
        fit = np.zeros(len(pa), dtype=np.uint8)
        code = np.zeros(len(pa), dtype=np.uint8)
        reason = np.zeros(len(pa), dtype=np.uint32)
        self.sync()
        pre = ALL_PLUGINS if pre_mask is None else pre_mask
        filt = ALL_PLUGINS if filt_mask is None else filt_mask
        self._pcheck(self._P.ykpred_query(self.engine, len(pa), pa.ctypes.data, na.ctypes.data, pre, filt, fit.ctypes.data,
                                          code.ctypes.data, reason.ctypes.data))
        return fit, code, reason

    def round_info(self):
        """ykpred_get_round_info: how the allocation rounds so far were decided (in batches / by the sequential kernel)."""
        out = np.zeros(6, dtype=np.int64)
        self._pcheck(self._P.ykpred_get_round_info(self.engine, out.ctypes.data))
        return dict(zip(("rounds_batched", "batched_asks", "batches", "exchanges", "rounds_sequential", "sequential_asks"), (int(v) for v in out)))

    def counters(self):
        out = np.zeros(6, dtype=np.int64)
        self._pcheck(self._P.ykpred_get_counters(self.engine, out.ctypes.data))
        return dict(zip(["full_evals", "node_patches", "row_patches", "queries", "gathers", "uploads"], out.tolist()))

    def timing(self):
        t = _ffi.YkpredTiming()
        self._pcheck(self._P.ykpred_last_timing(self.engine, C.byref(t)))
        return {"total_ms": t.total_ms,
                "kernels": [(t.kernel_name[i].decode(), t.kernel_ms[i]) for i in range(t.num_kernels)]}
