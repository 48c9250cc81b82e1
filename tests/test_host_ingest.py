"""The ingest path the Go manager drives: v1.Node / v1.Pod documents as JSON text through the cache hooks, one by one or as
one buffer (ykhost_update_nodes_batch / ykhost_update_pods_batch). The scanner of csrc/host/jsonscan.h lets a pod whose
template text was seen before skip the JSON tree; the result must be indistinguishable from the full parser's."""
import importlib
import json

import pytest

pkg = importlib.import_module("yunikorn-k8shim_amd")


@pytest.fixture()
def cluster_docs():
    src = pkg.GpuPredicateManager(device=-1)
    src.generate_kwok(seed=5, num_nodes=300, num_pods=4000, num_templates=120, node_affinity=1, spread=1)
    docs = [src.dump_documents(k) for k in (0, 1, 2)]
    n_asks = src.num_pods
    src.close()
    return docs, n_asks


def test_template_memo_path_equals_the_full_parser(cluster_docs):
    docs, _ = cluster_docs
    fast, slow = pkg.GpuPredicateManager(device=-1), pkg.GpuPredicateManager(device=-1)
    try:
        assert [fast.update_documents(k, docs[k]) for k in (0, 1, 2)] == [d.count(b"\n") for d in docs]
        st = fast.ingest_stats()
        assert st["template_reused"] > 10 * st["full_parse"] > 0
        # a harmless status.resize makes the scanner hand every pod to the full parser (in-place-resize inputs are its business)
        slow.update_documents(0, docs[0])
        for k in (1, 2):
            slow.update_documents(k, docs[k].replace(b'"status":{', b'"status":{"resize":"InProgress",'))
        assert slow.ingest_stats()["template_reused"] == 0
        assert fast.dump_snapshot() == slow.dump_snapshot()
        assert fast.encoded_tables() == slow.encoded_tables()
    finally:
        fast.close()
        slow.close()


def test_scanner_against_the_full_parser_on_reshaped_documents(cluster_docs, monkeypatch):
    """The one-pass scanner (jsonscan.h: scan_pod) against the tree parser on documents encoding/json would not produce but the API
    server may relay: members in another order (nodeName last, status first), whitespace between tokens, unknown members,
    escaped characters inside ignored strings — the mirror must not depend on which path read a pod."""
    import random
    docs, _ = cluster_docs
    rng = random.Random(7)

    def reshape(line):
        d = json.loads(line)
        meta, spec, status = d["metadata"], d["spec"], d.get("status", {})
        keys = list(spec)
        rng.shuffle(keys)
        if "nodeName" in spec and rng.random() < 0.5:
            keys.remove("nodeName")
            keys.append("nodeName")  # the last member: cut out without a trailing comma
        spec = {k: spec[k] for k in keys}
        if rng.random() < 0.3:
            spec["schedulerName"] = 'yuni"korn {x} [y]'
        meta = dict(reversed(list(meta.items())))
        if rng.random() < 0.3:
            meta["annotations"] = {"note": "brace } bracket ] quote \" backslash \\ done"}
        top = [("status", status), ("kind", "Pod"), ("metadata", meta), ("spec", spec)]
        rng.shuffle(top)
        seps = rng.choice([(",", ":"), (", ", ": "), (" ,\t", " : ")])
        return json.dumps(dict(top), separators=seps).encode()

    shaped = [b"\n".join(reshape(l) for l in docs[k].splitlines()) for k in (1, 2)]
    outcomes = []
    for threads in ("1", "4"):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
        fast, slow = pkg.GpuPredicateManager(device=-1), pkg.GpuPredicateManager(device=-1)
        try:
            fast.update_documents(0, docs[0])
            slow.update_documents(0, docs[0])
            for k, text in zip((1, 2), shaped):
                assert fast.update_documents(k, text) == text.count(b"\n") + 1
                forced = b"\n".join(json.dumps(dict(json.loads(l), status=dict(json.loads(l).get("status", {}), resize="InProgress"))).encode()
                                    for l in text.splitlines())
                assert slow.update_documents(k, forced) == text.count(b"\n") + 1
            assert slow.ingest_stats()["template_reused"] == 0 and fast.ingest_stats()["template_reused"] > 0
            assert fast.dump_snapshot() == slow.dump_snapshot()
            assert fast.encoded_tables() == slow.encoded_tables()
            outcomes.append(fast.dump_snapshot())
        finally:
            fast.close()
            slow.close()
    assert outcomes[0] == outcomes[1]


def test_documents_with_a_key_twice_read_like_the_tree_parser(cluster_docs):
    """a `spec` (or `metadata` / `status`) key that occurs twice is legal JSON and means last-wins to the tree parser.
    The one-pass scanner used to cut the nodeName member of the FIRST spec out of the text of the LAST one (a negative length:
    the document was rejected and the batch stopped) and to merge the fields of two metadata objects; such documents now take
    the full parser, and the mirror equals the one of the de-duplicated document."""
    docs, _ = cluster_docs
    lines = docs[1].splitlines()[:40]

    def twice(line, key):
        d = json.loads(line)
        first = dict(d[key])
        if key == "spec":
            first["nodeName"] = "some-other-node"   # an EARLIER spec with a nodeName member ...
            last = {k: v for k, v in d[key].items()}  # ... and the last one, the one that counts
        else:
            first = dict(first, name="shadowed-name", uid="shadowed-uid")
            last = d[key]
        body = json.dumps(d)
        dup = json.dumps({key: first})[1:-1] + ", " + body[1:]
        return ("{" + dup).encode()

    for key in ("spec", "metadata"):
        dup, plain = pkg.GpuPredicateManager(device=-1), pkg.GpuPredicateManager(device=-1)
        try:
            for m in (dup, plain):
                m.update_documents(0, docs[0])
            text = b"\n".join(twice(l, key) for l in lines)
            assert json.loads(text.splitlines()[0]) == json.loads(lines[0])  # (python's reader is last-wins as well)
            assert dup.update_documents(1, text) == len(lines), dup._L.ykhost_last_error(dup._h)
            assert plain.update_documents(1, b"\n".join(lines)) == len(lines)
            assert dup.dump_snapshot() == plain.dump_snapshot()
        finally:
            dup.close()
            plain.close()


def test_batch_equals_one_call_per_object(cluster_docs):
    docs, n_asks = cluster_docs
    one, batch = pkg.GpuPredicateManager(device=-1), pkg.GpuPredicateManager(device=-1)
    try:
        for d in docs[0].splitlines():
            one._L.ykhost_update_node(one._h, d)
        for k in (1, 2):
            for d in docs[k].splitlines():
                one._L.ykhost_update_pod(one._h, d)
        for k in (0, 1, 2):
            batch.update_documents(k, docs[k])
        pinned = sum(1 for d in docs[2].splitlines() if b'"nodeName"' in d)
        assert one.num_pods == batch.num_pods == n_asks - pinned  # an ask that names a node is a bound pod for the hooks
        assert one.dump_snapshot() == batch.dump_snapshot()
    finally:
        one.close()
        batch.close()


def test_real_world_shaped_documents_and_errors():
    """encoding/json output of real objects: members the mirror ignores (managedFields, ownerReferences, status.images ...),
    escaped strings, null members; a malformed document stops the batch at its position."""
    m = pkg.GpuPredicateManager(device=-1)
    try:
        node = {"kind": "Node", "metadata": {"name": "n1", "labels": {"zone": "a"}, "managedFields": [{"manager": "kubelet", "fieldsV1": {"f:x": {}}}]},
                "spec": {"taints": None, "podCIDR": "10.0.0.0/24"},
                "status": {"allocatable": {"cpu": "4", "memory": "8Gi", "pods": "10"}, "images": [{"names": ['a}]"b'], "sizeBytes": 1}] * 50,
                           "conditions": [{"type": "Ready", "status": "True", "message": 'kubelet is "ready" {ok}'}]}}
        text = json.dumps(node).encode().replace(b'"name": "n1"', b'"name": "n\\u0031"')  # an escaped name: still node n1
        assert m.update_documents(0, text) == 1 and m.node_index("n1") == 0
        pods = [{"metadata": {"name": f"p-{i}", "uid": f"u-{i}", "namespace": "ns", "labels": {"app": "x"}, "ownerReferences": [{"kind": "ReplicaSet", "name": "rs"}]},
                 "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": "1"}}}], "schedulerName": "yunikorn", "nodeName": None},
                 "status": {"phase": "Pending", "conditions": [{"type": "PodScheduled", "status": "False", "lastTransitionTime": f"2026-01-01T00:00:{i:02d}Z"}]}} for i in range(20)]
        assert m.update_pods_batch(pods) == 20 and m.num_pods == 20
        assert m.ingest_stats() == {"template_reused": 19, "full_parse": 1}
        weird = dict(pods[0], metadata=dict(pods[0]["metadata"], name='quo"te', uid="u-weird"))  # an escape in a captured field: full parser
        assert m.update_pods_batch([weird]) == 1 and m.pod_index("u-weird") == 20
        bad = json.dumps(pods[1]).encode() + b"\n{\"metadata\": {\"uid\": \"x\"\n" + json.dumps(pods[2]).encode()
        rc = m._L.ykhost_update_pods_batch(m._h, bad, len(bad))
        assert rc == -2 and b"malformed" in m._L.ykhost_last_error(m._h)  # -1 - (one document applied)
    finally:
        m.close()


def test_large_batch_takes_the_scanning_threads_and_equals_one_thread(cluster_docs, monkeypatch):
    """A batch of a few hundred KB is cut into pieces scanned by several threads (YKHOST_INGEST_THREADS pins the count: the
    build container reports more cores than it has); mirror, encoded tables and ingest counters equal the one-thread path's."""
    docs, _ = cluster_docs
    assert len(docs[1]) > 4 * 65536
    results = []
    for threads in ("1", "4"):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
        m = pkg.GpuPredicateManager(device=-1)
        try:
            assert [m.update_documents(k, docs[k]) for k in (0, 1, 2)] == [d.count(b"\n") for d in docs]
            results.append((m.dump_snapshot(), m.encoded_tables(), m.ingest_stats()))
        finally:
            m.close()
    assert results[0] == results[1]
    # a pretty-printed batch (raw newlines inside documents) cannot be cut: it takes the one-thread path and still loads
    monkeypatch.setenv("YKHOST_INGEST_THREADS", "4")
    pretty = b"\n".join(json.dumps(json.loads(d), indent=1).encode() for d in docs[2].splitlines()[:600])
    m = pkg.GpuPredicateManager(device=-1)
    try:
        m.update_documents(0, docs[0])
        assert m.update_documents(2, pretty) == 600
    finally:
        m.close()


def test_node_batch_parses_on_every_core_and_stops_where_one_thread_would(cluster_docs, monkeypatch):
    """ykhost_update_nodes_batch parses its documents on the scanning threads and applies them in order: the mirror equals the
    one-thread form's, and a document that does not parse (or a malformed one) stops the batch at its position — the nodes in
    front of it are in the cache, the ones behind it are not."""
    docs, _ = cluster_docs
    lines = docs[0].splitlines()
    assert len(lines) >= 256
    broken = b"\n".join(lines[:100] + [b'{"metadata": {"name": "half"'] + lines[100:])
    wrong = b"\n".join(lines[:70] + [b'{"metadata": {"name": 7, "labels": "x"}, "status": {"allocatable": {"cpu": "lots"}}}'] + lines[70:])
    seen = []
    for threads in ("1", "4"):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
        m = pkg.GpuPredicateManager(device=-1)
        try:
            assert m.update_documents(0, docs[0]) == len(lines)
            full = m.dump_snapshot()
        finally:
            m.close()
        outcome = [full]
        for text in (broken, wrong):
            m = pkg.GpuPredicateManager(device=-1)
            try:
                rc = m._L.ykhost_update_nodes_batch(m._h, text, len(text))
                outcome.append((rc, m._L.ykhost_last_error(m._h), m.num_nodes))
            finally:
                m.close()
        seen.append(outcome)
    assert seen[0] == seen[1]
    assert seen[0][1][0] == -101 and seen[0][1][2] == 100


def test_full_encode_on_several_threads_equals_one_thread(monkeypatch):
    """The node loop of a full encode runs on the host's cores when there are enough nodes and no selector classes (per-thread
    label-tuple memo; YKHOST_INGEST_THREADS pins the count); the pass over every pod that looks for required anti-affinity terms
    on running pods is skipped when no template of the pool carries one. The tables equal the one-thread encode's."""
    tables = []
    for threads in ("1", "5"):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
        m = pkg.GpuPredicateManager(device=-1)
        try:
            m.generate_kwok(seed=31, num_nodes=6000, num_pods=3000, num_templates=150, node_affinity=1, spread=0)
            tables.append(m.encoded_tables())
        finally:
            m.close()
    assert tables[0] == tables[1] and tables[0]["N"] == 6000 and tables[0]["KS"] == 0


@pytest.mark.parametrize("interpod", [False, True], ids=["shapes-stand-for-templates", "anti-affinity-visits-every-template"])
def test_full_encode_of_many_asks_on_several_threads_equals_one_thread(monkeypatch, interpod):
    """With >= 65 536 pending asks the full encode numbers the specs, finds the first template of every dictionary shape, prepares what
    those templates ask of the dictionaries, encodes the spec rows and places the variable-length columns on the host's cores; the
    ordered dictionary loop only consumes what was prepared. The asks here carry every dictionary kind the encoder knows (selector
    requirements and metadata.name fields, toleration lists, host ports, scalar resources, hard spread constraints, and — second
    case — pod (anti)affinity terms, which make the loop visit every template instead of one per shape). Tables, spec numbering and
    the routed asks equal the one-thread encode's, and so do they when thread creation fails half-way."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    import _gen
    snap = _gen.random_snapshot(77, 200, 66_000, scalars=True, spread=True, interpod=False)
    if interpod:  # (a few dozen templates with pod (anti)affinity terms: every one of them is matched against every ask's labels)
        extra = _gen.random_snapshot(78, 200, 48, scalars=True, spread=True, interpod=True)["pods"]
        for i, pod in enumerate(extra):
            pod["metadata"] = dict(pod["metadata"], name=f"ipa-{i}", uid=f"ipa-{i}")
        snap["pods"] = snap["pods"][:33_000] + extra + snap["pods"][33_000:]
    snap = json.dumps(snap)
    tables = []
    for threads, limit in (("1", None), ("6", None), ("6", "2")):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
        if limit is None:
            monkeypatch.delenv("YKHOST_TEST_THREAD_LIMIT", raising=False)
        else:
            monkeypatch.setenv("YKHOST_TEST_THREAD_LIMIT", limit)
        m = pkg.GpuPredicateManager(device=-1)
        try:
            m.load_snapshot(snap)
            assert m.num_pods >= 65_536
            tables.append((m.encoded_tables(), m.routing_stats()))
        finally:
            m.close()
    one = tables[0][0]
    assert len(one["spec_flags"]) > 8192 and one["KP"] > 0 and one["KD"] > 0 and one["R"] > 3 and one["W"] > 0
    assert tables[1] == tables[0] and tables[2] == tables[0]


def _load(monkeypatch, threads, batches):
    monkeypatch.setenv("YKHOST_INGEST_THREADS", threads)
    m = pkg.GpuPredicateManager(device=-1)
    try:
        counts = [m.update_documents(k, d) for k, d in batches]
        return counts, m.dump_snapshot(), m.encoded_tables(), m.ingest_stats(), m.ingest_timing()
    finally:
        m.close()


def test_bulk_cache_pass_equals_the_ordered_pass(cluster_docs, monkeypatch):
    """The start-up replay (every pod new, everything to be encoded anyway) takes the bulk cache pass — pod slots and ask rows by
    piece, the uid index by shard, the nodes' pod lists by node group, each on its own thread; anything it does not cover falls
    back to the ordered pass: a uid that occurs twice in the batch or is cached already, a terminated pod, a pod without a uid.
    Mirror, encoded tables and ingest counters equal the one-thread path's in every case; orphans go through the bulk pass. (A batch
    behind an ordered pass that recycled pod slots is ordered as well: the bulk pass does not reuse freed slots.)"""
    docs, _ = cluster_docs
    on_nodes, asks = docs[1].splitlines(), docs[2].splitlines()
    assert len(docs[1]) > 4 * 65536
    ghost = [json.dumps(dict(json.loads(d), spec=dict(json.loads(d)["spec"], nodeName="ghost-node"))).encode() for d in on_nodes[:40]]
    done = [json.dumps(dict(json.loads(d), status={"phase": "Succeeded"})).encode() for d in on_nodes[40:44]]
    cases = {
        "plain": ([(0, docs[0]), (1, docs[1]), (2, docs[2])], 2),
        "orphans": ([(0, docs[0]), (1, b"\n".join(ghost + on_nodes[40:])), (2, docs[2])], 2),
        "uid twice in the batch": ([(0, docs[0]), (1, b"\n".join(on_nodes + on_nodes[:30])), (2, docs[2])], 0),
        "uid cached already": ([(0, docs[0]), (1, docs[1]), (1, docs[1]), (2, docs[2])], 1),
        "terminated pod": ([(0, docs[0]), (1, b"\n".join(on_nodes[:40] + done + on_nodes[44:])), (2, docs[2])], 1),
    }
    for name, (batches, want_bulk) in cases.items():
        one = _load(monkeypatch, "1", batches)
        four = _load(monkeypatch, "4", batches)
        assert one[:4] == four[:4], name
        assert four[4]["bulk_batches"] == want_bulk and one[4]["bulk_batches"] == 0, (name, four[4])


def test_steady_state_bursts_equal_the_one_thread_pass_and_the_single_hooks(cluster_docs, monkeypatch):
    """Informer traffic WHILE scheduling (context.go:184-193,320-352), as batches on a mirror that is loaded and encoded: the scanning
    threads look every uid up in the (current) uid index and hand the ordered cache pass the version they found; the pass trusts that
    hint only while it still is a live pod of that uid. The burst mixes what can go wrong with it: unchanged pods (resync), asks that
    were bound meanwhile, ONE uid five times in a row moving from node to node (the third version is stored in the slot the first one
    vacated — the hint then points at the pod being applied), a pod that terminates and comes back under its uid, pods nobody has
    seen. Mirror, encoded tables and counters equal the one-thread batch and one hook call per document."""
    docs, _ = cluster_docs
    nodes = [json.loads(d)["metadata"]["name"] for d in docs[0].splitlines()]
    on_nodes, asks = docs[1].splitlines(), docs[2].splitlines()

    def on(doc, node, phase=None, uid=None):
        d = json.loads(doc)
        d["spec"] = dict(d["spec"], nodeName=node)
        if phase:
            d["status"] = dict(d.get("status") or {}, phase=phase)
        if uid:
            d["metadata"] = dict(d["metadata"], uid=uid, name=uid)
        return json.dumps(d).encode()

    wanderer = [on(on_nodes[7], nodes[k % len(nodes)]) for k in (3, 9, 4, 9, 1)]
    back = [on(on_nodes[11], nodes[2], phase="Succeeded"), on(on_nodes[11], nodes[5])]
    fresh = [on(asks[k], "", uid=f"fresh-{k}") for k in range(0, 60, 3)]
    bound = [on(a, nodes[(3 * i) % len(nodes)], phase="Running" if i % 2 else None) for i, a in enumerate(asks[100:1300:4])]
    burst = on_nodes[0:4000:5] + bound[:150] + wanderer[:2] + fresh[:10] + wanderer[2:] + back + bound[150:] + fresh[10:] + on_nodes[1:4000:7]
    text = b"\n".join(burst) + b"\n"
    assert len(text) > 2 * 65536  # (enough for two scanning threads)
    outcomes = []
    for mode in ("1", "4", "hooks"):
        monkeypatch.setenv("YKHOST_INGEST_THREADS", "4" if mode == "4" else "1")
        m = pkg.GpuPredicateManager(device=-1)
        try:
            for k in (0, 1, 2):
                m.update_documents(k, docs[k])
            m.encoded_tables()  # (a full encode: from here on the mirror keeps its per-row / per-node bookkeeping)
            before = m.ingest_timing()["parallel_batches"]
            if mode == "hooks":
                for doc in burst:
                    m.update_pod(json.loads(doc))
            else:
                assert m.update_documents(1, text) == len(burst)
                assert (m.ingest_timing()["parallel_batches"] - before) == (1 if mode == "4" else 0)
            outcomes.append((m.dump_snapshot(), m.encoded_tables(), m.num_pods))
        finally:
            m.close()
    assert outcomes[0] == outcomes[1], "four scanning threads vs one"
    assert outcomes[0] == outcomes[2], "the batch vs one hook call per document"


def test_threads_that_cannot_be_created_degrade_to_fewer_threads(cluster_docs, monkeypatch):
    """std::thread's constructor throws EAGAIN under a pids / thread cgroup limit. Every per-batch spawn goes
    through run_on_threads (host.cpp): the threads that did start — at worst the caller alone — take the items of the ones that
    did not, nothing is left joinable, nothing terminates. YKHOST_TEST_THREAD_LIMIT=k makes creation fail after k workers: the
    pod batches (scan + bulk cache pass), the node batch and the node loop of the full encode give the results of four real
    threads with zero and with one worker."""
    docs, _ = cluster_docs
    batches = [(0, docs[0]), (1, docs[1]), (2, docs[2])]
    want = _load(monkeypatch, "4", batches)
    for limit in ("0", "1"):
        monkeypatch.setenv("YKHOST_TEST_THREAD_LIMIT", limit)
        got = _load(monkeypatch, "4", batches)
        assert got[:4] == want[:4], limit
        assert got[4]["bulk_batches"] == want[4]["bulk_batches"] == 2  # the bulk pass itself ran, on fewer threads
    tables = []
    for limit in (None, "0", "2"):
        if limit is None:
            monkeypatch.delenv("YKHOST_TEST_THREAD_LIMIT", raising=False)
        else:
            monkeypatch.setenv("YKHOST_TEST_THREAD_LIMIT", limit)
        monkeypatch.setenv("YKHOST_INGEST_THREADS", "5")
        m = pkg.GpuPredicateManager(device=-1)
        try:
            m.generate_kwok(seed=31, num_nodes=6000, num_pods=3000, num_templates=150, node_affinity=1, spread=0)
            tables.append(m.encoded_tables())
        finally:
            m.close()
    assert tables[0] == tables[1] == tables[2]


def test_parallel_ingest_under_thread_sanitizer(cluster_docs, tmp_path):
    """libykhost's sources + tests/c/ingest_tsan.c under -fsanitize=thread: the scanning threads of the batch forms, the bulk cache
    pass, the full encode at the end (spec numbering, shape representatives, prepared dictionary items, node rows, spec rows and their
    placement — every one on the host's cores: >= 65 536 asks of their own templates) and two concurrent reader threads on the same
    handle produce no data-race report."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    lib = os.path.join(root, "yunikorn-k8shim_amd", "lib")
    pkg.build_all()
    src = pkg.GpuPredicateManager(device=-1)  # (>= 4096 nodes, >= 65 536 asks: the encode at the end of the harness takes its parallel loops)
    src.generate_kwok(seed=6, num_nodes=4500, num_pods=66_000, num_templates=0, node_affinity=1, spread=0)
    docs = [src.dump_documents(k) for k in (0, 1, 2)]
    src.close()
    paths = []
    for k, d in enumerate(docs):
        paths.append(str(tmp_path / f"docs{k}.ndjson"))
        with open(paths[-1], "wb") as f:
            f.write(d)
    exe = str(tmp_path / "ingest_tsan")
    san = ["-fsanitize=thread", "-fno-omit-frame-pointer", "-g", "-O1"]
    subprocess.check_call(["gcc", "-std=c11"] + san + ["-I" + os.path.join(root, "include"), "-c", os.path.join(root, "tests", "c", "ingest_tsan.c"),
                           "-o", str(tmp_path / "ingest_tsan.o")])
    subprocess.check_call(["g++", "-std=c++17"] + san + ["-I" + os.path.join(root, "include"), os.path.join(root, "yunikorn-k8shim_amd", "csrc", "host", "host.cpp"),
                           str(tmp_path / "ingest_tsan.o"), "-o", exe, "-L" + lib, "-lykpred", "-lpthread", "-Wl,-rpath," + lib])
    out = subprocess.run([exe] + paths, capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, YKHOST_INGEST_THREADS="4", TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0"))
    assert out.returncode == 0 and "ingest ok" in out.stdout and "bulk batches 2" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])
    assert "WARNING: ThreadSanitizer" not in out.stderr, out.stderr[-4000:]
