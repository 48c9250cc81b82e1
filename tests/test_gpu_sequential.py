"""Conflict-resolved (sequential) allocation rounds on the device against the oracle run SEQUENTIALLY — decide ask i, AssumePod
it, decide ask i + 1 (the loop yunikorn-core drives: scheduler_callback.go:203-205 → context.go:696-716, then
scheduler_callback.go:49-98 → context.go:828-885). Bar: every decision of the round identical, and the state the round leaves
behind (mirror + device tables after the assumes) identical to the oracle's mutated snapshot on the whole grid."""
import importlib
import json

import numpy as np
import pytest

import _oracle as orc
import _seqgen

pytestmark = pytest.mark.gpu
pkg = importlib.import_module("yunikorn-k8shim_amd")


@pytest.fixture(scope="module")
def pm():
    m = pkg.GpuPredicateManager()
    yield m
    m.close()


def unpack(bitmap, n):
    return np.unpackbits(bitmap.view(np.uint8), axis=1, bitorder="little")[:, :n]


def round_against_oracle(pm, snap, asks=None, expect_device=True):
    pm.load_snapshot(snap)
    before = pm.round_stats()
    o = orc.Oracle(pm.dump_snapshot())
    want = o.allocate_sequential(pods=asks)
    got = pm.allocate_round(asks=asks)
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{len(bad)} decisions differ, first at position {bad[0]}: gpu={got[bad[0]]} oracle={want[bad[0]]}"
    st = pm.round_stats()
    if expect_device:
        assert st["rounds_on_device"] == before["rounds_on_device"] + 1 and st["asks_one_by_one"] == before["asks_one_by_one"]
    else:
        assert st["asks_one_by_one"] > before["asks_one_by_one"]
    # the state the round left behind: the mirror after its assumes, evaluated on the device, against the oracle's mutated snapshot
    # (an assumed ask keeps its row in the ask table; the mirror dumps it under its node, not among the pending pods: the grid is
    # compared on the asks the round left pending)
    pm.evaluate(allocate=True)
    lay = pm.layout()
    o2 = orc.Oracle(pm.dump_snapshot())
    grid = o2.eval_grid(threads=8)
    listed = np.arange(lay.num_pods) if asks is None else np.asarray(asks)
    pending = np.setdiff1d(np.arange(lay.num_pods), listed[got >= 0])
    assert o2.num_pods == len(pending)
    assert np.array_equal(unpack(pm.read_bitmap(), lay.num_nodes)[pending], grid)
    for n in range(o.num_nodes):  # Requested / pod counts of every node: mirror == the oracle that ran the loop
        assert o.node_info(n) == o2.node_info(n), n
    return got


@pytest.mark.parametrize("seed", range(10))
def test_allocation_round_competing_asks(pm, seed):
    got = round_against_oracle(pm, _seqgen.competing(seed, scalars=bool(seed % 2)))
    assert (got >= 0).sum() > 10  # the round really allocates


@pytest.mark.parametrize("seed", range(3))
def test_allocation_round_in_a_given_order(pm, seed):
    snap = _seqgen.competing(100 + seed, n_nodes=25, n_pods=90)
    order = np.random.default_rng(seed).permutation(90)[:70].astype(np.int32)
    round_against_oracle(pm, snap, asks=order)


KINDS = {"spread": dict(spread=True), "ports": dict(ports=True), "ipa": dict(ipa=True), "all": dict(spread=True, ports=True, ipa=True)}


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("kind", sorted(KINDS))
def test_allocation_round_on_the_device_with_topology_constraints_and_host_ports(pm, seed, kind):
    """More than node resources couples the asks of these rounds: an assumed pod moves the PodTopologySpread / InterPodAffinity
    match counts of its topology domains (for every later ask whose selector it matches, in both directions of an anti-affinity
    rule) and occupies host ports on its node. The device round keeps that state live (ykpred_set_spec_effects) — every decision
    equals the oracle's sequential loop, and nothing is decided ask by ask on the host."""
    snap = _seqgen.competing(200 + 10 * seed + len(kind), n_nodes=24 + 3 * seed, n_pods=90, **KINDS[kind])
    got = round_against_oracle(pm, snap, expect_device=True)
    assert (got >= 0).sum() > 5


@pytest.mark.parametrize("seed,kind", [(0, "spread"), (1, "ports"), (2, "all")])
def test_allocation_round_goes_ask_by_ask_without_the_specs_effects(pm, seed, kind, monkeypatch):
    """The path node-sharded engines (and hosts that do not upload the specs' effects) still take: AssumePod + column patch per
    ask through the resident answer — same decisions. YKHOST_ROUND_ON_HOST withholds the effects."""
    monkeypatch.setenv("YKHOST_ROUND_ON_HOST", "1")
    snap = _seqgen.competing(200 + seed, n_nodes=20, n_pods=40, **KINDS[kind])
    round_against_oracle(pm, snap, expect_device=False)


def test_allocation_round_many_spread_constraints_move_their_minimum_at_once(pm):
    """Twenty templates spread over three zones by the SAME selector with twenty different maxSkew values: twenty constraints
    count the same pods, so their minima rise in the same ask — more at once than the workgroup's shared list holds (kRoundDirty),
    the rest are recomputed by their owner threads. Hostname-keyed anti-affinity on top (one domain per node)."""
    nodes = [{"metadata": {"name": f"n{i:03d}", "labels": {"zone": "abc"[i % 3], "kubernetes.io/hostname": f"n{i:03d}"}}, "spec": {},
              "status": {"allocatable": {"cpu": "8", "memory": "16Gi", "pods": "6"}}, "pods": []} for i in range(30)]
    pods = []
    for k in range(150):
        t = k % 20
        spec = {"containers": [{"name": "c", "resources": {"requests": {"cpu": "250m", "memory": "256Mi"}}}],
                "topologySpreadConstraints": [{"maxSkew": 1 + t, "topologyKey": "zone", "whenUnsatisfiable": "DoNotSchedule",
                                               "labelSelector": {"matchLabels": {"app": "shared"}}}]}
        if t % 5 == 0:
            spec["affinity"] = {"podAntiAffinity": {"requiredDuringSchedulingIgnoredDuringExecution": [
                {"labelSelector": {"matchLabels": {"tier": f"x{t}"}}, "topologyKey": "kubernetes.io/hostname"}]}}
        pods.append({"metadata": {"name": f"ask-{k}", "uid": f"ask-{k}", "namespace": "default", "labels": {"app": "shared", "tier": f"x{t}"}}, "spec": spec})
    got = round_against_oracle(pm, {"nodes": nodes, "pods": pods}, expect_device=True)
    assert (got >= 0).sum() > 60


def test_allocation_round_kwok_cluster_with_hard_spread_constraints(pm):
    """The configs[4] ask mix at test size: KWOK-style nodes in 16 zones, 6 000 asks of 60 templates, a tenth of the templates with
    a hard zone spread constraint — on the device, against the oracle's sequential loop."""
    pm.generate_kwok(seed=0x59554E49 + 11, num_nodes=400, num_pods=6000, num_templates=60, node_affinity=1, spread=1)
    before = pm.round_stats()
    o = orc.Oracle(pm.dump_snapshot())
    want = o.allocate_sequential(prefilter_once=True)
    got = pm.allocate_round()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    st = pm.round_stats()
    assert st["rounds_on_device"] == before["rounds_on_device"] + 1 and st["asks_one_by_one"] == before["asks_one_by_one"]
    assert (got >= 0).sum() > 1000 and pm.layout().num_classes > 60


@pytest.mark.parametrize("seed", range(4))
def test_allocation_round_runs_of_one_template(pm, seed):
    """Asks arrive template by template (a Deployment's replicas, a task group's members): a run of one spec lands on one node while
    it fits, and the device decides such a run in one step — as many asks as the node's free resources and pod slots hold, at most
    up to the next multiple of 64 of the round. Small nodes (3 to 8 pod slots, a few cores) end the runs early and in the middle
    of the 64-ask windows; pins, host ports and topology constraints inside the stream break them."""
    snap = _seqgen.competing(500 + seed, n_nodes=30, n_pods=400, ports=seed == 2, spread=seed == 3)
    snap["pods"].sort(key=lambda p: (p["metadata"]["labels"]["app"], p["metadata"]["name"]))
    got = round_against_oracle(pm, snap, expect_device=True)
    assert (got >= 0).sum() > 40


def test_allocation_round_kwok_cluster(pm):
    """KWOK-style nodes (random utilisation, 110 slots, taints, selectors) and 4 000 asks of 40 templates: long runs of asks pile
    onto the same node until its slots or resources run out."""
    pm.generate_kwok(seed=0x59554E49 + 7, num_nodes=300, num_pods=4000, num_templates=40, node_affinity=1)
    o = orc.Oracle(pm.dump_snapshot())
    want = o.allocate_sequential()
    got = pm.allocate_round()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    assert (got >= 0).sum() > 1000
    pm.evaluate(allocate=True)
    o2 = orc.Oracle(pm.dump_snapshot())
    for n in range(o.num_nodes):
        assert o.node_info(n) == o2.node_info(n), n


@pytest.mark.parametrize("kind", ["resources", "spread", "ports", "spread+ports"])
def test_allocation_round_moves_thousands_of_nodes(pm, kind):
    """the part of k_allocate_round that makes long rounds hard — candidate B over THOUSANDS of
    moved-node slots, 512 slots per step — checked on every decision: 3 000 nodes with 3 to 6 pod slots x 12 000 asks of 24
    templates in random order, ≈ 2 900 nodes receive a pod (six 512-slot steps), with and without hard spread constraints and
    host ports. Every decision equals the oracle's sequential loop, on the device, and so does the state the round leaves."""
    snap = _seqgen.small_slots(7, spread="spread" in kind, ports="ports" in kind)
    pm.load_snapshot(snap)
    before = pm.round_stats()
    o = orc.Oracle(pm.dump_snapshot())
    want = o.allocate_sequential(prefilter_once=True)
    got = pm.allocate_round()
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{len(bad)} decisions differ, first at position {bad[0]}: gpu={got[bad[0]]} oracle={want[bad[0]]}"
    st = pm.round_stats()
    assert st["rounds_on_device"] == before["rounds_on_device"] + 1 and st["asks_one_by_one"] == before["asks_one_by_one"]
    assert len(np.unique(got[got >= 0])) > 2000 and (got >= 0).sum() > 10000
    pm.evaluate(allocate=True)
    o2 = orc.Oracle(pm.dump_snapshot())
    for n in range(o.num_nodes):
        assert o.node_info(n) == o2.node_info(n), n


def _binpacking_cases():
    import os
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "binpacking_cases.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("case", _binpacking_cases(), ids=lambda c: c["name"])
def test_binpacking_e2e_node_order_on_the_device(pm, case):
    """Verify_BinPacking_Node_Order_Memory (test/e2e/bin_packing/bin_packing_test.go:52-189) through ykhost_allocate_round: the
    padding pods go where their nodeSelector sends them, job A (no requests) piles onto the node with the least available memory,
    job B (hostname anti-affinity to the padding pod assumed EARLIER IN THE SAME ROUND) onto the second — on the device."""
    pm.load_snapshot({"nodes": case["nodes"], "pods": case["pods"]})
    before = pm.round_stats()
    names = [n["metadata"]["name"] for n in case["nodes"]]
    got = pm.allocate_round()
    assert [names[i] if i >= 0 else None for i in got] == case["expect"], case["source"]
    st = pm.round_stats()
    assert st["rounds_on_device"] == before["rounds_on_device"] + 1 and st["asks_one_by_one"] == before["asks_one_by_one"]


@pytest.mark.parametrize("run_decide", [1, 0])
def test_allocation_round_on_a_population_of_sweep_runs(monkeypatch, run_decide):
    """Every ask its own cpu request (index rows, sweep runs): the snapshot decisions of these classes come from k_run_decide, which
    leaves the rank-ordered WINDOWS of their index rows unwritten — the round's class descriptors read every window, so the round
    completes them first (complete_windows). Every decision of the round against the oracle's sequential loop, with the run-level
    decisions on and off."""
    monkeypatch.setenv("YKPRED_TUNE", f"run_decide={run_decide}")
    m = pkg.GpuPredicateManager()
    try:
        m.generate_kwok(seed=0x59554E49 + 31, num_nodes=700, num_pods=4000, num_templates=0, node_affinity=1, unique_requests=1)
        m.evaluate(decisions=True)
        assert m.layout().sweep_rows > 3000
        o = orc.Oracle(m.dump_snapshot())
        dec = m.read_decisions()
        for p in range(0, 4000, 97):  # the snapshot decisions themselves
            assert o.decide(p)[1] == int(dec[p]), p
        want = o.allocate_sequential()
        got = m.allocate_round()
        assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
        assert (got >= 0).sum() > 500
    finally:
        m.close()


def test_allocation_round_reference_perf_shape(pm):
    """scheduler_perf_test.go's shape at a tenth of its size (the full 5 000 x 50 000 is bench.py's `allocation_round` leg): the
    asks fill node after node in NodeID order, 110 pods each."""
    snap = _seqgen.perf_shape(500, 5000)
    got = round_against_oracle(pm, snap)
    assert (got >= 0).all()
    counts = np.bincount(got, minlength=500)
    assert sorted(counts[counts > 0].tolist(), reverse=True)[:3] == [110, 110, 110] and (counts > 0).sum() == 46


def test_second_round_continues_from_the_first(pm):
    snap = _seqgen.competing(300, n_nodes=30, n_pods=100)
    pm.load_snapshot(snap)
    o = orc.Oracle(pm.dump_snapshot())
    first, second = np.arange(0, 50, dtype=np.int32), np.arange(50, 100, dtype=np.int32)
    want = np.concatenate([o.allocate_sequential(pods=first), o.allocate_sequential(pods=second)])
    got = np.concatenate([pm.allocate_round(asks=first), pm.allocate_round(asks=second)])
    assert np.array_equal(got, want)


def test_round_without_apply_leaves_the_cluster_alone(pm):
    snap = _seqgen.competing(301, n_nodes=30, n_pods=60)
    pm.load_snapshot(snap)
    before = json.loads(pm.dump_snapshot())
    a = pm.allocate_round(apply=False)
    b = pm.allocate_round(apply=False)
    assert np.array_equal(a, b) and json.loads(pm.dump_snapshot()) == before


@pytest.fixture(scope="module")
def pm_batched():
    """A manager whose rounds ALWAYS run in batches (YKPRED_TUNE round_batched=1, read when the engine is created): parallel
    proposals, pair bits, the host's replay, node-by-node assume — the form node-sharded engines use, here on one GPU."""
    import os
    old = os.environ.get("YKPRED_TUNE")
    os.environ["YKPRED_TUNE"] = "round_batched=1"
    try:
        m = pkg.GpuPredicateManager()
    finally:
        if old is None:
            del os.environ["YKPRED_TUNE"]
        else:
            os.environ["YKPRED_TUNE"] = old
    yield m
    m.close()


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("kind", ["plain"] + sorted(KINDS))
def test_batched_rounds_equal_the_oracle(pm_batched, seed, kind):
    """Batched rounds (round 6) against the oracle's sequential loop, every decision and the state left behind: competing asks of a
    few templates on small nodes — candidates that fill up inside a batch (the next list entry takes over), accepted nodes that
    overtake a later ask's candidate (pair bit + NodeResourcesFit on the exchanged columns), pins, and with `kind` host ports
    (an accepted node in front of a port ask ends the batch), PodTopologySpread / InterPodAffinity (an ask behind a contribution to
    a class it counts ends the batch; such batches are assumed ask after ask)."""
    kw = {} if kind == "plain" else KINDS[kind]
    snap = _seqgen.competing(900 + 10 * seed + len(kind), n_nodes=24 + 3 * seed, n_pods=120, scalars=bool(seed % 2), **kw)
    got = round_against_oracle(pm_batched, snap, expect_device=True)
    assert (got >= 0).sum() > 5


@pytest.mark.parametrize("seed", range(3))
def test_batched_rounds_in_a_given_order_and_in_runs(pm_batched, seed):
    snap = _seqgen.competing(950 + seed, n_nodes=30, n_pods=400, ports=seed == 1, spread=seed == 2)
    order = np.random.default_rng(seed).permutation(400)[:300].astype(np.int32)
    round_against_oracle(pm_batched, snap, asks=order)
    snap = _seqgen.competing(960 + seed, n_nodes=30, n_pods=400, ports=seed == 1, spread=seed == 2)
    snap["pods"].sort(key=lambda p: (p["metadata"]["labels"]["app"], p["metadata"]["name"]))  # (runs of one template: accepted at once)
    round_against_oracle(pm_batched, snap)


@pytest.mark.parametrize("kind", ["resources", "spread", "ports", "spread+ports"])
def test_batched_round_moves_thousands_of_nodes(pm_batched, kind):
    """The 12 000-ask round that moves 2 900 nodes (test_allocation_round_moves_thousands_of_nodes), in batches: every decision."""
    snap = _seqgen.small_slots(7, spread="spread" in kind, ports="ports" in kind)
    pm_batched.load_snapshot(snap)
    o = orc.Oracle(pm_batched.dump_snapshot())
    want = o.allocate_sequential(prefilter_once=True)
    got = pm_batched.allocate_round()
    bad = np.flatnonzero(got != want)
    assert bad.size == 0, f"{len(bad)} decisions differ, first at position {bad[0]}: gpu={got[bad[0]]} oracle={want[bad[0]]}"
    assert len(np.unique(got[got >= 0])) > 2000
    pm_batched.evaluate(allocate=True)
    o2 = orc.Oracle(pm_batched.dump_snapshot())
    for n in range(o.num_nodes):
        assert o.node_info(n) == o2.node_info(n), n


def test_batched_round_kwok_cluster_and_binpacking_pin(pm_batched):
    """KWOK-style nodes x 4 000 asks of 40 templates in batches, then the reference's bin_packing e2e (the decision-order pin)."""
    pm_batched.generate_kwok(seed=0x59554E49 + 7, num_nodes=300, num_pods=4000, num_templates=40, node_affinity=1)
    o = orc.Oracle(pm_batched.dump_snapshot())
    want = o.allocate_sequential()
    got = pm_batched.allocate_round()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    pm_batched.generate_kwok(seed=0x59554E49 + 8, num_nodes=2000, num_pods=6000, num_templates=700, node_affinity=1)
    o = orc.Oracle(pm_batched.dump_snapshot())
    want = o.allocate_sequential()
    got = pm_batched.allocate_round()
    assert np.array_equal(got, want), np.flatnonzero(got != want)[:5]
    for case in _binpacking_cases():
        pm_batched.load_snapshot({"nodes": case["nodes"], "pods": case["pods"]})
        names = [n["metadata"]["name"] for n in case["nodes"]]
        got = pm_batched.allocate_round()
        assert [names[i] if i >= 0 else None for i in got] == case["expect"], case["source"]


def test_round_fuzz_seeds():
    """A short slice of scripts/fuzz_rounds.py (random clusters and ask streams, every feature switched on at random, two rounds per
    cluster): the batched form, the sequential kernel and the oracle's loop agree on every decision. The long sweep
    (profiles/r06_fuzz_rounds.log: 3 000 seeds) runs outside the suite."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k != "YKPRED_TUNE"}
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "fuzz_rounds.py"), "720000", "60"], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert "60 ok, 0 bad" in r.stdout


@pytest.mark.parametrize("world,total_nodes,n_pods,n_templates,spread", [(2, 200, 600, 40, 0), (3, 330, 900, 6, 0), (2, 130, 500, 1, 0),
                                                                         (2, 256, 700, 40, 1), (3, 330, 600, 20, 1)],
                         ids=["two-shards", "three-shards-long-runs", "one-template-and-a-two-node-shard", "two-shards-hard-spread",
                              "three-shards-hard-spread"])
def test_allocation_rounds_on_a_node_sharded_cluster(tmp_path, world, total_nodes, n_pods, n_templates, spread):
    """Rounds on node-sharded engines (world 2 and 3 on this box's one GPU, the collectives through tests/c/rccl_stub.cpp): every shard
    proposes its 8 best nodes per ask of a batch (+ a bit per (ask, proposed node) pair), the proposals are all-gathered, every rank replays the loop —
    runs of one template land on one node while it fits — and the owners assume. Both rounds (apply = 1, then apply = 0 on top of it)
    equal the oracle's sequential loop over the whole cluster on every rank, in cluster-wide node indices, and stay on the device.
    hard-spread (round 6): a tenth of the templates carry a DoNotSchedule zone constraint — the histograms are cluster-wide state on
    every shard: the owner of an accepted node records what its assume added (delta record), a second all-gather hands it to the
    others (k_round_apply_deltas), and a batch ends in front of the first ask whose signature counts a class an accepted pod added to."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    stub = str(tmp_path / "librccl_stub.so")
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O1", "-fPIC", "-shared", "-std=c++17", os.path.join(root, "tests", "c", "rccl_stub.cpp"), "-o", stub, "-lrt"])
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(29800 + world * 11 + total_nodes % 83), os.path.join(root, "tests", "_shard_round_worker.py"),
           str(total_nodes), str(n_pods), str(n_templates), str(spread)]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=dict(os.environ, SHARD_RCCL_STUB=stub))
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-2500:])
    assert out.stdout.count("sharded rounds True on_device True") == world, (out.stdout[-1500:], out.stderr[-1500:])

