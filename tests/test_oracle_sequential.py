"""CPU: the oracle's sequential allocation loop (orc_allocate_sequential) against its own one-ask-at-a-time form — decide on a
freshly loaded snapshot, move the pod onto the node in the JSON, load again — and the host mirror's AssumePod bookkeeping
(mirror-only handle, no GPU) against the oracle's mutated state."""
import copy

import numpy as np
import pytest

import _oracle as orc
import _seqgen


def one_at_a_time(snap):
    snap = copy.deepcopy(snap)
    out = []
    for k in range(len(snap["pods"])):
        _, best = orc.Oracle(snap).decide(k)
        out.append(best)
        if best >= 0:
            pod = copy.deepcopy(snap["pods"][k])
            pod["spec"]["nodeName"] = snap["nodes"][best]["metadata"]["name"]
            snap["nodes"][best].setdefault("pods", []).append(pod)
    return np.array(out, dtype=np.int32), snap


@pytest.mark.parametrize("seed,kw", [(0, {}), (1, {"scalars": True}), (2, {"spread": True}), (3, {"ports": True}), (4, {"spread": True, "ports": True}),
                                     (5, {"ipa": True}), (6, {"ipa": True}), (7, {"spread": True, "ports": True, "ipa": True})])
def test_sequential_loop_equals_one_ask_at_a_time(seed, kw):
    snap = _seqgen.competing(seed, n_nodes=15, n_pods=40, **kw)
    o = orc.Oracle(snap)
    got = o.allocate_sequential()
    want, final = one_at_a_time(snap)
    assert np.array_equal(got, want)
    assert (got >= 0).sum() >= 5
    assert np.array_equal(orc.Oracle(snap).allocate_sequential(early_exit=False), want)  # the argmin form: same answers
    assert np.array_equal(orc.Oracle(snap).allocate_sequential(prefilter_once=True), want)  # PreFilter once per ask: same answers
    o2 = orc.Oracle(final)
    for n in range(o.num_nodes):
        assert o.node_info(n) == o2.node_info(n)


def test_perf_shape_fills_node_after_node():
    snap = _seqgen.perf_shape(20, 300)
    got = orc.Oracle(snap).allocate_sequential()
    assert got.tolist() == [k // 110 for k in range(300)]  # equal scores: NodeID order; a node that got a pod is the fullest: it goes first
