"""Table-driven checker of the ENCODED tables (the structure-of-arrays form documented in include/ykpred.h), for the
plugins that need no cluster-wide histogram: NodeAffinity PreFilter, NodeUnschedulable, NodeName, TaintToleration,
NodeAffinity, NodePorts, NodeResourcesFit, in the order and with the early exit of runFilterPlugins
(/root/reference/pkg/plugin/predicates/predicate_manager.go:206-283).

Test infrastructure only: it lets the CPU suite check the host ENCODER against the per-pair object-model results without
a device. It is written from the table layout, not from the HIP kernels.
"""
UNSCHED, NODE_NAME, TAINT, AFFINITY, PORTS, FIT, SPREAD, INTERPOD = 1, 2, 4, 8, 16, 32, 64, 128
SPEC_TOLERATES_UNSCHEDULABLE, SPEC_AFFINITY_SKIP, SPEC_PREFILTER_REJECT, SPEC_PREFILTER_NAMES, SPEC_UNSUPPORTED = 1, 2, 4, 8, 16
NO_NODE_NAME = -1


def _dnf(terms, t0, t1, lb, W):
    for t in range(t0, t1):
        if all((lb[w] & terms[t * W + w]) == terms[t * W + w] for w in range(W)):
            return True
    return False


def eval_pair(t, p, n, pre, filt):
    """(fit, failing plugin code) of ask row p on node n; codes as in ykpred.h (0 = PreFilter rejection / fit)."""
    N, R, KT, W, KP = t["N"], t["R"], t["KT"], t["W"], t["KP"]
    s, pin = t["pod_spec"][p], t["pod_node_name"][p]
    f = t["spec_flags"][s]
    if f & SPEC_UNSUPPORTED:  # not evaluated by the engine: routed to the CPU manager (YKPRED_CODE_UNSUPPORTED)
        return 0, 255
    lb = [t["label_bits"][w * N + n] for w in range(W)]
    if pre & AFFINITY and not f & SPEC_AFFINITY_SKIP:
        if f & SPEC_PREFILTER_REJECT:
            return 0, 0
        if f & SPEC_PREFILTER_NAMES and not _dnf(t["pre_terms"], t["pre_term_off"][s], t["pre_term_off"][s + 1], lb, W):
            return 0, 4
    if filt & UNSCHED and t["node_flags"][n] & 1 and not f & SPEC_TOLERATES_UNSCHEDULABLE:
        return 0, 1
    if filt & NODE_NAME and pin != NO_NODE_NAME and pin != n:
        return 0, 2
    if filt & TAINT and any(t["taint_bits"][k * N + n] & ~t["tolerated"][s * KT + k] for k in range(KT)):
        return 0, 3
    if filt & AFFINITY:
        skip = pre & AFFINITY and f & SPEC_AFFINITY_SKIP
        if not skip and not _dnf(t["aff_terms"], t["aff_term_off"][s], t["aff_term_off"][s + 1], lb, W):
            return 0, 4
    if filt & PORTS:
        if not pre & PORTS:
            return 0, 5
        want = [t["wanted_ports"][s * KP + k] for k in range(KP)]
        if any(want) and any(t["port_bits"][k * N + n] & want[k] for k in range(KP)):
            return 0, 5
    if filt & FIT:
        if not pre & FIT:
            return 0, 6
        if t["pod_count"][n] + 1 > t["allowed_pods"][n]:
            return 0, 6
        for r in range(R):
            q = t["requests"][s * R + r]
            if q > 0 and q > t["allocatable"][r * N + n] - t["requested"][r * N + n]:
                return 0, 6
    if filt & SPREAD and not pre & SPREAD:
        return 0, 7
    if filt & INTERPOD and not pre & INTERPOD:
        return 0, 8
    return 1, 0
