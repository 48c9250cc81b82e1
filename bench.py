#!/usr/bin/env python3
"""bench.py — pod×node predicate evaluations/s of the MI355X engine on BASELINE.json's metric config.

A "step" is ONE pass of the hot path over one snapshot: every pending ask against every node through the whole default
allocation-phase Filter set (NodeUnschedulable, NodeName, TaintToleration, NodeAffinity, NodeResourcesFit; the topology
plugins skip — no ask carries hard constraints unless --spread), producing the P×N feasibility bitmap, the per-ask
feasible-node count and the per-ask bin-pack decision. Tables are resident in HBM when the timed region starts.

  N = 1   workload = configs[2]: 50 000 nodes × 1 000 000 asks (KWOK-style synthetic, 2 000 pod templates).
  N > 1   workload = configs[3]: the SAME 50 000 nodes sharded N-way on the node axis (multiple-of-64 shards, one common
          row stride) × 1 000 000 gang-placeholder asks (10 000 task groups × 100 members) — STRONG scaling. A step =
          shard evaluation + all-gather of the shard bitmaps into [N][rows][row_stride] + the per-ask decision exchange
          (SUM count, MIN key, MIN global node), all behind the C ABI over RCCL. The gather moves the shards' CLASS rows
          over xGMI and expands the N slabs locally at HBM speed (ykpred_gather_bitmap_compressed; same layout as the
          plain ncclAllGather of the P-row bitmaps, which `--raw-gather` selects).
          `--weak` keeps round 1's weak-scaling variant (50 000 nodes PER GPU, decision exchange only).

Prints ONE JSON line (rank 0): `value` = whole-job evals/s of the timed steps; `roofline` describes the kernel with the
largest share of a step (HIP events on the launch stream) and the whole step; `variants` re-times the two other ask
populations of SURVEY.md §8d (every ask its own template; adversarial distinct request vectors); `end_to_end` is one cold
pass including encode + upload + class build; `cpu_baseline` is the oracle on the host cores.
"""
import argparse
import importlib
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)
SEED = 0x59554E49  # "YUNI"


def baseline_metric():
    """BASELINE.json's metric string (evals/sec is `value`, decisions/sec rides along as `decisions_per_sec`)."""
    try:
        with open(os.path.join(ROOT, "BASELINE.json")) as f:
            return json.load(f)["metric"]
    except (OSError, KeyError, ValueError):
        return "pod×node predicate evals/sec + decisions/sec, 50k nodes × 1M pods"


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--nodes", type=int, default=50_000, help="nodes of the cluster (with --weak: per GPU)")
    ap.add_argument("--pods", type=int, default=1_000_000)
    ap.add_argument("--templates", type=int, default=2000, help="distinct pod templates (0 = every pod draws its own)")
    ap.add_argument("--unique-requests", action="store_true", help="adversarial: a distinct cpu request per pod")
    ap.add_argument("--no-affinity", action="store_true", help="configs[1] plugin mix (no nodeSelector/affinity on pods)")
    ap.add_argument("--spread", action="store_true", help="configs[4] plugin mix: 10 %% of the templates carry a hard zone-spread constraint")
    ap.add_argument("--gang", type=int, default=-1, help="asks are gang placeholders, this many identical members per task group "
                                                          "(default: 0 at N=1, 100 at N>1 = configs[3])")
    ap.add_argument("--weak", action="store_true", help="N>1: weak scaling, --nodes per GPU, decision exchange only")
    ap.add_argument("--no-gather", action="store_true", help="N>1: leave the bitmap all-gather out of the step")
    ap.add_argument("--raw-gather", action="store_true",
                    help="N>1: all-gather the P-row shard bitmaps themselves over xGMI instead of their class rows + local expansion")
    ap.add_argument("--direct", action="store_true", help="time the per-pair kernel instead of the plane/class path")
    ap.add_argument("--cpu-seconds", type=float, default=12.0, help="budget of the cpu_baseline leg (0 = skip)")
    ap.add_argument("--profile-steps", type=int, default=5, help="steps run with per-kernel HIP events for `roofline`")
    ap.add_argument("--no-variants", action="store_true", help="N=1: skip the `variants` and `end_to_end` legs")
    ap.add_argument("--variant-steps", type=int, default=5)
    ap.add_argument("--no-ingest", action="store_true", help="N=1: leave the JSON ingest leg out of `end_to_end`")
    ap.add_argument("--no-configs4", action="store_true", help="N=1: leave the configs[4] (100k x 5M, one GPU) leg out of `variants`")
    ap.add_argument("--seed-offset", type=int, default=0, help="added to the workload seed (the configs[4] leg of the default run uses +2)")
    ap.add_argument("--no-verify", action="store_true", help="skip the checks that make every published number self-verifying (verify_leg)")
    return ap.parse_args()


def cpu_baseline(pm, budget_s, seed):
    """The reference's path cannot run here (Go); this times the oracle — a per-pair, object-model restatement of
    Predicates() — single-threaded (the core drives Predicates serially) on a bounded sample of the same workload."""
    if budget_s <= 0:
        return None
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    rng = np.random.default_rng(seed)
    n_nodes = min(pm.num_nodes, 4096)
    nodes = np.sort(rng.choice(pm.num_nodes, n_nodes, replace=False)).astype(np.int32)
    pods = np.sort(rng.choice(pm.num_pods, min(pm.num_pods, 16384), replace=False)).astype(np.int32)
    o = orc.Oracle(pm.dump_snapshot(pods=pods, nodes=nodes))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(8, dtype=np.int32), threads=1)
    per_pair = (time.perf_counter() - t0) / (8 * n_nodes)
    use = int(max(8, min(len(pods), budget_s / max(per_pair * n_nodes, 1e-9))))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(use, dtype=np.int32), threads=1)
    dt = time.perf_counter() - t0
    # second, stronger figure: the same per-pair port spread over every host core, on a sample sized for
    # about a quarter of the budget
    cores = os.cpu_count() or 1
    use_mt = int(min(len(pods), max(use, use * cores // 4)))
    t0 = time.perf_counter()
    o.eval_grid(pods=np.arange(use_mt, dtype=np.int32), threads=cores)
    dt_mt = time.perf_counter() - t0
    out = {"value": use * n_nodes / dt, "unit": "evals/s", "cores": 1, "kind": "port",
           "sample": f"{use} sampled pods x {n_nodes} sampled nodes of the same workload ({use * n_nodes} Predicates() calls, {dt:.1f} s)",
           "all_cores": {"value": use_mt * n_nodes / dt_mt, "cores": cores,
                         "sample": f"{use_mt} pods x {n_nodes} nodes, {dt_mt:.1f} s"}}
    # third figure: the strongest CPU formulation we know without the GPU path's planes / classes — the ENCODED tables
    # evaluated per pair with bitmask compares on every core (oracle/soa_cpu.c, checked against the oracle in tests)
    try:
        import _soa_cpu
        mirror = importlib.import_module("yunikorn-k8shim_amd").GpuPredicateManager(device=-1)
        mirror.load_snapshot(pm.dump_snapshot(pods=pods, nodes=nodes))
        tables = mirror.encoded_tables()
        mirror.close()
        if not tables["KD"] and not tables["spread_constraints"]:
            prepared = _soa_cpu.prepare(tables)
            _soa_cpu.run(prepared, orc.ALL, orc.ALL, threads=cores)
            reps, t0 = 0, time.perf_counter()
            while time.perf_counter() - t0 < max(1.0, budget_s / 6):
                _soa_cpu.run(prepared, orc.ALL, orc.ALL, threads=cores)
                reps += 1
            dt_soa = time.perf_counter() - t0
            out["soa_all_cores"] = {"value": reps * tables["P"] * tables["N"] / dt_soa, "cores": cores, "kind": "encoded tables, per pair, OpenMP",
                                    "sample": f"{reps} passes over {tables['P']} pods x {tables['N']} nodes, {dt_soa:.1f} s"}
    except Exception as exc:  # noqa: BLE001 - the extra figure must never break the bench line
        out["soa_all_cores"] = {"error": str(exc)}
    return out


def verify_leg(pm, seed=11, n_classes=64, use_direct=True, oracle=True):
    """Makes a published number self-verifying (outside every timed region): the bitmap the timed steps produced is
      (1) class-consistent on the device — every one of the P rows equals the row of its class's representative, padding zero;
      (2) the oracle's on `n_classes` sampled pod classes x ALL nodes, per pair (predicate_manager.go:206-283; the per-pod PreFilter
          replay form with hard spread constraints) — rows, feasible counts, float64 bin-pack scores and decisions;
      (3) reproduced bit for bit (checksum of checksums) by the independent per-pair kernel k_direct.
    (1) + (2) say: all rows of the sampled classes are the oracle's; (3) ties every other row to a second formulation."""
    t0 = time.perf_counter()
    out = {}
    lay = pm.layout()
    out["class_rows_bad_words"] = int(pm.check_class_rows())
    ok = out["class_rows_bad_words"] == 0
    plane_sum = pm.checksum()
    if oracle:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import _oracle as orc
        pod_class, rep = pm.pod_classes()
        rng = np.random.default_rng(seed)
        pick = np.sort(rng.choice(len(rep), min(n_classes, len(rep)), replace=False))
        reps = rep[pick].astype(np.int32)
        o = orc.Oracle(pm.dump_snapshot(pods=reps, compact=True))
        want = o.eval_grid(threads=min(os.cpu_count() or 8, 64), prefilter_once=True)
        rows = pm.read_rows(reps)
        got = np.unpackbits(rows.view(np.uint8), axis=1, bitorder="little")[:, :lay.num_nodes]
        rows_ok = bool(np.array_equal(got, want))
        counts, dec = pm.read_counts(), pm.read_decisions()
        counts_ok = bool(np.array_equal(counts[reps], want.sum(axis=1)))
        scores = o.binpack_scores()
        scores_ok = bool(np.array_equal(scores, pm.read_scores()))
        order = np.lexsort((np.arange(lay.num_nodes), scores))  # KWOK node names are zero-padded: NodeID order = index order
        in_order = want[:, order]
        first = in_order.argmax(axis=1)
        want_dec = np.where(in_order[np.arange(len(reps)), first] != 0, order[first], -1).astype(np.int32)
        dec_ok = bool(np.array_equal(dec[reps], want_dec))
        members_ok = bool(np.array_equal(dec, dec[rep][pod_class]) and np.array_equal(counts, counts[rep][pod_class]))
        out.update({"oracle_classes": int(len(reps)), "oracle_pairs": int(want.size), "rows": rows_ok, "counts": counts_ok, "scores_bit_equal": scores_ok,
                    "decisions": dec_ok, "members_share_count_and_decision": members_ok})
        ok = ok and rows_ok and counts_ok and scores_ok and dec_ok and members_ok
        o.close()
    if use_direct:
        pm.evaluate(direct=True, counts=False, decisions=False)
        pm.synchronize()
        out["direct_checksum_equal"] = bool(pm.checksum() == plane_sum)
        ok = ok and out["direct_checksum_equal"]
        pm.evaluate()  # the plane path's answer again (later legs read it)
        pm.synchronize()
    out["seconds"] = round(time.perf_counter() - t0, 1)
    out["ok"] = bool(ok)
    return out


def node_row_bytes(pm):
    st = pm.stats()
    return 8 * 2 * st["R"] + 4 + 4 + 4 + 8 * st["KT"] + 8 * st["W"]


def algorithmic_bytes(pm, lay):
    """ALGORITHMIC bytes of one pass over the local table, SURVEY.md §8(d) to the letter:
    bytes(P, N) = P·N/8 (bitmap written once) + N·B_node (node SoA read once) + P·B_pod (pod table read once) + 4·P feasible counts
    + 8·P decisions, B_node = 8·R + 4 + 4 + 8·k + 8·w + 4·K, B_pod = 8·R + 8·k + 4 + T·8·w with the config's own R, k (taint words),
    w (label words), K (topology keys) and T = 4 selector terms per ask (the survey's figure for configs[2]). The implementation's
    own intermediates — signature planes, index rows, class tables — are NOT algorithmic bytes (round 5 added
    `plane_bytes` here and overstated the fraction of the small-class populations by 4 points)."""
    st = pm.stats()
    R, KT, W = st["R"], st["KT"], st["W"]
    b_node = 8 * R + 4 + 4 + 8 * KT + 8 * W + 4 * st.get("KD", 0)
    b_pod = 8 * R + 8 * KT + 4 + 4 * 8 * W
    return lay.num_pods * lay.row_words * 8 + lay.num_nodes * b_node + lay.num_pods * b_pod + 12 * lay.num_pods


def plane_bytes(lay):
    """Signature planes of one pass: 8 bytes per 64-node word, except the request-value rows of many-valued dimensions, which
    are index rows of one byte per word (layout.index_rows)."""
    return (lay.plane_rows - lay.index_rows) * lay.row_words * 8 + lay.index_rows * lay.row_words


def profile_kernels(pm, run_step, n):
    kern = {}
    for _ in range(max(n, 0)):
        run_step(True)
        for name, ms in pm.timing()["kernels"]:
            kern.setdefault(name, []).append(ms)
    return {k: float(np.mean(v)) for k, v in kern.items()}


BITMAP_WRITERS = ("k_expand_bands", "k_combine", "k_combine_wave", "k_walk_rows", "k_sweep_rows", "k_class_runs", "k_direct")


def measured_traffic(workload, pods, nodes):
    """profiles/traffic_r04.json (scripts/pmc_passes.sh + summarize_pmc.py: separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes,
    calibrated and corrected as MI355X_MICROARCH.md prescribes): HBM bytes per launch of every engine kernel for this population."""
    for name in ("traffic_r06.json", "traffic_r05.json", "traffic_r04.json", "traffic_r03.json"):
        try:
            with open(os.path.join(ROOT, "profiles", name)) as f:
                tj = json.load(f)
        except (OSError, ValueError):
            continue
        w = tj.get("workloads", {}).get(workload)
        if w and (w.get("pods", tj.get("pods")), w.get("nodes", tj.get("nodes"))) == (pods, nodes):
            return dict(w, source="profiles/" + name)
    return None


def roofline_of(kern, algo_bytes, ms_per_step, traffic=None, lay=None, b_node=0):
    """The kernel with the largest average duration + the whole step, both against the HBM peak.

    `achieved` prices the kernel with ITS algorithmic bytes: the band writer is charged the bitmap rows it writes
    (layout.band_rows), the class-by-class writer the rest of the step's bytes (its rows + every input read once), k_walk_rows its
    rows + the index rows; the plane
    kernels are charged the planes they write plus the node table they read; kernels without a byte model here (the decision pass reads class rows out of L2 and
    is bound by the ordered scan, not by HBM) report achieved / frac = null — `whole_step_frac` always stands."""
    if not kern:
        return None
    dom = max(kern, key=kern.get)
    base = dom.split("(")[0]
    step_traffic = traffic_source = int_ops = None
    if isinstance(traffic, dict) and lay is not None:
        # "report int-ops/eval to expose ALU-boundness" — wave-level VALU instructions (SQ_INSTS_VALU, its own rocprofv3
        # --pmc pass) x 64 lanes / the evaluations of a step, for the dominant kernel and for the whole step
        kv = traffic.get("kernels_per_step", {}).get(base, {}).get("valu_insts")
        sv = traffic.get("step_valu_insts")
        pairs = float(lay.num_pods) * lay.num_nodes
        if kv is not None or sv is not None:
            int_ops = {"kernel": None if kv is None else round(kv * 64 / pairs, 4), "step": None if sv is None else round(sv * 64 / pairs, 4),
                       "unit": "VALU lane-operations per (pod, node) evaluation", "source": traffic.get("source")}
    if isinstance(traffic, dict):  # measured_traffic(): pick the dominant kernel's bytes per launch, keep the step's total
        # (the PMC passes are separate rocprofv3 runs — scripts/pmc_passes.sh; this line replays the committed summary for the same
        # workload and sizes, it does not count bytes itself)
        traffic_source = f"replayed from {traffic.get('source')} (separate rocprofv3 --pmc WRITE_SIZE / FETCH_SIZE passes of the same workload and sizes, not counted in this run)"
        step_traffic = traffic.get("step_hbm_bytes")
        k = traffic.get("kernels_per_step", {}).get(base)
        traffic = int(k["hbm_bytes"] / max(k.get("launches_per_step", 1.0), 1.0)) if k else None
    own = None
    band_bytes = (lay.band_rows * lay.row_words * 8) if lay is not None else 0
    if base == "k_expand_bands":
        own = band_bytes if lay is not None else algo_bytes
    elif base == "k_class_runs" and lay is not None:  # the rows it writes (a run's plane rows are read once per run, from L2)
        own = lay.run_rows * lay.row_words * 8
    elif base == "k_sweep_rows" and lay is not None:  # the rows it writes — it reads no index row and a plane row per run start
        own = lay.sweep_rows * lay.row_words * 8
    elif base == "k_walk_rows" and lay is not None:  # its bitmap rows + the index rows it decodes (one byte per word), nothing else
        own = (lay.num_pods - lay.sweep_rows) * lay.row_words * 8 - band_bytes + lay.index_rows_walked * lay.row_words
    elif base in BITMAP_WRITERS:
        own = max(algo_bytes - band_bytes, 0)
    elif lay is not None and base in ("k_sig_planes", "k_base_planes", "k_planes", "k_dim_walk"):
        own = plane_bytes(lay) + lay.num_nodes * b_node
    achieved = own / (kern[dom] * 1e-3) / 1e9 if own else None
    return {"bound": "hbm", "kernel": dom, "achieved": round(achieved, 1) if own else None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
            "frac": round(achieved / HBM_PEAK_GBS, 4) if own else None, "traffic": traffic, "avg_launch_ms": round(kern[dom], 4),
            "traffic_source": traffic_source,
            "algorithmic_bytes": int(own) if own else None, "step_algorithmic_bytes": int(algo_bytes), "step_traffic": step_traffic,
            "whole_step_frac": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            # (the same number under the name VERDICT r5 asked for: `step_algorithmic_bytes` IS the §8(d) formula since round 6)
            "whole_step_frac_8d": round(algo_bytes / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
            "int_ops_per_eval": int_ops}


def json_ingest_leg(pkg, dev, a, gang):
    """The path the Go manager drives (integration/gpu_predicate_manager.go): every Node and Pod of the default workload as JSON
    text through the batch forms of the cache hooks (Context.InitializeState's replay, /root/reference/pkg/cache/context.go:
    1411-1484: nodes, then every pod), then encode + upload + first evaluation — and the bitmap checksum of the rebuilt mirror
    against the directly generated one (asks that name a node are bound pods for the hooks, so the comparison runs without them)."""
    src = pkg.GpuPredicateManager(device=dev.index)
    out = {}
    try:
        src.generate_kwok(seed=SEED + 2, num_nodes=a.nodes, num_pods=a.pods, num_templates=a.templates, node_affinity=0 if a.no_affinity else 1,
                          spread=1 if a.spread else 0, gang_size=gang)
        t0 = time.perf_counter()
        docs = [src.dump_documents(k) for k in (0, 1, 2)]
        out["serialise_s"] = round(time.perf_counter() - t0, 2)
        out["documents"] = {"nodes": docs[0].count(b"\n"), "pods_on_nodes": docs[1].count(b"\n"), "pending_asks": docs[2].count(b"\n")}
        out["text_MB"] = round(sum(map(len, docs)) / 1e6, 1)
    finally:
        src.close()
    dst = pkg.GpuPredicateManager(device=dev.index)
    try:
        t0 = time.perf_counter()
        dst.update_documents(0, docs[0])
        t_nodes = time.perf_counter() - t0
        dst.update_documents(1, docs[1])
        dst.update_documents(2, docs[2])
        t_ingest = time.perf_counter() - t0
        dst.sync()
        t_sync = time.perf_counter() - t0
        dst.evaluate()
        dst.synchronize()
        t_all = time.perf_counter() - t0
        n_docs = sum(out["documents"].values())
        out.update({"json_ingest_ms": round(t_ingest * 1e3, 1), "nodes_ms": round(t_nodes * 1e3, 1), "us_per_document": round(t_ingest / max(n_docs, 1) * 1e6, 2),
                    "encode_upload_ms": round((t_sync - t_ingest) * 1e3, 1), "first_evaluation_ms": round((t_all - t_sync) * 1e3, 1),
                    "asks_mirrored": dst.num_pods, "templates": dst.stats()["templates"], "ingest": dst.ingest_stats(), "ingest_timing": dst.ingest_timing(),
                    "host_cores": len(os.sched_getaffinity(0)),
                    "note": "one cgo-shaped crossing per object kind (ykhost_update_nodes_batch / ykhost_update_pods_batch); the pod batches are "
                            "scanned on `ingest_timing.threads` threads (scan_ms), the cache pass runs on the caller's thread (apply_ms)"})
        try:
            out["steady_state_burst"] = steady_state_burst(dst, docs[1])
        except Exception as exc:  # noqa: BLE001 — a side leg never takes the line down
            out["steady_state_burst"] = {"error": str(exc)}
    finally:
        dst.close()
    return out


def steady_state_burst(dst, pod_docs):
    """Informer traffic on the LOADED, encoded mirror (context.go:184-193,320-352): every 10th pod document of the cluster again in one
    batch — an informer resync, every uid cached. The scanning threads resolve the uids, the ordered cache pass applies the batch."""
    lines = pod_docs.split(b"\n")
    burst = b"\n".join(lines[:-1][::10]) + b"\n"
    del lines
    t0 = time.perf_counter()
    n = dst.update_documents(1, burst)
    dt = time.perf_counter() - t0
    return {"documents": int(n), "ms": round(dt * 1e3, 1), "us_per_document": round(dt / max(n, 1) * 1e6, 2),
            "note": "every 10th pod document again on the loaded mirror (resync: uids cached, per-node / per-row bookkeeping live)"}


def device_rounds(pm, sizes, check_prefix):
    """Conflict-resolved rounds of the first n asks of a loaded manager for every n in `sizes` (apply = False: the tables stay as they
    are, so every round starts from the same state and the rounds are prefixes of one another), the first `check_prefix` decisions
    of each against ONE run of the oracle's sequential loop on one host core (its PreFilter-once form: the per-candidate form needs
    seconds per ask once topology constraints are on)."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    pm.evaluate(decisions=True)
    pm.synchronize()
    k = min(check_prefix, pm.num_pods)
    o = orc.Oracle(pm.dump_snapshot(pods=np.arange(k, dtype=np.int32), compact=True))
    t0 = time.perf_counter()
    want = o.allocate_sequential(prefilter_once=True)
    t_cpu = time.perf_counter() - t0
    out = []
    for n_asks in sizes:
        asks = np.arange(min(n_asks, pm.num_pods), dtype=np.int32)
        before = pm.round_stats()
        pm.allocate_round(asks=asks[:64], apply=False)  # (first call: scratch allocation, the specs' effects)
        info0 = pm.round_info()
        t0 = time.perf_counter()
        got = pm.allocate_round(asks=asks, apply=False)
        t_round = time.perf_counter() - t0
        after = pm.round_stats()
        info1 = pm.round_info()
        in_batches = info1["rounds_batched"] > info0["rounds_batched"]
        kk = min(k, len(asks))
        out.append({"nodes": pm.num_nodes, "asks": int(len(asks)), "allocated": int((got >= 0).sum()),
                    "distinct_nodes": int(len(np.unique(got[got >= 0]))),
                    "allocations_per_sec": len(asks) / t_round, "round_ms": round(t_round * 1e3, 2), "us_per_ask": round(t_round / len(asks) * 1e6, 2),
                    "on_device": bool(after["rounds_on_device"] == before["rounds_on_device"] + 2),
                    "form": "batched: parallel proposals, pair bits, host replay, node-by-node assume" if in_batches else "sequential kernel (one workgroup)",
                    "batches": int(info1["batches"] - info0["batches"]),
                    "asks_one_by_one": int(after["asks_one_by_one"] - before["asks_one_by_one"]),
                    "cpu_sequential_per_sec": k / t_cpu, "cpu_cores": 1, "checked_decisions": int(kk), "verified": bool(np.array_equal(got[:kk], want[:kk]))})
    return out


def allocation_round_leg(pkg, dev, big_pm=None, big_asks=2000):
    """CONFLICT-RESOLVED decisions — the metric's second half as the reference's loop defines it: the core decides an ask, the shim
    assumes it (context.go:828-885), the next Predicates() sees it. (a) The reference's own perf shape, scheduler_perf_test.go:62-66,
    283-352: 5 000 empty nodes of 16 000 m / 16 G / 110 pods x 50 000 asks of 10 m / 1 M — its "allocations/s". One
    ykhost_allocate_round call; every decision compared with the oracle run SEQUENTIALLY on one host core (timed beside it).
    (b) A round of the first `big_asks` asks of the main workload (configs[2]: 50 000 nodes), same check."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import _oracle as orc
    import _seqgen
    out = {}
    pm = pkg.GpuPredicateManager(device=dev.index)
    try:
        snap = _seqgen.perf_shape(5000, 50_000)
        pm.load_snapshot(snap)
        pm.evaluate(decisions=True)
        pm.synchronize()
        n = pm.num_pods
        text = pm.dump_snapshot()
        pm.allocate_round(n=64, apply=False)  # (first call: scratch allocation)
        t0 = time.perf_counter()
        got = pm.allocate_round(apply=False)
        t_round = time.perf_counter() - t0
        t0 = time.perf_counter()
        got2 = pm.allocate_round(apply=True)   # + the mirror's AssumePod bookkeeping for every allocation ...
        pm.evaluate_dirty(decisions=True)      # ... and the engine brought up to date (touched node rows, columns, decisions)
        pm.synchronize()
        t_cycle = time.perf_counter() - t0
        o = orc.Oracle(text)
        t0 = time.perf_counter()
        want = o.allocate_sequential()
        t_cpu = time.perf_counter() - t0
        out["reference_perf_shape"] = {
            "nodes": 5000, "asks": int(n), "allocated": int((got >= 0).sum()),
            "allocations_per_sec": n / t_round, "round_ms": round(t_round * 1e3, 2),
            "allocations_per_sec_incl_mirror_and_resync": n / t_cycle, "cycle_ms": round(t_cycle * 1e3, 2),
            "cpu_sequential_per_sec": n / t_cpu, "cpu_cores": 1, "cpu_kind": "port (oracle: per-pair Predicates() over a (score, NodeID)-ordered node set, first fit, AssumePod)",
            "verified": bool(np.array_equal(got, want) and np.array_equal(got2, want)), "stats": pm.round_stats()}
    finally:
        pm.close()
    if big_pm is not None:
        # a round of the first `big_asks` asks of the main workload, and the round that moves thousands of nodes: 20 000 asks of the
        # same workload. EVERY decision of both is checked (the moved-slot scan over thousands of slots lies behind
        # any short prefix): one run of the sequential oracle over all 20 000 asks, ≈ 40 s on one host core.
        # BENCH_ROUND_CHECK=<n> bounds the checked prefix (a prefix of a sequential round depends on nothing behind it).
        out["main_workload_round"], out["main_workload_round_20k"] = device_rounds(
            big_pm, [big_asks, 20_000], int(os.environ.get("BENCH_ROUND_CHECK", "20000")))
    out["definition"] = ("decisions/sec, conflict-resolved: ask i is decided with asks 0..i-1 of the round assumed on their nodes — identical to the "
                         "oracle run sequentially; `decisions_per_sec` of the line is the SNAPSHOT form (every ask against one state). `form`: the engine picks per round "
                         "between the sequential kernel and the batched form (ykpred.h, ykpred_allocate_round) — identical decisions")
    return out


def predicates_callback_leg(pm, P, N):
    """The seam the core really calls — Predicates(ask, node) one pair at a time (scheduler_callback.go:203-205) — served from the
    RESIDENT answer of a current evaluation (host mirror of the class rows): wall time per call through the ctypes
    binding (≈ 1-2 us of it is Python), a random pair per call and the core's pattern of one ask tried on many nodes."""
    pm.evaluate(decisions=True)
    pm.synchronize()
    rng = np.random.default_rng(3)
    pods, nodes = rng.integers(0, P, 2200), rng.integers(0, N, 2200)
    for i in range(200):  # (the first callback after an evaluation mirrors the class rows: outside the timed calls)
        pm.predicates(int(pods[i]), int(nodes[i]), True)
    t0 = time.perf_counter()
    for i in range(200, 2200):
        pm.predicates(int(pods[i]), int(nodes[i]), True)
    random_us = (time.perf_counter() - t0) / 2000 * 1e6
    t0 = time.perf_counter()
    calls = 0
    for pod in range(100, 110):
        for node in range(0, N, max(N // 200, 1)):
            pm.predicates(pod, node, True)
            calls += 1
    return {"resident_us_per_call_random_pair": round(random_us, 2),
            "resident_us_per_call_one_ask_many_nodes": round((time.perf_counter() - t0) / calls * 1e6, 2),
            "served": pm.resident_stats(),
            "note": "Predicates() through the C ABI + ctypes after a current evaluation; a failing pair fetches its plugin code from the device"}


def timed_leg(pkg, dev, stream, a, steps, warmup, profile_steps, **kwok):
    """A fresh manager on the same GPU: generate, upload, then `steps` timed full passes. Used for `variants` / `end_to_end`."""
    workload = kwok.pop("_workload", None)
    verify = kwok.pop("_verify", not a.no_verify)
    round_asks = kwok.pop("_round", 0)
    pm = pkg.GpuPredicateManager(device=dev.index)
    out = {}
    try:
        t0 = time.perf_counter()
        pm.generate_kwok(**kwok)
        out["generate_s"] = round(time.perf_counter() - t0, 2)
        P = pm.num_pods
        counts = torch.empty(P, dtype=torch.int32, device=dev)
        decisions = torch.empty(P, dtype=torch.int32, device=dev)
        # cold pass: encode (objects → dictionaries + tables), upload, pod classes, evaluation
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        pm.sync()
        t_sync = time.perf_counter() - t0
        pm.evaluate_into(counts=counts, decisions=decisions, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        t_cold = time.perf_counter() - t0
        for _ in range(warmup):
            pm.evaluate_into(counts=counts, decisions=decisions, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(steps):
            pm.evaluate_into(counts=counts, decisions=decisions, stream=stream.cuda_stream)
        torch.cuda.synchronize(dev)
        ms = (time.perf_counter() - t0) / steps * 1e3
        lay = pm.layout()
        kern = profile_kernels(pm, lambda prof: pm.evaluate_into(counts=counts, decisions=decisions, stream=stream.cuda_stream, profile=prof),
                               profile_steps)
        algo = algorithmic_bytes(pm, lay)
        out.update({"ms_per_step": round(ms, 4), "evals_per_sec": float(P) * lay.num_nodes / (ms * 1e-3), "pod_classes": lay.num_classes,
                    "signature_planes": lay.plane_rows, "distinct_evals_per_step": lay.num_classes * lay.num_nodes,
                    "roofline": roofline_of(kern, algo, ms, traffic=measured_traffic(workload, P, lay.num_nodes), lay=lay,
                                            b_node=node_row_bytes(pm)), "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
                    "cold_pass": {"encode_upload_ms": round(t_sync * 1e3, 1), "class_build_and_first_eval_ms": round((t_cold - t_sync) * 1e3, 1),
                                  "total_ms": round(t_cold * 1e3, 1), "encode_ms": round(pm.stats()["encode_us"] / 1e3, 1)}})
        if verify:
            # the timed steps wrote caller-owned counts / decisions; the checker reads the engine's own buffers: one more pass
            pm.evaluate()
            pm.synchronize()
            out["verification"] = verify_leg(pm)
            out["verified"] = out["verification"]["ok"]
        if round_asks:
            # a conflict-resolved round of this workload's first asks (configs[4]: hard spread constraints on a tenth of the
            # templates — the histograms move with every assumed pod, on the device)
            try:
                # (the oracle's PreFilter-once loop decides ≈ 30 asks/s at 100 000 nodes with spread constraints: 2 000 checked decisions)
                out["allocation_round"] = device_rounds(pm, [round_asks], int(os.environ.get("BENCH_ROUND_CHECK_TOPOLOGY", "2000")))[0]
            except Exception as exc:  # noqa: BLE001
                out["allocation_round"] = {"error": str(exc)}
    finally:
        pm.close()
    return out


def free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(a):
    """`python bench.py --gpus N` without a launcher: re-executes this script as N ranks of one node (one per GPU) under
    torch.distributed.run — the same command form the driver uses itself — so that the flag can never silently measure one GPU."""
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    have = torch.cuda.device_count()
    if backend == "nccl" and have < a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} needs {a.gpus} GPUs (RCCL wants one device per rank), {have} visible; "
                 f"BENCH_BACKEND=gloo runs the ranks on the GPUs there are with the torch.distributed reference exchanges")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    a = parse_args()
    if a.gpus < 1:
        sys.exit("bench.py: --gpus must be >= 1")
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(a)  # does not return
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != a.gpus:
        sys.exit(f"bench.py: --gpus {a.gpus} but the launcher started {world} rank(s): the line would report the wrong job")
    dist = None
    # BENCH_BACKEND=gloo lets the N>1 path be exercised on a single-GPU box (all ranks share cuda:0; the exchanges then use
    # the torch.distributed reference forms — RCCL refuses two ranks on one device). The driver's multi-GPU runs use the
    # default: one rank per GPU, nccl (= RCCL) for the bootstrap group, the data-path exchanges through the C ABI.
    backend = os.environ.get("BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank = local_rank % max(torch.cuda.device_count(), 1)
    if world > 1:
        import torch.distributed as dist
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    pkg = importlib.import_module("yunikorn-k8shim_amd")
    shard = importlib.import_module("yunikorn-k8shim_amd.sharding")
    strong = world > 1 and not a.weak
    gang = a.gang if a.gang >= 0 else (100 if strong else 0)
    if strong:
        ranges = shard.shard_ranges(a.nodes, world)
        total_nodes = a.nodes
    else:
        ranges = [(r * a.nodes, a.nodes) for r in range(world)]
        total_nodes = a.nodes * world
    first, count = ranges[rank]
    pm = pkg.GpuPredicateManager(device=local_rank)
    t_gen = time.perf_counter()
    kwok = dict(seed=SEED + 2 + a.seed_offset, num_pods=a.pods, num_templates=a.templates, node_affinity=0 if a.no_affinity else 1,
                unique_requests=1 if a.unique_requests else 0, spread=1 if a.spread else 0, gang_size=gang)
    pm.generate_kwok(num_nodes=count, node_index_offset=first, total_nodes=total_nodes, **kwok)
    if world > 1:
        pm.set_row_stride(shard.common_row_stride(ranges))
        pm.set_row_capacity(shard.common_row_capacity(a.pods))
    pm.sync()
    t_gen = time.perf_counter() - t_gen
    P, N = pm.num_pods, pm.num_nodes

    # ---- the exchanges of the N>1 step: C ABI over RCCL, or (single-GPU test rig / RCCL unavailable) torch.distributed
    collectives = None
    if world > 1:
        collectives = "c-abi rccl"
        if backend != "nccl" or os.environ.get("BENCH_COLLECTIVES") == "torch":
            collectives = "torch.distributed reference (" + backend + ")"
        else:
            try:
                shard.attach_communicator(pm, dist, rank, world, first)
            except Exception as exc:  # noqa: BLE001 - reported below; whether the run goes on is the caller's choice
                collectives = f"torch.distributed fallback (ykpred_comm_init failed: {exc})"
        flags_all = [None] * world
        dist.all_gather_object(flags_all, collectives)
        if any(f != "c-abi rccl" for f in flags_all) and collectives == "c-abi rccl":
            pm.comm_destroy()  # all ranks must take the same path
            collectives = "torch.distributed fallback (another rank could not create the communicator)"
        wanted_abi = backend == "nccl" and os.environ.get("BENCH_COLLECTIVES") != "torch"
        if wanted_abi and collectives != "c-abi rccl" and os.environ.get("BENCH_ALLOW_FALLBACK") != "1":
            # a scaling line produced over torch.distributed is not a measurement of the C-ABI path. The run
            # stops here with a non-zero exit and a line that says so (value null); BENCH_ALLOW_FALLBACK=1 runs the labelled fallback.
            if rank == 0:
                print(json.dumps({"metric": baseline_metric(), "value": None, "unit": "evals/s", "n_gpus": world, "steps": a.steps,
                                  "warmup": a.warmup, "ms_per_step": None, "higher_is_better": True, "scaling": "strong" if strong else "weak",
                                  "vs_baseline": None, "dtype": "int64", "data": "synthetic",
                                  "config": {"workload": "not run", "collectives": collectives},
                                  "error": "the C-ABI communicator (ykpred_comm_init over librccl) could not be created on every rank; "
                                           "set BENCH_ALLOW_FALLBACK=1 to time the torch.distributed reference exchanges instead",
                                  "per_rank": flags_all}))
            dist.barrier()
            dist.destroy_process_group()
            sys.exit(4)
    use_abi = collectives == "c-abi rccl"
    do_gather = strong and not a.no_gather

    # caller-owned outputs (torch tensors). Two sets at N>1: the exchanges of step k (comm stream) overlap the evaluation of
    # step k+1 (main stream).
    lay0 = pm.layout()
    nbuf = 2 if world > 1 else 1
    outs = [(torch.empty(P, dtype=torch.int32, device=dev), torch.empty(P, dtype=torch.int32, device=dev),
             torch.empty(P, dtype=torch.int64, device=dev)) for _ in range(nbuf)]
    rows_cap = shard.common_row_capacity(a.pods) if world > 1 else 0
    bitmaps = [torch.empty((rows_cap, lay0.row_stride), dtype=torch.int64, device=dev) for _ in range(nbuf)] if do_gather else [None] * nbuf
    gathered = torch.empty((world, rows_cap, lay0.row_stride), dtype=torch.int64, device=dev) if do_gather else None
    # A dedicated (non-default) stream: the engine launches on the stream it is handed, and handle 0 — torch's default
    # stream — would mean "use the engine's own stream", which the events below could not order against.
    stream = torch.cuda.Stream(device=dev)
    torch.cuda.set_stream(stream)
    comm = torch.cuda.Stream(device=dev) if world > 1 else None
    exchanged = [torch.cuda.Event() for _ in range(nbuf)]  # the exchange that last used buffer set b has finished
    gather_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(nbuf)]
    # C-ABI path: the shards exchange CLASS rows and every GPU expands the world slabs at HBM speed (ykpred_gather_bitmap_compressed);
    # --raw-gather moves the P-row bitmaps over xGMI instead. Same gathered layout either way.
    compressed = [use_abi and not a.raw_gather]
    gather_note = []
    gather_on = [True]  # switched off for the extra "step without the gather" leg after the timed region
    step_no = [0]

    def step(profile=False):
        b = step_no[0] % nbuf
        step_no[0] += 1
        counts, decisions, keys = outs[b]
        if world > 1:
            stream.wait_event(exchanged[b])  # the eval below overwrites set b: its previous exchange must be done
        pm.evaluate_into(bitmap=bitmaps[b], counts=counts, decisions=decisions, keys=keys if world > 1 else None,
                         stream=stream.cuda_stream, profile=profile, direct=a.direct)
        if world > 1:
            evaluated = torch.cuda.Event()
            evaluated.record(stream)
            comm.wait_event(evaluated)
            with torch.cuda.stream(comm):
                if do_gather and gather_on[0]:
                    gather_ev[b][0].record(comm)
                    if use_abi:
                        if compressed[0]:
                            try:
                                pm.gather_bitmap(gathered=gathered, stream=comm.cuda_stream, compressed=True)
                            except RuntimeError as exc:  # e.g. the shards built different layouts: the plain form always works
                                compressed[0] = False
                                gather_note.append(f"class-compressed gather refused ({exc}); plain all-gather used")
                        if not compressed[0]:
                            pm.gather_bitmap(gathered=gathered, stream=comm.cuda_stream)
                    else:
                        dist.all_gather_into_tensor(gathered.view(-1), bitmaps[b].view(-1))
                    gather_ev[b][1].record(comm)
                if use_abi:
                    pm.exchange_decisions(stream=comm.cuda_stream)
                else:
                    shard.ref_exchange_decisions(counts, decisions, keys, first, dist)
                exchanged[b].record(comm)

    def drain():
        if comm is not None:
            stream.wait_stream(comm)
        torch.cuda.synchronize(dev)

    for _ in range(a.warmup):
        step()
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step()
    drain()
    if dist:
        dist.barrier()
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    if dist:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / a.steps * 1e3

    lay = pm.layout()
    # ---- roofline (HIP events on the launch stream, bracketing each kernel)
    def prof_step(prof):
        step(profile=prof)
        drain()
    kern = profile_kernels(pm, prof_step, a.profile_steps)
    algo_bytes = algorithmic_bytes(pm, lay)
    traffic = None
    if world == 1 and not a.direct and not a.spread and gang == 0:
        traffic = measured_traffic("unique_request_vectors" if a.unique_requests else ("own_template_per_ask" if a.templates == 0 else
                                   ("default" if a.templates == 2000 else None)), P, N)
    roof = roofline_of(kern, algo_bytes, ms_per_step, traffic, lay=lay, b_node=node_row_bytes(pm))

    gather = None
    if do_gather:
        # extra leg, outside the timed region: the same step without the bitmap gather (shard evaluation + decision exchange)
        gather_on[0] = False
        step()
        drain()
        dist.barrier()
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(a.steps):
            step()
        drain()
        dist.barrier()
        torch.cuda.synchronize(dev)
        t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        no_gather_ms = float(t.item()) / a.steps * 1e3
        gather_on[0] = True
        g_ms = max(gather_ev[b][0].elapsed_time(gather_ev[b][1]) for b in range(nbuf))
        t = torch.tensor([g_ms], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        g_ms = float(t.item())
        nbytes = rows_cap * lay.row_stride * 8
        if compressed[0]:
            link = lay.num_classes * lay.row_stride * 8
            gather = {"mode": "class rows over xGMI + local expansion of the world slabs (ykpred_gather_bitmap_compressed)",
                      "shard_bytes": nbytes, "link_bytes_per_shard": link, "ms": round(g_ms, 3),
                      "expanded_GBps_per_gpu": round(nbytes * world / (g_ms * 1e-3) / 1e9, 1),
                      "layout": "[G][rows][row_stride] u64 (shard-major), identical to the plain all-gather",
                      "note": "device time of the last gather (max over ranks): collect class rows, ncclAllGather of "
                              f"{link} B per shard, {world} slab expansions by the writer kernels"}
        else:
            gather = {"mode": "plain ncclAllGather of the shard bitmaps" if use_abi else "all_gather_into_tensor of the shard bitmaps (torch.distributed reference)",
                      "shard_bytes": nbytes, "ms": round(g_ms, 3),
                      "recv_GBps_per_gpu": round(nbytes * (world - 1) / (g_ms * 1e-3) / 1e9, 1),
                      "per_peer_link_GBps": round(nbytes / (g_ms * 1e-3) / 1e9, 1), "layout": "[G][rows][row_stride] u64 (shard-major)",
                      "note": "device time of the last all-gather (max over ranks); every GPU receives one shard bitmap from each of its "
                              "world-1 peers, per_peer_link = one shard / that time" + ("; " + gather_note[0] if gather_note else "")}
        # every rank holds the same gathered bitmap: a wrapping 64-bit sum of its words, compared across the ranks (outside the timed region)
        checksums = [None] * world
        dist.all_gather_object(checksums, int(gathered.view(-1).sum().item()))
        gather["gathered_checksum"] = checksums[0]
        gather["ranks_hold_the_same_gathered_bitmap"] = bool(len(set(checksums)) == 1)
        gather["step_without_gather_ms"] = round(no_gather_ms, 4)
        gather["evals_per_sec_without_gather"] = float(P) * total_nodes / (no_gather_ms * 1e-3)

    # ---- the headline number verifies itself (outside the timed region): class rows, sampled classes x all nodes against the
    # oracle incl. decisions, and the independent per-pair kernel's checksum. Shards verify their own slab on the device.
    verification = None
    if not a.no_verify and not a.direct:
        try:
            pm.evaluate()
            pm.synchronize()
            verification = verify_leg(pm, oracle=(world == 1))
            if world > 1:
                flags = [None] * world
                dist.all_gather_object(flags, bool(verification["ok"]))
                verification["all_shards_ok"] = all(flags)
                verification["ok"] = bool(all(flags))
        except Exception as exc:  # noqa: BLE001 - a failed CHECK must show in the line, not take it down
            verification = {"ok": False, "error": str(exc)}
    cpu = cpu_baseline(pm, a.cpu_seconds, SEED) if rank == 0 else None
    callbacks = rounds = None
    if rank == 0 and world == 1 and not a.no_variants and not a.direct:
        try:
            callbacks = predicates_callback_leg(pm, P, N)
        except Exception as exc:  # noqa: BLE001 — a side leg never takes the line down
            callbacks = {"error": str(exc)}
        try:
            rounds = allocation_round_leg(pkg, dev, big_pm=pm if not a.spread else None)
        except Exception as exc:  # noqa: BLE001
            rounds = {"error": str(exc)}
    stats = pm.stats()
    try:
        rt = pm.routing_stats()
        routing = {"unsupported_asks": rt["unsupported_asks"], "fraction": rt["unsupported_asks"] / max(P, 1)}
    except Exception:  # noqa: BLE001
        routing = None
    comm_report = None
    if world > 1:
        # what every rank's engine says about its communicator (ykpred_comm_info): rank, world and first node of the shard — the
        # first multi-GPU run shows at a glance that N communicators of world N exist and that the shards tile the cluster
        mine = dict(zip(("rank", "world", "node_offset"), pm.comm_info())) if use_abi else {"rank": rank, "world": world, "node_offset": first, "c_abi": False}
        mine["nodes"] = N
        reports = [None] * world
        dist.all_gather_object(reports, mine)
        comm_report = reports
    if use_abi:
        pm.comm_destroy()
    pm.close()
    del outs, bitmaps, gathered
    torch.cuda.empty_cache()

    variants = end_to_end = None
    if rank == 0 and world == 1 and not a.no_variants and not a.direct:
        base = dict(seed=SEED + 2, num_nodes=a.nodes, num_pods=a.pods, node_affinity=0 if a.no_affinity else 1, spread=1 if a.spread else 0)
        variants = {}
        for name, kw in (("own_template_per_ask", dict(num_templates=0)),
                         ("unique_request_vectors", dict(num_templates=0, unique_requests=1))):
            try:
                variants[name] = timed_leg(pkg, dev, stream, a, a.variant_steps, 2, 2, **base, **kw, _workload=name)
            except Exception as exc:  # noqa: BLE001
                variants[name] = {"error": str(exc)}
        if (a.nodes, a.pods) == (50_000, 1_000_000) and not a.no_configs4:
            # BASELINE configs[4] whole on ONE GPU: 100 000 nodes x 5 000 000 asks, the full Filter set (10 % of the templates carry
            # a hard zone-spread constraint: PodTopologySpread PreFilter histograms + Filter) + bin-pack decisions; 62.6 GB bitmap
            try:
                variants["configs4_one_gpu"] = dict(
                    timed_leg(pkg, dev, stream, a, a.variant_steps, 1, 2, seed=SEED + 4, num_nodes=100_000, num_pods=5_000_000,
                              num_templates=a.templates, node_affinity=1, spread=1, _workload="configs4_one_gpu", _round=20_000),
                    workload="configs[4]: 100k nodes x 5M pods, full Filter set (NodeResourcesFit, TaintToleration, NodeAffinity, "
                             "PodTopologySpread DoNotSchedule on 10 % of the templates, ...) + bin-pack decisions, one GPU")
            except Exception as exc:  # noqa: BLE001
                variants["configs4_one_gpu"] = {"error": str(exc)}
        try:
            e2e = timed_leg(pkg, dev, stream, a, 1, 0, 0, **base, num_templates=a.templates, gang_size=gang, _verify=False)
            end_to_end = dict(e2e["cold_pass"], note="one cold pass of the default workload: objects → dictionaries/tables (encode) → H2D "
                                                      "upload → pod classes → evaluation, bitmap stays on the device")
        except Exception as exc:  # noqa: BLE001
            end_to_end = {"error": str(exc)}

    if rank == 0 and world == 1 and end_to_end is not None and "error" not in end_to_end and not a.no_ingest:
        try:
            end_to_end["json_ingest"] = json_ingest_leg(pkg, dev, a, gang)
        except Exception as exc:  # noqa: BLE001
            end_to_end["json_ingest"] = {"error": str(exc)}

    if rank == 0:
        evals = float(P) * float(total_nodes) * a.steps
        if world == 1:
            workload = ("configs[2]: 50k nodes x 1M pods, NodeResourcesFit + TaintToleration + NodeAffinity"
                        if (N, P, a.no_affinity, a.spread, gang) == (50_000, 1_000_000, False, False, 0) else
                        f"{N} nodes x {P} pods, affinity={'off' if a.no_affinity else 'on'}")
            parallelism = "single GPU"
        elif strong:
            workload = (f"configs[3]: {total_nodes} nodes sharded {world}-way x {P} gang-placeholder asks ({gang} members per task group), "
                        f"bitmap all-gather {'in' if do_gather else 'NOT in'} the step")
            parallelism = (f"node-axis shards x{world} ({ranges[0][1]} nodes per shard, last {ranges[-1][1]}; row stride {lay.row_stride} words); "
                           f"all-gather of shard bitmaps + decision exchange of step k overlap the evaluation of step k+1")
        else:
            workload = f"weak: {N} nodes/GPU x {P} pods, affinity={'off' if a.no_affinity else 'on'}"
            parallelism = f"node-axis shards x{world}; per-pod decision exchange of step k overlaps the evaluation of step k+1"
        out = {
            "metric": baseline_metric(), "value": evals / elapsed, "unit": "evals/s",
            "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": ms_per_step,
            "higher_is_better": True, "scaling": None if world == 1 else ("strong" if strong else "weak"), "vs_baseline": None, "dtype": "int64", "data": "synthetic",
            "config": {"workload": workload, "nodes_per_gpu": N, "total_nodes": total_nodes, "pods": P, "templates": a.templates,
                       "pod_classes": lay.num_classes, "signature_planes": lay.plane_rows, "unique_requests": bool(a.unique_requests),
                       "spread": bool(a.spread), "gang_size": gang, "path": "direct" if a.direct else "planes+combine",
                       "parallelism": parallelism, "collectives": collectives},
            "decisions_per_sec": float(P) * a.steps / elapsed,
            "distinct_evals_per_step": lay.num_classes * N,
            "verified": None if verification is None else bool(verification["ok"]), "verification": verification,
            "roofline": roof, "cpu_baseline": cpu,
            "kernel_ms": {k: round(v, 4) for k, v in kern.items()},
            "host_setup_s": round(t_gen, 2), "encode_ms": round(stats["encode_us"] / 1e3, 1),
            # asks the engine does NOT evaluate (volumes, DRA claims, a pod-affinity namespaceSelector with requirements, dictionary
            # overflow): routed to the CPU manager one by one — the fraction belongs beside every throughput figure (0 on the synthetic
            # KWOK populations, which carry none of those)
            "routed_asks": routing,
        }
        if comm_report is not None:
            out["communicators"] = comm_report
        if gather:
            out["bitmap_allgather"] = gather
            # the same step for consumers that need decisions + feasible counts only (the core's allocation loop): nothing but 16 bytes
            # per ask crosses the links (ykpred_exchange_decisions) — the configuration that can scale with the shard evaluation;
            # `python bench.py --gpus N --no-gather` times it as the line's own step
            out["decisions_only"] = {"ms_per_step": gather["step_without_gather_ms"], "evals_per_sec": gather["evals_per_sec_without_gather"],
                                     "decisions_per_sec": float(P) / (gather["step_without_gather_ms"] * 1e-3), "link_bytes_per_ask": 16,
                                     "note": "measured outside the timed region of this line (same engine, same outputs, gather switched off)"}
        elif world > 1:
            out["decisions_only"] = {"ms_per_step": ms_per_step, "evals_per_sec": evals / elapsed, "decisions_per_sec": float(P) * a.steps / elapsed,
                                     "link_bytes_per_ask": 16, "note": "this line: the bitmap gather is not in the step"}
        if variants is not None:
            out["variants"] = variants
        if end_to_end is not None:
            out["end_to_end"] = end_to_end
        if callbacks is not None:
            out["predicates_callback"] = callbacks
        if rounds is not None:
            out["allocation_round"] = rounds
            shape = rounds.get("reference_perf_shape") if isinstance(rounds, dict) else None
            if shape:
                # the reference's "allocations/s" includes the shim's AssumePod bookkeeping: the promoted figure is the whole cycle
                # (device round + the mirror's AssumePod for every allocation + the engine brought up to date); the device call
                # alone rides beside it
                out["allocations_per_sec"] = shape["allocations_per_sec_incl_mirror_and_resync"]
                out["allocations_per_sec_device_round_only"] = shape["allocations_per_sec"]
        print(json.dumps(out))
    if dist:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
