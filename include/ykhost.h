/*
 * ykhost.h — C API of libykhost.so: the host-side mirror of the reference's predicate interface.
 *
 * In the real system this layer is Go: a `gpuPredicateManager` in pkg/plugin/predicates that keeps an encoded
 * mirror of pkg/cache/external.SchedulerCache and talks to libykpred.so through cgo. Go is not
 * available in this image, so the same layer is written in C++ above the SAME C ABI (include/ykpred.h) and
 * exposed here so that tests and bench.py can drive it with Kubernetes-JSON objects. Names follow the reference:
 *
 *   ykhost_predicates            PredicateManager.Predicates(pod, node, allocate) (string, error)
 *                                /root/reference/pkg/plugin/predicates/predicate_manager.go:134-139
 *   ykhost_preemption_predicates PredicateManager.PreemptionPredicates(pod, node, victims, startIndex) int   (:141-179)
 *   ykhost_set_plugins           newPredicateManagerInternal(handle, resPre, allocPre, resFilt, allocFilt) (:378-424)
 *   ykhost_update_node / ykhost_remove_node / ykhost_update_pod / ykhost_remove_pod / ykhost_assume_pod /
 *   ykhost_forget_pod            SchedulerCache.UpdateNode / RemoveNode / UpdatePod / RemovePod / AssumePod / ForgetPod
 *                                /root/reference/pkg/cache/external/scheduler_cache.go:148-239,303-484
 *   ykhost_evaluate              the batched form of the core's loop over (ask, node) → Predicates()
 *
 * All functions return >= 0 on success, negative on error (text via ykhost_last_error). One lock per handle serialises
 * the entry points: the reference's callers hold read locks only (pkg/cache/context.go:697,709), so several threads may be
 * inside ykhost_predicates at once.
 */
#ifndef YKHOST_H_
#define YKHOST_H_
#include <stdint.h>

#include "ykpred.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ykhost ykhost_t;

/* NULL on failure (no GPU ⇒ failure). device < 0 makes a mirror-only handle: cache bookkeeping, request vectors and
 * snapshot dumps work, every evaluation fails (there is no CPU evaluation path). */
ykhost_t* ykhost_create(int32_t device, char* err, int32_t errlen);
void ykhost_destroy(ykhost_t* h);
const char* ykhost_last_error(const ykhost_t* h);

/* plugin lists of the four phases; defaults = NewPredicateManager (predicate_manager.go:321-373) */
int32_t ykhost_set_plugins(ykhost_t* h, uint32_t reservation_prefilters, uint32_t allocation_prefilters,
                           uint32_t reservation_filters, uint32_t allocation_filters);

/* --- cluster state (SchedulerCache mirror). JSON uses Kubernetes field names; see INTEGRATION.md. */
int32_t ykhost_load_snapshot(ykhost_t* h, const char* json); /* {"nodes":[{..., "pods":[...]}], "pods":[pending asks]} — replaces all state */
/* The six functions below restate the bookkeeping of scheduler_cache.go (podsMap / assignedPods / assumedPods /
 * orphanedPods), pinned by the scenarios of pkg/cache/external/scheduler_cache_test.go (tests/test_host_cache.py). */
int32_t ykhost_update_node(ykhost_t* h, const char* node_json);  /* → number of orphaned pods the (new) node adopted (:165-172) */
int32_t ykhost_remove_node(ykhost_t* h, const char* node_name);  /* → number of pods orphaned; assumed pods are reverted to pending (:205-219);
                                                                    unknown node: 0 */
/* → 1, or 0 when the pod was stored as an orphan (spec.nodeName names an unknown node, :352-360). spec.nodeName set ⇒
 * accounted on that node; unset ⇒ a pending ask (it inherits the node of the cached version, :336-339). status.phase
 * Running clears the assumed mark, Failed / Succeeded drops the pod (:344-347,374-383). */
int32_t ykhost_update_pod(ykhost_t* h, const char* pod_json);
int32_t ykhost_remove_pod(ykhost_t* h, const char* uid);         /* → 1 removed, 0 unknown uid (:390-417) */
/* The two update hooks for MANY objects at once — what the start-up replay of the informer caches amounts to
 * (Context.InitializeState, /root/reference/pkg/cache/context.go:1411-1484: every node, then every pod). `text` holds `len`
 * bytes of JSON documents one after the other (newline- or comma-separated; the body of a JSON array works): one cgo crossing,
 * one lock, and the pods share the template memo (pods of one Deployment / task group differ in name and uid only: all but the
 * first skip the JSON tree). → documents applied; on a malformed or rejected document -1 - (documents applied before it). */
int32_t ykhost_update_nodes_batch(ykhost_t* h, const char* text, int64_t len);
int32_t ykhost_update_pods_batch(ykhost_t* h, const char* text, int64_t len);
/* AssumePod: the cached pod gets spec.nodeName = node and is accounted there (moved from a node it was assumed on
 * before); the ask keeps its index (row) and is skipped by dump. Unknown pod / node: error. */
int32_t ykhost_assume_pod(ykhost_t* h, const char* uid, const char* node_name);
/* ForgetPod: the assumed mark is dropped; the cached pod keeps its node name and STAYS accounted on the node
 * (scheduler_cache.go:471-476 re-runs updatePod on the cached pod). → 1, or 0 for an unknown uid. */
int32_t ykhost_forget_pod(ykhost_t* h, const char* uid);
/* cache introspection: bit0 in podsMap, bit1 assigned (accounted on a node), bit2 assumed, bit3 orphan, bit4 holds an ask
 * row; node_out = spec.nodeName of the cached pod. 0 = unknown uid. */
int32_t ykhost_pod_state(ykhost_t* h, const char* uid, char* node_out, int32_t node_len);
int32_t ykhost_node_pod_count(ykhost_t* h, const char* node_name); /* len(NodeInfo.Pods), -1 = unknown node */

/* Gang scheduling. `task_groups_json` is the value of the pod annotation yunikorn.apache.org/task-groups (TaskGroup schema,
 * /root/reference/pkg/cache/amprotocol.go:47-57). ykhost_validate_task_groups restates GetTaskGroupsFromAnnotation +
 * validateTaskGroupResources (pkg/cache/utils.go:33-121: name / minMember / minResource present, no negative quantity, no
 * int64 overflow of minMember x minResource or of the cross-group aggregate, no cpu/vcore or explicit-"pods" collision) →
 * number of groups, negative = rejected (reason in ykhost_last_error). ykhost_add_task_groups also creates the minMember
 * placeholder pods of every group as pending asks, built like newPlaceholder (pkg/cache/placeholder.go:40-157): identical
 * labels (+ app id, queue), requests = minResource, nodeSelector, tolerations, affinity, topologySpreadConstraints — one
 * pod class per group. `app_json`: {"applicationId", "queue", "namespace"}. → number of placeholders created. */
int32_t ykhost_validate_task_groups(ykhost_t* h, const char* task_groups_json);
int32_t ykhost_add_task_groups(ykhost_t* h, const char* app_json, const char* task_groups_json);

/* The scheduler-interface callbacks as the core issues them — by allocation key (pod UID) and node id:
 * Context.IsPodFitNode (/root/reference/pkg/cache/context.go:696-716) behind AsyncRMCallback.Predicates
 * (pkg/cache/scheduler_callback.go:203-205). → 1 fit (err = ""), 0 does not fit (err = "failed plugin: '<name>'\n<message>",
 * the errors.Join of :713), YKHOST_E_POD_NOT_FOUND / YKHOST_E_NODE_NOT_FOUND with the texts of ErrorPodNotFound /
 * ErrorNodeNotFound (:66-69), YKHOST_E_NOT_AN_ASK for a cached pod that holds no ask row. */
#define YKHOST_E_POD_NOT_FOUND (-10)
#define YKHOST_E_NODE_NOT_FOUND (-11)
#define YKHOST_E_NOT_AN_ASK (-12)
#define YKHOST_E_UNSUPPORTED (-13) /* the ask is not evaluated by the engine (volumes, DRA claims, dictionary limits, specs the API
                                      server rejects): the caller's CPU PredicateManager answers this one; msg / err = the reason */
int32_t ykhost_is_pod_fit_node(ykhost_t* h, const char* allocation_key, const char* node_id, int32_t allocate, char* err, int32_t err_len);
/* Context.IsPodFitNodeViaPreemption (context.go:718-742) behind AsyncRMCallback.PreemptionPredicates
 * (scheduler_callback.go:207-216): → Index, or -1 for {Success: false} (no prefix of victims helps, unknown ask or node). */
int32_t ykhost_is_pod_fit_node_via_preemption(ykhost_t* h, const char* allocation_key, const char* node_id,
                                              const char* const* preempt_allocation_keys, int32_t num_keys, int32_t start_index);

/* synthetic KWOK-style cluster, replaces all state */
typedef struct ykhost_kwok {
  uint64_t seed;
  int32_t num_nodes;
  int32_t num_pods;
  int32_t num_templates;   /* distinct pod templates ("deployments"); 0 = every pod draws its own */
  int32_t node_affinity;   /* 0 = pods carry no nodeSelector/affinity (configs 1-2), 1 = config-3 mix */
  int32_t tolerations;     /* 0 = no tolerations and no node taints (config 1), 1 = KWOK taints + tolerations */
  int32_t unique_requests; /* 1 = adversarial: every pod a distinct cpu request (no signature sharing) */
  int32_t gang_size;       /* >0: pods are gang placeholders, `gang_size` identical members per task group */
  int32_t node_index_offset; /* node-sharded clusters: this shard holds global nodes [offset, offset+num_nodes); every node
                                draws from a stream keyed by its global index, so a shard holds exactly those nodes of the
                                unsharded cluster; pod draws depend on `seed` only (identical on all shards) */
  int32_t spread;            /* 1 = 10 % of the templates carry one DoNotSchedule zone-spread constraint (configs[4]) */
  int32_t total_nodes;       /* node-sharded clusters: node count of the whole cluster (0 = num_nodes); the asks' node pins
                                and matchFields names are drawn against it, so every shard holds the same asks */
  int32_t reserved[2];
} ykhost_kwok_t;
int32_t ykhost_generate_kwok(ykhost_t* h, const ykhost_kwok_t* cfg);

int32_t ykhost_num_nodes(const ykhost_t* h);
int32_t ykhost_num_pods(const ykhost_t* h); /* pending asks */
int32_t ykhost_pod_index(const ykhost_t* h, const char* uid);       /* index among pending asks, -1 = unknown */
int32_t ykhost_node_index(const ykhost_t* h, const char* node_name); /* -1 = unknown */

/* Serialises pending pods `pods[0..np)` (NULL = all) and nodes `nodes[0..nn)` (NULL = all, with their assigned pods)
 * as a snapshot document. Returns the required length (incl. NUL); writes at most `len` bytes. */
int64_t ykhost_dump_snapshot(ykhost_t* h, const int32_t* pods, int32_t np, const int32_t* nodes, int32_t nn, char* out, int64_t len);
/* The mirror's objects as the newline-separated documents the cache hooks would deliver: kind 0 = Node objects, 1 = pods on
 * nodes (spec.nodeName, status.phase Running), 2 = pending asks. Feeds ykhost_update_*_batch in bench.py's ingest leg and tests.
 * ykhost_ingest_stats: out[0] = pod documents that reused a known template without a parse, out[1] = full parses. */
int64_t ykhost_dump_documents(ykhost_t* h, int32_t kind, char* out, int64_t len);
int32_t ykhost_ingest_stats(ykhost_t* h, int64_t* out2);
/* Where the time of the pod batches went (summed over the handle's life): out[0] = scanning threads of the last batch that was
 * cut into pieces, out[1] = microseconds of the parallel scan, out[2] = microseconds of the ordered cache pass on the calling
 * thread (or of the bulk pass), out[3] = batches that took the parallel path, out[4] = those of them whose cache pass was the
 * bulk pass on every core (all pods new, everything to be re-encoded anyway — the start-up replay). */
int32_t ykhost_ingest_timing(ykhost_t* h, int64_t* out5);
/* on != 0: ykhost_dump_snapshot writes runs of on-node pods that share a pod template once, with "replicas": k (the loaders
 * expand them; their uids get a "#r" suffix) — what keeps the dump of a 50 000-node cluster small enough to hand to the
 * oracle for full-grid parity. */
int32_t ykhost_set_dump_compact(ykhost_t* h, int32_t on);

/* The encoded (structure-of-arrays) tables that cross the C ABI, as JSON (64-bit masks as hex strings); works on a
 * mirror-only handle. For encoder tests on machines without a device. Returns the required length like dump_snapshot. */
int64_t ykhost_encoded_tables_json(ykhost_t* h, char* out, int64_t len);

/* encode + upload whatever changed since the last sync (called implicitly by the functions below) */
int32_t ykhost_sync(ykhost_t* h);
ykpred_engine_t* ykhost_engine(ykhost_t* h); /* the underlying engine, for layout / readback / timing calls */

/* Node-sharded clusters (one handle per GPU, each mirroring a contiguous shard of the nodes and all asks): the shards agree
 * on one bitmap row stride (that of the largest shard) so that ykpred_gather_bitmap yields the regular layout
 * [world][P][row_stride]; ykhost_comm_init uploads the tables and attaches the RCCL communicator to the engine
 * (ykpred_comm_init). The gather / decision exchange are then called on ykhost_engine(h). */
int32_t ykhost_set_row_stride(ykhost_t* h, int32_t words /* multiple of 16; 0 = automatic */);
int32_t ykhost_set_row_capacity(ykhost_t* h, int32_t rows /* physical bitmap rows every shard allocates; 0 = automatic */);
int32_t ykhost_comm_init(ykhost_t* h, const uint8_t* id /* [YKPRED_COMM_ID_BYTES] */, int32_t rank, int32_t world, int32_t node_offset);
int32_t ykhost_comm_destroy(ykhost_t* h);

/* batched evaluation of every pending ask against every node: phase selects the plugin lists */
int32_t ykhost_evaluate(ykhost_t* h, int32_t allocate, uint32_t options /* YKPRED_OUT_* | YKPRED_EVAL_* */);

/* Like ykhost_evaluate, but when only node rows changed since the last evaluation of this phase (AssumePod / ForgetPod /
 * pods added to or removed from nodes) it patches just those node columns (ykpred_eval_nodes). *columns_patched receives
 * the number of columns re-evaluated, or -1 when a full evaluation was required. On a node-sharded handle (ykhost_comm_init)
 * whose asks carry topology constraints the call is COLLECTIVE: every shard's host makes it for the step, with or without
 * changes of its own — the engine exchanges the topology histograms inside. */
int32_t ykhost_evaluate_dirty(ykhost_t* h, int32_t allocate, uint32_t options, int32_t* columns_patched);

/* PredicateManager.Predicates for pending pod #pod on node #node. Returns 1 = fits ("", nil), 0 = error returned.
 * plugin receives the failing plugin name ("" when a PreFilter plugin rejected the pod), msg the status message. */
int32_t ykhost_predicates(ykhost_t* h, int32_t pod, int32_t node, int32_t allocate, char* plugin, int32_t plugin_len, char* msg,
                          int32_t msg_len);
/* The callbacks above are served from the RESIDENT answer whenever an evaluation of the phase is current: on the first
 * callback after ykhost_evaluate the class-compressed bitmap ([classes][row_words]; YKHOST_RESIDENT_MB, default 256, bounds it —
 * beyond that one row per ask) is mirrored to the host, and a callback becomes ask → class → bit. A node whose column changed
 * since (AssumePod between two asks) is answered per pair from the tables; a pair that does not fit costs one packed
 * whole-class query for its failing plugin (cached). Without a current evaluation: one ykpred_query_pod per ask, as before.
 * ykhost_candidates: the first k feasible nodes of the ask in bin-pack order — the node loop of the core in one call.
 * → nodes written, YKPRED_E_STATE (-4) if no evaluation WITH decisions is current. ykhost_resident_stats: out[0] callbacks served
 * from the mirror, [1] per pair (changed column), [2] by whole-ask queries, [3] answer fetches, [4] failing-plugin fetches. */
int32_t ykhost_candidates(ykhost_t* h, int32_t pod, int32_t allocate, int32_t k, int32_t* out_nodes /* [k] */);
int32_t ykhost_resident_stats(ykhost_t* h, int64_t* out5);

/* One scheduling ROUND with conflict-resolved decisions — what the core's loop decides for the asks asks[0..n) (indices into
 * the ask table, in decision order; NULL = asks 0..n-1): ask i is decided with asks 0..i-1 ASSUMED on their nodes (the core
 * decides an ask, AsyncRMCallback.UpdateAllocation → Context.AssumePod runs, scheduler_callback.go:49-98 / context.go:828-885,
 * and the next Predicates() call sees the node's new Requested / pod list). out_nodes[i] = node index, -1 = no node fits,
 * -2 = the ask is routed to the CPU manager (not evaluated by the engine) and takes no part in the round.
 * The whole round is ONE device call (ykpred_allocate_round): node resources, pod slots, host ports and the match counts behind
 * PodTopologySpread / InterPodAffinity are kept live on the device (the host uploads what a pod of every spec adds to its node:
 * ykpred_set_spec_effects); the engine runs it with its sequential kernel or in batches — parallel proposals, the loop replayed on
 * the host — whichever the ask list favours, with identical decisions (ykpred.h: ykpred_allocate_round, ykpred_get_round_info). Only when those effects are withheld (YKHOST_ROUND_ON_HOST=1, tests) does the host decide ask by ask
 * (decision → AssumePod → column patch → next decision) — that path needs apply != 0 to see its own allocations and refuses
 * apply == 0 with YKHOST_E_UNSUPPORTED.
 * NODE-SHARDED cluster (a communicator with world > 1 is attached to the engine): the call is COLLECTIVE — every rank calls with the
 * same asks in the same order — and node indices in out_nodes are indices in the WHOLE cluster (this shard's node 0 = its
 * node_offset), identical on every rank; an ask that went to another shard's node is assumed here as well (it leaves the pending
 * asks) without touching a node of this mirror. Topology constraints are part of the collective form (the owner of an accepted node
 * hands what its assume added to the histograms to the other shards: ykpred.h). A sharded round the collective form does not cover
 * (spec effects withheld; a pod that moves more histogram cells than a delta record holds) returns YKHOST_E_UNSUPPORTED — it is never
 * decided shard by shard.
 * apply != 0: every ask that got a node is assumed in the mirror exactly as ykhost_assume_pod would. → number of asks that got a
 * node, or a negative error. ykhost_round_stats: out[0] rounds decided by one device call, [1] asks decided in them, [2] asks
 * decided ask by ask, [3] asks routed. */
int32_t ykhost_allocate_round(ykhost_t* h, int32_t n, const int32_t* asks, int32_t apply, int32_t* out_nodes /* [n] */);
int32_t ykhost_round_stats(ykhost_t* h, int64_t* out4);

/* Engine calls that came back YKPRED_E_DEVICE / YKPRED_E_NOMEM so far (failed allocation, lost device). Each one marks the whole
 * device state stale: the failing call returns its error (Predicates() < 0: the Go manager routes the ask to the CPU predicate
 * manager — SURVEY.md §5, "must degrade, never fail scheduling"), the mirror stays intact, and the next ykhost_sync /
 * ykhost_evaluate re-uploads every table and runs a full pass. */
int64_t ykhost_device_errors(ykhost_t* h);

/* victims: UIDs of pods assigned to the node (NULL / unknown UID = nil victim). Returns the index or -1. */
int32_t ykhost_preemption_predicates(ykhost_t* h, int32_t pod, int32_t node, const char* const* victim_uids, int32_t num_victims,
                                     int32_t start_index);

/* Batched form: query q = (pods[q], nodes[q], victim UIDs victim_uids[victim_off[q] .. victim_off[q+1]), start_index[q]). */
int32_t ykhost_preemption_predicates_batch(ykhost_t* h, int32_t num_queries, const int32_t* pods, const int32_t* nodes,
                                           const int32_t* victim_off, const char* const* victim_uids, const int32_t* start_index,
                                           int32_t* out_index);

/* Individual routing: 1 = pending pod #pod is evaluated on the device, 0 = it is routed to the CPU predicate manager
 * (`reason` says why). ykhost_routing_stats: out[0] = asks marked unsupported at the last encode, out[1] = Predicates() calls
 * answered YKHOST_E_UNSUPPORTED so far (what the Go side exports as its fallback counter), out[2] = new asks whose selector
 * requirements were added to the dictionaries in place (one label-word column uploaded, nothing re-encoded). */
int32_t ykhost_ask_supported(ykhost_t* h, int32_t pod, char* reason, int32_t reason_len);
int32_t ykhost_routing_stats(ykhost_t* h, int64_t* out3);

/* request vector of pending pod #pod as JSON {"cpu": milli, "memory": bytes, ...} */
int32_t ykhost_pod_request_json(ykhost_t* h, int32_t pod, char* out, int32_t len);

/* encoder statistics: out[0]=R, [1]=KT, [2]=W, [3]=#taints, [4]=#requirements, [5]=#templates, [6]=#specs, [7]=last encode µs */
int32_t ykhost_stats(const ykhost_t* h, int64_t* out8);

#ifdef __cplusplus
}
#endif
#endif /* YKHOST_H_ */
