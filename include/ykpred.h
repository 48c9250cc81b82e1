/*
 * ykpred.h — C ABI of libykpred.so, the MI355X (gfx950) batched predicate engine.
 *
 * This is the drop-in boundary for the yunikorn-k8shim predicate hot path. A Go
 * `gpuPredicateManager` implementing the reference's
 *     type PredicateManager interface { EventsToRegister; Predicates; PreemptionPredicates }
 *     (/root/reference/pkg/plugin/predicates/predicate_manager.go:47-55)
 * binds these entry points through cgo (binding sketch: INTEGRATION.md) and is installed at the one
 * construction site /root/reference/pkg/cache/context.go:130; pkg/shim and the scheduler-interface
 * callback (/root/reference/pkg/cache/scheduler_callback.go:203-216) stay untouched.
 *
 * Conventions
 *   - plain C types, caller-owned flat arrays (structure-of-arrays), copied during the call: the library
 *     never retains a caller pointer (cgo rule: no Go pointers kept after return);
 *   - every function returns YKPRED_OK (0) or a negative YKPRED_E_* code; ykpred_last_error() gives text;
 *   - no CPU fallback exists in this library: if the HIP device is missing, create() fails;
 *   - evals may run concurrently with queries only if the caller serialises them against uploads — the same
 *     contract the reference gets from Context.lock / SchedulerCache.lock (context.go:697,709).
 *
 * What replaces what
 *   ykpred_set_nodes / ykpred_update_node   the per-call reads of framework.NodeInfo{Allocatable, Requested,
 *                                           Pods, Node().Spec/Labels} (scheduler_cache.go:84-145, a14 in SURVEY §8a)
 *   ykpred_set_specs / ykpred_set_pods      the pod side of every Predicates(pod, ...) call: the request vector of
 *                                           pkg/common/resource.go:56-109 and the toleration / node-selector ASTs,
 *                                           pre-encoded once per pod update instead of once per (pod,node) pair
 *   ykpred_eval                             the P×N loop of Predicates() calls: runPreFilterPlugins +
 *                                           runFilterPlugins (predicate_manager.go:221-283) for every pending
 *                                           ask against every node of one snapshot
 *   ykpred_query                            one Predicates() result: fit + first failing plugin (:206-219)
 *   ykpred_preemption                       PreemptionPredicates (:141-179)
 */
#ifndef YKPRED_H_
#define YKPRED_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 4: ykpred_set_spec_effects + ykpred_spec_effects_t, ykpred_comm_info (round 5 added them without a bump: a host built against the
 *    header could not tell an older library apart), ykpred_layout_t.sweep_rows / index_rows_walked / run_rows / fused_rows, ykpred_get_round_info
 * 3: ykpred_eval_args_t.bitmap_rows (a caller-owned bitmap states its size), ykpred_peek_row (the resident
 *    answer served to single Predicates() callbacks), ykpred_eval_nodes is collective on a sharded engine with topology signatures
 * 2: bitmap rows addressed through ykpred_layout_t.row_of_pod, ykpred_nodes_t.name_rank, per-ask unsupported flag, communicator /
 *    gather / exchange entry points (version 1 = the round-1 ABI: rows in ask order, no collectives) */
#define YKPRED_ABI_VERSION 4

/* status codes */
#define YKPRED_OK 0
#define YKPRED_E_INVALID (-1)   /* bad argument / inconsistent sizes */
#define YKPRED_E_DEVICE (-2)    /* HIP runtime error (message in ykpred_last_error) */
#define YKPRED_E_NOMEM (-3)     /* device or host allocation failed */
#define YKPRED_E_STATE (-4)     /* call sequence error (e.g. eval before nodes/specs/pods were uploaded) */
#define YKPRED_E_UNSUPPORTED (-5)

/* Plugin bits. Bit order = Filter order of predicate_manager.go:339-352. */
#define YKPRED_PLUGIN_NODE_UNSCHEDULABLE (1u << 0)
#define YKPRED_PLUGIN_NODE_NAME (1u << 1)
#define YKPRED_PLUGIN_TAINT_TOLERATION (1u << 2)
#define YKPRED_PLUGIN_NODE_AFFINITY (1u << 3)
#define YKPRED_PLUGIN_NODE_PORTS (1u << 4)
#define YKPRED_PLUGIN_NODE_RESOURCES_FIT (1u << 5)
#define YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD (1u << 6)
#define YKPRED_PLUGIN_INTER_POD_AFFINITY (1u << 7)
#define YKPRED_PLUGIN_ALL 0xffu

/* "failing plugin" codes returned by ykpred_query (0 = "" — a PreFilter plugin rejected the pod, :236-238) */
#define YKPRED_CODE_NONE 0
#define YKPRED_CODE_NODE_UNSCHEDULABLE 1
#define YKPRED_CODE_NODE_NAME 2
#define YKPRED_CODE_TAINT_TOLERATION 3
#define YKPRED_CODE_NODE_AFFINITY 4
#define YKPRED_CODE_NODE_PORTS 5
#define YKPRED_CODE_NODE_RESOURCES_FIT 6
#define YKPRED_CODE_POD_TOPOLOGY_SPREAD 7
#define YKPRED_CODE_INTER_POD_AFFINITY 8
#define YKPRED_CODE_UNSUPPORTED 255 /* the ask's spec carries YKPRED_SPEC_UNSUPPORTED: not evaluated, route it to the CPU manager */

/* reason bits returned by ykpred_query next to the plugin code (the host composes the status message) */
#define YKPRED_REASON_TOO_MANY_PODS (1u << 0)
#define YKPRED_REASON_PREFILTER_NODE_NOT_ELIGIBLE (1u << 1)
#define YKPRED_REASON_PREFILTER_REJECTED (1u << 2)
#define YKPRED_REASON_MISSING_TOPOLOGY_LABEL (1u << 3)
#define YKPRED_REASON_RESOURCE_SHIFT 8 /* bit (8+r) set: insufficient resource dimension r */

/* node flags */
#define YKPRED_NODE_UNSCHEDULABLE (1u << 0)

/* spec flags */
#define YKPRED_SPEC_TOLERATES_UNSCHEDULABLE (1u << 0) /* tolerates node.kubernetes.io/unschedulable:NoSchedule */
#define YKPRED_SPEC_AFFINITY_SKIP (1u << 1)           /* NodeAffinity.PreFilter returns Skip (no selector, no required affinity) */
#define YKPRED_SPEC_PREFILTER_REJECT (1u << 2)        /* NodeAffinity.PreFilter: conflicting metadata.name terms */
#define YKPRED_SPEC_PREFILTER_NAMES (1u << 3)         /* NodeAffinity.PreFilter returned a NodeNames set (pre_terms) */
#define YKPRED_SPEC_UNSUPPORTED (1u << 4)             /* the host could not encode this ask (volumes, DRA claims, dictionary limits ...):
                                                         its bitmap row is all zero, its count 0, its decision -1 and every query
                                                         answers YKPRED_CODE_UNSUPPORTED, whatever the plugin lists — the host
                                                         routes exactly these asks to the CPU predicate manager */

/* special values for pods.node_name_index */
#define YKPRED_NO_NODE_NAME (-1)      /* pod.Spec.NodeName == "" */
#define YKPRED_UNKNOWN_NODE_NAME (-2) /* names a node that is not in the table: matches no node */

typedef struct ykpred_engine ykpred_engine_t;

typedef struct ykpred_config {
  int32_t abi_version;      /* YKPRED_ABI_VERSION */
  int32_t device;           /* HIP device ordinal */
  int32_t num_resources;    /* R >= 3: 0 = cpu (milli), 1 = memory, 2 = ephemeral-storage, 3.. = scalar resources */
  int32_t taint_words;      /* KT >= 1: 64-bit words of the taint dictionary */
  int32_t label_words;      /* W  >= 1: 64-bit words of the node-selector requirement dictionary */
  int32_t topology_keys;    /* KD >= 0: topology keys used by hard spread constraints */
  int32_t selector_classes; /* KS >= 0: distinct (namespace, labelSelector) classes of spread constraints */
  int32_t port_words;       /* KP >= 0: 64-bit words of the host-port dictionary (NodePorts) */
  int32_t reserved[8];      /* [0..7]: engine tunables for experiments, 0 = defaults; [3] == 1 replays a repeated
                               ykpred_eval as a hipGraph (opt-in: measured equal to plain launches); [4] = number of distinct
                               request values per resource dimension from which the sorted-walk plane kernels are used (256);
                               [5] = average members per combine chunk below which the wave-per-chunk combine runs (16, -1 never);
                               [6] = band height of the zone-A row layout in windows (4..256, multiple of 4; 0 = chosen from the row
                               length so that classes of ~100 asks still get band rows; -1 = no band layout); [7] unused.
                               The tests force paths through the environment instead: YKPRED_TUNE="walk_rows=1,sig_wpl=2,..." */
} ykpred_config_t;

/* Node table, structure-of-arrays. Arrays documented [A][count] are A consecutive runs of `count` values. */
typedef struct ykpred_nodes {
  int32_t count;
  const int64_t* allocatable;    /* [R][count]   NodeInfo.Allocatable */
  const int64_t* requested;      /* [R][count]   NodeInfo.Requested (assumed pods included) */
  const int32_t* allowed_pods;   /* [count]      Allocatable.AllowedPodNumber */
  const int32_t* pod_count;      /* [count]      len(NodeInfo.Pods) */
  const uint32_t* flags;         /* [count]      YKPRED_NODE_* */
  const uint64_t* taint_bits;    /* [KT][count]  bit t: node carries dictionary taint t (NoSchedule/NoExecute only) */
  const uint64_t* label_bits;    /* [W][count]   bit q: node satisfies dictionary requirement q */
  const int32_t* domain_id;      /* [KD][count]  id of the node's value for topology key k, -1 = label missing */
  const int32_t* selector_count; /* [KS][count]  # pods on the node matching selector class s */
  const int32_t* domain_sizes;   /* [KD]         number of distinct values (domain ids 0..size-1) of topology key k */
  const uint64_t* port_bits;     /* [KP][count]  bit k: some pod on the node uses a host port that conflicts with dictionary
                                                 port k (HostPortInfo.CheckConflict: same protocol+port, equal or wildcard IP) */
  const int32_t* name_rank;      /* [count]      position of the node's NodeID in lexicographic order: the tie-break of the
                                                 bin-pack order between nodes of equal score (yunikorn-core sorts by score,
                                                 then node id). NULL = ties by node index. In a node-sharded cluster the
                                                 shards are ranges of a name-sorted node list, so that the cross-shard
                                                 tie-break (global node index) agrees with it. */
} ykpred_nodes_t;

/* One topology constraint of a pod spec: a hard (DoNotSchedule) topology spread constraint, or one InterPodAffinity
 * rule. All kinds share the mechanics "per-node match counts (selector_count) → histogram per topology domain → test of
 * the candidate node's domain"; a spec's PodTopologySpread constraints must precede its InterPodAffinity ones. */
typedef struct ykpred_spread {
  int32_t topology_key;   /* index into domain_id[KD] */
  int32_t selector_class; /* index into selector_count[KS]; -1 = counts nothing (nil / empty selector) */
  int32_t max_skew;       /* SPREAD only */
  int32_t min_domains;    /* SPREAD only; nil => 1 */
  int32_t self_match;     /* SPREAD: the pod's labels match the selector. POD_AFFINITY: the pod matches ALL its own
                             affinity terms (same value on every affinity term of the spec) */
  uint32_t flags;         /* SPREAD only. bit0: nodeAffinityPolicy == Honor, bit1: nodeTaintsPolicy == Honor */
  int32_t kind;           /* YKPRED_CONSTRAINT_* */
  int32_t reserved;
} ykpred_spread_t;
#define YKPRED_SPREAD_HONOR_AFFINITY (1u << 0)
#define YKPRED_SPREAD_HONOR_TAINTS (1u << 1)
#define YKPRED_CONSTRAINT_SPREAD 0
#define YKPRED_CONSTRAINT_POD_AFFINITY 1          /* selector_class counts pods matching ALL required affinity terms of the spec */
#define YKPRED_CONSTRAINT_POD_ANTI_AFFINITY 2     /* selector_class counts pods matching this required anti-affinity term */
#define YKPRED_CONSTRAINT_EXISTING_ANTI_AFFINITY 3 /* selector_class counts (pod, anti-affinity term on this key) pairs matching the spec's pod */

/* Pod specs (one per distinct pod template / task group; pods reference them by index). */
typedef struct ykpred_specs {
  int32_t count;
  const int64_t* requests;       /* [count][R]  upstream PodRequests: cpu milli, others Value() */
  const uint64_t* tolerated;     /* [count][KT] bit t: some toleration tolerates dictionary taint t */
  const uint32_t* flags;         /* [count]     YKPRED_SPEC_* */
  const int32_t* aff_term_off;   /* [count+1]   Filter DNF: spec s owns aff_terms rows [off[s], off[s+1]) */
  const uint64_t* aff_terms;     /* [n][W]      a term matches a node iff (label_bits & term) == term;
                                                no rows = matches no node; one all-zero row = matches every node */
  const int32_t* pre_term_off;   /* [count+1]   PreFilter NodeNames set as DNF (only if YKPRED_SPEC_PREFILTER_NAMES) */
  const uint64_t* pre_terms;     /* [m][W] */
  const int32_t* spread_off;     /* [count+1]   may be NULL when no spec has hard spread constraints */
  const ykpred_spread_t* spread; /* [k] */
  const uint64_t* wanted_ports;  /* [count][KP] bit k: the pod requests dictionary host port k (NodePorts); may be NULL if KP == 0 */
} ykpred_specs_t;

typedef struct ykpred_pods {
  int32_t count;
  const int32_t* spec_index;      /* [count] */
  const int32_t* node_name_index; /* [count] node index named by pod.Spec.NodeName, or YKPRED_NO_/UNKNOWN_NODE_NAME */
} ykpred_pods_t;

#define YKPRED_OUT_BITMAP (1u << 0)    /* P x N feasibility bitmap */
#define YKPRED_OUT_COUNTS (1u << 1)    /* per-pod number of feasible nodes */
#define YKPRED_OUT_DECISIONS (1u << 2) /* per-pod best feasible node under the bin-pack order (-1 = none) */
#define YKPRED_OUT_DECISION_KEYS (1u << 3) /* per-pod order key of that node (int64; INT64_MAX = none): lets shards of a
                                              node-sharded cluster pick the global best with one MIN all-reduce */
#define YKPRED_EVAL_PROFILE (1u << 8)  /* bracket every kernel with HIP events (see ykpred_last_timing) */
#define YKPRED_EVAL_DIRECT (1u << 9)   /* use the per-pair reference kernel instead of the plane/class path */
#define YKPRED_EVAL_SPREAD_COUNT_ONLY (1u << 10)   /* node-sharded clusters: only build this shard's PodTopologySpread
                                                      histograms (layout.spread_counts / spread_present) and return */
#define YKPRED_EVAL_SPREAD_COUNTS_READY (1u << 11) /* the histograms already hold the cluster-wide (all-reduced) values */
#define YKPRED_EVAL_SKIP_BITMAP (1u << 12)         /* refresh planes / counts scatter / decisions only; the bitmap and the class
                                                      counts are already current (used by ykpred_eval_nodes) */
#define YKPRED_EVAL_DIRTY_CLASSES (1u << 13)       /* internal (ykpred_eval_nodes): rewrite only the rows of classes whose topology
                                                      signature changed (flagged on the device), keep every other row */

typedef struct ykpred_eval_args {
  uint32_t prefilter_plugins; /* enabled PreFilter plugins (YKPRED_PLUGIN_* bits) */
  uint32_t filter_plugins;    /* enabled Filter plugins */
  uint32_t options;           /* YKPRED_OUT_* | YKPRED_EVAL_* */
  uint32_t bitmap_rows;       /* caller-owned bitmap: the physical rows it holds (row_stride words each). The engine never writes
                                 a row >= bitmap_rows: an evaluation or ask-table patch that would need one fails with
                                 YKPRED_E_INVALID / invalidates the evaluation instead. 0 = ykpred_set_row_capacity rows (one
                                 of the two must be given with a caller-owned bitmap) */
  void* bitmap;               /* optional caller-owned DEVICE buffer of bitmap_rows x row_stride x 8 bytes; NULL = engine-owned */
  void* stream;               /* hipStream_t to launch on; NULL = the engine's own stream */
  void* counts;               /* optional caller-owned DEVICE int32[P]; NULL = engine-owned */
  void* decisions;            /* optional caller-owned DEVICE int32[P]; NULL = engine-owned */
  void* decision_keys;        /* optional caller-owned DEVICE int64[P]; NULL = engine-owned */
} ykpred_eval_args_t;

typedef struct ykpred_layout {
  int32_t num_nodes;  /* N */
  int32_t num_pods;   /* P */
  int32_t num_specs;
  int32_t num_classes;
  int32_t row_words;   /* ceil(N/64): meaningful 64-bit words per bitmap row */
  int32_t row_stride;  /* words between consecutive rows (>= row_words, multiple of 16 = 128 B; padding words are 0) */
  int32_t num_chunks;
  int32_t plane_rows;  /* total signature planes evaluated per eval */
  uint64_t bitmap_bytes;
  void* bitmap;        /* device pointer of the last evaluated bitmap; rows are addressed through row_of_pod (below) */
  void* counts;        /* device int32[P] */
  void* decisions;     /* device int32[P] */
  void* decision_keys; /* device int64[P] */
  void* spread_counts;  /* device int32[spread_cells]: matching pods per (spread constraint, topology domain) — SUM across shards */
  void* spread_present; /* device int32[spread_cells]: 1 = an eligible node carries the domain — MAX across shards */
  int64_t spread_cells;
  int32_t num_rows;     /* physical rows of the bitmap (>= num_pods: rows are laid out for the writer and never reused between two
                           class builds); bitmap_bytes = num_rows * row_stride * 8 — what a caller-owned bitmap must hold */
  int32_t band_rows;    /* rows [0, band_rows) are written by the band writer (k_expand_bands), the rest class by class (k_combine) */
  void* row_of_pod;     /* device int32[P]: the bitmap row of pod p. Bit (n & 63) of word [row_of_pod[p] * row_stride + (n >> 6)]
                           says whether pod p fits node n */
  int32_t index_rows;   /* of plane_rows: request-value rows of many-valued resource dimensions, kept as INDEX rows (one byte per
                           64-node word instead of an 8-byte plane word) */
  int32_t band_steps;   /* band height (windows) the current row layout was built with */
  int32_t sweep_rows;   /* rows of zone B laid out as RUNS of one signature in ascending order of a many-valued request dimension:
                           a full pass writes them with k_sweep_rows (a row = its predecessor minus the nodes the larger value loses) */
  int32_t index_rows_walked; /* of index_rows: the ones a full pass still materialises (k_dim_walk) — the sweep runs read none */
  int32_t run_rows;     /* rows of zone B whose classes (no index row, staged request-value rows) are written run by run: k_class_runs */
  int32_t fused_rows;   /* rows of zone B whose classes no run kernel takes and whose rows are plain plane rows: written from records resolved
                           at class-build time (k_fused_rows) */
} ykpred_layout_t;

#define YKPRED_MAX_TIMED_KERNELS 24
typedef struct ykpred_timing {
  int32_t num_kernels;
  float total_ms;                         /* first launch to last completion, device time */
  float kernel_ms[YKPRED_MAX_TIMED_KERNELS];
  const char* kernel_name[YKPRED_MAX_TIMED_KERNELS];
} ykpred_timing_t;

/* lifecycle */
int32_t ykpred_create(const ykpred_config_t* cfg, ykpred_engine_t** out);
void ykpred_destroy(ykpred_engine_t* e);
const char* ykpred_last_error(const ykpred_engine_t* e); /* e may be NULL: error of the last failed create() */
int32_t ykpred_abi_version(void);

/* state upload (externally serialised against evals) */
int32_t ykpred_set_nodes(ykpred_engine_t* e, const ykpred_nodes_t* nodes);
int32_t ykpred_update_node(ykpred_engine_t* e, int32_t index, const ykpred_nodes_t* one_node); /* count must be 1 */
/* Dictionary growth without a re-upload: replaces ONE 64-bit word column of label_bits ([count] values, one per node) — how a
 * requirement that no spec used before gets its bit. Bits that no uploaded spec references may change freely: the last
 * evaluation stays valid. */
int32_t ykpred_update_label_word(ykpred_engine_t* e, int32_t word, const uint64_t* column /* [N] */);
/* A table whose first specs are byte-identical to the previous one (new specs appended) keeps the engine's pod classes and
 * the last evaluation valid, so that a new pod template is followed by ykpred_update_pods / ykpred_eval_pods, not by a full
 * pass (unless the new specs bring a new topology-constraint signature). */
int32_t ykpred_set_specs(ykpred_engine_t* e, const ykpred_specs_t* specs);
int32_t ykpred_set_pods(ykpred_engine_t* e, const ykpred_pods_t* pods);

/* evaluation of every pending pod against every node of the uploaded snapshot */
int32_t ykpred_eval(ykpred_engine_t* e, const ykpred_eval_args_t* args);
/* Incremental form for the sequential scheduling loop (AssumePod / ForgetPod / UpdateNode between two asks,
 * /root/reference/pkg/cache/context.go:828-898): after ykpred_update_node on `num_nodes` nodes, re-evaluates ONLY those
 * node columns of the bitmap produced by the last ykpred_eval (same plugin lists, same output buffers) and patches the
 * feasible counts; decisions are recomputed when YKPRED_OUT_DECISIONS is set. With PodTopologySpread / InterPodAffinity
 * signatures active the histograms are rebuilt (they couple all nodes), the signatures whose PreFilter state moved are
 * found on the device, and the classes that use them get their WHOLE rows rewritten — all other classes still only the
 * touched columns. YKPRED_E_STATE if there is no matching previous evaluation. On a node-sharded engine (communicator attached)
 * with topology signatures the call is COLLECTIVE: every shard enters it — with its own, possibly empty, node list — and
 * performs exactly the histogram exchange of a full ykpred_eval (SUM of matches, MAX of "domain present"), so a shard may
 * answer the same step with ykpred_eval instead. Hosts that move the histograms themselves split the call like ykpred_eval:
 * YKPRED_EVAL_SPREAD_COUNT_ONLY (rebuild this shard's histograms, return), then YKPRED_EVAL_SPREAD_COUNTS_READY. */
int32_t ykpred_eval_nodes(ykpred_engine_t* e, const ykpred_eval_args_t* args, int32_t num_nodes, const int32_t* node_index);
/* Row-level maintenance of the ask table (SchedulerCache.UpdatePod for a new / changed / finished ask,
 * /root/reference/pkg/cache/external/scheduler_cache.go:303-388) without re-uploading it: the table gets `num_pods_after`
 * rows (rows beyond it are dropped, every appended row must be listed) and the `count` listed rows (each at most once) take
 * new (spec, nodeName) values. The engine moves those rows between pod classes in O(count); their bitmap rows are stale
 * until ykpred_eval_pods (or a full ykpred_eval) has run. The specs must already be uploaded. */
int32_t ykpred_update_pods(ykpred_engine_t* e, int32_t num_pods_after, int32_t count, const int32_t* rows, const int32_t* spec_index,
                           const int32_t* node_name_index);
/* Re-evaluates ONLY the listed bitmap rows (per pair, against every node) into the bitmap of the last ykpred_eval (same
 * plugin lists, same output buffers; engine-owned outputs grow with the table, caller-owned ones must already hold
 * layout.num_pods rows), with their feasible counts and — YKPRED_OUT_DECISIONS / _KEYS — their decisions.
 * YKPRED_E_STATE if there is no matching previous evaluation, or decisions are requested while the bin-pack order of
 * the last evaluation is stale (a node changed since). */
int32_t ykpred_eval_pods(ykpred_engine_t* e, const ykpred_eval_args_t* args, int32_t num_rows, const int32_t* rows);
int32_t ykpred_synchronize(ykpred_engine_t* e);
int32_t ykpred_get_layout(const ykpred_engine_t* e, ykpred_layout_t* out);
int32_t ykpred_last_timing(const ykpred_engine_t* e, ykpred_timing_t* out);
/* Engine counters since create: out[0] full evaluations, [1] ykpred_eval_nodes calls, [2] ykpred_eval_pods calls, [3] query calls,
 * [4] bitmap gathers, [5] table uploads / patches. Every upload / eval / gather entry point is also bracketed by a roctx
 * range ("ykpred:eval", "ykpred:upload_nodes", "ykpred:gather_bitmap" ...) when the process runs under a profiler that has
 * the marker library mapped, or with YKPRED_ROCTX=1. */
int32_t ykpred_get_counters(const ykpred_engine_t* e, int64_t* out6);

/* readback (device -> caller host buffers); each implies a synchronize */
int32_t ykpred_read_bitmap(ykpred_engine_t* e, int32_t first_pod, int32_t num_pods, uint64_t* out /* [num_pods][row_words] */);
int32_t ykpred_read_counts(ykpred_engine_t* e, int32_t* out /* [P] */);
int32_t ykpred_read_decisions(ykpred_engine_t* e, int32_t* out /* [P] */);
int32_t ykpred_read_scores(ykpred_engine_t* e, double* out /* [N] bin-pack score per node */);
int32_t ykpred_bitmap_checksum(ykpred_engine_t* e, uint64_t* out); /* order-independent hash of (pod,word,value) */
/* Parity support for bitmaps too large to read back (50k x 1M = 6.27 GB): the rows of the listed pods, densely; the pod →
 * class map with one representative pod per class (-1 = class without live member); and the number of bitmap words that
 * differ between a pod's row and the row of its class's representative, plus non-zero padding words. With these a checker
 * evaluates ONE pod per class against every node on the CPU and still covers every (pod, node) pair of the bitmap. */
int32_t ykpred_read_rows(ykpred_engine_t* e, int32_t n, const int32_t* pod_index, uint64_t* out /* [n][row_words] */);
int32_t ykpred_read_row_map(ykpred_engine_t* e, int32_t* out /* [P]: layout.row_of_pod on the host */);
int32_t ykpred_read_pod_classes(ykpred_engine_t* e, int32_t* pod_class /* [P] or NULL */, int32_t* class_rep /* [num_classes] or NULL */);
int32_t ykpred_check_class_rows(ykpred_engine_t* e, uint64_t* bad_words);

/* CONFLICT-RESOLVED decisions: the sequential loop yunikorn-core drives through the shim. For asks[0], asks[1], ... in this
 * order: the first node of the bin-pack order that passes Predicates() under the state the EARLIER asks of the round left
 * behind (AsyncRMCallback.Predicates, /root/reference/pkg/cache/scheduler_callback.go:203-205 → Context.IsPodFitNode,
 * context.go:696-716), then AssumePod (context.go:828-885 → SchedulerCache.AssumePod, scheduler_cache.go:443-461 →
 * NodeInfo.AddPod): the node's Requested grows by the ask's request vector, len(Pods) by one, its bin-pack score moves.
 * out_nodes[i] = node index or -1 (no node fits; nothing is assumed for that ask). The engine's TABLES ARE NOT CHANGED — the
 * round runs on a scratch copy of the node columns; the caller applies the allocations the core accepts through the cache
 * hooks (ykhost_assume_pod → ykpred_update_node) as it does for any AssumePod.
 * Needs a current evaluation WITH decisions of the same plugin lists (YKPRED_E_STATE otherwise). YKPRED_E_UNSUPPORTED — decide ask by
 * ask instead — when something other than node resources couples the asks of the round (active PodTopologySpread /
 * InterPodAffinity signatures, an ask that requests a host port) while the specs' effects (ykpred_set_spec_effects, below) are not
 * uploaded for the current spec table.
 * ONE GPU, two forms with identical decisions. SEQUENTIAL: one workgroup runs the loop (k_allocate_round; the asks are a dependency
 * chain, the parallelism is inside an ask) — a run of asks of one spec that lands on one node is decided by arithmetic (millions of
 * asks per second), every other ask costs ~10 us. BATCHED: the asks of a batch are proposed in parallel against one frozen state
 * (k_round_propose: a workgroup per ask, its 8 best feasible nodes with the columns their keys are made of), every ask is evaluated
 * against every node the batch proposed (k_round_cross: a bit per pair), the host replays the loop over the batch exactly — an
 * accepted node by its bit + NodeResourcesFit on the exchanged columns, a candidate that is full by the next entry of the ask's
 * list — and the accepted pods are assumed node by node: 100 k -> 570 k asks/s on a 20 000-ask round of configs[2]. Chosen per round
 * (YKPRED_TUNE round_batched: -1 = by the list, the default; 0 / 1 = never / always): batched for rounds of 512 asks and more
 * without live topology constraints whose mean run of one spec is shorter than four asks. ykpred_get_round_info counts both.
 * NODE-SHARDED engines (communicator attached, world > 1): the call is COLLECTIVE — every rank passes the same asks in the same
 * order — and out_nodes holds GLOBAL node indices (the winner's shard offset + its index there), identical on every rank. The
 * round always runs in batches: proposals and pair bits of a batch in ONE all-gather, the same replay on every rank, the owners
 * assume (engine.hip has the rule and its proof sketch). Equal keys across shards are ordered by global node index, as in
 * ykpred_exchange_decisions. Before the first batch the ranks agree on status, ask count and a hash of the ask list (a rank that
 * cannot run the round makes every rank return the same error). With active topology signatures the histograms are cluster-wide
 * state on every shard: the owner of an accepted node records what its assume added (constraint, domain, count), a second
 * all-gather of the batch hands the records to the other shards, and a batch ends in front of the first ask whose topology signature
 * counts a selector class an accepted pod of the batch added to (its verdicts may have turned from fail to fit anywhere).
 * YKPRED_E_UNSUPPORTED when one pod moves more than 10 histogram cells (every rank sees the same record and stops). */
int32_t ykpred_allocate_round(ykpred_engine_t* e, uint32_t prefilter_plugins, uint32_t filter_plugins, int32_t n_asks,
                              const int32_t* asks /* host, [n_asks] ask indices in decision order */, int32_t* out_nodes /* host, [n_asks] */);
/* How the rounds so far were decided: out[0] rounds run in batches, [1] their asks, [2] their batches, [3] collective exchanges of
 * sharded rounds (proposals + histogram deltas), [4] rounds run by the sequential kernel, [5] their asks. */
int32_t ykpred_get_round_info(const ykpred_engine_t* e, int64_t* out6);
/* What NodeInfo.AddPod (behind SchedulerCache.AssumePod, /root/reference/pkg/cache/external/scheduler_cache.go:443-461) adds to a
 * node BESIDES the pod's request vector and len(Pods) += 1, per pod spec — the part of the upstream NodeInfo the Filters of
 * predicate_manager.go:339-352 read again for the NEXT ask: UsedPorts (NodePorts), and the pod itself in Pods / PodsWithAffinity /
 * PodsWithRequiredAntiAffinity, which this ABI encodes as the per-node match counts selector_count[KS][N] (PodTopologySpread's
 * countPodsMatchSelector, InterPodAffinity's topologyToMatchedTermCount). A pod's contribution to a count column is a function of
 * (column, pod template) alone, so the host states it once per spec:
 *   contrib_*      spec s adds contrib_count[k] to column contrib_class[k] of the node it lands on, k in [contrib_off[s], contrib_off[s+1])
 *   occupied_ports [count][KP] bit k: once a pod of the spec is on a node, the node's port_bits gain bit k (HostPortInfo.CheckConflict of
 *                  the pod's host ports against dictionary port k)
 * With the effects of the CURRENT spec table uploaded (ykpred_set_specs drops them), ykpred_allocate_round keeps the topology
 * histograms and the port words of its scratch state current ask by ask on the device, and active PodTopologySpread /
 * InterPodAffinity signatures or host-port asks no longer make it return YKPRED_E_UNSUPPORTED. The shards' dictionaries must
 * list the selector classes of every template that can be ASSUMED during a round, not only of the pods already on nodes (the
 * host encoder does: encoder.h, existing_anti_templates_). */
typedef struct ykpred_spec_effects {
  int32_t count;                  /* == the uploaded spec table's count */
  const int32_t* contrib_off;     /* [count+1]; NULL = no spec adds to any count column */
  const int32_t* contrib_class;   /* [contrib_off[count]] column of selector_count, 0 <= class < KS */
  const int32_t* contrib_count;   /* [contrib_off[count]] > 0 */
  const uint64_t* occupied_ports; /* [count][KP]; NULL = no spec occupies a dictionary host port (or KP == 0) */
} ykpred_spec_effects_t;
int32_t ykpred_set_spec_effects(ykpred_engine_t* e, const ykpred_spec_effects_t* fx /* NULL: drop the uploaded effects */);

/* Debugging aid (no reference counterpart). With YKPRED_GUARD_PAGES=1 (2) in the environment every device block of the engine
 * ends (starts) on the last (first) byte of its mapping with an unmapped granule behind (in front of) it. The self-test proves the
 * guard is armed: it allocates a block the same way and reads `offset` bytes past its end (offset >= 0) or |offset| bytes in front
 * of its start (offset < 0). Under the guard an offset outside the block kills the process with a GPU memory access fault — that
 * IS the expected outcome (tests/test_gpu_guard.py runs it in a child process); offsets inside the block return YKPRED_OK.
 * Without the guard the call returns YKPRED_E_STATE and touches nothing. */
int32_t ykpred_guard_selftest(ykpred_engine_t* e, int32_t offset);

/* The RESIDENT answer served to single callbacks. yunikorn-core walks asks, not nodes: for one ask it calls Predicates() node
 * after node (Context.IsPodFitNode, /root/reference/pkg/cache/context.go:696-716, behind scheduler_callback.go:203-205). When
 * the last evaluation is still current, every fit bit of that ask already sits in its bitmap row: ykpred_peek_row copies that
 * ONE row (row_words words; plus the ask's feasible count and bin-pack decision when the pointers are given) to the host on a
 * copy stream that waits for the evaluation's completion event — no kernel, no device-wide synchronize. The host then serves
 * the ask's callbacks from memory and needs ykpred_query only to name the failing plugin of a pair that does not fit.
 * YKPRED_E_STATE when the bitmap does not describe the current tables and these plugin lists (a node or ask changed and was not
 * re-evaluated): the caller falls back to ykpred_query_pod. ykpred_read_order: the bin-pack order of the last evaluation that
 * produced decisions (perm[i] = node at position i) — with a row, "the first k feasible nodes in bin-pack order". */
int32_t ykpred_peek_row(ykpred_engine_t* e, int32_t pod_index, uint32_t prefilter_plugins, uint32_t filter_plugins,
                        uint64_t* out_row /* [row_words] */, int32_t* out_count /* may be NULL */, int32_t* out_decision /* may be NULL */);
int32_t ykpred_read_order(ykpred_engine_t* e, int32_t* out_perm /* [N] */);
/* Building blocks of a host-side mirror of the resident answer (libykhost keeps one): YKPRED_OK iff the bitmap of the last
 * evaluation is the answer for the current tables and these plugin lists (*num_classes = classes of that evaluation); the class
 * of an ask in the engine's current class index (host-side lookup: asks that join an existing class after the evaluation are
 * answered by that class's row); the whole answer in class-compressed form, [num_classes][row_words] on the host. */
int32_t ykpred_answer_state(ykpred_engine_t* e, uint32_t prefilter_plugins, uint32_t filter_plugins, int32_t* num_classes /* may be NULL */);
int32_t ykpred_pod_class(ykpred_engine_t* e, int32_t pod_index, int32_t* out_class);
int32_t ykpred_read_class_rows(ykpred_engine_t* e, uint64_t* out /* [num_classes][row_words] */);

/* one Predicates() answer per (pod,node) pair, evaluated on the device straight from the tables */
int32_t ykpred_query(ykpred_engine_t* e, int32_t num_pairs, const int32_t* pod_index, const int32_t* node_index,
                     uint32_t prefilter_plugins, uint32_t filter_plugins, uint8_t* fit /* [num_pairs] */,
                     uint8_t* plugin_code /* [num_pairs], may be NULL */, uint32_t* reason /* [num_pairs], may be NULL */);

/* All Predicates() answers of ONE pod in one call: out arrays have one entry per node. The core tries an ask on many nodes
 * in a row (one callback per node); the Go side fetches the ask's answers once and serves those callbacks from host memory. */
int32_t ykpred_query_pod(ykpred_engine_t* e, int32_t pod_index, uint32_t prefilter_plugins, uint32_t filter_plugins,
                         uint8_t* fit /* [N] */, uint8_t* plugin_code /* [N], may be NULL */, uint32_t* reason /* [N], may be NULL */);

/* ykpred_query_pod with one 4-byte word per node and a single device → host copy: bits 0-7 plugin code, bit 8 fit, bits 9-12
 * reason bits 0-3, bits 13.. the insufficient-resource bits (reason >> YKPRED_REASON_RESOURCE_SHIFT). */
int32_t ykpred_query_pod_packed(ykpred_engine_t* e, int32_t pod_index, uint32_t prefilter_plugins, uint32_t filter_plugins, uint32_t* out /* [N] */);

/* PreemptionPredicates (predicate_manager.go:141-179): victims are described by their request vectors, in order. */
int32_t ykpred_preemption(ykpred_engine_t* e, int32_t pod_index, int32_t node_index, int32_t num_victims,
                          const int64_t* victim_requests /* [num_victims][R]; a nil victim is an all-zero row with present=0 */,
                          const uint8_t* victim_present /* [num_victims] */, int32_t start_index,
                          uint32_t prefilter_plugins, uint32_t filter_plugins, int32_t* out_index);
/* Same, for nodes whose victims hold host ports: port_bits_after[i][KP] = the node's port_bits once victims 0..i are gone. */
int32_t ykpred_preemption_ports(ykpred_engine_t* e, int32_t pod_index, int32_t node_index, int32_t num_victims,
                                const int64_t* victim_requests, const uint8_t* victim_present, const uint64_t* port_bits_after,
                                int32_t start_index, uint32_t prefilter_plugins, uint32_t filter_plugins, int32_t* out_index);

/* Batched PreemptionPredicates: query q uses victims [victim_off[q], victim_off[q+1]) of the flattened victim arrays
 * (same per-victim meaning as above) and writes out_index[q]. One launch, one thread per query. */
int32_t ykpred_preemption_batch(ykpred_engine_t* e, int32_t num_queries, const int32_t* pod_index, const int32_t* node_index,
                                const int32_t* victim_off /* [num_queries+1] */, const int64_t* victim_requests /* [total][R] */,
                                const uint8_t* victim_present /* [total] */, const uint64_t* port_bits_after /* [total][KP] or NULL */,
                                const int32_t* start_index /* [num_queries] */, uint32_t prefilter_plugins, uint32_t filter_plugins,
                                int32_t* out_index /* [num_queries] */);

/* ------------------------------------------------------------------------------------------------------------------
 * Multi-GPU: node-axis shards. One engine per GPU / process holds a contiguous shard of the node table and
 * the full ask table; every (ask, node) pair is independent, so each shard evaluates its own bitmap columns with no
 * data-path collective. The exchanges below run over RCCL (xGMI inside one box) on the stream they are given, ordered after
 * the work already queued there; librccl is loaded on first use (dlopen), so a single-GPU process never maps it.
 *
 *   ykpred_comm_unique_id   rank 0 draws the 128-byte id; the host distributes it out of band (the Go side: over whatever
 *                           already connects the shard processes; bench.py: its torch.distributed bootstrap group)
 *   ykpred_comm_init        ncclCommInitRank; node_offset = global index of this shard's node 0
 *   ykpred_gather_bitmap    all-gather of the shard bitmaps of the last ykpred_eval into the shard-major layout
 *                           [world][num_rows][row_stride] (BASELINE configs[3]) plus the shards' row_of_pod maps
 *                           [world][P]; every shard must use the same row_stride and row capacity
 *                           (ykpred_set_row_stride / ykpred_set_row_capacity) and hold the same asks in the same order
 *   ykpred_exchange_decisions  in place on the outputs of the last ykpred_eval (which must have produced decision keys):
 *                           counts → cluster-wide feasible counts (SUM); decisions → GLOBAL node index of the best
 *                           feasible node in bin-pack order, ties by global node index (MIN key, then MIN index), -1 = none
 *   PodTopologySpread / InterPodAffinity histograms: once a communicator is attached, ykpred_eval itself sums the
 *                           per-shard histograms (SUM of matches, MAX of "domain present") between its count and min
 *                           passes — the shards must share the topology-domain dictionaries — and ykpred_query /
 *                           ykpred_preemption reuse those cluster-wide histograms instead of rebuilding shard-local ones.
 * Not sharded: PreemptionPredicates (one node, sequential victim prefix) runs on the shard that owns the node. */
#define YKPRED_COMM_ID_BYTES 128
/* Which shared library provides the collectives (ncclGetUniqueId / ncclCommInitRank / ncclCommDestroy / ncclAllGather /
 * ncclAllReduce / ncclGetErrorString): by default librccl, loaded on the first ykpred_comm_* call. Effective only BEFORE that
 * first call of the process. The tests point it at tests/c/rccl_stub.cpp — the same entry points over shared memory between
 * processes that share one GPU, which RCCL itself refuses — so that the world > 1 branches run on a one-GPU box. */
int32_t ykpred_comm_use_library(const char* path);
int32_t ykpred_comm_unique_id(uint8_t* id /* [YKPRED_COMM_ID_BYTES] */);
int32_t ykpred_comm_init(ykpred_engine_t* e, const uint8_t* id, int32_t rank, int32_t world, int32_t node_offset);
int32_t ykpred_comm_destroy(ykpred_engine_t* e);
/* the attached communicator's geometry (rank 0, world 1, node_offset 0 without one); any of the pointers may be NULL */
int32_t ykpred_comm_info(const ykpred_engine_t* e, int32_t* rank, int32_t* world, int32_t* node_offset);
/* Rows of the NEXT ykpred_set_nodes get this stride (64-bit words, multiple of 16, >= the shard's own need); 0 = automatic.
 * Shards of unequal size agree on the stride of the largest one so that the gathered layout is regular. */
int32_t ykpred_set_row_stride(ykpred_engine_t* e, int32_t words);
/* The bitmap is sized for at least `rows` physical rows (0 = as many as needed). Shards agree on one capacity (>= every
 * shard's layout.num_rows) so that the gathered layout [world][rows][row_stride] is regular. */
int32_t ykpred_set_row_capacity(ykpred_engine_t* e, int32_t rows);
int32_t ykpred_gather_bitmap(ykpred_engine_t* e, void* gathered /* DEVICE [world][num_rows][row_stride] u64, NULL = engine-owned */, void* stream);
int32_t ykpred_exchange_decisions(ykpred_engine_t* e, void* stream);
/* Class-compressed form of the same gather. The member rows of a pod class are identical, so a shard's bitmap IS its
 * [num_classes][row_stride] class-row table plus its pod -> class map. xGMI is per-link bound (7 links x ~50 GB/s in a ring)
 * while the local HBM takes > 5 TB/s of writes: the shards exchange class rows (MBs instead of GBs) and every GPU writes all
 * `world` slabs locally. Result: [world][num_rows][row_stride] like ykpred_gather_bitmap, with EVERY slab in the row order
 * of the receiving engine (its own row_of_pod; ykpred_read_gathered hides the difference).
 * Shards merge signatures relative to their own taint / node-name dictionaries, so their class partitions may differ:
 *   ykpred_layout_hash            64-bit digest of the class partition + row layout; a peer with MY digest is expanded with my
 *                                 writer kernels and tables (band writer + class-by-class writer), any other peer ask by ask
 *                                 row by row through its pod -> class map (k_expand_by_row)
 *   ykpred_collect_class_rows     out = DEVICE [num_classes][row_stride] u64 of the last evaluation
 *   ykpred_expand_class_rows      bitmap_out (DEVICE [num_rows][row_stride]) = the bitmap whose class rows are `class_rows`
 *                                 (indexed by this engine's classes, or by a peer's with that peer's pod -> class map), in this
 *                                 engine's row layout
 *   ykpred_gather_bitmap_compressed  header exchange (digest, class count), ncclAllGather of the class rows (and of the pod ->
 *                                 class maps when digests differ), `world` expansions; the slab of this rank is skipped when
 *                                 `gathered` + rank * slab is the bitmap the last evaluation wrote */
int32_t ykpred_layout_hash(ykpred_engine_t* e, uint64_t* out);
int32_t ykpred_collect_class_rows(ykpred_engine_t* e, void* out, void* stream);
int32_t ykpred_expand_class_rows(ykpred_engine_t* e, const void* class_rows, const int32_t* pod_class /* DEVICE int32[P]: the pod -> class map
    the table is indexed by (a peer's); NULL = this engine's own classes */, void* bitmap_out, void* stream);
int32_t ykpred_gather_bitmap_compressed(ykpred_engine_t* e, void* gathered /* DEVICE [world][num_rows][row_stride] u64, NULL = engine-owned */, void* stream);
/* readback of the engine-owned gathered bitmap: the rows of pods [first_pod, first_pod + num_pods) in shard `shard` (through
 * that shard's gathered row_of_pod map), row_stride words each */
int32_t ykpred_read_gathered(ykpred_engine_t* e, int32_t shard, int32_t first_pod, int32_t num_pods, uint64_t* out);

#ifdef __cplusplus
}
#endif
#endif /* YKPRED_H_ */
