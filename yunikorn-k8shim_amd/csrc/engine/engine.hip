// engine.hip — C-ABI implementation of libykpred.so (include/ykpred.h) for MI355X / gfx950.
//
// Host side of the engine: owns the device tables (structure-of-arrays), groups pod specs into per-plugin
// signatures and pods into classes (pure bookkeeping on the upload path, O(1) per pod), and launches the
// kernels of kernels.hip.h. There is no CPU evaluation path in this file: every verdict comes from a kernel.
#include "../../../include/ykpred.h"

#include <hip/hip_runtime.h>
#include <dlfcn.h>
#include <rccl/rccl.h>  // types only: the library is loaded on first use of a ykpred_comm_* entry point

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <string>
#include <unordered_map>
#include <vector>

#include "kernels.hip.h"

using ykk::i64;
using ykk::u64;

namespace {

thread_local std::string g_create_error;

// Debugging aid, OFF unless YKPRED_GUARD_PAGES is set: every device buffer is mapped through the virtual-memory API with an unmapped
// granule next to it, so that an out-of-bounds access faults at the instruction that makes it instead of touching whatever was
// allocated next. YKPRED_GUARD_PAGES=1: the buffer ENDS on the last byte of its mapping (over-reads / over-writes fault; the size is
// NOT rounded up — the base is then only as aligned as the size is, which every kernel of this file tolerates: global memory takes
// unaligned vector accesses); =2: the buffer STARTS on the first byte of its mapping with the unmapped granule in front (accesses
// below the buffer fault); =3: POISON — plain blocks of exactly the size asked for, filled with 0xA5 (a read of memory nobody wrote
// shows up as a parity failure instead of passing on zeros). Costs one allocation granule per buffer: tests and small clusters only
// (tests/test_gpu_guard.py).
struct GuardAlloc {
  void* va = nullptr;      // start of the reserved address range
  void* map_at = nullptr;  // start of the mapped part
  size_t mapped = 0, reserved = 0;
  hipMemGenericAllocationHandle_t handle{};
};
inline int guard_mode() {
  static const int mode = [] {
    const char* v = getenv("YKPRED_GUARD_PAGES");
    return v ? atoi(v) : 0;
  }();
  return mode;
}
inline bool guard_pages_on() { return guard_mode() != 0; }
// YKPRED_TRACE_KERNELS=1 (debugging, with YKPRED_GUARD_PAGES): every stage of an evaluation is waited for and named on stderr —
// after a device fault the last line names the last stage that completed
inline bool trace_kernels_on() {
  static const bool on = [] {
    const char* v = getenv("YKPRED_TRACE_KERNELS");
    return v && atoi(v) != 0;
  }();
  return on;
}
inline hipError_t guard_alloc(size_t bytes, GuardAlloc* g, void** out) {
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  hipMemAllocationProp prop{};
  prop.type = hipMemAllocationTypePinned;
  prop.location.type = hipMemLocationTypeDevice;
  prop.location.id = dev;
  size_t gran = 0;
  if ((e = hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityMinimum)) != hipSuccess) return e;
  if (gran == 0) gran = 2u << 20;
  const bool front = guard_mode() == 2;
  g->mapped = (bytes + gran - 1) / gran * gran;
  g->reserved = g->mapped + gran;  // one granule of address space stays unmapped behind (mode 1) or in front of (mode 2) the buffer
  if ((e = hipMemAddressReserve(&g->va, g->reserved, gran, nullptr, 0)) != hipSuccess) return e;
  g->map_at = front ? (void*)((char*)g->va + gran) : g->va;
  if ((e = hipMemCreate(&g->handle, g->mapped, &prop, 0)) != hipSuccess) {
    (void)hipMemAddressFree(g->va, g->reserved);
    return e;
  }
  if ((e = hipMemMap(g->map_at, g->mapped, 0, g->handle, 0)) != hipSuccess) {
    (void)hipMemRelease(g->handle);
    (void)hipMemAddressFree(g->va, g->reserved);
    return e;
  }
  hipMemAccessDesc acc{};
  acc.location = prop.location;
  acc.flags = hipMemAccessFlagsProtReadWrite;
  if ((e = hipMemSetAccess(g->map_at, g->mapped, &acc, 1)) != hipSuccess) {
    (void)hipMemUnmap(g->map_at, g->mapped);
    (void)hipMemRelease(g->handle);
    (void)hipMemAddressFree(g->va, g->reserved);
    return e;
  }
  *out = front ? g->map_at : (void*)((char*)g->map_at + (g->mapped - bytes));
  // (with the stage trace: every block's address range — the address of a fault names the block that was overrun)
  if (trace_kernels_on()) fprintf(stderr, "ykpred: guard block %zu bytes [%p, %p)\n", bytes, *out, (void*)((char*)*out + bytes));
  return hipSuccess;
}
inline void guard_free(GuardAlloc* g) {
  if (!g->va) return;
  (void)hipDeviceSynchronize();
  (void)hipMemUnmap(g->map_at, g->mapped);
  (void)hipMemRelease(g->handle);
  // The address range is NEVER handed back (no hipMemAddressFree): on this stack (ROCm 7.2, gfx950) a range that is freed and
  // reserved again for a later block is served through stale translations — wrong data, or faults at addresses no block ever
  // had. That, not an engine defect, was the "deterministic fault under the guard" of round 3 (profiles/r04_guard_sessions.txt:
  // the same sweep with ranges reused: 3 of 4 clusters fail; never reused: 0 failures, no fault, in both modes). A freed block's
  // range stays reserved and unmapped, so a use-after-free faults as well.
  *g = GuardAlloc{};
}

struct DevBuf {
  void* p = nullptr;
  size_t cap = 0;
  GuardAlloc guard;  // only with YKPRED_GUARD_PAGES=1
  hipError_t raw_alloc(size_t bytes, void** out, GuardAlloc* g) {
    if (guard_mode() == 3) {  // POISON: a plain block of exactly the size asked for, filled with a pattern no table holds
      hipError_t e = hipMalloc(out, bytes);
      if (e == hipSuccess) e = hipMemset(*out, 0xA5, bytes);
      // (the fill runs on the null stream and the engine's streams are non-blocking: without this wait it can land AFTER the first
      // upload into the block — seen once as a spurious mismatch of this mode itself)
      if (e == hipSuccess) e = hipDeviceSynchronize();
      return e;
    }
    if (guard_pages_on()) return guard_alloc(bytes, g, out);
    return hipMalloc(out, bytes);
  }
  void raw_free(void* q, GuardAlloc* g) {
    if (g->va)
      guard_free(g);
    else if (q)
      (void)hipFree(q);
  }
  hipError_t ensure(size_t bytes) {
    if (bytes <= cap && p) return hipSuccess;
    raw_free(p, &guard);
    p = nullptr;
    cap = 0;
    if (bytes == 0) bytes = 8;
    hipError_t e = raw_alloc(bytes, &p, &guard);
    if (e == hipSuccess) cap = bytes;
    return e;
  }
  // Grows to at least `bytes` (with slack) and KEEPS the first `used` bytes. Other streams may still read the old block:
  // the device is drained before it is freed (growth is rare: the slack absorbs the row-by-row updates).
  hipError_t reserve_keep(size_t bytes, size_t used) {
    if (bytes <= cap && p) return hipSuccess;
    size_t ncap = guard_pages_on() ? bytes : bytes + bytes / 4 + 4096;  // (guard mode: no slack, the end of the buffer is the guard)
    void* q = nullptr;
    GuardAlloc ng;
    hipError_t e = raw_alloc(ncap, &q, &ng);
    if (e != hipSuccess) return e;
    (void)hipDeviceSynchronize();  // the engine's streams do not synchronise with the null stream used below
    if (p && used) {
      e = hipMemcpy(q, p, std::min(used, cap), hipMemcpyDeviceToDevice);
      if (e != hipSuccess) {
        raw_free(q, &ng);
        return e;
      }
    }
    raw_free(p, &guard);
    p = q;
    guard = ng;
    cap = ncap;
    return hipSuccess;
  }
  void release() {
    raw_free(p, &guard);
    p = nullptr;
    cap = 0;
  }
  template <class T>
  T* as() const {
    return static_cast<T*>(p);
  }
};

// A pod class: the four plugin-family signatures of the pod's spec + its pinned node (spec.nodeName).
struct ClassKey {
  int32_t a, b, c, d, pin;
  bool operator==(const ClassKey& o) const { return a == o.a && b == o.b && c == o.c && d == o.d && pin == o.pin; }
};
struct ClassKeyHash {
  size_t operator()(const ClassKey& k) const {
    uint64_t h = (uint64_t)(uint32_t)k.a * 0x9e3779b97f4a7c15ull;
    h ^= ((uint64_t)(uint32_t)k.b + 0x7f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
    h ^= ((uint64_t)(uint32_t)k.c + 0x1ce4e5b9ull) * 0x94d049bb133111ebull;
    h ^= ((uint64_t)(uint32_t)k.pin + 0x133111ebull) * 0xd6e8feb86659fd93ull;
    h ^= ((uint64_t)(uint32_t)k.d + 0x6659fd93ull) * 0xff51afd7ed558ccdull;
    return (size_t)(h ^ (h >> 29));
  }
};

struct Family {
  int D = 0;     // number of signatures
  int base = 0;  // first row of this family in the shared plane buffers
};

}  // namespace

struct ykpred_engine {
  // The reference's PredicateManager is called by concurrent readers (context.go:697,709 take read locks only), and
  // every entry point below uses engine-owned staging buffers and streams: calls on one engine are serialised here.
  mutable std::recursive_mutex mu;
  ykpred_config_t cfg{};
  int R = 3, KT = 1, W = 1;
  hipStream_t own_stream = nullptr;
  std::string err;

  // --- node table
  int N = 0;
  int row_words = 0, row_stride = 0;
  DevBuf d_alloc, d_req, d_allowed, d_count, d_nflags, d_taints, d_labels, d_domain, d_selcount, d_ports;
  int KD = 0, KS = 0, KP = 0;
  std::vector<int32_t> h_domain_sizes;
  DevBuf d_score, d_key, d_rank, d_perm, d_rankbuf, d_member_key, d_name_rank, d_member_tie;
  bool has_name_rank = false;
  bool nodes_set = false;

  // --- specs (host copies kept for class building)
  int S = 0;
  std::vector<i64> h_req;
  std::vector<u64> h_tol, h_wanted;
  std::vector<uint32_t> h_sflags;
  std::vector<int32_t> h_aff_off, h_pre_off, h_spread_off;
  std::vector<ykpred_spread_t> h_spread;
  std::vector<u64> h_aff_terms, h_pre_terms;
  DevBuf d_sreq, d_stol, d_sflags, d_aff_off, d_aff_terms, d_pre_off, d_pre_terms;  // per-spec tables (k_query / k_direct)
  // per-family signature tables
  std::vector<int32_t> spec_sig_res, spec_sig_tol, spec_sig_aff;
  Family fam_res, fam_tol, fam_aff, fam_spread;
  DevBuf planes_canon, planes_ranked;  // [total rows][row_stride] u64; families are row ranges (res, spread, tol, aff)
  DevBuf base_canon, base_ranked;      // bit-sliced dictionaries: [64W requirement | 64KT taint | unsched | exists][row_stride]
  u64 taint_used[ykk::kMaxKT] = {0, 0, 0, 0};  // OR of every node's taint words
  int plane_rows_alloc = 0;
  // PodTopologySpread signatures (host copies; device tables are rebuilt when nodes or specs change)
  std::vector<int32_t> spec_sig_spread;                 // [S] -1 = no hard constraints
  std::vector<std::vector<ykpred_spread_t>> spread_sig; // constraints of each signature
  std::vector<int32_t> spread_sig_aff, spread_sig_tol;  // eligibility signatures
  DevBuf d_spec_spread, d_sp_coff, d_sp_c, d_sp_aff, d_sp_tol, d_sp_cnt, d_sp_present, d_sp_min;
  DevBuf d_sp_cnt_prev, d_sp_present_prev, d_sp_min_prev, d_sig_changed, d_class_dirty;  // incremental path: what moved
  int spread_constraints = 0;
  int64_t spread_cells = 0;
  bool spread_dirty = true;
  // NodeResourcesFit value planes: row 0 = pod-independent part, then one row per (dimension, distinct request value) in
  // first-use order (kernels.hip.h: plane_dim). A request vector = res_rows[vec][1 + R] rows ANDed together.
  int res_vectors = 0;                                    // distinct request vectors (class key component)
  std::vector<i64> h_dim_val;                             // [rows]
  std::vector<int32_t> h_dim_of;                          // [rows] dimension, -1 for row 0
  std::vector<int32_t> h_stage_rows;                      // ballot rows (not index rows) of the family, at most ykk::kWalkMaxStage: k_walk_rows keeps them in LDS
  DevBuf d_dim_val, d_dim_order, d_dim_chunk_dim, d_dim_chunk_begin, d_dim_chunk_len, d_res_rows;
  int dim_chunks = 0;
  // dimensions with >= walk_rows distinct values are evaluated by the sorted walk (k_dim_sort / k_dim_walk)
  int walk_rows = 256;  // tunable: cfg.reserved[4]
  int wave_combine_below = 16;  // tunable: cfg.reserved[5] — average members per chunk below which k_combine_wave is used
  int n_big = 0, walk_chunks = 0, index_rows = 0;
  int NCB = 0;                      // zone-B chunks (d_chunk_list_b)
  DevBuf d_big_dim, d_walk_big, d_walk_begin, d_walk_len, d_sfree_c, d_pmask_c, d_sfree_r, d_pmask_r, d_rbits_c;
  // Sweep runs (k_sweep_rows): zone-B classes of one (toleration, affinity, spread, staged plane row) signature in ascending order of
  // their walked value. set_specs keeps what build_classes needs to find them: the request-vector table, every value row's position
  // in its dimension's sorted order, the sorted values themselves (device: k_dim_sort turns free values into positions).
  std::vector<int32_t> h_res_rows;   // [vectors][1 + R] as uploaded (index rows tagged with their walked dimension)
  std::vector<int32_t> h_row_pos;    // [rows] position in the walked dimension's ascending value order, -1: not a walked row
  std::vector<int32_t> h_row_big;    // [rows] its walked dimension (index into big_dim), -1
  std::vector<int32_t> h_sorted_off; // [n_big + 1]
  DevBuf d_sorted, d_sorted_off, d_ent_c;  // sorted values; cursor lists [n_big][65][row_words] (k_dim_sort, canonical order)
  int sweep_min_run = 1;             // tunable (YKPRED_TUNE sweep_min_run): runs shorter than this stay with k_walk_rows; 0 = no sweep
  int sweep_groups = 0;              // tunable (sweep_groups): persistent workgroups per segment; 0 = one per compute unit
  bool sweep_ready = false;          // the current class build has sweep rows and nothing has touched their classes since
  std::vector<int32_t> h_class_sweep;  // [C] 1 = the class is a row of a sweep run
  int sweep_rows[ykk::kMaxIdxRows] = {0, 0}, sweep_row_off[ykk::kMaxIdxRows + 1] = {0, 0, 0}, sweep_runs = 0;  // per walked dimension
  int index_rows_needed = 0;         // index rows some class OUTSIDE the sweep runs reads (the full pass walks only those)
  DevBuf d_agree;  // sharded rounds: what the ranks agree on before the first batch
  // fused rows (k_fused_rows): the zone-B classes no run kernel takes whose rows are plain plane rows, resolved to records at class-build time
  int fuse_rows = 1;                 // tunable (YKPRED_TUNE fuse_rows): 0 = those classes stay with the chunk writers
  int fuse_wpl = 0;                  // tunable (YKPRED_TUNE fuse_wpl): words per lane of k_fused_rows (1, 2, 5); 0 = from the row width
  int fuse_count = 0, fuse_row_count = 0;  // classes / bitmap rows
  bool fuse_ready = false;           // the current class build has fused classes and nothing has touched them since
  DevBuf d_fuse_rec;                 // FuseRec[fuse_count]
  int fuse_combos = 0;               // distinct (toleration, request vector) pairs among them: their ANDs are written first (0: records name their rows directly)
  int fuse_combine = 1;              // tunable (YKPRED_TUNE fuse_combine): 0 = never
  DevBuf d_fuse_combo_rec, d_fuse_combo;  // FuseRec[fuse_combos]; the combination table [fuse_combos][row_stride]
  std::vector<uint8_t> h_class_fused;
  // decisions of the sweep runs (k_run_decide): one range per run of the sweep row list, the classes k_decide leaves to it
  int run_decide = 1;                // tunable (YKPRED_TUNE run_decide): 0 = k_decide scans every class
  int run_ranges = 0, n_decide_list = 0, run_decide_classes = -1;  // (classes at the build the lists describe)
  bool win_partial = false;          // the rank-ordered windows of the last decision pass cover only the rows k_decide read (a round walks them all first)
  DevBuf d_run_ranges, d_no_decide, d_glin_r;  // RunRange; the classes k_decide still scans (int list); g of every node's free value in rank order [n_big][row_words * 64]
  // class runs (k_class_runs): zone-B classes WITHOUT an index row whose request-value rows are all staged, run by run
  int class_runs = 1;                // tunable (YKPRED_TUNE class_runs): 0 = those classes stay with the chunk writers
  int class_runs_min_rows = 16;      // tunable (class_runs_min_rows): rows of a signature from which its classes are written as a run
  int run_classes = 0, run_units = 0, run_rows = 0;
  DevBuf d_run_classes, d_run_units; // RunClass; unit bounds [run_units + 1]
  DevBuf d_sweep_rows, d_sweep_runs, d_sweep_units; // SweepRow {class, bitmap row, position, run}; SweepRun; unit bounds [units + 1] per walked dimension
  int sweep_units[ykk::kMaxIdxRows] = {0, 0}, sweep_unit_off[ykk::kMaxIdxRows + 1] = {0, 0, 0};
  DevBuf d_chunk_list_b0;            // [NCB0] the zone-B chunks outside the sweep runs (a pass with the sweep runs the chunk writers over these)
  int NCB0 = 0;
  DevBuf d_walk2_order, d_walk2_big, d_walk2_begin, d_walk2_len;  // walk chunks over the needed index rows only
  int walk2_chunks = -1;             // -1: no reduced walk (every index row is walked)
  DevBuf d_first_r;         // rank-ordered planes: first non-zero word per plane row (k_decide's starting point)
  static constexpr bool decide_skip = true;  // k_decide starts a class's scan where its rows can first have a bit
  DevBuf d_chunk_list_b;    // [NCB] numbers of the zone-B chunks (ascending)
  DevBuf d_slice_general;   // one int: chunks of the pass that k_walk_rows leaves to k_combine_wave
  DevBuf d_slice_desc;      // [NC] chunk descriptors of k_walk_rows (k_slice_desc, refilled per pass)
  DevBuf d_pfx_r;           // [n_big][row_words] running maximum of the free values along the bin-pack order (k_dim_prefix_max)
  DevBuf d_idx_c;           // index rows of the walked dimensions: [fam_res.D][idx_stride] bytes, canonical order
  DevBuf d_win_r;           // rank order: the 64-byte WINDOW of every index row (k_dim_walk_window): [fam_res.D][64]
  int idx_stride = 0;
  DevBuf d_sig_tol, d_sig_tolflags, d_sig_ports, d_swanted;                    // [Dtol][KT], [Dtol], [Dtol][KP]; [S][KP]
  DevBuf d_sig_aff_flags, d_sig_aff_off, d_sig_aff_terms, d_sig_pre_off, d_sig_pre_terms;
  bool specs_set = false;

  // --- pods / classes
  int P = 0, C = 0, NC = 0;
  std::vector<int32_t> h_pod_spec, h_pod_pin;
  DevBuf d_pod_spec, d_pod_pin, d_pod_class;
  DevBuf d_class_sig, d_class_pin, d_class_first, d_class_word, d_chunk_class, d_chunk_begin, d_chunk_len, d_chunk_first, d_members;
  // class index kept on the host so that ykpred_update_pods can move single rows between classes (mirrors of the device
  // tables above; `h_members` has -1 holes where a row left its class, chunks are only ever appended between rebuilds)
  std::unordered_map<ClassKey, int32_t, ClassKeyHash> class_ids;
  std::vector<int32_t> h_pod_class, h_pod_slot;                       // per pod: class, entry in h_members
  // Physical bitmap rows. The row of pod p is h_pod_row[p] (device: d_pod_row, ykpred_layout_t.row_of_pod): rows are laid
  // out for the WRITER — zone A [0, rows_a) in band order (see k_expand_bands), zone B after it class by class, rows of asks
  // that arrived or changed class since the last build appended at the end. Rows are never reused between two builds.
  std::vector<int32_t> h_pod_row;
  int rows_total = 0, rows_a = 0, row_capacity = 0;  // rows in use; rows of zone A; ykpred_set_row_capacity (0 = automatic)
  DevBuf d_pod_row;
  // zone A: band layout tables (built by build_classes, read by k_class_rows / k_expand_bands / k_fix_rows)
  bool bands_enabled = true;       // tunable: cfg.reserved[6] == -1 disables the band layout (every class in zone B)
  int band_steps = 0;              // tunable: cfg.reserved[6] > 0 (4..256); 0 = chosen per node table from the row length
  int band_steps_now = 128;        // the band height the current class build used
  int round_batched = -1;           // round_batched: rounds on one GPU in batches (parallel proposals, the host replays the loop; sharded engines always): 1 = always, 0 = never, -1 = by the ask list's mean run length
  int round_node_assume = 1;        // round_node_assume: 0 = batched rounds always assume ask after ask (k_allocate_round's assume mode)
  int round_prof = 0;               // round_prof: 1 = k_allocate_round counts thread 0's ticks per phase, printed on stderr
  int fail_after = 0;               // fail_after: failure injection — the n-th checked device call fails (0 = off)
  // Test knobs (YKPRED_TUNE, see ykpred_create): they force paths the populations of the test suite would not choose themselves.
  int sig_wpl = 0;                  // sig_wpl: row words per lane of k_sig_planes (1, 2, 4); 0 = from the row width
  int combine_slices = 1;           // combine_slices: 0 = index-row populations take the wave-per-chunk writer instead of k_walk_rows
  int decide_groups_from = 16384;   // decide_groups_from: classes from which k_decide serves four classes per wave
  int early_counts = 1;             // early_counts: 0 = class counts by the writers and the per-ask scatter behind them (the round-4 order)
  int max_lds_bytes = 64 * 1024;   // opt-in dynamic LDS limit of the device (hipDeviceAttributeMaxSharedMemoryPerBlock)
  int num_cus = 256;               // compute units of the device (hipDeviceAttributeMultiprocessorCount)
  int n_bands = 0, n_band_steps = 0, n_classes_a = 0, n_fix_rows = 0;
  std::vector<int32_t> h_class_slot_a;  // [C] index into the class-row table, -1 = zone B class
  DevBuf d_band_tab, d_class_rows_a, d_class_list_a, d_class_slot_a, d_fix_row, d_fix_slot, d_chunk_zone;
  DevBuf d_class_list_b;            // the zone-B classes (k_class_rows counts them too when there are few: ykpred_eval)
  int n_classes_b = 0;
  std::vector<int32_t> h_ch_zone;
  std::vector<uint8_t> h_row_stale;  // bitmap row rewritten by ykpred_update_pods and not re-evaluated yet: it must not serve
                                     // as the representative row of its class (k_column_class reads that row)
  std::vector<int32_t> h_class_sig, h_class_pin, h_class_first, h_class_live;
  std::vector<std::vector<int32_t>> h_class_chunks;                   // chunk ids of a class, creation order
  std::vector<int32_t> h_members, h_ch_class, h_ch_begin, h_ch_len, h_ch_first;
  int patch_chunks = 0;     // chunks appended since the last build_classes
  bool rank_valid = false;  // d_rank / d_perm / d_key describe the current node table
  // what the rank-ordered planes (and their first-word table) of the last decision pass describe: ykpred_allocate_round scans them
  uint64_t specs_version = 1, ranked_specs_version = 0, ranked_nodes_epoch = 0;
  unsigned ranked_pre = 0, ranked_filt = 0;
  bool ranked_has_first = false;
  DevBuf d_round;           // scratch of ykpred_allocate_round
  // what an assumed pod of a spec adds to its node besides resources (ykpred_set_spec_effects): valid for specs_version == fx_version
  DevBuf d_fx_off, d_fx_cls, d_fx_cnt, d_fx_occ, d_sp_sig_of;
  uint64_t fx_version = 0;
  bool fx_contrib = false, fx_ports = false;  // some spec adds to a count column / occupies a dictionary host port
  std::vector<int32_t> h_fx_off;              // [S + 1] the contribution rows' offsets, host copy
  std::vector<int32_t> h_fx_cls;              // the selector class of every contribution row, host copy
  std::vector<uint8_t> h_fx_occupies;         // [S] the spec's pods occupy a host port once on a node
  DevBuf d_patches, d_rows, d_row_count, d_row_best;
  unsigned last_pre = 0, last_filt = 0;  // plugin lists of the last full evaluation (ykpred_eval_nodes must match them)
  bool last_eval_valid = false;
  DevBuf d_class_count, d_class_best;
  bool pods_set = false, classes_dirty = true;
  int chunk_members = ykk::kChunkMembers;  // tunable: cfg.reserved[0] (1..64)
  bool chunk_sorted = true;                // tunable: cfg.reserved[1] == 1 disables the pod-order dispatch
  int combine_lds_bytes = 40000;           // tunable: cfg.reserved[2] — LDS reserved per k_combine block = occupancy cap of
                                           // 4 blocks (16 waves) per CU: measured 1.11 ms vs 1.19 ms at 8 blocks/CU — the
                                           // write path prefers fewer, longer streams (cf. scripts/fill_probe.hip)

  // --- outputs
  DevBuf d_bitmap, d_counts, d_decisions, d_keys, d_scratch;
  void* last_bitmap = nullptr;
  int64_t last_bitmap_rows = 0;  // rows the buffer of the last evaluation holds (caller-owned: eval_args.bitmap_rows / row capacity)
  void *last_counts = nullptr, *last_decisions = nullptr, *last_keys = nullptr;

  // --- decision stream (score → rank → ranked planes → decide run beside the bitmap branch)
  hipStream_t aux_stream = nullptr;
  hipEvent_t ev_fork = nullptr, ev_planes = nullptr, ev_join = nullptr, ev_counts = nullptr, ev_zero = nullptr;
  // --- the resident answer served to single callbacks (ykpred_peek_row): a copy stream ordered after the
  // last evaluation by an event, pinned staging memory
  hipStream_t peek_stream = nullptr;
  hipEvent_t ev_eval_done = nullptr;
  void* peek_pinned = nullptr;
  size_t peek_pinned_bytes = 0;

  // --- hipGraph cache of the full evaluation: the ~15 launches / memsets / cross-stream events of one pass are captured
  // once per (tables version, plugin lists, options, output buffers) and replayed; any table change bumps the version
  struct GraphKey {
    uint64_t version;
    unsigned pre, filt, options;
    void *bitmap, *counts, *decisions, *keys;
    bool operator==(const GraphKey& o) const {
      return version == o.version && pre == o.pre && filt == o.filt && options == o.options && bitmap == o.bitmap && counts == o.counts &&
             decisions == o.decisions && keys == o.keys;
    }
  };
  std::vector<std::pair<GraphKey, hipGraphExec_t>> graphs;
  std::vector<GraphKey> seen;  // a pass is captured the SECOND time it is asked for unchanged: one-off passes (after every
                               // table change of the incremental path) never pay for capture + instantiation
  uint64_t tables_version = 1;
  bool graph_disabled = true;   // opt-in (cfg.reserved[3] == 1 / YKPRED_GRAPH=1); also set when a capture failed once

  // --- multi-GPU (node-axis shards): RCCL communicator, this shard's place in the cluster, exchange scratch
  ncclComm_t comm = nullptr;
  int comm_rank = 0, comm_world = 1, node_offset = 0;
  int64_t round_exchanges = 0;      // proposal exchanges of sharded allocation rounds so far
  void* h_round_pinned = nullptr;   // page-locked landing buffer of a batched round's proposals (world x one shard's bytes)
  size_t h_round_pinned_bytes = 0;
  int64_t rounds_batched = 0, round_batched_asks = 0, round_batches = 0, rounds_sequential = 0, round_sequential_asks = 0;  // ykpred_get_round_info
  int forced_stride = 0;  // ykpred_set_row_stride
  DevBuf d_gathered, d_gathered_map, d_xkey, d_xcand;
  bool last_has_keys = false;
  // class-compressed gather (ykpred_gather_bitmap_compressed): this shard's class rows, everybody's, expansion scratch
  DevBuf d_class_rows_all, d_gathered_classes, d_class_rows_slot, d_class_sig_ident, d_expand_count, d_layout_hash, d_gathered_pod_class, d_row_pod;
  int ident_classes = 0;                                     // d_class_sig_ident is filled for this many classes
  uint64_t layout_version = 1, layout_hashed_version = 0, layout_hash_value = 0, row_pod_version = 0;
  // Bumped by every call that can change the class layout (set_nodes / set_specs / set_pods / update_pods / set_row_*): the
  // shards of a cluster make those calls in lockstep (they hold the same asks), so "ask_epoch moved" is a COLLECTIVE fact and
  // the header exchange of the compressed gather can be cached by it without one shard skipping a collective the others enter.
  uint64_t ask_epoch = 1, hdr_epoch = 0;
  std::vector<uint64_t> hdr_cache;  // [world][4] of the last header exchange
  // PodTopologySpread / InterPodAffinity histograms: valid for the node / spec tables of `hist_epoch`
  uint64_t nodes_epoch = 1, hist_epoch = 0;
  uint64_t bitmap_epoch = 0;  // nodes_epoch the bitmap of the last evaluation (full pass or column patch) describes

  // --- counters (ykpred_get_counters)
  int64_t n_full_evals = 0, n_node_patches = 0, n_row_patches = 0, n_queries = 0, n_gathers = 0, n_uploads = 0;

  // --- timing: one (start, stop) event pair per kernel, recorded on the stream the kernel is launched on
  hipEvent_t ev[2 * YKPRED_MAX_TIMED_KERNELS + 2]{};
  bool ev_ready = false;
  int timed = 0;
  const char* timed_name[YKPRED_MAX_TIMED_KERNELS]{};
  bool timing_valid = false;
};

namespace {

int fail(ykpred_engine* e, int code, const std::string& msg) {
  if (e) e->err = msg;
  if (e && (code == YKPRED_E_DEVICE || code == YKPRED_E_NOMEM)) {
    // a device call failed somewhere inside an entry point: whatever that call was building is unfinished. Nothing derived from
    // it may be served — the caller re-uploads its tables (the host library does: host.cpp fail()) and evaluates again.
    e->last_eval_valid = false;
    e->rank_valid = false;
    e->classes_dirty = true;
    e->spread_dirty = true;
    e->hist_epoch = 0;
    e->ranked_nodes_epoch = 0;
    e->tables_version++;
  }
  return code;
}
// Failure injection (tests only, YKPRED_TUNE fail_after=n): the n-th checked device call of the engine fails WITHOUT being made.
inline bool inject_failure(ykpred_engine* e) { return e && e->fail_after > 0 && --e->fail_after == 0; }
#define HIPCHK(call)                                                                                        \
  do {                                                                                                      \
    hipError_t _s = inject_failure(e) ? hipErrorUnknown : (call);                                          \
    if (_s != hipSuccess)                                                                                   \
      return fail(e, _s == hipErrorOutOfMemory ? YKPRED_E_NOMEM : YKPRED_E_DEVICE,                          \
                  std::string(#call) + ": " + hipGetErrorString(_s));                                       \
  } while (0)

template <class T>
int upload(ykpred_engine* e, DevBuf& b, const T* src, size_t n, hipStream_t st) {
  HIPCHK(b.ensure(n * sizeof(T)));
  if (n) HIPCHK(hipMemcpyAsync(b.p, src, n * sizeof(T), hipMemcpyHostToDevice, st));
  return YKPRED_OK;
}
#define TRY(x)                     \
  do {                             \
    int _r = (x);                  \
    if (_r != YKPRED_OK) return _r; \
  } while (0)

ykk::NodeTable node_table(const ykpred_engine* e) {
  ykk::NodeTable t;
  t.n = e->N;
  t.R = e->R;
  t.KT = e->KT;
  t.W = e->W;
  t.alloc = e->d_alloc.as<i64>();
  t.req = e->d_req.as<i64>();
  t.allowed = e->d_allowed.as<int>();
  t.count = e->d_count.as<int>();
  t.flags = e->d_nflags.as<unsigned>();
  t.taints = e->d_taints.as<u64>();
  t.labels = e->d_labels.as<u64>();
  t.KD = e->KD;
  t.KS = e->KS;
  t.domain = e->d_domain.as<int>();
  t.selcount = e->d_selcount.as<int>();
  t.KP = e->KP;
  t.ports = e->d_ports.as<u64>();
  return t;
}
ykk::SpreadSigs spread_sigs(const ykpred_engine* e) {
  ykk::SpreadSigs sp;
  sp.D = e->fam_spread.D;
  sp.c_off = e->d_sp_coff.as<int>();
  sp.c = e->d_sp_c.as<ykk::SpreadC>();
  sp.aff_sig = e->d_sp_aff.as<int>();
  sp.tol_sig = e->d_sp_tol.as<int>();
  sp.cnt = e->d_sp_cnt.as<int>();
  sp.present = e->d_sp_present.as<int>();
  sp.minv = e->d_sp_min.as<int>();
  return sp;
}
ykk::SpecTable spec_table(const ykpred_engine* e) {
  ykk::SpecTable s;
  s.R = e->R;
  s.KT = e->KT;
  s.W = e->W;
  s.req = e->d_sreq.as<i64>();
  s.tol = e->d_stol.as<u64>();
  s.flags = e->d_sflags.as<unsigned>();
  s.aff.flags = e->d_sflags.as<unsigned>();
  s.aff.term_off = e->d_aff_off.as<int>();
  s.aff.terms = e->d_aff_terms.as<u64>();
  s.aff.pre_off = e->d_pre_off.as<int>();
  s.aff.pre_terms = e->d_pre_terms.as<u64>();
  s.KP = e->KP;
  s.wanted_ports = e->d_swanted.as<u64>();
  s.spread_sig = e->d_spec_spread.as<int>();
  s.spread = spread_sigs(e);
  return s;
}

// Groups pods into classes (same signatures + same pinned node) and classes into chunks of <= kChunkMembers pods.
int build_classes(ykpred_engine* e, hipStream_t st) {
  e->tables_version++;
  const int P = e->P;
  auto& ids = e->class_ids;
  ids.clear();
  ids.reserve(1024);
  auto& pod_class = e->h_pod_class;
  pod_class.assign((size_t)P, 0);
  auto &class_sig = e->h_class_sig, &class_pin = e->h_class_pin, &class_size = e->h_class_live;
  class_sig.clear();
  class_pin.clear();
  class_size.clear();
  for (int p = 0; p < P; ++p) {
    int s = e->h_pod_spec[(size_t)p];
    ClassKey k{e->spec_sig_res[(size_t)s], e->spec_sig_tol[(size_t)s], e->spec_sig_aff[(size_t)s], e->spec_sig_spread[(size_t)s],
               e->h_pod_pin[(size_t)p]};
    auto it = ids.find(k);
    int32_t c;
    if (it == ids.end()) {
      c = (int32_t)class_pin.size();
      ids.emplace(k, c);
      class_sig.insert(class_sig.end(), {k.a, k.b, k.c, k.d});
      class_pin.push_back(k.pin);
      class_size.push_back(0);
    } else {
      c = it->second;
    }
    pod_class[(size_t)p] = c;
    class_size[(size_t)c]++;
  }
  const int C = (int)class_pin.size();
  std::vector<int32_t> class_off((size_t)C + 1, 0);
  for (int c = 0; c < C; ++c) class_off[(size_t)c + 1] = class_off[(size_t)c] + class_size[(size_t)c];
  auto& members = e->h_members;
  members.assign((size_t)P, 0);
  e->h_pod_slot.assign((size_t)P, 0);
  std::vector<int32_t> cursor(class_off.begin(), class_off.end() - 1);
  for (int p = 0; p < P; ++p) {
    int slot = cursor[(size_t)pod_class[(size_t)p]]++;
    members[(size_t)slot] = p;
    e->h_pod_slot[(size_t)p] = slot;
  }
  // ---- physical rows. Zone A (band layout): the bitmap is written like a linear fill — ykk::kBandGroups
  // workgroups, 4 KiB aligned tiles, a window of kBandGroups * 4 KiB that advances one step at a time — so the rows are
  // permuted for the writer: inside a band of `band_steps` windows a class owns the rows whose start offset inside their
  // window, X(r) = (r * row_bytes) mod window, falls into one interval. Workgroup b always writes window bytes
  // [4096 b, 4096 b + 4096): during a band it needs the rows of at most kBandClasses classes, which it keeps in LDS.
  const long row_b = (long)e->row_stride * 8, win = (long)ykk::kBandGroups * 4096;
  const int KC = ykk::kBandClasses;
  // a piece of a class inside one band must be wide enough in X that no workgroup range (4096 + row bytes) meets more
  // than KC pieces: (rows of the piece / S) * row_b >= (4096 + row_b) / (KC - 1), with a margin; verified below
  auto piece_min_of = [&](int steps) { return (int)((double)steps * (4096.0 + (double)row_b) / (double)row_b / (double)(KC - 1) * 1.3) + 3; };
  // Band height S. A class is admitted to the band layout from 2 * piece_min(S) rows on, and piece_min grows with S and with
  // (4096 + row bytes) / row bytes — 190 rows at 50 000 nodes, but 620 at the 6 272-node shard of an 8-way cluster, where the
  // 100-member task groups of BASELINE configs[3] would all fall to the class-by-class writer. So S is chosen per build from
  // the class sizes: the tallest band that still admits (nearly) every row a short band would, but never so short that the
  // class rows re-read per band (KC rows per S tiles of 4 KiB) become a visible share of the traffic.
  int S = e->band_steps;
  if (S <= 0) {
    int s_min = 8;
    while (s_min < 128 && (long)s_min * 4096 < 5L * KC * row_b) s_min *= 2;
    auto coverage = [&](int steps) {
      const int need = 2 * piece_min_of(steps);
      long rows = 0;
      for (int c = 0; c < C; ++c)
        if (class_size[(size_t)c] >= need) rows += class_size[(size_t)c];
      return rows;
    };
    // (measured, profiles/r03_writer_knobs.txt: where a tall band already takes most rows, admitting the smaller classes too
    // LOSES — many short pieces per band slow the band writer more than the chunk writer gains; a shorter band is for
    // populations the tall one would leave almost entirely to the chunk writer)
    const long best = coverage(s_min);
    S = 128;
    while (S > s_min && (double)coverage(S) < 0.5 * (double)best) S /= 2;
  }
  while (S > 4 && (long)S * win / row_b >= (1L << 21)) S -= 4;  // the builder's sort key holds the row-in-band in 21 bits
  e->band_steps_now = S;
  const int piece_min = piece_min_of(S);
  e->h_class_slot_a.assign((size_t)C, -1);
  e->h_pod_row.assign((size_t)P, -1);
  std::vector<ykk::BandEntry> band_tab;
  std::vector<int32_t> class_list_a, fix_row, fix_slot;
  int n_steps = 0, n_bands = 0, rows_a = 0;
  const bool geometry_ok = e->bands_enabled && e->N > 0 && (size_t)2 * KC * (size_t)row_b <= (size_t)std::min(ykk::kBandMaxLds, e->max_lds_bytes);
  if (geometry_ok) {
    long rows_needed = 0;
    for (int c = 0; c < C; ++c)
      if (class_size[(size_t)c] >= 2 * piece_min) {
        e->h_class_slot_a[(size_t)c] = (int32_t)class_list_a.size();
        class_list_a.push_back(c);
        rows_needed += class_size[(size_t)c];
      }
    if (rows_needed < (long)S * win / row_b / 4) {  // not even a quarter of a band: the chunk kernel does it all
      class_list_a.clear();
      std::fill(e->h_class_slot_a.begin(), e->h_class_slot_a.end(), -1);
    }
  }
  if (!class_list_a.empty()) {
    // band by band: the band's rows in (X, step) order, filled with class pieces in class order
    struct Piece {
      int32_t x, s, slot;  // key of the piece's first row, class-row slot
    };
    size_t ci = 0;            // next zone-A class
    int32_t left = class_size[(size_t)class_list_a[0]], taken = 0;  // rows of class ci still to place / already placed
    bool ok = true;
    std::vector<uint64_t> keys;
    std::vector<Piece> pieces;
    int band = 0;
    while (ci < class_list_a.size() && ok) {
      // rows still to place decide the height of this band (the last one is shorter: no tail of unused windows)
      const long band_rows = (long)S * win / row_b;
      long remaining = left;
      for (size_t k = ci + 1; k < class_list_a.size() && remaining <= band_rows; ++k) remaining += class_size[(size_t)class_list_a[k]];
      const long s_lo = (long)band * S;
      long band_steps_now = S;
      if (remaining + piece_min + 2 < band_rows) {  // the last band: only as many windows as the remaining rows need
        band_steps_now = ((remaining + piece_min + 2) * row_b + win - 1) / win;
        band_steps_now = std::min<long>(S, std::max<long>(4, (band_steps_now + 3) / 4 * 4));
      }
      const long s_hi = s_lo + band_steps_now;
      const long r_lo = (s_lo * win + row_b - 1) / row_b, r_hi = (s_hi * win + row_b - 1) / row_b;
      // key = X (< 2^20 … 2^22) | step | row-in-band: sorting the keys orders the band's rows by (X, step)
      keys.resize((size_t)(r_hi - r_lo));
      for (size_t i = 0; i < keys.size(); ++i) {
        const long r = r_lo + (long)i, start = r * row_b;
        keys[i] = ((uint64_t)(start % win) << 42) | ((uint64_t)(start / win) << 21) | (uint64_t)i;
      }
      std::sort(keys.begin(), keys.end());
      pieces.clear();
      size_t pos = 0;  // next free position of the sorted order
      const size_t cap = keys.size();
      while (pos < cap && ci < class_list_a.size()) {
        const size_t room = cap - pos;
        if (room < (size_t)piece_min) break;  // the rest of the band stays unused
        size_t take = std::min<size_t>((size_t)left, room);
        if ((size_t)left > take && (size_t)left - take < (size_t)piece_min) {
          // the remainder would be too narrow in the next band: leave it piece_min rows
          if (take < (size_t)piece_min + (size_t)piece_min) break;
          take = (size_t)left - (size_t)piece_min;
        }
        const int c = class_list_a[ci];
        const uint64_t k0 = keys[pos];
        pieces.push_back({(int32_t)(k0 >> 42), (int32_t)((k0 >> 21) & 0x1fffff), (int32_t)ci});
        for (size_t i = 0; i < take; ++i) {
          const long r = r_lo + (long)(keys[pos + i] & 0x1fffff);
          const int p = members[(size_t)(class_off[(size_t)c] + taken + (int32_t)i)];
          e->h_pod_row[(size_t)p] = (int32_t)r;
        }
        pos += take;
        taken += (int32_t)take;
        left -= (int32_t)take;
        if (left == 0) {
          ++ci;
          taken = 0;
          if (ci < class_list_a.size()) left = class_size[(size_t)class_list_a[ci]];
        }
      }
      // what every workgroup needs during this band
      for (int b = 0; b < ykk::kBandGroups && ok; ++b) {
        ykk::BandEntry be{};
        for (int i = 0; i < KC; ++i) be.slot[i] = pieces.empty() ? 0 : pieces[0].slot;
        for (int i = 0; i < KC - 1; ++i) {
          be.xb[i] = 0x7fffffff;
          be.sb[i] = 0;
        }
        const long x_lo = std::max<long>(0, (long)b * 4096 - row_b + 1), x_hi = (long)(b + 1) * 4096;  // rows starting in [x_lo, x_hi) meet the tile
        // first piece: the last one whose first key is not after (x_lo, first step)
        size_t ia = 0;
        for (size_t i = 0; i < pieces.size(); ++i)
          if ((long)pieces[i].x < x_lo) ia = i;
        if (!pieces.empty()) be.slot[0] = pieces[ia].slot;
        int n = 1;
        for (size_t i = ia + 1; i < pieces.size() && (long)pieces[i].x < x_hi; ++i) {
          if (n >= KC) {
            ok = false;
            break;
          }
          be.slot[n] = pieces[i].slot;
          be.xb[n - 1] = pieces[i].x;
          be.sb[n - 1] = pieces[i].s;
          ++n;
        }
        for (int i = n; i < KC; ++i) be.slot[i] = be.slot[n - 1];
        be.n = n;
        be.steps = (int32_t)band_steps_now;
        be.first_step = (int32_t)s_lo;
        band_tab.push_back(be);
      }
      ++band;
      n_steps = (int)s_hi;
      rows_a = (int)r_hi;
      if (band_steps_now < S) break;  // a shorter band is the last one
    }
    if (ci < class_list_a.size()) ok = false;  // (a shortened last band that turned out too small)
    if (!ok) {
      // fall back: everything in zone B
      class_list_a.clear();
      std::fill(e->h_class_slot_a.begin(), e->h_class_slot_a.end(), -1);
      std::fill(e->h_pod_row.begin(), e->h_pod_row.end(), -1);
      band_tab.clear();
      n_steps = n_bands = rows_a = 0;
    } else {
      n_bands = band;
      // the rows that straddle a window boundary: the band kernel writes only the part of a row that lies in the window it
      // starts in — one row per window is finished by k_fix_rows
      std::vector<int32_t> row_slot((size_t)rows_a, -1);
      for (int c : class_list_a)
        for (int i = class_off[(size_t)c]; i < class_off[(size_t)c + 1]; ++i) row_slot[(size_t)e->h_pod_row[(size_t)members[(size_t)i]]] = e->h_class_slot_a[(size_t)c];
      for (long s1 = 1; s1 <= n_steps; ++s1) {
        const long r = (s1 * win - 1) / row_b;  // the row that holds the last byte before the boundary
        if (r * row_b + row_b > s1 * win && r < rows_a && row_slot[(size_t)r] >= 0) {
          fix_row.push_back((int32_t)r);
          fix_slot.push_back(row_slot[(size_t)r]);
        }
      }
    }
  }
  // zone B: the remaining classes, class by class after zone A — in the order of their SIGNATURES (node-affinity signature
  // first: the family with the most planes), so that the chunk kernel's neighbouring workgroups AND the same plane rows at the
  // same time and find them in L2 instead of fetching 6 KB rows from the far cache or HBM once per class
  // SWEEP RUNS (k_sweep_rows). A single-member, unpinned zone-B class whose request vector is the pod-independent row, at most one
  // staged ballot row and exactly ONE index row differs from the classes of the same (affinity, toleration, spread, ballot row)
  // only in the VALUE of its walked dimension. Inside such a group the classes are laid out in ascending order of that value:
  // a run. Runs of at least sweep_min_run rows are written by k_sweep_rows, which keeps a row's words in registers and only
  // clears the bits of the nodes the next value loses; everything else stays with the class-by-class writers.
  int next_row = rows_a;
  e->h_class_sweep.assign((size_t)C, 0);
  e->sweep_ready = false;
  e->sweep_runs = 0;
  e->run_ranges = 0;
  e->walk2_chunks = -1;
  e->index_rows_needed = e->index_rows;
  for (int b = 0; b < ykk::kMaxIdxRows; ++b) e->sweep_rows[b] = 0;
  {
    std::vector<int32_t> order_b;
    for (int c = 0; c < C; ++c)
      if (e->h_class_slot_a[(size_t)c] < 0) order_b.push_back(c);
    auto key = [&](int32_t c, int f) { return class_sig[(size_t)c * 4 + (size_t)f]; };
    const int R1 = 1 + e->R, big_mask = (1 << ykk::kRowBigShift) - 1;
    bool sweep_possible = e->sweep_min_run > 0 && e->n_big > 0 && !e->h_res_rows.empty() && (int)e->h_sorted_off.size() == e->n_big + 1;
    for (int b = 0; sweep_possible && b < e->n_big; ++b)
      sweep_possible = e->h_sorted_off[(size_t)b + 1] - e->h_sorted_off[(size_t)b] < (1 << 26);  // (a position and a node share 32 bits)
    // candidate classes: the walked dimension, the second ballot row, the position of the value (slot < 0: not a candidate)
    std::vector<int32_t> cand_big, cand_slot, cand_pos;
    if (sweep_possible) {
      cand_big.assign((size_t)C, -1);
      cand_slot.assign((size_t)C, -1);
      cand_pos.assign((size_t)C, -1);
      for (int32_t c : order_b) {
        const int sr = key(c, 0);
        if (class_size[(size_t)c] != 1 || class_pin[(size_t)c] != -1 || sr < 0 || (size_t)(sr + 1) * (size_t)R1 > e->h_res_rows.size()) continue;
        const int32_t* rr = e->h_res_rows.data() + (size_t)sr * (size_t)R1;
        int np = 0, ni = 0, prow = -1, irow = -1;
        for (int k = 0; k < R1; ++k) {
          const int r = rr[k];
          if (r < 0) continue;
          if (r >> ykk::kRowBigShift) {
            if (ni == 0) irow = r;
            ++ni;
          } else {
            if (np == 1) prow = r;
            ++np;
          }
        }
        if (ni != 1 || np < 1 || np > 2) continue;
        const int slot = np == 2 ? prow : 0;  // the second ballot row of the request family (0: none — row 0 is in every class)
        const int pos = e->h_row_pos[(size_t)(irow & big_mask)];
        if (pos < 0) continue;
        cand_big[(size_t)c] = (irow >> ykk::kRowBigShift) - 1;
        cand_slot[(size_t)c] = slot;
        cand_pos[(size_t)c] = pos;
      }
    }
    auto is_cand = [&](int32_t c) { return sweep_possible && cand_slot[(size_t)c] >= 0; };
    // BALLOT-ROW classes for k_class_runs: unpinned, no index row, at most four request-value rows, every one of them staged in LDS
    // by the writers (h_stage_rows); run_slots = their slots, a byte each (n_stage: none — the all-ones row)
    std::vector<int32_t> run_slots;
    const bool runs_possible = e->class_runs != 0 && !e->h_res_rows.empty() && e->h_stage_rows.size() < 255;
    if (runs_possible) {
      run_slots.assign((size_t)C, -1);
      const int n_stage = (int)e->h_stage_rows.size();
      for (int32_t c : order_b) {
        const int sr = key(c, 0);
        if (class_pin[(size_t)c] != -1 || sr < 0 || (size_t)(sr + 1) * (size_t)R1 > e->h_res_rows.size()) continue;
        const int32_t* rr = e->h_res_rows.data() + (size_t)sr * (size_t)R1;
        int n = 0, packed = 0;
        bool ok = true;
        for (int k = 0; k < R1 && ok; ++k) {
          const int r = rr[k];
          if (r <= 0) continue;  // unused slot, or row 0 (part of every run's base)
          if (r >> ykk::kRowBigShift) ok = false;
          int slot = -1;
          for (int q = 0; q < n_stage && ok; ++q)
            if (e->h_stage_rows[(size_t)q] == r) slot = q;
          if (slot < 0 || n >= 4) ok = false;
          else packed |= slot << (8 * n++);
        }
        if (!ok) continue;
        for (; n < 4; ++n) packed |= n_stage << (8 * n);
        run_slots[(size_t)c] = packed;
      }
    }
    auto kind_of = [&](int32_t c) { return is_cand(c) ? 2 : (runs_possible && run_slots[(size_t)c] >= 0 ? 1 : 0); };
    // The classes the run kernels will NOT take — shapes without a run form, signatures with too few rows — go to the END of zone B,
    // side by side: the chunk writers then store consecutive rows (their chunks run in row order) instead of single rows scattered
    // between the runs of 6 GB.
    std::vector<uint8_t> tail((size_t)C, 0);
    {
      struct SigKey {
        int32_t a, t, s, k, b, p;
        bool operator==(const SigKey& o) const { return a == o.a && t == o.t && s == o.s && k == o.k && b == o.b && p == o.p; }
      };
      struct SigHash {
        size_t operator()(const SigKey& q) const {
          uint64_t h = 0x9e3779b97f4a7c15ull;
          for (int32_t v : {q.a, q.t, q.s, q.k, q.b, q.p}) h = (h ^ (uint64_t)(uint32_t)v) * 0x100000001b3ull;
          return (size_t)h;
        }
      };
      std::unordered_map<SigKey, long, SigHash> rows_of_sig;
      auto sig_key = [&](int32_t c, int kind) {
        return SigKey{key(c, 2), key(c, 1), key(c, 3), kind, kind == 2 ? cand_big[(size_t)c] : 0, kind == 2 ? cand_slot[(size_t)c] : 0};
      };
      for (int32_t c : order_b) {
        const int kind = kind_of(c);
        if (kind) rows_of_sig[sig_key(c, kind)] += class_size[(size_t)c];
      }
      for (int32_t c : order_b) {
        const int kind = kind_of(c);
        tail[(size_t)c] = kind == 0 || rows_of_sig[sig_key(c, kind)] < (kind == 1 ? (long)e->class_runs_min_rows : (long)e->sweep_min_run);
      }
    }
    std::sort(order_b.begin(), order_b.end(), [&](int32_t x, int32_t y) {
      if (tail[(size_t)x] != tail[(size_t)y]) return tail[(size_t)x] < tail[(size_t)y];
      if (key(x, 2) != key(y, 2)) return key(x, 2) < key(y, 2);  // aff
      if (key(x, 1) != key(y, 1)) return key(x, 1) < key(y, 1);  // tol
      if (key(x, 3) != key(y, 3)) return key(x, 3) < key(y, 3);  // spread
      const int kx = kind_of(x), ky = kind_of(y);
      if (kx != ky) return kx < ky;                                // the others first, then the ballot-row classes, by request vector as before
      if (kx == 2) {
        if (cand_big[(size_t)x] != cand_big[(size_t)y]) return cand_big[(size_t)x] < cand_big[(size_t)y];
        if (cand_slot[(size_t)x] != cand_slot[(size_t)y]) return cand_slot[(size_t)x] < cand_slot[(size_t)y];
        if (cand_pos[(size_t)x] != cand_pos[(size_t)y]) return cand_pos[(size_t)x] < cand_pos[(size_t)y];
        return x < y;
      }
      if (key(x, 0) != key(y, 0)) return key(x, 0) < key(y, 0);  // request vector
      return x < y;
    });
    std::vector<int32_t> first_row_of((size_t)C, -1);
    for (int32_t c : order_b) {
      first_row_of[(size_t)c] = next_row;
      for (int i = class_off[(size_t)c]; i < class_off[(size_t)c + 1]; ++i) e->h_pod_row[(size_t)members[(size_t)i]] = next_row++;
    }
    e->n_classes_b = (int)order_b.size();
    TRY(upload(e, e->d_class_list_b, order_b.data(), order_b.size(), st));
    std::vector<ykk::SweepRun> runs;  // of both kinds
    e->run_classes = e->run_units = e->run_rows = 0;
    if (runs_possible) {
      std::vector<ykk::RunClass> rcs;
      std::vector<int32_t> starts;  // 1 = the class starts a run
      // A run is worth its start — a round of global loads and the wait for it, on a kernel with two workgroups per CU — from
      // class_runs_min_rows rows on; the signatures with a handful of rows stay with k_combine_wave, whose many waves hide exactly
      // that latency (measured on the own-template population, 410 k zone-B rows: every run through k_class_runs 0.89 ms, none 0.77 ms, the runs of 16 rows
      // and more 0.42 ms = 5.0 TB/s for their 333 k rows + 0.34 ms of k_combine_wave for the 77 k single-row classes).
      for (size_t i = 0; i < order_b.size();) {
        const int32_t c = order_b[i];
        if (kind_of(c) != 1) {
          ++i;
          continue;
        }
        size_t j = i;
        long rows_in_run = 0;
        while (j < order_b.size() && kind_of(order_b[j]) == 1 && key(order_b[j], 2) == key(c, 2) && key(order_b[j], 1) == key(c, 1) &&
               key(order_b[j], 3) == key(c, 3)) {
          rows_in_run += class_size[(size_t)order_b[j]];
          ++j;
        }
        if (rows_in_run >= e->class_runs_min_rows) {
          ykk::SweepRun run{};
          run.st = key(c, 1), run.sa = key(c, 2), run.ss = key(c, 3), run.prow = 0;
          runs.push_back(run);
          for (size_t k = i; k < j; ++k) {
            const int32_t m = order_b[k];
            ykk::RunClass rc{};
            rc.cls = m, rc.dest0 = first_row_of[(size_t)m], rc.len = class_size[(size_t)m], rc.slots = run_slots[(size_t)m], rc.run = (int32_t)runs.size() - 1;
            rcs.push_back(rc);
            starts.push_back(k == i ? 1 : 0);
            e->h_class_sweep[(size_t)m] = 1;
          }
        }
        i = j;
      }
      if (!rcs.empty()) {
        const int n = (int)rcs.size();
        auto cost = [&](int i) { return (long)rcs[(size_t)i].len + ykk::kRunsClassCost + (starts[(size_t)i] ? ykk::kSweepRunCost : 0); };
        long total = 0;
        for (int i = 0; i < n; ++i) total += cost(i);
        const int want = std::max(1, std::min(n / 2, 2 * e->num_cus * ykk::kRunsWaves * 8));
        std::vector<int32_t> units;
        long acc = 0;
        for (int i = 0; i < n; ++i) {
          if ((long)units.size() < want && acc * want >= (long)units.size() * total) units.push_back(i);
          acc += cost(i);
        }
        e->run_units = (int)units.size();
        units.push_back(n);
        e->run_classes = n;
        for (const ykk::RunClass& rc : rcs) e->run_rows += rc.len;
        TRY(upload(e, e->d_run_classes, rcs.data(), rcs.size(), st));
        TRY(upload(e, e->d_run_units, units.data(), units.size(), st));
      }
    }
    if (sweep_possible) {
      std::vector<int32_t> rows_of[ykk::kMaxIdxRows];  // flattened int4 {class, bitmap row, position, run}
      for (size_t i = 0; i < order_b.size();) {
        const int32_t c = order_b[i];
        if (!is_cand(c)) {
          ++i;
          continue;
        }
        size_t j = i + 1;
        while (j < order_b.size() && is_cand(order_b[j]) && key(order_b[j], 2) == key(c, 2) && key(order_b[j], 1) == key(c, 1) &&
               key(order_b[j], 3) == key(c, 3) && cand_big[(size_t)order_b[j]] == cand_big[(size_t)c] &&
               cand_slot[(size_t)order_b[j]] == cand_slot[(size_t)c])
          ++j;
        if ((int)(j - i) >= e->sweep_min_run) {
          const int big = cand_big[(size_t)c];
          ykk::SweepRun run{};
          run.st = key(c, 1), run.sa = key(c, 2), run.ss = key(c, 3), run.prow = cand_slot[(size_t)c];
          const int32_t id = (int32_t)runs.size();
          runs.push_back(run);
          for (size_t k = i; k < j; ++k) {
            const int32_t m = order_b[k];
            e->h_class_sweep[(size_t)m] = 1;
            rows_of[big].insert(rows_of[big].end(), {m, first_row_of[(size_t)m], cand_pos[(size_t)m], id});
          }
        }
        i = j;
      }
      e->sweep_runs = (int)runs.size();
      std::vector<int32_t> all_rows, all_units;
      std::vector<ykk::RunRange> ranges;
      std::vector<uint8_t> no_decide((size_t)C, 0);
      e->sweep_row_off[0] = 0;
      e->sweep_unit_off[0] = 0;
      for (int b = 0; b < ykk::kMaxIdxRows; ++b) {
        const int n = (int)(rows_of[b].size() / 4);
        e->sweep_rows[b] = n;
        for (int i = 0; i < n;) {  // one range per run of this dimension's list
          int j = i + 1;
          while (j < n && rows_of[b][(size_t)j * 4 + 3] == rows_of[b][(size_t)i * 4 + 3]) ++j;
          for (int q = i; q < j; q += ykk::kRunDecideRows)
            ranges.push_back(ykk::RunRange{rows_of[b][(size_t)i * 4 + 3], b, (int32_t)(all_rows.size() / 4) + q, std::min(ykk::kRunDecideRows, j - q)});
          i = j;
        }
        for (int i = 0; i < n; ++i) no_decide[(size_t)rows_of[b][(size_t)i * 4]] = 1;
        all_rows.insert(all_rows.end(), rows_of[b].begin(), rows_of[b].end());
        e->sweep_row_off[b + 1] = e->sweep_row_off[b] + n;
        // units of equal estimated cost (a row = 1, a run start = kSweepRunCost): eight per wave of a full launch, none empty
        const int want = std::max(1, std::min(n / 4, e->num_cus * ykk::kSweepWaves * 8));
        long total = 0;
        for (int i = 0; i < n; ++i) total += 1 + ((i == 0 || rows_of[b][(size_t)i * 4 + 3] != rows_of[b][(size_t)(i - 1) * 4 + 3]) ? ykk::kSweepRunCost : 0);
        const size_t first = all_units.size();
        long acc = 0;
        for (int i = 0; i < n; ++i) {
          const long unit = (long)(all_units.size() - first);
          if (unit < want && acc * want >= unit * total) all_units.push_back(i);  // (unit k begins at the first row whose prefix cost reaches k / want)
          acc += 1 + ((i == 0 || rows_of[b][(size_t)i * 4 + 3] != rows_of[b][(size_t)(i - 1) * 4 + 3]) ? ykk::kSweepRunCost : 0);
        }
        e->sweep_units[b] = (int)(all_units.size() - first);
        all_units.push_back(n);
        e->sweep_unit_off[b + 1] = (int)all_units.size();
      }
      all_units.push_back(0);
      TRY(upload(e, e->d_sweep_units, all_units.data(), all_units.size(), st));
      e->run_ranges = 0;
      if (e->sweep_row_off[ykk::kMaxIdxRows] > 0) {
        all_rows.push_back(0);
        TRY(upload(e, e->d_sweep_rows, all_rows.data(), all_rows.size(), st));
        e->run_ranges = (int)ranges.size();
        TRY(upload(e, e->d_run_ranges, ranges.data(), ranges.size(), st));
        std::vector<int32_t> decide_list;
        for (int c = 0; c < C; ++c)
          if (!no_decide[(size_t)c]) decide_list.push_back(c);
        e->n_decide_list = (int)decide_list.size();
        e->run_decide_classes = C;
        decide_list.push_back(0);
        TRY(upload(e, e->d_no_decide, decide_list.data(), decide_list.size(), st));
        // The index rows somebody OUTSIDE the runs still reads (zone-A classes, short runs, shapes without a fast path): the full
        // pass walks only those (k_dim_walk over the reduced chunk lists below); a dirty-class pass walks them all.
        const int n_rows = (int)e->h_row_pos.size();
        std::vector<uint8_t> needed((size_t)n_rows, 0);
        for (int c = 0; c < C; ++c) {
          const int sr = key(c, 0);
          if (e->h_class_sweep[(size_t)c] || sr < 0 || (size_t)(sr + 1) * (size_t)R1 > e->h_res_rows.size()) continue;
          for (int k = 0; k < R1; ++k) {
            const int r = e->h_res_rows[(size_t)sr * (size_t)R1 + (size_t)k];
            if (r > 0 && (r >> ykk::kRowBigShift)) needed[(size_t)(r & big_mask)] = 1;
          }
        }
        // in ascending value order per walked dimension: bucket by position
        std::vector<int32_t> order2, wbig2, wbegin2, wlen2;
        for (int b = 0; b < e->n_big; ++b) {
          const int cnt = e->h_sorted_off[(size_t)b + 1] - e->h_sorted_off[(size_t)b];
          std::vector<int32_t> at_pos((size_t)cnt, -1);
          for (int r = 0; r < n_rows; ++r)
            if (needed[(size_t)r] && e->h_row_pos[(size_t)r] >= 0 && e->h_row_big[(size_t)r] == b) at_pos[(size_t)e->h_row_pos[(size_t)r]] = r;
          const int begin = (int)order2.size();
          for (int32_t r : at_pos)
            if (r >= 0) order2.push_back(r);
          // (few rows: chunks of ONE row. A chunk is one thread's sequential walk down every word's sorted list — a chain of dependent
          // loads that grows with the distance between the chunk's values, and the rows that are left are far apart: 16-row chunks of
          // 1 112 rows made k_dim_walk_window the longest kernel of the decision stream, 1.1 ms beside the sweep)
          const int left = (int)order2.size() - begin;
          const int clen = std::min(ykk::kWalkRows, std::max(1, left / 2048));
          for (int q = begin; q < (int)order2.size(); q += clen) {
            wbig2.push_back(b);
            wbegin2.push_back(q);
            wlen2.push_back(std::min(clen, (int)order2.size() - q));
          }
        }
        e->index_rows_needed = (int)order2.size();
        e->walk2_chunks = (int)wbig2.size();
        order2.push_back(0);
        TRY(upload(e, e->d_walk2_order, order2.data(), order2.size(), st));
        TRY(upload(e, e->d_walk2_big, wbig2.data(), wbig2.size(), st));
        TRY(upload(e, e->d_walk2_begin, wbegin2.data(), wbegin2.size(), st));
        TRY(upload(e, e->d_walk2_len, wlen2.data(), wlen2.size(), st));
      }
    }
    if (getenv("YKPRED_TRACE_RUNS")) fprintf(stderr, "runs: %zu runs, %d run classes (%d rows, %d units), %d sweep rows\n", runs.size(), e->run_classes, e->run_rows, e->run_units, e->sweep_row_off[ykk::kMaxIdxRows]);
    if (!runs.empty()) {
      TRY(upload(e, e->d_sweep_runs, runs.data(), runs.size(), st));
      e->sweep_runs = (int)runs.size();
      e->sweep_ready = true;
    }
    // FUSED ROWS: what is left to the chunk writers — the signatures with a handful of rows, the asks with a selector of their own — is
    // written by k_fused_rows from RECORDS resolved here (see FuseRec): unpinned classes without a topology signature whose rows are
    // plain plane rows, in row order. They need a full pass with NodeResourcesFit and NodeAffinity evaluated and counts added by the
    // writers (the run kernels' pass conditions).
    e->fuse_count = e->fuse_row_count = 0;
    e->fuse_ready = false;
    e->h_class_fused.assign((size_t)C, 0);
    if (e->fuse_rows != 0 && e->fam_aff.D > 0 && !e->h_res_rows.empty()) {
      std::vector<int32_t> users((size_t)e->fam_aff.D, 0);
      for (int c = 0; c < C; ++c)
        if (key(c, 2) >= 0 && key(c, 2) < e->fam_aff.D) users[(size_t)key(c, 2)]++;
      std::vector<ykk::FuseRec> recs;
      for (int32_t c : order_b) {
        const int sa = key(c, 2), sr = key(c, 0), stl = key(c, 1);
        if (!tail[(size_t)c] || e->h_class_sweep[(size_t)c] || class_size[(size_t)c] < 1 || class_pin[(size_t)c] != -1 || key(c, 3) >= 0) continue;
        if (sa < 0 || sa >= e->fam_aff.D || sr < 0 || (size_t)(sr + 1) * (size_t)R1 > e->h_res_rows.size()) continue;
        ykk::FuseRec rec{};
        rec.cls = c, rec.dest = first_row_of[(size_t)c], rec.len = class_size[(size_t)c];
        rec.row[rec.n++] = (2 << ykk::kFuseFamShift) | sa;  // (family-relative: the families' first rows are known per pass, ensure_planes)
        if (stl >= 0) rec.row[rec.n++] = (1 << ykk::kFuseFamShift) | stl;
        bool ok = true;
        for (int k = 0; k < R1 && ok; ++k) {
          const int r = e->h_res_rows[(size_t)sr * (size_t)R1 + (size_t)k];
          if (r < 0) continue;
          if ((r >> ykk::kRowBigShift) || rec.n >= 8) ok = false;
          else rec.row[rec.n++] = r;
        }
        if (!ok) continue;
        recs.push_back(rec);
        e->h_class_fused[(size_t)c] = 1;
        e->fuse_count++;
        e->fuse_row_count += rec.len;
      }
      e->fuse_combos = 0;
      if (e->fuse_count > 0) {
        // Classes that differ only in their node-affinity signature share the AND of their other rows: one COMBINATION per distinct
        // (toleration, request vector) pair, written first by the same kernel; a class is then two rows. Worth it while a
        // combination serves several classes (else the records keep naming their rows).
        std::unordered_map<uint64_t, int32_t> combo_of;
        std::vector<ykk::FuseRec> combo_recs;
        std::vector<int32_t> combo_ix(recs.size(), -1);
        for (size_t i = 0; i < recs.size(); ++i) {
          const int32_t c = recs[i].cls;
          const uint64_t k2 = ((uint64_t)(uint32_t)key(c, 1) << 32) | (uint32_t)key(c, 0);
          auto it = combo_of.find(k2);
          if (it == combo_of.end()) {
            it = combo_of.emplace(k2, (int32_t)combo_recs.size()).first;
            ykk::FuseRec cr{};
            cr.cls = 0, cr.dest = it->second, cr.len = 1;
            for (int k = 1; k < recs[i].n; ++k) cr.row[cr.n++] = recs[i].row[k];
            combo_recs.push_back(cr);
          }
          combo_ix[i] = it->second;
        }
        bool combine = e->fuse_combine != 0 && combo_recs.size() * 4 <= recs.size() && combo_recs.size() <= 16384;
        for (const ykk::FuseRec& cr : combo_recs) combine = combine && cr.n >= 1;  // (a class of the affinity row alone has nothing to combine)
        if (combine) {
          for (size_t i = 0; i < recs.size(); ++i) {
            recs[i].n = 2;
            recs[i].row[1] = (3 << ykk::kFuseFamShift) | combo_ix[i];
          }
          e->fuse_combos = (int)combo_recs.size();
          TRY(upload(e, e->d_fuse_combo_rec, combo_recs.data(), combo_recs.size(), st));
          HIPCHK(e->d_fuse_combo.ensure((size_t)e->fuse_combos * (size_t)e->row_stride * sizeof(u64)));
        }
        TRY(upload(e, e->d_fuse_rec, recs.data(), recs.size(), st));
        e->fuse_ready = true;
      }
      if (getenv("YKPRED_TRACE_RUNS")) {
        // what the chunk writers keep: tail classes by size and by the number of classes that share their affinity signature
        long t_cls = 0, t_rows = 0, t_single = 0, t_single_shared = 0, t_single_big = 0, t_single_spread = 0, t_pinned = 0;
        for (int32_t c : order_b) {
          if (!tail[(size_t)c] || e->h_class_sweep[(size_t)c] || e->h_class_fused[(size_t)c]) continue;
          ++t_cls, t_rows += class_size[(size_t)c];
          if (class_size[(size_t)c] == 1) {
            ++t_single;
            const int sa = key(c, 2);
            if (sa >= 0 && sa < e->fam_aff.D && users[(size_t)sa] != 1) ++t_single_shared;
            if (key(c, 3) >= 0) ++t_single_spread;
            if (class_pin[(size_t)c] != -1) ++t_pinned;
            const int sr = key(c, 0);
            if (sr >= 0)
              for (int k = 0; k < R1; ++k)
                if (e->h_res_rows[(size_t)sr * (size_t)R1 + (size_t)k] > 0 && (e->h_res_rows[(size_t)sr * (size_t)R1 + (size_t)k] >> ykk::kRowBigShift)) { ++t_single_big; break; }
          }
        }
        fprintf(stderr, "fused: %d classes, %d rows, %d combinations; chunk writers keep %ld classes / %ld rows of zone B (%ld single-row: %ld share their affinity signature, %ld with an index row, %ld spread, %ld pinned)\n",
                e->fuse_count, e->fuse_row_count, e->fuse_combos, t_cls, t_rows, t_single, t_single_shared, t_single_big, t_single_spread, t_pinned);
      }
    }
  }
  e->rows_total = next_row;
  e->rows_a = rows_a;
  e->n_bands = n_bands;
  e->n_band_steps = n_steps;
  e->n_classes_a = (int)class_list_a.size();
  e->n_fix_rows = (int)fix_row.size();
  if (e->row_capacity && e->rows_total > e->row_capacity)
    return fail(e, YKPRED_E_INVALID, "the bitmap needs more rows than ykpred_set_row_capacity allows");
  // Chunks of <= chunk_members members of one class with their explicit row lists. The full pass runs them for zone B only
  // (zone A is written by the band kernel); the incremental "dirty classes" pass runs them for any class.
  const int cm = e->chunk_members;
  struct Chunk {
    int32_t cls, begin, len, first, first_row;
  };
  std::vector<Chunk> chunks;
  for (int c = 0; c < C; ++c)
    for (int b = class_off[(size_t)c]; b < class_off[(size_t)c + 1]; b += cm)
      chunks.push_back({c, b, std::min(cm, class_off[(size_t)c + 1] - b), b == class_off[(size_t)c] ? 1 : 0, e->h_pod_row[(size_t)members[(size_t)b]]});
  if (e->chunk_sorted)
    std::stable_sort(chunks.begin(), chunks.end(), [](const Chunk& x, const Chunk& y) { return x.first_row < y.first_row; });
  auto &ch_class = e->h_ch_class, &ch_begin = e->h_ch_begin, &ch_len = e->h_ch_len, &ch_first = e->h_ch_first;
  ch_class.clear();
  ch_begin.clear();
  ch_len.clear();
  ch_first.clear();
  e->h_ch_zone.clear();
  e->h_class_chunks.assign((size_t)C, {});
  for (const Chunk& k : chunks) {
    e->h_class_chunks[(size_t)k.cls].push_back((int32_t)ch_class.size());
    ch_class.push_back(k.cls);
    ch_begin.push_back(k.begin);
    ch_len.push_back(k.len);
    ch_first.push_back(k.first);
    e->h_ch_zone.push_back(e->h_class_slot_a[(size_t)k.cls] >= 0 ? 1 : (e->h_class_sweep[(size_t)k.cls] ? 2 : (e->h_class_fused[(size_t)k.cls] ? 3 : 0)));
  }
  // the zone-B chunks by number: the full pass launches the class-by-class writer over this list only
  std::vector<int32_t> chunk_list_b;
  for (size_t k = 0; k < e->h_ch_zone.size(); ++k)
    if (e->h_ch_zone[k] != 1) chunk_list_b.push_back((int32_t)k);  // (the chunks of sweep runs too: a pass without the sweep writes them here)
  e->NCB = (int)chunk_list_b.size();
  {
    std::vector<int32_t> chunk_list_b0;
    for (size_t k = 0; k < e->h_ch_zone.size(); ++k)
      if (e->h_ch_zone[k] == 0) chunk_list_b0.push_back((int32_t)k);
    e->NCB0 = (int)chunk_list_b0.size();
    TRY(upload(e, e->d_chunk_list_b0, chunk_list_b0.data(), chunk_list_b0.size(), st));
  }
  e->h_class_first.assign((size_t)C, -1);
  for (int c = 0; c < C; ++c) e->h_class_first[(size_t)c] = members[(size_t)class_off[(size_t)c]];
  std::vector<int32_t> member_rows((size_t)P);
  for (int i = 0; i < P; ++i) member_rows[(size_t)i] = e->h_pod_row[(size_t)members[(size_t)i]];
  e->h_row_stale.assign((size_t)P, 0);
  e->C = C;
  e->NC = (int)ch_class.size();
  e->patch_chunks = 0;
  TRY(upload(e, e->d_pod_class, pod_class.data(), pod_class.size(), st));
  TRY(upload(e, e->d_class_sig, class_sig.data(), class_sig.size(), st));
  TRY(upload(e, e->d_class_pin, class_pin.data(), class_pin.size(), st));
  TRY(upload(e, e->d_class_first, e->h_class_first.data(), e->h_class_first.size(), st));
  TRY(upload(e, e->d_members, member_rows.data(), member_rows.size(), st));
  TRY(upload(e, e->d_pod_row, e->h_pod_row.data(), e->h_pod_row.size(), st));
  TRY(upload(e, e->d_chunk_zone, e->h_ch_zone.data(), e->h_ch_zone.size(), st));
  TRY(upload(e, e->d_chunk_list_b, chunk_list_b.data(), chunk_list_b.size(), st));
  TRY(upload(e, e->d_band_tab, band_tab.data(), band_tab.size(), st));
  TRY(upload(e, e->d_class_list_a, class_list_a.data(), class_list_a.size(), st));
  TRY(upload(e, e->d_class_slot_a, e->h_class_slot_a.data(), e->h_class_slot_a.size(), st));
  TRY(upload(e, e->d_fix_row, fix_row.data(), fix_row.size(), st));
  TRY(upload(e, e->d_fix_slot, fix_slot.data(), fix_slot.size(), st));
  HIPCHK(e->d_class_rows_a.ensure((size_t)std::max<size_t>(class_list_a.size(), 1) * (size_t)e->row_stride * sizeof(u64)));
  TRY(upload(e, e->d_chunk_class, ch_class.data(), ch_class.size(), st));
  TRY(upload(e, e->d_chunk_begin, ch_begin.data(), ch_begin.size(), st));
  TRY(upload(e, e->d_chunk_len, ch_len.data(), ch_len.size(), st));
  TRY(upload(e, e->d_chunk_first, ch_first.data(), ch_first.size(), st));
  HIPCHK(e->d_class_count.ensure((size_t)std::max(C, 1) * sizeof(int)));
  HIPCHK(e->d_class_best.ensure((size_t)std::max(C, 1) * sizeof(int)));
  HIPCHK(hipStreamSynchronize(st));  // the uploads read pageable host vectors
  e->classes_dirty = false;
  e->layout_version++;
  e->last_eval_valid = false;
  return YKPRED_OK;
}

// Plane buffers: one allocation per node order, families are consecutive row ranges. (Re)allocated buffers are zeroed
// ON THE LAUNCH STREAM: the engine's streams are non-blocking, so a null-stream hipMemset would not be ordered
// against the kernels that follow. The zero fill matters for the padding words [row_words, row_stride).
int ensure_planes(ykpred_engine* e, hipStream_t st) {
  int rows = 0;
  for (Family* f : {&e->fam_res, &e->fam_spread, &e->fam_tol, &e->fam_aff}) {  // ballot families first (see k_permute_planes)
    f->base = rows;
    rows += std::max(f->D, 1);
  }
  // (+ 64 words: k_walk_rows reads whole 64-word groups, the lanes past the end of the LAST row stay inside the buffer)
  size_t need = ((size_t)rows * (size_t)e->row_stride + 64) * sizeof(u64);
  for (DevBuf* b : {&e->planes_canon, &e->planes_ranked}) {
    if (b->cap < need || e->plane_rows_alloc != rows) {
      e->tables_version++;
      HIPCHK(b->ensure(need));
      HIPCHK(hipMemsetAsync(b->p, 0, need, st));
    }
  }
  e->plane_rows_alloc = rows;
  size_t base_need = (size_t)(64 * (e->W + e->KT + e->KP) + 2) * (size_t)e->row_stride * sizeof(u64);
  for (DevBuf* b : {&e->base_canon, &e->base_ranked}) {
    if (b->cap < base_need) {
      e->tables_version++;
      HIPCHK(b->ensure(base_need));
      HIPCHK(hipMemsetAsync(b->p, 0, base_need, st));
    }
  }
  return YKPRED_OK;
}

// Device tables of the PodTopologySpread signatures: constraint rows with their histogram cell ranges (which depend on
// the per-key domain counts of the current node table).
int build_spread_tables(ykpred_engine* e, hipStream_t st) {
  e->tables_version++;
  std::vector<int32_t> coff{0}, aff, tol;
  std::vector<ykk::SpreadC> rows;
  int64_t cells = 0;
  for (size_t d = 0; d < e->spread_sig.size(); ++d) {
    for (const ykpred_spread_t& c : e->spread_sig[d]) {
      if (c.topology_key < 0 || c.topology_key >= e->KD || c.selector_class >= e->KS)
        return fail(e, YKPRED_E_INVALID, "spread constraint references a topology key / selector class outside the node table");
      ykk::SpreadC r;
      r.kd = c.topology_key;
      r.ks = c.selector_class;
      r.max_skew = c.max_skew;
      r.min_domains = c.min_domains;
      r.self_match = c.self_match;
      r.flags = c.flags;
      r.kind = c.kind;
      r.dom_size = (size_t)c.topology_key < e->h_domain_sizes.size() ? e->h_domain_sizes[(size_t)c.topology_key] : 0;
      r.cnt_off = (int)cells;
      cells += r.dom_size;
      if (cells > 0x7fffffff) return fail(e, YKPRED_E_UNSUPPORTED, "spread histograms exceed 2^31 cells");
      rows.push_back(r);
    }
    coff.push_back((int32_t)rows.size());
    aff.push_back(e->spread_sig_aff[d]);
    tol.push_back(e->spread_sig_tol[d]);
  }
  e->spread_constraints = (int)rows.size();
  e->spread_cells = cells;
  TRY(upload(e, e->d_sp_coff, coff.data(), coff.size(), st));
  TRY(upload(e, e->d_sp_c, rows.data(), rows.size(), st));
  TRY(upload(e, e->d_sp_aff, aff.data(), aff.size(), st));
  TRY(upload(e, e->d_sp_tol, tol.data(), tol.size(), st));
  HIPCHK(e->d_sp_cnt.ensure((size_t)std::max<int64_t>(cells, 1) * sizeof(int)));
  HIPCHK(e->d_sp_present.ensure((size_t)std::max<int64_t>(cells, 1) * sizeof(int)));
  HIPCHK(e->d_sp_min.ensure((size_t)std::max(e->spread_constraints, 1) * sizeof(int)));
  HIPCHK(hipStreamSynchronize(st));
  e->spread_dirty = false;
  return YKPRED_OK;
}

struct Timer {
  ykpred_engine* e;
  bool on;
  void start(hipStream_t st) {
    e->timed = 0;
    e->timing_valid = false;
    if (on) (void)hipEventRecord(e->ev[2 * YKPRED_MAX_TIMED_KERNELS], st);
  }
  // bracket one kernel: begin() before the launch, end() after it, both on the launch stream
  void begin(hipStream_t st) {
    if (on && e->timed < YKPRED_MAX_TIMED_KERNELS) (void)hipEventRecord(e->ev[2 * e->timed], st);
  }
  void end(hipStream_t st, const char* name) {
    if (trace_kernels_on()) {
      const hipError_t s = hipStreamSynchronize(st);
      fprintf(stderr, "ykpred: %s %s\n", name, s == hipSuccess ? "done" : hipGetErrorString(s));
      fflush(stderr);
    }
    if (!on || e->timed >= YKPRED_MAX_TIMED_KERNELS) return;
    (void)hipEventRecord(e->ev[2 * e->timed + 1], st);
    e->timed_name[e->timed] = name;
    e->timed++;
  }
  void done(hipStream_t st) {
    if (on) (void)hipEventRecord(e->ev[2 * YKPRED_MAX_TIMED_KERNELS + 1], st);
    e->timing_valid = on;
  }
};

// Engine-owned outputs of the last evaluation grow with the ask table and keep their contents (caller-owned ones must
// already hold layout.num_pods rows): the incremental kernels address rows up to the CURRENT table length.
int grow_owned_outputs(ykpred_engine* e) {
  const size_t P = (size_t)std::max(e->P, 1);
  if (e->last_bitmap && e->last_bitmap == e->d_bitmap.p) {
    HIPCHK(e->d_bitmap.reserve_keep((size_t)std::max(std::max(e->rows_total, e->row_capacity), 1) * (size_t)e->row_stride * sizeof(u64), e->d_bitmap.cap));
    e->last_bitmap = e->d_bitmap.p;
  }
  if (e->last_counts && e->last_counts == e->d_counts.p) {
    HIPCHK(e->d_counts.reserve_keep(P * sizeof(int), e->d_counts.cap));
    e->last_counts = e->d_counts.p;
  }
  if (e->last_decisions && e->last_decisions == e->d_decisions.p) {
    HIPCHK(e->d_decisions.reserve_keep(P * sizeof(int), e->d_decisions.cap));
    e->last_decisions = e->d_decisions.p;
  }
  if (e->last_keys && e->last_keys == e->d_keys.p) {
    HIPCHK(e->d_keys.reserve_keep(P * sizeof(i64), e->d_keys.cap));
    e->last_keys = e->d_keys.p;
  }
  return YKPRED_OK;
}

// PodTopologySpread PreFilter for every spread signature: histogram over nodes, then the per-constraint minimum.
// count_only / counts_ready split the two halves for node-sharded clusters (all-reduce of the histograms in between).
int run_spread_prefilter(ykpred_engine* e, hipStream_t st, Timer* tm, bool do_count, bool do_min) {
  if (e->spread_dirty) TRY(build_spread_tables(e, st));
  if (e->fam_spread.D == 0 || e->N == 0) return YKPRED_OK;
  ykk::NodeTable nt = node_table(e);
  ykk::SpreadSigs sp = spread_sigs(e);
  if (do_count) {
    HIPCHK(hipMemsetAsync(e->d_sp_cnt.p, 0, (size_t)e->spread_cells * sizeof(int), st));
    HIPCHK(hipMemsetAsync(e->d_sp_present.p, 0, (size_t)e->spread_cells * sizeof(int), st));
    ykk::AffSigs as{e->d_sig_aff_flags.as<unsigned>(), e->d_sig_aff_off.as<int>(), e->d_sig_aff_terms.as<u64>(), e->d_sig_pre_off.as<int>(),
                    e->d_sig_pre_terms.as<u64>()};
    if (tm) tm->begin(st);
    hipLaunchKernelGGL(ykk::k_spread_count, dim3((unsigned)e->fam_spread.D, (unsigned)((e->N + ykk::kBlock - 1) / ykk::kBlock)),
                       dim3(ykk::kBlock), 0, st, nt, sp, as, e->d_sig_tol.as<u64>());
    if (tm) tm->end(st, "k_spread_count");
  }
  if (do_min) {
    if (tm) tm->begin(st);
    hipLaunchKernelGGL(ykk::k_spread_min, dim3((unsigned)e->spread_constraints), dim3(ykk::kWave), 0, st, sp, e->spread_constraints);
    if (tm) tm->end(st, "k_spread_min");
  }
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
}

// roctx ranges around upload / eval / gather: visible in rocprofv3 --marker-trace timelines. The marker
// library is only used when it is already mapped (the process runs under a profiler) or YKPRED_ROCTX=1 asks for it.
struct Roctx {
  int (*push)(const char*) = nullptr;
  int (*pop)() = nullptr;
};
Roctx* roctx() {
  static Roctx r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* lib = nullptr;
    const char* names[] = {"librocprofiler-sdk-roctx.so.1", "libroctx64.so.4", "libroctx64.so"};
    for (const char* n : names)
      if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD);
    const char* want = getenv("YKPRED_ROCTX");
    if (!lib && want && want[0] == '1')
      for (const char* n : names)
        if (!lib) lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    if (!lib) return;
    r.push = (int (*)(const char*))dlsym(lib, "roctxRangePushA");
    r.pop = (int (*)())dlsym(lib, "roctxRangePop");
    if (!r.push || !r.pop) r.push = nullptr;
  });
  return &r;
}
struct Range {
  bool on;
  explicit Range(const char* name) : on(roctx()->push != nullptr) {
    if (on) roctx()->push(name);
  }
  ~Range() {
    if (on) roctx()->pop();
  }
};

// librccl, resolved on first use. torch ships a librccl.so.1 of its own: a copy that is already mapped is reused.
struct Rccl {
  void* lib = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string error;
};
std::string& rccl_library_override() {
  static std::string path;  // ykpred_comm_use_library
  return path;
}
Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    if (!rccl_library_override().empty()) {
      r.lib = dlopen(rccl_library_override().c_str(), RTLD_NOW | RTLD_LOCAL);
      if (!r.lib) {
        r.error = std::string("cannot load ") + rccl_library_override() + ": " + dlerror();
        return;
      }
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.lib) break;
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
    }
    for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      if (r.lib) break;
      r.lib = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!r.lib) {
      r.error = std::string("cannot load librccl: ") + dlerror();
      return;
    }
    auto sym = [&](const char* n) {
      void* p = dlsym(r.lib, n);
      if (!p && r.error.empty()) r.error = std::string("librccl lacks ") + n;
      return p;
    };
    r.GetUniqueId = (decltype(r.GetUniqueId))sym("ncclGetUniqueId");
    r.CommInitRank = (decltype(r.CommInitRank))sym("ncclCommInitRank");
    r.CommDestroy = (decltype(r.CommDestroy))sym("ncclCommDestroy");
    r.AllGather = (decltype(r.AllGather))sym("ncclAllGather");
    r.AllReduce = (decltype(r.AllReduce))sym("ncclAllReduce");
    r.GetErrorString = (decltype(r.GetErrorString))sym("ncclGetErrorString");
  });
  return &r;
}
#define NCCLCHK(call)                                                                                     \
  do {                                                                                                    \
    ncclResult_t _s = (call);                                                                             \
    if (_s != ncclSuccess) return fail(e, YKPRED_E_DEVICE, std::string(#call) + ": " + rccl()->GetErrorString(_s)); \
  } while (0)

// Cluster-wide PreFilter state of the topology plugins on a node-sharded cluster: matches per (constraint, domain) add up,
// "an eligible node carries the domain" is an OR (MAX of 0/1). Two small all-reduces (KBs) between count and min.
int allreduce_spread(ykpred_engine* e, hipStream_t st) {
  if (!e->comm || e->comm_world <= 1 || e->spread_cells == 0) return YKPRED_OK;
  NCCLCHK(rccl()->AllReduce(e->d_sp_cnt.p, e->d_sp_cnt.p, (size_t)e->spread_cells, ncclInt32, ncclSum, e->comm, st));
  NCCLCHK(rccl()->AllReduce(e->d_sp_present.p, e->d_sp_present.p, (size_t)e->spread_cells, ncclInt32, ncclMax, e->comm, st));
  return YKPRED_OK;
}

// The topology histograms for the per-pair entry points (query / preemption): reuse what the last full pass left when the
// tables have not moved since — on a sharded engine those are the cluster-wide sums, which a local rebuild would destroy.
int ensure_histograms(ykpred_engine* e, hipStream_t st) {
  if (e->spread_dirty) TRY(build_spread_tables(e, st));
  if (e->fam_spread.D == 0 || e->hist_epoch == e->nodes_epoch) return YKPRED_OK;
  if (e->comm && e->comm_world > 1)
    return fail(e, YKPRED_E_STATE, "topology histograms are stale on a sharded engine: run ykpred_eval on every shard first (it sums them across shards)");
  TRY(run_spread_prefilter(e, st, nullptr, true, true));
  e->hist_epoch = e->nodes_epoch;
  return YKPRED_OK;
}

// The rank-ordered planes as the decision kernels address them (ykpred_eval's decision branch; ykpred_allocate_round re-reads
// what that branch left in the buffers).
// The rank-ordered windows of EVERY index row (k_dim_walk_window) from the sorted lists and the running maxima the last decision pass
// left: a pass whose sweep classes got their decisions from k_run_decide wrote only the windows k_decide read; a round's class
// descriptors read them all.
int complete_windows(ykpred_engine* e, hipStream_t st) {
  if (!e->win_partial || e->n_big == 0 || e->walk_chunks == 0) return YKPRED_OK;
  ykk::DimWalk dw{e->d_dim_val.as<i64>(), e->d_dim_order.as<int>(), e->d_big_dim.as<int>(), e->d_walk_big.as<int>(), e->d_walk_begin.as<int>(),
                  e->d_walk_len.as<int>(), e->d_sfree_r.as<i64>(), e->d_pmask_r.as<u64>(), e->n_big, e->walk_chunks, e->row_words, nullptr,
                  nullptr, e->d_sorted.as<i64>(), e->d_sorted_off.as<int>(), nullptr};
  const unsigned gy = (unsigned)((e->row_words + ykk::kBlock * ykk::kWalkWords - 1) / (ykk::kBlock * ykk::kWalkWords));
  hipLaunchKernelGGL(ykk::k_dim_walk_window<ykk::kWalkBatch>, dim3((unsigned)e->walk_chunks, gy), dim3(ykk::kBlock), 0, st, dw, e->d_pfx_r.as<i64>(), e->d_win_r.as<unsigned char>());
  HIPCHK(hipGetLastError());
  e->win_partial = false;
  return YKPRED_OK;
}

ykk::Planes ranked_planes_of(const ykpred_engine* e, unsigned pre, unsigned filt, bool spread_on, bool with_first) {
  const bool res_on = filt & YKPRED_PLUGIN_NODE_RESOURCES_FIT, aff_on = (filt | pre) & YKPRED_PLUGIN_NODE_AFFINITY;
  const int fit_error = (pre & YKPRED_PLUGIN_NODE_RESOURCES_FIT) ? 0 : 1;
  auto ranked_of = [&](const Family& f) { return e->planes_ranked.as<u64>() + (size_t)f.base * e->row_stride; };
  int* first_r = with_first ? e->d_first_r.as<int>() : nullptr;
  ykk::Planes pr{res_on ? ranked_of(e->fam_res) : nullptr, ranked_of(e->fam_tol), aff_on ? ranked_of(e->fam_aff) : nullptr,
                 spread_on ? ranked_of(e->fam_spread) : nullptr, e->row_stride, e->d_res_rows.as<int>(), 1 + e->R, nullptr,
                 e->idx_stride, e->d_pmask_r.as<u64>(), e->row_words, first_r, e->fam_res.base, e->fam_tol.base, e->fam_aff.base,
                 e->fam_spread.base, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  pr.n_big = res_on ? e->n_big : 0;
  if (res_on && e->n_big > 0) {
    // index rows in rank order: a 64-byte window per row + the words' sorted free lists for everything outside it
    pr.res_win = e->d_win_r.as<unsigned char>();
    pr.sfree = e->d_sfree_r.as<i64>();
    pr.res_val = e->d_dim_val.as<i64>();
    if (!fit_error) pr.pfx = e->d_pfx_r.as<i64>();  // (Filter without PreFilter state: every mask table is empty, no window is consulted)
  }
  return pr;
}

}  // namespace

// null handles fall through to the entry point's own argument check
#define YK_SERIALISE(eng) \
  std::unique_lock<std::recursive_mutex> engine_lock; \
  if (eng) engine_lock = std::unique_lock<std::recursive_mutex>((eng)->mu)

extern "C" {

int32_t ykpred_abi_version(void) { return YKPRED_ABI_VERSION; }

const char* ykpred_last_error(const ykpred_engine_t* e) { return e ? e->err.c_str() : g_create_error.c_str(); }

int32_t ykpred_create(const ykpred_config_t* cfg, ykpred_engine_t** out) {
  if (!cfg || !out) {
    g_create_error = "null argument";
    return YKPRED_E_INVALID;
  }
  if (cfg->abi_version != YKPRED_ABI_VERSION) {
    g_create_error = "ABI version mismatch";
    return YKPRED_E_INVALID;
  }
  if (cfg->num_resources < 3 || cfg->num_resources > ykk::kMaxR || cfg->taint_words < 1 || cfg->taint_words > ykk::kMaxKT ||
      cfg->label_words < 1 || cfg->label_words > ykk::kMaxWTotal) {
    g_create_error = "config out of range: need 3<=R<=8, 1<=KT<=4, 1<=W<=32";
    return YKPRED_E_UNSUPPORTED;
  }
  if (cfg->topology_keys < 0 || cfg->topology_keys > ykk::kMaxKD || cfg->selector_classes < 0 || cfg->selector_classes > 4096 ||
      cfg->port_words < 0 || cfg->port_words > ykk::kMaxKP) {
    g_create_error = "config out of range: need 0<=KD<=8 topology keys, 0<=KS<=4096 selector classes, 0<=KP<=4 port words";
    return YKPRED_E_UNSUPPORTED;
  }
  int ndev = 0;
  hipError_t s = hipGetDeviceCount(&ndev);
  if (s != hipSuccess || ndev <= 0) {
    g_create_error = std::string("no HIP device: ") + hipGetErrorString(s) + " (libykpred has no CPU fallback)";
    return YKPRED_E_DEVICE;
  }
  if (cfg->device < 0 || cfg->device >= ndev) {
    g_create_error = "device ordinal out of range";
    return YKPRED_E_INVALID;
  }
  s = hipSetDevice(cfg->device);
  if (s != hipSuccess) {
    g_create_error = std::string("hipSetDevice: ") + hipGetErrorString(s);
    return YKPRED_E_DEVICE;
  }
  auto* e = new ykpred_engine();
  e->cfg = *cfg;
  e->R = cfg->num_resources;
  e->KT = cfg->taint_words;
  e->W = cfg->label_words;
  e->KD = cfg->topology_keys;
  e->KS = cfg->selector_classes;
  e->KP = cfg->port_words;
  if (cfg->reserved[0] >= 1 && cfg->reserved[0] <= ykk::kChunkMembers) e->chunk_members = cfg->reserved[0];
  e->chunk_sorted = cfg->reserved[1] != 1;
  if (cfg->reserved[2] > 0 && cfg->reserved[2] <= 160 * 1024) e->combine_lds_bytes = cfg->reserved[2];
  if (cfg->reserved[2] < 0) e->combine_lds_bytes = 0;
  if (cfg->reserved[4] > 0) e->walk_rows = cfg->reserved[4];
  if (cfg->reserved[5] > 0) e->wave_combine_below = cfg->reserved[5];
  if (cfg->reserved[5] < 0) e->wave_combine_below = 0;
  if (cfg->reserved[6] < 0) e->bands_enabled = false;
  // the band kernel packs the step-in-band into 8 bits of its class key: at most 256 windows per band
  if (cfg->reserved[6] > 0) e->band_steps = std::min(256, std::max(4, (cfg->reserved[6] + 3) / 4 * 4));
  e->graph_disabled = cfg->reserved[3] != 1;  // tunable: replay a repeated pass as a hipGraph (measured: no gain)
  {
    int lds = 0;  // what a workgroup may ask for with hipFuncAttributeMaxDynamicSharedMemorySize (64 KiB on gfx90a / gfx942, 160 KiB on gfx950)
    if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, cfg->device) == hipSuccess && lds > 0) e->max_lds_bytes = lds;
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, cfg->device) == hipSuccess && cus > 0) e->num_cus = cus;
  }
  // YKPRED_TUNE="key=value,key=value": the ONE environment hook for engine tunables — what the tests use to force a path (index rows
  // on tiny clusters, a k_sig_planes width, the sub-wave decision kernel ...). Unknown keys are an error, not ignored.
  if (const char* tune = getenv("YKPRED_TUNE")) {
    std::string text(tune);
    size_t at = 0;
    while (at < text.size()) {
      size_t stop = text.find(',', at);
      if (stop == std::string::npos) stop = text.size();
      const std::string item = text.substr(at, stop - at);
      at = stop + 1;
      if (item.empty()) continue;
      const size_t eq = item.find('=');
      const std::string key = item.substr(0, eq);
      const int val = eq == std::string::npos ? 1 : atoi(item.c_str() + eq + 1);
      if (key == "walk_rows") e->walk_rows = std::max(val, 1);
      else if (key == "sig_wpl") e->sig_wpl = val;
      else if (key == "combine_slices") e->combine_slices = val;
      else if (key == "decide_groups_from") e->decide_groups_from = val;
      else if (key == "graph") e->graph_disabled = val != 1;
      else if (key == "early_counts") e->early_counts = val;
      else if (key == "band_steps") {
        e->bands_enabled = val >= 0;
        e->band_steps = val > 0 ? std::min(256, std::max(4, (val + 3) / 4 * 4)) : 0;
      } else if (key == "chunk_members") e->chunk_members = std::min(std::max(val, 1), (int)ykk::kChunkMembers);
      else if (key == "wave_combine_below") e->wave_combine_below = std::max(val, 0);
      else if (key == "fail_after") e->fail_after = std::max(val, 0);
      else if (key == "round_prof") e->round_prof = val;
      else if (key == "round_batched") e->round_batched = val;
      else if (key == "round_node_assume") e->round_node_assume = val;
      else if (key == "sweep_min_run") e->sweep_min_run = std::max(val, 0);
      else if (key == "sweep_groups") e->sweep_groups = std::max(val, 0);
      else if (key == "class_runs") e->class_runs = val;
      else if (key == "run_decide") e->run_decide = val;
      else if (key == "fuse_rows") e->fuse_rows = val;
      else if (key == "fuse_wpl") e->fuse_wpl = val;
      else if (key == "fuse_combine") e->fuse_combine = val;
      else if (key == "class_runs_min_rows") e->class_runs_min_rows = std::max(val, 1);
      else {
        g_create_error = "YKPRED_TUNE: unknown key '" + key + "'";
        delete e;
        return YKPRED_E_INVALID;
      }
    }
  }
  s = hipStreamCreateWithFlags(&e->own_stream, hipStreamNonBlocking);
  if (s != hipSuccess) {
    g_create_error = std::string("hipStreamCreate: ") + hipGetErrorString(s);
    delete e;
    return YKPRED_E_DEVICE;
  }
  {
    int lo = 0, hi = 0;  // numerically lower = higher priority
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);
    // the decision stream gets the greatest priority (measured: no effect either way on any population)
    if (hipStreamCreateWithPriority(&e->aux_stream, hipStreamNonBlocking, hi) != hipSuccess)
      (void)hipStreamCreateWithFlags(&e->aux_stream, hipStreamNonBlocking);
  }
  if (hipStreamCreateWithFlags(&e->peek_stream, hipStreamNonBlocking) != hipSuccess) e->peek_stream = nullptr;
  (void)hipEventCreateWithFlags(&e->ev_eval_done, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&e->ev_join, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&e->ev_planes, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&e->ev_counts, hipEventDisableTiming);
  (void)hipEventCreateWithFlags(&e->ev_zero, hipEventDisableTiming);
  for (auto& ev : e->ev) (void)hipEventCreate(&ev);
  e->ev_ready = true;
  *out = e;
  return YKPRED_OK;
}

void ykpred_destroy(ykpred_engine_t* e) {
  if (!e) return;
  (void)hipSetDevice(e->cfg.device);
  (void)hipDeviceSynchronize();
  if (e->comm) (void)rccl()->CommDestroy(e->comm);
  e->comm = nullptr;
  for (DevBuf* b : {&e->d_gathered, &e->d_gathered_map, &e->d_xkey, &e->d_xcand}) b->release();
  for (auto& g : e->graphs) (void)hipGraphExecDestroy(g.second);
  e->graphs.clear();
  for (DevBuf* b : {&e->d_alloc, &e->d_req, &e->d_allowed, &e->d_count, &e->d_nflags, &e->d_taints, &e->d_labels, &e->d_domain, &e->d_selcount, &e->d_ports, &e->d_sig_ports, &e->d_swanted,
                    &e->d_sp_cnt_prev, &e->d_sp_present_prev, &e->d_sp_min_prev, &e->d_sig_changed, &e->d_class_dirty,
                    &e->d_spec_spread, &e->d_sp_coff, &e->d_sp_c, &e->d_sp_aff, &e->d_sp_tol, &e->d_sp_cnt, &e->d_sp_present, &e->d_sp_min,
                    &e->planes_canon, &e->planes_ranked, &e->base_canon, &e->base_ranked, &e->d_rankbuf, &e->d_score, &e->d_key,
                    &e->d_rank, &e->d_perm, &e->d_sreq, &e->d_stol, &e->d_sflags, &e->d_aff_off, &e->d_aff_terms, &e->d_pre_off,
                    &e->d_pre_terms, &e->d_dim_val, &e->d_dim_order, &e->d_dim_chunk_dim, &e->d_dim_chunk_begin, &e->d_dim_chunk_len,
                    &e->d_res_rows, &e->d_big_dim, &e->d_walk_big, &e->d_walk_begin, &e->d_walk_len, &e->d_sfree_c, &e->d_pmask_c,
                    &e->d_sfree_r, &e->d_pmask_r, &e->d_rbits_c, &e->d_sorted, &e->d_sorted_off, &e->d_ent_c, &e->d_agree, &e->d_fuse_rec, &e->d_fuse_combo_rec, &e->d_fuse_combo, &e->d_run_ranges, &e->d_no_decide, &e->d_glin_r, &e->d_run_classes, &e->d_run_units, &e->d_sweep_rows, &e->d_sweep_runs, &e->d_sweep_units, &e->d_chunk_list_b0, &e->d_walk2_order, &e->d_walk2_big, &e->d_walk2_begin, &e->d_walk2_len, &e->d_idx_c, &e->d_win_r, &e->d_pfx_r, &e->d_slice_desc, &e->d_slice_general, &e->d_chunk_list_b, &e->d_first_r, &e->d_sig_tol, &e->d_sig_tolflags, &e->d_sig_aff_flags, &e->d_sig_aff_off,
                    &e->d_sig_aff_terms, &e->d_sig_pre_off, &e->d_sig_pre_terms, &e->d_pod_spec, &e->d_pod_pin, &e->d_pod_class,
                    &e->d_class_sig, &e->d_class_pin, &e->d_class_first, &e->d_class_word, &e->d_chunk_class, &e->d_chunk_begin, &e->d_chunk_len, &e->d_chunk_first,
                    &e->d_pod_row, &e->d_band_tab, &e->d_class_rows_a, &e->d_class_list_a, &e->d_class_slot_a, &e->d_fix_row, &e->d_fix_slot, &e->d_chunk_zone,
                    &e->d_members, &e->d_patches, &e->d_rows, &e->d_row_count, &e->d_row_best, &e->d_class_count, &e->d_class_best, &e->d_bitmap, &e->d_counts, &e->d_decisions, &e->d_keys, &e->d_scratch,
                    &e->d_member_key, &e->d_name_rank, &e->d_member_tie, &e->d_class_rows_all, &e->d_gathered_classes, &e->d_class_rows_slot,
                    &e->d_class_sig_ident, &e->d_expand_count, &e->d_layout_hash, &e->d_gathered_pod_class, &e->d_row_pod,
                    &e->d_round, &e->d_fx_off, &e->d_fx_cls, &e->d_fx_cnt, &e->d_fx_occ, &e->d_sp_sig_of, &e->d_class_list_b, &e->d_gathered, &e->d_gathered_map,
                    &e->d_xkey, &e->d_xcand})
    b->release();
  if (e->ev_ready)
    for (auto& ev : e->ev) (void)hipEventDestroy(ev);
  if (e->ev_fork) (void)hipEventDestroy(e->ev_fork);
  if (e->ev_join) (void)hipEventDestroy(e->ev_join);
  if (e->ev_planes) (void)hipEventDestroy(e->ev_planes);
  if (e->ev_counts) (void)hipEventDestroy(e->ev_counts);
  if (e->ev_zero) (void)hipEventDestroy(e->ev_zero);
  if (e->h_round_pinned) (void)hipHostFree(e->h_round_pinned);
  if (e->ev_eval_done) (void)hipEventDestroy(e->ev_eval_done);
  if (e->peek_pinned) (void)hipHostFree(e->peek_pinned);
  if (e->peek_stream) (void)hipStreamDestroy(e->peek_stream);
  if (e->aux_stream) (void)hipStreamDestroy(e->aux_stream);
  if (e->own_stream) (void)hipStreamDestroy(e->own_stream);
  delete e;
}

int32_t ykpred_set_nodes(ykpred_engine_t* e, const ykpred_nodes_t* n) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:upload_nodes");
  if (e) e->n_uploads++;
  if (e) e->tables_version++;
  if (e) e->nodes_epoch++;
  if (e) e->ask_epoch++;
  if (e) e->last_eval_valid = false;  // every row and column of an earlier bitmap is stale (and N / row_stride may change)
  if (!e || !n || n->count < 0) return fail(e, YKPRED_E_INVALID, "set_nodes: bad argument");
  if (n->count > 0 && (!n->allocatable || !n->requested || !n->allowed_pods || !n->pod_count || !n->flags || !n->taint_bits ||
                       !n->label_bits))
    return fail(e, YKPRED_E_INVALID, "set_nodes: null column");
  if ((e->KD > 0 && (!n->domain_sizes || (n->count > 0 && !n->domain_id))) || (e->KS > 0 && n->count > 0 && !n->selector_count))
    return fail(e, YKPRED_E_INVALID, "set_nodes: topology / selector columns missing (config has KD/KS > 0)");
  if (e->KP > 0 && n->count > 0 && !n->port_bits) return fail(e, YKPRED_E_INVALID, "set_nodes: port_bits missing (config has KP > 0)");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  e->rank_valid = false;
  const size_t N = (size_t)n->count;
  TRY(upload(e, e->d_alloc, n->allocatable, N * (size_t)e->R, st));
  TRY(upload(e, e->d_req, n->requested, N * (size_t)e->R, st));
  TRY(upload(e, e->d_allowed, n->allowed_pods, N, st));
  TRY(upload(e, e->d_count, n->pod_count, N, st));
  TRY(upload(e, e->d_nflags, n->flags, N, st));
  TRY(upload(e, e->d_taints, n->taint_bits, N * (size_t)e->KT, st));
  TRY(upload(e, e->d_labels, n->label_bits, N * (size_t)e->W, st));
  TRY(upload(e, e->d_domain, n->domain_id, N * (size_t)e->KD, st));
  TRY(upload(e, e->d_selcount, n->selector_count, N * (size_t)e->KS, st));
  TRY(upload(e, e->d_ports, n->port_bits, N * (size_t)e->KP, st));
  e->has_name_rank = n->name_rank != nullptr && N > 0;
  if (e->has_name_rank) TRY(upload(e, e->d_name_rank, n->name_rank, N, st));
  HIPCHK(e->d_member_tie.ensure(std::max<size_t>(N, 1) * sizeof(int)));
  for (int k = 0; k < ykk::kMaxKT; ++k) e->taint_used[k] = 0;
  for (int k = 0; k < e->KT; ++k)
    for (size_t i = 0; i < N; ++i) e->taint_used[k] |= n->taint_bits[(size_t)k * N + i];
  e->h_domain_sizes.assign(n->domain_sizes, n->domain_sizes + e->KD);
  e->spread_dirty = true;
  HIPCHK(e->d_score.ensure(N * sizeof(double)));
  HIPCHK(e->d_key.ensure(N * sizeof(u64)));
  HIPCHK(e->d_rank.ensure(N * sizeof(int)));
  HIPCHK(e->d_perm.ensure(N * sizeof(int)));
  HIPCHK(e->d_rankbuf.ensure((3 * (size_t)ykk::kRankBuckets + 1 + N) * sizeof(int)));
  HIPCHK(e->d_member_key.ensure(N * sizeof(u64)));
  HIPCHK(hipStreamSynchronize(st));
  if (e->N != n->count) {
    // plane rows change length: drop them so ensure_planes() re-zeroes the padding
    e->planes_canon.release();
    e->planes_ranked.release();
    e->base_canon.release();
    e->base_ranked.release();
  }
  e->N = n->count;
  e->row_words = (e->N + 63) / 64;
  // rows start on 128-byte lines (16 words): a line shared by two rows would be written in two partial pieces
  e->row_stride = std::max(16, (e->row_words + 15) / 16 * 16);
  if (e->forced_stride) {
    if (e->forced_stride < e->row_stride) return fail(e, YKPRED_E_INVALID, "set_nodes: ykpred_set_row_stride is smaller than this shard needs");
    e->row_stride = e->forced_stride;
  }
  e->nodes_set = true;
  return YKPRED_OK;
}

int32_t ykpred_update_node(ykpred_engine_t* e, int32_t idx, const ykpred_nodes_t* n) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:update_node");
  if (e) e->n_uploads++;
  if (e) e->tables_version++;
  if (!e || !n || n->count != 1) return fail(e, YKPRED_E_INVALID, "update_node: count must be 1");
  if (!e->nodes_set || idx < 0 || idx >= e->N) return fail(e, YKPRED_E_INVALID, "update_node: index out of range");
  for (int k = 0; k < e->KD; ++k)  // validate before any column is overwritten
    if (n->domain_id[k] >= e->h_domain_sizes[(size_t)k]) return fail(e, YKPRED_E_INVALID, "update_node: new topology domain — re-upload the node table");
  e->nodes_epoch++;
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  e->rank_valid = false;  // the node's score may move
  const size_t N = (size_t)e->N;
  for (int r = 0; r < e->R; ++r) {
    HIPCHK(hipMemcpyAsync(e->d_alloc.as<i64>() + (size_t)r * N + idx, n->allocatable + r, sizeof(i64), hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(e->d_req.as<i64>() + (size_t)r * N + idx, n->requested + r, sizeof(i64), hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(e->d_allowed.as<int>() + idx, n->allowed_pods, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(e->d_count.as<int>() + idx, n->pod_count, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(e->d_nflags.as<unsigned>() + idx, n->flags, sizeof(unsigned), hipMemcpyHostToDevice, st));
  for (int k = 0; k < e->KT; ++k) {
    e->taint_used[k] |= n->taint_bits[k];
    HIPCHK(hipMemcpyAsync(e->d_taints.as<u64>() + (size_t)k * N + idx, n->taint_bits + k, sizeof(u64), hipMemcpyHostToDevice, st));
  }
  for (int w = 0; w < e->W; ++w)
    HIPCHK(hipMemcpyAsync(e->d_labels.as<u64>() + (size_t)w * N + idx, n->label_bits + w, sizeof(u64), hipMemcpyHostToDevice, st));
  for (int k = 0; k < e->KD; ++k) {
    HIPCHK(hipMemcpyAsync(e->d_domain.as<int>() + (size_t)k * N + idx, n->domain_id + k, sizeof(int), hipMemcpyHostToDevice, st));
  }
  for (int k = 0; k < e->KS; ++k)
    HIPCHK(hipMemcpyAsync(e->d_selcount.as<int>() + (size_t)k * N + idx, n->selector_count + k, sizeof(int), hipMemcpyHostToDevice, st));
  for (int k = 0; k < e->KP; ++k)
    HIPCHK(hipMemcpyAsync(e->d_ports.as<u64>() + (size_t)k * N + idx, n->port_bits + k, sizeof(u64), hipMemcpyHostToDevice, st));
  if (e->has_name_rank && n->name_rank)
    HIPCHK(hipMemcpyAsync(e->d_name_rank.as<int>() + idx, n->name_rank, sizeof(int), hipMemcpyHostToDevice, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_update_label_word(ykpred_engine_t* e, int32_t word, const uint64_t* column) {
  YK_SERIALISE(e);
  if (!e || !column || word < 0 || word >= (e ? e->W : 0)) return fail(e, YKPRED_E_INVALID, "update_label_word: bad argument");
  if (!e->nodes_set) return fail(e, YKPRED_E_STATE, "update_label_word: no node table");
  e->tables_version++;
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->N) {
    HIPCHK(hipMemcpyAsync(e->d_labels.as<u64>() + (size_t)word * (size_t)e->N, column, (size_t)e->N * sizeof(u64), hipMemcpyHostToDevice, e->own_stream));
    HIPCHK(hipStreamSynchronize(e->own_stream));
  }
  return YKPRED_OK;
}

int32_t ykpred_set_specs(ykpred_engine_t* e, const ykpred_specs_t* s) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:upload_specs");
  if (e) e->n_uploads++;
  if (e) e->tables_version++;
  if (e) e->ask_epoch++;
  if (e) e->specs_version++;
  if (!e || !s || s->count < 0) return fail(e, YKPRED_E_INVALID, "set_specs: bad argument");
  if (s->count > 0 && (!s->requests || !s->tolerated || !s->flags || !s->aff_term_off || !s->pre_term_off))
    return fail(e, YKPRED_E_INVALID, "set_specs: null column");
  if (s->spread_off && s->count > 0 && s->spread_off[s->count] != 0 && !s->spread)
    return fail(e, YKPRED_E_INVALID, "set_specs: spread_off without spread rows");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  const int S = s->count, R = e->R, KT = e->KT, W = e->W, KP = e->KP;
  if (KP > 0 && S > 0 && !s->wanted_ports) return fail(e, YKPRED_E_INVALID, "set_specs: wanted_ports missing (config has KP > 0)");
  const int T = S ? s->aff_term_off[S] : 0, M = S ? s->pre_term_off[S] : 0;
  if (T < 0 || M < 0 || (T > 0 && !s->aff_terms) || (M > 0 && !s->pre_terms)) return fail(e, YKPRED_E_INVALID, "set_specs: bad term tables");
  // A spec table that only APPENDS to the previous one (the first old-S specs byte-identical) keeps every signature id of
  // the old specs — signatures are numbered in first-use order — so the class index, the last evaluation and the
  // incremental paths stay valid: a new pod template costs one small upload, not a rebuild of 10^6 pod classes.
  const int oldS = e->S, old_spread_D = e->fam_spread.D;
  bool append_only = e->specs_set && !e->classes_dirty && oldS <= S;
  if (append_only && oldS > 0) {
    auto same = [](const void* a, const void* b, size_t n) { return n == 0 || memcmp(a, b, n) == 0; };
    const int oT = e->h_aff_off[(size_t)oldS], oM = e->h_pre_off[(size_t)oldS];
    append_only = same(e->h_req.data(), s->requests, (size_t)oldS * R * sizeof(i64)) &&
                  same(e->h_tol.data(), s->tolerated, (size_t)oldS * KT * sizeof(u64)) &&
                  (KP == 0 || same(e->h_wanted.data(), s->wanted_ports, (size_t)oldS * KP * sizeof(u64))) &&
                  same(e->h_sflags.data(), s->flags, (size_t)oldS * sizeof(uint32_t)) &&
                  same(e->h_aff_off.data(), s->aff_term_off, (size_t)(oldS + 1) * sizeof(int32_t)) &&
                  same(e->h_pre_off.data(), s->pre_term_off, (size_t)(oldS + 1) * sizeof(int32_t)) &&
                  same(e->h_aff_terms.data(), s->aff_terms, (size_t)oT * W * sizeof(u64)) &&
                  same(e->h_pre_terms.data(), s->pre_terms, (size_t)oM * W * sizeof(u64));
    const bool had_spread = !e->h_spread_off.empty() && e->h_spread_off[(size_t)oldS] > 0;
    if (append_only && (had_spread || s->spread_off)) {
      if (e->h_spread_off.empty() || !s->spread_off) {
        append_only = !had_spread && s->spread_off && s->spread_off[oldS] == 0;
      } else {
        append_only = same(e->h_spread_off.data(), s->spread_off, (size_t)(oldS + 1) * sizeof(int32_t)) &&
                      same(e->h_spread.data(), s->spread, (size_t)e->h_spread_off[(size_t)oldS] * sizeof(ykpred_spread_t));
      }
    }
  }
  if (s->spread_off && S > 0) {
    e->h_spread_off.assign(s->spread_off, s->spread_off + S + 1);
    e->h_spread.assign(s->spread, s->spread + s->spread_off[S]);
  } else {
    e->h_spread_off.assign((size_t)S + 1, 0);
    e->h_spread.clear();
  }
  e->S = S;
  e->h_req.assign(s->requests, s->requests + (size_t)S * R);
  e->h_tol.assign(s->tolerated, s->tolerated + (size_t)S * KT);
  if (KP > 0) e->h_wanted.assign(s->wanted_ports, s->wanted_ports + (size_t)S * KP); else e->h_wanted.clear();
  TRY(upload(e, e->d_swanted, e->h_wanted.data(), e->h_wanted.size(), st));
  e->h_sflags.assign(s->flags, s->flags + S);
  e->h_aff_off.assign(s->aff_term_off, s->aff_term_off + S + (S ? 1 : 0));
  e->h_pre_off.assign(s->pre_term_off, s->pre_term_off + S + (S ? 1 : 0));
  if (!S) {
    e->h_aff_off.assign(1, 0);
    e->h_pre_off.assign(1, 0);
  }
  e->h_aff_terms.assign(s->aff_terms, s->aff_terms + (size_t)T * W);
  e->h_pre_terms.assign(s->pre_terms, s->pre_terms + (size_t)M * W);
  TRY(upload(e, e->d_sreq, e->h_req.data(), e->h_req.size(), st));
  TRY(upload(e, e->d_stol, e->h_tol.data(), e->h_tol.size(), st));
  TRY(upload(e, e->d_sflags, e->h_sflags.data(), e->h_sflags.size(), st));
  TRY(upload(e, e->d_aff_off, e->h_aff_off.data(), e->h_aff_off.size(), st));
  TRY(upload(e, e->d_pre_off, e->h_pre_off.data(), e->h_pre_off.size(), st));
  TRY(upload(e, e->d_aff_terms, e->h_aff_terms.data(), e->h_aff_terms.size(), st));
  TRY(upload(e, e->d_pre_terms, e->h_pre_terms.data(), e->h_pre_terms.size(), st));

  // ---- per-plugin signatures: identical byte strings share one plane
  e->spec_sig_res.assign((size_t)S, 0);
  e->spec_sig_tol.assign((size_t)S, 0);
  e->spec_sig_aff.assign((size_t)S, 0);
  std::unordered_map<std::string, int32_t> m_tol, m_aff;
  // Request vectors → signature and request value → plane row (per dimension) are looked up once per spec and dimension, and the
  // adversarial population has 10^6 distinct ones: open-addressing tables of int32 (no node per entry, no key copy — a slot names
  // the first spec that carried the vector, resp. the row whose value it is), sized once for the worst case.
  auto mix64 = [](u64 z) {
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  };
  size_t tab_cap = 64;
  while (tab_cap < (size_t)S * 2 + 2) tab_cap <<= 1;
  std::vector<int32_t> res_tab(tab_cap, -1);           // slot → signature id
  std::vector<int32_t> res_first;                      // signature id → first spec with that request vector
  std::vector<std::vector<int32_t>> dim_tab((size_t)R);  // per dimension, grown on demand: slot → plane row (its value: h_dim_val)
  std::vector<size_t> dim_used((size_t)R, 0);
  auto dim_row = [&](int r, i64 q) -> int32_t {  // the plane row of value q in dimension r (a new one when nobody had it)
    std::vector<int32_t>& tab = dim_tab[(size_t)r];
    if (tab.empty()) tab.assign(1024, -1);
    if ((dim_used[(size_t)r] + 1) * 2 > tab.size()) {  // keep the table at most half full
      std::vector<int32_t> bigger(tab.size() * 4, -1);
      for (int32_t row : tab)
        if (row >= 0) {
          size_t at = (size_t)mix64((u64)e->h_dim_val[(size_t)row]) & (bigger.size() - 1);
          while (bigger[at] >= 0) at = (at + 1) & (bigger.size() - 1);
          bigger[at] = row;
        }
      tab.swap(bigger);
    }
    size_t at = (size_t)mix64((u64)q) & (tab.size() - 1);
    while (tab[at] >= 0) {
      if (e->h_dim_val[(size_t)tab[at]] == q) return tab[at];
      at = (at + 1) & (tab.size() - 1);
    }
    const int32_t row = (int32_t)e->h_dim_val.size();
    tab[at] = row;
    dim_used[(size_t)r]++;
    e->h_dim_val.push_back(q);
    e->h_dim_of.push_back(r);
    return row;
  };
  std::vector<int32_t> res_rows;                                   // [vectors][1 + R]
  e->h_dim_val.assign(1, 0);
  e->h_dim_of.assign(1, -1);
  std::vector<u64> sig_tol, sig_ports, sig_aff_terms, sig_pre_terms;
  std::vector<uint32_t> sig_tolflags, sig_aff_flags;
  std::vector<int32_t> sig_aff_off{0}, sig_pre_off{0};
  const uint32_t aff_flag_mask = YKPRED_SPEC_AFFINITY_SKIP | YKPRED_SPEC_PREFILTER_REJECT | YKPRED_SPEC_PREFILTER_NAMES;
  // (10^6 specs: the key strings are three buffers reused by every spec — a hit costs no allocation —, and the tables are sized for the
  // population where every spec is its own request vector, so they do not rehash on the way there)
  m_tol.reserve(4096);
  m_aff.reserve(4096);
  res_rows.reserve((size_t)std::min(S, 1 << 16) * (size_t)(R + 1));
  std::string kt, ka;
  for (int i = 0; i < S; ++i) {
    const int64_t* rq = s->requests + (size_t)i * R;
    u64 hv = 0x9e3779b97f4a7c15ull;
    for (int r = 0; r < R; ++r) hv = mix64(hv ^ (u64)rq[r]);
    size_t at = (size_t)hv & (tab_cap - 1);
    int32_t sig = -1;
    while (res_tab[at] >= 0) {
      if (memcmp(s->requests + (size_t)res_first[(size_t)res_tab[at]] * R, rq, (size_t)R * sizeof(i64)) == 0) {
        sig = res_tab[at];
        break;
      }
      at = (at + 1) & (tab_cap - 1);
    }
    if (sig < 0) {
      sig = (int32_t)res_first.size();
      res_tab[at] = sig;
      res_first.push_back(i);
      res_rows.push_back(0);  // the pod-independent row
      for (int r = 0; r < R; ++r) {
        const i64 q = (i64)rq[r];
        res_rows.push_back(q > 0 ? dim_row(r, q) : -1);  // "req_r > 0 ∧ req_r > free_r" fails: a non-positive request never does
      }
    }
    e->spec_sig_res[(size_t)i] = sig;

    uint32_t tf = s->flags[i] & (YKPRED_SPEC_TOLERATES_UNSCHEDULABLE | YKPRED_SPEC_UNSUPPORTED);
    kt.assign((const char*)(s->tolerated + (size_t)i * KT), (size_t)KT * sizeof(u64));
    kt.append((const char*)&tf, sizeof(tf));
    if (KP > 0) kt.append((const char*)(s->wanted_ports + (size_t)i * KP), (size_t)KP * sizeof(u64));  // NodePorts rides in this family
    auto jt = m_tol.find(kt);
    if (jt == m_tol.end()) {
      jt = m_tol.emplace(kt, (int32_t)m_tol.size()).first;
      sig_tol.insert(sig_tol.end(), s->tolerated + (size_t)i * KT, s->tolerated + (size_t)(i + 1) * KT);
      sig_tolflags.push_back(tf);
      if (KP > 0) sig_ports.insert(sig_ports.end(), s->wanted_ports + (size_t)i * KP, s->wanted_ports + (size_t)(i + 1) * KP);
    }
    e->spec_sig_tol[(size_t)i] = jt->second;

    uint32_t af = s->flags[i] & aff_flag_mask;
    int t0 = s->aff_term_off[i], t1 = s->aff_term_off[i + 1], p0 = s->pre_term_off[i], p1 = s->pre_term_off[i + 1];
    if (t1 < t0 || p1 < p0) return fail(e, YKPRED_E_INVALID, "set_specs: offsets not monotone");
    ka.assign((const char*)&af, sizeof(af));
    int32_t nt = t1 - t0;
    ka.append((const char*)&nt, sizeof(nt));
    if (nt) ka.append((const char*)(s->aff_terms + (size_t)t0 * W), (size_t)nt * W * sizeof(u64));
    if (p1 > p0) ka.append((const char*)(s->pre_terms + (size_t)p0 * W), (size_t)(p1 - p0) * W * sizeof(u64));
    auto kt2 = m_aff.find(ka);
    if (kt2 == m_aff.end()) {
      kt2 = m_aff.emplace(ka, (int32_t)m_aff.size()).first;
      sig_aff_flags.push_back(af);
      if (nt) sig_aff_terms.insert(sig_aff_terms.end(), s->aff_terms + (size_t)t0 * W, s->aff_terms + (size_t)t1 * W);
      if (p1 > p0) sig_pre_terms.insert(sig_pre_terms.end(), s->pre_terms + (size_t)p0 * W, s->pre_terms + (size_t)p1 * W);
      sig_aff_off.push_back((int32_t)(sig_aff_terms.size() / (size_t)W));
      sig_pre_off.push_back((int32_t)(sig_pre_terms.size() / (size_t)W));
    }
    e->spec_sig_aff[(size_t)i] = kt2->second;
  }
  // ---- PodTopologySpread signatures: constraints + the eligibility inputs they honour
  e->spec_sig_spread.assign((size_t)S, -1);
  e->spread_sig.clear();
  e->spread_sig_aff.clear();
  e->spread_sig_tol.clear();
  std::unordered_map<std::string, int32_t> m_spread;
  for (int i = 0; i < S && s->spread_off; ++i) {
    int c0 = s->spread_off[i], c1 = s->spread_off[i + 1];
    if (c1 < c0) return fail(e, YKPRED_E_INVALID, "set_specs: spread offsets not monotone");
    if (c1 == c0) continue;
    bool honor_aff = false, honor_tol = false;
    for (int c = c0; c < c1; ++c) {
      honor_aff |= (s->spread[c].flags & YKPRED_SPREAD_HONOR_AFFINITY) != 0;
      honor_tol |= (s->spread[c].flags & YKPRED_SPREAD_HONOR_TAINTS) != 0;
    }
    int32_t ea = honor_aff ? e->spec_sig_aff[(size_t)i] : 0, et = honor_tol ? e->spec_sig_tol[(size_t)i] : 0;
    std::string k((const char*)(s->spread + c0), (size_t)(c1 - c0) * sizeof(ykpred_spread_t));
    k.append((const char*)&ea, sizeof(ea));
    k.append((const char*)&et, sizeof(et));
    auto it = m_spread.find(k);
    if (it == m_spread.end()) {
      it = m_spread.emplace(std::move(k), (int32_t)m_spread.size()).first;
      e->spread_sig.emplace_back(s->spread + c0, s->spread + c1);
      e->spread_sig_aff.push_back(ea);
      e->spread_sig_tol.push_back(et);
    }
    e->spec_sig_spread[(size_t)i] = it->second;
  }
  e->fam_spread.D = (int)m_spread.size();
  // appended specs that bring no new topology signature leave the device tables and the histograms of the last pass valid
  const bool spread_unchanged = append_only && e->fam_spread.D == old_spread_D && !e->spread_dirty;
  if (!spread_unchanged) e->spread_dirty = true;
  if (!spread_unchanged) e->nodes_epoch++;  // the histograms of the last pass do not describe the new signatures
  TRY(upload(e, e->d_spec_spread, e->spec_sig_spread.data(), e->spec_sig_spread.size(), st));
  e->res_vectors = (int)res_first.size();
  e->fam_res.D = (int)e->h_dim_val.size();
  e->fam_tol.D = (int)m_tol.size();
  e->fam_aff.D = (int)m_aff.size();
  {
    // rows grouped by dimension. Dimensions with few distinct values are cut into ballot chunks of one dimension each (a
    // plane block reads one node-table column); dimensions with many are sorted by value and cut into walk chunks.
    const int rows = e->fam_res.D;
    std::vector<int32_t> order((size_t)rows), start((size_t)R + 2, 0);
    for (int d = 0; d < rows; ++d) start[(size_t)(e->h_dim_of[(size_t)d] + 1) + 1]++;
    for (size_t g = 1; g < start.size(); ++g) start[g] += start[g - 1];
    std::vector<int32_t> cursor(start.begin(), start.end() - 1);
    for (int d = 0; d < rows; ++d) order[(size_t)cursor[(size_t)(e->h_dim_of[(size_t)d] + 1)]++] = d;
    std::vector<int32_t> cdim, cbegin, clen, big_dim, wbig, wbegin, wlen;
    for (int g = 0; g <= R; ++g) {
      const int b0 = start[(size_t)g], b1 = start[(size_t)g + 1];
      if (g > 0 && b1 - b0 >= e->walk_rows && (int)big_dim.size() < ykk::kMaxIdxRows) {
        std::sort(order.begin() + b0, order.begin() + b1, [&](int32_t x, int32_t y) { return e->h_dim_val[(size_t)x] < e->h_dim_val[(size_t)y]; });
        for (int b = b0; b < b1; b += ykk::kWalkRows) {
          wbig.push_back((int32_t)big_dim.size());
          wbegin.push_back(b);
          wlen.push_back(std::min(ykk::kWalkRows, b1 - b));
        }
        big_dim.push_back(g - 1);
        continue;
      }
      for (int b = b0; b < b1; b += ykk::kDimRowsPerBlock) {
        cdim.push_back(g - 1);
        cbegin.push_back(b);
        clen.push_back(std::min(ykk::kDimRowsPerBlock, b1 - b));
      }
    }
    // rows of a walked dimension are INDEX rows: their entries of the request-vector table say so (and which mask table decodes them)
    if (!big_dim.empty()) {
      if (rows >= (1 << ykk::kRowBigShift)) return fail(e, YKPRED_E_UNSUPPORTED, "more than 2^28 request-value rows");
      std::vector<int32_t> big_of_dim((size_t)R, 0);
      for (size_t b = 0; b < big_dim.size(); ++b) big_of_dim[(size_t)big_dim[b]] = (int32_t)b + 1;
      for (int32_t& r : res_rows)
        if (r > 0 && big_of_dim[(size_t)e->h_dim_of[(size_t)r]]) r |= big_of_dim[(size_t)e->h_dim_of[(size_t)r]] << ykk::kRowBigShift;
    }
    e->index_rows = 0;
    for (int32_t len : wlen) e->index_rows += len;
    // what the sweep runs are built from (build_classes) and what k_dim_sort needs to turn a free value into a position
    e->h_res_rows = res_rows;
    e->h_row_pos.assign((size_t)rows, -1);
    e->h_row_big.assign((size_t)rows, -1);
    e->sweep_ready = false;  // (positions move with every new value: the runs are rebuilt with the classes)
    e->h_sorted_off.assign(1, 0);
    {
      std::vector<i64> sorted;
      for (size_t b = 0; b < big_dim.size(); ++b) {
        const int g = big_dim[b] + 1, b0 = start[(size_t)g], b1 = start[(size_t)g + 1];
        for (int i = b0; i < b1; ++i) {
          e->h_row_pos[(size_t)order[(size_t)i]] = i - b0;
          e->h_row_big[(size_t)order[(size_t)i]] = (int32_t)b;
          sorted.push_back(e->h_dim_val[(size_t)order[(size_t)i]]);
        }
        e->h_sorted_off.push_back((int32_t)sorted.size());
      }
      sorted.push_back(0);  // (never empty)
      TRY(upload(e, e->d_sorted, sorted.data(), sorted.size(), st));
      TRY(upload(e, e->d_sorted_off, e->h_sorted_off.data(), e->h_sorted_off.size(), st));
    }
    e->h_stage_rows.clear();
    {
      std::vector<uint8_t> walked((size_t)R, 0);
      for (int32_t d : big_dim) walked[(size_t)d] = 1;
      for (int d = 1; d < rows && (int)e->h_stage_rows.size() < ykk::kWalkMaxStage; ++d)
        if (!walked[(size_t)e->h_dim_of[(size_t)d]]) e->h_stage_rows.push_back(d);
    }
    e->dim_chunks = (int)cdim.size();
    e->n_big = (int)big_dim.size();
    e->walk_chunks = (int)wbig.size();
    TRY(upload(e, e->d_dim_val, e->h_dim_val.data(), e->h_dim_val.size(), st));
    TRY(upload(e, e->d_dim_order, order.data(), order.size(), st));
    TRY(upload(e, e->d_dim_chunk_dim, cdim.data(), cdim.size(), st));
    TRY(upload(e, e->d_dim_chunk_begin, cbegin.data(), cbegin.size(), st));
    TRY(upload(e, e->d_dim_chunk_len, clen.data(), clen.size(), st));
    TRY(upload(e, e->d_big_dim, big_dim.data(), big_dim.size(), st));
    TRY(upload(e, e->d_walk_big, wbig.data(), wbig.size(), st));
    TRY(upload(e, e->d_walk_begin, wbegin.data(), wbegin.size(), st));
    TRY(upload(e, e->d_walk_len, wlen.data(), wlen.size(), st));
    TRY(upload(e, e->d_res_rows, res_rows.data(), res_rows.size(), st));
  }
  TRY(upload(e, e->d_sig_tol, sig_tol.data(), sig_tol.size(), st));
  TRY(upload(e, e->d_sig_tolflags, sig_tolflags.data(), sig_tolflags.size(), st));
  TRY(upload(e, e->d_sig_ports, sig_ports.data(), sig_ports.size(), st));
  TRY(upload(e, e->d_sig_aff_flags, sig_aff_flags.data(), sig_aff_flags.size(), st));
  TRY(upload(e, e->d_sig_aff_off, sig_aff_off.data(), sig_aff_off.size(), st));
  TRY(upload(e, e->d_sig_pre_off, sig_pre_off.data(), sig_pre_off.size(), st));
  TRY(upload(e, e->d_sig_aff_terms, sig_aff_terms.data(), sig_aff_terms.size(), st));
  TRY(upload(e, e->d_sig_pre_terms, sig_pre_terms.data(), sig_pre_terms.size(), st));
  HIPCHK(hipStreamSynchronize(st));
  e->specs_set = true;
  if (!append_only) e->classes_dirty = true;
  if (!spread_unchanged) e->last_eval_valid = false;  // new topology signatures have no histogram yet: full pass first
  return YKPRED_OK;
}

int32_t ykpred_set_pods(ykpred_engine_t* e, const ykpred_pods_t* p) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:upload_asks");
  if (e) e->n_uploads++;
  if (e) e->tables_version++;
  if (e) e->ask_epoch++;
  if (!e || !p || p->count < 0) return fail(e, YKPRED_E_INVALID, "set_pods: bad argument");
  if (p->count > 0 && (!p->spec_index || !p->node_name_index)) return fail(e, YKPRED_E_INVALID, "set_pods: null column");
  HIPCHK(hipSetDevice(e->cfg.device));
  e->P = p->count;
  e->h_pod_spec.assign(p->spec_index, p->spec_index + p->count);
  e->h_pod_pin.assign(p->node_name_index, p->node_name_index + p->count);
  TRY(upload(e, e->d_pod_spec, e->h_pod_spec.data(), e->h_pod_spec.size(), e->own_stream));
  TRY(upload(e, e->d_pod_pin, e->h_pod_pin.data(), e->h_pod_pin.size(), e->own_stream));
  HIPCHK(hipStreamSynchronize(e->own_stream));
  e->pods_set = true;
  e->classes_dirty = true;
  return YKPRED_OK;
}

static int validate_state(ykpred_engine* e) {
  if (!e->nodes_set || !e->specs_set || !e->pods_set) return fail(e, YKPRED_E_STATE, "nodes, specs and pods must be uploaded first");
  for (int p = 0; p < e->P; ++p) {
    int s = e->h_pod_spec[(size_t)p], pin = e->h_pod_pin[(size_t)p];
    if (s < 0 || s >= e->S) return fail(e, YKPRED_E_INVALID, "pod references a spec that was not uploaded");
    if (pin < YKPRED_UNKNOWN_NODE_NAME || pin >= e->N) return fail(e, YKPRED_E_INVALID, "pod node_name_index out of range");
  }
  return YKPRED_OK;
}

int32_t ykpred_eval(ykpred_engine_t* e, const ykpred_eval_args_t* a) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:eval");
  if (!e || !a) return fail(e, YKPRED_E_INVALID, "eval: bad argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = a->stream ? (hipStream_t)a->stream : e->own_stream;
  if (e->classes_dirty) {
    // A class build renumbers the classes and re-packs the bitmap rows: a pass that leaves the bitmap alone (decision refresh,
    // dirty-class rewrite) would publish an evaluation whose rows belong to the OLD layout. The caller runs a full pass instead.
    if (a->options & (YKPRED_EVAL_SKIP_BITMAP | YKPRED_EVAL_DIRTY_CLASSES))
      return fail(e, YKPRED_E_STATE, "eval: the pod classes are due for a rebuild — run a full ykpred_eval (bitmap included)");
    TRY(validate_state(e));
    TRY(build_classes(e, e->own_stream));
  }
  const unsigned pre = a->prefilter_plugins, filt = a->filter_plugins;
  const bool spread_filt = filt & YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD, spread_pre = pre & YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD;
  const bool ipa_filt = filt & YKPRED_PLUGIN_INTER_POD_AFFINITY, ipa_pre = pre & YKPRED_PLUGIN_INTER_POD_AFFINITY;
  // Filter without PreFilter state (PodTopologySpread, NodePorts, InterPodAffinity): every pair fails with an Error status
  const bool spread_err = (spread_filt && !spread_pre) || (ipa_filt && !ipa_pre) ||
                          ((filt & YKPRED_PLUGIN_NODE_PORTS) && !(pre & YKPRED_PLUGIN_NODE_PORTS));
  if (e->spread_dirty) TRY(build_spread_tables(e, e->own_stream));
  const bool pts_en = spread_filt && spread_pre, ipa_en = ipa_filt && ipa_pre;
  const bool spread_on = (pts_en || ipa_en) && e->fam_spread.D > 0;  // the topology-constraint family (spread + inter-pod affinity)
  const int N = e->N, P = e->P;
  const size_t bitmap_bytes = (size_t)std::max(std::max(e->rows_total, e->row_capacity), 1) * (size_t)e->row_stride * sizeof(u64);
  u64* bitmap = (u64*)a->bitmap;
  int64_t bitmap_rows = 0;
  if (!bitmap || (void*)bitmap == e->d_bitmap.p) {  // (ykpred_eval_nodes hands the engine's own buffer back)
    HIPCHK(e->d_bitmap.ensure(bitmap_bytes));
    bitmap = e->d_bitmap.as<u64>();
    bitmap_rows = (int64_t)(e->d_bitmap.cap / ((size_t)std::max(e->row_stride, 1) * sizeof(u64)));
  } else {
    // A caller-owned bitmap has to say how many rows it holds: rows are laid out for the writer (band padding) and every
    // changed or appended ask takes a fresh row, so the row count is not the ask count — and the engine must never store
    // past the end of memory it does not own.
    bitmap_rows = a->bitmap_rows ? (int64_t)a->bitmap_rows : (int64_t)e->row_capacity;
    if (bitmap_rows == 0)
      return fail(e, YKPRED_E_INVALID, "eval: a caller-owned bitmap needs eval_args.bitmap_rows or ykpred_set_row_capacity (it must hold layout.num_rows rows)");
    if ((int64_t)e->rows_total > bitmap_rows)
      return fail(e, YKPRED_E_INVALID, "eval: the caller-owned bitmap holds fewer rows than the layout needs (layout.num_rows)");
  }
  e->last_bitmap = bitmap;
  e->last_bitmap_rows = bitmap_rows;
  HIPCHK(e->d_counts.ensure((size_t)std::max(P, 1) * sizeof(int)));
  HIPCHK(e->d_decisions.ensure((size_t)std::max(P, 1) * sizeof(int)));
  HIPCHK(e->d_keys.ensure((size_t)std::max(P, 1) * sizeof(i64)));
  e->last_counts = a->counts ? a->counts : e->d_counts.p;
  e->last_decisions = a->decisions ? a->decisions : e->d_decisions.p;
  e->last_keys = a->decision_keys ? a->decision_keys : e->d_keys.p;
  TRY(ensure_planes(e, st));
  if (e->n_big > 0) {
    const size_t cells = (size_t)e->n_big * (size_t)std::max(e->row_words, 1);
    for (DevBuf* b : {&e->d_sfree_c, &e->d_sfree_r}) HIPCHK(b->ensure(cells * 64 * sizeof(i64)));
    for (DevBuf* b : {&e->d_pmask_c, &e->d_pmask_r}) HIPCHK(b->ensure(cells * 65 * sizeof(u64)));
    HIPCHK(e->d_rbits_c.ensure(cells * ykk::kRankBits * sizeof(u64)));
    HIPCHK(e->d_ent_c.ensure(cells * 65 * sizeof(unsigned)));
    HIPCHK(e->d_glin_r.ensure(cells * 65 * sizeof(unsigned)));  // (g per node + the largest g per word)
    // (a forced row stride — unequal shards — may exceed the rounded row length: consumers address index bytes by row word)
    const int idx_stride_before = e->idx_stride;
    e->idx_stride = (std::max(e->row_words, e->row_stride) + 63) / 64 * 64;
    for (DevBuf* b : {&e->d_idx_c}) {
      const size_t need = (size_t)e->fam_res.D * (size_t)e->idx_stride;
      if (b->cap < need || !b->p || idx_stride_before != e->idx_stride) {
        HIPCHK(b->ensure(need));
        HIPCHK(hipMemsetAsync(b->p, 64, need, st));  // bytes past the row decode to entry 64 of a mask table: no node
      }
    }
    {  // (window bytes nobody wrote are never read: a word is looked up in a window only where the walk stored it)
      const size_t need = (size_t)e->fam_res.D * 64;
      if (e->d_win_r.cap < need || !e->d_win_r.p) {
        HIPCHK(e->d_win_r.ensure(need));
        HIPCHK(hipMemsetAsync(e->d_win_r.p, 64, need, st));
      }
    }
  }
  Timer tm{e, (a->options & YKPRED_EVAL_PROFILE) != 0};
  tm.start(st);
  if (N == 0 || P == 0) {
    tm.done(st);
    return YKPRED_OK;
  }
  ykk::NodeTable nt = node_table(e);
  const bool want_keys = a->options & YKPRED_OUT_DECISION_KEYS;
  const bool want_dec = (a->options & YKPRED_OUT_DECISIONS) || want_keys;
  const bool want_cnt = a->options & YKPRED_OUT_COUNTS;
  const int nblk_nodes = (N + ykk::kBlock - 1) / ykk::kBlock;

  // Everything below only ENQUEUES work (kernels, memsets, cross-stream events) on `st` / the decision stream, so one pass
  // can be captured into a hipGraph. Returns 1 when the pass ends early (histograms only, per-pair kernel).
  bool use_run_decide_pass = false;  // (set by enqueue: the pass left windows only for the rows k_decide read)
  auto enqueue = [&]() -> int {
  if (spread_on || (a->options & YKPRED_EVAL_SPREAD_COUNT_ONLY)) {
    const bool ready = a->options & YKPRED_EVAL_SPREAD_COUNTS_READY, only = a->options & YKPRED_EVAL_SPREAD_COUNT_ONLY;
    TRY(run_spread_prefilter(e, st, &tm, !ready, false));
    if (only) return 1;
    if (!ready) TRY(allreduce_spread(e, st));  // node-sharded cluster: the histograms become cluster-wide here
    TRY(run_spread_prefilter(e, st, &tm, false, true));
    e->hist_epoch = e->nodes_epoch;
  }
  if (a->options & YKPRED_EVAL_DIRECT) {
    ykk::SpecTable stbl = spec_table(e);
    dim3 grid((unsigned)((P + ykk::kWave - 1) / ykk::kWave), (unsigned)((e->row_stride + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock));
    tm.begin(st);
    hipLaunchKernelGGL(ykk::k_direct, grid, dim3(ykk::kBlock), 0, st, nt, stbl, P, e->d_pod_spec.as<int>(), e->d_pod_pin.as<int>(),
                       e->d_pod_row.as<int>(), pre, filt, bitmap, e->row_words, e->row_stride);
    tm.end(st, "k_direct");
    HIPCHK(hipGetLastError());
    return 1;
  }

  const unsigned wgroups = (unsigned)((e->row_words + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock);
  const bool res_on = filt & YKPRED_PLUGIN_NODE_RESOURCES_FIT;
  const bool aff_on = (filt | pre) & YKPRED_PLUGIN_NODE_AFFINITY;
  const int fit_error = (pre & YKPRED_PLUGIN_NODE_RESOURCES_FIT) ? 0 : 1;
  auto sig_chunks = [](int D) {
    const int spb = ykk::sigs_per_block(D);
    return (unsigned)((D + spb - 1) / spb);
  };
  ykk::AffSigs as{e->d_sig_aff_flags.as<unsigned>(), e->d_sig_aff_off.as<int>(), e->d_sig_aff_terms.as<u64>(), e->d_sig_pre_off.as<int>(),
                  e->d_sig_pre_terms.as<u64>()};
  auto canon_of = [&](const Family& f) { return e->planes_canon.as<u64>() + (size_t)f.base * e->row_stride; };
  auto ranked_of = [&](const Family& f) { return e->planes_ranked.as<u64>() + (size_t)f.base * e->row_stride; };
  ykk::PlaneOut o_res{canon_of(e->fam_res), ranked_of(e->fam_res), e->row_stride, e->fam_res.D, nullptr};
  ykk::PlaneOut o_tol{canon_of(e->fam_tol), ranked_of(e->fam_tol), e->row_stride, e->fam_tol.D, nullptr};
  ykk::PlaneOut o_aff{canon_of(e->fam_aff), ranked_of(e->fam_aff), e->row_stride, e->fam_aff.D, nullptr};
  ykk::PlaneOut o_spread{canon_of(e->fam_spread), ranked_of(e->fam_spread), e->row_stride, e->fam_spread.D, nullptr};
  // k_sweep_rows takes the sweep runs of a FULL pass with NodeResourcesFit evaluated (the reservation phase has no request rows: every
  // class of a signature shares one row and the chunk writers do it); a dirty-class pass, a pass without PreFilter state for the
  // Filter, a class build whose runs were touched since — all fall back to the chunk writers, which know every chunk.
  // (Not where the class counts are known before a row is written — counts_early below, a small zone B: those writers add nothing to
  // the counts and the run kernels would count a class twice.)
  const bool full_pass = !(a->options & (YKPRED_EVAL_SKIP_BITMAP | YKPRED_EVAL_DIRTY_CLASSES));
  const bool counts_early_here = e->early_counts != 0 && full_pass && e->n_classes_a > 0 && e->patch_chunks == 0 &&
                                 !((long)e->NC * e->wave_combine_below > (long)P) && e->n_classes_b <= 16384;
  const bool zero_counts_beside = full_pass && !counts_early_here && e->ev_zero != nullptr && e->C >= 65536 &&
                                  (a->options & (YKPRED_OUT_DECISIONS | YKPRED_OUT_DECISION_KEYS)) != 0;
  const bool use_sweep = e->sweep_ready && e->sweep_runs > 0 && res_on && !fit_error && full_pass && !counts_early_here;
  const bool sweep_rows_on = use_sweep && e->n_big > 0 && e->sweep_row_off[ykk::kMaxIdxRows] > 0;  // k_sweep_rows has rows (k_class_runs: run_classes)
  // k_fused_rows takes the fused classes of a full pass whose writers add the counts (NodeResourcesFit and node affinity evaluated,
  // every Filter with its PreFilter state)
  const bool fuse_on = e->fuse_ready && res_on && !fit_error && full_pass && !counts_early_here && aff_on && !spread_err;
  // the decisions of the sweep runs come from k_run_decide wherever decisions are produced from walked request rows (any pass, also a
  // decision refresh without the bitmap): k_decide skips those classes and no window of their index rows is written
  const bool use_run_decide = e->run_decide != 0 && e->sweep_ready && e->run_ranges > 0 && res_on && !fit_error && e->n_big > 0 &&
                              (a->options & (YKPRED_OUT_DECISIONS | YKPRED_OUT_DECISION_KEYS)) && e->walk2_chunks >= 0 &&
                              e->C == e->run_decide_classes;  // (a class that row patches added since the build is in neither list: k_decide takes them all then)
  use_run_decide_pass = use_run_decide;
  ykk::ClassTable ct{e->d_class_sig.as<int>(), e->d_class_pin.as<int>(),  e->d_chunk_class.as<int>(), e->d_chunk_begin.as<int>(),
                     e->d_chunk_len.as<int>(), e->d_chunk_first.as<int>(), e->d_members.as<int>(), e->d_chunk_zone.as<int>(), (use_sweep ? 1 : 0) | (fuse_on ? 2 : 0),
                     use_run_decide ? e->d_no_decide.as<int>() : nullptr};
  ykk::Planes pc{res_on ? o_res.canon : nullptr, o_tol.canon, aff_on ? o_aff.canon : nullptr, spread_on ? o_spread.canon : nullptr,
                 e->row_stride, e->d_res_rows.as<int>(), 1 + e->R, e->d_idx_c.as<unsigned char>(), e->idx_stride, e->d_pmask_c.as<u64>(), e->row_words,
                 nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  // first non-zero word of every rank-ordered plane row (whole buffer: families at their base rows), reset per pass
  int* first_r = nullptr;
  if ((a->options & (YKPRED_OUT_DECISIONS | YKPRED_OUT_DECISION_KEYS)) && e->decide_skip) {
    HIPCHK(e->d_first_r.ensure((size_t)std::max(e->plane_rows_alloc, 1) * sizeof(int)));
    first_r = e->d_first_r.as<int>();
  }
  if (want_dec && res_on && e->n_big > 0 && !fit_error) HIPCHK(e->d_pfx_r.ensure((size_t)e->n_big * (size_t)std::max(e->row_words, 1) * sizeof(i64)));
  ykk::Planes pr = ranked_planes_of(e, pre, filt, spread_on, first_r != nullptr);
  pc.n_big = pr.n_big;
  pc.rbits = e->d_rbits_c.as<u64>();
  const int pin_on = ((filt & YKPRED_PLUGIN_NODE_NAME) ? 1 : 0) | (spread_err ? 2 : 0);
  hipStream_t sb = e->aux_stream;

  // ---- stream B (decision branch), part 1: bin-pack order. Independent of the planes, runs beside them.
  if (want_dec) {
    HIPCHK(hipEventRecord(e->ev_fork, st));
    HIPCHK(hipStreamWaitEvent(sb, e->ev_fork, 0));
    if (zero_counts_beside) {
      // the class counts of a pass whose writers add to them are zeroed here, beside the plane kernels (4 MB at 10^6 classes: a fill
      // that took 60 us in front of the writers on the launch stream)
      HIPCHK(hipMemsetAsync(e->d_class_count.p, 0, (size_t)e->C * sizeof(int), sb));
      HIPCHK(hipEventRecord(e->ev_zero, sb));
    }
    if (first_r) HIPCHK(hipMemsetAsync(first_r, 0x7f, (size_t)std::max(e->plane_rows_alloc, 1) * sizeof(int), sb));  // = ykk::kNoWord
    tm.begin(sb);
    hipLaunchKernelGGL(ykk::k_score, dim3((unsigned)nblk_nodes), dim3(ykk::kBlock), 0, sb, nt, e->d_score.as<double>(), e->d_key.as<u64>());
    tm.end(sb, "k_score");
    // bucketed exact sort by (score, node index): histogram → scan → bucket fill → rank within bucket
    int* hist = e->d_rankbuf.as<int>();
    int* bucket_off = hist + ykk::kRankBuckets;
    int* cursor = bucket_off + ykk::kRankBuckets + 1;
    int* members = cursor + ykk::kRankBuckets;
    HIPCHK(hipMemsetAsync(hist, 0, ykk::kRankBuckets * sizeof(int), sb));
    tm.begin(sb);
    hipLaunchKernelGGL(ykk::k_rank_hist, dim3((unsigned)nblk_nodes), dim3(ykk::kBlock), 0, sb, N, e->d_score.as<double>(), hist);
    hipLaunchKernelGGL(ykk::k_rank_scan, dim3(1), dim3(ykk::kRankBuckets), 0, sb, hist, bucket_off, cursor);
    hipLaunchKernelGGL(ykk::k_rank_fill, dim3((unsigned)nblk_nodes), dim3(ykk::kBlock), 0, sb, N, e->d_score.as<double>(), e->d_key.as<u64>(),
                       e->has_name_rank ? e->d_name_rank.as<int>() : (const int*)nullptr, cursor, members, e->d_member_tie.as<int>(),
                       e->d_member_key.as<u64>());
    hipLaunchKernelGGL(ykk::k_rank_final, dim3((unsigned)ykk::kRankBuckets), dim3(ykk::kBlock), 0, sb, bucket_off, members,
                       e->d_member_tie.as<int>(), e->d_member_key.as<u64>(), e->d_rank.as<int>(), e->d_perm.as<int>());
    tm.end(sb, "k_rank");
  }
  // Bit-sliced dictionaries + signature planes of one node order (perm == nullptr: canonical) on stream s.
  auto base_of = [&](DevBuf& buf) {
    ykk::BasePlanes bp;
    bp.req = buf.as<u64>();
    bp.taint = bp.req + (size_t)64 * e->W * e->row_stride;
    bp.port = bp.taint + (size_t)64 * e->KT * e->row_stride;
    bp.unsched = bp.port + (size_t)64 * e->KP * e->row_stride;
    bp.exists = bp.unsched + e->row_stride;
    bp.stride = e->row_stride;
    return bp;
  };
  auto launch_dictionary_planes = [&](hipStream_t s, const int* perm, bool ranked, const char* base_name, const char* sig_name, bool base_done = false) -> int {
    ykk::BasePlanes bp = base_of(ranked ? e->base_ranked : e->base_canon);
    if (!base_done) {
      tm.begin(s);
      hipLaunchKernelGGL(ykk::k_base_planes, dim3((unsigned)(e->W + e->KT + e->KP + 1), wgroups), dim3(ykk::kBlock), 0, s, nt, perm, bp,
                         e->row_words);
      tm.end(s, base_name);
    }
    ykk::SigPlaneArgs sa{};
    sa.base = bp;
    sa.tol = o_tol;
    sa.aff = o_aff;
    if (ranked) {
      sa.tol.canon = o_tol.ranked;
      sa.aff.canon = o_aff.ranked;
      sa.tol.first = first_r ? first_r + e->fam_tol.base : nullptr;
      sa.aff.first = first_r ? first_r + e->fam_aff.base : nullptr;
    }
    if (!aff_on) sa.aff.D = 0;
    sa.sig_tol = e->d_sig_tol.as<u64>();
    sa.sig_tolflags = e->d_sig_tolflags.as<unsigned>();
    sa.sig_ports = e->d_sig_ports.as<u64>();
    sa.KP = e->KP;
    for (int k = 0; k < ykk::kMaxKT; ++k) sa.taint_used[k] = e->taint_used[k];
    sa.affs = as;
    sa.KT = e->KT;
    sa.W = e->W;
    sa.pre_mask = pre;
    sa.filt_mask = filt;
    sa.n_words = e->row_words;
    tm.begin(s);
    // words per lane: wide rows give a lane 4 (or 2) words 64 apart — the per-signature latency chain moves more bytes
    const int wpl = e->sig_wpl > 0 ? e->sig_wpl : (e->row_words >= 4 * ykk::kWave ? 4 : (e->row_words >= 2 * ykk::kWave ? 2 : 1));
    const unsigned chunks = std::max((unsigned)((std::max(sa.tol.D, sa.aff.D) + ykk::kBitSigsPerBlock - 1) / ykk::kBitSigsPerBlock), 1u);
    const unsigned ygroups = (unsigned)((e->row_words + ykk::kBlock * wpl - 1) / (ykk::kBlock * wpl));
    if (wpl >= 4)
      hipLaunchKernelGGL(ykk::k_sig_planes<4>, dim3(chunks, ygroups, 2u), dim3(ykk::kBlock), 0, s, sa);
    else if (wpl >= 2)
      hipLaunchKernelGGL(ykk::k_sig_planes<2>, dim3(chunks, ygroups, 2u), dim3(ykk::kBlock), 0, s, sa);
    else
      hipLaunchKernelGGL(ykk::k_sig_planes<1>, dim3(chunks, ygroups, 2u), dim3(ykk::kBlock), 0, s, sa);
    tm.end(s, sig_name);
    return YKPRED_OK;
  };
  // ---- stream B, part 2: dictionary planes in rank order (needs only the bin-pack order). With few signature planes
  // altogether the rank-ordered copies of ALL of them come from one bit permutation of the canonical planes (part 3) — a
  // fraction of the work of evaluating the dictionary families a second time, and less traffic beside the band writer.
  if (want_dec) TRY(launch_dictionary_planes(sb, e->d_perm.as<int>(), true, "k_base_planes(ranked)", "k_sig_planes(ranked)"));
  // ---- stream A: canonical planes. Ballot families (request vectors, spread) in one launch, then the bit-sliced ones.
  auto ranked_walk = [](const int* perm) { return perm != nullptr; };
  auto launch_ballot_planes = [&](hipStream_t s, const int* perm, const char* name, bool with_base = false) {
    ykk::PlaneArgs pa{};
    pa.perm = perm;
    pa.res = o_res;
    pa.spread = o_spread;
    if (perm && first_r) {
      pa.res.first = first_r + e->fam_res.base;
      pa.spread.first = first_r + e->fam_spread.base;
    }
    if (!spread_on) pa.spread.D = 0;
    pa.dims = ykk::DimPlanes{e->d_dim_val.as<i64>(), e->d_dim_order.as<int>(), e->d_dim_chunk_dim.as<int>(), e->d_dim_chunk_begin.as<int>(),
                             e->d_dim_chunk_len.as<int>(), res_on ? e->dim_chunks : 0};
    pa.spreads = spread_sigs(e);
    pa.fit_error = fit_error;
    pa.n_words = e->row_words;
    pa.spread_en = pts_en ? 1 : 0;
    pa.ipa_en = ipa_en ? 1 : 0;
    unsigned xchunks = std::max((unsigned)pa.dims.n_chunks, sig_chunks(pa.spread.D));
    tm.begin(s);
    if (with_base) {
      // the bit-sliced dictionaries of the same node order ride in the same launch (k_node_planes: one boundary instead of two)
      xchunks = std::max(xchunks, (unsigned)(e->W + e->KT + e->KP + 1));
      hipLaunchKernelGGL(ykk::k_node_planes, dim3(std::max(xchunks, 1u), wgroups, 3u), dim3(ykk::kBlock), 0, s, nt, pa,
                         base_of(perm ? e->base_ranked : e->base_canon));
    } else {
      hipLaunchKernelGGL(ykk::k_planes, dim3(std::max(xchunks, 1u), wgroups, spread_on ? 2u : 1u), dim3(ykk::kBlock), 0, s, nt, pa);
    }
    tm.end(s, name);
    if (res_on && e->n_big > 0 && !fit_error) {
      // many-valued dimensions: sort every word's free values once, then one thread per word walks the sorted rows
      const bool ranked = perm != nullptr;
      ykk::DimWalk dw{e->d_dim_val.as<i64>(), e->d_dim_order.as<int>(), e->d_big_dim.as<int>(), e->d_walk_big.as<int>(), e->d_walk_begin.as<int>(),
                      e->d_walk_len.as<int>(), (ranked ? e->d_sfree_r : e->d_sfree_c).as<i64>(), (ranked ? e->d_pmask_r : e->d_pmask_c).as<u64>(),
                      e->n_big, e->walk_chunks, e->row_words, ranked ? nullptr : e->d_rbits_c.as<u64>(),
                      (!ranked && sweep_rows_on) ? e->d_ent_c.as<unsigned>() : nullptr, e->d_sorted.as<i64>(), e->d_sorted_off.as<int>(),
                      (ranked && use_run_decide) ? e->d_glin_r.as<unsigned>() : nullptr};
      tm.begin(s);
      hipLaunchKernelGGL(ykk::k_dim_sort, dim3((unsigned)e->n_big, wgroups), dim3(ykk::kBlock), 0, s, nt, perm, dw);
      tm.end(s, ranked ? "k_dim_sort(ranked)" : "k_dim_sort");
      tm.begin(s);
      const unsigned walk_gy = (unsigned)((e->row_words + ykk::kBlock * ykk::kWalkWords - 1) / (ykk::kBlock * ykk::kWalkWords));
      const dim3 walk_grid((unsigned)e->walk_chunks, walk_gy);
      if (!ranked && use_sweep && e->walk2_chunks >= 0) {
        // the sweep runs read no index row: only the rows somebody else decodes are walked (build_classes)
        if (e->walk2_chunks > 0) {
          ykk::DimWalk dw2 = dw;
          dw2.order = e->d_walk2_order.as<int>();
          dw2.chunk_big = e->d_walk2_big.as<int>();
          dw2.chunk_begin = e->d_walk2_begin.as<int>();
          dw2.chunk_len = e->d_walk2_len.as<int>();
          dw2.n_chunks = e->walk2_chunks;
          hipLaunchKernelGGL(ykk::k_dim_walk, dim3((unsigned)e->walk2_chunks, walk_gy), dim3(ykk::kBlock), 0, s, dw2, e->d_idx_c.as<unsigned char>(), e->idx_stride);
        }
      } else if (!ranked) {
        hipLaunchKernelGGL(ykk::k_dim_walk, walk_grid, dim3(ykk::kBlock), 0, s, dw, e->d_idx_c.as<unsigned char>(), e->idx_stride);
      } else {
        // running maximum of the free values along the bin-pack order: where a value's row can start (k_decide) — and the window
        // of the row that is worth writing in rank order (k_dim_walk_window)
        hipLaunchKernelGGL(ykk::k_dim_prefix_max, dim3((unsigned)e->n_big), dim3(ykk::kPfxBlock), 0, s, dw, e->d_pfx_r.as<i64>());
        if (use_run_decide) {
          // only the windows k_decide reads (the classes outside the sweep runs); a round completes them first (complete_windows)
          if (e->walk2_chunks > 0) {
            ykk::DimWalk dw2 = dw;
            dw2.order = e->d_walk2_order.as<int>();
            dw2.chunk_big = e->d_walk2_big.as<int>();
            dw2.chunk_begin = e->d_walk2_begin.as<int>();
            dw2.chunk_len = e->d_walk2_len.as<int>();
            dw2.n_chunks = e->walk2_chunks;
            hipLaunchKernelGGL(ykk::k_dim_walk_window<8>, dim3((unsigned)e->walk2_chunks, walk_gy), dim3(ykk::kBlock), 0, s, dw2, e->d_pfx_r.as<i64>(), e->d_win_r.as<unsigned char>());
          }
        } else {
          hipLaunchKernelGGL(ykk::k_dim_walk_window<ykk::kWalkBatch>, walk_grid, dim3(ykk::kBlock), 0, s, dw, e->d_pfx_r.as<i64>(), e->d_win_r.as<unsigned char>());
        }
      }
      tm.end(s, ranked ? "k_dim_walk(ranked)" : "k_dim_walk");
    } else if (res_on && e->n_big > 0) {
      // Filter without PreFilter state: the walked rows fit nowhere like every other row of the family — position 64 of every
      // mask table is the empty mask (the tables are only written by k_dim_sort: clear them as well)
      if (!ranked_walk(perm)) (void)hipMemsetAsync(e->d_idx_c.p, 64, (size_t)e->fam_res.D * (size_t)e->idx_stride, s);
      (void)hipMemsetAsync((ranked_walk(perm) ? e->d_pmask_r : e->d_pmask_c).p, 0, (size_t)e->n_big * (size_t)e->row_words * 65 * sizeof(u64), s);
      if (!ranked_walk(perm)) (void)hipMemsetAsync(e->d_rbits_c.p, 0, (size_t)e->n_big * (size_t)e->row_words * ykk::kRankBits * sizeof(u64), s);
    }
  };
  const bool fused_planes = res_on || spread_on;
  if (fused_planes) launch_ballot_planes(st, nullptr, "k_planes+k_base_planes", true);
  TRY(launch_dictionary_planes(st, nullptr, false, "k_base_planes", "k_sig_planes", fused_planes));
  // ---- stream B, part 3 (after the canonical ballot planes): their rank-ordered copies by bit permutation, then the
  // first feasible node of every class. Overlaps the start of k_combine.
  if (want_dec) {
    HIPCHK(hipEventRecord(e->ev_planes, st));
    HIPCHK(hipStreamWaitEvent(sb, e->ev_planes, 0));
    const int ballot_rows = e->fam_tol.base;  // res + spread rows come first in the plane buffers
    if (ballot_rows <= ykk::kManySigs && e->n_big == 0) {  // (index rows of walked dimensions are not bit planes: they are re-walked in rank order)
      tm.begin(sb);
      hipLaunchKernelGGL(ykk::k_permute_planes, dim3(sig_chunks(ballot_rows), wgroups), dim3(ykk::kBlock), 0, sb, N, e->d_perm.as<int>(),
                         e->planes_canon.as<u64>(), e->planes_ranked.as<u64>(), e->row_stride, ballot_rows, e->row_words, first_r);
      tm.end(sb, "k_permute_planes");
    } else if (res_on || spread_on) {
      // very many request / spread signatures: the bit gather would touch one cache line per lane and row; evaluating
      // the signatures again in permuted node order is cheaper
      launch_ballot_planes(sb, e->d_perm.as<int>(), "k_planes(ranked)");
    }
    if (use_run_decide) {
      tm.begin(sb);
      hipLaunchKernelGGL(ykk::k_run_decide, dim3((unsigned)((e->run_ranges + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, sb, pr,
                         e->d_sweep_runs.as<ykk::SweepRun>(), e->d_run_ranges.as<ykk::RunRange>(), e->run_ranges, e->d_sweep_rows.as<ykk::SweepRow>(),
                         e->d_glin_r.as<unsigned>(), e->d_sorted.as<i64>(), e->d_sorted_off.as<int>(), e->d_perm.as<int>(), e->row_words, N, pin_on,
                         e->d_class_best.as<int>());
      tm.end(sb, "k_run_decide");
    }
    tm.begin(sb);
    const int n_decide = use_run_decide ? e->n_decide_list : e->C;
    if (n_decide == 0) {
      // (every class is a row of a sweep run)
    } else if (n_decide >= e->decide_groups_from) {
      // many classes: four per wave (k_decide_groups)
      const int per_block = ykk::kWavesPerBlock * ykk::kDecideGroups;
      hipLaunchKernelGGL(ykk::k_decide_groups, dim3((unsigned)((n_decide + per_block - 1) / per_block)), dim3(ykk::kBlock), 0, sb, ct, pr, n_decide,
                         e->row_words, e->d_perm.as<int>(), e->d_rank.as<int>(), pin_on, e->d_class_best.as<int>());
    } else {
      hipLaunchKernelGGL(ykk::k_decide, dim3((unsigned)((n_decide + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, sb, ct,
                         pr, n_decide, e->row_words, e->d_perm.as<int>(), e->d_rank.as<int>(), pin_on, e->d_class_best.as<int>(), n_decide <= 4096 ? 1 : 0);
    }
    tm.end(sb, "k_decide");
  }
  // ---- stream A: combine → bitmap (skipped by ykpred_eval_nodes' decision refresh: bitmap and class counts were
  // patched incrementally)
  const bool skip_combine = a->options & YKPRED_EVAL_SKIP_BITMAP;
  const bool dirty_only = a->options & YKPRED_EVAL_DIRTY_CLASSES;
  const int* class_dirty = nullptr;
  if (!skip_combine && dirty_only) {
    // classes whose topology signature changed (d_sig_changed was filled by ykpred_eval_nodes): flag them, zero their counts
    HIPCHK(e->d_class_dirty.ensure((size_t)std::max(e->C, 1) * sizeof(int)));
    hipLaunchKernelGGL(ykk::k_mark_dirty_classes, dim3((unsigned)((e->C + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, e->C,
                       e->d_class_sig.as<int>(), e->d_sig_changed.as<int>(), e->d_class_dirty.as<int>(), e->d_class_count.as<int>());
    class_dirty = e->d_class_dirty.as<int>();
  }
  // Threads per chunk of the class-by-class writer decide its form; with a SMALL zone B (the band layout took nearly every class)
  // k_class_rows counts the zone-B classes as well: every class count is known before a bitmap row is written, so the per-ask
  // scatter runs on the decision stream beside the band writer instead of behind it, and the writers add nothing to the counts.
  const bool small_chunks = (long)e->NC * e->wave_combine_below > (long)P;
  const bool counts_early = e->early_counts != 0 && !skip_combine && !dirty_only && e->n_classes_a > 0 && e->patch_chunks == 0 && !small_chunks && e->n_classes_b <= 16384;
  bool scattered = false;
  if (!skip_combine && !dirty_only && !counts_early) {
    if (zero_counts_beside) HIPCHK(hipStreamWaitEvent(st, e->ev_zero, 0));
    else HIPCHK(hipMemsetAsync(e->d_class_count.p, 0, (size_t)e->C * sizeof(int), st));
  }
  if (!skip_combine) {
    // threads per group of k_combine: the smallest whole number of waves (64/128/256) whose single pass covers a row
    int tpg = ykk::kBlock;
    while (tpg > ykk::kWave && (tpg / 2) * 2 >= e->row_stride) tpg /= 2;
    const int seg = tpg * ykk::kCombineUnroll * 2;
    // the full pass runs the class-by-class writer over the zone-B chunks only (the dirty-class pass over every chunk)
    // (chunks appended by ykpred_update_pods since the class build are not in the list: then every chunk runs, as the dirty pass does)
    const bool listed = !dirty_only && e->patch_chunks == 0;
    // (the short list — zone 0 alone — when both the run kernels and the fused rows are on, or no class is fused; else every zone-B chunk
    // goes through the writers' zone filter)
    const bool short_list = use_sweep && (fuse_on || e->fuse_count == 0);
    const int* chunk_list = listed ? (short_list ? e->d_chunk_list_b0 : e->d_chunk_list_b).as<int>() : nullptr;
    const int n_run = listed ? (short_list ? e->NCB0 : e->NCB) : e->NC;
    dim3 grid((unsigned)std::max(n_run, 1), (unsigned)((e->row_stride + seg - 1) / seg));
    // Both writers run on the launch stream, the band writer first. (Measured in round 3, profiles/r03_writer_knobs.txt: the
    // class-by-class writer BESIDE the band writer on a third stream upsets the one-workgroup-per-CU placement the band writer
    // lives on, 1.22 -> 1.45 ms, and even a 241-row zone B costs 0.25 ms through the fork / join; zone B first gains nothing.)
    hipStream_t sz = st;
    if (!dirty_only && e->n_classes_a > 0) {
      // zone A: class rows → table (and the classes' feasible counts), then the fill-pattern expansion over the band layout
      tm.begin(st);
      const int n_counted_b = counts_early ? e->n_classes_b : 0;
      hipLaunchKernelGGL(ykk::k_class_rows, dim3((unsigned)((e->n_classes_a + n_counted_b + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, st,
                         ct, pc, e->d_class_list_a.as<int>(), e->n_classes_a, e->row_words, e->row_stride, pin_on, e->d_class_rows_a.as<u64>(),
                         e->d_class_count.as<int>(), (const int*)nullptr, e->d_class_list_b.as<int>(), n_counted_b);
      tm.end(st, "k_class_rows");
      if (counts_early && want_dec) {
        // counts (this stream) and decisions (decision stream) are both final: per-ask outputs now, beside the band writer
        HIPCHK(hipEventRecord(e->ev_counts, st));
        HIPCHK(hipStreamWaitEvent(sb, e->ev_counts, 0));
        tm.begin(sb);
        hipLaunchKernelGGL(ykk::k_scatter, dim3((unsigned)((P + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, sb, P,
                           e->d_pod_class.as<int>(), e->d_class_count.as<int>(), e->d_class_best.as<int>(), e->d_key.as<u64>(),
                           want_cnt ? (int*)e->last_counts : nullptr, (int*)e->last_decisions, want_keys ? (i64*)e->last_keys : nullptr);
        tm.end(sb, "k_scatter");
        scattered = true;
      }
      const size_t lds_bytes = (size_t)2 * ykk::kBandClasses * (size_t)e->row_stride * sizeof(u64);
      if (lds_bytes > 64 * 1024)
        HIPCHK(hipFuncSetAttribute((const void*)ykk::k_expand_bands, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
      tm.begin(st);
      hipLaunchKernelGGL(ykk::k_expand_bands, dim3((unsigned)ykk::kBandGroups), dim3(ykk::kBandBlock), lds_bytes, st, bitmap,
                         e->d_class_rows_a.as<u64>(), e->d_band_tab.as<ykk::BandEntry>(), e->n_bands, e->row_stride);
      if (e->n_fix_rows > 0 && !counts_early)
        hipLaunchKernelGGL(ykk::k_fix_rows, dim3((unsigned)e->n_fix_rows), dim3(ykk::kBlock), 0, st, bitmap, e->d_class_rows_a.as<u64>(),
                           e->d_fix_row.as<int>(), e->d_fix_slot.as<int>(), e->n_fix_rows, e->row_stride);
      tm.end(st, "k_expand_bands");
    }
    if (use_sweep && e->run_classes > 0) {
      // the runs of ballot-row classes (k_class_runs): the staged request-value rows in LDS, one persistent workgroup (or two) per CU
      ykk::WalkStage rstage{};
      rstage.n = (int)e->h_stage_rows.size();
      for (int k = 0; k < rstage.n; ++k) rstage.row[k] = e->h_stage_rows[(size_t)k];
      const int its = (e->row_stride + ykk::kWave - 1) / ykk::kWave;
      int segs = (its + ykk::kWalkMaxIt - 1) / ykk::kWalkMaxIt, nit = (its + segs - 1) / segs;
      while (nit > 1 && ykk::runs_lds_bytes(rstage.n, nit) > (size_t)e->max_lds_bytes) {
        ++segs;
        nit = (its + segs - 1) / segs;
      }
      const int n_long = segs - (nit * segs - its);
      const size_t lds = ykk::runs_lds_bytes(rstage.n, nit);
      const int per_cu = lds * 2 <= (size_t)e->max_lds_bytes ? 2 : 1;
      const int groups = std::max(1, std::min(per_cu * e->num_cus, (e->run_units + ykk::kRunsWaves - 1) / ykk::kRunsWaves));
      tm.begin(sz);
      auto launch_runs = [&](auto kernel) -> int {
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3((unsigned)groups, (unsigned)segs), dim3(ykk::kRunsThreads), lds, sz, pc, e->d_run_classes.as<ykk::RunClass>(),
                           e->d_sweep_runs.as<ykk::SweepRun>(), e->run_classes, e->d_run_units.as<int>(), e->run_units, bitmap, e->row_words,
                           e->row_stride, pin_on, e->d_class_count.as<int>(), n_long, rstage);
        return YKPRED_OK;
      };
      switch (nit) {
        case 1: TRY(launch_runs(ykk::k_class_runs<1>)); break;
        case 2: TRY(launch_runs(ykk::k_class_runs<2>)); break;
        case 3: TRY(launch_runs(ykk::k_class_runs<3>)); break;
        case 4: TRY(launch_runs(ykk::k_class_runs<4>)); break;
        case 5: TRY(launch_runs(ykk::k_class_runs<5>)); break;
        case 6: TRY(launch_runs(ykk::k_class_runs<6>)); break;
        default: TRY(launch_runs(ykk::k_class_runs<7>)); break;
      }
      tm.end(sz, "k_class_runs");
    }
    if (sweep_rows_on) {
      // the sweep runs: one launch per walked dimension, the row's segments in grid.y (whole 64-word groups, at most kWalkMaxIt per
      // lane, what the LDS holds — 65 list dwords per word), one persistent workgroup per compute unit and segment claiming batches of rows
      const int its = (e->row_stride + ykk::kWave - 1) / ykk::kWave;
      int segs = (its + ykk::kWalkMaxIt - 1) / ykk::kWalkMaxIt, nit = (its + segs - 1) / segs;
      while (nit > 1 && ykk::sweep_lds_bytes(nit) > (size_t)e->max_lds_bytes) {
        ++segs;
        nit = (its + segs - 1) / segs;
      }
      const int n_long = segs - (nit * segs - its);
      tm.begin(sz);
      for (int b = 0; b < e->n_big; ++b) {
        const int n_rows = e->sweep_rows[b];
        if (n_rows == 0) continue;
        const int n_units = e->sweep_units[b];
        const int groups = std::max(1, std::min(e->sweep_groups > 0 ? e->sweep_groups : e->num_cus, (n_units + ykk::kSweepWaves - 1) / ykk::kSweepWaves));
        auto launch_sweep = [&](auto kernel) -> int {
          const size_t lds = ykk::sweep_lds_bytes(nit);
          if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
          hipLaunchKernelGGL(kernel, dim3((unsigned)groups, (unsigned)segs), dim3(ykk::kSweepThreads), lds, sz, pc,
                             e->d_ent_c.as<unsigned>() + (size_t)b * 65 * (size_t)e->row_words, e->d_rbits_c.as<u64>() + (size_t)b * ykk::kRankBits * (size_t)e->row_words,
                             e->d_sweep_rows.as<ykk::SweepRow>() + e->sweep_row_off[b], e->d_sweep_runs.as<ykk::SweepRun>(), n_rows,
                             e->d_sweep_units.as<int>() + e->sweep_unit_off[b], n_units, bitmap, e->row_words, e->row_stride, pin_on,
                             e->d_class_count.as<int>(), n_long);
          return YKPRED_OK;
        };
        switch (nit) {
          case 1: TRY(launch_sweep(ykk::k_sweep_rows<1>)); break;
          case 2: TRY(launch_sweep(ykk::k_sweep_rows<2>)); break;
          case 3: TRY(launch_sweep(ykk::k_sweep_rows<3>)); break;
          case 4: TRY(launch_sweep(ykk::k_sweep_rows<4>)); break;
          case 5: TRY(launch_sweep(ykk::k_sweep_rows<5>)); break;
          case 6: TRY(launch_sweep(ykk::k_sweep_rows<6>)); break;
          default: TRY(launch_sweep(ykk::k_sweep_rows<7>)); break;
        }
      }
      tm.end(sz, "k_sweep_rows");
    }
    if (fuse_on) {
      // the classes no run kernel takes, from their records: three compute waves of WPL words per lane cover a group of 192 * WPL
      // words (50 k nodes = 782 words: one group at WPL 5; a 6 250-node shard = 98 words: WPL 1)
      const int wpl = e->fuse_wpl >= 5 ? 5 : (e->fuse_wpl >= 2 ? 2 : (e->fuse_wpl == 1 ? 1 : (e->row_words > 384 ? 5 : (e->row_words > 192 ? 2 : 1))));
      const int gw = ykk::kFuseComputeWaves * ykk::kWave * wpl;
      const unsigned ygroups = (unsigned)((e->row_words + gw - 1) / gw);
      ykk::FuseSrc fsrc{e->planes_canon.as<u64>(), e->fuse_combos > 0 ? e->d_fuse_combo.as<u64>() : nullptr, {e->fam_res.base, e->fam_tol.base, e->fam_aff.base}};
      tm.begin(sz);
      auto launch_fused = [&](auto kernel, const ykk::FuseRec* recs, int n, u64* out, int* counts) {
        hipLaunchKernelGGL(kernel, dim3((unsigned)((n + ykk::kFuseRecsPerBlock - 1) / ykk::kFuseRecsPerBlock), ygroups), dim3(ykk::kBlock), 0, sz, recs, n, fsrc,
                           e->row_words, e->row_stride, out, counts);
      };
      auto launch_wide = [&](const ykk::FuseRec* recs, int n, u64* out, int* counts) {  // records that name up to eight rows
        if (wpl == 5) launch_fused(ykk::k_fused_rows<5, 8>, recs, n, out, counts);
        else if (wpl == 2) launch_fused(ykk::k_fused_rows<2, 8>, recs, n, out, counts);
        else launch_fused(ykk::k_fused_rows<1, 8>, recs, n, out, counts);
      };
      if (e->fuse_combos > 0) {
        launch_wide(e->d_fuse_combo_rec.as<ykk::FuseRec>(), e->fuse_combos, e->d_fuse_combo.as<u64>(), nullptr);
        if (wpl == 5) launch_fused(ykk::k_fused_rows<5, 2>, e->d_fuse_rec.as<ykk::FuseRec>(), e->fuse_count, bitmap, e->d_class_count.as<int>());
        else if (wpl == 2) launch_fused(ykk::k_fused_rows<2, 2>, e->d_fuse_rec.as<ykk::FuseRec>(), e->fuse_count, bitmap, e->d_class_count.as<int>());
        else launch_fused(ykk::k_fused_rows<1, 2>, e->d_fuse_rec.as<ykk::FuseRec>(), e->fuse_count, bitmap, e->d_class_count.as<int>());
      } else {
        launch_wide(e->d_fuse_rec.as<ykk::FuseRec>(), e->fuse_count, bitmap, e->d_class_count.as<int>());
      }
      tm.end(sz, "k_fused_rows");
    }
    tm.begin(sz);
    // Index rows to decode: k_walk_rows — a wave writes whole rows, the rank planes of the walked dimensions (56 bytes per word) staged
    // in LDS per workgroup; rows wider than kWalkMaxIt x 64 words go segment by segment (grid.y). Only where the device grants the LDS.
    // Segments of whole 64-word groups: as few as possible with at most kWalkMaxIt groups (register budget: 4 waves per SIMD) and an
    // LDS footprint that lets two workgroups share a CU; sizes differ by at most one — `walk_long` segments of walk_nit groups, then
    // the rest with walk_nit - 1; the last one holds the tail.
    int walk_nit = 0, walk_segs = 0, walk_long = 0;
    ykk::WalkStage wstage{};
    wstage.n = (int)e->h_stage_rows.size();
    for (int k = 0; k < wstage.n; ++k) wstage.row[k] = e->h_stage_rows[(size_t)k];
    // (with every run swept — sweep_min_run 1 — no chunk is left that k_walk_rows has a fast path for: what remains goes through the
    // listed chunk writers below, without the descriptor pass over all chunks)
    bool slices = small_chunks && e->combine_slices != 0 && pc.n_big > 0 && pc.res != nullptr && !(use_sweep && e->sweep_min_run <= 1 && listed);
    if (slices) {
      const int its = (e->row_stride + ykk::kWave - 1) / ykk::kWave;
      const size_t lds_cap = std::min((size_t)e->max_lds_bytes, (size_t)80 * 1024);
      walk_segs = (its + ykk::kWalkMaxIt - 1) / ykk::kWalkMaxIt;
      walk_nit = (its + walk_segs - 1) / walk_segs;
      while (walk_nit > 1 && ykk::walk_lds_bytes(pc.n_big, wstage.n, walk_nit) > lds_cap) {
        ++walk_segs;
        walk_nit = (its + walk_segs - 1) / walk_segs;
      }
      walk_long = walk_segs - (walk_nit * walk_segs - its);
      slices = ykk::walk_lds_bytes(pc.n_big, wstage.n, walk_nit) <= (size_t)e->max_lds_bytes && its * ykk::kWave <= e->idx_stride;
    }
    if (slices) {
      // chunk descriptors first, one thread per chunk (what a wave needs to know about a chunk, resolved once per pass)
      HIPCHK(e->d_slice_desc.ensure((size_t)std::max(e->NC, 1) * sizeof(ykk::SliceDesc)));
      HIPCHK(e->d_slice_general.ensure(sizeof(int)));
      HIPCHK(hipMemsetAsync(e->d_slice_general.p, 0, sizeof(int), sz));
      hipLaunchKernelGGL(ykk::k_slice_desc, dim3((unsigned)((e->NC + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, sz, ct, pc, e->NC, class_dirty,
                         pin_on, e->d_slice_desc.as<ykk::SliceDesc>(), e->d_slice_general.as<int>(), wstage);
      tm.end(sz, "k_slice_desc");
      tm.begin(sz);
      const int per_group = ykk::kRowsWaves * ykk::kWalkChunksPerWave;
      const unsigned gx = (unsigned)((e->NC + per_group - 1) / per_group);
      auto launch_walk = [&](auto kernel, int nit, int segs, int w_base) -> int {
        const size_t lds = ykk::walk_lds_bytes(pc.n_big, wstage.n, nit);
        if (lds > 64 * 1024) HIPCHK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(kernel, dim3(gx, (unsigned)segs), dim3(ykk::kRowsThreads), lds, sz, pc, e->d_slice_desc.as<ykk::SliceDesc>(), bitmap,
                           e->row_words, e->row_stride, pin_on, e->d_class_count.as<int>(), e->NC, w_base, wstage);
        return YKPRED_OK;
      };
      // (segs segments of nit groups from word w_base; the row's last segment is launched on its own with the tail predicate)
      auto launch_segments = [&](int nit, int segs, int w_base, bool tail) -> int {
        if (segs <= 0) return YKPRED_OK;
        switch (nit * 2 + (tail ? 1 : 0)) {
#define YK_WALK_CASE(N)                                                         \
  case (N) * 2: return launch_walk(ykk::k_walk_rows<(N), false>, nit, segs, w_base); \
  case (N) * 2 + 1: return launch_walk(ykk::k_walk_rows<(N), true>, nit, segs, w_base);
          YK_WALK_CASE(1) YK_WALK_CASE(2) YK_WALK_CASE(3) YK_WALK_CASE(4) YK_WALK_CASE(5) YK_WALK_CASE(6) YK_WALK_CASE(7)
#undef YK_WALK_CASE
        }
        return fail(e, YKPRED_E_INVALID, "k_walk_rows: no kernel for " + std::to_string(nit) + " word groups per lane");
      };
      {
        const int n_short = walk_segs - walk_long, nit_last = n_short > 0 ? walk_nit - 1 : walk_nit;
        const int long_plain = n_short > 0 ? walk_long : walk_long - 1, short_plain = n_short > 0 ? n_short - 1 : 0;
        TRY(launch_segments(walk_nit, long_plain, 0, false));
        TRY(launch_segments(walk_nit - 1, short_plain, walk_long * walk_nit * ykk::kWave, false));
        TRY(launch_segments(nit_last, 1, (long_plain * walk_nit + short_plain * (walk_nit - 1)) * ykk::kWave, true));
      }
      tm.end(sz, "k_walk_rows");
      tm.begin(sz);
      // chunks the fast path does not cover (several member rows, pins to unknown nodes, other row shapes): wave per chunk
      hipLaunchKernelGGL(ykk::k_combine_wave, dim3((unsigned)((e->NC + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, sz, ct,
                         pc, bitmap, e->row_words, e->row_stride, pin_on, e->d_class_count.as<int>(), e->NC, class_dirty,
                         e->d_slice_desc.as<ykk::SliceDesc>(), e->d_slice_general.as<int>(), (const int*)nullptr);
    } else if (counts_early) {
      // one launch behind the band writer: the zone-B chunks (their counts are known: none added here) and, in further workgroups,
      // the straddling rows of the band layout (k_fix_rows' work)
      const ykk::FixRows fix{e->d_class_rows_a.as<u64>(), e->d_fix_row.as<int>(), e->d_fix_slot.as<int>(), e->n_fix_rows};
      const int blocks = n_run + e->n_fix_rows;
      if (blocks > 0)
        hipLaunchKernelGGL(ykk::k_combine, dim3((unsigned)blocks, grid.y), dim3(ykk::kBlock), (size_t)e->combine_lds_bytes, sz, ct, pc, bitmap, e->row_words,
                           e->row_stride, pin_on, (int*)nullptr, tpg, class_dirty, chunk_list, fix, n_run);
    } else if (n_run == 0) {
      // (no chunk outside the band layout: nothing to launch)
    } else if (small_chunks) {
      // few members per chunk: one wave per chunk (see k_combine_wave)
      const dim3 wgrid((unsigned)((std::max(n_run, 1) + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock));
      hipLaunchKernelGGL(ykk::k_combine_wave, wgrid, dim3(ykk::kBlock), 0, sz, ct, pc, bitmap, e->row_words, e->row_stride, pin_on,
                         e->d_class_count.as<int>(), n_run, class_dirty, (const ykk::SliceDesc*)nullptr, (const int*)nullptr, chunk_list);
    } else {
      // dynamic LDS is requested only to cap the blocks resident per CU (see combine_lds_bytes)
      hipLaunchKernelGGL(ykk::k_combine, grid, dim3(ykk::kBlock), (size_t)e->combine_lds_bytes, sz, ct, pc, bitmap, e->row_words, e->row_stride,
                         pin_on, e->d_class_count.as<int>(), tpg, class_dirty, chunk_list, ykk::FixRows{nullptr, nullptr, nullptr, 0}, n_run);
    }
    tm.end(sz, dirty_only ? "k_combine(dirty classes)" : "k_combine");
  }
  if (want_dec) {
    HIPCHK(hipEventRecord(e->ev_join, sb));
    HIPCHK(hipStreamWaitEvent(st, e->ev_join, 0));
  }
  if ((want_dec || want_cnt) && !scattered) {
    tm.begin(st);
    hipLaunchKernelGGL(ykk::k_scatter, dim3((unsigned)((P + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, P,
                       e->d_pod_class.as<int>(), e->d_class_count.as<int>(), e->d_class_best.as<int>(), e->d_key.as<u64>(),
                       want_cnt ? (int*)e->last_counts : nullptr, want_dec ? (int*)e->last_decisions : nullptr,
                       want_keys ? (i64*)e->last_keys : nullptr);
    tm.end(st, "k_scatter");
  }
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
  };
  const bool graphable = !tm.on && !e->graph_disabled && !(e->comm && e->comm_world > 1) && !(a->options & (YKPRED_EVAL_DIRECT | YKPRED_EVAL_SPREAD_COUNT_ONLY));
  hipGraphExec_t exec = nullptr;
  if (graphable) {
    ykpred_engine::GraphKey key{e->tables_version, pre, filt, a->options, (void*)bitmap, e->last_counts, e->last_decisions, e->last_keys};
    for (size_t i = 0; i < e->graphs.size();) {
      if (e->graphs[i].first.version != e->tables_version) {
        (void)hipGraphExecDestroy(e->graphs[i].second);
        e->graphs.erase(e->graphs.begin() + (long)i);
      } else {
        if (e->graphs[i].first == key) exec = e->graphs[i].second;
        ++i;
      }
    }
    bool repeated = false;
    for (size_t i = 0; i < e->seen.size();) {
      if (e->seen[i].version != e->tables_version) {
        e->seen.erase(e->seen.begin() + (long)i);
        continue;
      }
      repeated = repeated || e->seen[i] == key;
      ++i;
    }
    if (!repeated && e->seen.size() < 8) e->seen.push_back(key);
    if (!exec && repeated) {
      hipGraph_t g = nullptr;
      int crc = YKPRED_E_DEVICE;
      if (hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal) == hipSuccess) {
        crc = enqueue();
        if (hipStreamEndCapture(st, &g) != hipSuccess) crc = YKPRED_E_DEVICE;
      }
      if (crc == YKPRED_OK && g && hipGraphInstantiate(&exec, g, nullptr, nullptr, 0) == hipSuccess) {
        if (e->graphs.size() >= 8) {
          (void)hipGraphExecDestroy(e->graphs.front().second);
          e->graphs.erase(e->graphs.begin());
        }
        e->graphs.emplace_back(key, exec);
      } else {
        exec = nullptr;
        e->graph_disabled = true;  // plain launches from now on
        (void)hipGetLastError();
      }
      if (g) (void)hipGraphDestroy(g);
    }
  }
  if (exec) {
    HIPCHK(hipGraphLaunch(exec, st));
  } else {
    int rc = enqueue();
    if (rc < 0) return rc;
    if (rc == 1) {
      tm.done(st);
      return YKPRED_OK;
    }
  }
  tm.done(st);
  if (e->ev_eval_done) (void)hipEventRecord(e->ev_eval_done, st);
  e->bitmap_epoch = e->nodes_epoch;
  e->last_pre = pre;
  e->last_filt = filt;
  e->last_eval_valid = true;
  if (!(a->options & (YKPRED_EVAL_SKIP_BITMAP | YKPRED_EVAL_DIRTY_CLASSES))) e->n_full_evals++;
  if (!(a->options & (YKPRED_EVAL_SKIP_BITMAP | YKPRED_EVAL_DIRTY_CLASSES))) std::fill(e->h_row_stale.begin(), e->h_row_stale.end(), 0);
  if (want_dec) {
    e->win_partial = use_run_decide_pass;
    e->rank_valid = true;
    e->ranked_specs_version = e->specs_version;
    e->ranked_nodes_epoch = e->nodes_epoch;
    e->ranked_pre = pre;
    e->ranked_filt = filt;
    e->ranked_has_first = e->decide_skip;
  }
  e->last_has_keys = want_keys;
  return YKPRED_OK;
}

int32_t ykpred_eval_nodes(ykpred_engine_t* e, const ykpred_eval_args_t* a, int32_t num_nodes, const int32_t* node_index) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:eval_nodes");
  if (e) e->n_node_patches++;
  if (!e || !a || num_nodes < 0 || (num_nodes > 0 && !node_index)) return fail(e, YKPRED_E_INVALID, "eval_nodes: bad argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = a->stream ? (hipStream_t)a->stream : e->own_stream;
  const unsigned pre = a->prefilter_plugins, filt = a->filter_plugins;
  u64* bitmap = a->bitmap ? (u64*)a->bitmap : e->d_bitmap.as<u64>();
  if (e->classes_dirty || !e->last_eval_valid || pre != e->last_pre || filt != e->last_filt || (void*)bitmap != e->last_bitmap)
    return fail(e, YKPRED_E_STATE, "eval_nodes: no matching previous ykpred_eval (tables, plugin lists or bitmap changed)");
  const bool topo = (pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY)) && e->fam_spread.D > 0;
  // On a node-sharded engine the topology histograms couple the shards: with a communicator attached the call is COLLECTIVE
  // when topology signatures are active — every shard enters it (with its own, possibly empty, node list) and performs exactly
  // the histogram exchange a full ykpred_eval performs (SUM of matches, MAX of "domain present"), so shards may even mix the
  // two calls. Hosts that move the histograms themselves use the same two-phase flags as ykpred_eval
  // (YKPRED_EVAL_SPREAD_COUNT_ONLY, then YKPRED_EVAL_SPREAD_COUNTS_READY).
  const bool sharded = e->comm && e->comm_world > 1;
  const bool hist_ready = a->options & YKPRED_EVAL_SPREAD_COUNTS_READY, hist_only = a->options & YKPRED_EVAL_SPREAD_COUNT_ONLY;
  const bool topo_pass = topo && e->P > 0 && (num_nodes > 0 || sharded || hist_ready || hist_only);
  if (topo && e->spread_dirty) return fail(e, YKPRED_E_STATE, "eval_nodes: topology tables changed — run ykpred_eval");
  for (int i = 0; i < num_nodes; ++i)
    if (node_index[i] < 0 || node_index[i] >= e->N) return fail(e, YKPRED_E_INVALID, "eval_nodes: node index out of range");
  Timer tm{e, (a->options & YKPRED_EVAL_PROFILE) != 0};
  tm.start(st);
  const bool want_cnt = a->options & YKPRED_OUT_COUNTS;
  const bool want_keys = a->options & YKPRED_OUT_DECISION_KEYS;
  const bool want_dec = (a->options & YKPRED_OUT_DECISIONS) || want_keys;
  if (topo_pass) {
    // The PreFilter state of the topology plugins couples all nodes: rebuild the histograms (cheap: one thread per
    // signature and node), then find the signatures whose cells or minima moved.
    const size_t cells = (size_t)std::max<int64_t>(e->spread_cells, 1) * sizeof(int), mins = (size_t)std::max(e->spread_constraints, 1) * sizeof(int);
    HIPCHK(e->d_sp_cnt_prev.ensure(cells));
    HIPCHK(e->d_sp_present_prev.ensure(cells));
    HIPCHK(e->d_sp_min_prev.ensure(mins));
    HIPCHK(e->d_sig_changed.ensure((size_t)e->fam_spread.D * sizeof(int)));
    if (e->hist_epoch == 0) return fail(e, YKPRED_E_STATE, "eval_nodes: no topology histograms from a previous ykpred_eval");
    if (!hist_ready) {
      HIPCHK(hipMemcpyAsync(e->d_sp_cnt_prev.p, e->d_sp_cnt.p, cells, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(e->d_sp_present_prev.p, e->d_sp_present.p, cells, hipMemcpyDeviceToDevice, st));
      HIPCHK(hipMemcpyAsync(e->d_sp_min_prev.p, e->d_sp_min.p, mins, hipMemcpyDeviceToDevice, st));
      TRY(run_spread_prefilter(e, st, &tm, true, false));
    }
    if (hist_only) {  // the caller sums layout.spread_counts / spread_present over the shards and comes back with COUNTS_READY
      tm.done(st);
      return YKPRED_OK;
    }
    if (!hist_ready) TRY(allreduce_spread(e, st));
    TRY(run_spread_prefilter(e, st, &tm, false, true));
    e->hist_epoch = e->nodes_epoch;
    tm.begin(st);
    hipLaunchKernelGGL(ykk::k_spread_diff, dim3((unsigned)e->fam_spread.D), dim3(ykk::kWave), 0, st, spread_sigs(e), e->d_sp_cnt_prev.as<int>(),
                       e->d_sp_present_prev.as<int>(), e->d_sp_min_prev.as<int>(), e->d_sig_changed.as<int>());
    tm.end(st, "k_spread_diff");
  }
  if (num_nodes > 0 && e->P > 0) {
    // unique nodes, grouped by bitmap word, at most kMaxColGroups words / 64 nodes per launch
    std::vector<int32_t> nodes(node_index, node_index + num_nodes);
    std::sort(nodes.begin(), nodes.end());
    nodes.erase(std::unique(nodes.begin(), nodes.end()), nodes.end());
    HIPCHK(e->d_class_word.ensure((size_t)std::max(e->C, 1) * ykk::kMaxColGroups * sizeof(u64)));
    ykk::NodeTable nt = node_table(e);
    ykk::SpecTable stbl = spec_table(e);
    size_t i = 0;
    while (i < nodes.size()) {
      ykk::ColumnGroups cg{};
      int filled = 0;
      while (i < nodes.size() && cg.n_groups < ykk::kMaxColGroups && filled < 64) {
        int w = nodes[i] >> 6;
        cg.word[cg.n_groups] = w;
        cg.first[cg.n_groups] = filled;
        while (i < nodes.size() && (nodes[i] >> 6) == w && filled < 64) cg.nodes[filled++] = nodes[i++];
        cg.n_groups++;
        if (i < nodes.size() && (nodes[i] >> 6) == w) break;  // word split across launches: its second half goes next
      }
      cg.first[cg.n_groups] = filled;
      tm.begin(st);
      hipLaunchKernelGGL(ykk::k_column_class, dim3((unsigned)((e->C + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, nt, stbl, cg,
                         e->C, e->d_class_first.as<int>(), e->d_class_pin.as<int>(), e->d_pod_spec.as<int>(), e->d_pod_row.as<int>(), pre, filt, bitmap,
                         e->row_stride, e->d_class_word.as<u64>(), e->d_class_count.as<int>());
      tm.end(st, "k_column_class");
      tm.begin(st);
      hipLaunchKernelGGL(ykk::k_column_patch, dim3((unsigned)((e->P + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, cg, e->P,
                         e->d_pod_class.as<int>(), e->d_pod_row.as<int>(), e->d_class_word.as<u64>(), e->d_class_count.as<int>(), bitmap, e->row_stride,
                         want_cnt ? (int*)(a->counts ? a->counts : e->last_counts) : nullptr);
      tm.end(st, "k_column_patch");
    }
  }
  HIPCHK(hipGetLastError());
  tm.done(st);
  if (e->ev_eval_done) (void)hipEventRecord(e->ev_eval_done, st);
  e->bitmap_epoch = e->nodes_epoch;  // the caller lists every node it changed since the last evaluation
  if (topo_pass) {
    // planes again (cheap), then the whole rows of the classes whose topology signature moved; with decisions requested the
    // same pass refreshes the bin-pack order and the class decisions
    ykpred_eval_args_t b = *a;
    b.bitmap = bitmap;
    b.options = (a->options & ~(uint32_t)YKPRED_EVAL_SPREAD_COUNT_ONLY) | YKPRED_OUT_BITMAP | YKPRED_EVAL_DIRTY_CLASSES | YKPRED_EVAL_SPREAD_COUNTS_READY;
    int rc = ykpred_eval(e, &b);
    if (rc != YKPRED_OK) return rc;
  } else if (want_dec) {
    // the bin-pack order moved with the node's Requested: rerun the (cheap) plane + decision kernels, not the bitmap
    ykpred_eval_args_t b = *a;
    b.options = (a->options & ~(uint32_t)YKPRED_OUT_BITMAP) | YKPRED_EVAL_SKIP_BITMAP;
    int rc = ykpred_eval(e, &b);
    if (rc != YKPRED_OK) return rc;
  }
  return YKPRED_OK;
}

// Row-level maintenance of the ask table. See ykpred.h.
int32_t ykpred_update_pods(ykpred_engine_t* e, int32_t num_pods_after, int32_t count, const int32_t* rows, const int32_t* spec_index,
                           const int32_t* node_name_index) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:update_asks");
  if (e) e->n_uploads++;
  if (e) e->tables_version++;
  if (e) e->layout_version++;
  if (e) e->ask_epoch++;
  if (!e || num_pods_after < 0 || count < 0 || (count > 0 && (!rows || !spec_index || !node_name_index)))
    return fail(e, YKPRED_E_INVALID, "update_pods: bad argument");
  if (!e->pods_set || !e->specs_set) return fail(e, YKPRED_E_STATE, "update_pods: ykpred_set_specs and ykpred_set_pods come first");
  HIPCHK(hipSetDevice(e->cfg.device));
  const int oldP = e->P, newP = num_pods_after;
  {
    std::vector<uint8_t> listed((size_t)std::max(newP - oldP, 0), 0);
    std::vector<int32_t> sorted(rows, rows + count);
    std::sort(sorted.begin(), sorted.end());
    if (std::adjacent_find(sorted.begin(), sorted.end()) != sorted.end()) return fail(e, YKPRED_E_INVALID, "update_pods: a row is listed twice");
    for (int i = 0; i < count; ++i) {
      if (rows[i] < 0 || rows[i] >= newP) return fail(e, YKPRED_E_INVALID, "update_pods: row out of range");
      if (spec_index[i] < 0 || spec_index[i] >= e->S) return fail(e, YKPRED_E_INVALID, "update_pods: pod references a spec that was not uploaded");
      if (node_name_index[i] < YKPRED_UNKNOWN_NODE_NAME || (e->nodes_set && node_name_index[i] >= e->N))
        return fail(e, YKPRED_E_INVALID, "update_pods: node_name_index out of range");
      if (rows[i] >= oldP) listed[(size_t)(rows[i] - oldP)] = 1;
    }
    for (uint8_t l : listed)
      if (!l) return fail(e, YKPRED_E_INVALID, "update_pods: every appended row must be listed");
  }
  hipStream_t st = e->own_stream;
  if (!e->classes_dirty) {
    // Every listed row takes a fresh physical row at the end of the bitmap. If those do not fit the row capacity the shards
    // agreed on — or the caller-owned bitmap of the last evaluation — nothing is patched in place: the class index is dropped
    // BEFORE any mirror is touched and the call takes the table-only branch below; the next ykpred_eval re-packs the rows.
    const int64_t after = (int64_t)e->rows_total + count;
    const bool caller_owned = e->last_bitmap && e->last_bitmap != e->d_bitmap.p;
    if ((e->row_capacity && after > (int64_t)e->row_capacity) || (caller_owned && e->last_eval_valid && after > e->last_bitmap_rows)) {
      e->classes_dirty = true;
      e->last_eval_valid = false;
    }
  }
  if (e->classes_dirty) {
    // no class index yet (or it is due for a rebuild): only the pod table changes, the next ykpred_eval builds the classes
    e->h_pod_spec.resize((size_t)newP);
    e->h_pod_pin.resize((size_t)newP);
    for (int i = 0; i < count; ++i) {
      e->h_pod_spec[(size_t)rows[i]] = spec_index[i];
      e->h_pod_pin[(size_t)rows[i]] = node_name_index[i];
    }
    e->P = newP;
    TRY(upload(e, e->d_pod_spec, e->h_pod_spec.data(), e->h_pod_spec.size(), st));
    TRY(upload(e, e->d_pod_pin, e->h_pod_pin.data(), e->h_pod_pin.size(), st));
    HIPCHK(hipStreamSynchronize(st));
    return YKPRED_OK;
  }

  enum { T_POD_SPEC, T_POD_PIN, T_POD_CLASS, T_MEMBERS, T_CH_CLASS, T_CH_BEGIN, T_CH_LEN, T_CH_FIRST, T_CLASS_SIG, T_CLASS_PIN, T_CLASS_FIRST, T_POD_ROW, T_CH_ZONE };
  std::vector<ykk::TablePatch> patches;
  auto put = [&](int table, int index, int value) { patches.push_back({table, index, value, 0}); };
  std::vector<int32_t> orphaned;  // classes whose representative row left
  auto leave = [&](int p) {
    const int c = e->h_pod_class[(size_t)p], slot = e->h_pod_slot[(size_t)p];
    e->h_members[(size_t)slot] = -1;
    put(T_MEMBERS, slot, -1);
    if ((size_t)c < e->h_class_sweep.size() && (e->h_class_sweep[(size_t)c] || e->h_class_fused[(size_t)c])) e->sweep_ready = e->fuse_ready = false;  // (its row list names a row that is gone: the chunk writers take the runs until the next class build)
    e->h_class_live[(size_t)c]--;
    if (e->h_class_first[(size_t)c] == p) {
      e->h_class_first[(size_t)c] = -1;
      orphaned.push_back(c);
    }
  };
  for (int p = newP; p < oldP; ++p) leave(p);  // truncated rows
  for (int i = 0; i < count; ++i)
    if (rows[i] < oldP) leave(rows[i]);
  e->h_pod_spec.resize((size_t)newP);
  e->h_pod_pin.resize((size_t)newP);
  e->h_pod_class.resize((size_t)newP);
  e->h_pod_slot.resize((size_t)newP);
  e->h_pod_row.resize((size_t)newP, -1);
  const int cm = e->chunk_members;
  for (int i = 0; i < count; ++i) {
    const int p = rows[i], sp = spec_index[i], pin = node_name_index[i];
    e->h_pod_spec[(size_t)p] = sp;
    e->h_pod_pin[(size_t)p] = pin;
    put(T_POD_SPEC, p, sp);
    put(T_POD_PIN, p, pin);
    ClassKey k{e->spec_sig_res[(size_t)sp], e->spec_sig_tol[(size_t)sp], e->spec_sig_aff[(size_t)sp], e->spec_sig_spread[(size_t)sp], pin};
    auto it = e->class_ids.find(k);
    int c;
    if (it == e->class_ids.end()) {
      c = e->C++;
      e->class_ids.emplace(k, c);
      e->h_class_sig.insert(e->h_class_sig.end(), {k.a, k.b, k.c, k.d});
      e->h_class_pin.push_back(pin);
      e->h_class_first.push_back(-1);
      e->h_class_live.push_back(0);
      e->h_class_slot_a.push_back(-1);
      e->h_class_chunks.emplace_back();
      put(T_CLASS_SIG, c * 4 + 0, k.a);
      put(T_CLASS_SIG, c * 4 + 1, k.b);
      put(T_CLASS_SIG, c * 4 + 2, k.c);
      put(T_CLASS_SIG, c * 4 + 3, k.d);
      put(T_CLASS_PIN, c, pin);
    } else {
      c = it->second;
    }
    // a row of its own at the end of the bitmap: the row it had may lie inside another class's band interval (zone A), and
    // rows are never reused between two class builds
    const int row = e->rows_total++;
    e->h_pod_row[(size_t)p] = row;
    put(T_POD_ROW, p, row);
    const int slot = (int)e->h_members.size();
    e->h_members.push_back(p);
    put(T_MEMBERS, slot, row);
    auto& cc = e->h_class_chunks[(size_t)c];
    const int tail = cc.empty() ? -1 : cc.back();
    if (tail >= 0 && e->h_ch_begin[(size_t)tail] + e->h_ch_len[(size_t)tail] == slot && e->h_ch_len[(size_t)tail] < cm) {
      e->h_ch_len[(size_t)tail]++;  // consecutive appends to one class share a chunk
      put(T_CH_LEN, tail, e->h_ch_len[(size_t)tail]);
    } else {
      const int ch = e->NC++;
      e->h_ch_class.push_back(c);
      e->h_ch_begin.push_back(slot);
      e->h_ch_len.push_back(1);
      e->h_ch_first.push_back(cc.empty() ? 1 : 0);  // the first chunk of a class adds the class's feasible count
      e->h_ch_zone.push_back(0);                     // appended rows lie behind both zones: the chunk kernel writes them
      put(T_CH_ZONE, ch, 0);
      put(T_CH_CLASS, ch, c);
      put(T_CH_BEGIN, ch, slot);
      put(T_CH_LEN, ch, 1);
      put(T_CH_FIRST, ch, cc.empty() ? 1 : 0);
      cc.push_back(ch);
      e->patch_chunks++;
    }
    e->h_pod_class[(size_t)p] = c;
    e->h_pod_slot[(size_t)p] = slot;
    put(T_POD_CLASS, p, c);
    e->h_class_live[(size_t)c]++;
    if (e->h_class_first[(size_t)c] < 0) e->h_class_first[(size_t)c] = p;
  }
  // Representative rows. k_column_class reads the class's bitmap row from its representative, so that row must be a LIVE
  // member whose bitmap row is current: not one rewritten by this (or an earlier, not yet evaluated) update. Only when
  // every member of a class is stale may a stale row represent it — then all of its rows are rebuilt by ykpred_eval_pods.
  e->h_row_stale.resize((size_t)newP, 0);
  std::vector<int32_t> touched(orphaned);
  for (int i = 0; i < count; ++i) {
    e->h_row_stale[(size_t)rows[i]] = 1;
    touched.push_back(e->h_pod_class[(size_t)rows[i]]);
  }
  std::sort(touched.begin(), touched.end());
  touched.erase(std::unique(touched.begin(), touched.end()), touched.end());
  for (int c : touched) {
    const int first = e->h_class_first[(size_t)c];
    int want = first;
    if (e->h_class_live[(size_t)c] == 0) {
      want = -1;
    } else if (first < 0 || e->h_row_stale[(size_t)first]) {
      int fresh = -1, any = -1;
      for (int ch : e->h_class_chunks[(size_t)c]) {
        for (int j = 0; j < e->h_ch_len[(size_t)ch] && fresh < 0; ++j) {
          const int m = e->h_members[(size_t)(e->h_ch_begin[(size_t)ch] + j)];
          if (m < 0) continue;
          if (any < 0) any = m;
          if (!e->h_row_stale[(size_t)m]) fresh = m;
        }
        if (fresh >= 0) break;
      }
      want = fresh >= 0 ? fresh : any;
    }
    e->h_class_first[(size_t)c] = want;
    put(T_CLASS_FIRST, c, want);
  }
  e->P = newP;
  TRY(grow_owned_outputs(e));

  // device side: grow what has to grow (contents kept), then apply every change with one copy + one launch
  const size_t I = sizeof(int32_t);
  HIPCHK(e->d_pod_spec.reserve_keep((size_t)newP * I, (size_t)oldP * I));
  HIPCHK(e->d_pod_pin.reserve_keep((size_t)newP * I, (size_t)oldP * I));
  HIPCHK(e->d_pod_class.reserve_keep((size_t)newP * I, (size_t)oldP * I));
  HIPCHK(e->d_pod_row.reserve_keep((size_t)newP * I, (size_t)oldP * I));
  HIPCHK(e->d_members.reserve_keep(e->h_members.size() * I, e->h_members.size() * I));
  for (DevBuf* b : {&e->d_chunk_class, &e->d_chunk_begin, &e->d_chunk_len, &e->d_chunk_first, &e->d_chunk_zone})
    HIPCHK(b->reserve_keep((size_t)e->NC * I, (size_t)e->NC * I));
  HIPCHK(e->d_class_sig.reserve_keep((size_t)e->C * 4 * I, (size_t)e->C * 4 * I));
  for (DevBuf* b : {&e->d_class_pin, &e->d_class_first, &e->d_class_count, &e->d_class_best}) HIPCHK(b->reserve_keep((size_t)e->C * I, (size_t)e->C * I));
  if (!patches.empty()) {
    // one thread per patch: several patches of one (table, index) — a chunk length that grows member by member within this
    // call — must not race; only the LAST value of each cell is uploaded
    {
      std::unordered_map<uint64_t, size_t> last;
      last.reserve(patches.size());
      for (size_t i = 0; i < patches.size(); ++i) last[((uint64_t)(uint32_t)patches[i].table << 32) | (uint32_t)patches[i].index] = i;
      if (last.size() != patches.size()) {
        std::vector<ykk::TablePatch> unique;
        unique.reserve(last.size());
        for (size_t i = 0; i < patches.size(); ++i)
          if (last[((uint64_t)(uint32_t)patches[i].table << 32) | (uint32_t)patches[i].index] == i) unique.push_back(patches[i]);
        patches.swap(unique);
      }
    }
    ykk::TablePtrs tp{};
    tp.t[T_POD_SPEC] = e->d_pod_spec.as<int>();
    tp.t[T_POD_PIN] = e->d_pod_pin.as<int>();
    tp.t[T_POD_CLASS] = e->d_pod_class.as<int>();
    tp.t[T_MEMBERS] = e->d_members.as<int>();
    tp.t[T_CH_CLASS] = e->d_chunk_class.as<int>();
    tp.t[T_CH_BEGIN] = e->d_chunk_begin.as<int>();
    tp.t[T_CH_LEN] = e->d_chunk_len.as<int>();
    tp.t[T_CH_FIRST] = e->d_chunk_first.as<int>();
    tp.t[T_CLASS_SIG] = e->d_class_sig.as<int>();
    tp.t[T_CLASS_PIN] = e->d_class_pin.as<int>();
    tp.t[T_CLASS_FIRST] = e->d_class_first.as<int>();
    tp.t[T_POD_ROW] = e->d_pod_row.as<int>();
    tp.t[T_CH_ZONE] = e->d_chunk_zone.as<int>();
    TRY(upload(e, e->d_patches, patches.data(), patches.size(), st));
    hipLaunchKernelGGL(ykk::k_apply_patches, dim3((unsigned)((patches.size() + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, tp,
                       (int)patches.size(), e->d_patches.as<ykk::TablePatch>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));  // `patches` is pageable host memory
  }
  // many small chunks make k_combine re-read planes for little output: rebuild the classes at the next full evaluation
  if (e->patch_chunks > std::max(4096, (e->NC - e->patch_chunks) / 4) || e->rows_total > newP + std::max(8192, newP / 4)) e->classes_dirty = true;
  return YKPRED_OK;
}

int32_t ykpred_eval_pods(ykpred_engine_t* e, const ykpred_eval_args_t* a, int32_t num_rows, const int32_t* rows) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:eval_pods");
  if (e) e->n_row_patches++;
  if (!e || !a || num_rows < 0 || (num_rows > 0 && !rows)) return fail(e, YKPRED_E_INVALID, "eval_pods: bad argument");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = a->stream ? (hipStream_t)a->stream : e->own_stream;
  const unsigned pre = a->prefilter_plugins, filt = a->filter_plugins;
  const bool own_bitmap = !a->bitmap && e->last_bitmap == e->d_bitmap.p;
  if (e->classes_dirty || !e->last_eval_valid || pre != e->last_pre || filt != e->last_filt || (a->bitmap && a->bitmap != e->last_bitmap) ||
      (!a->bitmap && !own_bitmap))
    return fail(e, YKPRED_E_STATE, "eval_pods: no matching previous ykpred_eval (tables, plugin lists or bitmap changed)");
  const bool want_cnt = a->options & YKPRED_OUT_COUNTS;
  const bool want_keys = a->options & YKPRED_OUT_DECISION_KEYS;
  const bool want_dec = (a->options & YKPRED_OUT_DECISIONS) || want_keys;
  if (want_dec && !e->rank_valid) return fail(e, YKPRED_E_STATE, "eval_pods: the bin-pack order is stale — run ykpred_eval with decisions");
  for (int i = 0; i < num_rows; ++i)
    if (rows[i] < 0 || rows[i] >= e->P) return fail(e, YKPRED_E_INVALID, "eval_pods: row out of range");
  TRY(grow_owned_outputs(e));
  Timer tm{e, (a->options & YKPRED_EVAL_PROFILE) != 0};
  tm.start(st);
  if (num_rows > 0 && e->N > 0) {
    u64* bitmap = (u64*)e->last_bitmap;
    TRY(upload(e, e->d_rows, rows, (size_t)num_rows, st));
    HIPCHK(e->d_row_count.ensure((size_t)num_rows * sizeof(int)));
    HIPCHK(e->d_row_best.ensure((size_t)num_rows * sizeof(int)));
    HIPCHK(hipMemsetAsync(e->d_row_count.p, 0, (size_t)num_rows * sizeof(int), st));
    HIPCHK(hipMemsetAsync(e->d_row_best.p, 0x7f, (size_t)num_rows * sizeof(int), st));  // 0x7f7f7f7f > any rank; mapped to "none" below
    ykk::NodeTable nt = node_table(e);
    ykk::SpecTable stbl = spec_table(e);
    const unsigned wgroups = (unsigned)((e->row_stride + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock);
    tm.begin(st);
    hipLaunchKernelGGL(ykk::k_rows, dim3((unsigned)num_rows, wgroups), dim3(ykk::kBlock), 0, st, nt, stbl, num_rows, e->d_rows.as<int>(),
                       e->d_pod_spec.as<int>(), e->d_pod_pin.as<int>(), e->d_pod_row.as<int>(), pre, filt, bitmap, e->row_words, e->row_stride,
                       want_dec ? e->d_rank.as<int>() : nullptr, e->d_row_count.as<int>(), e->d_row_best.as<int>());
    tm.end(st, "k_rows");
    tm.begin(st);
    hipLaunchKernelGGL(ykk::k_rows_finish, dim3((unsigned)((num_rows + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, num_rows,
                       e->d_rows.as<int>(), e->d_pod_class.as<int>(), e->d_row_count.as<int>(), e->d_row_best.as<int>(),
                       want_dec ? e->d_perm.as<int>() : nullptr, e->d_key.as<u64>(), e->d_class_count.as<int>(), e->d_class_best.as<int>(),
                       want_cnt ? (int*)(a->counts ? a->counts : e->last_counts) : nullptr,
                       (a->options & YKPRED_OUT_DECISIONS) ? (int*)(a->decisions ? a->decisions : e->last_decisions) : nullptr,
                       want_keys ? (i64*)(a->decision_keys ? a->decision_keys : e->last_keys) : nullptr);
    tm.end(st, "k_rows_finish");
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(st));  // `rows` was staged from caller memory
    for (int i = 0; i < num_rows; ++i)
      if ((size_t)rows[i] < e->h_row_stale.size()) e->h_row_stale[(size_t)rows[i]] = 0;
  }
  tm.done(st);
  if (e->ev_eval_done) (void)hipEventRecord(e->ev_eval_done, st);
  return YKPRED_OK;
}

// Conflict-resolved decisions for a sequence of asks. See ykpred.h.
int32_t ykpred_set_spec_effects(ykpred_engine_t* e, const ykpred_spec_effects_t* fx) {
  YK_SERIALISE(e);
  if (!e) return fail(e, YKPRED_E_INVALID, "set_spec_effects: bad argument");
  if (!fx) {  // drop them: rounds with topology constraints / host-port asks answer YKPRED_E_UNSUPPORTED again
    e->fx_version = 0;
    return YKPRED_OK;
  }
  if (!e->specs_set || fx->count != e->S) return fail(e, YKPRED_E_STATE, "set_spec_effects: the effects must describe the uploaded spec table (same count)");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  const size_t S = (size_t)e->S;
  e->fx_version = 0;
  e->fx_contrib = false;
  e->fx_ports = false;
  e->h_fx_off.clear();
  e->h_fx_cls.clear();
  if (fx->contrib_off) {
    const int total = fx->contrib_off[S];
    if (fx->contrib_off[0] != 0 || total < 0 || (total > 0 && (!fx->contrib_class || !fx->contrib_count)))
      return fail(e, YKPRED_E_INVALID, "set_spec_effects: malformed contribution rows");
    for (size_t i = 0; i < S; ++i)
      if (fx->contrib_off[i + 1] < fx->contrib_off[i]) return fail(e, YKPRED_E_INVALID, "set_spec_effects: contrib_off must not decrease");
    for (int k = 0; k < total; ++k)
      if (fx->contrib_class[k] < 0 || fx->contrib_class[k] >= e->KS || fx->contrib_count[k] <= 0)
        return fail(e, YKPRED_E_INVALID, "set_spec_effects: contribution outside the selector classes of the node table, or not positive");
    TRY(upload(e, e->d_fx_off, fx->contrib_off, S + 1, st));
    e->h_fx_off.assign(fx->contrib_off, fx->contrib_off + S + 1);  // (a batched round's prefix rule asks which specs move which histograms)
    if (total > 0) e->h_fx_cls.assign(fx->contrib_class, fx->contrib_class + total);
    TRY(upload(e, e->d_fx_cls, fx->contrib_class, (size_t)std::max(total, 1), st));
    TRY(upload(e, e->d_fx_cnt, fx->contrib_count, (size_t)std::max(total, 1), st));
    e->fx_contrib = total > 0;
  }
  e->h_fx_occupies.assign(S, 0);
  if (fx->occupied_ports && e->KP > 0) {
    TRY(upload(e, e->d_fx_occ, fx->occupied_ports, S * (size_t)e->KP, st));
    for (size_t i = 0; i < S * (size_t)e->KP; ++i)
      if (fx->occupied_ports[i] != 0) {
        e->fx_ports = true;
        e->h_fx_occupies[i / (size_t)e->KP] = 1;
      }
  }
  HIPCHK(hipStreamSynchronize(st));  // (the caller's arrays are not retained)
  e->fx_version = e->specs_version;
  return YKPRED_OK;
}

// Host forms of k_score's arithmetic (the same operation order; the file is compiled with -ffp-contract=off for host and device):
// what a node's key becomes after k more pods of a spec, computed by every rank of a sharded round from the exchanged columns.
static double host_node_score(const i64 (&total)[2], const i64 (&used)[2]) {
  double sum = 0.0, wsum = 0.0;
  for (int r = 0; r < 2; ++r) {
    if (total[r] <= 0) continue;
    const double avail = (double)(total[r] - used[r]);
    const double share = 1.0 - avail / (double)total[r];
    sum = sum + share;
    wsum = wsum + 1.0;
  }
  if (wsum == 0.0) return 1.0;
  return 1.0 - sum / wsum;
}
static u64 host_sortable_key(double v) {
  u64 b;
  memcpy(&b, &v, sizeof b);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
constexpr size_t kShardBatchMax = 256;  // asks proposed per batch of a batched round (the batch adapts below this)
constexpr size_t kBatchBytesMax = kShardBatchMax * ykk::kPropK * sizeof(ykk::RoundProposal) +
                                  kShardBatchMax * ((kShardBatchMax * ykk::kPropK + 63) / 64) * sizeof(u64);  // proposals + pair bits of one shard
static_assert(kShardBatchMax * ykk::kPropK * 2 <= (size_t)ykk::kDistinctSlots, "k_round_distinct's table holds a batch's proposals");

int32_t ykpred_allocate_round(ykpred_engine_t* e, uint32_t pre, uint32_t filt, int32_t n_asks, const int32_t* asks, int32_t* out_nodes) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:allocate_round");
  if (!e || n_asks < 0 || (n_asks > 0 && (!asks || !out_nodes))) return fail(e, YKPRED_E_INVALID, "allocate_round: bad argument");
  const bool sharded = e->comm && e->comm_world > 1;
  const bool stale = e->classes_dirty || !e->last_eval_valid || !e->rank_valid || e->bitmap_epoch != e->nodes_epoch || pre != e->last_pre || filt != e->last_filt ||
                     e->ranked_pre != pre || e->ranked_filt != filt || e->ranked_nodes_epoch != e->nodes_epoch || e->ranked_specs_version != e->specs_version;
  if (sharded) {
    // The sharded round is collective: a rank that left through a rank-local exit (no current evaluation, a patched ask, a list that
    // differs from the other ranks' — per-shard dictionaries can route different asks away) would leave the others blocked in the
    // first all-gather. So the ranks AGREE first — status, ask count, a hash of the list — and all of them return the same error.
    struct Agree {
      int32_t rc, n;
      u64 hash;
    };
    Agree mine{stale ? YKPRED_E_STATE : YKPRED_OK, n_asks, 0x9e3779b97f4a7c15ull};
    for (int i = 0; i < n_asks; ++i) {
      mine.hash = (mine.hash ^ (u64)(uint32_t)asks[i]) * 0x100000001b3ull;
      if (asks[i] < 0 || asks[i] >= e->P) mine.rc = YKPRED_E_INVALID;
      else if ((size_t)asks[i] < e->h_row_stale.size() && e->h_row_stale[(size_t)asks[i]]) mine.rc = YKPRED_E_STATE;
    }
    HIPCHK(hipSetDevice(e->cfg.device));
    const int W = e->comm_world;
    HIPCHK(e->d_agree.ensure((size_t)(W + 1) * sizeof(Agree)));
    Agree* d = e->d_agree.as<Agree>();
    HIPCHK(hipMemcpyAsync(d + W, &mine, sizeof(Agree), hipMemcpyHostToDevice, e->own_stream));
    NCCLCHK(rccl()->AllGather(d + W, d, sizeof(Agree), ncclInt8, e->comm, e->own_stream));
    std::vector<Agree> all((size_t)W);
    HIPCHK(hipMemcpyAsync(all.data(), d, (size_t)W * sizeof(Agree), hipMemcpyDeviceToHost, e->own_stream));
    HIPCHK(hipStreamSynchronize(e->own_stream));
    for (int g = 0; g < W; ++g) {
      if (all[(size_t)g].rc != YKPRED_OK)
        return fail(e, all[(size_t)g].rc, "allocate_round (sharded): rank " + std::to_string(g) + " cannot run the round (no current evaluation with decisions, "
                                          "an ask patched and not re-evaluated, or an ask index out of range): no rank runs it");
      if (all[(size_t)g].n != mine.n || all[(size_t)g].hash != mine.hash)
        return fail(e, YKPRED_E_INVALID, "allocate_round (sharded): rank " + std::to_string(g) + " was handed a different ask list: every rank passes the same asks in the same order");
    }
  }
  if (stale)
    return fail(e, YKPRED_E_STATE, "allocate_round: no current evaluation WITH decisions of these plugin lists (run ykpred_eval with YKPRED_OUT_DECISIONS)");
  const bool fx_current = e->fx_version == e->specs_version;
  const bool topo_on = (pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY)) && e->fam_spread.D > 0;
  if (topo_on && !fx_current)
    return fail(e, YKPRED_E_UNSUPPORTED, "allocate_round: topology constraints are active (an assumed pod's labels move the histograms of later asks) and the "
                                         "specs' effects are not uploaded (ykpred_set_spec_effects): decide ask by ask");
  const bool ports_on = (pre & filt & YKPRED_PLUGIN_NODE_PORTS) && e->KP > 0;
  bool port_asks = false;  // some ask of the round requests a host port: the round keeps the nodes' port words live
  for (int i = 0; i < n_asks; ++i) {
    if (asks[i] < 0 || asks[i] >= e->P) return fail(e, YKPRED_E_INVALID, "allocate_round: ask index out of range");
    if ((size_t)asks[i] < e->h_row_stale.size() && e->h_row_stale[(size_t)asks[i]])
      return fail(e, YKPRED_E_STATE, "allocate_round: an ask of the round was patched and not re-evaluated yet");
    if (ports_on && !port_asks) {
      const size_t sp = (size_t)e->h_pod_spec[(size_t)asks[i]];
      for (int k = 0; k < e->KP; ++k) port_asks = port_asks || e->h_wanted[sp * (size_t)e->KP + (size_t)k] != 0;
    }
  }
  if (port_asks && !fx_current)
    return fail(e, YKPRED_E_UNSUPPORTED, "allocate_round: an ask of the round requests a host port (an assumed pod's ports change later answers) and the specs' "
                                         "effects are not uploaded (ykpred_set_spec_effects): decide ask by ask");
  if (n_asks == 0) return YKPRED_OK;
  // One GPU: the sequential kernel decides runs of one spec by arithmetic (millions of asks per second) and pays ~10 us for every
  // other ask; the batched form pays per batch and wins on a mix of specs (configs[2]: 100 k -> 250 k+ asks/s). With topology
  // constraints live its batches end at every ask behind a contribution to a class it counts (configs[4]'s mix: 51 k against 59 k/s).
  // round_batched < 0 (default): batched without topology constraints when the list's mean run is shorter than four asks.
  bool batched = sharded || e->round_batched > 0;
  if (!sharded && e->round_batched < 0 && !topo_on && n_asks >= 512) {
    int changes = 1;
    for (int i = 1; i < n_asks; ++i) changes += e->h_pod_spec[(size_t)asks[i]] != e->h_pod_spec[(size_t)asks[i - 1]] ? 1 : 0;
    batched = (int64_t)changes * 4 >= (int64_t)n_asks;
  }
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  if (e->ev_eval_done) HIPCHK(hipStreamWaitEvent(st, e->ev_eval_done, 0));
  if (topo_on) TRY(ensure_histograms(e, st));  // (current after the evaluation this round builds on; a no-op then)
  const bool live_ports = port_asks || (ports_on && e->fx_ports && fx_current);
  const size_t N = (size_t)std::max(e->N, 1), R = (size_t)e->R, C = (size_t)std::max(e->C, 1), RW = (size_t)std::max(e->row_words, 1);
  const size_t G = topo_on ? (size_t)std::max(e->spread_constraints, 1) : 0, cells = topo_on ? (size_t)std::max<int64_t>(e->spread_cells, 1) : 0;
  // scratch layout (8-byte aligned pieces): the node-indexed copies of what a round changes (Requested, pod counts, port words), the
  // round's bookkeeping (moved bits in rank order, class cursors, node -> slot), the slot columns of the moved nodes, the bitsets
  // over slots (dead nodes; per class: slots it failed on), asks / decisions, and the live PreFilter state of the topology plugins
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t at = off;
    off += (bytes + 7) / 8 * 8;
    return at;
  };
  const size_t cap = N, cap64 = (N + 63) / 64, Wc = (size_t)std::min(e->W, ykk::kMaxW), KP1 = (size_t)std::max(e->KP, 1), KD1 = (size_t)std::max(e->KD, 1);
  // the "failed before" bits of every class: kept while they fit in a quarter of a GB (2 061 classes x 50 000 nodes: 13 MB; a
  // population with 10^6 classes does without them and evaluates every live slot)
  const bool keep_failed = C * cap64 * sizeof(u64) <= ((size_t)256 << 20);
  const size_t o_req = take(R * N * sizeof(i64)), o_bits = take(RW * sizeof(u64)),
               o_ports = take(live_ports ? (size_t)e->KP * N * sizeof(u64) : 0), o_cnt = take(N * sizeof(int)), o_cur = take(C * sizeof(int)),
               o_slot = take(N * sizeof(int)), o_mnode = take(cap * sizeof(int)), o_mtie = take(cap * sizeof(int)), o_mkey = take(cap * sizeof(u64)),
               o_mfree = take(R * cap * sizeof(i64)), o_mroom = take(cap * sizeof(int)), o_mports = take(KP1 * cap * sizeof(u64)),
               o_mtaint = take((size_t)std::max(e->KT, 1) * cap * sizeof(u64)), o_mlabel = take(std::max<size_t>(Wc, 1) * cap * sizeof(u64)), o_mdom = take(KD1 * cap * sizeof(int)),
               o_mflags = take(cap * sizeof(unsigned)), o_dead = take(cap64 * sizeof(u64)), o_failed = take(keep_failed ? C * cap64 * sizeof(u64) : 0),
               o_asks = take((size_t)n_asks * sizeof(int)), o_out = take((size_t)n_asks * sizeof(int)),
               o_nm = take(sizeof(int)), o_hist = take(cells * sizeof(int)), o_minv = take(G * sizeof(int)), o_mn = take(G * sizeof(int)),
               o_at = take(G * sizeof(int)), o_nd = take(G * sizeof(int)), o_prof = take(16 * sizeof(i64)),
               o_rkey = take(N * sizeof(u64)), o_rtie = take(N * sizeof(int)), o_cdesc = take(C * ykk::kDescWords * sizeof(u64)),
               o_prop = take(batched ? kBatchBytesMax : 0),
               o_allprop = take(sharded ? (size_t)e->comm_world * kBatchBytesMax : 0),
               o_list = take(batched ? (kShardBatchMax * ykk::kPropK + 2) * sizeof(int) : 0),
               o_forced = take(batched ? (size_t)n_asks * sizeof(int) : 0), o_runlen = take(batched ? (size_t)n_asks * sizeof(int) : 0),
               o_deltas = take(batched ? kShardBatchMax * sizeof(ykk::RoundNodeDelta) : 0),
               o_delta = take((sharded && topo_on) ? (size_t)n_asks * ykk::kDeltaStride * sizeof(int) : 0),
               o_alldelta = take((sharded && topo_on) ? (size_t)e->comm_world * kShardBatchMax * ykk::kDeltaStride * sizeof(int) : 0);
  HIPCHK(e->d_round.ensure(off));
  char* base = (char*)e->d_round.p;
  HIPCHK(hipMemcpyAsync(base + o_req, e->d_req.p, R * (size_t)e->N * sizeof(i64), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemcpyAsync(base + o_cnt, e->d_count.p, (size_t)e->N * sizeof(int), hipMemcpyDeviceToDevice, st));
  if (live_ports) HIPCHK(hipMemcpyAsync(base + o_ports, e->d_ports.p, (size_t)e->KP * (size_t)e->N * sizeof(u64), hipMemcpyDeviceToDevice, st));
  HIPCHK(hipMemsetAsync(base + o_bits, 0, RW * sizeof(u64), st));
  HIPCHK(hipMemsetAsync(base + o_cur, 0xff, C * sizeof(int), st));
  HIPCHK(hipMemsetAsync(base + o_slot, 0xff, N * sizeof(int), st));
  HIPCHK(hipMemsetAsync(base + o_dead, 0, cap64 * sizeof(u64), st));
  if (keep_failed) HIPCHK(hipMemsetAsync(base + o_failed, 0, C * cap64 * sizeof(u64), st));
  HIPCHK(hipMemsetAsync(base + o_nm, 0, sizeof(int), st));
  if (sharded && topo_on) HIPCHK(hipMemsetAsync(base + o_delta, 0, (size_t)n_asks * ykk::kDeltaStride * sizeof(int), st));
  HIPCHK(hipMemsetAsync(base + o_prof, 0, 16 * sizeof(i64), st));
  HIPCHK(hipMemcpyAsync(base + o_asks, asks, (size_t)n_asks * sizeof(int), hipMemcpyHostToDevice, st));
  ykk::NodeTable nt = node_table(e);
  nt.req = (const i64*)(base + o_req);
  nt.count = (const int*)(base + o_cnt);
  if (live_ports) nt.ports = (const u64*)(base + o_ports);
  ykk::ClassTable ct{e->d_class_sig.as<int>(), e->d_class_pin.as<int>(), e->d_chunk_class.as<int>(), e->d_chunk_begin.as<int>(),
                     e->d_chunk_len.as<int>(), e->d_chunk_first.as<int>(), e->d_members.as<int>(), e->d_chunk_zone.as<int>(), 0, nullptr};
  const bool spread_err = ((filt & YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD) && !(pre & YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD)) ||
                          ((filt & YKPRED_PLUGIN_INTER_POD_AFFINITY) && !(pre & YKPRED_PLUGIN_INTER_POD_AFFINITY)) ||
                          ((filt & YKPRED_PLUGIN_NODE_PORTS) && !(pre & YKPRED_PLUGIN_NODE_PORTS));
  ykk::RoundArgs ra{};
  ra.asks = (const int*)(base + o_asks);
  ra.pod_spec = e->d_pod_spec.as<int>();
  ra.pod_pin = e->d_pod_pin.as<int>();
  ra.pod_class = e->d_pod_class.as<int>();
  ra.perm = e->d_perm.as<int>();
  ra.rank = e->d_rank.as<int>();
  ra.key0 = e->d_key.as<u64>();
  ra.name_rank = e->has_name_rank ? e->d_name_rank.as<int>() : nullptr;
  ra.rkey = (const u64*)(base + o_rkey);
  ra.rtie = (const int*)(base + o_rtie);
  if (e->N > 0)
    hipLaunchKernelGGL(ykk::k_round_ranked_keys, dim3((unsigned)((e->N + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, e->N, ra.perm, ra.key0,
                       ra.name_rank, (u64*)(base + o_rkey), (int*)(base + o_rtie));
  ra.pre = pre;
  ra.filt = filt;
  ra.row_words = e->row_words;
  ra.all_fail = spread_err ? 1 : 0;
  ra.req = (i64*)(base + o_req);
  ra.count = (int*)(base + o_cnt);
  ra.ports = live_ports ? (u64*)(base + o_ports) : nullptr;
  ra.moved_bits = (u64*)(base + o_bits);
  ra.cursor = (int*)(base + o_cur);
  ra.n_moved = (int*)(base + o_nm);
  ra.cap = (int)cap;
  ra.cap64 = (int)cap64;
  ra.slot_of = (int*)(base + o_slot);
  ra.m_node = (int*)(base + o_mnode);
  ra.m_tie = (int*)(base + o_mtie);
  ra.m_key = (u64*)(base + o_mkey);
  ra.m_free = (i64*)(base + o_mfree);
  ra.m_room = (int*)(base + o_mroom);
  ra.m_ports = (u64*)(base + o_mports);
  ra.m_taint = (u64*)(base + o_mtaint);
  ra.m_label = (u64*)(base + o_mlabel);
  ra.m_dom = (int*)(base + o_mdom);
  ra.m_flags = (unsigned*)(base + o_mflags);
  ra.dead = (u64*)(base + o_dead);
  ra.failed = keep_failed ? (u64*)(base + o_failed) : nullptr;
  ra.out = (int*)(base + o_out);
  ra.prof = e->round_prof ? (i64*)(base + o_prof) : nullptr;
  ra.mode = ykk::kRoundDecide;
  ra.node_offset = e->node_offset;
  if (fx_current) {
    ra.fx.off = e->fx_contrib ? e->d_fx_off.as<int>() : nullptr;
    ra.fx.cls = e->d_fx_cls.as<int>();
    ra.fx.cnt = e->d_fx_cnt.as<int>();
    ra.fx.occupied = (live_ports && e->fx_ports) ? e->d_fx_occ.as<u64>() : nullptr;
  }
  ykk::SpecTable stbl = spec_table(e);
  if (topo_on) {
    // the round's own copy of the PreFilter state of the topology plugins (the engine's stays what the evaluation left) + what keeps
    // a spread constraint's minimum current; constraint → signature for the eligibility test of the node an ask lands on
    HIPCHK(hipMemcpyAsync(base + o_hist, e->d_sp_cnt.p, (size_t)e->spread_cells * sizeof(int), hipMemcpyDeviceToDevice, st));
    HIPCHK(hipMemcpyAsync(base + o_minv, e->d_sp_min.p, (size_t)e->spread_constraints * sizeof(int), hipMemcpyDeviceToDevice, st));
    stbl.spread.cnt = (int*)(base + o_hist);
    stbl.spread.minv = (int*)(base + o_minv);
    std::vector<int32_t> sig_of;
    for (size_t d = 0; d < e->spread_sig.size(); ++d) sig_of.insert(sig_of.end(), e->spread_sig[d].size(), (int32_t)d);
    sig_of.resize(G, 0);
    TRY(upload(e, e->d_sp_sig_of, sig_of.data(), sig_of.size(), st));
    HIPCHK(hipStreamSynchronize(st));  // (sig_of is a local)
    ra.topo_on = 1;
    ra.G = e->spread_constraints;
    ra.sig_aff = ykk::AffSigs{e->d_sig_aff_flags.as<unsigned>(), e->d_sig_aff_off.as<int>(), e->d_sig_aff_terms.as<u64>(), e->d_sig_pre_off.as<int>(),
                              e->d_sig_pre_terms.as<u64>()};
    ra.sig_tol = e->d_sig_tol.as<u64>();
    ra.sig_of = e->d_sp_sig_of.as<int>();
    ra.mn = (int*)(base + o_mn);
    ra.at_min = (int*)(base + o_at);
    ra.nd = (int*)(base + o_nd);
    if (e->spread_constraints > 0)
      hipLaunchKernelGGL(ykk::k_round_topo_init, dim3((unsigned)e->spread_constraints), dim3(ykk::kWave), 0, st, stbl.spread, e->spread_constraints,
                         (int*)(base + o_mn), (int*)(base + o_at), (int*)(base + o_nd));
  }
  TRY(complete_windows(e, st));  // (the class descriptors read the window of every index row)
  const ykk::Planes pr = ranked_planes_of(e, pre, filt, false, e->ranked_has_first);
  // the plane rows of every class, resolved once (the loop reads a class's descriptor in one load round instead of walking the tables)
  ra.cdesc = (const u64*)(base + o_cdesc);
  if (e->C > 0)
    hipLaunchKernelGGL(ykk::k_round_class_desc, dim3((unsigned)((e->C + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, ct, pr, e->C,
                       (u64*)(base + o_cdesc));
  if (batched) {
    // ---- A round in BATCHES: node-sharded engines (every rank holds the whole ask table and its own node shard), and one GPU
    // under YKPRED_TUNE round_batched=1. Per batch:
    //   1. PROPOSE, the asks of the batch in parallel against the state the accepted asks left (k_round_propose: a workgroup per
    //      ask, its kPropK best feasible nodes in (key, NodeID) order with the columns the key is made of); the distinct nodes a
    //      shard proposed are numbered (k_round_distinct) and EVERY ask of the batch is evaluated against every one of them
    //      (k_round_cross: a bit per pair);
    //   2. one all-gather of proposals + bits (world > 1), one copy to the host;
    //   3. every rank replays the sequential loop over the batch, ask after ask, as far as it is exact. A verdict only ever turns from
    //      fit to fail as a node fills (no topology signature) and an assume moves only its own node, towards the front. So the
    //      answer of ask j under the state the accepted asks left is the smallest (key, NodeID) among
    //        - the first entry of j's merged candidate list that is NOT an accepted node (unchanged since the proposal; the merged
    //          list is exact up to the last entry of the first shard list that came back full), and
    //        - the accepted nodes j passes: its bit (every Filter, at the state of the proposal) and NodeResourcesFit on the node's
    //          columns after the accepted pods — arithmetic on what the proposals carry.
    //      The prefix ends where that is not enough: every list entry is an accepted node that is full and the lists were cut
    //      (kPropK), an accepted node in front whose verdict is not arithmetic (the ask wants host ports or has a topology
    //      signature), an ask with a topology signature behind an accepted contribution to the histograms. Runs of one spec are
    //      accepted at once (k_allocate_round's run argument);
    //   4. the owners of the accepted nodes assume them (k_allocate_round, assume mode), everybody moves on behind the prefix.
    // The decisions equal the sequential loop's; across shards the tie-break between equal keys is the cluster-wide node index (as in
    // ykpred_exchange_decisions: NodeID order wherever names are zero-padded), on one GPU the NodeID rank itself.
    Rccl* r = sharded ? rccl() : nullptr;
    const int W = sharded ? e->comm_world : 1, me = sharded ? e->comm_rank : 0;
    constexpr int K = ykk::kPropK;
    auto cw_of = [](int b) { return (b * K + 63) / 64; };
    auto bytes_of = [&](int b) { return (size_t)b * K * sizeof(ykk::RoundProposal) + (size_t)b * (size_t)cw_of(b) * sizeof(u64); };
    char* d_send = base + o_prop;
    char* d_all = base + o_allprop;
    int* d_list = (int*)(base + o_list);
    int* d_nlist = d_list + kShardBatchMax * K;
    int* d_forced = (int*)(base + o_forced);
    int* d_runlen = (int*)(base + o_runlen);
    ykk::RoundNodeDelta* d_deltas = (ykk::RoundNodeDelta*)(base + o_deltas);
    std::vector<ykk::RoundNodeDelta> deltas;
    std::vector<int32_t> run_len;
    // (the proposals land in page-locked memory: a pageable copy of a few hundred KB per batch is staged and costs tens of microseconds)
    if (e->h_round_pinned_bytes < (size_t)W * kBatchBytesMax) {
      if (e->h_round_pinned) (void)hipHostFree(e->h_round_pinned);
      e->h_round_pinned = nullptr;
      e->h_round_pinned_bytes = 0;
      if (hipHostMalloc(&e->h_round_pinned, (size_t)W * kBatchBytesMax, hipHostMallocDefault) == hipSuccess) e->h_round_pinned_bytes = (size_t)W * kBatchBytesMax;
      else (void)hipGetLastError();
    }
    std::vector<char> all_pageable(e->h_round_pinned ? 0 : (size_t)W * kBatchBytesMax);
    char* const all = e->h_round_pinned ? (char*)e->h_round_pinned : all_pageable.data();
    std::vector<int32_t> forced;
    // Topology signatures: the histograms are cluster-wide state every shard holds. An accepted ask whose pod adds to a selector class
    // moves them — on its owner in the assume, on the others from the owner's delta record (second all-gather of the batch,
    // k_round_apply_deltas) — and it can turn verdicts of asks whose topology signature COUNTS THAT CLASS from fail to fit
    // anywhere: the prefix ends in front of the first such ask behind an accepted contribution to one of its classes. An ask whose
    // constraints count other classes saw the histograms it reads when the batch was proposed: its bits and lists stand.
    int* d_delta = (sharded && topo_on) ? (int*)(base + o_delta) : nullptr;
    int* d_alldelta = (sharded && topo_on) ? (int*)(base + o_alldelta) : nullptr;
    std::vector<int32_t> h_alldelta;
    auto spec_has_signature = [&](int spec) { return topo_on && e->spec_sig_spread[(size_t)spec] >= 0; };
    auto spec_contributes = [&](int spec) {
      return topo_on && e->fx_contrib && (size_t)spec + 1 < e->h_fx_off.size() && e->h_fx_off[(size_t)spec + 1] > e->h_fx_off[(size_t)spec];
    };
    std::vector<uint8_t> touched((size_t)std::max(e->KS, 1), 0);  // selector classes an accepted ask of this batch added to
    std::vector<int32_t> touched_list;
    auto touches = [&](int spec) {  // ... and whether a constraint of the spec's signature counts one of them
      if (touched_list.empty()) return false;
      const int d = e->spec_sig_spread[(size_t)spec];
      if (d < 0 || (size_t)d >= e->spread_sig.size()) return true;
      for (const ykpred_spread_t& c : e->spread_sig[(size_t)d])
        if (c.selector_class >= 0 && c.selector_class < e->KS && touched[(size_t)c.selector_class]) return true;
      return false;
    };
    struct Accepted {
      int64_t ord;     // the tie-break between equal keys: cluster-wide node index (world > 1) / NodeID rank (one GPU)
      int64_t gnode;   // cluster-wide node index
      u64 key0, key1;  // the node's key when it was proposed / after the accepted pods
      i64 alloc[ykk::kMaxR], used[ykk::kMaxR];  // its resource columns after the accepted pods
      int room;        // pod slots left after them
      int rank, node, didx;  // its shard, its index there, its number among the shard's distinct nodes of this batch
      int pods;        // pods the batch put on it, and what they request together (k_round_assume_nodes)
      i64 add[ykk::kMaxR];
    };
    const bool fit_on = (filt & YKPRED_PLUGIN_NODE_RESOURCES_FIT) && (pre & YKPRED_PLUGIN_NODE_RESOURCES_FIT);
    // Does a pod of `spec` still fit a node whose resource columns are (alloc, used) with `room` pod slots — NodeResourcesFit alone:
    // what the other Filters said of the pair when the batch was proposed cannot have changed for a spec without a topology signature
    // that wants no host port (taints, labels and names do not move during a round).
    auto still_fits = [&](int spec, const Accepted& n) {
      if (!fit_on) return true;
      if (n.room < 1) return false;
      for (int rr = 0; rr < e->R && rr < ykk::kMaxR; ++rr) {
        const i64 q = e->h_req[(size_t)spec * (size_t)e->R + (size_t)rr];
        if (q > 0 && q > n.alloc[rr] - n.used[rr]) return false;
      }
      return true;
    };
    auto spec_wants_ports = [&](int spec) {
      for (int kp = 0; kp < e->KP; ++kp)
        if (e->h_wanted[(size_t)spec * (size_t)e->KP + (size_t)kp] != 0) return true;
      return false;
    };
    auto less2 = [](u64 ka, int64_t oa, u64 kb, int64_t ob) { return ka < kb || (ka == kb && oa < ob); };
    struct Entry {
      u64 key;
      int64_t ord;
      int g, q;
    };
    std::vector<Accepted> acc;
    std::unordered_map<int64_t, int32_t> acc_of;  // cluster-wide node -> its entry of acc
    std::vector<Entry> ents;
    const int R = e->R;
    int pos = 0, batch = 32;
    int64_t exchanges = 0, delta_exchanges = 0, delta_cells = 0, batches = 0;
    double t_prop = 0, t_host = 0, t_assume = 0;  // (round_prof: seconds in the proposal kernels + copy, the replay, the assume)
    auto now_s = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    int64_t ends[4] = {0, 0, 0, 0};  // what ended the prefixes: the batch itself, a topology ask behind a contribution, candidate lists used up, an accepted node in front of an ask that wants host ports
    while (pos < n_asks) {
      const int b = std::min(batch, n_asks - pos), cw = cw_of(b);
      const size_t nbytes = bytes_of(b);
      const double t0 = e->round_prof ? now_s() : 0.0;
      ykk::RoundProposal* d_props = (ykk::RoundProposal*)d_send;
      u64* d_cross = (u64*)(d_send + (size_t)b * K * sizeof(ykk::RoundProposal));
      ra.mode = ykk::kRoundDecide;  // (the proposal kernels read the state; nothing is assumed)
      ra.first = pos;
      ra.n_asks = b;
      {
        const ykk::RoundCtx ctx{nt, stbl, ct, pr, ra};
        hipLaunchKernelGGL(ykk::k_round_propose, dim3((unsigned)b), dim3(ykk::kProposeThreads), 0, st, ctx, d_props);
        hipLaunchKernelGGL(ykk::k_round_distinct, dim3(1), dim3(ykk::kBlock), 0, st, d_props, b * K, d_list, d_nlist);
        hipLaunchKernelGGL(ykk::k_round_cross, dim3((unsigned)b), dim3(ykk::kBlock), 0, st, ctx, (const int*)d_list, (const int*)d_nlist, d_cross, cw);
      }
      HIPCHK(hipGetLastError());
      if (W > 1) {
        NCCLCHK(r->AllGather(d_send, d_all, nbytes, ncclInt8, e->comm, st));
        HIPCHK(hipMemcpyAsync(all, d_all, (size_t)W * nbytes, hipMemcpyDeviceToHost, st));
        ++exchanges;
      } else {
        HIPCHK(hipMemcpyAsync(all, d_send, nbytes, hipMemcpyDeviceToHost, st));
      }
      HIPCHK(hipStreamSynchronize(st));
      const double t1 = e->round_prof ? now_s() : 0.0;
      ++batches;
      auto prop_of = [&](int g, int m, int q) -> const ykk::RoundProposal& {
        return ((const ykk::RoundProposal*)(all + (size_t)g * nbytes))[(size_t)m * K + (size_t)q];
      };
      auto bit_of = [&](const Accepted& a2, int m) {
        const u64* cross = (const u64*)(all + (size_t)a2.rank * nbytes + (size_t)b * K * sizeof(ykk::RoundProposal));
        return a2.didx >= 0 && ((cross[(size_t)m * (size_t)cw + (size_t)(a2.didx >> 6)] >> (a2.didx & 63)) & 1ull) != 0;
      };
      auto ord_of = [&](const ykk::RoundProposal& p) { return W > 1 ? (int64_t)p.gnode : (int64_t)p.pad; };
      acc.clear();
      acc_of.clear();
      forced.assign((size_t)b, -1);
      run_len.assign((size_t)b, 1);
      for (int32_t c2 : touched_list) touched[(size_t)c2] = 0;
      touched_list.clear();
      int m = 0;
      bool contributed = false;  // an ask accepted in this batch moved a histogram
      bool plain = true;         // ... and none moved anything but resources and pod counts
      while (m < b) {
        const int ask0 = asks[pos + m];
        const int spec = e->h_pod_spec[(size_t)ask0];
        const bool has_sig = spec_has_signature(spec), ports = spec_wants_ports(spec);
        if (contributed && has_sig && touches(spec)) {  // (proposed again against the new histograms)
          ++ends[1];
          break;
        }
        // the ask's candidates of every shard, merged; exact up to the last entry of the first list that came back full
        ents.clear();
        bool cut = false;
        u64 h_key = ~0ull;
        int64_t h_ord = INT64_MAX;
        for (int g = 0; g < W; ++g) {
          int n = 0;
          for (int q = 0; q < K; ++q) {
            const ykk::RoundProposal& p = prop_of(g, m, q);
            if (p.node < 0) break;
            ents.push_back(Entry{p.key, ord_of(p), g, q});
            ++n;
          }
          if (n == K) {
            const Entry& last = ents.back();
            if (!cut || less2(last.key, last.ord, h_key, h_ord)) h_key = last.key, h_ord = last.ord;
            cut = true;
          }
        }
        std::sort(ents.begin(), ents.end(), [&](const Entry& x, const Entry& y) { return less2(x.key, x.ord, y.key, y.ord); });
        const Entry* fresh = nullptr;  // the first candidate no accepted ask has touched
        for (const Entry& en : ents) {
          if (cut && less2(h_key, h_ord, en.key, en.ord)) break;
          if (acc_of.find((int64_t)prop_of(en.g, m, en.q).gnode) == acc_of.end()) {
            fresh = &en;
            break;
          }
        }
        const bool beyond = !fresh && cut;  // (a node behind the lists' horizon may fit: only its shard knows)
        Accepted* best = nullptr;           // the accepted node the ask passes that stands first
        const Accepted* unknown = nullptr;  // ... and the first one whose verdict is not arithmetic
        for (Accepted& a2 : acc) {
          if (fresh && !less2(a2.key1, a2.ord, fresh->key, fresh->ord)) continue;  // (behind the untouched candidate: cannot be the answer)
          if (!bit_of(a2, m)) continue;  // (failed when the batch was proposed: a node only fills)
          if (ports) {  // (the node's port words moved with the pods it took: not arithmetic here)
            if (!unknown || less2(a2.key1, a2.ord, unknown->key1, unknown->ord)) unknown = &a2;
          } else if (still_fits(spec, a2)) {
            if (!best || less2(a2.key1, a2.ord, best->key1, best->ord)) best = &a2;
          }
        }
        if (beyond && !(best && less2(best->key1, best->ord, h_key, h_ord))) {
          ++ends[2];
          break;
        }
        const bool take_fresh = fresh && (!best || less2(fresh->key, fresh->ord, best->key1, best->ord));
        if (unknown && ((!fresh && !best) || (take_fresh ? less2(unknown->key1, unknown->ord, fresh->key, fresh->ord)
                                                          : less2(unknown->key1, unknown->ord, best->key1, best->ord)))) {
          ++ends[3];
          break;
        }
        if (!fresh && !best) {  // no node of any shard fits, and none will: verdicts only turn to fail
          out_nodes[pos + m] = -1;
          ++m;
          continue;
        }
        Accepted* same = take_fresh ? nullptr : best;
        Accepted node_now{};
        int w_fits = 1;
        if (same) {
          node_now = *same;
        } else {
          const ykk::RoundProposal& w = prop_of(fresh->g, m, fresh->q);
          node_now.ord = fresh->ord;
          node_now.gnode = w.gnode;
          node_now.key0 = node_now.key1 = w.key;
          for (int rr = 0; rr < ykk::kMaxR; ++rr) node_now.alloc[rr] = w.alloc[rr], node_now.used[rr] = w.req[rr];
          node_now.room = w.room;
          node_now.rank = fresh->g;
          node_now.node = w.node;
          node_now.didx = w.didx;
          w_fits = w.fits;
        }
        // a run of asks with this spec and no pin lands on this node while it fits
        int k = 1;
        const bool run_ok = e->h_pod_pin[(size_t)ask0] == YKPRED_NO_NODE_NAME && (same ? (fit_on && !spec_contributes(spec)) : true);
        if (run_ok) {
          // how many pods of the spec the node holds as it stands now (a node first met in this batch: the proposal's count)
          i64 holds = same ? (i64)node_now.room : (i64)w_fits;
          if (same)
            for (int rr = 0; rr < R && rr < ykk::kMaxR; ++rr) {
              const i64 q = e->h_req[(size_t)spec * (size_t)R + (size_t)rr];
              if (q > 0) holds = std::min(holds, (node_now.alloc[rr] - node_now.used[rr]) / q);
            }
          while (k < holds && m + k < b && e->h_pod_spec[(size_t)asks[pos + m + k]] == spec && e->h_pod_pin[(size_t)asks[pos + m + k]] == YKPRED_NO_NODE_NAME) ++k;
        }
        for (int rr = 0; rr < R && rr < ykk::kMaxR; ++rr) {
          node_now.used[rr] += e->h_req[(size_t)spec * (size_t)R + (size_t)rr] * k;
          node_now.add[rr] += e->h_req[(size_t)spec * (size_t)R + (size_t)rr] * k;
        }
        node_now.room -= k;
        node_now.pods += k;
        plain = plain && !spec_contributes(spec) && !((size_t)spec < e->h_fx_occupies.size() && e->h_fx_occupies[(size_t)spec]) && !ports;
        const i64 total[2] = {node_now.alloc[0], node_now.alloc[1]};
        const i64 used[2] = {node_now.used[0], node_now.used[1]};
        node_now.key1 = host_sortable_key(host_node_score(total, used));
        for (int q = 0; q < k; ++q) {
          out_nodes[pos + m + q] = (int32_t)node_now.gnode;
          if (node_now.rank == me) forced[(size_t)(m + q)] = node_now.node;
        }
        for (int q = 0; q < k;) {  // (the assume takes a run in pieces that stay inside one 64-ask window of its loop)
          const int piece = std::min(k - q, 64 - ((m + q) & 63));
          run_len[(size_t)(m + q)] = piece;
          q += piece;
        }
        if (same) {
          *same = node_now;
        } else {
          acc_of.emplace(node_now.gnode, (int32_t)acc.size());
          acc.push_back(node_now);
        }
        if (spec_contributes(spec)) {
          contributed = true;
          if (e->h_fx_cls.size() >= (size_t)e->h_fx_off[(size_t)spec + 1])
            for (int q = e->h_fx_off[(size_t)spec]; q < e->h_fx_off[(size_t)spec + 1]; ++q) {
              const int32_t c2 = e->h_fx_cls[(size_t)q];
              if (c2 >= 0 && c2 < e->KS && !touched[(size_t)c2]) {
                touched[(size_t)c2] = 1;
                touched_list.push_back(c2);
              }
            }
        }
        m += k;
      }
      if (m >= b) ++ends[0];
      const double t2 = e->round_prof ? now_s() : 0.0;
      // the owners assume what was accepted (the asks behind the prefix are proposed again)
      ra.mode = ykk::kRoundAssume;
      ra.first = pos;
      ra.n_asks = m;
      if (plain && e->round_node_assume != 0) {
        // nothing but resources and pod counts moved: node by node, a wave each, from the sums of the replay
        deltas.clear();
        for (const Accepted& a2 : acc)
          if (a2.rank == me && a2.pods > 0) {
            ykk::RoundNodeDelta nd{};
            nd.node = a2.node;
            nd.pods = a2.pods;
            for (int rr = 0; rr < ykk::kMaxR; ++rr) nd.add[rr] = a2.add[rr];
            deltas.push_back(nd);
          }
        if (!deltas.empty()) {
          HIPCHK(hipMemcpyAsync(d_deltas, deltas.data(), deltas.size() * sizeof(ykk::RoundNodeDelta), hipMemcpyHostToDevice, st));
          hipLaunchKernelGGL(ykk::k_round_assume_nodes, dim3((unsigned)deltas.size()), dim3(ykk::kWave), 0, st, ykk::RoundCtx{nt, stbl, ct, pr, ra},
                             (const ykk::RoundNodeDelta*)d_deltas, (int)deltas.size());
        }
      } else {
        HIPCHK(hipMemcpyAsync(d_forced + pos, forced.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, st));
        HIPCHK(hipMemcpyAsync(d_runlen + pos, run_len.data(), (size_t)m * sizeof(int), hipMemcpyHostToDevice, st));
        ra.run_len = d_runlen;
        ra.forced = d_forced;
        ra.delta = d_delta;
        hipLaunchKernelGGL(ykk::k_allocate_round, dim3(1), dim3(ykk::kRoundThreads), 0, st, ykk::RoundCtx{nt, stbl, ct, pr, ra});
      }
      HIPCHK(hipGetLastError());
      if (W > 1 && topo_on && contributed) {
        // what the owners' assumes added to the histograms: every rank applies the others' records (its own are in its copy already)
        const size_t rec_bytes = (size_t)m * ykk::kDeltaStride * sizeof(int);
        NCCLCHK(r->AllGather(d_delta + (size_t)pos * ykk::kDeltaStride, d_alldelta, rec_bytes, ncclInt8, e->comm, st));
        h_alldelta.resize((size_t)W * (size_t)m * ykk::kDeltaStride);
        HIPCHK(hipMemcpyAsync(h_alldelta.data(), d_alldelta, (size_t)W * rec_bytes, hipMemcpyDeviceToHost, st));
        for (int g = 0; g < W; ++g)
          if (g != me)
            hipLaunchKernelGGL(ykk::k_round_apply_deltas, dim3(1), dim3(ykk::kBlock), 0, st, stbl.spread, ra.mn, ra.at_min, ra.nd,
                               d_alldelta + (size_t)g * (size_t)m * ykk::kDeltaStride, m);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(st));
        ++exchanges;
        ++delta_exchanges;
        for (size_t q = 0; q < (size_t)W * (size_t)m; ++q) delta_cells += h_alldelta[q * ykk::kDeltaStride];
        for (size_t q = 0; q < (size_t)W * (size_t)m; ++q)  // (every rank reads the same records: the same verdict everywhere)
          if (h_alldelta[q * ykk::kDeltaStride] > ykk::kDeltaMax)
            return fail(e, YKPRED_E_UNSUPPORTED, "allocate_round (sharded): an assumed pod moves more histogram cells than a delta record holds (" +
                                                 std::to_string(ykk::kDeltaMax) + "): the round stops here on every rank");
      }
      HIPCHK(hipStreamSynchronize(st));  // (`forced` is reused by the next batch)
      if (e->round_prof) t_prop += t1 - t0, t_host += t2 - t1, t_assume += now_s() - t2;
      pos += m;
      batch = (int)std::min<size_t>(kShardBatchMax, (size_t)std::max(32, 2 * m + 16));
    }
    e->round_exchanges += exchanges;
    e->rounds_batched++;
    e->round_batched_asks += n_asks;
    e->round_batches += batches;
    if (e->round_prof)
      fprintf(stderr, "round_prof batched round (world %d): %d asks in %lld batches (%.1f accepted per batch), %lld exchanges, %lld of them histogram deltas (%lld cells); prefixes ended by: the batch %lld, a topology ask behind a contribution %lld, candidate lists used up %lld, an accepted node in front of an ask with host ports %lld; ms: proposals %.1f, replay %.1f, assume %.1f\n",
              W, n_asks, (long long)batches, batches ? (double)n_asks / (double)batches : 0.0, (long long)exchanges, (long long)delta_exchanges, (long long)delta_cells,
              (long long)ends[0], (long long)ends[1], (long long)ends[2], (long long)ends[3], t_prop * 1e3, t_host * 1e3, t_assume * 1e3);
    return YKPRED_OK;
  }
  // one launch per 32 768 asks: the loop is a single workgroup, and a bounded launch keeps the queue responsive (the state of the
  // round — scratch tables, moved list — lives in memory between the launches)
  const int per_launch = 32768;
  for (int first = 0; first < n_asks; first += per_launch) {
    ra.first = first;
    ra.n_asks = std::min(per_launch, n_asks - first);
    hipLaunchKernelGGL(ykk::k_allocate_round, dim3(1), dim3(ykk::kRoundThreads), 0, st, ykk::RoundCtx{nt, stbl, ct, pr, ra});
  }
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out_nodes, base + o_out, (size_t)n_asks * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  e->rounds_sequential++;
  e->round_sequential_asks += n_asks;
  if (e->round_prof) {
    i64 pr_[16];
    HIPCHK(hipMemcpy(pr_, base + o_prof, sizeof(pr_), hipMemcpyDeviceToHost));
    if (pr_[8] > 0)  // (the ticks exist in a -DYK_ROUND_PROF build of the engine only)
      fprintf(stderr, "round_prof asks=%d iters=%lld again=%lld | us: header %.0f again %.0f pin %.0f A %.0f B %.0f exchange %.0f assume %.0f tail %.0f\n",
              n_asks, (long long)pr_[8], (long long)pr_[9], pr_[0] / 100.0, pr_[1] / 100.0, pr_[2] / 100.0, pr_[3] / 100.0, pr_[4] / 100.0,
              pr_[5] / 100.0, pr_[6] / 100.0, pr_[7] / 100.0);
    if (pr_[13] > 0) fprintf(stderr, "round_prof shader clock %.0f MHz\n", 100.0 * (double)pr_[12] / (double)pr_[13]);
  }
  return YKPRED_OK;
}

int32_t ykpred_synchronize(ykpred_engine_t* e) {
  YK_SERIALISE(e);
  if (!e) return YKPRED_E_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  return YKPRED_OK;
}

int32_t ykpred_get_layout(const ykpred_engine_t* e, ykpred_layout_t* o) {
  YK_SERIALISE(e);
  if (!e || !o) return YKPRED_E_INVALID;
  o->num_nodes = e->N;
  o->num_pods = e->P;
  o->num_specs = e->S;
  o->num_classes = e->C;
  o->row_words = e->row_words;
  o->row_stride = e->row_stride;
  o->num_chunks = e->NC;
  o->plane_rows = e->fam_res.D + e->fam_tol.D + e->fam_aff.D + e->fam_spread.D;
  o->num_rows = std::max(e->rows_total, e->row_capacity);
  o->band_rows = e->rows_a;
  o->row_of_pod = e->d_pod_row.p;
  o->index_rows = e->index_rows;
  o->band_steps = e->band_steps_now;
  o->sweep_rows = e->sweep_ready ? e->sweep_row_off[ykk::kMaxIdxRows] : 0;
  o->index_rows_walked = (e->sweep_ready && e->walk2_chunks >= 0) ? e->index_rows_needed : e->index_rows;
  o->run_rows = e->sweep_ready ? e->run_rows : 0;
  o->fused_rows = e->fuse_ready ? e->fuse_row_count : 0;
  o->bitmap_bytes = (uint64_t)std::max(o->num_rows, 1) * (uint64_t)e->row_stride * sizeof(u64);
  o->bitmap = e->last_bitmap;
  o->counts = e->last_counts ? e->last_counts : e->d_counts.p;
  o->decisions = e->last_decisions ? e->last_decisions : e->d_decisions.p;
  o->decision_keys = e->last_keys ? e->last_keys : e->d_keys.p;
  o->spread_counts = e->d_sp_cnt.p;
  o->spread_present = e->d_sp_present.p;
  o->spread_cells = e->spread_cells;
  return YKPRED_OK;
}

int32_t ykpred_last_timing(const ykpred_engine_t* ce, ykpred_timing_t* o) {
  YK_SERIALISE(ce);
  ykpred_engine_t* e = const_cast<ykpred_engine_t*>(ce);
  if (!e || !o) return YKPRED_E_INVALID;
  if (!e->timing_valid) return fail(e, YKPRED_E_STATE, "last eval was not run with YKPRED_EVAL_PROFILE");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  o->num_kernels = e->timed;
  o->total_ms = 0.f;
  for (int i = 0; i < e->timed; ++i) {
    float ms = 0.f;
    HIPCHK(hipEventElapsedTime(&ms, e->ev[2 * i], e->ev[2 * i + 1]));
    o->kernel_ms[i] = ms;
    o->kernel_name[i] = e->timed_name[i];
  }
  HIPCHK(hipEventElapsedTime(&o->total_ms, e->ev[2 * YKPRED_MAX_TIMED_KERNELS], e->ev[2 * YKPRED_MAX_TIMED_KERNELS + 1]));
  return YKPRED_OK;
}

int32_t ykpred_read_bitmap(ykpred_engine_t* e, int32_t first, int32_t num, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out || first < 0 || num < 0 || first + num > e->P) return fail(e, YKPRED_E_INVALID, "read_bitmap: range");
  if (!e->last_bitmap) return fail(e, YKPRED_E_STATE, "read_bitmap: no eval yet");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  if (num == 0 || e->row_words == 0) return YKPRED_OK;
  // rows are addressed through row_of_pod: gather them densely on the device (in slabs), then copy
  const size_t row_bytes = (size_t)e->row_words * sizeof(u64);
  const int slab = (int)std::max<size_t>(1, std::min<size_t>((size_t)num, ((size_t)1 << 30) / row_bytes));
  HIPCHK(e->d_scratch.ensure((size_t)slab * row_bytes));
  for (int done = 0; done < num; done += slab) {
    const int n = std::min(slab, num - done);
    hipLaunchKernelGGL(ykk::k_gather_rows, dim3((unsigned)n), dim3(ykk::kBlock), 0, e->own_stream, (const u64*)e->last_bitmap, n, (const int*)nullptr,
                       first + done, e->d_pod_row.as<int>(), e->row_words, e->row_stride, e->d_scratch.as<u64>());
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(e->own_stream));
    HIPCHK(hipMemcpy((char*)out + (size_t)done * row_bytes, e->d_scratch.p, (size_t)n * row_bytes, hipMemcpyDeviceToHost));
  }
  return YKPRED_OK;
}

int32_t ykpred_read_counts(ykpred_engine_t* e, int32_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  if (e->P) HIPCHK(hipMemcpy(out, e->last_counts ? e->last_counts : e->d_counts.p, (size_t)e->P * sizeof(int), hipMemcpyDeviceToHost));
  return YKPRED_OK;
}
int32_t ykpred_read_decisions(ykpred_engine_t* e, int32_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  if (e->P) HIPCHK(hipMemcpy(out, e->last_decisions ? e->last_decisions : e->d_decisions.p, (size_t)e->P * sizeof(int), hipMemcpyDeviceToHost));
  return YKPRED_OK;
}
int32_t ykpred_read_scores(ykpred_engine_t* e, double* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  if (!e->nodes_set) return fail(e, YKPRED_E_STATE, "read_scores: no nodes");
  HIPCHK(hipSetDevice(e->cfg.device));
  if (e->N) {
    ykk::NodeTable nt = node_table(e);
    hipLaunchKernelGGL(ykk::k_score, dim3((unsigned)((e->N + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, e->own_stream, nt,
                       e->d_score.as<double>(), e->d_key.as<u64>());
    HIPCHK(hipStreamSynchronize(e->own_stream));
    HIPCHK(hipMemcpy(out, e->d_score.p, (size_t)e->N * sizeof(double), hipMemcpyDeviceToHost));
  }
  return YKPRED_OK;
}

int32_t ykpred_bitmap_checksum(ykpred_engine_t* e, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  if (!e->last_bitmap) return fail(e, YKPRED_E_STATE, "checksum: no eval yet");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(e->d_scratch.ensure(sizeof(u64)));
  HIPCHK(hipMemsetAsync(e->d_scratch.p, 0, sizeof(u64), e->own_stream));
  if (e->P && e->row_words)
    hipLaunchKernelGGL(ykk::k_checksum, dim3(2048), dim3(ykk::kBlock), 0, e->own_stream, (const u64*)e->last_bitmap, e->P, e->row_words,
                       e->row_stride, e->d_pod_row.as<int>(), e->d_scratch.as<u64>());
  HIPCHK(hipStreamSynchronize(e->own_stream));
  HIPCHK(hipMemcpy(out, e->d_scratch.p, sizeof(u64), hipMemcpyDeviceToHost));
  return YKPRED_OK;
}

int32_t ykpred_get_counters(const ykpred_engine_t* e, int64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  out[0] = e->n_full_evals;
  out[1] = e->n_node_patches;
  out[2] = e->n_row_patches;
  out[3] = e->n_queries;
  out[4] = e->n_gathers;
  out[5] = e->n_uploads;
  return YKPRED_OK;
}

int32_t ykpred_get_round_info(const ykpred_engine_t* e, int64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  out[0] = e->rounds_batched;
  out[1] = e->round_batched_asks;
  out[2] = e->round_batches;
  out[3] = e->round_exchanges;
  out[4] = e->rounds_sequential;
  out[5] = e->round_sequential_asks;
  return YKPRED_OK;
}

int32_t ykpred_read_rows(ykpred_engine_t* e, int32_t n, const int32_t* pods, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || n < 0 || (n > 0 && (!pods || !out))) return fail(e, YKPRED_E_INVALID, "read_rows: bad argument");
  if (!e->last_bitmap) return fail(e, YKPRED_E_STATE, "read_rows: no eval yet");
  for (int i = 0; i < n; ++i)
    if (pods[i] < 0 || pods[i] >= e->P) return fail(e, YKPRED_E_INVALID, "read_rows: pod index out of range");
  if (n == 0 || e->row_words == 0) return YKPRED_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  hipStream_t st = e->own_stream;
  const size_t idx_bytes = ((size_t)n * sizeof(int32_t) + 15) / 16 * 16, out_bytes = (size_t)n * (size_t)e->row_words * sizeof(u64);
  HIPCHK(e->d_scratch.ensure(idx_bytes + out_bytes));
  char* base = (char*)e->d_scratch.p;
  HIPCHK(hipMemcpyAsync(base, pods, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ykk::k_gather_rows, dim3((unsigned)n), dim3(ykk::kBlock), 0, st, (const u64*)e->last_bitmap, n, (const int*)base, 0,
                     e->d_pod_row.as<int>(), e->row_words, e->row_stride, (u64*)(base + idx_bytes));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, base + idx_bytes, out_bytes, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

// The resident answer, one ask at a time. See ykpred.h.
int32_t ykpred_peek_row(ykpred_engine_t* e, int32_t pod, uint32_t pre, uint32_t filt, uint64_t* out_row, int32_t* out_count, int32_t* out_decision) {
  YK_SERIALISE(e);
  if (!e || !out_row) return fail(e, YKPRED_E_INVALID, "peek_row: bad argument");
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap || e->bitmap_epoch != e->nodes_epoch || pre != e->last_pre || filt != e->last_filt)
    return fail(e, YKPRED_E_STATE, "peek_row: no current evaluation of these plugin lists (tables changed since)");
  if (pod < 0 || pod >= e->P || (size_t)pod >= e->h_pod_row.size()) return fail(e, YKPRED_E_INVALID, "peek_row: ask index out of range");
  if ((size_t)pod < e->h_row_stale.size() && e->h_row_stale[(size_t)pod]) return fail(e, YKPRED_E_STATE, "peek_row: the ask's row was patched and not re-evaluated yet");
  const int row = e->h_pod_row[(size_t)pod];
  if (row < 0 || (int64_t)row >= e->last_bitmap_rows) return fail(e, YKPRED_E_STATE, "peek_row: the ask has no bitmap row");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->peek_stream ? e->peek_stream : e->own_stream;
  const size_t row_bytes = (size_t)e->row_words * sizeof(u64), need = row_bytes + 16;
  if (e->peek_pinned_bytes < need) {
    if (e->peek_pinned) (void)hipHostFree(e->peek_pinned);
    e->peek_pinned = nullptr;
    e->peek_pinned_bytes = 0;
    HIPCHK(hipHostMalloc(&e->peek_pinned, need + 4096, hipHostMallocDefault));
    e->peek_pinned_bytes = need + 4096;
  }
  // ordered after the evaluation by its completion event, not by draining the device: other streams keep running
  if (e->ev_eval_done) HIPCHK(hipStreamWaitEvent(st, e->ev_eval_done, 0));
  char* pin = (char*)e->peek_pinned;
  if (row_bytes) HIPCHK(hipMemcpyAsync(pin + 16, (const u64*)e->last_bitmap + (size_t)row * (size_t)e->row_stride, row_bytes, hipMemcpyDeviceToHost, st));
  if (out_count) HIPCHK(hipMemcpyAsync(pin, (const int*)e->last_counts + pod, sizeof(int), hipMemcpyDeviceToHost, st));
  if (out_decision) HIPCHK(hipMemcpyAsync(pin + 4, (const int*)e->last_decisions + pod, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  if (row_bytes) memcpy(out_row, pin + 16, row_bytes);
  if (out_count) memcpy(out_count, pin, sizeof(int));
  if (out_decision) memcpy(out_decision, pin + 4, sizeof(int));
  return YKPRED_OK;
}

// Is the bitmap of the last evaluation the answer for the current tables and these plugin lists?
int32_t ykpred_answer_state(ykpred_engine_t* e, uint32_t pre, uint32_t filt, int32_t* num_classes) {
  YK_SERIALISE(e);
  if (!e) return YKPRED_E_INVALID;
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap || e->bitmap_epoch != e->nodes_epoch || pre != e->last_pre || filt != e->last_filt)
    return fail(e, YKPRED_E_STATE, "answer_state: no current evaluation of these plugin lists");
  for (uint8_t stale : e->h_row_stale)
    if (stale) return fail(e, YKPRED_E_STATE, "answer_state: ask rows were patched and not re-evaluated yet");
  if (num_classes) *num_classes = e->C;
  return YKPRED_OK;
}

// The class of an ask in the engine's current class index (host-side lookup, no device work)
int32_t ykpred_pod_class(ykpred_engine_t* e, int32_t pod, int32_t* out_class) {
  YK_SERIALISE(e);
  if (!e || !out_class) return YKPRED_E_INVALID;
  if (e->classes_dirty || pod < 0 || pod >= e->P || (size_t)pod >= e->h_pod_class.size()) return fail(e, YKPRED_E_STATE, "pod_class: no current class index for this ask");
  *out_class = e->h_pod_class[(size_t)pod];
  return YKPRED_OK;
}

// perm[i] = the node at position i of the bin-pack order of the last evaluation that produced decisions
int32_t ykpred_read_order(ykpred_engine_t* e, int32_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return fail(e, YKPRED_E_INVALID, "read_order: bad argument");
  if (!e->rank_valid) return fail(e, YKPRED_E_STATE, "read_order: no current bin-pack order (run ykpred_eval with YKPRED_OUT_DECISIONS)");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->peek_stream ? e->peek_stream : e->own_stream;
  if (e->ev_eval_done) HIPCHK(hipStreamWaitEvent(st, e->ev_eval_done, 0));
  if (e->N) HIPCHK(hipMemcpyAsync(out, e->d_perm.p, (size_t)e->N * sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_read_row_map(ykpred_engine_t* e, int32_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return YKPRED_E_INVALID;
  if (e->classes_dirty || (int)e->h_pod_row.size() < e->P) return fail(e, YKPRED_E_STATE, "read_row_map: no current class build (run ykpred_eval)");
  std::copy(e->h_pod_row.begin(), e->h_pod_row.begin() + e->P, out);
  return YKPRED_OK;
}

int32_t ykpred_read_pod_classes(ykpred_engine_t* e, int32_t* pod_class, int32_t* class_rep) {
  YK_SERIALISE(e);
  if (!e) return YKPRED_E_INVALID;
  if (e->classes_dirty || !e->last_eval_valid) return fail(e, YKPRED_E_STATE, "read_pod_classes: no current evaluation");
  if (pod_class) std::copy(e->h_pod_class.begin(), e->h_pod_class.begin() + e->P, pod_class);
  if (class_rep) std::copy(e->h_class_first.begin(), e->h_class_first.begin() + e->C, class_rep);
  return YKPRED_OK;
}

int32_t ykpred_check_class_rows(ykpred_engine_t* e, uint64_t* bad_words) {
  YK_SERIALISE(e);
  if (!e || !bad_words) return YKPRED_E_INVALID;
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap) return fail(e, YKPRED_E_STATE, "check_class_rows: no current evaluation");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(e->d_scratch.ensure(sizeof(u64)));
  HIPCHK(hipMemsetAsync(e->d_scratch.p, 0, sizeof(u64), e->own_stream));
  if (e->P && e->row_stride)
    hipLaunchKernelGGL(ykk::k_check_class_rows, dim3(4096), dim3(ykk::kBlock), 0, e->own_stream, (const u64*)e->last_bitmap, e->P, e->row_words,
                       e->row_stride, e->d_pod_class.as<int>(), e->d_class_first.as<int>(), e->d_pod_row.as<int>(), e->d_scratch.as<u64>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->own_stream));
  HIPCHK(hipMemcpy(bad_words, e->d_scratch.p, sizeof(u64), hipMemcpyDeviceToHost));
  return YKPRED_OK;
}

// reads one byte at p[off] (the guard self-test)
__global__ void k_guard_probe(const unsigned char* p, long off, unsigned* sink) {
  if (threadIdx.x == 0) *sink = p[off];
}
int32_t ykpred_guard_selftest(ykpred_engine_t* e, int32_t offset) {
  YK_SERIALISE(e);
  if (!e) return YKPRED_E_INVALID;
  if (!guard_pages_on()) return fail(e, YKPRED_E_STATE, "guard_selftest: YKPRED_GUARD_PAGES is not set");
  HIPCHK(hipSetDevice(e->cfg.device));
  const size_t bytes = 1000;  // deliberately not a multiple of anything
  DevBuf block, sink;
  HIPCHK(block.ensure(bytes));
  HIPCHK(sink.ensure(sizeof(unsigned)));
  HIPCHK(hipMemsetAsync(block.p, 0x5a, bytes, e->own_stream));
  const long off = offset >= 0 ? (long)bytes - 1 + offset : (long)offset;  // offset 0 = the last byte, -1 = the byte before the first
  hipLaunchKernelGGL(k_guard_probe, dim3(1), dim3(64), 0, e->own_stream, block.as<unsigned char>(), off, sink.as<unsigned>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->own_stream));
  block.release();
  sink.release();
  return YKPRED_OK;
}

int32_t ykpred_query(ykpred_engine_t* e, int32_t n, const int32_t* pods, const int32_t* nodes, uint32_t pre, uint32_t filt, uint8_t* fit,
                     uint8_t* code, uint32_t* reason) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:query");
  if (e) e->n_queries++;
  if (!e || n < 0 || (n > 0 && (!pods || !nodes || !fit))) return fail(e, YKPRED_E_INVALID, "query: bad argument");
  if (!e->nodes_set || !e->specs_set || !e->pods_set) return fail(e, YKPRED_E_STATE, "query: tables not uploaded");
  for (int i = 0; i < n; ++i)
    if (pods[i] < 0 || pods[i] >= e->P || nodes[i] < 0 || nodes[i] >= e->N) return fail(e, YKPRED_E_INVALID, "query: index out of range");
  if (n == 0) return YKPRED_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  // scratch layout: pods | nodes | reason | fit | code
  size_t need = (size_t)n * (3 * sizeof(int32_t) + 2);
  HIPCHK(e->d_scratch.ensure(need + 64));
  int32_t* d_p = e->d_scratch.as<int32_t>();
  int32_t* d_n = d_p + n;
  uint32_t* d_r = (uint32_t*)(d_n + n);
  uint8_t* d_f = (uint8_t*)(d_r + n);
  uint8_t* d_c = d_f + n;
  if ((pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY))) TRY(ensure_histograms(e, st));
  else if (e->spread_dirty) TRY(build_spread_tables(e, st));
  HIPCHK(hipMemcpyAsync(d_p, pods, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(d_n, nodes, (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ykk::k_query, dim3((unsigned)((n + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, node_table(e),
                     spec_table(e), n, d_p, d_n, e->d_pod_spec.as<int>(), e->d_pod_pin.as<int>(), pre, filt, d_f, d_c, d_r);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(fit, d_f, (size_t)n, hipMemcpyDeviceToHost, st));
  if (code) HIPCHK(hipMemcpyAsync(code, d_c, (size_t)n, hipMemcpyDeviceToHost, st));
  if (reason) HIPCHK(hipMemcpyAsync(reason, d_r, (size_t)n * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_query_pod(ykpred_engine_t* e, int32_t pod, uint32_t pre, uint32_t filt, uint8_t* fit, uint8_t* code, uint32_t* reason) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:query_pod");
  if (e) e->n_queries++;
  if (!e || !fit) return fail(e, YKPRED_E_INVALID, "query_pod: bad argument");
  if (!e->nodes_set || !e->specs_set || !e->pods_set) return fail(e, YKPRED_E_STATE, "query_pod: tables not uploaded");
  if (pod < 0 || pod >= e->P) return fail(e, YKPRED_E_INVALID, "query_pod: index out of range");
  if (e->N == 0) return YKPRED_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  if ((pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY))) TRY(ensure_histograms(e, st));
  else if (e->spread_dirty) TRY(build_spread_tables(e, st));
  const size_t N = (size_t)e->N;
  HIPCHK(e->d_scratch.ensure(N * 6 + 64));
  uint32_t* d_r = e->d_scratch.as<uint32_t>();
  uint8_t* d_f = (uint8_t*)(d_r + N);
  uint8_t* d_c = d_f + N;
  hipLaunchKernelGGL(ykk::k_query_pod, dim3((unsigned)((N + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, node_table(e),
                     spec_table(e), e->h_pod_spec[(size_t)pod], e->h_pod_pin[(size_t)pod], pre, filt, d_f, d_c, d_r);
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(fit, d_f, N, hipMemcpyDeviceToHost, st));
  if (code) HIPCHK(hipMemcpyAsync(code, d_c, N, hipMemcpyDeviceToHost, st));
  if (reason) HIPCHK(hipMemcpyAsync(reason, d_r, N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

// ykpred_query_pod with ONE 4-byte word per node and one device → host copy: bits 0-7 plugin code, bit 8 fit, bits 9-12 the
// four reason flags, bits 13.. the insufficient-resource flags (reason >> YKPRED_REASON_RESOURCE_SHIFT).
int32_t ykpred_query_pod_packed(ykpred_engine_t* e, int32_t pod, uint32_t pre, uint32_t filt, uint32_t* out) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:query_pod");
  if (e) e->n_queries++;
  if (!e || !out) return fail(e, YKPRED_E_INVALID, "query_pod_packed: bad argument");
  if (!e->nodes_set || !e->specs_set || !e->pods_set) return fail(e, YKPRED_E_STATE, "query_pod_packed: tables not uploaded");
  if (pod < 0 || pod >= e->P) return fail(e, YKPRED_E_INVALID, "query_pod_packed: index out of range");
  if (e->N == 0) return YKPRED_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  if ((pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY))) TRY(ensure_histograms(e, st));
  else if (e->spread_dirty) TRY(build_spread_tables(e, st));
  const size_t N = (size_t)e->N;
  HIPCHK(e->d_scratch.ensure(N * sizeof(uint32_t) + 64));
  hipLaunchKernelGGL(ykk::k_query_pod_packed, dim3((unsigned)((N + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, node_table(e),
                     spec_table(e), e->h_pod_spec[(size_t)pod], e->h_pod_pin[(size_t)pod], pre, filt, e->d_scratch.as<uint32_t>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, e->d_scratch.p, N * sizeof(uint32_t), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_preemption_batch(ykpred_engine_t* e, int32_t nq, const int32_t* pods, const int32_t* nodes, const int32_t* voff,
                                const int64_t* vreq, const uint8_t* vpresent, const uint64_t* ports_after, const int32_t* start,
                                uint32_t pre, uint32_t filt, int32_t* out) {
  YK_SERIALISE(e);
  if (!e || nq < 0 || (nq > 0 && (!pods || !nodes || !voff || !start || !out))) return fail(e, YKPRED_E_INVALID, "preemption: bad argument");
  if (!e->nodes_set || !e->specs_set || !e->pods_set) return fail(e, YKPRED_E_STATE, "preemption: tables not uploaded");
  if (nq == 0) return YKPRED_OK;
  const int total = voff[nq];
  if (voff[0] != 0 || total < 0 || (total > 0 && (!vreq || !vpresent))) return fail(e, YKPRED_E_INVALID, "preemption: bad victim arrays");
  for (int q = 0; q < nq; ++q)
    if (pods[q] < 0 || pods[q] >= e->P || nodes[q] < 0 || nodes[q] >= e->N || voff[q + 1] < voff[q] || start[q] < 0)
      return fail(e, YKPRED_E_INVALID, "preemption: index out of range");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->own_stream;
  if ((pre & filt & (YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD | YKPRED_PLUGIN_INTER_POD_AFFINITY))) TRY(ensure_histograms(e, st));
  else if (e->spread_dirty) TRY(build_spread_tables(e, st));
  // scratch layout (8-byte aligned pieces): vreq | ports_after | q_pod | q_node | voff | q_start | out | vpresent
  const size_t vbytes = (size_t)total * (size_t)e->R * sizeof(i64);
  const size_t pbytes = (ports_after && e->KP > 0) ? (size_t)total * (size_t)e->KP * sizeof(u64) : 0;
  const size_t ibytes = ((size_t)nq * 4 + 7) / 8 * 8, obytes = ((size_t)(nq + 1) * 4 + 7) / 8 * 8;
  size_t off = 0;
  const size_t o_vreq = off; off += vbytes;
  const size_t o_ports = off; off += pbytes;
  const size_t o_pod = off; off += ibytes;
  const size_t o_node = off; off += ibytes;
  const size_t o_voff = off; off += obytes;
  const size_t o_start = off; off += ibytes;
  const size_t o_out = off; off += ibytes;
  const size_t o_pres = off; off += (size_t)total + 8;
  HIPCHK(e->d_scratch.ensure(off + 64));
  char* base = (char*)e->d_scratch.p;
  if (total) {
    HIPCHK(hipMemcpyAsync(base + o_vreq, vreq, vbytes, hipMemcpyHostToDevice, st));
    if (pbytes) HIPCHK(hipMemcpyAsync(base + o_ports, ports_after, pbytes, hipMemcpyHostToDevice, st));
    HIPCHK(hipMemcpyAsync(base + o_pres, vpresent, (size_t)total, hipMemcpyHostToDevice, st));
  }
  HIPCHK(hipMemcpyAsync(base + o_pod, pods, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(base + o_node, nodes, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(base + o_voff, voff, (size_t)(nq + 1) * 4, hipMemcpyHostToDevice, st));
  HIPCHK(hipMemcpyAsync(base + o_start, start, (size_t)nq * 4, hipMemcpyHostToDevice, st));
  hipLaunchKernelGGL(ykk::k_preempt, dim3((unsigned)((nq + ykk::kWave - 1) / ykk::kWave)), dim3(ykk::kWave), 0, st, node_table(e), spec_table(e),
                     nq, (const int*)(base + o_pod), (const int*)(base + o_node), (const int*)(base + o_voff), e->d_pod_spec.as<int>(),
                     e->d_pod_pin.as<int>(), (const i64*)(base + o_vreq), (const unsigned char*)(base + o_pres),
                     pbytes ? (const u64*)(base + o_ports) : (const u64*)nullptr, (const int*)(base + o_start), pre, filt,
                     (int*)(base + o_out));
  HIPCHK(hipGetLastError());
  HIPCHK(hipMemcpyAsync(out, base + o_out, (size_t)nq * 4, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_preemption_ports(ykpred_engine_t* e, int32_t pod, int32_t node, int32_t nv, const int64_t* vreq, const uint8_t* vpresent,
                                const uint64_t* ports_after, int32_t start, uint32_t pre, uint32_t filt, int32_t* out) {
  if (!out || nv < 0) return fail(e, YKPRED_E_INVALID, "preemption: bad argument");
  const int32_t voff[2] = {0, nv};
  return ykpred_preemption_batch(e, 1, &pod, &node, voff, vreq, vpresent, ports_after, &start, pre, filt, out);
}

int32_t ykpred_preemption(ykpred_engine_t* e, int32_t pod, int32_t node, int32_t nv, const int64_t* vreq, const uint8_t* vpresent,
                          int32_t start, uint32_t pre, uint32_t filt, int32_t* out) {
  return ykpred_preemption_ports(e, pod, node, nv, vreq, vpresent, nullptr, start, pre, filt, out);
}

// ---------------------------------------------------------------------------------------------------
// multi-GPU exchanges over RCCL (see ykpred.h)
// ---------------------------------------------------------------------------------------------------
int32_t ykpred_comm_use_library(const char* path) {
  if (!path) return YKPRED_E_INVALID;
  rccl_library_override() = path;
  return YKPRED_OK;
}

int32_t ykpred_comm_unique_id(uint8_t* id) {
  if (!id) return YKPRED_E_INVALID;
  Rccl* r = rccl();
  if (!r->error.empty()) {
    g_create_error = r->error;
    return YKPRED_E_DEVICE;
  }
  ncclUniqueId u;
  ncclResult_t s = r->GetUniqueId(&u);
  if (s != ncclSuccess) {
    g_create_error = std::string("ncclGetUniqueId: ") + r->GetErrorString(s);
    return YKPRED_E_DEVICE;
  }
  static_assert(sizeof(u) == YKPRED_COMM_ID_BYTES, "ncclUniqueId size");
  memcpy(id, &u, sizeof u);
  return YKPRED_OK;
}

int32_t ykpred_comm_init(ykpred_engine_t* e, const uint8_t* id, int32_t rank, int32_t world, int32_t node_offset) {
  YK_SERIALISE(e);
  if (!e || !id || world < 1 || rank < 0 || rank >= world || node_offset < 0) return fail(e, YKPRED_E_INVALID, "comm_init: bad argument");
  if (e->comm) return fail(e, YKPRED_E_STATE, "comm_init: a communicator is already attached");
  Rccl* r = rccl();
  if (!r->error.empty()) return fail(e, YKPRED_E_DEVICE, r->error);
  HIPCHK(hipSetDevice(e->cfg.device));
  ncclUniqueId u;
  memcpy(&u, id, sizeof u);
  NCCLCHK(r->CommInitRank(&e->comm, world, u, rank));
  e->comm_rank = rank;
  e->comm_world = world;
  e->node_offset = node_offset;
  e->hist_epoch = 0;  // shard-local histograms are not the cluster's
  return YKPRED_OK;
}

int32_t ykpred_comm_info(const ykpred_engine_t* e, int32_t* rank, int32_t* world, int32_t* node_offset) {
  if (!e) return YKPRED_E_INVALID;
  if (rank) *rank = e->comm ? e->comm_rank : 0;
  if (world) *world = e->comm ? e->comm_world : 1;
  if (node_offset) *node_offset = e->comm ? e->node_offset : 0;
  return YKPRED_OK;
}

int32_t ykpred_comm_destroy(ykpred_engine_t* e) {
  YK_SERIALISE(e);
  if (!e) return YKPRED_E_INVALID;
  if (e->comm) {
    (void)hipSetDevice(e->cfg.device);
    (void)hipDeviceSynchronize();
    (void)rccl()->CommDestroy(e->comm);
    e->comm = nullptr;
  }
  e->comm_rank = 0;
  e->comm_world = 1;
  e->node_offset = 0;
  e->hist_epoch = 0;
  return YKPRED_OK;
}

int32_t ykpred_set_row_capacity(ykpred_engine_t* e, int32_t rows) {
  YK_SERIALISE(e);
  if (!e || rows < 0) return fail(e, YKPRED_E_INVALID, "set_row_capacity: bad argument");
  if (rows && e->rows_total > rows) return fail(e, YKPRED_E_INVALID, "set_row_capacity: the bitmap already uses more rows");
  if (rows != e->row_capacity) e->last_eval_valid = false;  // the bitmap buffer changes size
  e->row_capacity = rows;
  e->ask_epoch++;
  return YKPRED_OK;
}

int32_t ykpred_set_row_stride(ykpred_engine_t* e, int32_t words) {
  YK_SERIALISE(e);
  if (!e || words < 0 || words % 16 != 0) return fail(e, YKPRED_E_INVALID, "set_row_stride: need a multiple of 16 words (0 = automatic)");
  e->forced_stride = words;
  e->ask_epoch++;
  return YKPRED_OK;
}

int32_t ykpred_gather_bitmap(ykpred_engine_t* e, void* gathered, void* stream) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:gather_bitmap");
  if (e) e->n_gathers++;
  if (!e) return YKPRED_E_INVALID;
  if (!e->comm) return fail(e, YKPRED_E_STATE, "gather_bitmap: no communicator (ykpred_comm_init)");
  if (!e->last_eval_valid || !e->last_bitmap) return fail(e, YKPRED_E_STATE, "gather_bitmap: no current evaluation");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
  if (e->comm_world > 1 && e->row_capacity == 0)
    return fail(e, YKPRED_E_STATE, "gather_bitmap: the shards must agree on a row capacity first (ykpred_set_row_capacity)");
  const size_t rows = (size_t)std::max(std::max(e->rows_total, e->row_capacity), 1);
  const size_t words = rows * (size_t)e->row_stride;
  if (!gathered) {
    HIPCHK(e->d_gathered.ensure(words * (size_t)e->comm_world * sizeof(u64)));
    gathered = e->d_gathered.p;
  }
  NCCLCHK(rccl()->AllGather(e->last_bitmap, gathered, words, ncclUint64, e->comm, st));
  // every shard lays its rows out for its own writer: the row_of_pod maps travel with the bitmaps
  HIPCHK(e->d_gathered_map.ensure((size_t)std::max(e->P, 1) * (size_t)e->comm_world * sizeof(int)));
  if (e->P) NCCLCHK(rccl()->AllGather(e->d_pod_row.p, e->d_gathered_map.p, (size_t)e->P, ncclInt32, e->comm, st));
  return YKPRED_OK;
}

// ---- class-compressed gather -------------------------------------------------------------------------------------------------
namespace {
uint64_t layout_digest(ykpred_engine_t* e) {
  if (e->layout_hashed_version == e->layout_version) return e->layout_hash_value;
  uint64_t h = 1469598103934665603ull;  // FNV-1a over everything the expansion of a REMOTE class-row table relies on
  auto mix = [&](uint64_t v) {
    h ^= v;
    h *= 1099511628211ull;
  };
  mix((uint64_t)e->C);
  mix((uint64_t)e->P);
  mix((uint64_t)e->row_stride);  // (not row_words: shards of unequal node counts share the stride, class rows travel at full stride)
  mix((uint64_t)std::max(e->rows_total, e->row_capacity));
  mix((uint64_t)e->rows_a);
  for (int p = 0; p < e->P; ++p) mix(((uint64_t)(uint32_t)e->h_pod_row[(size_t)p] << 32) | (uint32_t)e->h_pod_class[(size_t)p]);
  e->layout_hash_value = h;
  e->layout_hashed_version = e->layout_version;
  return h;
}
// the bitmap whose class rows are `class_rows` ([C][row_stride], device), written with this engine's row layout
int expand_class_rows_into(ykpred_engine_t* e, const u64* class_rows, u64* out, hipStream_t st) {
  const int C = e->C, P = e->P;
  if (C == 0 || P == 0) return YKPRED_OK;
  if (e->ident_classes < C) {
    HIPCHK(e->d_class_sig_ident.ensure((size_t)C * 4 * sizeof(int)));
    hipLaunchKernelGGL(ykk::k_identity_sigs, dim3((unsigned)((C + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, C, e->d_class_sig_ident.as<int>());
    e->ident_classes = C;
  }
  HIPCHK(e->d_expand_count.ensure((size_t)C * sizeof(int)));
  ykk::ClassTable ct{e->d_class_sig_ident.as<int>(), e->d_class_pin.as<int>(), e->d_chunk_class.as<int>(), e->d_chunk_begin.as<int>(),
                     e->d_chunk_len.as<int>(), e->d_chunk_first.as<int>(), e->d_members.as<int>(), e->d_chunk_zone.as<int>(), 0, nullptr};
  ykk::Planes pl{nullptr, class_rows, nullptr, nullptr, e->row_stride, nullptr, 0, nullptr, 0, nullptr, 0, nullptr, 0, 0, 0, 0, 0, nullptr, nullptr, nullptr, nullptr, nullptr};
  if (e->n_classes_a > 0) {
    HIPCHK(e->d_class_rows_slot.ensure((size_t)e->n_classes_a * (size_t)e->row_stride * sizeof(u64)));
    hipLaunchKernelGGL(ykk::k_pick_class_rows, dim3((unsigned)e->n_classes_a), dim3(ykk::kBlock), 0, st, class_rows, e->d_class_list_a.as<int>(),
                       e->n_classes_a, e->row_stride, e->d_class_rows_slot.as<u64>());
    const size_t lds_bytes = (size_t)2 * ykk::kBandClasses * (size_t)e->row_stride * sizeof(u64);
    if (lds_bytes > 64 * 1024)
      HIPCHK(hipFuncSetAttribute((const void*)ykk::k_expand_bands, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes));
    hipLaunchKernelGGL(ykk::k_expand_bands, dim3((unsigned)ykk::kBandGroups), dim3(ykk::kBandBlock), lds_bytes, st, out, e->d_class_rows_slot.as<u64>(),
                       e->d_band_tab.as<ykk::BandEntry>(), e->n_bands, e->row_stride);
    if (e->n_fix_rows > 0)
      hipLaunchKernelGGL(ykk::k_fix_rows, dim3((unsigned)e->n_fix_rows), dim3(ykk::kBlock), 0, st, out, e->d_class_rows_slot.as<u64>(),
                         e->d_fix_row.as<int>(), e->d_fix_slot.as<int>(), e->n_fix_rows, e->row_stride);
  }
  // classes outside the band layout (and rows appended since the last class build): chunk by chunk, like the evaluation
  if ((long)e->NC * e->wave_combine_below > (long)P) {
    hipLaunchKernelGGL(ykk::k_combine_wave, dim3((unsigned)((e->NC + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, st, ct, pl,
                       out, e->row_stride, e->row_stride, 0, e->d_expand_count.as<int>(), e->NC, (const int*)nullptr, (const ykk::SliceDesc*)nullptr,
                       (const int*)nullptr, (const int*)nullptr);
  } else {
    int tpg = ykk::kBlock;
    while (tpg > ykk::kWave && (tpg / 2) * 2 >= e->row_stride) tpg /= 2;
    const int seg = tpg * ykk::kCombineUnroll * 2;
    dim3 grid((unsigned)e->NC, (unsigned)((e->row_stride + seg - 1) / seg));
    hipLaunchKernelGGL(ykk::k_combine, grid, dim3(ykk::kBlock), (size_t)e->combine_lds_bytes, st, ct, pl, out, e->row_stride, e->row_stride, 0,
                       e->d_expand_count.as<int>(), tpg, (const int*)nullptr, (const int*)nullptr, ykk::FixRows{nullptr, nullptr, nullptr, 0}, e->NC);
  }
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
}
// a PEER's class rows (indexed through the peer's ask -> class map) into this engine's row order
int expand_peer_rows_into(ykpred_engine_t* e, const u64* class_rows, const int* pod_class, u64* out, hipStream_t st) {
  if (e->P == 0) return YKPRED_OK;
  const int n_rows = std::max(std::max(e->rows_total, e->row_capacity), 1);
  if (e->row_pod_version != e->layout_version) {
    HIPCHK(e->d_row_pod.ensure((size_t)n_rows * sizeof(int)));
    HIPCHK(hipMemsetAsync(e->d_row_pod.p, 0xff, (size_t)n_rows * sizeof(int), st));
    hipLaunchKernelGGL(ykk::k_invert_row_map, dim3((unsigned)((e->P + ykk::kBlock - 1) / ykk::kBlock)), dim3(ykk::kBlock), 0, st, e->d_pod_row.as<int>(), e->P,
                       e->d_row_pod.as<int>());
    e->row_pod_version = e->layout_version;
  }
  const int row_groups = (n_rows + ykk::kExpandRows - 1) / ykk::kExpandRows;
  hipLaunchKernelGGL(ykk::k_expand_by_row, dim3((unsigned)((row_groups + ykk::kWavesPerBlock - 1) / ykk::kWavesPerBlock)), dim3(ykk::kBlock), 0, st, out, class_rows,
                     pod_class, e->d_row_pod.as<int>(), n_rows, e->row_stride);
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
}
int collect_class_rows_into(ykpred_engine_t* e, u64* out, hipStream_t st) {
  if (e->C == 0) return YKPRED_OK;
  hipLaunchKernelGGL(ykk::k_collect_class_rows, dim3((unsigned)e->C), dim3(ykk::kBlock), 0, st, (const u64*)e->last_bitmap, e->C,
                     e->d_class_first.as<int>(), e->d_pod_row.as<int>(), e->row_stride, out);
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
}
}  // namespace

int32_t ykpred_layout_hash(ykpred_engine_t* e, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return fail(e, YKPRED_E_INVALID, "layout_hash: null argument");
  if (e->classes_dirty || (int)e->h_pod_row.size() < e->P) return fail(e, YKPRED_E_STATE, "layout_hash: no current class build (run ykpred_eval)");
  *out = layout_digest(e);
  return YKPRED_OK;
}

int32_t ykpred_collect_class_rows(ykpred_engine_t* e, void* out, void* stream) {
  YK_SERIALISE(e);
  if (!e || !out) return fail(e, YKPRED_E_INVALID, "collect_class_rows: null argument");
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap) return fail(e, YKPRED_E_STATE, "collect_class_rows: no current evaluation");
  HIPCHK(hipSetDevice(e->cfg.device));
  return collect_class_rows_into(e, (u64*)out, stream ? (hipStream_t)stream : e->own_stream);
}

// The whole answer in class-compressed form on the host: [num_classes][row_words] (classes without a live ask: zeros)
int32_t ykpred_read_class_rows(ykpred_engine_t* e, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out) return fail(e, YKPRED_E_INVALID, "read_class_rows: null argument");
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap) return fail(e, YKPRED_E_STATE, "read_class_rows: no current evaluation");
  if (e->C == 0 || e->row_words == 0) return YKPRED_OK;
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = e->peek_stream ? e->peek_stream : e->own_stream;
  if (e->ev_eval_done) HIPCHK(hipStreamWaitEvent(st, e->ev_eval_done, 0));
  const size_t bytes = (size_t)e->C * (size_t)e->row_stride * sizeof(u64);
  HIPCHK(e->d_class_rows_all.ensure(bytes));
  TRY(collect_class_rows_into(e, e->d_class_rows_all.as<u64>(), st));
  HIPCHK(hipMemcpy2DAsync(out, (size_t)e->row_words * sizeof(u64), e->d_class_rows_all.p, (size_t)e->row_stride * sizeof(u64), (size_t)e->row_words * sizeof(u64),
                          (size_t)e->C, hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));
  return YKPRED_OK;
}

int32_t ykpred_expand_class_rows(ykpred_engine_t* e, const void* class_rows, const int32_t* pod_class, void* bitmap_out, void* stream) {
  YK_SERIALISE(e);
  if (!e || !class_rows || !bitmap_out) return fail(e, YKPRED_E_INVALID, "expand_class_rows: null argument");
  if (e->classes_dirty) return fail(e, YKPRED_E_STATE, "expand_class_rows: no current class build (run ykpred_eval)");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
  if (!pod_class) return expand_class_rows_into(e, (const u64*)class_rows, (u64*)bitmap_out, st);
  return expand_peer_rows_into(e, (const u64*)class_rows, (const int*)pod_class, (u64*)bitmap_out, st);
}

int32_t ykpred_gather_bitmap_compressed(ykpred_engine_t* e, void* gathered, void* stream) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:gather_bitmap_compressed");
  if (e) e->n_gathers++;
  if (!e) return YKPRED_E_INVALID;
  if (!e->comm) return fail(e, YKPRED_E_STATE, "gather_bitmap_compressed: no communicator (ykpred_comm_init)");
  if (e->classes_dirty || !e->last_eval_valid || !e->last_bitmap) return fail(e, YKPRED_E_STATE, "gather_bitmap_compressed: no current evaluation");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
  const int G = e->comm_world;
  if (G > 1 && e->row_capacity == 0)
    return fail(e, YKPRED_E_STATE, "gather_bitmap_compressed: the shards must agree on a row capacity first (ykpred_set_row_capacity)");
  Rccl* r = rccl();
  // header exchange: {layout digest, class count, ask count} of every shard. Shards merge signatures relative to their OWN
  // dictionaries, so class counts (and layouts) may differ; a peer with my digest is expanded with my writer tables, any
  // other peer ask by ask through its pod -> class map.
  struct Header {
    uint64_t digest, classes, pods, pad;
  };
  std::vector<Header> hdr((size_t)G);
  const Header mine{layout_digest(e), (uint64_t)e->C, (uint64_t)e->P, 0};
  static_assert(sizeof(Header) == 4 * sizeof(uint64_t), "header layout");
  if (G > 1 && e->hdr_epoch == e->ask_epoch && e->hdr_cache.size() == (size_t)G * 4) {
    if (e->hdr_cache[(size_t)e->comm_rank * 4] != mine.digest)  // cannot happen by the rule below; an error beats a lone rank in a collective
      return fail(e, YKPRED_E_STATE, "gather_bitmap_compressed: the class layout moved without an ask-table call");
    // Headers only move with the class layout, and that only moves in calls every shard makes (ask_epoch): the exchange — a
    // host round trip that would hold the host thread until the evaluation in front of it has drained — runs once per epoch;
    // in the steady state of a scheduling cycle the gather is enqueued without waiting for anything.
    memcpy(hdr.data(), e->hdr_cache.data(), (size_t)G * sizeof(Header));
  } else if (G > 1) {
    HIPCHK(e->d_layout_hash.ensure((size_t)(G + 1) * sizeof(Header)));
    HIPCHK(hipMemcpyAsync(e->d_layout_hash.p, &mine, sizeof(mine), hipMemcpyHostToDevice, st));
    NCCLCHK(r->AllGather(e->d_layout_hash.p, (char*)e->d_layout_hash.p + sizeof(Header), sizeof(Header) / 8, ncclUint64, e->comm, st));
    HIPCHK(hipMemcpyAsync(hdr.data(), (char*)e->d_layout_hash.p + sizeof(Header), (size_t)G * sizeof(Header), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    e->hdr_cache.resize((size_t)G * 4);
    memcpy(e->hdr_cache.data(), hdr.data(), (size_t)G * sizeof(Header));
    e->hdr_epoch = e->ask_epoch;
  } else {
    hdr[0] = mine;
  }
  uint64_t cmax = 1;
  bool same_everywhere = true;
  for (const Header& h : hdr) {
    if (h.pods != mine.pods) return fail(e, YKPRED_E_STATE, "gather_bitmap_compressed: the shards hold different ask tables");
    cmax = std::max(cmax, h.classes);
    same_everywhere = same_everywhere && h.digest == mine.digest;
  }
  const size_t rows = (size_t)std::max(std::max(e->rows_total, e->row_capacity), 1);
  const size_t words = rows * (size_t)e->row_stride;
  if (!gathered) {
    HIPCHK(e->d_gathered.ensure(words * (size_t)G * sizeof(u64)));
    gathered = e->d_gathered.p;
  }
  const size_t class_words = (size_t)cmax * (size_t)e->row_stride;  // every shard sends cmax rows (its own count padded)
  HIPCHK(e->d_class_rows_all.ensure(class_words * sizeof(u64)));
  HIPCHK(e->d_gathered_classes.ensure(class_words * (size_t)G * sizeof(u64)));
  TRY(collect_class_rows_into(e, e->d_class_rows_all.as<u64>(), st));
  NCCLCHK(r->AllGather(e->d_class_rows_all.p, e->d_gathered_classes.p, class_words, ncclUint64, e->comm, st));
  if (!same_everywhere && e->P) {  // the peers' pod -> class maps
    HIPCHK(e->d_gathered_pod_class.ensure((size_t)e->P * (size_t)G * sizeof(int)));
    NCCLCHK(r->AllGather(e->d_pod_class.p, e->d_gathered_pod_class.p, (size_t)e->P, ncclInt32, e->comm, st));
  }
  for (int g = 0; g < G; ++g) {
    u64* slab = (u64*)gathered + (size_t)g * words;
    if (g == e->comm_rank && (void*)slab == e->last_bitmap) continue;  // the evaluation wrote this rank's slab itself
    const u64* rows_g = e->d_gathered_classes.as<u64>() + (size_t)g * class_words;
    if (hdr[(size_t)g].digest == mine.digest) {
      TRY(expand_class_rows_into(e, rows_g, slab, st));
    } else {
      TRY(expand_peer_rows_into(e, rows_g, e->d_gathered_pod_class.as<int>() + (size_t)g * (size_t)e->P, slab, st));
    }
  }
  HIPCHK(hipGetLastError());
  // every slab is written in THIS engine's row order: ykpred_read_gathered reads all of them through this engine's map
  HIPCHK(e->d_gathered_map.ensure((size_t)std::max(e->P, 1) * (size_t)G * sizeof(int)));
  for (int g = 0; g < G && e->P; ++g)
    HIPCHK(hipMemcpyAsync(e->d_gathered_map.as<int>() + (size_t)g * (size_t)e->P, e->d_pod_row.p, (size_t)e->P * sizeof(int), hipMemcpyDeviceToDevice, st));
  return YKPRED_OK;
}

int32_t ykpred_exchange_decisions(ykpred_engine_t* e, void* stream) {
  YK_SERIALISE(e);
  Range roctx_range("ykpred:exchange_decisions");
  if (!e) return YKPRED_E_INVALID;
  if (!e->comm) return fail(e, YKPRED_E_STATE, "exchange_decisions: no communicator (ykpred_comm_init)");
  if (!e->last_eval_valid || !e->last_has_keys)
    return fail(e, YKPRED_E_STATE, "exchange_decisions: the last ykpred_eval must have produced counts, decisions and decision keys");
  HIPCHK(hipSetDevice(e->cfg.device));
  hipStream_t st = stream ? (hipStream_t)stream : e->own_stream;
  const size_t P = (size_t)e->P;
  if (P == 0) return YKPRED_OK;
  HIPCHK(e->d_xkey.ensure(P * sizeof(i64)));
  HIPCHK(e->d_xcand.ensure(P * sizeof(int)));
  Rccl* r = rccl();
  NCCLCHK(r->AllReduce(e->last_counts, e->last_counts, P, ncclInt32, ncclSum, e->comm, st));
  NCCLCHK(r->AllReduce(e->last_keys, e->d_xkey.p, P, ncclInt64, ncclMin, e->comm, st));
  const unsigned blocks = (unsigned)((P + ykk::kBlock - 1) / ykk::kBlock);
  hipLaunchKernelGGL(ykk::k_decision_candidates, dim3(blocks), dim3(ykk::kBlock), 0, st, (int)P, (const int*)e->last_decisions,
                     (const i64*)e->last_keys, e->d_xkey.as<i64>(), e->node_offset, e->d_xcand.as<int>());
  NCCLCHK(r->AllReduce(e->d_xcand.p, e->d_xcand.p, P, ncclInt32, ncclMin, e->comm, st));
  hipLaunchKernelGGL(ykk::k_decision_finalize, dim3(blocks), dim3(ykk::kBlock), 0, st, (int)P, e->d_xcand.as<int>(), e->d_xkey.as<i64>(),
                     (int*)e->last_decisions, (i64*)e->last_keys);
  HIPCHK(hipGetLastError());
  return YKPRED_OK;
}

int32_t ykpred_read_gathered(ykpred_engine_t* e, int32_t shard, int32_t first, int32_t num, uint64_t* out) {
  YK_SERIALISE(e);
  if (!e || !out || shard < 0 || shard >= e->comm_world || first < 0 || num < 0 || first + num > e->P) return fail(e, YKPRED_E_INVALID, "read_gathered: range");
  if (!e->d_gathered.p) return fail(e, YKPRED_E_STATE, "read_gathered: no engine-owned gathered bitmap (ykpred_gather_bitmap with gathered = NULL)");
  HIPCHK(hipSetDevice(e->cfg.device));
  HIPCHK(hipDeviceSynchronize());
  if (num == 0) return YKPRED_OK;
  const size_t rows = (size_t)std::max(std::max(e->rows_total, e->row_capacity), 1);
  const u64* shard_bitmap = e->d_gathered.as<u64>() + (size_t)shard * rows * (size_t)e->row_stride;
  const int* shard_map = e->d_gathered_map.as<int>() + (size_t)shard * (size_t)e->P;
  HIPCHK(e->d_scratch.ensure((size_t)num * (size_t)e->row_stride * sizeof(u64)));
  hipLaunchKernelGGL(ykk::k_gather_rows, dim3((unsigned)num), dim3(ykk::kBlock), 0, e->own_stream, shard_bitmap, num, (const int*)nullptr, first, shard_map,
                     e->row_stride, e->row_stride, e->d_scratch.as<u64>());
  HIPCHK(hipGetLastError());
  HIPCHK(hipStreamSynchronize(e->own_stream));
  HIPCHK(hipMemcpy(out, e->d_scratch.p, (size_t)num * (size_t)e->row_stride * sizeof(u64), hipMemcpyDeviceToHost));
  return YKPRED_OK;
}

}  // extern "C"
