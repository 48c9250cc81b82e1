// host.cpp — libykhost.so: SchedulerCache mirror + encoder + GpuPredicateManager above the C ABI of libykpred.so.
//
// Stand-in for the Go layer of the drop-in (see include/ykhost.h). It never evaluates a
// predicate itself: every fit / no-fit verdict and every failing-plugin code comes back from the engine
// (ykpred_eval / ykpred_query / ykpred_preemption). The only string work done here for a verdict is composing
// the human-readable status message AFTER the device has named the failing plugin.
#include "../../../include/ykhost.h"

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <exception>
#include <limits>
#include <map>
#include <memory>
#include <mutex>
#include <set>
#include <sched.h>
#include <string>
#include <system_error>
#include <thread>
#include <unordered_map>
#include <vector>

#include "encoder.h"
#include "jsonscan.h"
#include "objects.h"

namespace ykh {

// Default plugin lists of NewPredicateManager (predicate_manager.go:321-373), restricted to the engine's plugins.
constexpr uint32_t kAllPlugins = YKPRED_PLUGIN_ALL;
constexpr uint32_t kReservationPre = YKPRED_PLUGIN_NODE_AFFINITY | YKPRED_PLUGIN_NODE_PORTS | YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD |
                                     YKPRED_PLUGIN_INTER_POD_AFFINITY;
constexpr uint32_t kReservationFilt = YKPRED_PLUGIN_NODE_UNSCHEDULABLE | YKPRED_PLUGIN_NODE_NAME | YKPRED_PLUGIN_TAINT_TOLERATION |
                                      YKPRED_PLUGIN_NODE_AFFINITY | YKPRED_PLUGIN_NODE_PORTS | YKPRED_PLUGIN_POD_TOPOLOGY_SPREAD |
                                      YKPRED_PLUGIN_INTER_POD_AFFINITY;

static const char* kPluginNames[] = {"", "NodeUnschedulable", "NodeName", "TaintToleration", "NodeAffinity", "NodePorts",
                                     "NodeResourcesFit", "PodTopologySpread", "InterPodAffinity"};

struct Rng {  // splitmix64
  uint64_t s;
  uint64_t next() {
    uint64_t z = (s += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
  }
  uint32_t below(uint32_t n) { return n ? (uint32_t)(next() % n) : 0; }
  bool chance(uint32_t num, uint32_t den) { return below(den) < num; }
};

}  // namespace ykh

using namespace ykh;

// The encoded (structure-of-arrays) form of the mirror: exactly what crosses the C ABI in ykpred_set_nodes / set_specs.
// A vector whose resize() leaves new elements uninitialised: the per-spec columns of a full encode are sized once and then written
// whole by the threads that encode the rows — the first touch of a fresh page (the kernel's zero fill, a fault each) happens on
// those threads instead of one after the other on the caller's.
template <class T>
struct NoInitAlloc {
  using value_type = T;
  NoInitAlloc() = default;
  template <class U>
  NoInitAlloc(const NoInitAlloc<U>&) {}
  T* allocate(size_t n) { return std::allocator<T>().allocate(n); }
  void deallocate(T* p, size_t n) { std::allocator<T>().deallocate(p, n); }
  template <class U, class... A>
  void construct(U* p, A&&... a) {
    if constexpr (sizeof...(A) == 0) ::new ((void*)p) U;
    else ::new ((void*)p) U(std::forward<A>(a)...);
  }
  template <class U>
  bool operator==(const NoInitAlloc<U>&) const { return true; }
  template <class U>
  bool operator!=(const NoInitAlloc<U>&) const { return false; }
};
template <class T>
using RawVec = std::vector<T, NoInitAlloc<T>>;
struct EncodedTables {
  std::vector<uint64_t> ports, taints, labels, wanted;
  RawVec<uint64_t> stol, aff_terms, pre_terms;
  std::vector<int64_t> alloc, req;
  RawVec<int64_t> sreq;
  std::vector<int32_t> allowed, count, domain, selcount, dsizes, name_rank;
  RawVec<int32_t> aff_off, pre_off, spread_off;
  std::vector<uint32_t> flags;
  RawVec<uint32_t> sflags;
  RawVec<ykpred_spread_t> spread;
  ykpred_nodes_t nt{};
  ykpred_specs_t sp{};
  uint64_t dummy = 0;
  ykpred_spread_t no_spread{};
};

// podsMap of the SchedulerCache (scheduler_cache.go:57): uid -> the cached version of the pod. Sharded by a hash of the uid so
// that a bulk load can fill the shards on different threads (update_pods_bulk); every other caller sees the subset of the
// std::unordered_map interface it used before (find / end / operator[] / erase / iteration).
struct UidIndex {
  static constexpr int kShards = 64;
  using Map = std::unordered_map<std::string, Pod*>;
  Map shard[kShards];
  static unsigned shard_of(const std::string& uid) { return (unsigned)((std::hash<std::string>{}(uid) * 0x9E3779B97F4A7C15ull) >> 58); }
  struct iterator {
    UidIndex* ix;
    int s;
    Map::iterator it;
    Map::value_type& operator*() const { return *it; }
    Map::value_type* operator->() const { return &*it; }
    bool operator==(const iterator& o) const { return s == o.s && (s == kShards || it == o.it); }
    bool operator!=(const iterator& o) const { return !(*this == o); }
    void settle() {
      while (s < kShards && it == ix->shard[s].end()) {
        ++s;
        if (s < kShards) it = ix->shard[s].begin();
      }
    }
    iterator& operator++() {
      ++it;
      settle();
      return *this;
    }
  };
  iterator end() { return iterator{this, kShards, Map::iterator{}}; }
  iterator begin() {
    iterator i{this, 0, shard[0].begin()};
    i.settle();
    return i;
  }
  iterator find(const std::string& uid) {
    const int s = (int)shard_of(uid);
    auto it = shard[s].find(uid);
    return it == shard[s].end() ? end() : iterator{this, s, it};
  }
  Pod* lookup(const std::string& uid) const {  // (read-only: safe from several threads while nobody writes the index)
    const Map& m = shard[shard_of(uid)];
    auto it = m.find(uid);
    return it == m.end() ? nullptr : it->second;
  }
  Pod*& operator[](const std::string& uid) { return shard[shard_of(uid)][uid]; }
  void erase(iterator i) { shard[i.s].erase(i.it); }
  size_t erase(const std::string& uid) { return shard[shard_of(uid)].erase(uid); }
  void clear() {
    for (Map& m : shard) m.clear();
  }
  void reserve(size_t n) {
    for (Map& m : shard) m.reserve(n / kShards + n / (8 * kShards) + 16);
  }
  size_t size() const {
    size_t n = 0;
    for (const Map& m : shard) n += m.size();
    return n;
  }
};

struct ykhost {
  // Context.IsPodFitNode runs under read locks only (context.go:697,709), so several core goroutines may be inside
  // Predicates() at once while the mirror's answer cache and lazy sync mutate state: one lock around every entry point.
  mutable std::recursive_mutex mu;
  ykpred_engine_t* eng = nullptr;
  int device = 0;
  std::string err;
  uint32_t res_pre = kReservationPre, alloc_pre = kAllPlugins, res_filt = kReservationFilt, alloc_filt = kAllPlugins;

  // ---- SchedulerCache mirror
  TemplatePool pool;
  // raw template text (namespace | labels | spec without nodeName) → interned template: the pods of one Deployment / task group
  // differ in name and uid only, so all but the first skip the JSON tree, read_template and the canonical re-serialisation
  std::unordered_map<std::string, const PodTemplate*> tpl_memo;
  int64_t ingest_fast = 0, ingest_full = 0;  // pod documents that took the memo / the full parser
  int64_t ingest_threads = 0, ingest_scan_us = 0, ingest_apply_us = 0, ingest_parallel_batches = 0, ingest_bulk_batches = 0;  // ykhost_ingest_timing
  std::deque<NodeInfo> node_store;
  std::vector<NodeInfo*> nodes;  // index = engine node index
  std::unordered_map<std::string, int> node_ix;
  std::deque<Pod> pod_store;
  std::deque<std::vector<Pod>> pod_blocks;  // pods of bulk loads (update_pods_bulk): one block per scanned piece, filled on its thread
  std::vector<Pod*> free_pods;  // slots of pod versions nothing refers to any more (reused by the next update)
  std::vector<Pod*> pending;  // index = engine pod index
  UidIndex by_uid;
  bool uid_index = true;  // false for generated clusters until first needed

  // ---- encoder state
  Encoder enc;
  bool dirty_all = true, dirty_pods = false;
  std::vector<int> dirty_nodes;       // node rows to re-upload
  std::vector<int> eval_dirty_nodes;  // node columns changed since the last evaluation
  std::vector<int> dirty_rows;        // ask rows to re-upload (ykpred_update_pods)
  std::vector<int> eval_dirty_rows;   // bitmap rows changed since the last evaluation
  bool table_shrunk = false;          // the ask table lost rows at its end since the last upload
  bool decisions_stale = true;        // something was (re)evaluated without decisions since they were last produced
  uint64_t placeholder_serial = 0;    // nonce source of generated placeholder names
  std::shared_ptr<EncodedTables> tables;  // what the last full encode uploaded; spec rows are appended for new templates
  bool specs_dirty = false;               // spec rows were appended since the last ykpred_set_specs
  bool fx_uploaded = false;               // ykpred_set_spec_effects describes the spec table the engine holds (allocation rounds)
  int64_t fx_uploads = 0;
  // answers of one ask against every node (ykpred_query_pod), so that the core's per-node Predicates() callbacks of a
  // scheduling attempt are served from host memory; dropped whenever any table changes
  struct AskAnswers {
    int pod = -1, phase = -1;
    std::vector<uint8_t> fit, code;
    std::vector<uint32_t> reason;
  } answers;
  // The RESIDENT answer. yunikorn-core walks asks, and for one ask calls Predicates() node after node; when an evaluation is
  // current every one of those answers already sits in the bitmap. The mirror holds it in class-compressed form on the host
  // ([classes][row_words], fetched once per evaluation on the first callback): a callback is then a lookup — ask → class
  // (the engine's class index) → bit. Node columns that changed since the mirror was taken (AssumePod between two asks) are
  // answered per pair by the device straight from the tables (ykpred_query); only a pair that does NOT fit needs the device
  // for its failing plugin (one packed ykpred_query_pod per class, cached).
  struct Resident {
    bool valid = false;
    int phase = -1, C = 0, N = 0, row_words = 0;
    std::vector<uint64_t> rows;        // [C][row_words]; empty when C * row_words is over the budget (then: one row per ask, below)
    std::vector<uint8_t> node_dirty;   // [N] 1 = the node changed since the mirror was taken
    int n_dirty = 0;
    std::unordered_map<int, std::vector<uint32_t>> class_codes;  // class → packed answers of ykpred_query_pod_packed (clean columns)
    int peek_pod = -1;                 // per-ask form: the row of the ask that was asked about last
    std::vector<uint64_t> peek_row;
    std::vector<int32_t> order;        // bin-pack order of the evaluation (perm), fetched on the first ykhost_candidates call
  } resident;
  int64_t resident_budget_bytes = 256ll << 20;  // YKHOST_RESIDENT_MB
  int64_t served_resident = 0, served_dirty_column = 0, served_query = 0, resident_fetches = 0, code_fetches = 0;
  int last_eval_phase = -1;           // 1 allocate / 0 reserve / -1 none: the phase of the bitmap on the device
  uint32_t last_eval_options = 0;
  bool dump_compact = false;                 // ykhost_set_dump_compact
  int row_stride_words = 0;                  // ykhost_set_row_stride: bitmap row stride shared by the shards of a cluster
  int row_capacity = 0;                      // ykhost_set_row_capacity: bitmap rows shared by the shards of a cluster
  bool comm_attached = false;                // ykhost_comm_init: the engine carries an RCCL communicator
  std::vector<PodTemplate*> spec_templates;  // spec id → template
  int64_t last_encode_us = 0;
  int64_t unsupported_asks = 0;    // asks whose template the encoder marked unsupported at the last full encode
  int64_t routed_to_cpu = 0;
  int64_t device_errors = 0;       // engine calls that came back YKPRED_E_DEVICE / YKPRED_E_NOMEM (each forces a full re-upload)
  int64_t rounds_on_device = 0, round_asks_on_device = 0, round_asks_one_by_one = 0, round_asks_routed = 0;
  int64_t dictionary_growths = 0;  // new asks whose selector requirements were added to the dictionaries in place
  // label key → value → nodes carrying it (built on the first dictionary growth, dropped whenever a node object changes): a
  // new requirement bit is then computed per DISTINCT value of its key instead of per node
  std::unordered_map<std::string, std::unordered_map<std::string, std::vector<int32_t>>> label_index;
  bool label_index_valid = false;       // Predicates() calls answered YKHOST_E_UNSUPPORTED (the Go side's fallback counter)
  int cfgR = 0, cfgKT = 0, cfgW = 0, cfgKD = -1, cfgKS = -1, cfgKP = -1;

  void clear_state() {
    pool.clear();
    tpl_memo.clear();
    node_store.clear();
    nodes.clear();
    node_ix.clear();
    pod_store.clear();
    pod_blocks.clear();
    free_pods.clear();
    pending.clear();
    by_uid.clear();
    uid_index = true;
    dirty_all = true;
    label_index_valid = false;
    dirty_nodes.clear();
    eval_dirty_nodes.clear();
    dirty_rows.clear();
    eval_dirty_rows.clear();
    table_shrunk = false;
    last_eval_phase = -1;
    resident.valid = false;
    spec_templates.clear();
  }
};

namespace {

int fail(ykhost* h, const std::string& m, int code = -1) {
  h->err = m;
  if (code == YKPRED_E_DEVICE || code == YKPRED_E_NOMEM) {
    // The engine lost a device call (failed allocation, lost device): nothing on the device can be trusted to describe the
    // mirror any more. The mirror itself is intact — it is the source of truth — so the next sync re-encodes and re-uploads
    // everything and the next evaluation is a full pass; until then callbacks fail and the Go manager routes them to the CPU
    // predicate manager (the engine must degrade, never fail scheduling).
    h->dirty_all = true;
    h->resident.valid = false;
    h->answers.pod = -1;
    h->last_eval_phase = -1;
    h->decisions_stale = true;
    h->device_errors++;
  }
  return code;
}
void copy_out(const std::string& s, char* out, int64_t len) {
  if (!out || len <= 0) return;
  size_t n = std::min((size_t)(len - 1), s.size());
  memcpy(out, s.data(), n);
  out[n] = 0;
}

void ensure_uid_index(ykhost* h) {
  if (h->uid_index) return;
  h->by_uid.clear();
  size_t n = h->pod_store.size();
  for (auto& b : h->pod_blocks) n += b.size();
  h->by_uid.reserve(n);
  for (Pod& p : h->pod_store)
    if (p.tpl) h->by_uid[p.uid] = &p;
  for (auto& b : h->pod_blocks)
    for (Pod& p : b)
      if (p.tpl) h->by_uid[p.uid] = &p;
  h->uid_index = true;
}

Pod* store_pod(ykhost* h, Pod&& p);
Pod* add_pod_object(ykhost* h, const mj::Value& v, size_t* anon) {
  Pod p;
  if (const mj::Value* md = v.get_nn("metadata")) {
    p.uid = md->str_or("uid", "");
    p.name = md->str_or("name", "");
    p.terminating = md->get_nn("deletionTimestamp") != nullptr;
  }
  if (const mj::Value* spec = v.get_nn("spec")) p.node_name = spec->str_or("nodeName", "");
  if (p.uid.empty()) p.uid = "anon-" + std::to_string((*anon)++);
  p.tpl = h->pool.intern(read_template(v));
  return store_pod(h, std::move(p));
}
Pod* store_pod(ykhost* h, Pod&& p) {
  if (!h->free_pods.empty()) {
    Pod* slot = h->free_pods.back();
    h->free_pods.pop_back();
    *slot = std::move(p);
    return slot;
  }
  h->pod_store.push_back(std::move(p));
  return &h->pod_store.back();
}
// A pod version that is in no map, on no node and holds no row any more: its slot can be reused.
void recycle_pod(ykhost* h, Pod* p) {
  *p = Pod{};  // tpl == nullptr marks the slot as empty for ensure_uid_index
  h->free_pods.push_back(p);
}

// ---- encode + upload -------------------------------------------------------------------------------
int recreate_engine(ykhost* h) {
  if (h->eng && h->cfgR == h->enc.R && h->cfgKT == h->enc.KT && h->cfgW == h->enc.W && h->cfgKD == h->enc.KD && h->cfgKS == h->enc.KS &&
      h->cfgKP == h->enc.KP)
    return 0;
  if (h->eng && h->comm_attached)
    return fail(h, "the dictionary shape changed on an engine that carries an RCCL communicator: the shard has to be re-created "
                   "(ykhost_comm_init is collective and cannot be repeated silently)", YKPRED_E_STATE);
  if (h->eng) ykpred_destroy(h->eng);
  h->eng = nullptr;
  h->fx_uploaded = false;
  if (h->device < 0) return fail(h, "mirror-only handle (device < 0): no device engine, nothing can be evaluated", YKPRED_E_STATE);
  ykpred_config_t c{};
  c.abi_version = YKPRED_ABI_VERSION;
  c.device = h->device;
  c.num_resources = h->enc.R;
  c.taint_words = h->enc.KT;
  c.label_words = h->enc.W;
  c.topology_keys = h->enc.KD;
  c.selector_classes = h->enc.KS;
  c.port_words = h->enc.KP;
  // (engine tunables for tests come from YKPRED_TUNE, read by ykpred_create itself)
  int r = ykpred_create(&c, &h->eng);
  if (r != YKPRED_OK) return fail(h, std::string("ykpred_create: ") + ykpred_last_error(nullptr), r);
  h->cfgR = c.num_resources;
  h->cfgKT = c.taint_words;
  h->cfgW = c.label_words;
  h->cfgKD = c.topology_keys;
  h->cfgKS = c.selector_classes;
  h->cfgKP = c.port_words;
  return 0;
}

void refresh_spec_view(ykhost* h, EncodedTables* T);

// Dictionaries + node rows + spec rows from the current objects (no device involved).
// Threads for the batch forms and the node loop of a full encode: YKHOST_INGEST_THREADS, else the cores this process may really use — a container with a CPU quota
// (cgroup v2 cpu.max) still reports every core of its host through hardware_concurrency (the GPU boxes of this pool: 256 visible,
// 16 granted), and 64 scanning threads on 16 cores only add context switches — at most 64.
unsigned host_threads() {
  const char* env = getenv("YKHOST_INGEST_THREADS");  // (read per batch: the tests switch it inside one process)
  const int want = env ? atoi(env) : 0;
  if (want > 0) return (unsigned)want;
  unsigned hw = std::max(1u, std::thread::hardware_concurrency());
  // the CPUs this process may run on (cpuset cgroups, taskset): hardware_concurrency does not look at the affinity mask
  cpu_set_t set;
  CPU_ZERO(&set);
  if (sched_getaffinity(0, sizeof set, &set) == 0) {
    const int n = CPU_COUNT(&set);
    if (n > 0) hw = std::min<unsigned>(hw, (unsigned)n);
  }
  auto cap_by_quota = [&](long long quota, long long period) {
    if (quota > 0 && period > 0) hw = std::min<unsigned>(hw, (unsigned)std::max<long long>(1, (quota + period - 1) / period));
  };
  if (FILE* f = fopen("/sys/fs/cgroup/cpu.max", "r")) {  // cgroup v2: "<quota|max> <period>"
    long long quota = 0, period = 0;
    if (fscanf(f, "%lld %lld", &quota, &period) == 2) cap_by_quota(quota, period);
    fclose(f);
  } else {  // cgroup v1: cpu.cfs_quota_us (-1 = no quota) / cpu.cfs_period_us, under either mount name
    for (const char* dir : {"/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"}) {
      long long quota = 0, period = 0;
      bool have = false;
      if (FILE* q = fopen((std::string(dir) + "/cpu.cfs_quota_us").c_str(), "r")) {
        have = fscanf(q, "%lld", &quota) == 1;
        fclose(q);
      }
      if (!have) continue;
      if (FILE* pf = fopen((std::string(dir) + "/cpu.cfs_period_us").c_str(), "r")) {
        if (fscanf(pf, "%lld", &period) == 1) cap_by_quota(quota, period);
        fclose(pf);
      }
      break;
    }
  }
  return std::min(hw, 64u);
}
// f(i) for every i in [0, n), claimed one at a time by at most `threads` threads — the caller is one of them. The design rule of
// this library is "degrade, never fail scheduling": a thread that cannot be created (std::system_error / EAGAIN under a pids or
// thread cgroup limit) is simply not there and the threads that did start — at worst the caller alone — take its items; an
// exception thrown by f on a worker is caught there, stops the hand-out, and is rethrown on the caller once every thread that
// started has been joined (a joinable std::thread must never be destroyed: std::terminate would take the scheduler down).
// YKHOST_TEST_THREAD_LIMIT=k (tests): thread creation "fails" after k workers.
template <class F>
void run_on_threads(int threads, int n, F&& f) {
  if (n <= 0) return;
  std::atomic<int> next{0};
  std::mutex err_mu;
  std::exception_ptr err;
  auto body = [&]() {
    try {
      for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) f(i);
    } catch (...) {
      std::lock_guard<std::mutex> lock(err_mu);
      if (!err) err = std::current_exception();
      next.store(n);
    }
  };
  const char* lim = getenv("YKHOST_TEST_THREAD_LIMIT");
  const int limit = lim ? atoi(lim) : -1;
  std::vector<std::thread> pool;
  const int want = std::min(threads, n) - 1;
  try {
    pool.reserve((size_t)std::max(want, 0));
    for (int t = 0; t < want; ++t) {
      if (limit >= 0 && t >= limit) throw std::system_error(std::make_error_code(std::errc::resource_unavailable_try_again));
      pool.emplace_back(body);
    }
  } catch (...) {
    // no more threads: carry on with the ones that exist
  }
  body();
  for (auto& th : pool) th.join();
  if (err) std::rethrow_exception(err);
}
int encode_tables(ykhost* h, EncodedTables* T) {
  auto tp0 = std::chrono::steady_clock::now();
  const bool trace = getenv("YKHOST_TRACE_ENCODE") != nullptr;
  h->enc.trace = trace;
  auto lap = [&](const char* what) {
    if (!trace) return;
    auto n = std::chrono::steady_clock::now();
    fprintf(stderr, "encode %s %.1f ms\n", what, std::chrono::duration<double, std::milli>(n - tp0).count());
    tp0 = n;
  };
  // templates of pending asks, in first-use order → spec ids. With many asks the three passes (reset, first use, numbering) are
  // walks over a million scattered objects: they run on the host's cores — a template's id is the number of templates whose first
  // row lies in front of its own, which every range of rows can count for itself.
  h->spec_templates.clear();
  const std::vector<PodTemplate*>& all_templates = h->pool.all();
  const size_t n_pending = h->pending.size();
  const int id_threads = n_pending >= 65536 ? (int)std::min<size_t>(host_threads(), 32) : 1;
  auto atomic_min = [](int32_t* at, int32_t v) {
    int32_t cur = __atomic_load_n(at, __ATOMIC_RELAXED);
    while (v < cur && !__atomic_compare_exchange_n(at, &cur, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {
    }
  };
  bool ids_done = false;
  if (id_threads > 1) {
    try {
      const size_t A = all_templates.size(), K = (size_t)id_threads;
      std::vector<size_t> firsts(K + 1, 0);
      run_on_threads(id_threads, id_threads, [&](int k) {
        for (size_t i = A * (size_t)k / K; i < A * (size_t)(k + 1) / K; ++i) {
          all_templates[i]->spec_id = -1;
          all_templates[i]->first_row = 0x7fffffff;
        }
      });
      run_on_threads(id_threads, id_threads, [&](int k) {
        for (size_t r = n_pending * (size_t)k / K; r < n_pending * (size_t)(k + 1) / K; ++r)
          atomic_min(&const_cast<PodTemplate*>(h->pending[r]->tpl)->first_row, (int32_t)r);
      });
      run_on_threads(id_threads, id_threads, [&](int k) {
        size_t n = 0;
        for (size_t r = n_pending * (size_t)k / K; r < n_pending * (size_t)(k + 1) / K; ++r) n += h->pending[r]->tpl->first_row == (int32_t)r ? 1 : 0;
        firsts[(size_t)k + 1] = n;
      });
      for (size_t k = 0; k < K; ++k) firsts[k + 1] += firsts[k];
      h->spec_templates.assign(firsts[K], nullptr);
      run_on_threads(id_threads, id_threads, [&](int k) {
        size_t id = firsts[(size_t)k];
        for (size_t r = n_pending * (size_t)k / K; r < n_pending * (size_t)(k + 1) / K; ++r) {
          PodTemplate* t = const_cast<PodTemplate*>(h->pending[r]->tpl);
          if (t->first_row != (int32_t)r) continue;
          t->spec_id = (int32_t)id;
          h->spec_templates[id++] = t;
        }
      });
      ids_done = true;
    } catch (const std::exception&) {
      h->spec_templates.clear();  // (no memory for the counters: the one-thread walk below)
    }
  }
  if (!ids_done) {
    for (PodTemplate* t : all_templates) t->spec_id = -1;
    for (Pod* p : h->pending) {
      PodTemplate* t = const_cast<PodTemplate*>(p->tpl);
      if (t->spec_id < 0) {
        t->spec_id = (int32_t)h->spec_templates.size();
        h->spec_templates.push_back(t);
      }
    }
  }
  const bool any_anti = h->pool.any_anti_affinity();
  // the first template of every dictionary shape, in spec order: what build_dictionaries visits when shapes may stand for their
  // templates (it decides; see there). Found by the same kind of walk; a template without a shape id turns the shortcut off.
  std::vector<int32_t> shape_reps;
  bool have_reps = false;
  if (ids_done && h->pool.num_shapes() > 0) {
    try {
      const size_t S0 = h->spec_templates.size(), K = (size_t)id_threads;
      std::vector<int32_t> shape_first(h->pool.num_shapes(), 0x7fffffff);
      std::atomic<bool> unshaped{false};
      run_on_threads(id_threads, id_threads, [&](int k) {
        for (size_t s = S0 * (size_t)k / K; s < S0 * (size_t)(k + 1) / K; ++s) {
          const int32_t sh = h->spec_templates[s]->shape_id;
          if (sh < 0 || (size_t)sh >= shape_first.size()) unshaped.store(true);
          else atomic_min(&shape_first[(size_t)sh], (int32_t)s);
        }
      });
      if (!unshaped.load()) {
        for (int32_t s : shape_first)
          if (s != 0x7fffffff) shape_reps.push_back(s);
        std::sort(shape_reps.begin(), shape_reps.end());
        have_reps = true;
      }
    } catch (const std::exception&) {
      have_reps = false;
    }
  }
  lap("spec ids");
  const Encoder::ParallelFor on_cores = [&](int n, const std::function<void(int)>& f) {
    // (blocks of 256 items: a work item per template would be an atomic per microsecond of work)
    const int kBlock = 256, blocks = (n + kBlock - 1) / kBlock;
    run_on_threads(id_threads, blocks, [&](int b) {
      for (int i = b * kBlock; i < std::min(n, (b + 1) * kBlock); ++i) f(i);
    });
  };
  bool dict_ok = false;
  try {
    dict_ok = h->enc.build_dictionaries(h->nodes, h->spec_templates, any_anti, have_reps ? &shape_reps : nullptr, id_threads > 1 ? &on_cores : nullptr);
  } catch (const std::exception&) {
    // (a worker of the preparation ran out of memory: the build starts from cleared dictionaries — once more, everything on this thread)
    try {
      dict_ok = h->enc.build_dictionaries(h->nodes, h->spec_templates, any_anti, have_reps ? &shape_reps : nullptr, nullptr);
    } catch (const std::exception& ex) {
      return fail(h, std::string("encoder: dictionaries: ") + ex.what(), YKPRED_E_NOMEM);
    }
  }
  if (!dict_ok) return fail(h, "encoder: " + h->enc.error, YKPRED_E_UNSUPPORTED);
  lap("dictionaries");
  h->unsupported_asks = 0;
  if (!h->enc.unsupported.empty())
    for (const Pod* p : h->pending) h->unsupported_asks += h->enc.unsupported.count(p->tpl) ? 1 : 0;
  const int R = h->enc.R, KT = h->enc.KT, W = h->enc.W, KD = h->enc.KD, KS = h->enc.KS, KP = h->enc.KP;
  const size_t N = h->nodes.size();
  T->ports.assign(N * KP + 1, 0);
  T->alloc.assign(N * R, 0);
  T->req.assign(N * R, 0);
  T->allowed.assign(N, 0);
  T->count.assign(N, 0);
  T->domain.assign(N * KD + 1, 0);
  T->selcount.assign(N * KS + 1, 0);
  T->flags.assign(N, 0);
  T->taints.assign(N * KT, 0);
  T->labels.assign(N * W, 0);
  // One row per node. The rows are independent; what is shared is read-only after build_dictionaries except two caches: the
  // label-tuple memo (every thread gets its own) and the selector-class memo of encode_node_spread (KS > 0: one thread).
  auto encode_rows = [&](size_t n0, size_t n1, Encoder::LabelMemo* memo) {
    std::vector<uint64_t> p1(KP + 1), t1(KT), l1(W);
    std::vector<int64_t> a1(R), r1(R);
    std::vector<int32_t> d1(KD + 1), s1(KS + 1);
    for (size_t n = n0; n < n1; ++n) {
      h->nodes[n]->index = (int32_t)n;
      h->enc.encode_node(*h->nodes[n], a1.data(), r1.data(), &T->allowed[n], &T->count[n], &T->flags[n], t1.data(), l1.data(), memo);
      h->enc.encode_node_spread(*h->nodes[n], d1.data(), s1.data());
      for (int k = 0; k < KD; ++k) T->domain[(size_t)k * N + n] = d1[(size_t)k];
      for (int k = 0; k < KS; ++k) T->selcount[(size_t)k * N + n] = s1[(size_t)k];
      h->enc.encode_ports(h->nodes[n]->pods, p1.data());
      for (int k = 0; k < KP; ++k) T->ports[(size_t)k * N + n] = p1[(size_t)k];
      for (int r = 0; r < R; ++r) {
        T->alloc[(size_t)r * N + n] = a1[(size_t)r];
        T->req[(size_t)r * N + n] = r1[(size_t)r];
      }
      for (int k = 0; k < KT; ++k) T->taints[(size_t)k * N + n] = t1[(size_t)k];
      for (int w = 0; w < W; ++w) T->labels[(size_t)w * N + n] = l1[(size_t)w];
    }
  };
  const int threads = (KS == 0 && N >= 4096) ? (int)std::min<size_t>(host_threads(), N / 1024) : 1;
  bool rows_done = false;
  if (threads > 1) {
    // (one node range and one label memo per item; an item runs on exactly one thread, whichever claims it)
    try {
      std::vector<Encoder::LabelMemo> memos((size_t)threads);
      run_on_threads(threads, threads, [&](int t) {
        encode_rows(N * (size_t)t / (size_t)threads, N * (size_t)(t + 1) / (size_t)threads, &memos[(size_t)t]);
      });
      rows_done = true;
    } catch (const std::exception&) {
      // a worker ran out of memory: the rows are independent and idempotent — once more on this thread alone
    }
  }
  if (!rows_done) {
    try {
      encode_rows(0, N, nullptr);
    } catch (const std::exception& ex) {
      return fail(h, std::string("encoder: node rows: ") + ex.what(), YKPRED_E_NOMEM);
    }
  }
  lap("node rows");
  ykpred_nodes_t& nt = T->nt;
  nt = ykpred_nodes_t{};
  nt.count = (int32_t)N;
  nt.allocatable = T->alloc.data();
  nt.requested = T->req.data();
  nt.allowed_pods = T->allowed.data();
  nt.pod_count = T->count.data();
  nt.flags = T->flags.data();
  nt.taint_bits = T->taints.data();
  nt.label_bits = T->labels.data();
  T->dsizes = h->enc.domain_sizes();
  T->dsizes.push_back(0);
  nt.domain_id = T->domain.data();
  nt.selector_count = T->selcount.data();
  nt.domain_sizes = T->dsizes.data();
  nt.port_bits = T->ports.data();
  {  // NodeID order: the tie-break of the bin-pack order between nodes of equal score
    std::vector<int32_t> by_name(N);
    for (size_t n = 0; n < N; ++n) by_name[n] = (int32_t)n;
    std::sort(by_name.begin(), by_name.end(), [&](int32_t x, int32_t y) { return h->nodes[(size_t)x]->node.name < h->nodes[(size_t)y]->node.name; });
    T->name_rank.assign(N, 0);
    for (size_t r = 0; r < N; ++r) T->name_rank[(size_t)by_name[r]] = (int32_t)r;
    nt.name_rank = T->name_rank.data();
  }

  lap("name ranks");
  const size_t S = h->spec_templates.size();
  T->sreq.resize(S * R);  // (left uninitialised: every row is written whole by the block that encodes it)
  T->stol.resize(S * KT);
  T->wanted.assign(S * KP + 1, 0);
  T->sflags.resize(S);
  T->aff_terms.clear();
  T->pre_terms.clear();
  T->aff_off.assign(1, 0);
  T->pre_off.assign(1, 0);
  T->spread_off.assign(1, 0);
  T->spread.clear();
  // One row per spec. encode_spec only READS the dictionaries, so the rows are encoded on the host's cores, a block of specs at a
  // time: the fixed-width columns are written in place, the variable-length ones (Filter / PreFilter DNF terms, topology
  // constraints) into the block's own vectors, which are then appended in block order — the tables are the one-thread encode's.
  struct SpecBlock {
    std::vector<ykpred_spread_t> spread;
    std::vector<uint64_t> aff, pre;
    std::vector<int32_t> n_spread, n_aff, n_pre;  // per spec of the block
  };
  const size_t kSpecBlock = 2048;
  const size_t n_blocks = (S + kSpecBlock - 1) / kSpecBlock;
  std::vector<SpecBlock> blocks(n_blocks);
  auto encode_block = [&](int b) {
    SpecBlock& blk = blocks[(size_t)b];
    const size_t s0 = (size_t)b * kSpecBlock, s1 = std::min(S, s0 + kSpecBlock);
    blk.n_spread.reserve(s1 - s0);
    blk.n_aff.reserve(s1 - s0);
    blk.n_pre.reserve(s1 - s0);
    blk.aff.reserve((s1 - s0) * (size_t)W * 2);  // (two Filter terms per spec is the usual size: fewer regrowths of the block's column)
    for (size_t s = s0; s < s1; ++s) {
      EncodedSpec es = h->enc.encode_spec(*h->spec_templates[s]);
      blk.spread.insert(blk.spread.end(), es.spread.begin(), es.spread.end());
      blk.n_spread.push_back((int32_t)es.spread.size());
      std::copy(es.req.begin(), es.req.end(), T->sreq.begin() + (long)(s * R));
      std::copy(es.tol.begin(), es.tol.end(), T->stol.begin() + (long)(s * KT));
      T->sflags[s] = es.flags;
      h->enc.encode_wanted_ports(*h->spec_templates[s], T->wanted.data() + s * KP);
      for (auto& t : es.terms) blk.aff.insert(blk.aff.end(), t.begin(), t.end());
      for (auto& t : es.pre_terms) blk.pre.insert(blk.pre.end(), t.begin(), t.end());
      blk.n_aff.push_back((int32_t)es.terms.size());
      blk.n_pre.push_back((int32_t)es.pre_terms.size());
    }
  };
  {
    const int spec_threads = S >= 4 * kSpecBlock ? (int)std::min<size_t>(host_threads(), n_blocks) : 1;
    bool done = false;
    if (spec_threads > 1) {
      try {
        run_on_threads(spec_threads, (int)n_blocks, encode_block);
        done = true;
        lap("spec rows (blocks)");
      } catch (const std::exception&) {
        for (SpecBlock& blk : blocks) blk = SpecBlock{};  // (a worker ran out of memory: once more, on this thread alone)
      }
    }
    try {
      if (!done)
        for (size_t b = 0; b < n_blocks; ++b) encode_block((int)b);
      // The blocks' variable-length columns go into the tables at offsets known from the blocks' sizes: the tables are sized once and
      // every block copies its own piece (and writes the offset rows of its specs) — on the host's cores again.
      std::vector<size_t> at_spread(n_blocks + 1, 0), at_aff(n_blocks + 1, 0), at_pre(n_blocks + 1, 0);
      std::vector<int32_t> n0_spread(n_blocks + 1, 0), n0_aff(n_blocks + 1, 0), n0_pre(n_blocks + 1, 0);  // rows in front of the block
      for (size_t b = 0; b < n_blocks; ++b) {
        const SpecBlock& blk = blocks[b];
        at_spread[b + 1] = at_spread[b] + blk.spread.size();
        at_aff[b + 1] = at_aff[b] + blk.aff.size();
        at_pre[b + 1] = at_pre[b] + blk.pre.size();
        int32_t ns = 0, na = 0, np = 0;
        for (size_t i = 0; i < blk.n_spread.size(); ++i) {
          ns += blk.n_spread[i];
          na += blk.n_aff[i];
          np += blk.n_pre[i];
        }
        n0_spread[b + 1] = n0_spread[b] + ns;
        n0_aff[b + 1] = n0_aff[b] + na;
        n0_pre[b + 1] = n0_pre[b] + np;
      }
      T->spread.resize(at_spread[n_blocks]);
      T->aff_terms.resize(at_aff[n_blocks]);
      T->pre_terms.resize(at_pre[n_blocks]);
      T->spread_off.resize(S + 1);
      T->aff_off.resize(S + 1);
      T->pre_off.resize(S + 1);
      T->spread_off[0] = T->aff_off[0] = T->pre_off[0] = 0;
      auto place_block = [&](int bi) {
        const size_t b = (size_t)bi;
        const SpecBlock& blk = blocks[b];
        std::copy(blk.spread.begin(), blk.spread.end(), T->spread.begin() + (long)at_spread[b]);
        std::copy(blk.aff.begin(), blk.aff.end(), T->aff_terms.begin() + (long)at_aff[b]);
        std::copy(blk.pre.begin(), blk.pre.end(), T->pre_terms.begin() + (long)at_pre[b]);
        int32_t os = n0_spread[b], oa = n0_aff[b], op = n0_pre[b];
        const size_t s0 = b * kSpecBlock;
        for (size_t i = 0; i < blk.n_spread.size(); ++i) {
          os += blk.n_spread[i];
          oa += blk.n_aff[i];
          op += blk.n_pre[i];
          T->spread_off[s0 + i + 1] = os;
          T->aff_off[s0 + i + 1] = oa;
          T->pre_off[s0 + i + 1] = op;
        }
      };
      bool placed = false;
      if (spec_threads > 1) {
        try {
          run_on_threads(spec_threads, (int)n_blocks, place_block);
          placed = true;
        } catch (const std::exception&) {
          // (no thread to be had: the pieces are independent and idempotent — once more on this thread alone)
        }
      }
      if (!placed)
        for (size_t b = 0; b < n_blocks; ++b) place_block((int)b);
    } catch (const std::out_of_range& ex) {
      // Not a memory problem: a template asked the dictionaries for an entry they do not hold — its shape's representative did not
      // register what it needs. The dictionaries are rebuilt once with every template visited; if that does not help either the
      // encoder is wrong about this cluster and says so (the caller keeps the CPU manager), it never reports "out of memory".
      if (h->enc.use_shapes) {
        h->enc.use_shapes = false;
        const int rc = encode_tables(h, T);
        h->enc.use_shapes = true;
        return rc;
      }
      return fail(h, std::string("encoder: spec rows: a template needs a dictionary entry that was not registered (") + ex.what() + ")", YKPRED_E_UNSUPPORTED);
    } catch (const std::exception& ex) {
      return fail(h, std::string("encoder: spec rows: ") + ex.what(), YKPRED_E_NOMEM);
    }
  }
  lap("spec rows");
  refresh_spec_view(h, T);
  return 0;
}

// Points T->sp at the (possibly re-allocated) spec vectors.
void refresh_spec_view(ykhost* h, EncodedTables* T) {
  const size_t S = h->spec_templates.size();
  ykpred_specs_t& sp = T->sp;
  sp = ykpred_specs_t{};
  sp.count = (int32_t)S;
  sp.requests = T->sreq.data();
  sp.tolerated = T->stol.data();
  sp.flags = T->sflags.data();
  sp.aff_term_off = T->aff_off.data();
  sp.aff_terms = T->aff_terms.empty() ? &T->dummy : T->aff_terms.data();
  sp.pre_term_off = T->pre_off.data();
  sp.pre_terms = T->pre_terms.empty() ? &T->dummy : T->pre_terms.data();
  sp.spread_off = T->spread_off.data();
  sp.spread = T->spread.empty() ? &T->no_spread : T->spread.data();
  sp.wanted_ports = T->wanted.data();
}

// A pending ask arrived with a pod template the device has no spec for. If the existing dictionaries cover it, its spec
// row is appended (ykpred_set_specs keeps the pod classes for an append) and the ask is an ordinary new row; otherwise
// everything is re-encoded.
bool append_spec(ykhost* h, const PodTemplate* tpl) {
  if (h->dirty_all || !h->tables || !h->eng) return false;
  EncodedTables& T = *h->tables;
  EncodedSpec es;
  std::vector<uint64_t> wanted;
  if (!h->enc.encode_spec_if_covered(*tpl, &es, &wanted)) {
    // Dictionary growth: selector requirements nobody used before take spare bits of the allocated label words — each new bit
    // is evaluated on every node and only the touched word columns are uploaded (ykpred_update_label_word); nothing else of
    // the cluster is re-encoded.
    std::vector<int> new_bits;
    if (!h->enc.extend_requirements(*tpl, &new_bits)) return false;
    const size_t N = h->nodes.size();
    std::set<int> words;
    if (!h->label_index_valid) {
      h->label_index.clear();
      for (size_t n = 0; n < N; ++n)
        for (auto& kv : h->nodes[n]->node.labels) h->label_index[kv.first][kv.second].push_back((int32_t)n);
      h->label_index_valid = true;
    }
    for (int q : new_bits) {
      const DictReq& d = h->enc.req_dict[(size_t)q];
      uint64_t* col = T.labels.data() + (size_t)(q >> 6) * N;
      const uint64_t bit = 1ull << (q & 63);
      if (d.kind == DictReq::kLabel || d.kind == DictReq::kEquals) {
        // a label requirement depends on the node only through its value of ONE key: In / Equals touch just the nodes of
        // the listed values, the other operators are decided once per distinct value (and once for "key absent")
        auto ix = h->label_index.find(d.req.key);
        const bool listed = d.kind == DictReq::kEquals || d.req.op == "In" || d.req.op == "NotIn";
        const bool negated = d.kind == DictReq::kLabel && d.req.op == "NotIn";
        if (listed) {
          if (negated)
            for (size_t n = 0; n < N; ++n) col[n] |= bit;
          if (ix != h->label_index.end())
            for (auto& v : d.req.values) {
              auto vn = ix->second.find(v);
              if (vn == ix->second.end()) continue;
              for (int32_t n : vn->second) col[(size_t)n] = negated ? (col[(size_t)n] & ~bit) : (col[(size_t)n] | bit);
            }
        } else {
          Node probe;
          const bool absent_matches = d.eval(probe);
          if (absent_matches)
            for (size_t n = 0; n < N; ++n) col[n] |= bit;
          if (ix != h->label_index.end())
            for (auto& vn : ix->second) {
              probe.labels.clear();
              probe.labels[d.req.key] = vn.first;
              const bool m = d.eval(probe);
              if (m == absent_matches) continue;
              for (int32_t n : vn.second) col[(size_t)n] = m ? (col[(size_t)n] | bit) : (col[(size_t)n] & ~bit);
            }
        }
      } else {
        for (size_t n = 0; n < N; ++n)
          if (d.eval(h->nodes[n]->node)) col[n] |= bit;
      }
      words.insert(q >> 6);
    }
    for (int w : words)
      if (ykpred_update_label_word(h->eng, w, T.labels.data() + (size_t)w * N) != YKPRED_OK) return false;
    h->dictionary_growths++;
    if (!h->enc.encode_spec_if_covered(*tpl, &es, &wanted)) return false;
  }
  const size_t W = (size_t)h->enc.W, KP = (size_t)h->enc.KP;
  const size_t S = h->spec_templates.size();
  const_cast<PodTemplate*>(tpl)->spec_id = (int32_t)S;
  h->spec_templates.push_back(const_cast<PodTemplate*>(tpl));
  T.sreq.insert(T.sreq.end(), es.req.begin(), es.req.end());
  T.stol.insert(T.stol.end(), es.tol.begin(), es.tol.end());
  T.sflags.push_back(es.flags);
  T.wanted.resize(S * KP);  // drop the padding element, append, pad again
  T.wanted.insert(T.wanted.end(), wanted.begin(), wanted.begin() + (long)KP);
  T.wanted.push_back(0);
  for (auto& t : es.terms) T.aff_terms.insert(T.aff_terms.end(), t.begin(), t.end());
  for (auto& t : es.pre_terms) T.pre_terms.insert(T.pre_terms.end(), t.begin(), t.end());
  T.aff_off.push_back((int32_t)(T.aff_terms.size() / W));
  T.pre_off.push_back((int32_t)(T.pre_terms.size() / W));
  T.spread.insert(T.spread.end(), es.spread.begin(), es.spread.end());
  T.spread_off.push_back((int32_t)T.spread.size());
  refresh_spec_view(h, &T);
  h->specs_dirty = true;
  return true;
}

int full_sync(ykhost* h) {
  auto t0 = std::chrono::steady_clock::now();
  h->tables = std::make_shared<EncodedTables>();
  h->specs_dirty = false;
  EncodedTables& T = *h->tables;
  int rc = encode_tables(h, &T);
  if (rc) return rc;
  rc = recreate_engine(h);
  if (rc) return rc;
  rc = ykpred_set_row_stride(h->eng, h->row_stride_words);
  if (rc) return fail(h, std::string("ykpred_set_row_stride: ") + ykpred_last_error(h->eng), rc);
  rc = ykpred_set_row_capacity(h->eng, 0);  // (re-applied below: a smaller table may follow a larger one)
  if (rc == YKPRED_OK) rc = ykpred_set_row_capacity(h->eng, h->row_capacity);
  if (rc) return fail(h, std::string("ykpred_set_row_capacity: ") + ykpred_last_error(h->eng), rc);
  rc = ykpred_set_nodes(h->eng, &T.nt);
  if (rc) return fail(h, std::string("ykpred_set_nodes: ") + ykpred_last_error(h->eng), rc);
  h->fx_uploaded = false;
  rc = ykpred_set_specs(h->eng, &T.sp);
  if (rc) return fail(h, std::string("ykpred_set_specs: ") + ykpred_last_error(h->eng), rc);
  h->dirty_all = false;
  h->dirty_nodes.clear();
  h->dirty_pods = true;
  h->last_encode_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  return 0;
}

// (spec, nodeName pin) of one ask row as the engine sees it
void encode_row(ykhost* h, const Pod* pod, int32_t* spec, int32_t* pin) {
  *spec = pod->tpl->spec_id;
  if (pod->node_name.empty() || pod->assumed) {  // an assumed ask keeps evaluating its spec without the pin (see ykhost_assume_pod)
    *pin = YKPRED_NO_NODE_NAME;
  } else {
    auto it = h->node_ix.find(pod->node_name);
    *pin = it == h->node_ix.end() ? YKPRED_UNKNOWN_NODE_NAME : it->second;
  }
}

int pods_sync(ykhost* h) {
  const size_t P = h->pending.size();
  std::vector<int32_t> spec(P), pin(P);
  for (size_t p = 0; p < P; ++p) encode_row(h, h->pending[p], &spec[p], &pin[p]);
  ykpred_pods_t pp{};
  pp.count = (int32_t)P;
  pp.spec_index = spec.data();
  pp.node_name_index = pin.data();
  int rc = ykpred_set_pods(h->eng, &pp);
  if (rc) return fail(h, std::string("ykpred_set_pods: ") + ykpred_last_error(h->eng), rc);
  h->dirty_pods = false;
  h->dirty_rows.clear();
  h->eval_dirty_rows.clear();
  h->table_shrunk = false;
  return 0;
}

// Uploads only the ask rows that changed (new ask, ask bound / removed, node pin changed).
int rows_sync(ykhost* h) {
  std::sort(h->dirty_rows.begin(), h->dirty_rows.end());
  h->dirty_rows.erase(std::unique(h->dirty_rows.begin(), h->dirty_rows.end()), h->dirty_rows.end());
  const int P = (int)h->pending.size();
  while (!h->dirty_rows.empty() && h->dirty_rows.back() >= P) h->dirty_rows.pop_back();  // rows dropped again meanwhile
  std::vector<int32_t> spec(h->dirty_rows.size()), pin(h->dirty_rows.size());
  for (size_t i = 0; i < h->dirty_rows.size(); ++i) encode_row(h, h->pending[(size_t)h->dirty_rows[i]], &spec[i], &pin[i]);
  int32_t none = 0;
  int rc = ykpred_update_pods(h->eng, P, (int32_t)h->dirty_rows.size(), h->dirty_rows.empty() ? &none : h->dirty_rows.data(),
                              spec.empty() ? &none : spec.data(), pin.empty() ? &none : pin.data());
  if (rc) return fail(h, std::string("ykpred_update_pods: ") + ykpred_last_error(h->eng), rc);
  h->eval_dirty_rows.insert(h->eval_dirty_rows.end(), h->dirty_rows.begin(), h->dirty_rows.end());
  h->dirty_rows.clear();
  h->table_shrunk = false;
  return 0;
}

int node_row_sync(ykhost* h, int n) {
  const int R = h->enc.R, KT = h->enc.KT, W = h->enc.W;
  std::vector<int64_t> a((size_t)R), r((size_t)R);
  std::vector<uint64_t> t((size_t)KT), l((size_t)W);
  int32_t allowed, count;
  uint32_t flags;
  h->enc.encode_node(*h->nodes[(size_t)n], a.data(), r.data(), &allowed, &count, &flags, t.data(), l.data());
  std::vector<int32_t> dom((size_t)h->enc.KD + 1), sel((size_t)h->enc.KS + 1);
  if (!h->enc.encode_node_spread(*h->nodes[(size_t)n], dom.data(), sel.data())) {
    h->dirty_all = true;  // a topology value outside the dictionary: rebuild everything
    return full_sync(h);
  }
  std::vector<uint64_t> pb((size_t)h->enc.KP + 1);
  h->enc.encode_ports(h->nodes[(size_t)n]->pods, pb.data());
  ykpred_nodes_t nt{};
  nt.count = 1;
  nt.domain_id = dom.data();
  nt.selector_count = sel.data();
  nt.port_bits = pb.data();
  nt.allocatable = a.data();
  nt.requested = r.data();
  nt.allowed_pods = &allowed;
  nt.pod_count = &count;
  nt.flags = &flags;
  nt.taint_bits = t.data();
  nt.label_bits = l.data();
  const int32_t keep_rank = h->tables && (size_t)n < h->tables->name_rank.size() ? h->tables->name_rank[(size_t)n] : n;
  nt.name_rank = &keep_rank;  // a node keeps its name, hence its place in NodeID order
  int rc = ykpred_update_node(h->eng, n, &nt);
  if (rc) return fail(h, std::string("ykpred_update_node: ") + ykpred_last_error(h->eng), rc);
  return 0;
}

int sync(ykhost* h) {
  if (h->dirty_all || h->dirty_pods || !h->dirty_nodes.empty() || !h->dirty_rows.empty() || h->table_shrunk) h->answers.pod = -1;
  if (h->dirty_all || h->dirty_pods) h->resident.valid = false;  // dictionaries / class index are rebuilt
  if (h->dirty_all) {
    int rc = full_sync(h);
    if (rc) return rc;
  }
  {
    // node_row_sync may fall back to full_sync (a topology value outside the dictionaries), which clears dirty_nodes and
    // uploads every row: iterate over a detached copy and stop at that point
    std::vector<int> todo;
    todo.swap(h->dirty_nodes);
    std::sort(todo.begin(), todo.end());
    todo.erase(std::unique(todo.begin(), todo.end()), todo.end());
    for (int n : todo) {
      if ((size_t)n >= h->nodes.size()) continue;
      const std::shared_ptr<EncodedTables> before = h->tables;
      int rc = node_row_sync(h, n);
      if (rc) return rc;
      if (h->tables != before) break;  // everything was re-encoded and re-uploaded
    }
    h->dirty_nodes.clear();
  }
  if (h->specs_dirty && !h->dirty_all) {
    h->fx_uploaded = false;
    int rc = ykpred_set_specs(h->eng, &h->tables->sp);
    if (rc) return fail(h, std::string("ykpred_set_specs: ") + ykpred_last_error(h->eng), rc);
    h->specs_dirty = false;
    h->answers.pod = -1;
  }
  if (h->dirty_pods) return pods_sync(h);
  if (!h->dirty_rows.empty() || h->table_shrunk) return rows_sync(h);
  return 0;
}

// A node-level change that cannot introduce new dictionary entries (pod assumed / forgotten / removed).
void touch_node(ykhost* h, int n) {
  if (!h->dirty_all) {
    h->dirty_nodes.push_back(n);
    h->eval_dirty_nodes.push_back(n);
  }
  ykhost::Resident& R = h->resident;
  if (R.valid) {
    // topology constraints couple every node through the histograms: one changed node can move any column
    if (h->enc.KD > 0 || n < 0 || n >= R.N) {
      R.valid = false;
    } else if (!R.node_dirty[(size_t)n]) {
      R.node_dirty[(size_t)n] = 1;
      if (++R.n_dirty > 4096) R.valid = false;  // mostly stale: the next evaluation brings a fresh one
    }
  }
}

// ---- SchedulerCache bookkeeping (scheduler_cache.go:303-388) ---------------------------------------------
// Placing a pod on a node may need dictionary entries that were built from the pods already on nodes.
void touch_node_for(ykhost* h, int n, const Pod* p) {
  if (!h->enc.node_pod_known(*p->tpl)) h->dirty_all = true;
  touch_node(h, n);
}
// Drops the assignedPods entry of `p` and takes it off its NodeInfo (:321-340).
void detach_from_node(ykhost* h, Pod* p) {
  if (p->assigned_node.empty()) return;
  auto nt = h->node_ix.find(p->assigned_node);
  if (nt != h->node_ix.end() && h->nodes[(size_t)nt->second]->remove_pod(p->uid, p)) touch_node(h, nt->second);
  p->assigned_node.clear();
}
// cache.updatePod for the new version `p` of the cached pod `old` (null: not cached; == p: re-evaluate in place).
// Returns false when the pod ends up orphaned (:352-360).
bool cache_update_pod(ykhost* h, Pod* old, Pod* p, bool running, bool terminated) {
  if (old) {
    const std::string prev = old->assigned_node;
    const bool was_assumed = old->assumed;
    detach_from_node(h, old);
    old->orphan = false;
    if (!prev.empty() && p->node_name.empty()) p->node_name = prev;  // "new pod wasn't assigned to a node, so use existing assignment"
    p->assumed = was_assumed;
    // (an ask assumed on another shard's node carries that node across its versions, like `assumed` itself: an informer update in
    // front of the bind must not turn "allocated over there" into "no node fits")
    if (old != p) p->remote_node = was_assumed ? old->remote_node : -1;
  }
  if (running || terminated) {  // "pod has now been bound" (:344-347)
    p->assumed = false;
    p->remote_node = -1;
  }
  bool result = true;
  p->orphan = false;
  if (!p->node_name.empty() && !terminated) {
    auto nt = h->node_ix.find(p->node_name);
    if (nt == h->node_ix.end()) {
      p->orphan = true;
      result = false;
    } else {
      h->nodes[(size_t)nt->second]->add_pod(p);
      p->assigned_node = p->node_name;
      touch_node_for(h, nt->second, p);
    }
  }
  if (!terminated) {
    if (old != p) h->by_uid[p->uid] = p;  // (re-evaluated in place — AssumePod, ForgetPod, an adopted orphan: the index holds this very pod)
  } else {
    p->assumed = p->orphan = false;
    h->by_uid.erase(p->uid);
  }
  return result;
}
// Row bookkeeping of the ask table: `old` leaves its row (if it has one), `now` (may be null) takes it over in place or
// is appended. A vacated row is refilled with the LAST row's ask, so the table stays dense and all other rows keep their
// index. Touched rows are re-uploaded one by one (ykpred_update_pods) and re-evaluated alone (ykpred_eval_pods).
void mark_row(ykhost* h, int row) {
  if (h->dirty_all || h->dirty_pods) return;
  h->dirty_rows.push_back(row);
  if (h->dirty_rows.size() > 8192) {  // a bulk change: re-upload the whole ask table instead
    h->dirty_pods = true;
    h->dirty_rows.clear();
  }
}
void set_ask_row(ykhost* h, Pod* old, Pod* now) {
  int row = (old && old->ask) ? old->row : -1;
  if (old) {
    old->ask = false;
    old->row = -1;
  }
  if (now) {
    now->ask = true;
    if (row < 0) {
      row = (int)h->pending.size();
      h->pending.push_back(now);
    } else {
      h->pending[(size_t)row] = now;
    }
    now->row = row;
    if (now->tpl->spec_id < 0 && !append_spec(h, now->tpl))
      h->dirty_all = true;  // new template that needs new dictionary entries
    else
      mark_row(h, row);
  } else if (row >= 0) {
    Pod* last = h->pending.back();
    h->pending.pop_back();
    if (last != old) {
      h->pending[(size_t)row] = last;
      last->row = row;
      mark_row(h, row);
    }
    h->table_shrunk = true;
  }
}

// ---- status message for a device verdict -------------------------------------------------------------
std::string compose_message(ykhost* h, const Pod& pod, const NodeInfo& ni, int code, uint32_t reason) {
  switch (code) {
    case YKPRED_CODE_NONE: return "node(s) didn't match Pod's node affinity/selector";  // PreFilter rejected (conflicting terms)
    case YKPRED_CODE_NODE_UNSCHEDULABLE: return "node(s) were unschedulable";
    case YKPRED_CODE_NODE_NAME: return "node(s) didn't match the requested node name";
    case YKPRED_CODE_TAINT_TOLERATION: {
      for (auto& t : ni.node.taints) {
        if (t.effect != "NoSchedule" && t.effect != "NoExecute") continue;
        bool ok = false;
        for (auto& tol : pod.tpl->tolerations) ok = ok || toleration_tolerates(tol, t);
        if (!ok) return "node(s) had untolerated taint {" + t.key + ": " + t.value + "}";
      }
      return "node(s) had untolerated taint";
    }
    case YKPRED_CODE_NODE_AFFINITY:
      if (reason & YKPRED_REASON_PREFILTER_NODE_NOT_ELIGIBLE) return "node not eligible";
      return "node(s) didn't match Pod's node affinity/selector";
    case YKPRED_CODE_NODE_PORTS: return "node(s) didn't have free ports for the requested pod ports";
    case YKPRED_CODE_NODE_RESOURCES_FIT: {
      std::string m;
      auto add = [&](const std::string& s) {
        if (!m.empty()) m += ", ";
        m += s;
      };
      if (reason & YKPRED_REASON_TOO_MANY_PODS) add("Too many pods");
      static const char* base[] = {"cpu", "memory", "ephemeral-storage"};
      for (int r = 0; r < h->enc.R; ++r)
        if (reason & (1u << (YKPRED_REASON_RESOURCE_SHIFT + r))) add(std::string("Insufficient ") + (r < 3 ? base[r] : h->enc.scalar_names[(size_t)r - 3].c_str()));
      if (m.empty()) m = "running \"NodeResourcesFit\" filter plugin: reading \"PreFilterNodeResourcesFit\" from cycleState: not found";
      return m;
    }
    case YKPRED_CODE_POD_TOPOLOGY_SPREAD:
      if (reason & YKPRED_REASON_MISSING_TOPOLOGY_LABEL) return "node(s) didn't match pod topology spread constraints (missing required label)";
      return "node(s) didn't match pod topology spread constraints";
    case YKPRED_CODE_INTER_POD_AFFINITY: return "node(s) didn't match pod affinity/anti-affinity rules";
    default: return "unschedulable";
  }
}

// ---- KWOK-style generator --------------------------------------------------------------
const char* kCpuPalette[] = {"0", "10m", "100m", "250m", "500m", "1", "2", "4", "8"};
const char* kMemPalette[] = {"0", "1M", "128Mi", "500M", "1Gi", "4Gi", "16Gi"};
const char* kKernel[] = {"0204", "0206", "0510", "0601"};
const char* kKernelThreshold[] = {"0100", "0205", "0300", "0600"};

std::string fmt(const char* f, int v) {
  char b[64];
  snprintf(b, sizeof b, f, v);
  return b;
}

PodTemplate draw_template(Rng& g, const ykhost_kwok_t& c, int n_nodes) {
  PodTemplate t;
  t.ns = "default";
  t.labels["app"] = fmt("app-%d", (int)g.below(8));
  Container ct;
  ct.name = "main";
  const char* cpu = kCpuPalette[g.below(9)];
  const char* mem = kMemPalette[g.below(7)];
  if (strcmp(cpu, "0") != 0) ct.requests["cpu"] = cpu;
  if (strcmp(mem, "0") != 0) ct.requests["memory"] = mem;
  t.containers.push_back(ct);
  if (c.tolerations) {
    if (g.chance(90, 100)) t.tolerations.push_back({"kwok.x-k8s.io/node", "Exists", "", "NoSchedule"});
    if (g.chance(10, 100)) t.tolerations.push_back({"dedicated", "Equal", fmt("team-%d", (int)g.below(8)), "NoSchedule"});
    if (g.chance(2, 100)) t.tolerations.push_back({"", "Exists", "", ""});
    if (g.chance(3, 100)) t.tolerations.push_back({"node.example.com/maintenance", "Exists", "", "NoExecute"});
  }
  if (c.spread && g.chance(10, 100)) {  // configs[4]: 10 % of asks carry one hard zone-spread constraint, maxSkew 1-3
    SpreadConstraint sc;
    sc.max_skew = 1 + (int32_t)g.below(3);
    sc.topology_key = "topology.kubernetes.io/zone";
    sc.when_unsatisfiable = "DoNotSchedule";
    sc.selector.present = true;
    sc.selector.match_labels["app"] = t.labels["app"];
    if (g.chance(1, 4)) sc.node_taints_policy = "Honor";
    if (g.chance(1, 8)) {
      sc.has_min_domains = true;
      sc.min_domains = 2 + (int32_t)g.below(20);
    }
    sc.raw_selector = sc.selector;
    t.spread.push_back(sc);
  }
  if (c.node_affinity) {
    auto zone = [&](int i) { return fmt("zone-%02d", i % 16); };
    auto label_expr = [&]() {
      Requirement r;
      switch (g.below(6)) {
        case 0: {  // zone In / NotIn a run of 1-3 zones
          r.key = "topology.kubernetes.io/zone";
          r.op = g.chance(1, 2) ? "In" : "NotIn";
          int z = (int)g.below(16), len = 1 + (int)g.below(3);
          for (int i = 0; i < len; ++i) r.values.push_back(zone(z + i));
          break;
        }
        case 1:
          r.key = "node.kubernetes.io/instance-type";
          r.op = g.chance(1, 2) ? "In" : "NotIn";
          r.values.push_back(fmt("type-%d", (int)g.below(8)));
          break;
        case 2:
          r.key = "kernel-version";
          r.op = g.chance(1, 2) ? "Gt" : "Lt";
          r.values.push_back(kKernelThreshold[g.below(4)]);
          break;
        case 3:
          r.key = "gpu";
          r.op = g.chance(1, 2) ? "Exists" : "DoesNotExist";
          break;
        case 4:
          r.key = "topology.kubernetes.io/zone";
          r.op = "Exists";
          break;
        default:
          r.key = "type";
          r.op = "In";
          r.values.push_back(g.chance(9, 10) ? "kwok" : "real");
          break;
      }
      return r;
    };
    uint32_t d = g.below(100);
    if (d < 60) {
      // none
    } else if (d < 80) {
      t.has_node_selector = true;
      switch (g.below(3)) {
        case 0: t.node_selector["topology.kubernetes.io/zone"] = zone((int)g.below(16)); break;
        case 1: t.node_selector["node.kubernetes.io/instance-type"] = fmt("type-%d", (int)g.below(8)); break;
        default: t.node_selector["type"] = "kwok"; break;
      }
    } else if (d < 95) {
      t.has_required = true;
      SelectorTerm term;
      int ne = 1 + (int)g.below(2);
      for (int i = 0; i < ne; ++i) term.exprs.push_back(label_expr());
      t.terms.push_back(term);
    } else {
      t.has_required = true;
      int nt = 2 + (int)g.below(3);
      for (int i = 0; i < nt; ++i) {
        SelectorTerm term;
        if (g.chance(1, 3)) {
          Requirement f;
          f.key = "metadata.name";
          f.op = g.chance(3, 4) ? "In" : "NotIn";
          f.values.push_back(fmt("kwok-node-%06d", (int)g.below((uint32_t)std::min(n_nodes, 32))));
          term.fields.push_back(f);
          if (g.chance(1, 2)) term.exprs.push_back(label_expr());
        } else {
          term.exprs.push_back(label_expr());
        }
        t.terms.push_back(term);
      }
    }
  }
  return t;
}

int generate_kwok(ykhost* h, const ykhost_kwok_t& c) {
  h->clear_state();
  h->uid_index = false;
  Rng g{0};
  const int N = c.num_nodes, P = c.num_pods;
  // ---- nodes: every node draws from its own stream, keyed by its GLOBAL index, so that any sharding of a cluster
  // (node_index_offset, num_nodes) reproduces exactly the nodes of the unsharded one
  for (int n = 0; n < N; ++n) {
    g = Rng{(c.seed ^ 0x4e4f444553ull) + (uint64_t)(c.node_index_offset + n) * 0xd1342543de82ef95ull};
    g.next();
    Node nd;
    nd.name = fmt("kwok-node-%06d", c.node_index_offset + n);
    bool big = g.chance(80, 100);
    int64_t cpu_milli = big ? 32000 : 16000;
    int64_t mem = big ? (256ll << 30) : 16000000000ll;
    nd.allocatable["cpu"] = big ? "32" : "16";
    nd.allocatable["memory"] = big ? "256Gi" : "16G";
    nd.allocatable["pods"] = "110";
    nd.labels["kubernetes.io/hostname"] = nd.name;
    nd.labels["topology.kubernetes.io/zone"] = fmt("zone-%02d", (int)g.below(16));
    nd.labels["node.kubernetes.io/instance-type"] = fmt("type-%d", (int)g.below(8));
    nd.labels["type"] = "kwok";
    nd.labels["kernel-version"] = kKernel[g.below(4)];
    if (c.tolerations) {
      nd.taints.push_back({"kwok.x-k8s.io/node", "fake", "NoSchedule"});
      if (g.chance(10, 100)) nd.taints.push_back({"dedicated", fmt("team-%d", (int)g.below(8)), "NoSchedule"});
      if (g.chance(2, 100)) nd.taints.push_back({"node.example.com/maintenance", "true", "NoExecute"});
      if (g.chance(5, 100)) nd.taints.push_back({"prefer", "soft", "PreferNoSchedule"});
    }
    nd.unschedulable = g.chance(1, 100);
    h->node_store.emplace_back();
    NodeInfo& ni = h->node_store.back();
    ni.set_node(nd);
    ni.index = n;
    h->nodes.push_back(&ni);
    h->node_ix[nd.name] = n;
    // ---- pods already on the node: 1 pod carrying the node's Requested, the rest best-effort
    int k = (int)g.below(111);
    uint32_t special = g.below(1000);
    int64_t used_cpu = (int64_t)((double)cpu_milli * (double)g.below(951) / 1000.0);
    int64_t used_mem = (int64_t)((double)mem * (double)g.below(951) / 1000.0);
    if (special < 5) {  // exactly full: no pod slot left, cpu free hits a palette value exactly
      k = 110;
      used_cpu = cpu_milli - 1000;
    } else if (special < 10) {  // over-committed: negative free
      used_cpu = cpu_milli + cpu_milli / 10;
      used_mem = mem + mem / 10;
      if (k == 0) k = 1;
    }
    if (k > 0) {
      PodTemplate a;
      a.ns = "default";
      a.labels["app"] = fmt("app-%d", (int)g.below(8));
      Container ct;
      ct.name = "main";
      if (used_cpu > 0) ct.requests["cpu"] = std::to_string(used_cpu) + "m";
      if (used_mem > 0) ct.requests["memory"] = std::to_string(used_mem);
      a.containers.push_back(ct);
      const PodTemplate* at = h->pool.intern(std::move(a));
      Pod p;
      p.uid = p.name = fmt("n%d-p0", c.node_index_offset + n);
      p.node_name = p.assigned_node = nd.name;
      p.tpl = at;
      h->pod_store.push_back(std::move(p));
      ni.add_pod(&h->pod_store.back());
      PodTemplate b;
      b.ns = "default";
      b.labels["app"] = fmt("app-%d", (int)g.below(8));
      Container cb;
      cb.name = "main";
      b.containers.push_back(cb);
      const PodTemplate* bt = h->pool.intern(std::move(b));
      for (int i = 1; i < k; ++i) {
        Pod q;
        q.uid = q.name = "n" + std::to_string(c.node_index_offset + n) + "-p" + std::to_string(i);
        q.node_name = q.assigned_node = nd.name;
        q.tpl = bt;
        h->pod_store.push_back(std::move(q));
        ni.add_pod(&h->pod_store.back());
      }
    }
  }
  // ---- pending asks (own stream: identical on every shard of a node-sharded cluster)
  g = Rng{c.seed ^ 0x504f4453ull};
  std::vector<const PodTemplate*> tpls;
  int T = c.num_templates;
  if (c.gang_size > 0) T = (P + c.gang_size - 1) / c.gang_size;
  const int NT = c.total_nodes > 0 ? c.total_nodes : N;  // asks are drawn against the WHOLE cluster: identical on every shard
  for (int i = 0; i < T; ++i) tpls.push_back(h->pool.intern(draw_template(g, c, NT)));
  for (int p = 0; p < P; ++p) {
    Pod pod;
    pod.uid = pod.name = fmt("pod-%07d", p);
    if (c.unique_requests) {
      PodTemplate t = draw_template(g, c, NT);
      t.containers[0].requests["cpu"] = std::to_string(p + 1) + "m";
      pod.tpl = h->pool.intern(std::move(t));
    } else if (T > 0) {
      pod.tpl = c.gang_size > 0 ? tpls[(size_t)(p / c.gang_size)] : tpls[(size_t)(p % T)];
    } else {
      pod.tpl = h->pool.intern(draw_template(g, c, NT));
    }
    uint32_t pin = g.below(10000);
    if (NT > 0 && pin < 10)
      pod.node_name = fmt("kwok-node-%06d", (int)g.below((uint32_t)NT));  // a GLOBAL node name (on another shard: unknown here)
    else if (pin == 10)
      pod.node_name = "no-such-node";
    pod.ask = true;
    pod.row = (int32_t)h->pending.size();
    h->pod_store.push_back(std::move(pod));
    h->pending.push_back(&h->pod_store.back());
  }
  h->dirty_all = true;
  return 0;
}

}  // namespace

// =====================================================================================================
#define YKHOST_LOCKED(h) std::lock_guard<std::recursive_mutex> host_lock((h)->mu)

extern "C" {

ykhost_t* ykhost_create(int32_t device, char* err, int32_t errlen) {
  auto* h = new ykhost();
  h->device = device;
  // create the engine eagerly with the minimal shape so that a missing GPU fails here, loudly; device < 0 makes a
  // mirror-only handle (cache bookkeeping, request vectors, snapshot dumps) on which every evaluation fails
  h->enc.R = 3;
  h->enc.KT = 1;
  h->enc.W = 1;
  if (const char* v = getenv("YKHOST_RESIDENT_MB")) h->resident_budget_bytes = (int64_t)atoll(v) << 20;
  if (device >= 0 && recreate_engine(h) != 0) {
    copy_out(h->err, err, errlen);
    delete h;
    return nullptr;
  }
  return h;
}
void ykhost_destroy(ykhost_t* h) {
  if (!h) return;
  if (h->eng) ykpred_destroy(h->eng);
  delete h;
}
const char* ykhost_last_error(const ykhost_t* h) { return h ? h->err.c_str() : ""; }

int32_t ykhost_set_plugins(ykhost_t* h, uint32_t rp, uint32_t ap, uint32_t rf, uint32_t af) {
  YKHOST_LOCKED(h);
  h->answers.pod = -1;
  h->resident.valid = false;
  h->res_pre = rp;
  h->alloc_pre = ap;
  h->res_filt = rf;
  h->alloc_filt = af;
  return 0;
}

int32_t ykhost_load_snapshot(ykhost_t* h, const char* json) {
  YKHOST_LOCKED(h);
  try {
    mj::ValuePtr root = mj::parse(json);
    h->clear_state();
    size_t anon = 0;
    if (const mj::Value* nodes = root->get_nn("nodes"))
      for (auto& nv : nodes->arr) {
        h->node_store.emplace_back();
        NodeInfo& ni = h->node_store.back();
        ni.set_node(read_node(*nv));
        ni.index = (int32_t)h->nodes.size();
        h->node_ix[ni.node.name] = ni.index;
        h->nodes.push_back(&ni);
        if (const mj::Value* pods = nv->get_nn("pods"))
          for (auto& pv : pods->arr) {
            int64_t reps = pv->int_or("replicas", 1);
            for (int64_t r = 0; r < reps; ++r) {
              Pod* p = add_pod_object(h, *pv, &anon);
              if (reps > 1) p->uid += "#" + std::to_string(r);
              p->node_name = p->assigned_node = ni.node.name;
              ni.add_pod(p);
              h->by_uid[p->uid] = p;
            }
          }
      }
    if (const mj::Value* pods = root->get_nn("pods"))
      for (auto& pv : pods->arr) {
        Pod* p = add_pod_object(h, *pv, &anon);  // an ask of the snapshot; a spec.nodeName it carries is only the NodeName filter's input
        p->ask = true;
        p->row = (int32_t)h->pending.size();
        h->pending.push_back(p);
        h->by_uid[p->uid] = p;
      }
    h->dirty_all = true;
    return 0;
  } catch (const std::exception& e) {
    return fail(h, e.what());
  }
}

// SchedulerCache.UpdateNode (scheduler_cache.go:148-187): add or replace the node object, keep its pods; a NEW node
// adopts the orphaned pods whose spec.nodeName is its name (:165-172). Returns the number of adopted pods.
static int update_node_text(ykhost* h, js::Range doc);
static Node parse_node_text(js::Range doc);
static int apply_node(ykhost* h, const Node& n);
int32_t ykhost_update_node(ykhost_t* h, const char* node_json) {
  YKHOST_LOCKED(h);
  try {
    return update_node_text(h, js::Range{node_json, node_json + strlen(node_json)});
  } catch (const std::exception& e) {
    return fail(h, e.what());
  }
}
// Many Node documents in one call (see ykhost_update_pods_batch). → documents applied, or -1 - (documents applied before the bad one).
int32_t ykhost_update_nodes_batch(ykhost_t* h, const char* text, int64_t len) {
  YKHOST_LOCKED(h);
  if (!text || len < 0) return fail(h, "bad argument");
  long applied = 0;
  try {
    // Documents are found and parsed first — the parse (90 % of a Node's cost) on every core — then go through the cache one by
    // one in the buffer's order; a document that does not parse stops the batch where the one-thread form would have stopped.
    std::vector<js::Range> docs;
    const long n = js::documents(text, text + len, [&](js::Range doc) { docs.push_back(doc); });
    const size_t D = docs.size();
    std::vector<Node> parsed(D);
    std::vector<std::string> errors(D);
    std::vector<char> bad(D, 0);
    {
      const int T = (int)std::max<size_t>(1, std::min<size_t>(host_threads(), D / 64));
      run_on_threads(T, (int)((D + 63) / 64), [&](int blk) {
        for (size_t j = (size_t)blk * 64; j < std::min(D, (size_t)blk * 64 + 64); ++j) {
          try {
            parsed[j] = parse_node_text(docs[j]);
          } catch (const std::exception& e) {
            bad[j] = 1;
            errors[j] = e.what();
          }
        }
      });
    }
    for (size_t i = 0; i < D; ++i) {
      if (bad[i]) return fail(h, std::string("document #") + std::to_string(applied) + ": " + errors[i], (int)(-1 - applied));
      apply_node(h, parsed[i]);
      ++applied;
    }
    if (n < 0) return fail(h, "malformed JSON document #" + std::to_string(-1 - n) + " in the batch", (int)(-1 - applied));
    return (int32_t)n;
  } catch (const std::exception& e) {
    return fail(h, std::string("document #") + std::to_string(applied) + ": " + e.what(), (int)(-1 - applied));
  }
}
// A real Node object is tens of kilobytes of status.images, conditions and addresses around the five fields the predicates
// read: the scanner cuts the document down to those before the tree parser sees it.
static Node parse_node_text(js::Range doc) {  // (no shared state: the batch form runs it on every core)
  std::string reduced;
  mj::ValuePtr v = js::reduce_node(doc, &reduced) ? mj::parse(reduced) : mj::parse(std::string(doc.b, doc.e));
  return read_node(*v);
}
static int update_node_text(ykhost* h, js::Range doc) { return apply_node(h, parse_node_text(doc)); }
static int apply_node(ykhost* h, const Node& n) {
  {
    int adopted = 0;
    h->label_index_valid = false;
    auto it = h->node_ix.find(n.name);
    if (it == h->node_ix.end()) {
      h->node_store.emplace_back();
      NodeInfo& ni = h->node_store.back();
      ni.set_node(n);
      ni.index = (int32_t)h->nodes.size();
      h->node_ix[n.name] = ni.index;
      h->nodes.push_back(&ni);
      ensure_uid_index(h);
      std::vector<Pod*> orphans;
      for (auto& kv : h->by_uid)
        if (kv.second->orphan && kv.second->node_name == n.name) orphans.push_back(kv.second);
      for (Pod* p : orphans)
        if (cache_update_pod(h, p, p, false, false)) ++adopted;
      h->dirty_all = true;  // the node axis grew: every table is re-uploaded
    } else {
      NodeInfo* ni = h->nodes[(size_t)it->second];
      ni->set_node(n);
      if (h->enc.node_known(*ni))
        touch_node(h, it->second);  // same dictionaries: only this node's row (and bitmap column) changes
      else
        h->dirty_all = true;  // a new taint or scalar resource extends the dictionaries
    }
    return adopted;
  }
}

// SchedulerCache.RemoveNode (:189-239). A pod that is still assumed was never bound: its assignment is reverted and it
// is a pending ask again (:205-219); every other pod of the node becomes an orphan that the node adopts again if it
// comes back. Returns the number of orphans; removing an unknown node is a no-op (:192-195).
int32_t ykhost_remove_node(ykhost_t* h, const char* name) {
  YKHOST_LOCKED(h);
  auto it = h->node_ix.find(name);
  if (it == h->node_ix.end()) return 0;
  h->label_index_valid = false;
  int idx = it->second;
  ensure_uid_index(h);
  int orphans = 0;
  for (const Pod* cp : h->nodes[(size_t)idx]->pods) {
    // every pod of the NodeInfo is handled (scheduler_cache.go:205-224), also one the uid index does not (or no longer)
    // point at: it becomes an orphan under its uid unless a newer version of the pod holds that uid
    auto pt = h->by_uid.find(cp->uid);
    if (pt != h->by_uid.end() && pt->second != cp) continue;  // a newer version owns the uid: the stale copy just leaves with the node
    Pod* p = const_cast<Pod*>(cp);                            // the mirror owns every Pod object (pod_store)
    if (pt == h->by_uid.end()) h->by_uid[p->uid] = p;
    const bool revert = p->assumed;
    p->assigned_node.clear();
    p->assumed = false;
    if (revert) {
      p->node_name.clear();  // (the whole table is re-uploaded below: node indices shift)
    } else {
      p->orphan = true;
      ++orphans;
    }
  }
  h->nodes[(size_t)idx]->pods.clear();
  h->nodes.erase(h->nodes.begin() + idx);
  h->node_ix.clear();
  for (size_t i = 0; i < h->nodes.size(); ++i) {
    h->nodes[i]->index = (int32_t)i;
    h->node_ix[h->nodes[i]->node.name] = (int)i;
  }
  h->dirty_all = true;
  return orphans;
}

// SchedulerCache.UpdatePod (:303-388). Returns 1, or 0 when the pod was stored as an orphan (its node is unknown).
// Ask table: an unassigned, not yet running pod is a pending ask and holds a row; a pod that arrives with a
// spec.nodeName of its own was bound by the cluster and holds none. A pod that already holds a row keeps it (in place)
// while it is neither running nor terminated, so that rows stay stable while binds are in flight.
static int apply_pod(ykhost* h, Pod* p, const std::string& phase, Pod* hint = nullptr);
static int update_pod_value(ykhost* h, const mj::Value& v) {
  ensure_uid_index(h);
  size_t anon = h->pod_store.size();
  Pod* p = add_pod_object(h, v, &anon);
  std::string phase;
  if (const mj::Value* st = v.get_nn("status")) phase = st->str_or("phase", "");
  return apply_pod(h, p, phase);
}
// One pod document as TEXT. The scanner (jsonscan.h) pulls out name / uid / nodeName / phase and the raw template text; a
// template that was seen before costs no parse at all. Anything the scanner does not vouch for takes the full parser.
static int update_pod_text(ykhost* h, js::Range doc) {
  js::PodScan sc;
  if (!js::scan_pod(doc, &sc) || sc.needs_full_parse) {
    h->ingest_full++;
    mj::ValuePtr v = mj::parse(std::string(doc.b, doc.e));
    return update_pod_value(h, *v);
  }
  const PodTemplate* tpl = nullptr;
  auto it = h->tpl_memo.find(sc.key);
  if (it != h->tpl_memo.end()) {
    tpl = it->second;
    h->ingest_fast++;
  } else {
    h->ingest_full++;
    mj::ValuePtr v = mj::parse(std::string(doc.b, doc.e));
    tpl = h->pool.intern(read_template(*v));
    if (h->tpl_memo.size() < 262144) h->tpl_memo.emplace(std::move(sc.key), tpl);  // (bounded: every-ask-its-own-template populations)
  }
  ensure_uid_index(h);
  Pod p;
  p.uid = sc.uid.str();
  p.name = sc.name.str();
  p.terminating = sc.terminating;
  p.node_name = sc.node_name.str();
  if (p.uid.empty()) p.uid = "anon-" + std::to_string(h->pod_store.size());
  p.tpl = tpl;
  return apply_pod(h, store_pod(h, std::move(p)), sc.phase.str());
}
// hint: what the uid index held for this uid when the batch was scanned (null: nothing, or not looked up). The batch may have moved on
// since — an earlier document of the same uid replaced or removed that version, and its slot may have been handed to another pod,
// even to `p` itself — so the hint counts only if it still IS a live pod with this uid other than p: pod slots are recycled empty
// (no template), and the cache never holds two live versions of one uid besides the one being applied. Anything else looks the uid up.
static int apply_pod(ykhost* h, Pod* p, const std::string& phase, Pod* hint) {
  const bool running = phase == "Running";                           // utils.IsPodRunning (utils.go:89-91)
  const bool terminated = phase == "Failed" || phase == "Succeeded";  // utils.IsPodTerminated (:93-95)
  Pod* old;
  if (hint && hint != p && hint->tpl && hint->uid == p->uid) {
    old = hint;
  } else {
    auto it = h->by_uid.find(p->uid);
    old = it == h->by_uid.end() ? nullptr : it->second;
  }
  const bool bound_by_cluster = !p->node_name.empty();
  const bool wants_row = !terminated && !running && (old && old->ask ? true : !bound_by_cluster);
  // A version that changes nothing the cache mirrors — an informer resync re-delivers every pod as it is — leaves the cache as it
  // is: the reference's updatePod would take the pod off its node and put it back (scheduler_cache.go:321-360), with the same
  // sums, the same maps and nothing for a predicate to see; here it would also mark the node and the ask row for re-evaluation.
  if (old && old != p && !terminated && old->tpl == p->tpl && old->node_name == p->node_name && old->name == p->name &&
      old->terminating == p->terminating && !old->assumed && !old->orphan && old->ask == wants_row &&
      (old->node_name.empty() || old->assigned_node == old->node_name)) {
    recycle_pod(h, p);
    return 1;
  }
  const bool ok = cache_update_pod(h, old, p, running, terminated);
  set_ask_row(h, old, wants_row ? p : nullptr);
  if (old && old != p) recycle_pod(h, old);  // replaced by the new version everywhere
  if (terminated) recycle_pod(h, p);          // dropped from every map (scheduler_cache.go:377-383)
  return ok ? 1 : 0;
}
int32_t ykhost_update_pod(ykhost_t* h, const char* pod_json) {
  YKHOST_LOCKED(h);
  try {
    return update_pod_text(h, js::Range{pod_json, pod_json + strlen(pod_json)});
  } catch (const std::exception& e) {
    return fail(h, e.what());
  }
}

// The same hooks for MANY objects in one call: `text` holds `len` bytes of JSON documents one after the other (newline- or
// comma-separated; a JSON array body works). What InitializeState (context.go:1411-1484) replays at start-up — every node,
// every pod — crosses cgo once, takes the handle's lock once, and the pods share the template memo. → documents applied;
// on a malformed or rejected document: -1 - (documents applied before it), reason in ykhost_last_error.
// ---- the batch in parallel -------------------------------------------------------------------------------------------
// What costs the time of a large batch is not the cache bookkeeping but reading the text: finding the documents, the five
// fields and the template key of each, and parsing the templates nobody has seen yet (1.2-1.4 us per document on one thread:
// 5.4 s for the 3.8 M documents of InitializeState at configs[2] size). That part has no shared state, so it runs on every
// host core: the buffer is cut at newlines into one piece per thread, each thread scans its documents into finished Pod
// objects (strings allocated there) and parses the templates the memo does not know into a list of its own. The cache
// bookkeeping — SchedulerCache.UpdatePod is order-dependent: a later version of a pod replaces an earlier one — then runs on
// the calling thread, document by document in the buffer's order, exactly as the one-thread form does.
// The cut assumes what encoding/json.Marshal produces: no raw newline inside a document. A piece whose first or last
// document does not parse (the cut fell inside a pretty-printed document) makes the whole batch take the one-thread path:
// nothing has been applied at that point.
namespace {
// templates the memo does not know yet, shared by the scanning threads: the first thread that meets a key parses the
// document, the others find the entry (a thread-local map in front keeps the shard locks off the common path)
struct SharedTemplates {
  static constexpr int kShards = 64;
  struct Shard {
    std::mutex mu;
    std::unordered_map<std::string, int> index;
    std::deque<PodTemplate> tpls;
    std::deque<std::string> keys;
  } shard[kShards];
};
struct ScannedPod {
  Pod pod;                 // uid / name / node_name / terminating; tpl when the memo had it
  std::string phase;
  int tpl_shard = -1, tpl_index = -1;  // else: the shared new template ...
  bool first = false;                  // ... and whether this document is the one it was parsed from
  js::Range full{};        // else: the document takes the full parser (in the ordered pass)
  Pod* cached = nullptr;   // the version of this uid the cache held when the batch was SCANNED (uid index current then): a hint for the
                           // ordered pass, which checks it before it trusts it (apply_pod) — the lookup is the pass's longest step
};
struct ScannedPiece {
  std::vector<ScannedPod> pods;
  bool ok = true;
  std::string error;
  // what decides whether the batch can take the bulk pass (update_pods_bulk)
  bool plain = true;                  // no document for the full parser, no terminated pod, no pod without a uid
  std::vector<int32_t> shared_tpl;    // indices of the pods that carry a template from the shared (new) list
};
void scan_piece(const ykhost* h, SharedTemplates* shared, const char* b, const char* e, ScannedPiece* out) {
  try {
    std::unordered_map<std::string, std::pair<int, int>> local;
    out->pods.reserve((size_t)(e - b) / 160 + 16);
    long n = js::documents(b, e, [&](js::Range doc) {
      if (memchr(doc.b, '\n', doc.size())) out->ok = false;  // (not the layout the cut relies on)
      out->pods.emplace_back();
      ScannedPod& sp = out->pods.back();
      js::PodScan sc;
      if (!js::scan_pod(doc, &sc) || sc.needs_full_parse) {
        sp.full = doc;
        return;
      }
      auto it = h->tpl_memo.find(sc.key);  // (read-only while the pieces are scanned)
      if (it != h->tpl_memo.end()) {
        sp.pod.tpl = it->second;
      } else {
        auto lt = local.find(sc.key);
        if (lt == local.end()) {
          SharedTemplates::Shard& sh = shared->shard[std::hash<std::string>{}(sc.key) % SharedTemplates::kShards];
          int idx = -1;
          {
            std::lock_guard<std::mutex> lock(sh.mu);
            auto st = sh.index.find(sc.key);
            if (st != sh.index.end()) idx = st->second;
          }
          if (idx < 0) {
            PodTemplate parsed = read_template(*mj::parse(std::string(doc.b, doc.e)));  // (outside the lock; a race parses twice, keeps one)
            TemplatePool::prepare(parsed);  // canonical key + request vector: the ordered part of interning is a map insert
            std::lock_guard<std::mutex> lock(sh.mu);
            auto st = sh.index.find(sc.key);
            if (st != sh.index.end()) {
              idx = st->second;
            } else {
              idx = (int)sh.tpls.size();
              sh.tpls.push_back(std::move(parsed));
              sh.keys.push_back(sc.key);
              sh.index.emplace(sc.key, idx);
              sp.first = true;
            }
          }
          lt = local.emplace(std::move(sc.key), std::make_pair((int)(&sh - shared->shard), idx)).first;
        }
        sp.tpl_shard = lt->second.first;
        sp.tpl_index = lt->second.second;
      }
      sp.pod.uid = sc.uid.str();
      sp.pod.name = sc.name.str();
      sp.pod.terminating = sc.terminating;
      sp.pod.node_name = sc.node_name.str();
      sp.phase = sc.phase.str();
      if (h->uid_index && !sp.pod.uid.empty()) sp.cached = h->by_uid.lookup(sp.pod.uid);
      if (sp.tpl_shard >= 0) out->shared_tpl.push_back((int32_t)out->pods.size() - 1);
      if (sp.pod.uid.empty() || sp.phase == "Failed" || sp.phase == "Succeeded") out->plain = false;
    });
    if (n < 0) out->ok = false;
    for (const ScannedPod& sp : out->pods)
      if (sp.full.present()) out->plain = false;
  } catch (const std::exception& ex) {
    out->ok = false;
    out->error = ex.what();
  }
}
// The cache pass of a BULK load on every core. SchedulerCache.UpdatePod is order-dependent only per uid (a later version of a
// pod replaces an earlier one) and per shared structure (a node's pod list, the ask table) — and at start-up, when
// Context.InitializeState replays the informer caches (context.go:1411-1484), neither bites: every pod is new. So when
//   * every table is going to be re-encoded anyway (dirty_all: no per-row / per-node patch bookkeeping to keep),
//   * no document needs the full parser, no pod is terminated or anonymous, no freed slot waits for reuse,
//   * and no uid of the batch is already cached or occurs twice in it (found out while the shards of the uid index are filled:
//     a clash rolls the batch back and the ordered pass takes it),
// the pass is split by what it writes: pod slots and ask rows by PIECE (document order = slot order = ask order), the uid
// index by uid SHARD, the nodes' pod lists and Requested sums by NODE GROUP — each structure is written by exactly one thread,
// in document order. The result is the ordered pass's, field by field (tests/test_host_ingest.py holds them equal; the same
// pass runs under ThreadSanitizer). Measured at configs[2] size on a 16-core quota: the ordered pass 3.8 s for 3.76 M
// documents (it was 93 % of the ingest), this one: DESIGN.md section 4.9.
// -> false: nothing was applied, the caller runs the ordered pass.
bool update_pods_bulk(ykhost* h, std::vector<ScannedPiece>& pieces, SharedTemplates* shared, long* applied_out) {
  const int T = (int)pieces.size();
  if (!h->dirty_all || !h->free_pods.empty()) return false;
  for (const ScannedPiece& pc : pieces)
    if (!pc.plain) return false;
  std::vector<size_t> offset((size_t)T + 1, 0);
  for (int t = 0; t < T; ++t) offset[(size_t)t + 1] = offset[(size_t)t] + pieces[(size_t)t].pods.size();
  const size_t M = offset[(size_t)T];
  if (M == 0) return false;
  ensure_uid_index(h);
  std::vector<std::vector<Pod>> block((size_t)T);  // the pods' final storage, one block per piece (published once no uid clashes)
  std::vector<Pod*> slot_base((size_t)T, nullptr);
  auto slot_of = [&](int t, int32_t local) { return slot_base[(size_t)t] + local; };
  const int G = std::max(1, std::min<int>(4 * T, (int)h->nodes.size()));  // node groups (contiguous index ranges)
  const size_t N = h->nodes.size();
  auto group_of = [&](int node) { return (int)((size_t)node * (size_t)G / std::max<size_t>(N, 1)); };
  std::vector<std::vector<std::vector<int32_t>>> by_shard((size_t)T, std::vector<std::vector<int32_t>>(UidIndex::kShards));
  std::vector<std::vector<std::vector<int32_t>>> by_group((size_t)T, std::vector<std::vector<int32_t>>((size_t)G));
  std::vector<std::vector<int32_t>> node_of((size_t)T);
  std::vector<size_t> asks_in((size_t)T, 0);
  auto run = [&](int n, auto&& f) {  // f(0..n-1) on T threads (the caller is one of them); an exception of a worker is rethrown here
    try {
      run_on_threads(T, n, f);
    } catch (const std::exception& ex) {
      throw std::runtime_error(std::string("bulk cache pass: ") + ex.what());
    }
  };
  // P1, by piece: where every pod goes (uid shard, node, node group), how many asks the piece holds
  run(T, [&](int t) {
    const std::vector<ScannedPod>& pods = pieces[(size_t)t].pods;
    block[(size_t)t].resize(pods.size());
    slot_base[(size_t)t] = block[(size_t)t].data();
    node_of[(size_t)t].resize(pods.size());
    for (auto& v : by_shard[(size_t)t]) v.reserve(pods.size() / UidIndex::kShards + 8);
    size_t asks = 0;
    for (size_t i = 0; i < pods.size(); ++i) {
      const ScannedPod& sp = pods[i];
      by_shard[(size_t)t][UidIndex::shard_of(sp.pod.uid)].push_back((int32_t)i);
      int node = -1;  // -1: no spec.nodeName, -2: names a node the cache does not hold (orphan)
      if (!sp.pod.node_name.empty()) {
        auto nt = h->node_ix.find(sp.pod.node_name);
        node = nt == h->node_ix.end() ? -2 : nt->second;
      }
      node_of[(size_t)t][i] = node;
      if (node >= 0) by_group[(size_t)t][(size_t)group_of(node)].push_back((int32_t)i);
      if (node == -1 && sp.phase != "Running") ++asks;
    }
    asks_in[(size_t)t] = asks;
  });
  // P2, by uid shard: the index entries of the new slots; a uid that is already there ends the bulk pass
  std::atomic<bool> clash{false};
  run(UidIndex::kShards, [&](int s) {
    UidIndex::Map& m = h->by_uid.shard[s];
    size_t n = 0;
    for (int t = 0; t < T; ++t) n += by_shard[(size_t)t][(size_t)s].size();
    m.reserve(m.size() + n);
    for (int t = 0; t < T; ++t)
      for (int32_t i : by_shard[(size_t)t][(size_t)s])
        if (!m.emplace(pieces[(size_t)t].pods[(size_t)i].pod.uid, slot_of(t, i)).second) clash.store(true, std::memory_order_relaxed);
  });
  if (clash.load()) {
    run(UidIndex::kShards, [&](int s) {
      UidIndex::Map& m = h->by_uid.shard[s];
      for (int t = 0; t < T; ++t)
        for (int32_t i : by_shard[(size_t)t][(size_t)s]) {
          auto it = m.find(pieces[(size_t)t].pods[(size_t)i].pod.uid);
          if (it != m.end() && it->second == slot_of(t, i)) m.erase(it);
        }
    });
    return false;
  }
  for (auto& b : block)
    if (!b.empty()) h->pod_blocks.push_back(std::move(b));  // (a moved vector keeps its buffer: the slots stay where they are)
  // the new templates, interned in the order the ordered pass meets them
  std::vector<std::vector<const PodTemplate*>> interned(SharedTemplates::kShards);
  for (int k = 0; k < SharedTemplates::kShards; ++k) interned[(size_t)k].assign(shared->shard[k].tpls.size(), nullptr);
  int64_t new_templates = 0;
  for (int t = 0; t < T; ++t)
    for (int32_t i : pieces[(size_t)t].shared_tpl) {
      const ScannedPod& sp = pieces[(size_t)t].pods[(size_t)i];
      const PodTemplate*& slot = interned[(size_t)sp.tpl_shard][(size_t)sp.tpl_index];
      if (slot) continue;
      SharedTemplates::Shard& sh = shared->shard[sp.tpl_shard];
      slot = h->pool.intern_prepared(std::move(sh.tpls[(size_t)sp.tpl_index]));
      if (h->tpl_memo.size() < 262144) h->tpl_memo.emplace(std::move(sh.keys[(size_t)sp.tpl_index]), slot);
      ++new_templates;
    }
  h->ingest_full += new_templates;
  h->ingest_fast += (int64_t)M - new_templates;
  // P3, by piece: the pods move into their slots; asks take their rows (document order)
  const size_t pending_before = h->pending.size();
  std::vector<size_t> ask_base((size_t)T + 1, pending_before);
  for (int t = 0; t < T; ++t) ask_base[(size_t)t + 1] = ask_base[(size_t)t] + asks_in[(size_t)t];
  std::atomic<bool> any_on_node{false};
  // From here on the mirror is being changed. If P3 / P4 run out of memory half-way, what they did is made consistent again
  // before the exception leaves: the ask table keeps no empty row (the rows that were filled close up, in document
  // order) and every pod that carries a node is on that node's list exactly once. The caller re-encodes everything anyway.
  auto repair = [&]() {
    size_t keep = pending_before;
    for (size_t i = pending_before; i < h->pending.size(); ++i)
      if (Pod* p = h->pending[i]) {
        p->row = (int32_t)keep;
        h->pending[keep++] = p;
      }
    h->pending.resize(keep);
    for (int t = 0; t < T; ++t)
      for (size_t i = 0; i < offset[(size_t)t + 1] - offset[(size_t)t]; ++i) {
        const Pod* q = slot_of(t, (int32_t)i);  // (a slot P3 did not reach is an empty Pod: no node)
        if (q->assigned_node.empty()) continue;
        auto nt = h->node_ix.find(q->assigned_node);
        if (nt == h->node_ix.end()) continue;
        NodeInfo& ni = *h->nodes[(size_t)nt->second];
        if (std::find(ni.pods.begin(), ni.pods.end(), q) == ni.pods.end()) ni.add_pod(q);
      }
    h->resident.valid = false;
  };
  try {
  h->pending.resize(ask_base[(size_t)T]);
  run(T, [&](int t) {
    std::vector<ScannedPod>& pods = pieces[(size_t)t].pods;
    size_t row = ask_base[(size_t)t];
    bool on_node = false;
    for (size_t i = 0; i < pods.size(); ++i) {
      ScannedPod& sp = pods[i];
      Pod* p = slot_of(t, (int32_t)i);
      if (sp.tpl_shard >= 0) sp.pod.tpl = interned[(size_t)sp.tpl_shard][(size_t)sp.tpl_index];
      *p = std::move(sp.pod);
      const int node = node_of[(size_t)t][i];
      if (node >= 0) {
        p->assigned_node = p->node_name;
        on_node = true;
      } else if (node == -2) {
        p->orphan = true;
      } else if (sp.phase != "Running") {
        p->ask = true;
        p->row = (int32_t)row;
        h->pending[row++] = p;
      }
    }
    if (on_node) any_on_node.store(true, std::memory_order_relaxed);
    std::vector<ScannedPod>().swap(pods);  // (freed here, on the piece's thread)
  });
  // P4, by node group: pod lists and Requested sums
  run(G, [&](int g) {
    for (int t = 0; t < T; ++t)
      for (int32_t i : by_group[(size_t)t][(size_t)g]) h->nodes[(size_t)node_of[(size_t)t][(size_t)i]]->add_pod(slot_of(t, i));
  });
  } catch (...) {
    repair();
    throw;
  }
  if (any_on_node.load()) h->resident.valid = false;  // (touch_node: node columns changed under a resident answer)
  *applied_out = (long)M;
  return true;
}

// → documents applied, -1 when the batch has to take the one-thread path (nothing applied), or the error code of a rejected
// document (-1 - applied, like the one-thread form)
long update_pods_parallel(ykhost* h, const char* text, int64_t len, bool* fallback) {
  *fallback = true;
  // YKHOST_INGEST_THREADS: a container with a CPU quota still reports every core of its host (hardware_concurrency), and
  // scanning threads that share two real cores are slower than one; 1 = the one-thread path
  const unsigned hw = host_threads();
  const int64_t min_piece = 1 << 16;
  const int T = (int)std::min<int64_t>(hw, len / min_piece);
  if (T < 2) return -1;
  std::vector<const char*> cut((size_t)T + 1);
  cut[0] = text;
  cut[(size_t)T] = text + len;
  for (int t = 1; t < T; ++t) {
    const char* p = text + len * t / T;
    p = (const char*)memchr(p, '\n', (size_t)(text + len - p));
    cut[(size_t)t] = p ? p + 1 : text + len;
  }
  if (!h->dirty_all) ensure_uid_index(h);  // (steady state: the scanning threads look the uids up; a start-up replay fills the index in its bulk pass)
  const auto t_begin = std::chrono::steady_clock::now();
  std::vector<ScannedPiece> pieces((size_t)T);
  auto shared = std::make_unique<SharedTemplates>();
  try {
    run_on_threads(T, T, [&](int t) { scan_piece(h, shared.get(), cut[(size_t)t], cut[(size_t)t + 1], &pieces[(size_t)t]); });
  } catch (const std::exception&) {
    return -1;  // (a piece that threw — bad_alloc — has applied nothing: the one-thread path takes the batch)
  }
  for (const ScannedPiece& pc : pieces)
    if (!pc.ok) return -1;
  *fallback = false;
  const auto t_scanned = std::chrono::steady_clock::now();
  auto account = [&]() {
    h->ingest_threads = T;
    h->ingest_parallel_batches++;
    h->ingest_scan_us += std::chrono::duration_cast<std::chrono::microseconds>(t_scanned - t_begin).count();
    h->ingest_apply_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_scanned).count();
  };
  {
    long bulk_applied = 0;
    if (update_pods_bulk(h, pieces, shared.get(), &bulk_applied)) {
      h->ingest_bulk_batches++;
      account();
      return bulk_applied;
    }
  }
  // ---- the ordered pass: a new template is interned when the first pod that carries it is reached — the order the one-thread
  // form interns them in — then the pod goes through the cache
  std::vector<std::vector<const PodTemplate*>> interned(SharedTemplates::kShards);
  for (int k = 0; k < SharedTemplates::kShards; ++k) interned[(size_t)k].assign(shared->shard[k].tpls.size(), nullptr);
  long applied = 0;
  ensure_uid_index(h);
  try {
    for (ScannedPiece& pc : pieces) {
      for (ScannedPod& sp : pc.pods) {
        if (sp.full.present()) {
          h->ingest_full++;
          mj::ValuePtr v = mj::parse(std::string(sp.full.b, sp.full.e));
          update_pod_value(h, *v);
        } else {
          if (sp.tpl_shard >= 0) {
            const PodTemplate*& slot = interned[(size_t)sp.tpl_shard][(size_t)sp.tpl_index];
            if (!slot) {
              SharedTemplates::Shard& sh = shared->shard[sp.tpl_shard];
              slot = h->pool.intern_prepared(std::move(sh.tpls[(size_t)sp.tpl_index]));
              if (h->tpl_memo.size() < 262144) h->tpl_memo.emplace(std::move(sh.keys[(size_t)sp.tpl_index]), slot);
              h->ingest_full++;  // (one full parse per new template, as on one thread)
            } else {
              h->ingest_fast++;
            }
            sp.pod.tpl = slot;
          } else {
            h->ingest_fast++;
          }
          if (sp.pod.uid.empty()) sp.pod.uid = "anon-" + std::to_string(h->pod_store.size());
          apply_pod(h, store_pod(h, std::move(sp.pod)), sp.phase, sp.cached);
        }
        ++applied;
      }
    }
  } catch (const std::exception& ex) {
    fail(h, std::string("document #") + std::to_string(applied) + ": " + ex.what());
    account();
    return -1 - applied;
  }
  account();
  return applied;
}
}  // namespace

int32_t ykhost_update_pods_batch(ykhost_t* h, const char* text, int64_t len) {
  YKHOST_LOCKED(h);
  if (!text || len < 0) return fail(h, "bad argument");
  try {
    bool fallback = true;
    const long n = update_pods_parallel(h, text, len, &fallback);
    if (!fallback) return (int32_t)n;
  } catch (const std::exception& e) {
    // (out of memory in the middle of the bulk pass: the mirror may hold a part of the batch — everything is re-encoded anyway,
    // and the caller is told)
    h->dirty_all = true;
    h->uid_index = false;
    return fail(h, std::string("pod batch: ") + e.what());
  }
  long applied = 0;
  try {
    long n = js::documents(text, text + len, [&](js::Range doc) {
      update_pod_text(h, doc);
      ++applied;
    });
    if (n < 0) return fail(h, "malformed JSON document #" + std::to_string(-1 - n) + " in the batch", (int)(-1 - applied));
    return (int32_t)n;
  } catch (const std::exception& e) {
    return fail(h, std::string("document #") + std::to_string(applied) + ": " + e.what(), (int)(-1 - applied));
  }
}

// ---- gang scheduling: task groups → placeholder asks ----------------------------------------------------------
// GetTaskGroupsFromAnnotation + validateTaskGroupResources (pkg/cache/utils.go:33-121). Returns "" when the annotation
// value is acceptable, else the reason it is rejected.
static std::string validate_task_groups(const mj::Value& groups) {
  if (!groups.is_arr()) return "task-groups annotation is not a JSON array";
  const int64_t kMax = std::numeric_limits<int64_t>::max();
  std::map<std::string, int64_t> totals;
  for (auto& g : groups.arr) {
    if (!g->is_obj()) return "task group is not an object";
    const mj::Value* name = g->get_nn("name");
    if (name && !name->is_str()) return "taskGroup name is not a string";
    if (!name || name->s.empty()) return "can't get taskGroup Name from pod annotation";
    const mj::Value* mm = g->get_nn("minMember");
    if (mm && !mm->is_num()) return "taskGroup minMember is not a number";
    const mj::Value* mr = g->get_nn("minResource");
    if (!mr) return "can't get taskGroup MinResource from pod annotation";
    if (!mr->is_obj()) return "taskGroup minResource is not an object";
    const int64_t members = mm ? std::strtoll(mm->s.c_str(), nullptr, 10) : 0;
    if (mm && (mm->s.find_first_of(".eE") != std::string::npos || members > std::numeric_limits<int32_t>::max() ||
               members < std::numeric_limits<int32_t>::min()))
      return "taskGroup minMember is not an int32";
    if (members == 0) return "can't get taskGroup MinMember from pod annotation";
    if (members < 0) return "minMember cannot be negative";
    // validateTaskGroupResources (:79-121): the placeholder ask of the group, folded into the cross-group totals under
    // canonical keys ("cpu" → "vcore" in milli-units; the implicit "pods" count claims its key)
    std::set<std::string> seen{"pods"};
    if (totals["pods"] > kMax - members) return "aggregate placeholder request for \"pods\" overflows int64 across taskGroups";
    totals["pods"] += members;
    for (auto& kv : mr->obj) {
      if (!kv.second->is_str() && !kv.second->is_num()) return "minResource " + kv.first + " is not a quantity";
      Quantity q = parse_quantity(kv.second->s);
      if (!q.ok) return "minResource " + kv.first + " is not a quantity";
      if (q.mant < 0) return "minResource \"" + kv.first + "\" in taskGroup \"" + name->s + "\" cannot be negative";
      const bool cpu = kv.first == "cpu";
      const std::string canonical = cpu ? "vcore" : kv.first;
      if (!seen.insert(canonical).second)
        return "minResource \"" + kv.first + "\" in taskGroup \"" + name->s + "\" collides with another resource under canonical key \"" + canonical + "\"";
      if (quantity_cmp_int64(q, kMax / (members * (cpu ? 1000 : 1))) > 0)
        return "minResource \"" + kv.first + "\" in taskGroup \"" + name->s + "\" overflows int64 when scaled by minMember";
      const int64_t contribution = members * scaled_value(q, cpu ? -3 : 0);
      if (totals[canonical] > kMax - contribution) return "aggregate placeholder request for \"" + canonical + "\" overflows int64 across taskGroups";
      totals[canonical] += contribution;
    }
  }
  return "";
}

int32_t ykhost_validate_task_groups(ykhost_t* h, const char* task_groups_json) {
  YKHOST_LOCKED(h);
  try {
    mj::ValuePtr groups = mj::parse(task_groups_json);
    std::string why = validate_task_groups(*groups);
    if (!why.empty()) return fail(h, why);
    return (int32_t)groups->arr.size();
  } catch (const std::exception& e) {
    return fail(h, e.what());
  }
}

// PlaceholderManager.createAppPlaceholders + newPlaceholder (placeholder_manager.go:72-99, placeholder.go:40-157): every
// task group yields minMember placeholder pods with IDENTICAL scheduling-relevant fields — labels (the group's + app id +
// queue), one container whose requests are the group's minResource, nodeSelector, tolerations, affinity and
// topologySpreadConstraints — i.e. one pod class per group. They enter the cache like any pending pod.
int32_t ykhost_add_task_groups(ykhost_t* h, const char* app_json, const char* task_groups_json) {
  YKHOST_LOCKED(h);
  try {
    mj::ValuePtr app = mj::parse(app_json);
    mj::ValuePtr groups = mj::parse(task_groups_json);
    std::string why = validate_task_groups(*groups);
    if (!why.empty()) return fail(h, why);
    const std::string app_id = app->str_or("applicationId", ""), queue = app->str_or("queue", ""), ns = app->str_or("namespace", "");
    auto mk = [](mj::Kind k) {
      auto v = std::make_shared<mj::Value>();
      v->kind = k;
      return v;
    };
    auto str = [&](const std::string& s) {
      auto v = mk(mj::Kind::String);
      v->s = s;
      return v;
    };
    auto child = [](const mj::Value& o, const char* key) -> mj::ValuePtr {
      for (auto& kv : o.obj)
        if (kv.first == key && !kv.second->is_null()) return kv.second;
      return nullptr;
    };
    int created = 0;
    for (auto& g : groups->arr) {
      const std::string tg = g->str_or("name", "");
      auto labels = mk(mj::Kind::Object);
      if (auto l = child(*g, "labels")) labels->obj = l->obj;
      auto set_label = [&](const std::string& k, const std::string& v) {  // utils.MergeMaps: the app's entries win
        for (auto& kv : labels->obj)
          if (kv.first == k) {
            kv.second = str(v);
            return;
          }
        labels->obj.emplace_back(k, str(v));
      };
      set_label("yunikorn.apache.org/app-id", app_id);
      set_label("yunikorn.apache.org/queue", queue);
      auto requests = mk(mj::Kind::Object);  // GetPlaceholderResourceRequests (gang_utils.go:83-92): empty names dropped
      for (auto& kv : g->get_nn("minResource")->obj)
        if (!kv.first.empty()) requests->obj.emplace_back(kv.first, kv.second);
      auto resources = mk(mj::Kind::Object);
      resources->obj.emplace_back("requests", requests);
      resources->obj.emplace_back("limits", requests);
      auto container = mk(mj::Kind::Object);
      container->obj.emplace_back("name", str("pause"));
      container->obj.emplace_back("resources", resources);
      auto containers = mk(mj::Kind::Array);
      containers->arr.push_back(container);
      auto spec = mk(mj::Kind::Object);
      spec->obj.emplace_back("containers", containers);
      for (const char* field : {"nodeSelector", "tolerations", "affinity", "topologySpreadConstraints"})
        if (auto v = child(*g, field)) spec->obj.emplace_back(field, v);
      const int64_t members = g->int_or("minMember", 0);
      for (int64_t i = 0; i < members; ++i) {
        // GeneratePlaceholderName (gang_utils.go:61-67): "tg-%.28s-%.20s-<10 base-36 chars>"; the nonce is drawn from a
        // per-handle counter here so that runs are reproducible
        uint64_t z = (h->placeholder_serial++ + 0x9e3779b97f4a7c15ull) * 0xbf58476d1ce4e5b9ull;
        z ^= z >> 29;
        std::string nonce;
        for (int k = 0; k < 10; ++k) {
          nonce.push_back("abcdefghijklmnopqrstuvwxyz0123456789"[z % 36]);
          z = z / 36 + (z << 7);
        }
        const std::string pname = "tg-" + app_id.substr(0, 28) + "-" + tg.substr(0, 20) + "-" + nonce;
        auto meta = mk(mj::Kind::Object);
        meta->obj.emplace_back("name", str(pname));
        meta->obj.emplace_back("uid", str(pname));
        meta->obj.emplace_back("namespace", str(ns));
        meta->obj.emplace_back("labels", labels);
        auto pod = mk(mj::Kind::Object);
        pod->obj.emplace_back("metadata", meta);
        pod->obj.emplace_back("spec", spec);
        update_pod_value(h, *pod);
        ++created;
      }
    }
    return created;
  } catch (const std::exception& e) {
    return fail(h, e.what());
  }
}

// SchedulerCache.RemovePod (:390-417): unknown pods are ignored.
int32_t ykhost_remove_pod(ykhost_t* h, const char* uid) {
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  auto it = h->by_uid.find(uid);
  if (it == h->by_uid.end()) return 0;
  Pod* p = it->second;
  detach_from_node(h, p);
  set_ask_row(h, p, nullptr);
  h->by_uid.erase(it);
  recycle_pod(h, p);
  return 1;
}

// Context.AssumePod → SchedulerCache.AssumePod (context.go:828-885, scheduler_cache.go:443-461): the cached pod gets
// spec.nodeName = node and goes through updatePod, i.e. it is accounted on the node (moving there from a node it was
// assumed on before), and is marked assumed. Unknown pod or node: nothing happens (context.go:831,835).
int32_t ykhost_assume_pod(ykhost_t* h, const char* uid, const char* node_name) {
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  auto it = h->by_uid.find(uid);
  auto nt = h->node_ix.find(node_name);
  if (it == h->by_uid.end()) return fail(h, "pod not found");
  if (nt == h->node_ix.end()) return fail(h, "node not found");
  Pod* p = it->second;
  // The ask keeps its row in the ask table so that pod indices (bitmap rows) stay stable during a scheduling cycle; only
  // node rows and bitmap columns change. The core does not ask about an ask it has allocated, so the row is left
  // evaluating the spec without the nodeName pin until something else rebuilds the ask table.
  detach_from_node(h, p);
  p->node_name = node_name;
  cache_update_pod(h, p, p, false, false);
  p->assumed = true;
  return 0;
}

// Context.ForgetPod → SchedulerCache.ForgetPod (context.go:887-898, scheduler_cache.go:463-484): updatePod of the
// CACHED pod — which still carries the node name it was assumed on, so it stays accounted on that node — and the
// assumed mark is dropped. The ask's answers are from now on those of a pod with spec.nodeName set (NodeName filter).
int32_t ykhost_forget_pod(ykhost_t* h, const char* uid) {
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  auto it = h->by_uid.find(uid);
  if (it == h->by_uid.end()) return 0;  // "unable to forget pod: not found in cache" (context.go:897)
  Pod* p = it->second;
  cache_update_pod(h, p, p, false, false);
  if (p->assumed && p->ask) mark_row(h, p->row);  // the row now follows the cached pod: pinned to its node
  p->assumed = false;
  p->remote_node = -1;
  return 1;
}

// Cache introspection for tests: flags bit0 in podsMap, bit1 assigned (accounted on a node), bit2 assumed, bit3 orphan,
// bit4 holds an ask row. node_out receives spec.nodeName of the cached pod.
int32_t ykhost_pod_state(ykhost_t* h, const char* uid, char* node_out, int32_t node_len) {
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  copy_out("", node_out, node_len);
  auto it = h->by_uid.find(uid);
  if (it == h->by_uid.end()) return 0;
  const Pod* p = it->second;
  copy_out(p->node_name, node_out, node_len);
  return 1 | (p->assigned_node.empty() ? 0 : 2) | (p->assumed ? 4 : 0) | (p->orphan ? 8 : 0) | (p->ask ? 16 : 0);
}
// len(NodeInfo.Pods) of a cached node, -1 when the node is not in the cache.
int32_t ykhost_node_pod_count(ykhost_t* h, const char* name) {
  YKHOST_LOCKED(h);
  auto it = h->node_ix.find(name);
  return it == h->node_ix.end() ? -1 : (int32_t)h->nodes[(size_t)it->second]->pods.size();
}

int32_t ykhost_generate_kwok(ykhost_t* h, const ykhost_kwok_t* cfg) {
  YKHOST_LOCKED(h);
  if (!cfg || cfg->num_nodes < 0 || cfg->num_pods < 0) return fail(h, "bad kwok config");
  return generate_kwok(h, *cfg);
}

int32_t ykhost_num_nodes(const ykhost_t* h) { YKHOST_LOCKED(h); return (int32_t)h->nodes.size(); }
int32_t ykhost_num_pods(const ykhost_t* h) { YKHOST_LOCKED(h); return (int32_t)h->pending.size(); }
int32_t ykhost_pod_index(const ykhost_t* ch, const char* uid) {
  ykhost_t* h = const_cast<ykhost_t*>(ch);
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  auto it = h->by_uid.find(uid);
  if (it != h->by_uid.end() && it->second->ask) return it->second->row;
  // asks of a snapshot may share a uid with a pod on a node (or carry none that is unique): fall back to the table
  for (size_t i = 0; i < h->pending.size(); ++i)
    if (h->pending[i]->uid == uid) return (int32_t)i;
  return -1;
}
int32_t ykhost_node_index(const ykhost_t* h, const char* name) {
  YKHOST_LOCKED(h);
  auto it = h->node_ix.find(name);
  return it == h->node_ix.end() ? -1 : it->second;
}

int32_t ykhost_set_dump_compact(ykhost_t* h, int32_t on) {
  YKHOST_LOCKED(h);
  h->dump_compact = on != 0;
  return 0;
}

int64_t ykhost_dump_snapshot(ykhost_t* h, const int32_t* pods, int32_t np, const int32_t* nodes, int32_t nn, char* out, int64_t len) {
  YKHOST_LOCKED(h);
  std::string o = "{\"nodes\":[";
  int count_n = nodes ? nn : (int)h->nodes.size();
  for (int i = 0; i < count_n; ++i) {
    int idx = nodes ? nodes[i] : i;
    if (idx < 0 || idx >= (int)h->nodes.size()) return fail(h, "dump: node index out of range");
    if (i) o.push_back(',');
    node_json(*h->nodes[(size_t)idx], o, h->dump_compact);
  }
  o += "],\"pods\":[";
  int count_p = pods ? np : (int)h->pending.size();
  for (int i = 0; i < count_p; ++i) {
    int idx = pods ? pods[i] : i;
    if (idx < 0 || idx >= (int)h->pending.size()) return fail(h, "dump: pod index out of range");
    if (h->pending[(size_t)idx]->assumed) continue;  // bound asks are listed under their node
    if (o.back() != '[') o.push_back(',');
    pod_json(*h->pending[(size_t)idx], o);
  }
  o += "]}";
  if (out && len > 0) copy_out(o, out, len);
  return (int64_t)o.size() + 1;
}

// The mirror's objects as the documents the cache hooks would deliver, newline-separated: kind 0 = Node objects, 1 = the pods
// that sit on nodes (spec.nodeName set, status.phase Running), 2 = the pending asks. What bench.py feeds back through
// ykhost_update_nodes_batch / ykhost_update_pods_batch to time the JSON ingest. Returns the required length (incl. NUL).
int64_t ykhost_dump_documents(ykhost_t* h, int32_t kind, char* out, int64_t len) {
  YKHOST_LOCKED(h);
  std::string o;
  if (kind == 0) {
    NodeInfo bare;
    for (const NodeInfo* ni : h->nodes) {
      bare.node = ni->node;
      node_json(bare, o, false);
      o.push_back('\n');
    }
  } else if (kind == 1) {
    for (const NodeInfo* ni : h->nodes)
      for (const Pod* p : ni->pods) {
        pod_json(*p, o);
        o.pop_back();
        o += ",\"status\":{\"phase\":\"Running\"}}\n";
      }
  } else if (kind == 2) {
    for (const Pod* p : h->pending) {
      if (p->assumed) continue;
      pod_json(*p, o);
      o.pop_back();
      o += ",\"status\":{\"phase\":\"Pending\"}}\n";
    }
  } else {
    return fail(h, "dump_documents: kind must be 0 (nodes), 1 (pods on nodes) or 2 (pending asks)");
  }
  if (out && len > 0) {
    const size_t n = std::min((size_t)(len - 1), o.size());
    memcpy(out, o.data(), n);
    out[n] = 0;
  }
  return (int64_t)o.size() + 1;
}
// out[0] = pod documents that reused a known template without a parse, out[1] = pod documents that took the full parser
int32_t ykhost_ingest_stats(ykhost_t* h, int64_t* out2) {
  YKHOST_LOCKED(h);
  out2[0] = h->ingest_fast;
  out2[1] = h->ingest_full;
  return 0;
}
int32_t ykhost_ingest_timing(ykhost_t* h, int64_t* out5) {
  int64_t* out4 = out5;
  YKHOST_LOCKED(h);
  out4[0] = h->ingest_threads;
  out4[1] = h->ingest_scan_us;
  out4[2] = h->ingest_apply_us;
  out4[3] = h->ingest_parallel_batches;
  out4[4] = h->ingest_bulk_batches;
  return 0;
}

// The encoded tables as JSON (64-bit masks as hex strings) — lets the encoder be checked without a device: tests
// evaluate the tables with a straightforward table-driven checker and compare with the per-pair object-model results.
int64_t ykhost_encoded_tables_json(ykhost_t* h, char* out, int64_t len) {
  YKHOST_LOCKED(h);
  EncodedTables T;
  auto t0 = std::chrono::steady_clock::now();
  int rc = encode_tables(h, &T);
  h->last_encode_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t0).count();
  h->dirty_all = true;  // spec ids were re-assigned: the device tables (if any) are re-uploaded at the next sync
  if (rc) return rc;
  std::string o = "{";
  auto ints = [&](const char* name, auto begin, size_t n) {
    o += std::string("\"") + name + "\":[";
    for (size_t i = 0; i < n; ++i) {
      if (i) o.push_back(',');
      o += std::to_string(begin[i]);
    }
    o += "],";
  };
  auto masks = [&](const char* name, const uint64_t* v, size_t n) {
    o += std::string("\"") + name + "\":[";
    char buf[24];
    for (size_t i = 0; i < n; ++i) {
      if (i) o.push_back(',');
      snprintf(buf, sizeof buf, "\"%llx\"", (unsigned long long)v[i]);
      o += buf;
    }
    o += "],";
  };
  const size_t N = h->nodes.size(), S = h->spec_templates.size(), P = h->pending.size();
  const size_t R = (size_t)h->enc.R, KT = (size_t)h->enc.KT, W = (size_t)h->enc.W, KD = (size_t)h->enc.KD, KS = (size_t)h->enc.KS,
               KP = (size_t)h->enc.KP;
  o += "\"R\":" + std::to_string(R) + ",\"KT\":" + std::to_string(KT) + ",\"W\":" + std::to_string(W) + ",\"KD\":" + std::to_string(KD) +
       ",\"KS\":" + std::to_string(KS) + ",\"KP\":" + std::to_string(KP) + ",\"N\":" + std::to_string(N) + ",\"S\":" + std::to_string(S) +
       ",\"P\":" + std::to_string(P) + ",";
  ints("allocatable", T.alloc.data(), N * R);
  ints("requested", T.req.data(), N * R);
  ints("allowed_pods", T.allowed.data(), N);
  ints("pod_count", T.count.data(), N);
  ints("node_flags", T.flags.data(), N);
  masks("taint_bits", T.taints.data(), N * KT);
  masks("label_bits", T.labels.data(), N * W);
  masks("port_bits", T.ports.data(), N * KP);
  ints("domain_id", T.domain.data(), N * KD);
  ints("selector_count", T.selcount.data(), N * KS);
  ints("requests", T.sreq.data(), S * R);
  masks("tolerated", T.stol.data(), S * KT);
  ints("spec_flags", T.sflags.data(), S);
  ints("aff_term_off", T.aff_off.data(), S + 1);
  masks("aff_terms", T.aff_terms.data(), T.aff_terms.size());
  ints("pre_term_off", T.pre_off.data(), S + 1);
  masks("pre_terms", T.pre_terms.data(), T.pre_terms.size());
  masks("wanted_ports", T.wanted.data(), S * KP);
  ints("spread_off", T.spread_off.data(), S + 1);
  std::vector<int32_t> spec(P), pin(P);
  for (size_t p = 0; p < P; ++p) encode_row(h, h->pending[p], &spec[p], &pin[p]);
  ints("pod_spec", spec.data(), P);
  ints("pod_node_name", pin.data(), P);
  {  // what an assumed pod of each spec adds to its node besides resources (ykpred_spec_effects_t)
    std::vector<int32_t> fx_off, fx_cls, fx_cnt;
    std::vector<uint64_t> occupied;
    h->enc.spec_effects(h->spec_templates, &fx_off, &fx_cls, &fx_cnt, &occupied);
    ints("effect_off", fx_off.data(), fx_off.size());
    ints("effect_class", fx_cls.data(), fx_cls.size());
    ints("effect_count", fx_cnt.data(), fx_cnt.size());
    masks("occupied_ports", occupied.data(), S * KP);
  }
  o += "\"spread_constraints\":" + std::to_string(T.spread.size()) + "}";
  if (out && len > 0) copy_out(o, out, len);
  return (int64_t)o.size() + 1;
}

int32_t ykhost_sync(ykhost_t* h) { YKHOST_LOCKED(h); return sync(h); }

int32_t ykhost_set_row_stride(ykhost_t* h, int32_t words) {
  YKHOST_LOCKED(h);
  if (words < 0 || words % 16 != 0) return fail(h, "row stride must be a multiple of 16 words (0 = automatic)");
  if (words != h->row_stride_words) h->dirty_all = true;
  h->row_stride_words = words;
  return 0;
}

int32_t ykhost_set_row_capacity(ykhost_t* h, int32_t rows) {
  YKHOST_LOCKED(h);
  if (rows < 0) return fail(h, "row capacity must be >= 0 (0 = automatic)");
  if (rows != h->row_capacity) h->dirty_all = true;
  h->row_capacity = rows;
  return 0;
}

int32_t ykhost_comm_init(ykhost_t* h, const uint8_t* id, int32_t rank, int32_t world, int32_t node_offset) {
  YKHOST_LOCKED(h);
  int rc = sync(h);  // the engine of the final dictionary shape must exist before the communicator is attached to it
  if (rc) return rc;
  rc = ykpred_comm_init(h->eng, id, rank, world, node_offset);
  if (rc) return fail(h, std::string("ykpred_comm_init: ") + ykpred_last_error(h->eng), rc);
  h->comm_attached = true;
  return 0;
}

int32_t ykhost_comm_destroy(ykhost_t* h) {
  YKHOST_LOCKED(h);
  if (h->eng) ykpred_comm_destroy(h->eng);
  h->comm_attached = false;
  return 0;
}
ykpred_engine_t* ykhost_engine(ykhost_t* h) { return h->eng; }

int32_t ykhost_evaluate(ykhost_t* h, int32_t allocate, uint32_t options) {
  YKHOST_LOCKED(h);
  int rc = sync(h);
  if (rc) return rc;
  ykpred_eval_args_t a{};
  a.prefilter_plugins = allocate ? h->alloc_pre : h->res_pre;
  a.filter_plugins = allocate ? h->alloc_filt : h->res_filt;
  a.options = options;
  rc = ykpred_eval(h->eng, &a);
  if (rc) return fail(h, std::string("ykpred_eval: ") + ykpred_last_error(h->eng), rc);
  h->eval_dirty_nodes.clear();
  h->eval_dirty_rows.clear();
  h->decisions_stale = !(options & (YKPRED_OUT_DECISIONS | YKPRED_OUT_DECISION_KEYS));
  h->last_eval_phase = (options & YKPRED_EVAL_DIRECT) || !(options & YKPRED_OUT_BITMAP) ? -1 : (allocate ? 1 : 0);
  h->last_eval_options = options;
  h->resident.valid = false;  // a new answer: mirrored on the first callback that wants it
  return 0;
}

int32_t ykhost_evaluate_dirty(ykhost_t* h, int32_t allocate, uint32_t options, int32_t* columns_patched) {
  YKHOST_LOCKED(h);
  if (columns_patched) *columns_patched = -1;
  // topology constraints (spread, inter-pod affinity) couple all nodes through their histograms: ykpred_eval_nodes rebuilds
  // them and rewrites the rows of the classes whose PreFilter state moved (row changes alone never touch the histograms)
  const bool nodes_touched = !h->eval_dirty_nodes.empty();
  const bool incremental = !h->dirty_all && !h->dirty_pods && h->last_eval_phase == (allocate ? 1 : 0);
  if (!incremental) return ykhost_evaluate(h, allocate, options);
  int rc = sync(h);  // uploads the touched node rows and ask rows
  if (rc) return rc;
  if (h->dirty_all || h->dirty_pods) return ykhost_evaluate(h, allocate, options);  // the sync had to rebuild tables
  ykpred_eval_args_t a{};
  a.prefilter_plugins = allocate ? h->alloc_pre : h->res_pre;
  a.filter_plugins = allocate ? h->alloc_filt : h->res_filt;
  a.options = options;
  const bool want_dec = options & (YKPRED_OUT_DECISIONS | YKPRED_OUT_DECISION_KEYS);
  const bool rows_touched = !h->eval_dirty_rows.empty();
  // columns first (they patch every row through its class), then the changed rows as a whole
  if (want_dec && h->decisions_stale && !nodes_touched) {
    // earlier patches ran without decisions: refresh bin-pack order, class decisions and the per-pod scatter (no bitmap pass)
    ykpred_eval_args_t b = a;
    b.options = (options & ~(uint32_t)YKPRED_OUT_BITMAP) | YKPRED_EVAL_SKIP_BITMAP;
    rc = ykpred_eval(h->eng, &b);
    if (rc == YKPRED_E_STATE) return ykhost_evaluate(h, allocate, options);  // (the pod classes were due for a rebuild: full pass)
    if (rc) return fail(h, std::string("ykpred_eval: ") + ykpred_last_error(h->eng), rc);
  }
  // A node-sharded engine with topology constraints: the histograms couple the shards, so the step is COLLECTIVE — every
  // shard's host calls ykhost_evaluate_dirty for it, whether or not one of ITS nodes changed, and the engine call below is
  // entered with an empty node list too. Both ykpred_eval_nodes and the full ykpred_eval it may fall back to perform exactly
  // one histogram exchange, so the shards stay in step whichever of the two each of them takes.
  const bool collective_step = h->comm_attached && h->enc.KD > 0;
  // hosts that carry the histograms between the shards themselves (no communicator) split the step in two calls:
  // YKPRED_EVAL_SPREAD_COUNT_ONLY (this shard's histograms are rebuilt, nothing else happens yet), then ..._COUNTS_READY
  const bool two_phase = options & (YKPRED_EVAL_SPREAD_COUNT_ONLY | YKPRED_EVAL_SPREAD_COUNTS_READY);
  // The mirrored class rows only survive purely column-local patches of THIS host's own nodes: with topology constraints the
  // engine rewrites whole rows of every class whose histograms moved — on a sharded cluster also for a node of another shard,
  // which this host's touch_node never saw.
  if (h->enc.KD > 0 || collective_step || two_phase) h->resident.valid = false;
  if (nodes_touched || collective_step || two_phase) {
    int32_t none = 0;
    rc = ykpred_eval_nodes(h->eng, &a, (int32_t)h->eval_dirty_nodes.size(), h->eval_dirty_nodes.empty() ? &none : h->eval_dirty_nodes.data());
    if (rc == YKPRED_E_STATE || rc == YKPRED_E_UNSUPPORTED) return ykhost_evaluate(h, allocate, options);
    if (rc) return fail(h, std::string("ykpred_eval_nodes: ") + ykpred_last_error(h->eng), rc);
    if (options & YKPRED_EVAL_SPREAD_COUNT_ONLY) return 0;  // the dirty lists stay: the second call finishes the step
  }
  if (!h->eval_dirty_rows.empty()) {
    std::sort(h->eval_dirty_rows.begin(), h->eval_dirty_rows.end());
    h->eval_dirty_rows.erase(std::unique(h->eval_dirty_rows.begin(), h->eval_dirty_rows.end()), h->eval_dirty_rows.end());
    while (!h->eval_dirty_rows.empty() && h->eval_dirty_rows.back() >= (int)h->pending.size()) h->eval_dirty_rows.pop_back();
    rc = ykpred_eval_pods(h->eng, &a, (int32_t)h->eval_dirty_rows.size(), h->eval_dirty_rows.data());
    if (rc == YKPRED_E_STATE) return ykhost_evaluate(h, allocate, options);
    if (rc) return fail(h, std::string("ykpred_eval_pods: ") + ykpred_last_error(h->eng), rc);
  }
  if (columns_patched) {
    std::sort(h->eval_dirty_nodes.begin(), h->eval_dirty_nodes.end());
    *columns_patched = (int32_t)(std::unique(h->eval_dirty_nodes.begin(), h->eval_dirty_nodes.end()) - h->eval_dirty_nodes.begin());
  }
  h->eval_dirty_nodes.clear();
  h->eval_dirty_rows.clear();
  if (want_dec)
    h->decisions_stale = false;
  else if (nodes_touched || rows_touched)
    h->decisions_stale = true;
  h->resident.order.clear();     // the bin-pack order may have moved
  h->resident.peek_pod = -1;     // and so may any row
  return 0;
}

int32_t ykhost_predicates(ykhost_t* h, int32_t pod, int32_t node, int32_t allocate, char* plugin, int32_t plugin_len, char* msg,
                          int32_t msg_len) {
  YKHOST_LOCKED(h);
  if (pod < 0 || pod >= (int)h->pending.size() || node < 0 || node >= (int)h->nodes.size()) return fail(h, "index out of range", -1);
  int rc = sync(h);
  if (rc) return rc;
  {
    auto un = h->enc.unsupported.find(h->pending[(size_t)pod]->tpl);
    if (un != h->enc.unsupported.end()) {  // not evaluated by the engine: the caller's CPU predicate manager answers
      h->routed_to_cpu++;
      copy_out("", plugin, plugin_len);
      copy_out(un->second, msg, msg_len);
      return fail(h, "ask is not evaluated by the engine (route it to the CPU predicate manager): " + un->second, YKHOST_E_UNSUPPORTED);
    }
  }
  const uint32_t pre = allocate ? h->alloc_pre : h->res_pre, filt = allocate ? h->alloc_filt : h->res_filt;
  const int phase = allocate ? 1 : 0;
  int fit = -1, code = 0;
  uint32_t reason = 0;
  bool have_code = false;
  auto unpack = [&](uint32_t w) {
    fit = (w >> 8) & 1;
    code = (int)(w & 0xff);
    reason = ((w >> 9) & 0xfu) | ((w >> 13) << YKPRED_REASON_RESOURCE_SHIFT);
    have_code = true;
  };
  // (1) the resident answer: mirror it on the first callback after an evaluation of this phase
  ykhost::Resident& R = h->resident;
  if (!R.valid && h->last_eval_phase == phase && h->eval_dirty_nodes.empty() && h->eval_dirty_rows.empty()) {
    int32_t C = 0;
    if (ykpred_answer_state(h->eng, pre, filt, &C) == YKPRED_OK) {
      ykpred_layout_t lay{};
      ykpred_get_layout(h->eng, &lay);
      R = ykhost::Resident{};
      R.phase = phase;
      R.C = C;
      R.N = lay.num_nodes;
      R.row_words = lay.row_words;
      R.node_dirty.assign((size_t)R.N, 0);
      bool ok = true;
      if ((int64_t)C * R.row_words * 8 <= h->resident_budget_bytes) {
        R.rows.assign((size_t)C * (size_t)R.row_words, 0);
        ok = ykpred_read_class_rows(h->eng, R.rows.data()) == YKPRED_OK;
        h->resident_fetches++;
      }
      R.valid = ok;
    }
  }
  int cls = -1;
  if (R.valid && R.phase == phase && node < R.N && ykpred_pod_class(h->eng, pod, &cls) == YKPRED_OK && cls >= 0 && cls < R.C) {
    if (R.node_dirty[(size_t)node]) {
      // the column moved since the mirror was taken (AssumePod / UpdateNode): this one pair straight from the tables
      uint8_t f = 0, c = 0;
      uint32_t why = 0;
      const int32_t p1 = pod, n1 = node;
      rc = ykpred_query(h->eng, 1, &p1, &n1, pre, filt, &f, &c, &why);
      if (rc) return fail(h, std::string("ykpred_query: ") + ykpred_last_error(h->eng), rc);
      fit = f;
      code = c;
      reason = why;
      have_code = true;
      h->served_dirty_column++;
    } else if (!R.rows.empty()) {
      fit = (int)((R.rows[(size_t)cls * (size_t)R.row_words + (size_t)(node >> 6)] >> (node & 63)) & 1ull);
      h->served_resident++;
    } else {
      // too many classes to mirror: the ask's own row (one small copy per ask), kept for its following callbacks
      if (R.peek_pod != pod) {
        R.peek_row.assign((size_t)R.row_words, 0);
        R.peek_pod = ykpred_peek_row(h->eng, pod, pre, filt, R.peek_row.data(), nullptr, nullptr) == YKPRED_OK ? pod : -1;
        h->resident_fetches++;
      }
      if (R.peek_pod == pod) {
        fit = (int)((R.peek_row[(size_t)(node >> 6)] >> (node & 63)) & 1ull);
        h->served_resident++;
      }
    }
    if (fit == 0 && !have_code) {
      // the failing plugin of the pair: the packed per-node answers of this CLASS (its members share every projection a
      // plugin sees), one device round trip per class
      auto it = R.class_codes.find(cls);
      if (it == R.class_codes.end()) {
        if (R.class_codes.size() >= 1024) R.class_codes.clear();
        std::vector<uint32_t> packed((size_t)R.N, 0);
        rc = ykpred_query_pod_packed(h->eng, pod, pre, filt, packed.data());
        if (rc) return fail(h, std::string("ykpred_query_pod_packed: ") + ykpred_last_error(h->eng), rc);
        it = R.class_codes.emplace(cls, std::move(packed)).first;
        h->code_fetches++;
      }
      unpack(it->second[(size_t)node]);
    }
  }
  // (2) no resident answer for this pair: one device round trip per (ask, phase), the answers for all nodes cached
  if (fit < 0) {
    if (h->answers.pod != pod || h->answers.phase != phase) {
      const size_t N = h->nodes.size();
      h->answers.fit.assign(N, 0);
      h->answers.code.assign(N, 0);
      h->answers.reason.assign(N, 0);
      rc = ykpred_query_pod(h->eng, pod, pre, filt, h->answers.fit.data(), h->answers.code.data(), h->answers.reason.data());
      if (rc) return fail(h, std::string("ykpred_query_pod: ") + ykpred_last_error(h->eng), rc);
      h->answers.pod = pod;
      h->answers.phase = phase;
    }
    fit = h->answers.fit[(size_t)node];
    code = h->answers.code[(size_t)node];
    reason = h->answers.reason[(size_t)node];
    h->served_query++;
  }
  if (fit) {
    copy_out("", plugin, plugin_len);
    copy_out("", msg, msg_len);
    return 1;
  }
  copy_out(code < 9 ? kPluginNames[code] : "", plugin, plugin_len);
  copy_out(compose_message(h, *h->pending[(size_t)pod], *h->nodes[(size_t)node], code, reason), msg, msg_len);
  return 0;
}

// The first `k` feasible nodes of an ask in bin-pack order (what the core's node iterator would find, without one callback
// per node): the ask's row of the resident answer walked in the order of the evaluation that produced the decisions.
// → number of nodes written; YKPRED_E_STATE (-4) when no evaluation with decisions of this phase is current (a node or
// ask changed since: evaluate first), YKHOST_E_UNSUPPORTED for a routed ask.
int32_t ykhost_candidates(ykhost_t* h, int32_t pod, int32_t allocate, int32_t k, int32_t* out_nodes) {
  YKHOST_LOCKED(h);
  if (pod < 0 || pod >= (int)h->pending.size() || k < 0 || (k > 0 && !out_nodes)) return fail(h, "bad argument", -1);
  if (k == 0 || h->nodes.empty()) return 0;
  char plugin[8], msg[8];
  int rc = ykhost_predicates(h, pod, 0, allocate, plugin, sizeof plugin, msg, sizeof msg);  // sync + routing + mirrors the answer
  if (rc < 0) return rc;
  ykhost::Resident& R = h->resident;
  const uint32_t pre = allocate ? h->alloc_pre : h->res_pre, filt = allocate ? h->alloc_filt : h->res_filt;
  int cls = -1;
  if (!R.valid || R.phase != (allocate ? 1 : 0) || h->decisions_stale || !h->eval_dirty_nodes.empty() || !h->eval_dirty_rows.empty() ||
      ykpred_pod_class(h->eng, pod, &cls) != YKPRED_OK || cls < 0 || cls >= R.C)
    return fail(h, "no current evaluation with decisions for this ask (a node or the ask changed since): evaluate first", YKPRED_E_STATE);
  if (R.order.empty()) {
    R.order.assign((size_t)R.N, 0);
    if (ykpred_read_order(h->eng, R.order.data()) != YKPRED_OK) {
      R.order.clear();
      return fail(h, std::string("ykpred_read_order: ") + ykpred_last_error(h->eng), YKPRED_E_STATE);
    }
  }
  const uint64_t* row = nullptr;
  if (!R.rows.empty() && R.n_dirty == 0) {
    row = R.rows.data() + (size_t)cls * (size_t)R.row_words;
  } else {
    // columns were patched on the device since the mirror was taken (ykhost_evaluate_dirty): the ask's CURRENT row
    if (R.peek_pod != pod) {
      R.peek_row.assign((size_t)R.row_words, 0);
      R.peek_pod = ykpred_peek_row(h->eng, pod, pre, filt, R.peek_row.data(), nullptr, nullptr) == YKPRED_OK ? pod : -1;
    }
    if (R.peek_pod != pod) return fail(h, std::string("ykpred_peek_row: ") + ykpred_last_error(h->eng), YKPRED_E_STATE);
    row = R.peek_row.data();
  }
  int found = 0;
  for (int i = 0; i < R.N && found < k; ++i) {
    const int n = R.order[(size_t)i];
    if ((row[(size_t)(n >> 6)] >> (n & 63)) & 1ull) out_nodes[found++] = n;
  }
  return found;
}

// What AssumePod of an ask adds to its node besides its resources — the selector-class counts behind PodTopologySpread /
// InterPodAffinity and the host ports it occupies — per spec, so that the device round can keep that state current ask by ask
// (ykpred_set_spec_effects). Only a round needs it: built and uploaded on the first round after the spec table moved.
static int upload_spec_effects(ykhost* h) {
  if (h->fx_uploaded) return 0;
  if (h->enc.KS == 0 && h->enc.KP == 0) return 0;  // nothing but resources couples the asks
  std::vector<int32_t> off, cls, cnt;
  std::vector<uint64_t> occupied;
  h->enc.spec_effects(h->spec_templates, &off, &cls, &cnt, &occupied);
  cls.push_back(0);  // (never empty: the C side takes data() of them)
  cnt.push_back(0);
  ykpred_spec_effects_t fx{};
  fx.count = (int32_t)h->spec_templates.size();
  fx.contrib_off = off.data();
  fx.contrib_class = cls.data();
  fx.contrib_count = cnt.data();
  fx.occupied_ports = h->enc.KP > 0 ? occupied.data() : nullptr;
  const int rc = ykpred_set_spec_effects(h->eng, &fx);
  if (rc) return fail(h, std::string("ykpred_set_spec_effects: ") + ykpred_last_error(h->eng), rc);
  h->fx_uploaded = true;
  h->fx_uploads++;
  return 0;
}

// One scheduling round with conflict-resolved decisions. See ykhost.h.
int32_t ykhost_allocate_round(ykhost_t* h, int32_t n, const int32_t* asks, int32_t apply, int32_t* out_nodes) {
  YKHOST_LOCKED(h);
  if (n < 0 || (n > 0 && !out_nodes)) return fail(h, "bad argument", -1);
  const uint32_t want = YKPRED_OUT_BITMAP | YKPRED_OUT_COUNTS | YKPRED_OUT_DECISIONS;
  // a current evaluation with decisions (column / row patches where they suffice)
  int rc = ykhost_evaluate_dirty(h, 1, want, nullptr);
  if (rc) return rc;
  const int P = (int)h->pending.size();
  // A node-sharded engine (a communicator with world > 1 is attached to it): the round is COLLECTIVE — every rank calls with the same
  // asks — and node indices in and out of it are indices in the WHOLE cluster (this shard's first node = shard_offset). An ask that
  // went to another shard's node is assumed here as well (it is nobody's pending ask any more), without touching a node of this mirror.
  int32_t shard_world = 1, shard_offset = 0;
  ykpred_comm_info(h->eng, nullptr, &shard_world, &shard_offset);
  const bool sharded = shard_world > 1;
  std::vector<int32_t> list, slot;  // asks of the round that the engine evaluates, and their position in `asks`
  list.reserve((size_t)n);
  slot.reserve((size_t)n);
  for (int i = 0; i < n; ++i) {
    const int row = asks ? asks[i] : i;
    if (row < 0 || row >= P) return fail(h, "ask index out of range", -1);
    Pod* p = h->pending[(size_t)row];
    out_nodes[i] = -1;
    if (h->enc.unsupported.count(p->tpl)) {
      out_nodes[i] = -2;
      h->round_asks_routed++;
    } else if (p->assumed) {  // allocated by an earlier round and not bound yet: it keeps its node
      auto nt = h->node_ix.find(p->node_name);
      out_nodes[i] = p->remote_node >= 0 ? p->remote_node : nt == h->node_ix.end() ? -1 : nt->second + shard_offset;
    } else {
      list.push_back(row);
      slot.push_back(i);
    }
  }
  auto assume = [&](Pod* p, int node) {  // ykhost_assume_pod without the two name lookups
    detach_from_node(h, p);
    p->node_name = h->nodes[(size_t)node]->node.name;
    cache_update_pod(h, p, p, false, false);
    p->assumed = true;
  };
  int placed = 0;
  if (list.empty()) return 0;
  std::vector<int32_t> got(list.size(), -1);
  // YKHOST_ROUND_ON_HOST=1 (tests): the specs' effects are withheld, so that a round with topology constraints or host ports takes
  // the ask-by-ask path below — the path node-sharded engines still take
  const char* on_host_env = getenv("YKHOST_ROUND_ON_HOST");  // (read per round: the tests switch it inside one process)
  const bool round_on_host = on_host_env && atoi(on_host_env) != 0;
  if (round_on_host) {
    ykpred_set_spec_effects(h->eng, nullptr);
    h->fx_uploaded = false;
  }
  if (!round_on_host) {
    rc = upload_spec_effects(h);
    if (rc) return rc;
  }
  rc = ykpred_allocate_round(h->eng, h->alloc_pre, h->alloc_filt, (int32_t)list.size(), list.data(), got.data());
  if (rc == YKPRED_E_STATE) {
    // the patched evaluation is current but its rank-ordered planes are not (e.g. a template was appended since the last
    // decision pass): one full pass, then the round
    rc = ykhost_evaluate(h, 1, want);
    if (rc) return rc;
    if (!round_on_host) {
      rc = upload_spec_effects(h);
      if (rc) return rc;
    }
    rc = ykpred_allocate_round(h->eng, h->alloc_pre, h->alloc_filt, (int32_t)list.size(), list.data(), got.data());
  }
  if (rc == YKPRED_OK) {
    h->rounds_on_device++;
    h->round_asks_on_device += (int64_t)list.size();
    ensure_uid_index(h);
    for (size_t k = 0; k < list.size(); ++k) {
      out_nodes[(size_t)slot[k]] = got[k];
      if (got[k] < 0) continue;
      ++placed;
      if (!apply) continue;
      Pod* p = h->pending[(size_t)list[k]];
      const int local = got[k] - shard_offset;  // (the engine of a sharded cluster answers in cluster-wide indices)
      if (!sharded || (local >= 0 && local < (int)h->nodes.size())) {
        assume(p, sharded ? local : got[k]);
      } else {
        p->assumed = true;
        p->remote_node = got[k];
      }
    }
    return placed;
  }
  if (rc != YKPRED_E_UNSUPPORTED) return fail(h, std::string("ykpred_allocate_round: ") + ykpred_last_error(h->eng), rc);
  // A node-sharded cluster never decides ask by ask here: the path below takes this shard's own answer and assumes on this shard's
  // node — every rank would place the ask on a different node, in shard-local indices. What the engine's collective round does not
  // cover is refused as such; the caller keeps the per-pair Predicates() path of the CPU manager for these asks.
  if (sharded)
    return fail(h, std::string("a round on a node-sharded cluster that the engine's collective round does not cover (") + ykpred_last_error(h->eng) +
                       "): not decided shard by shard", YKHOST_E_UNSUPPORTED);
  // ---- ask by ask: something other than node resources couples the asks (topology histograms, host ports) and the specs' effects are withheld
  if (!apply) return fail(h, std::string("this round is decided ask by ask and has to assume as it goes (apply = 1): ") + ykpred_last_error(h->eng), YKHOST_E_UNSUPPORTED);
  ensure_uid_index(h);
  ykpred_layout_t lay{};
  ykpred_get_layout(h->eng, &lay);
  std::vector<uint64_t> row_buf((size_t)std::max(lay.row_words, 1));
  for (size_t k = 0; k < list.size(); ++k) {
    rc = ykhost_evaluate_dirty(h, 1, want, nullptr);  // the allocations so far are columns (and topology rows) to patch
    if (rc) return rc;
    ykpred_get_layout(h->eng, &lay);
    row_buf.resize((size_t)std::max(lay.row_words, 1));
    int32_t cnt = 0, dec = -1;
    rc = ykpred_peek_row(h->eng, list[k], h->alloc_pre, h->alloc_filt, row_buf.data(), &cnt, &dec);
    if (rc) return fail(h, std::string("ykpred_peek_row: ") + ykpred_last_error(h->eng), rc);
    h->round_asks_one_by_one++;
    out_nodes[(size_t)slot[k]] = dec;
    if (dec < 0) continue;
    ++placed;
    assume(h->pending[(size_t)list[k]], dec);
  }
  return placed;
}

int64_t ykhost_device_errors(ykhost_t* h) {
  YKHOST_LOCKED(h);
  return h->device_errors;
}

int32_t ykhost_round_stats(ykhost_t* h, int64_t* out4) {
  YKHOST_LOCKED(h);
  out4[0] = h->rounds_on_device;
  out4[1] = h->round_asks_on_device;
  out4[2] = h->round_asks_one_by_one;
  out4[3] = h->round_asks_routed;
  return 0;
}

// out[0] = Predicates() calls answered from the mirrored resident answer, [1] = answered per pair because the node's column
// changed since, [2] = answered by a whole-ask device query (no current evaluation), [3] = answer fetches (class-row mirror
// or single rows), [4] = failing-plugin fetches (one per class)
int32_t ykhost_resident_stats(ykhost_t* h, int64_t* out5) {
  YKHOST_LOCKED(h);
  out5[0] = h->served_resident;
  out5[1] = h->served_dirty_column;
  out5[2] = h->served_query;
  out5[3] = h->resident_fetches;
  out5[4] = h->code_fetches;
  return 0;
}

// Appends the victim columns of one PreemptionPredicates query: request vectors, "really removed" flags (nil victims and
// victims that are not on the node are ignored, predicate_manager.go:181-192) and the node's host-port bits after each removal.
static void append_victims(ykhost* h, const NodeInfo& ni, const char* const* victim_uids, int32_t nv, std::vector<int64_t>* vreq,
                           std::vector<uint8_t>* present, std::vector<uint64_t>* ports_after) {
  const int R = h->enc.R, KP = h->enc.KP;
  const size_t base = present->size();
  vreq->resize((base + (size_t)nv) * (size_t)R, 0);
  present->resize(base + (size_t)nv, 0);
  ports_after->resize((base + (size_t)nv) * (size_t)std::max(KP, 1), 0);
  std::vector<const Pod*> remaining(ni.pods.begin(), ni.pods.end());
  for (int i = 0; i < nv; ++i) {
    const Pod* v = nullptr;
    if (victim_uids && victim_uids[i])
      for (const Pod* q : remaining)
        if (q->uid == victim_uids[i]) v = q;
    if (v) {
      remaining.erase(std::find(remaining.begin(), remaining.end(), v));
      (*present)[base + (size_t)i] = 1;
      Resource r = to_resource(v->tpl->requests);
      int64_t* row = vreq->data() + (base + (size_t)i) * (size_t)R;
      row[0] = r.milli_cpu;
      row[1] = r.memory;
      row[2] = r.ephemeral;
      for (size_t s = 0; s < h->enc.scalar_names.size(); ++s) {
        auto it = r.scalar.find(h->enc.scalar_names[s]);
        if (it != r.scalar.end()) row[3 + s] = it->second;
      }
    }
    if (KP) h->enc.encode_ports(remaining, ports_after->data() + (base + (size_t)i) * (size_t)KP);  // NodeInfo.UsedPorts after this step
  }
}

int32_t ykhost_preemption_predicates_batch(ykhost_t* h, int32_t nq, const int32_t* pods, const int32_t* nodes, const int32_t* victim_off,
                                           const char* const* victim_uids, const int32_t* start_index, int32_t* out_index) {
  YKHOST_LOCKED(h);
  if (nq < 0 || (nq > 0 && (!pods || !nodes || !victim_off || !start_index || !out_index))) return fail(h, "bad argument", -2);
  int rc = sync(h);
  if (rc) return rc;
  ensure_uid_index(h);
  std::vector<int64_t> vreq;
  std::vector<uint8_t> present;
  std::vector<uint64_t> ports_after;
  for (int q = 0; q < nq; ++q) {
    if (pods[q] < 0 || pods[q] >= (int)h->pending.size() || nodes[q] < 0 || nodes[q] >= (int)h->nodes.size() || victim_off[q + 1] < victim_off[q])
      return fail(h, "index out of range", -2);
    append_victims(h, *h->nodes[(size_t)nodes[q]], victim_uids ? victim_uids + victim_off[q] : nullptr, victim_off[q + 1] - victim_off[q], &vreq,
                   &present, &ports_after);
  }
  int64_t no_req = 0;
  uint8_t no_flag = 0;
  rc = ykpred_preemption_batch(h->eng, nq, pods, nodes, victim_off, vreq.empty() ? &no_req : vreq.data(), present.empty() ? &no_flag : present.data(),
                               h->enc.KP ? ports_after.data() : nullptr, start_index, h->alloc_pre, h->alloc_filt, out_index);
  if (rc) return fail(h, std::string("ykpred_preemption_batch: ") + ykpred_last_error(h->eng), rc);
  return 0;
}

int32_t ykhost_preemption_predicates(ykhost_t* h, int32_t pod, int32_t node, const char* const* victim_uids, int32_t nv, int32_t start) {
  YKHOST_LOCKED(h);
  const int32_t off[2] = {0, nv};
  int32_t out = -1;
  int rc = ykhost_preemption_predicates_batch(h, 1, &pod, &node, off, victim_uids, &start, &out);
  return rc ? -2 : out;
}

// Context.IsPodFitNode (context.go:696-716) behind AsyncRMCallback.Predicates (scheduler_callback.go:203-205): the core
// names the ask by allocation key (= pod UID) and the node by id.
int32_t ykhost_is_pod_fit_node(ykhost_t* h, const char* allocation_key, const char* node_id, int32_t allocate, char* err, int32_t err_len) {
  YKHOST_LOCKED(h);
  copy_out("", err, err_len);
  ensure_uid_index(h);
  auto it = h->by_uid.find(allocation_key ? allocation_key : "");
  if (it == h->by_uid.end()) {
    copy_out("predicates were not run because pod was not found in cache", err, err_len);  // ErrorPodNotFound (context.go:67)
    return YKHOST_E_POD_NOT_FOUND;
  }
  auto nt = h->node_ix.find(node_id ? node_id : "");
  if (nt == h->node_ix.end()) {
    copy_out("predicates were not run because node was not found in cache", err, err_len);  // ErrorNodeNotFound (:68)
    return YKHOST_E_NODE_NOT_FOUND;
  }
  if (!it->second->ask) {
    copy_out("pod is cached but holds no row of the ask table (it is bound, not a pending ask)", err, err_len);
    return fail(h, "pod holds no ask row", YKHOST_E_NOT_AN_ASK);
  }
  char plugin[64], msg[1024];
  int rc = ykhost_predicates(h, it->second->row, nt->second, allocate, plugin, sizeof plugin, msg, sizeof msg);
  if (rc < 0) {  // includes YKHOST_E_UNSUPPORTED: the Go side calls the CPU manager for this ask
    copy_out(h->err, err, err_len);
    return rc;
  }
  if (rc == 0) copy_out(std::string("failed plugin: '") + plugin + "'\n" + msg, err, err_len);  // errors.Join (:713)
  return rc;
}

// Context.IsPodFitNodeViaPreemption (context.go:718-742) behind AsyncRMCallback.PreemptionPredicates
// (scheduler_callback.go:207-216): → index of the last victim that has to go, -1 = {Success: false} (also for an unknown
// ask or node; victims that are not in the cache count as nil pods).
int32_t ykhost_is_pod_fit_node_via_preemption(ykhost_t* h, const char* allocation_key, const char* node_id,
                                              const char* const* preempt_allocation_keys, int32_t num_keys, int32_t start_index) {
  YKHOST_LOCKED(h);
  ensure_uid_index(h);
  auto it = h->by_uid.find(allocation_key ? allocation_key : "");
  auto nt = h->node_ix.find(node_id ? node_id : "");
  if (it == h->by_uid.end() || nt == h->node_ix.end() || !it->second->ask) return -1;
  return ykhost_preemption_predicates(h, it->second->row, nt->second, preempt_allocation_keys, num_keys, start_index);
}

int32_t ykhost_pod_request_json(ykhost_t* h, int32_t pod, char* out, int32_t len) {
  YKHOST_LOCKED(h);
  if (pod < 0 || pod >= (int)h->pending.size()) return fail(h, "index out of range");
  std::string js = "{";
  bool first = true;
  for (auto& kv : h->pending[(size_t)pod]->tpl->requests) {
    if (!first) js += ",";
    first = false;
    js += "\"" + kv.first + "\":" + std::to_string(kv.second);
  }
  js += "}";
  copy_out(js, out, len);
  return 0;
}

// 1 = the ask is evaluated on the device, 0 = it is routed to the CPU predicate manager (reason copied out)
int32_t ykhost_ask_supported(ykhost_t* h, int32_t pod, char* reason, int32_t reason_len) {
  YKHOST_LOCKED(h);
  if (pod < 0 || pod >= (int)h->pending.size()) return fail(h, "index out of range");
  if (h->device < 0) {  // mirror-only handle: encode without a device (like ykhost_encoded_tables_json)
    EncodedTables T;
    int rc = encode_tables(h, &T);
    h->dirty_all = true;
    if (rc) return rc;
  } else {
    int rc = sync(h);
    if (rc) return rc;
  }
  auto un = h->enc.unsupported.find(h->pending[(size_t)pod]->tpl);
  copy_out(un == h->enc.unsupported.end() ? "" : un->second, reason, reason_len);
  return un == h->enc.unsupported.end() ? 1 : 0;
}
// out[0] = asks marked unsupported at the last encode, out[1] = Predicates() calls answered YKHOST_E_UNSUPPORTED so far,
// out[2] = new asks whose selector requirements were added to the dictionaries in place (no re-encode)
int32_t ykhost_routing_stats(ykhost_t* h, int64_t* out2) {
  YKHOST_LOCKED(h);
  out2[0] = h->unsupported_asks;
  out2[1] = h->routed_to_cpu;
  out2[2] = h->dictionary_growths;
  return 0;
}

int32_t ykhost_stats(const ykhost_t* h, int64_t* o) {
  YKHOST_LOCKED(h);
  o[0] = h->enc.R;
  o[1] = h->enc.KT;
  o[2] = h->enc.W;
  o[3] = (int64_t)h->enc.taint_dict.size();
  o[4] = (int64_t)h->enc.req_dict.size();
  o[5] = (int64_t)h->pool.all().size();
  o[6] = (int64_t)h->spec_templates.size();
  o[7] = h->last_encode_us;
  return 0;
}

}  // extern "C"
