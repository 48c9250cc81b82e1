// ykoracle.cpp — CPU ORACLE for the yunikorn-k8shim predicate hot path.
//
// *** TEST INFRASTRUCTURE ONLY. ***  Nothing under yunikorn-k8shim_amd/ may include, link or call this
// file; only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg use it, and only as the
// checker / reported baseline. It is a per-(pod,node), object-model restatement of what the reference
// computes for one `Predicates()` call — fresh per-call state, a PreFilter pass over ALL nodes followed
// by the ordered Filter list with early exit — written for fidelity, not speed.
//
// Provenance of every rule:
//   * Driver (authoritative, in-tree): /root/reference/pkg/plugin/predicates/predicate_manager.go
//       Predicates :134-139, predicatesReserve/Allocate :194-204, podFitsNode :206-219,
//       runPreFilterPlugins :221-254, runFilterPlugins :260-283, PreemptionPredicates :141-192,
//       phase plugin lists :321-373 (filter order :339-352).
//   * Request vectors (in-tree twin of upstream PodRequests): /root/reference/pkg/common/resource.go:56-182,273-301.
//   * NodeInfo bookkeeping as used by /root/reference/pkg/cache/external/scheduler_cache.go:166,179,324,363.
//   * The per-plugin arithmetic lives in k8s.io/kubernetes v1.36.1, k8s.io/component-helpers v0.36.1 and
//     k8s.io/apimachinery v0.36.1 (go.mod:36-46), which are NOT vendored under /root/reference and cannot be
//     built here (no Go toolchain, no network). Their published algorithms are restated below, plugin by
//     plugin, and pinned against every golden vector the reference's own tests hold for this path
//     (tests/golden/*.json, transcribed from predicate_manager_test.go and resource_test.go).
//   * PARITY UNPINNED at unit level in the reference (documented restatement only): TaintToleration Filter,
//     PodTopologySpread, bin-pack score (the latter lives in yunikorn-core, go.mod:24).
//
// Build: see oracle/Makefile (g++ -O2 -shared -fPIC -fopenmp).
#include <algorithm>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <limits>
#include <map>
#include <memory>
#include <set>
#include <string>
#include <tuple>
#include <unordered_map>
#include <vector>

#include "orc_json.h"
#include "orc_quantity.h"

namespace orc {

// ---------------------------------------------------------------------------------------------------
// Plugin identifiers. Order = default MultiPoint order documented at predicate_manager.go:339-352.
// ---------------------------------------------------------------------------------------------------
enum PluginBit : uint32_t {
  kNodeUnschedulable = 1u << 0,
  kNodeName = 1u << 1,
  kTaintToleration = 1u << 2,
  kNodeAffinity = 1u << 3,
  kNodePorts = 1u << 4,
  kNodeResourcesFit = 1u << 5,
  kPodTopologySpread = 1u << 6,
  kInterPodAffinity = 1u << 7,
};
// Codes returned as "failing plugin": 0 = "" (a PreFilter plugin rejected the pod itself, :236-238).
enum PluginCode : int {
  kCodeNone = 0,
  kCodeNodeUnschedulable = 1,
  kCodeNodeName = 2,
  kCodeTaintToleration = 3,
  kCodeNodeAffinity = 4,
  kCodeNodePorts = 5,
  kCodeNodeResourcesFit = 6,
  kCodePodTopologySpread = 7,
  kCodeInterPodAffinity = 8,
};

// ---------------------------------------------------------------------------------------------------
// Object model (subset of v1.Pod / v1.Node / framework.NodeInfo that the path reads).
// ---------------------------------------------------------------------------------------------------
using StrMap = std::map<std::string, std::string>;

struct Taint {
  std::string key, value, effect;
};
struct Toleration {
  std::string key, op, value, effect;
};
struct Requirement {  // v1.NodeSelectorRequirement / metav1.LabelSelectorRequirement
  std::string key, op;
  std::vector<std::string> values;
};
struct Term {  // v1.NodeSelectorTerm
  std::vector<Requirement> exprs, fields;
};
struct LabelSelector {
  bool present = false;  // nil *LabelSelector → labels.Nothing()
  StrMap match_labels;
  std::vector<Requirement> match_exprs;
};
struct SpreadConstraint {
  int32_t max_skew = 1;
  std::string topology_key, when_unsatisfiable;
  LabelSelector selector;
  bool has_min_domains = false;
  int32_t min_domains = 1;
  std::string node_affinity_policy = "Honor", node_taints_policy = "Ignore";
  std::vector<std::string> match_label_keys;
};
struct PodAffinityTerm {  // v1.PodAffinityTerm (required terms only)
  LabelSelector selector;
  std::vector<std::string> namespaces;
  // namespaceSelector: {} = labels.Everything() selects every namespace; a non-empty one needs Namespace objects (not modelled:
  // rejected at load time)
  bool has_namespace_selector = false;
  std::string topology_key;
};
struct HostPort {  // v1.ContainerPort with HostPort > 0, sanitized like HostPortInfo.sanitize ("" ip → 0.0.0.0, "" protocol → TCP)
  std::string protocol, ip;
  int64_t port = 0;
  bool operator<(const HostPort& o) const { return std::tie(ip, protocol, port) < std::tie(o.ip, o.protocol, o.port); }
};
struct Container {
  std::string name;
  StrMap requests;
  bool sidecar = false;  // initContainer with restartPolicy: Always (resource.go:184-186)
  std::vector<HostPort> host_ports;
};
struct Pod {
  std::string name, uid, ns;
  StrMap labels;
  std::string node_name;
  bool has_node_selector = false;  // pod.Spec.NodeSelector != nil
  StrMap node_selector;
  bool has_required = false;  // Affinity.NodeAffinity.RequiredDuringSchedulingIgnoredDuringExecution != nil
  std::vector<Term> terms;
  std::vector<Toleration> tolerations;
  std::vector<Container> containers, init_containers;
  bool has_overhead = false;
  StrMap overhead;
  StrMap pod_level_requests;
  std::vector<SpreadConstraint> spread;
  std::vector<PodAffinityTerm> pod_affinity, pod_anti_affinity;  // requiredDuringSchedulingIgnoredDuringExecution
  bool terminating = false;  // metadata.deletionTimestamp set
  std::string phase;
  // KEP-1287 in-place resize inputs of computeContainerResource (resource.go:110-127): status.containerStatuses and
  // status.initContainerStatuses by container name, and "the pending resize is infeasible" (isResizeInfeasible :132-142)
  struct ContainerStatus {
    StrMap allocated;        // allocatedResources
    bool has_resources = false;
    StrMap resources_requests;  // resources.requests
  };
  std::map<std::string, ContainerStatus> container_status;
  bool resize_infeasible = false;
};
struct Node {
  std::string name;
  StrMap labels;
  std::vector<Taint> taints;
  bool unschedulable = false;
  StrMap allocatable;
};

// framework.Resource
struct Resource {
  int64_t milli_cpu = 0, memory = 0, ephemeral = 0;
  int64_t allowed_pods = 0;
  std::map<std::string, int64_t> scalar;
};

// v1helper/schedutil IsScalarResourceName: extended (has '/', not kubernetes.io/, not "requests."-prefixed),
// hugepages-*, kubernetes.io/-prefixed native, attachable-volumes-*.
static bool has_prefix(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
static bool is_scalar_resource_name(const std::string& n) {
  bool native = n.find('/') == std::string::npos || n.find("kubernetes.io/") != std::string::npos;
  bool extended = !native && !has_prefix(n, "requests.");
  bool prefixed_native = n.find("kubernetes.io/") != std::string::npos;
  return extended || has_prefix(n, "hugepages-") || prefixed_native || has_prefix(n, "attachable-volumes-");
}

// int64 resource map in the units of resource.go:273-285 (cpu → MilliValue, everything else → Value).
using ResMap = std::map<std::string, int64_t>;
static ResMap get_resource(const StrMap& rl) {
  ResMap out;
  for (auto& kv : rl) out[kv.first] = kv.first == "cpu" ? quantity_milli(kv.second) : quantity_value(kv.second);
  return out;
}
static void res_add(ResMap& l, const ResMap& r) {  // resource.go Add
  for (auto& kv : r) l[kv.first] += kv.second;
}
static void res_update_max(ResMap& l, const ResMap& r) {  // resource.go:145-162
  for (auto& kv : r) {
    auto it = l.find(kv.first);
    if (it == l.end())
      l[kv.first] = kv.second;
    else if (kv.second > it->second)
      it->second = kv.second;
  }
}
static bool is_supported_pod_level(const std::string& n) {  // resourcehelper.IsSupportedPodLevelResource
  return n == "cpu" || n == "memory" || has_prefix(n, "hugepages-");
}

// Pod request vector. Follows resource.go:56-109 (GetPodResource) minus the YuniKorn-only "pods":1 entry
// (:58), i.e. upstream PodRequests — equality of the two is asserted by resource_test.go:201-202,219-220.
// computeContainerResource (:110-127): max(spec requests, status allocatedResources, status resources.requests) per
// resource, unless the pending resize is infeasible — then the status requests stand.
static ResMap container_requests(const Pod& p, const Container& c) {
  ResMap combined;
  res_update_max(combined, get_resource(c.requests));
  auto it = p.container_status.find(c.name);
  if (it != p.container_status.end()) {
    const Pod::ContainerStatus& st = it->second;
    if (p.resize_infeasible && st.has_resources) return get_resource(st.resources_requests);
    res_update_max(combined, get_resource(st.allocated));
    if (st.has_resources) res_update_max(combined, get_resource(st.resources_requests));
  }
  return combined;
}
static ResMap pod_requests(const Pod& p) {
  ResMap total;
  for (auto& c : p.containers) res_add(total, container_requests(p, c));  // :74-76
  if (!p.init_containers.empty()) {                                      // :81-83 → checkInitContainerRequest :164-182
    ResMap init_max, sidecars;
    for (auto& c : p.init_containers) {
      ResMap ic = container_requests(p, c);
      ResMap cur = ic;
      res_add(cur, sidecars);
      if (c.sidecar) res_add(sidecars, ic);
      res_update_max(init_max, cur);
    }
    res_add(total, sidecars);
    res_update_max(total, init_max);
  }
  if (!p.pod_level_requests.empty()) {  // :88-94
    for (auto& kv : get_resource(p.pod_level_requests))
      if (is_supported_pod_level(kv.first)) total[kv.first] = kv.second;
  }
  if (p.has_overhead) res_add(total, get_resource(p.overhead));  // :98-106
  return total;
}

// framework.Resource from a request map (upstream Resource.Add semantics: cpu/memory/ephemeral-storage/pods
// are first-class, scalar-named resources go to ScalarResources, anything else is dropped).
static Resource to_resource(const ResMap& m) {
  Resource r;
  for (auto& kv : m) {
    if (kv.first == "cpu")
      r.milli_cpu += kv.second;
    else if (kv.first == "memory")
      r.memory += kv.second;
    else if (kv.first == "ephemeral-storage")
      r.ephemeral += kv.second;
    else if (kv.first == "pods")
      r.allowed_pods += kv.second;
    else if (is_scalar_resource_name(kv.first))
      r.scalar[kv.first] += kv.second;
  }
  return r;
}

// schedutil.GetHostPorts: host ports of the containers and of the init containers that keep running (restartPolicy
// Always). k8s.io/kubernetes v1.36.1, not vendored; the reference pins only regular containers
// (predicate_manager_test.go:199-222).
static std::vector<HostPort> get_host_ports(const Pod& p) {
  std::vector<HostPort> out;
  for (auto& c : p.init_containers)
    if (c.sidecar) out.insert(out.end(), c.host_ports.begin(), c.host_ports.end());
  for (auto& c : p.containers) out.insert(out.end(), c.host_ports.begin(), c.host_ports.end());
  return out;
}
// framework.HostPortInfo.CheckConflict
static bool ports_conflict(const HostPort& want, const std::set<HostPort>& used) {
  if (want.port <= 0) return false;
  for (auto& u : used) {
    if (u.protocol != want.protocol || u.port != want.port) continue;
    if (want.ip == "0.0.0.0" || u.ip == "0.0.0.0" || u.ip == want.ip) return true;
  }
  return false;
}

// framework.NodeInfo: SetNode / AddPod / RemovePod as driven by scheduler_cache.go:166,179,324,363 and
// predicate_manager.go:158,185.
struct NodeInfo {
  Node node;
  std::vector<const Pod*> pods;
  Resource requested, allocatable;
  std::set<HostPort> used_ports;  // NodeInfo.UsedPorts (set semantics: Add / Remove per pod)

  void set_node(const Node& n) {
    node = n;
    allocatable = to_resource(get_resource(n.allocatable));
  }
  void add_pod(const Pod* p) {
    pods.push_back(p);
    for (auto& hp : get_host_ports(*p)) used_ports.insert(hp);
    Resource r = to_resource(pod_requests(*p));
    requested.milli_cpu += r.milli_cpu;
    requested.memory += r.memory;
    requested.ephemeral += r.ephemeral;
    for (auto& kv : r.scalar) requested.scalar[kv.first] += kv.second;
  }
  bool remove_pod(const Pod* p) {  // by UID, like upstream; false if absent (removePodFromNodeNoFail ignores)
    for (size_t i = 0; i < pods.size(); ++i) {
      if (pods[i]->uid == p->uid) {
        for (auto& hp : get_host_ports(*pods[i])) used_ports.erase(hp);
        Resource r = to_resource(pod_requests(*pods[i]));
        requested.milli_cpu -= r.milli_cpu;
        requested.memory -= r.memory;
        requested.ephemeral -= r.ephemeral;
        for (auto& kv : r.scalar) requested.scalar[kv.first] -= kv.second;
        pods.erase(pods.begin() + static_cast<long>(i));
        return true;
      }
    }
    return false;
  }
};

// ---------------------------------------------------------------------------------------------------
// apimachinery label validation (labels.NewRequirement) — a requirement that fails validation makes its
// whole node-selector TERM non-matching (pin: predicate_manager_test.go:793-820).
// ---------------------------------------------------------------------------------------------------
static bool is_alnum(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); }
static bool is_name_part(const std::string& s) {  // ([A-Za-z0-9][-A-Za-z0-9_.]*)?[A-Za-z0-9], 1..63
  if (s.empty() || s.size() > 63) return false;
  if (!is_alnum(s.front()) || !is_alnum(s.back())) return false;
  for (char c : s)
    if (!is_alnum(c) && c != '-' && c != '_' && c != '.') return false;
  return true;
}
static bool is_dns1123_subdomain(const std::string& s) {
  if (s.empty() || s.size() > 253) return false;
  size_t start = 0;
  while (true) {
    size_t dot = s.find('.', start);
    std::string lab = s.substr(start, dot == std::string::npos ? std::string::npos : dot - start);
    if (lab.empty()) return false;
    auto lower_alnum = [](char c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); };
    if (!lower_alnum(lab.front()) || !lower_alnum(lab.back())) return false;
    for (char c : lab)
      if (!lower_alnum(c) && c != '-') return false;
    if (dot == std::string::npos) break;
    start = dot + 1;
  }
  return true;
}
static bool is_qualified_name(const std::string& k) {
  size_t slash = k.find('/');
  if (slash == std::string::npos) return is_name_part(k);
  if (k.find('/', slash + 1) != std::string::npos) return false;
  return is_dns1123_subdomain(k.substr(0, slash)) && is_name_part(k.substr(slash + 1));
}
static bool is_valid_label_value(const std::string& v) { return v.empty() || is_name_part(v); }

// strconv.ParseInt(s, 10, 64)
static bool parse_int64(const std::string& s, int64_t* out) {
  size_t p = 0;
  bool neg = false;
  if (p < s.size() && (s[p] == '+' || s[p] == '-')) {
    neg = s[p] == '-';
    ++p;
  }
  if (p >= s.size()) return false;
  unsigned __int128 v = 0;
  for (; p < s.size(); ++p) {
    if (s[p] < '0' || s[p] > '9') return false;
    v = v * 10 + static_cast<unsigned>(s[p] - '0');
    if (v > (static_cast<unsigned __int128>(1) << 63)) return false;
  }
  if (!neg && v > static_cast<unsigned __int128>(std::numeric_limits<int64_t>::max())) return false;
  *out = neg ? static_cast<int64_t>(-static_cast<__int128>(v)) : static_cast<int64_t>(v);
  return true;
}

// labels.NewRequirement: returns false on a validation error.
static bool validate_label_requirement(const Requirement& r) {
  if (!is_qualified_name(r.key)) return false;
  const std::string& op = r.op;
  if (op == "In" || op == "NotIn") {
    if (r.values.empty()) return false;
  } else if (op == "Exists" || op == "DoesNotExist") {
    if (!r.values.empty()) return false;
  } else if (op == "Gt" || op == "Lt") {
    if (r.values.size() != 1) return false;
    int64_t tmp;
    if (!parse_int64(r.values[0], &tmp)) return false;
  } else {
    return false;  // operator not supported
  }
  for (auto& v : r.values)
    if (!is_valid_label_value(v)) return false;
  return true;
}

// labels.Requirement.Matches
static bool label_requirement_matches(const Requirement& r, const StrMap& labels) {
  auto it = labels.find(r.key);
  bool has = it != labels.end();
  auto has_value = [&](const std::string& v) { return std::find(r.values.begin(), r.values.end(), v) != r.values.end(); };
  if (r.op == "In") return has && has_value(it->second);
  if (r.op == "NotIn") return !has || !has_value(it->second);
  if (r.op == "Exists") return has;
  if (r.op == "DoesNotExist") return !has;
  if (r.op == "Gt" || r.op == "Lt") {
    if (!has) return false;
    int64_t lv, rv;
    if (!parse_int64(it->second, &lv)) return false;
    if (r.values.size() != 1 || !parse_int64(r.values[0], &rv)) return false;
    return r.op == "Gt" ? lv > rv : lv < rv;
  }
  return false;
}

// component-helpers nodeaffinity: nodeSelectorTerm.match (labels AND fields; a term with a parse error never matches).
static bool term_matches(const Term& t, const Node& node) {
  // newNodeSelectorTerm: parse both halves, collect errors.
  for (auto& e : t.exprs)
    if (!validate_label_requirement(e)) return false;
  for (auto& f : t.fields) {
    if (f.op != "In" && f.op != "NotIn") return false;
    if (f.values.size() != 1) return false;
  }
  for (auto& e : t.exprs)
    if (!label_requirement_matches(e, node.labels)) return false;
  // `t.matchFields != nil && len(nodeFields) > 0`: extractNodeFields only yields metadata.name when the
  // node has a non-empty name; with no fields the matchFields half is not consulted.
  if (!t.fields.empty() && !node.name.empty()) {
    for (auto& f : t.fields) {
      std::string fv = f.key == "metadata.name" ? node.name : std::string();
      bool eq = fv == f.values[0];
      if (f.op == "In" ? !eq : eq) return false;
    }
  }
  return true;
}

// nodeaffinity.GetRequiredNodeAffinity(pod).Match(node)
static bool required_node_affinity_matches(const Pod& p, const Node& node) {
  if (!p.node_selector.empty()) {  // labels.SelectorFromSet — plain equality, no validation
    for (auto& kv : p.node_selector) {
      auto it = node.labels.find(kv.first);
      if (it == node.labels.end() || it->second != kv.second) return false;
    }
  }
  if (p.has_required) {
    for (auto& t : p.terms) {
      if (t.exprs.empty() && t.fields.empty()) continue;  // isEmptyNodeSelectorTerm: selects no objects
      if (term_matches(t, node)) return true;
    }
    return false;  // nil / empty term list matches nothing (pins :550-607)
  }
  return true;
}

// v1.Toleration.ToleratesTaint
static bool tolerates(const Toleration& t, const Taint& taint) {
  if (!t.effect.empty() && t.effect != taint.effect) return false;
  if (!t.key.empty() && t.key != taint.key) return false;
  if (t.op.empty() || t.op == "Equal") return t.value == taint.value;
  if (t.op == "Exists") return true;
  return false;
}
static bool tolerations_tolerate(const std::vector<Toleration>& tols, const Taint& taint) {
  for (auto& t : tols)
    if (tolerates(t, taint)) return true;
  return false;
}
// v1helper.FindMatchingUntoleratedTaint with helper.DoNotScheduleTaintsFilterFunc (NoSchedule | NoExecute only).
static const Taint* find_untolerated_taint(const Node& n, const Pod& p) {
  for (auto& t : n.taints) {
    if (t.effect != "NoSchedule" && t.effect != "NoExecute") continue;
    if (!tolerations_tolerate(p.tolerations, t)) return &t;
  }
  return nullptr;
}

// metav1.LabelSelectorAsSelector(...).Matches(labels); `*err` set when the selector cannot be built.
static bool selector_matches(const LabelSelector& s, const StrMap& labels, bool* err) {
  if (!s.present) return false;  // labels.Nothing()
  for (auto& kv : s.match_labels) {
    if (!is_qualified_name(kv.first) || !is_valid_label_value(kv.second)) {
      *err = true;
      return false;
    }
    auto it = labels.find(kv.first);
    if (it == labels.end() || it->second != kv.second) return false;
  }
  for (auto& r : s.match_exprs) {
    if ((r.op != "In" && r.op != "NotIn" && r.op != "Exists" && r.op != "DoesNotExist") || !validate_label_requirement(r)) {
      *err = true;
      return false;
    }
    if (!label_requirement_matches(r, labels)) return false;
  }
  return true;
}
static bool selector_is_empty(const LabelSelector& s) {  // labels.Selector.Empty(): Everything() only
  return s.present && s.match_labels.empty() && s.match_exprs.empty();
}

// ---------------------------------------------------------------------------------------------------
// Snapshot
// ---------------------------------------------------------------------------------------------------
struct Snapshot {
  std::vector<size_t> nodes_with_anti;      // NodeInfoLister.HavePodsWithRequiredAntiAffinityList (nodeinfo_lister.go)
  std::vector<std::unique_ptr<Pod>> owned;  // every pod object (pending + assigned)
  std::vector<const Pod*> pending;          // the asks
  std::vector<NodeInfo> nodes;
  std::string load_error;
};

static StrMap read_strmap(const oj::Node* v) {
  StrMap out;
  if (v && v->is_obj())
    for (auto& kv : v->obj)
      if (kv.second->is_str() || kv.second->is_num()) out[kv.first] = kv.second->s;
  return out;
}
static std::vector<Requirement> read_requirements(const oj::Node* v) {
  std::vector<Requirement> out;
  if (v && v->is_arr())
    for (auto& e : v->arr) {
      Requirement r;
      r.key = e->str_or("key", "");
      r.op = e->str_or("operator", "");
      if (const oj::Node* vals = e->get_nn("values"))
        for (auto& x : vals->arr) r.values.push_back(x->s);
      out.push_back(std::move(r));
    }
  return out;
}
static std::vector<Container> read_containers(const oj::Node* v, bool init, std::string* err) {
  std::vector<Container> out;
  if (v && v->is_arr())
    for (auto& e : v->arr) {
      Container c;
      c.name = e->str_or("name", "");
      if (const oj::Node* res = e->get_nn("resources")) c.requests = read_strmap(res->get_nn("requests"));
      if (init) c.sidecar = e->str_or("restartPolicy", "") == "Always";
      if (const oj::Node* ports = e->get_nn("ports"))
        for (auto& pt : ports->arr) {
          int64_t hp = pt->int_or("hostPort", 0);
          if (hp <= 0) continue;  // "Only return ports with a host port specified"
          HostPort h;
          h.protocol = pt->str_or("protocol", "");
          h.ip = pt->str_or("hostIP", "");
          if (h.protocol.empty()) h.protocol = "TCP";
          if (h.ip.empty()) h.ip = "0.0.0.0";
          h.port = hp;
          c.host_ports.push_back(h);
        }
      (void)err;
      out.push_back(std::move(c));
    }
  return out;
}
static std::unique_ptr<Pod> read_pod(const oj::Node& v, std::string* err) {
  auto p = std::make_unique<Pod>();
  if (const oj::Node* md = v.get_nn("metadata")) {
    p->name = md->str_or("name", "");
    p->uid = md->str_or("uid", "");
    p->ns = md->str_or("namespace", "");
    p->labels = read_strmap(md->get_nn("labels"));
    p->terminating = md->get_nn("deletionTimestamp") != nullptr;
  }
  if (const oj::Node* st = v.get_nn("status")) {
    p->phase = st->str_or("phase", "");
    for (const char* field : {"containerStatuses", "initContainerStatuses"})  // resource.go:65-71: one map, init statuses last
      if (const oj::Node* css = st->get_nn(field))
        for (auto& cs : css->arr) {
          Pod::ContainerStatus out;
          out.allocated = read_strmap(cs->get_nn("allocatedResources"));
          if (const oj::Node* r = cs->get_nn("resources")) {
            out.has_resources = true;
            out.resources_requests = read_strmap(r->get_nn("requests"));
          }
          p->container_status[cs->str_or("name", "")] = std::move(out);
        }
    p->resize_infeasible = st->str_or("resize", "") == "Infeasible";  // v1.PodResizeStatusInfeasible
    if (const oj::Node* conds = st->get_nn("conditions"))
      for (auto& c : conds->arr)
        if (c->str_or("type", "") == "PodResizePending" && c->str_or("reason", "") == "Infeasible") p->resize_infeasible = true;
  }
  const oj::Node* spec = v.get_nn("spec");
  if (!spec) return p;
  p->node_name = spec->str_or("nodeName", "");
  if (const oj::Node* ns = spec->get_nn("nodeSelector")) {
    p->has_node_selector = true;
    p->node_selector = read_strmap(ns);
  }
  if (const oj::Node* aff = spec->get_nn("affinity")) {
    if (const oj::Node* na = aff->get_nn("nodeAffinity")) {
      if (const oj::Node* req = na->get_nn("requiredDuringSchedulingIgnoredDuringExecution")) {
        p->has_required = true;
        if (const oj::Node* terms = req->get_nn("nodeSelectorTerms"))
          for (auto& t : terms->arr) {
            Term term;
            term.exprs = read_requirements(t->get_nn("matchExpressions"));
            term.fields = read_requirements(t->get_nn("matchFields"));
            p->terms.push_back(std::move(term));
          }
      }
    }
    auto read_terms = [&](const oj::Node* pa, std::vector<PodAffinityTerm>* out) {
      if (!pa) return;
      const oj::Node* req = pa->get_nn("requiredDuringSchedulingIgnoredDuringExecution");
      if (!req || !req->is_arr()) return;
      for (auto& t : req->arr) {
        PodAffinityTerm term;
        if (const oj::Node* ls = t->get_nn("labelSelector")) {
          term.selector.present = true;
          term.selector.match_labels = read_strmap(ls->get_nn("matchLabels"));
          term.selector.match_exprs = read_requirements(ls->get_nn("matchExpressions"));
        }
        if (const oj::Node* nss = t->get_nn("namespaces"))
          for (auto& x : nss->arr) term.namespaces.push_back(x->s);
        term.topology_key = t->str_or("topologyKey", "");
        if (const oj::Node* nsel = t->get_nn("namespaceSelector")) {
          term.has_namespace_selector = true;
          const oj::Node* ml = nsel->get_nn("matchLabels");
          const oj::Node* me = nsel->get_nn("matchExpressions");
          if ((ml && !ml->obj.empty()) || (me && !me->arr.empty())) *err = "podAffinityTerm.namespaceSelector with requirements not modelled (needs Namespace labels)";
        }
        // matchLabelKeys / mismatchLabelKeys: the API server merges them into labelSelector when the pod is created; the
        // scheduler's framework.newAffinityTerm reads labelSelector, namespaces and namespaceSelector only
        out->push_back(std::move(term));
      }
    };
    read_terms(aff->get_nn("podAffinity"), &p->pod_affinity);
    read_terms(aff->get_nn("podAntiAffinity"), &p->pod_anti_affinity);
  }
  if (const oj::Node* tols = spec->get_nn("tolerations"))
    for (auto& t : tols->arr) {
      Toleration tol;
      tol.key = t->str_or("key", "");
      tol.op = t->str_or("operator", "");
      tol.value = t->str_or("value", "");
      tol.effect = t->str_or("effect", "");
      p->tolerations.push_back(std::move(tol));
    }
  p->containers = read_containers(spec->get_nn("containers"), false, err);
  p->init_containers = read_containers(spec->get_nn("initContainers"), true, err);
  if (const oj::Node* oh = spec->get_nn("overhead")) {
    p->has_overhead = true;
    p->overhead = read_strmap(oh);
  }
  if (const oj::Node* res = spec->get_nn("resources")) p->pod_level_requests = read_strmap(res->get_nn("requests"));
  if (const oj::Node* tsc = spec->get_nn("topologySpreadConstraints"))
    for (auto& c : tsc->arr) {
      SpreadConstraint sc;
      sc.max_skew = static_cast<int32_t>(c->int_or("maxSkew", 1));
      sc.topology_key = c->str_or("topologyKey", "");
      sc.when_unsatisfiable = c->str_or("whenUnsatisfiable", "DoNotSchedule");
      if (const oj::Node* ls = c->get_nn("labelSelector")) {
        sc.selector.present = true;
        sc.selector.match_labels = read_strmap(ls->get_nn("matchLabels"));
        sc.selector.match_exprs = read_requirements(ls->get_nn("matchExpressions"));
      }
      if (c->get_nn("minDomains")) {
        sc.has_min_domains = true;
        sc.min_domains = static_cast<int32_t>(c->int_or("minDomains", 1));
      }
      sc.node_affinity_policy = c->str_or("nodeAffinityPolicy", "Honor");
      sc.node_taints_policy = c->str_or("nodeTaintsPolicy", "Ignore");
      if (const oj::Node* mk = c->get_nn("matchLabelKeys"))
        for (auto& x : mk->arr) sc.match_label_keys.push_back(x->s);
      p->spread.push_back(std::move(sc));
    }
  return p;
}

static Snapshot* load_snapshot(const std::string& text) {
  auto snap = std::make_unique<Snapshot>();
  oj::NodePtr root = oj::parse(text);
  std::string err;
  size_t anon = 0;
  if (const oj::Node* nodes = root->get_nn("nodes")) {
    snap->nodes.reserve(nodes->arr.size());
    for (auto& nv : nodes->arr) {
      Node n;
      if (const oj::Node* md = nv->get_nn("metadata")) {
        n.name = md->str_or("name", "");
        n.labels = read_strmap(md->get_nn("labels"));
      }
      if (const oj::Node* spec = nv->get_nn("spec")) {
        n.unschedulable = spec->bool_or("unschedulable", false);
        if (const oj::Node* ts = spec->get_nn("taints"))
          for (auto& t : ts->arr) n.taints.push_back({t->str_or("key", ""), t->str_or("value", ""), t->str_or("effect", "")});
      }
      if (const oj::Node* st = nv->get_nn("status")) n.allocatable = read_strmap(st->get_nn("allocatable"));
      NodeInfo ni;
      ni.set_node(n);
      snap->nodes.push_back(std::move(ni));
      // NodeInfo.Pods: "pods" lists pod objects; an entry may carry "replicas": k (snapshot-format
      // extension) meaning k identical pods with distinct UIDs.
      if (const oj::Node* pods = nv->get_nn("pods"))
        for (auto& pv : pods->arr) {
          int64_t reps = pv->int_or("replicas", 1);
          for (int64_t r = 0; r < reps; ++r) {
            auto p = read_pod(*pv, &err);
            if (p->uid.empty()) p->uid = "anon-" + std::to_string(anon++);
            if (reps > 1) p->uid += "#" + std::to_string(r);
            snap->nodes.back().add_pod(p.get());
            snap->owned.push_back(std::move(p));
          }
        }
    }
  }
  if (const oj::Node* pods = root->get_nn("pods"))
    for (auto& pv : pods->arr) {
      auto p = read_pod(*pv, &err);
      if (p->uid.empty()) p->uid = "ask-" + std::to_string(anon++);
      snap->pending.push_back(p.get());
      snap->owned.push_back(std::move(p));
    }
  for (size_t i = 0; i < snap->nodes.size(); ++i)
    for (const Pod* p : snap->nodes[i].pods)
      if (!p->pod_anti_affinity.empty()) {
        snap->nodes_with_anti.push_back(i);
        break;
      }
  snap->load_error = err;
  return snap.release();
}

// ---------------------------------------------------------------------------------------------------
// Per-call cycle state + plugins
// ---------------------------------------------------------------------------------------------------
struct Status {
  enum Code { Success, Error, Unschedulable, UnschedulableAndUnresolvable, Skip } code = Success;
  std::string msg;
  bool is_success() const { return code == Success; }
  bool is_skip() const { return code == Skip; }
  bool is_rejected() const { return code == Unschedulable || code == UnschedulableAndUnresolvable; }
};

struct FitState {  // noderesources preFilterState
  bool written = false;
  Resource req;
};
// podtopologyspread preFilterState, restated for k8s.io/kubernetes v1.36.1 (go.mod:46): since the 1.3x refactor the state is
// held PER CONSTRAINT (TpValueToMatchNum []map[string]int, CriticalPaths []*criticalPaths — slices indexed by the constraint),
// no longer in maps keyed by {topologyKey, value} as up to 1.2x, where two hard constraints on the SAME topologyKey with
// different selectors shared (and overwrote) one counter. API validation rejects that shape anyway ("duplicate
// {topologyKey, whenUnsatisfiable}", ValidateTopologySpreadConstraints), and the product's encoder refuses it with that
// reason; the oracle still evaluates it constraint by constraint (tests/test_oracle_golden.py pins the shape).
struct SpreadState {
  bool written = false;
  std::vector<const SpreadConstraint*> constraints;
  std::vector<LabelSelector> selectors;                    // [constraint] labelSelector with the matchLabelKeys folded in
  std::vector<std::map<std::string, int>> value_to_match;  // [constraint] TpValueToMatchNum
  std::vector<int> min_match;                              // [constraint] CriticalPaths[i][0].MatchNum
};
using TopologyPair = std::pair<std::string, std::string>;
struct InterPodState {  // interpodaffinity preFilterState
  bool written = false;
  std::map<TopologyPair, int64_t> existing_anti, affinity, anti;  // existingAntiAffinityCounts, affinityCounts, antiAffinityCounts
};
struct CycleState {
  InterPodState ipa;
  bool affinity_written = false;
  bool ports_written = false;
  std::vector<HostPort> want_ports;
  FitState fit;
  SpreadState spread;
};

struct PreFilterResult {
  bool all_nodes = true;
  std::set<std::string> names;
  void merge(const PreFilterResult& o) {  // fwk.PreFilterResult.Merge
    if (o.all_nodes) return;
    if (all_nodes) {
      all_nodes = false;
      names = o.names;
      return;
    }
    std::set<std::string> inter;
    for (auto& n : names)
      if (o.names.count(n)) inter.insert(n);
    names.swap(inter);
  }
};

// --- NodeAffinity.PreFilter -------------------------------------------------------------------------
static Status nodeaffinity_prefilter(const Pod& p, CycleState& st, PreFilterResult* out) {
  bool no_affinity = !p.has_required;
  if (no_affinity && !p.has_node_selector) return {Status::Skip, ""};
  st.affinity_written = true;
  if (no_affinity || p.terms.empty()) return {};
  std::set<std::string> names;
  bool names_set = false;
  for (auto& t : p.terms) {
    bool term_set = false;
    std::set<std::string> term_names;
    for (auto& r : t.fields) {
      if (r.key == "metadata.name" && r.op == "In") {
        std::set<std::string> s(r.values.begin(), r.values.end());
        if (!term_set) {
          term_names = s;
          term_set = true;
        } else {
          std::set<std::string> inter;
          for (auto& n : term_names)
            if (s.count(n)) inter.insert(n);
          term_names.swap(inter);
        }
      }
    }
    if (!term_set) return {};  // a term without node-name affinity ⇒ all nodes eligible (terms are ORed)
    names.insert(term_names.begin(), term_names.end());
    names_set = true;
  }
  if (names_set && names.empty())
    return {Status::UnschedulableAndUnresolvable, "node(s) didn't match Pod's node affinity/selector"};  // errReasonConflict
  if (!names.empty()) {
    out->all_nodes = false;
    out->names = names;
  }
  return {};
}
static Status nodeaffinity_filter(const Pod& p, const NodeInfo& ni) {
  if (!required_node_affinity_matches(p, ni.node))
    return {Status::UnschedulableAndUnresolvable, "node(s) didn't match Pod's node affinity/selector"};
  return {};
}

// --- NodePorts (pins: predicate_manager_test.go:224-335, 1147-1157) -----------------------------------
static Status nodeports_prefilter(const Pod& p, CycleState& st) {
  std::vector<HostPort> want = get_host_ports(p);
  if (want.empty()) return {Status::Skip, ""};
  st.ports_written = true;
  st.want_ports = want;
  return {};
}
static Status nodeports_filter(const Pod& p, const CycleState& st, const NodeInfo& ni) {
  // Filter reads the PreFilter state; without it (PreFilter plugin disabled) the plugin returns an Error status
  if (!st.ports_written) return {Status::Error, "reading \"PreFilterNodePorts\" from cycleState: not found"};
  for (auto& w : st.want_ports)
    if (ports_conflict(w, ni.used_ports)) return {Status::Unschedulable, "node(s) didn't have free ports for the requested pod ports"};
  return {};
}

// --- NodeResourcesFit -------------------------------------------------------------------------------
static Status fit_prefilter(const Pod& p, CycleState& st) {
  st.fit.written = true;
  st.fit.req = to_resource(pod_requests(p));
  return {};
}
static Status fit_filter(const CycleState& st, const NodeInfo& ni) {
  if (!st.fit.written) return {Status::Error, "reading \"PreFilterNodeResourcesFit\" from cycleState: not found"};
  const Resource& rq = st.fit.req;
  std::string why;
  auto add = [&](const std::string& r) {
    if (!why.empty()) why += ", ";
    why += r;
  };
  if (static_cast<int64_t>(ni.pods.size()) + 1 > ni.allocatable.allowed_pods) add("Too many pods");
  bool empty_req = rq.milli_cpu == 0 && rq.memory == 0 && rq.ephemeral == 0 && rq.scalar.empty();
  if (!empty_req) {
    if (rq.milli_cpu > 0 && rq.milli_cpu > ni.allocatable.milli_cpu - ni.requested.milli_cpu) add("Insufficient cpu");
    if (rq.memory > 0 && rq.memory > ni.allocatable.memory - ni.requested.memory) add("Insufficient memory");
    if (rq.ephemeral > 0 && rq.ephemeral > ni.allocatable.ephemeral - ni.requested.ephemeral) add("Insufficient ephemeral-storage");
    for (auto& kv : rq.scalar) {
      if (kv.second == 0) continue;
      auto a = ni.allocatable.scalar.find(kv.first);
      auto u = ni.requested.scalar.find(kv.first);
      int64_t alloc = a == ni.allocatable.scalar.end() ? 0 : a->second;
      int64_t used = u == ni.requested.scalar.end() ? 0 : u->second;
      if (kv.second > alloc - used) add("Insufficient " + kv.first);
    }
  }
  if (!why.empty()) return {Status::Unschedulable, why};
  return {};
}

// --- PodTopologySpread ------------------------------------------------------------------------------
static Status spread_prefilter(const Pod& p, const std::vector<NodeInfo>& all, CycleState& st) {
  SpreadState& s = st.spread;
  for (auto& c : p.spread)
    if (c.when_unsatisfiable == "DoNotSchedule") s.constraints.push_back(&c);
  // no hard constraints (system defaults are ScheduleAnyway) ⇒ Skip
  if (s.constraints.empty()) return {Status::Skip, ""};
  // filterTopologySpreadConstraints: matchLabelKeys are folded into the selector — for every listed key the incoming pod
  // carries, "key = the pod's value" is ANDed on (mergeLabelSetWithSelector; labels.Nothing() of a nil selector stays Nothing)
  s.selectors.clear();
  for (auto* c : s.constraints) {
    LabelSelector merged = c->selector;
    if (merged.present)
      for (const std::string& key : c->match_label_keys) {
        auto own = p.labels.find(key);
        if (own == p.labels.end()) continue;
        Requirement eq;  // labels.SelectorFromSet: one equality requirement per key, ANDed with whatever the selector already asks of it
        eq.key = key;
        eq.op = "In";
        eq.values.push_back(own->second);
        merged.match_exprs.push_back(std::move(eq));
      }
    s.selectors.push_back(std::move(merged));
  }
  bool sel_err = false;
  for (auto& sel : s.selectors) {
    bool e = false;
    selector_matches(sel, p.labels, &e);
    sel_err |= e;
  }
  if (sel_err) return {Status::Error, "invalid label selector in topologySpreadConstraints"};
  s.written = true;
  s.value_to_match.assign(s.constraints.size(), {});
  s.min_match.assign(s.constraints.size(), 0);
  for (auto& ni : all) {
    const Node& node = ni.node;
    bool has_all = true;
    for (auto* c : s.constraints)
      if (!node.labels.count(c->topology_key)) has_all = false;
    if (!has_all) continue;  // nodeLabelsMatchSpreadConstraints
    for (size_t i = 0; i < s.constraints.size(); ++i) {
      const SpreadConstraint* c = s.constraints[i];
      // matchNodeInclusionPolicies
      if (c->node_affinity_policy == "Honor" && !required_node_affinity_matches(p, node)) continue;
      if (c->node_taints_policy == "Honor" && find_untolerated_taint(node, p)) continue;
      int count = 0;
      if (!selector_is_empty(s.selectors[i])) {  // countPodsMatchSelector
        for (const Pod* ep : ni.pods) {
          if (ep->terminating || ep->ns != p.ns) continue;
          bool e = false;
          if (selector_matches(s.selectors[i], ep->labels, &e)) ++count;
        }
      }
      s.value_to_match[i][node.labels.at(c->topology_key)] += count;
    }
  }
  for (size_t i = 0; i < s.constraints.size(); ++i) {
    int m = std::numeric_limits<int32_t>::max();
    for (auto& kv : s.value_to_match[i]) m = std::min(m, kv.second);
    s.min_match[i] = m;
  }
  return {};
}
static Status spread_filter(const Pod& p, const CycleState& st, const NodeInfo& ni) {
  const SpreadState& s = st.spread;
  if (!s.written) return {Status::Error, "reading \"PreFilterPodTopologySpread\" from cycleState: not found"};
  for (size_t i = 0; i < s.constraints.size(); ++i) {
    const SpreadConstraint* c = s.constraints[i];
    auto lit = ni.node.labels.find(c->topology_key);
    if (lit == ni.node.labels.end())
      return {Status::UnschedulableAndUnresolvable, "node(s) didn't match pod topology spread constraints (missing required label)"};
    int64_t min_match = s.min_match[i];
    const int domains = static_cast<int>(s.value_to_match[i].size());
    if (domains < (c->has_min_domains ? c->min_domains : 1)) min_match = 0;
    bool e = false;
    int64_t self = selector_matches(s.selectors[i], p.labels, &e) ? 1 : 0;
    auto mit = s.value_to_match[i].find(lit->second);
    int64_t match = mit == s.value_to_match[i].end() ? 0 : mit->second;
    int64_t skew = match + self - min_match;
    if (skew > c->max_skew) return {Status::Unschedulable, "node(s) didn't match pod topology spread constraints"};
  }
  return {};
}

// --- InterPodAffinity (pins: predicate_manager_test.go:1171-2113) --------------------------------------
// framework.AffinityTerm.Matches(pod, nil): the term's namespaces (default: the OWNING pod's namespace) contain the
// pod's namespace and the selector matches its labels.
static bool affinity_term_matches(const PodAffinityTerm& t, const std::string& owner_ns, const Pod& target) {
  // newAffinityTerm: no namespaces and a nil namespaceSelector = the owner's namespace; Matches: the namespace is listed OR the
  // namespaceSelector (here only ever Everything) selects it
  bool ns_ok = t.has_namespace_selector ||
               (t.namespaces.empty() ? target.ns == owner_ns : std::find(t.namespaces.begin(), t.namespaces.end(), target.ns) != t.namespaces.end());
  if (!ns_ok) return false;
  bool e = false;
  return selector_matches(t.selector, target.labels, &e);
}
static bool pod_matches_all_affinity_terms(const std::vector<PodAffinityTerm>& terms, const std::string& owner_ns, const Pod& target) {
  if (terms.empty()) return false;
  for (auto& t : terms)
    if (!affinity_term_matches(t, owner_ns, target)) return false;
  return true;
}
static void topo_update(std::map<TopologyPair, int64_t>& m, const Node& node, const std::string& key, int64_t v) {
  auto it = node.labels.find(key);
  if (it != node.labels.end()) m[{key, it->second}] += v;
}
static Status interpod_prefilter(const Pod& p, const std::vector<NodeInfo>& all, const std::vector<size_t>& nodes_with_anti,
                                 CycleState& st) {
  InterPodState& s = st.ipa;
  for (auto* terms : {&p.pod_affinity, &p.pod_anti_affinity})
    for (auto& t : *terms) {
      bool e = false;
      selector_matches(t.selector, p.labels, &e);
      if (e) return {Status::UnschedulableAndUnresolvable, "parsing pod: invalid label selector in pod (anti)affinity term"};
    }
  // getExistingAntiAffinityCounts: existing pods whose required anti-affinity terms match the incoming pod; upstream
  // walks only the nodes that hold such pods (HavePodsWithRequiredAntiAffinityList)
  for (size_t i : nodes_with_anti)
    for (const Pod* ep : all[i].pods)
      for (auto& t : ep->pod_anti_affinity)
        if (affinity_term_matches(t, ep->ns, p)) topo_update(s.existing_anti, all[i].node, t.topology_key, 1);
  // getIncomingAffinityAntiAffinityCounts (returns at once for a pod without required terms)
  if (!p.pod_affinity.empty() || !p.pod_anti_affinity.empty())
    for (auto& ni : all)
      for (const Pod* ep : ni.pods) {
      if (pod_matches_all_affinity_terms(p.pod_affinity, p.ns, *ep))
        for (auto& t : p.pod_affinity) topo_update(s.affinity, ni.node, t.topology_key, 1);
      for (auto& t : p.pod_anti_affinity)
        if (affinity_term_matches(t, p.ns, *ep)) topo_update(s.anti, ni.node, t.topology_key, 1);
    }
  if (s.existing_anti.empty() && p.pod_affinity.empty() && p.pod_anti_affinity.empty()) return {Status::Skip, ""};
  s.written = true;
  return {};
}
static Status interpod_filter(const Pod& p, const CycleState& st, const NodeInfo& ni) {
  const InterPodState& s = st.ipa;
  if (!s.written) return {Status::Error, "reading \"PreFilterInterPodAffinity\" from cycleState: not found"};
  const Node& node = ni.node;
  // satisfyPodAffinity
  bool pods_exist = true;
  for (auto& t : p.pod_affinity) {
    auto it = node.labels.find(t.topology_key);
    if (it == node.labels.end())
      return {Status::UnschedulableAndUnresolvable, "node(s) didn't match pod affinity rules"};  // all topology labels must exist
    auto c = s.affinity.find({t.topology_key, it->second});
    if (c == s.affinity.end() || c->second <= 0) pods_exist = false;
  }
  if (!pods_exist && !(s.affinity.empty() && pod_matches_all_affinity_terms(p.pod_affinity, p.ns, p)))
    return {Status::UnschedulableAndUnresolvable, "node(s) didn't match pod affinity rules"};
  // satisfyPodAntiAffinity
  for (auto& t : p.pod_anti_affinity) {
    auto it = node.labels.find(t.topology_key);
    if (it == node.labels.end()) continue;
    auto c = s.anti.find({t.topology_key, it->second});
    if (c != s.anti.end() && c->second > 0) return {Status::Unschedulable, "node(s) didn't match pod anti-affinity rules"};
  }
  // satisfyExistingPodsAntiAffinity
  for (auto& kv : node.labels) {
    auto c = s.existing_anti.find({kv.first, kv.second});
    if (c != s.existing_anti.end() && c->second > 0)
      return {Status::Unschedulable, "node(s) didn't satisfy existing pods anti-affinity rules"};
  }
  return {};
}

// --- the three stateless filters --------------------------------------------------------------------
static Status nodeunschedulable_filter(const Pod& p, const NodeInfo& ni) {
  if (!ni.node.unschedulable) return {};
  Taint t{"node.kubernetes.io/unschedulable", "", "NoSchedule"};
  if (!tolerations_tolerate(p.tolerations, t)) return {Status::UnschedulableAndUnresolvable, "node(s) were unschedulable"};
  return {};
}
static Status nodename_filter(const Pod& p, const NodeInfo& ni) {
  if (!p.node_name.empty() && p.node_name != ni.node.name)
    return {Status::UnschedulableAndUnresolvable, "node(s) didn't match the requested node name"};
  return {};
}
static Status tainttoleration_filter(const Pod& p, const NodeInfo& ni) {
  if (const Taint* t = find_untolerated_taint(ni.node, p))
    return {Status::UnschedulableAndUnresolvable, "node(s) had untolerated taint {" + t->key + ": " + t->value + "}"};
  return {};
}

// ---------------------------------------------------------------------------------------------------
// Driver: predicate_manager.go:206-287
// ---------------------------------------------------------------------------------------------------
struct Outcome {
  bool fit = true;
  int plugin = kCodeNone;
  std::string msg;
};

// runPreFilterPlugins (:221-254). Returns false when a PreFilter failed / excluded the node.
static bool run_prefilters(const Snapshot& snap, const Pod& p, const NodeInfo& target, uint32_t pre_mask, CycleState& st,
                           uint32_t* skip, Outcome* out) {
  PreFilterResult merged;
  struct Pre {
    uint32_t bit;
    int code;
  };
  // PreFilter-implementing plugins of this path, in MultiPoint order (comment block :307-318).
  static const Pre order[] = {{kNodeAffinity, kCodeNodeAffinity},
                              {kNodePorts, kCodeNodePorts},
                              {kNodeResourcesFit, kCodeNodeResourcesFit},
                              {kPodTopologySpread, kCodePodTopologySpread},
                              {kInterPodAffinity, kCodeInterPodAffinity}};
  for (const Pre& pl : order) {
    if (!(pre_mask & pl.bit)) continue;
    PreFilterResult res;
    Status s;
    if (pl.bit == kNodeAffinity)
      s = nodeaffinity_prefilter(p, st, &res);
    else if (pl.bit == kNodePorts)
      s = nodeports_prefilter(p, st);
    else if (pl.bit == kNodeResourcesFit)
      s = fit_prefilter(p, st);
    else if (pl.bit == kPodTopologySpread)
      s = spread_prefilter(p, snap.nodes, st);
    else
      s = interpod_prefilter(p, snap.nodes, snap.nodes_with_anti, st);
    if (s.is_skip()) {
      *skip |= pl.bit;  // :233-234
    } else if (!s.is_success()) {
      out->fit = false;
      out->msg = s.msg;
      out->plugin = s.is_rejected() ? kCodeNone : pl.code;  // :236-244
      return false;
    }
    merged.merge(res);  // :247
    if (!merged.all_nodes && !merged.names.count(target.node.name)) {
      out->fit = false;
      out->plugin = pl.code;
      out->msg = "node not eligible";  // :248-250
      return false;
    }
  }
  return true;
}

// runFilterPlugins (:260-283)
static void run_filters(const Pod& p, const NodeInfo& ni, uint32_t filt_mask, uint32_t skip, const CycleState& st, Outcome* out) {
  struct Filt {
    uint32_t bit;
    int code;
  };
  static const Filt order[] = {{kNodeUnschedulable, kCodeNodeUnschedulable}, {kNodeName, kCodeNodeName},
                               {kTaintToleration, kCodeTaintToleration},     {kNodeAffinity, kCodeNodeAffinity},
                               {kNodePorts, kCodeNodePorts},                 {kNodeResourcesFit, kCodeNodeResourcesFit},
                               {kPodTopologySpread, kCodePodTopologySpread}, {kInterPodAffinity, kCodeInterPodAffinity}};
  for (const Filt& pl : order) {
    if (!(filt_mask & pl.bit)) continue;
    if (skip & pl.bit) continue;  // :264-266
    Status s;
    switch (pl.bit) {
      case kNodeUnschedulable: s = nodeunschedulable_filter(p, ni); break;
      case kNodeName: s = nodename_filter(p, ni); break;
      case kTaintToleration: s = tainttoleration_filter(p, ni); break;
      case kNodeAffinity: s = nodeaffinity_filter(p, ni); break;
      case kNodePorts: s = nodeports_filter(p, st, ni); break;
      case kNodeResourcesFit: s = fit_filter(st, ni); break;
      case kPodTopologySpread: s = spread_filter(p, st, ni); break;
      case kInterPodAffinity: s = interpod_filter(p, st, ni); break;
      default: break;
    }
    if (!s.is_success()) {
      out->fit = false;
      out->plugin = pl.code;
      out->msg = s.is_rejected() ? s.msg : "running filter plugin for pod \"" + p.name + "\": " + s.msg;  // :269-277
      return;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// The same verdicts for ONE pod against MANY nodes with the PreFilter pass run once. Nothing a PreFilter plugin writes into
// the cycle state depends on the candidate node — the reference rebuilds identical state for every pair (:196,202,221-254)
// and only the membership test of the merged NodeNames result (:247-250) looks at the node. This form exists so that the
// parity tests can afford BASELINE configs[4] (100 000 nodes with hard spread constraints: the per-pair form costs O(N) node
// visits per PAIR); tests/test_oracle_golden.py holds it equal to pod_fits_node, pair by pair, on random clusters.
// ---------------------------------------------------------------------------------------------------
struct PreFilterReplay {
  CycleState st;
  uint32_t skip = 0;
  struct Step {
    bool failed = false;        // the plugin returned a non-Skip failure: every node that got this far fails with `fail`
    Outcome fail;
    PreFilterResult merged;     // result merged up to and including this plugin
    int code = kCodeNone;
  };
  std::vector<Step> steps;
};
static PreFilterReplay prefilter_once(const Snapshot& snap, const Pod& p, uint32_t pre_mask) {
  PreFilterReplay r;
  PreFilterResult merged;
  struct Pre {
    uint32_t bit;
    int code;
  };
  static const Pre order[] = {{kNodeAffinity, kCodeNodeAffinity},
                              {kNodePorts, kCodeNodePorts},
                              {kNodeResourcesFit, kCodeNodeResourcesFit},
                              {kPodTopologySpread, kCodePodTopologySpread},
                              {kInterPodAffinity, kCodeInterPodAffinity}};
  for (const Pre& pl : order) {
    if (!(pre_mask & pl.bit)) continue;
    PreFilterResult res;
    Status s;
    if (pl.bit == kNodeAffinity)
      s = nodeaffinity_prefilter(p, r.st, &res);
    else if (pl.bit == kNodePorts)
      s = nodeports_prefilter(p, r.st);
    else if (pl.bit == kNodeResourcesFit)
      s = fit_prefilter(p, r.st);
    else if (pl.bit == kPodTopologySpread)
      s = spread_prefilter(p, snap.nodes, r.st);
    else
      s = interpod_prefilter(p, snap.nodes, snap.nodes_with_anti, r.st);
    PreFilterReplay::Step step;
    step.code = pl.code;
    if (s.is_skip()) {
      r.skip |= pl.bit;
    } else if (!s.is_success()) {
      step.failed = true;
      step.fail.fit = false;
      step.fail.msg = s.msg;
      step.fail.plugin = s.is_rejected() ? kCodeNone : pl.code;
      r.steps.push_back(std::move(step));
      break;  // (:236-244 return here: later plugins never run)
    }
    merged.merge(res);
    step.merged = merged;
    r.steps.push_back(std::move(step));
  }
  return r;
}
static Outcome pod_fits_node_replayed(const PreFilterReplay& r, const Pod& p, const NodeInfo& ni, uint32_t filt_mask) {
  for (const PreFilterReplay::Step& step : r.steps) {
    if (step.failed) return step.fail;
    if (!step.merged.all_nodes && !step.merged.names.count(ni.node.name)) {
      Outcome out;
      out.fit = false;
      out.plugin = step.code;
      out.msg = "node not eligible";
      return out;
    }
  }
  Outcome out;
  run_filters(p, ni, filt_mask, r.skip, r.st, &out);
  return out;
}

// podFitsNode (:206-219) with a NEW CycleState per call (:196,202).
static Outcome pod_fits_node(const Snapshot& snap, const Pod& p, const NodeInfo& ni, uint32_t pre_mask, uint32_t filt_mask) {
  Outcome out;
  CycleState st;
  uint32_t skip = 0;
  if (!run_prefilters(snap, p, ni, pre_mask, st, &skip, &out)) return out;
  run_filters(p, ni, filt_mask, skip, st, &out);
  return out;
}

// PreemptionPredicates (:141-179). `victims[i] == nullptr` models a nil pod (:182-184).
static int preemption_predicates(const Snapshot& snap, const Pod& p, const NodeInfo& ni, const std::vector<const Pod*>& victims,
                                 int start, uint32_t pre_mask, uint32_t filt_mask) {
  Outcome out;
  CycleState st;
  uint32_t skip = 0;
  if (!run_prefilters(snap, p, ni, pre_mask, st, &skip, &out)) return -1;  // :146-155
  NodeInfo clone = ni;                                                      // node.Snapshot() :158
  int n = static_cast<int>(victims.size());
  for (int i = 0; i < start && i < n; ++i)
    if (victims[static_cast<size_t>(i)]) clone.remove_pod(victims[static_cast<size_t>(i)]);  // :161-163
  for (int i = start; i < n; ++i) {                                                           // :166-172
    if (victims[static_cast<size_t>(i)]) clone.remove_pod(victims[static_cast<size_t>(i)]);
    Outcome o;
    run_filters(p, clone, filt_mask, skip, st, &o);
    if (o.fit) return i;
  }
  return -1;
}

// Bin-pack node score. NOT in the reference (framework_handle.go:131-134 is a fatal stub); restates the
// recollection of yunikorn-core's `binpacking` node sorting policy from SURVEY.md Appendix A.9:
//   score = 1 - ( Σ_{r∈{vcore,memory}, total_r>0} (1 - available_r/total_r) ) / (number of such r)   [float64]
// nodes are tried in ascending score, ties by node index. PARITY UNPINNED. The exact operation order below
// is the contract the device kernel matches bit-for-bit.
static double binpack_score(const NodeInfo& ni) {
  const int64_t total[2] = {ni.allocatable.milli_cpu, ni.allocatable.memory};
  const int64_t used[2] = {ni.requested.milli_cpu, ni.requested.memory};
  double sum = 0.0, wsum = 0.0;
  for (int r = 0; r < 2; ++r) {
    if (total[r] <= 0) continue;
    double avail = static_cast<double>(total[r] - used[r]);
    double share = 1.0 - avail / static_cast<double>(total[r]);
    sum = sum + share;
    wsum = wsum + 1.0;
  }
  if (wsum == 0.0) return 1.0;
  return 1.0 - sum / wsum;
}

}  // namespace orc

// ---------------------------------------------------------------------------------------------------
// C API (ctypes)
// ---------------------------------------------------------------------------------------------------
using orc::Snapshot;

static void copy_str(const std::string& s, char* out, int len) {
  if (!out || len <= 0) return;
  size_t n = std::min(static_cast<size_t>(len - 1), s.size());
  std::memcpy(out, s.data(), n);
  out[n] = 0;
}

extern "C" {

void* orc_load(const char* json, char* err, int errlen) {
  try {
    Snapshot* s = orc::load_snapshot(json);
    if (!s->load_error.empty()) {
      copy_str(s->load_error, err, errlen);
      delete s;
      return nullptr;
    }
    return s;
  } catch (const std::exception& e) {
    copy_str(e.what(), err, errlen);
    return nullptr;
  }
}
void orc_free(void* h) { delete static_cast<Snapshot*>(h); }
int orc_num_nodes(void* h) { return static_cast<int>(static_cast<Snapshot*>(h)->nodes.size()); }
int orc_num_pods(void* h) { return static_cast<int>(static_cast<Snapshot*>(h)->pending.size()); }

// One Predicates() call. Returns 1 = fits ("" , nil), 0 = does not fit; *plugin = failing plugin code.
int orc_predicates(void* h, int pod, int node, unsigned pre_mask, unsigned filt_mask, int* plugin, char* msg, int msglen) {
  Snapshot* s = static_cast<Snapshot*>(h);
  if (pod < 0 || pod >= static_cast<int>(s->pending.size()) || node < 0 || node >= static_cast<int>(s->nodes.size())) return -1;
  orc::Outcome o = orc::pod_fits_node(*s, *s->pending[static_cast<size_t>(pod)], s->nodes[static_cast<size_t>(node)], pre_mask, filt_mask);
  if (plugin) *plugin = o.plugin;
  copy_str(o.msg, msg, msglen);
  return o.fit ? 1 : 0;
}

// pods[i] × nodes[j] → fit[i*nn+j] (0/1) and optionally plugin[i*nn+j]; per-pair calls, `threads` OpenMP threads.
int orc_eval_grid(void* h, const int* pods, int np, const int* nodes, int nn, unsigned pre_mask, unsigned filt_mask, uint8_t* fit,
                  uint8_t* plugin, int threads) {
  Snapshot* s = static_cast<Snapshot*>(h);
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int i = 0; i < np; ++i) {
    const orc::Pod& p = *s->pending[static_cast<size_t>(pods ? pods[i] : i)];
    for (int j = 0; j < nn; ++j) {
      const orc::NodeInfo& ni = s->nodes[static_cast<size_t>(nodes ? nodes[j] : j)];
      orc::Outcome o = orc::pod_fits_node(*s, p, ni, pre_mask, filt_mask);
      fit[static_cast<size_t>(i) * static_cast<size_t>(nn) + static_cast<size_t>(j)] = o.fit ? 1 : 0;
      if (plugin) plugin[static_cast<size_t>(i) * static_cast<size_t>(nn) + static_cast<size_t>(j)] = static_cast<uint8_t>(o.plugin);
    }
  }
  return 0;
}

// orc_eval_grid with the PreFilter pass of a pod run once for all its nodes (see prefilter_once)
int orc_eval_rows(void* h, const int* pods, int np, const int* nodes, int nn, unsigned pre_mask, unsigned filt_mask, uint8_t* fit,
                  uint8_t* plugin, int threads) {
  Snapshot* s = static_cast<Snapshot*>(h);
  if (threads < 1) threads = 1;
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads)
  for (int i = 0; i < np; ++i) {
    const orc::Pod& p = *s->pending[static_cast<size_t>(pods ? pods[i] : i)];
    const orc::PreFilterReplay replay = orc::prefilter_once(*s, p, pre_mask);
    for (int j = 0; j < nn; ++j) {
      const orc::NodeInfo& ni = s->nodes[static_cast<size_t>(nodes ? nodes[j] : j)];
      orc::Outcome o = orc::pod_fits_node_replayed(replay, p, ni, filt_mask);
      fit[static_cast<size_t>(i) * static_cast<size_t>(nn) + static_cast<size_t>(j)] = o.fit ? 1 : 0;
      if (plugin) plugin[static_cast<size_t>(i) * static_cast<size_t>(nn) + static_cast<size_t>(j)] = static_cast<uint8_t>(o.plugin);
    }
  }
  return 0;
}

// victims: indices into the node's pod list (order of the snapshot's "pods", replicas expanded); -1 = nil pod.
int orc_preemption(void* h, int pod, int node, const int* victims, int nv, int start, unsigned pre_mask, unsigned filt_mask) {
  Snapshot* s = static_cast<Snapshot*>(h);
  const orc::NodeInfo& ni = s->nodes[static_cast<size_t>(node)];
  std::vector<const orc::Pod*> v;
  for (int i = 0; i < nv; ++i) v.push_back(victims[i] < 0 ? nullptr : ni.pods[static_cast<size_t>(victims[i])]);
  return orc::preemption_predicates(*s, *s->pending[static_cast<size_t>(pod)], ni, v, start, pre_mask, filt_mask);
}

// Request vector of a pending pod as JSON {"cpu": milli, "memory": bytes, ...} (A.1 known-answer checks).
int orc_pod_request_json(void* h, int pod, char* out, int len) {
  Snapshot* s = static_cast<Snapshot*>(h);
  orc::ResMap m = orc::pod_requests(*s->pending[static_cast<size_t>(pod)]);
  std::string js = "{";
  bool first = true;
  for (auto& kv : m) {
    if (!first) js += ",";
    first = false;
    js += "\"" + kv.first + "\":" + std::to_string(kv.second);
  }
  js += "}";
  copy_str(js, out, len);
  return static_cast<int>(js.size());
}

// Node bookkeeping readback: out[0..5] = alloc cpu, alloc mem, alloc eph, allowed pods, pod count, (unused);
// out[6..8] = requested cpu, mem, eph.
int orc_node_info(void* h, int node, int64_t* out) {
  Snapshot* s = static_cast<Snapshot*>(h);
  const orc::NodeInfo& ni = s->nodes[static_cast<size_t>(node)];
  out[0] = ni.allocatable.milli_cpu;
  out[1] = ni.allocatable.memory;
  out[2] = ni.allocatable.ephemeral;
  out[3] = ni.allocatable.allowed_pods;
  out[4] = static_cast<int64_t>(ni.pods.size());
  out[5] = 0;
  out[6] = ni.requested.milli_cpu;
  out[7] = ni.requested.memory;
  out[8] = ni.requested.ephemeral;
  return 0;
}

int orc_binpack_scores(void* h, double* out) {
  Snapshot* s = static_cast<Snapshot*>(h);
  for (size_t i = 0; i < s->nodes.size(); ++i) out[i] = orc::binpack_score(s->nodes[i]);
  return 0;
}

// Snapshot decision for one pod: feasible count and the feasible node with the smallest (score, NodeID string) — the
// bin-packing order of yunikorn-core (recollection, A.9: nodes sorted by score, ties by node id; PARITY UNPINNED).
int orc_decide(void* h, int pod, unsigned pre_mask, unsigned filt_mask, int* count, int* best) {
  Snapshot* s = static_cast<Snapshot*>(h);
  const orc::Pod& p = *s->pending[static_cast<size_t>(pod)];
  int c = 0, b = -1;
  double bs = 0.0;
  for (size_t j = 0; j < s->nodes.size(); ++j) {
    if (!orc::pod_fits_node(*s, p, s->nodes[j], pre_mask, filt_mask).fit) continue;
    ++c;
    double sc = orc::binpack_score(s->nodes[j]);
    if (b < 0 || sc < bs || (sc == bs && s->nodes[j].node.name < s->nodes[static_cast<size_t>(b)].node.name)) {
      b = static_cast<int>(j);
      bs = sc;
    }
  }
  *count = c;
  *best = b;
  return 0;
}

// orc_decide with the pod's PreFilter pass run once for all nodes (see prefilter_once: same verdicts, O(N) instead of O(N^2)
// with hard spread constraints)
int orc_decide_once(void* h, int pod, unsigned pre_mask, unsigned filt_mask, int* count, int* best) {
  Snapshot* s = static_cast<Snapshot*>(h);
  const orc::Pod& p = *s->pending[static_cast<size_t>(pod)];
  const orc::PreFilterReplay replay = orc::prefilter_once(*s, p, pre_mask);
  int c = 0, b = -1;
  double bs = 0.0;
  for (size_t j = 0; j < s->nodes.size(); ++j) {
    if (!orc::pod_fits_node_replayed(replay, p, s->nodes[j], filt_mask).fit) continue;
    ++c;
    double sc = orc::binpack_score(s->nodes[j]);
    if (b < 0 || sc < bs || (sc == bs && s->nodes[j].node.name < s->nodes[static_cast<size_t>(b)].node.name)) {
      b = static_cast<int>(j);
      bs = sc;
    }
  }
  *count = c;
  *best = b;
  return 0;
}

// ---------------------------------------------------------------------------------------------------
// The SEQUENTIAL loop the reference actually runs: yunikorn-core decides one ask, the shim assumes it, and
// the next Predicates() call sees it. For ask pods[i], in order:
//   * the core walks the nodes in bin-pack order under the CURRENT state — ascending (score, NodeID), recollection of
//     yunikorn-core's `binpacking` policy, A.9 — and calls Predicates(ask, node) (scheduler_callback.go:203-205 →
//     context.go:696-716 → predicate_manager.go:134-139) until the first node that fits;
//   * the allocation comes back through AsyncRMCallback.UpdateAllocation (scheduler_callback.go:49-98) → Context.AssumePod
//     (context.go:828-885): assumedPod.Spec.NodeName = node, SchedulerCache.AssumePod (scheduler_cache.go:443-461) → updatePod
//     → NodeInfo.AddPod (:363): the node's Requested, pod list (labels for the topology plugins), used host ports grow.
// out[i] = node index or -1 (no node fits: the ask stays pending, nothing is assumed). The snapshot is MUTATED.
// The node order is kept in an ordered set keyed by (score, name) and only the allocated node is re-keyed — what the core's
// sorted node collection does; `early_exit` = 0 evaluates every node instead (the naive argmin form; same answers), 2 = early
// exit with the ask's PreFilter pass run once instead of once per candidate (same answers; what large clusters can afford).
int orc_allocate_sequential(void* h, const int* pods, int np, unsigned pre_mask, unsigned filt_mask, int* out, int early_exit) {
  Snapshot* s = static_cast<Snapshot*>(h);
  struct Key {
    double score;
    const std::string* name;
    size_t idx;
    bool operator<(const Key& o) const {
      if (score != o.score) return score < o.score;
      if (*name != *o.name) return *name < *o.name;
      return idx < o.idx;
    }
  };
  std::set<Key> order;
  std::vector<Key> key_of(s->nodes.size());
  for (size_t j = 0; j < s->nodes.size(); ++j) {
    key_of[j] = Key{orc::binpack_score(s->nodes[j]), &s->nodes[j].node.name, j};
    order.insert(key_of[j]);
  }
  std::set<size_t> anti(s->nodes_with_anti.begin(), s->nodes_with_anti.end());
  for (int i = 0; i < np; ++i) {
    const size_t pi = static_cast<size_t>(pods ? pods[i] : i);
    orc::Pod* p = const_cast<orc::Pod*>(s->pending[pi]);
    int best = -1;
    if (early_exit == 2) {
      // the PreFilter pass of the ask ONCE, its outcome replayed for every candidate node (orc_eval_rows' form, held equal to the
      // per-pair form by tests/test_oracle_golden.py and tests/test_oracle_sequential.py): nothing changes between the candidates
      // of one ask, and the per-pair form costs a walk over every pod of the cluster per CANDIDATE once an ask carries a topology
      // constraint — minutes per ask at 10^5 nodes
      const orc::PreFilterReplay replay = orc::prefilter_once(*s, *p, pre_mask);
      for (const Key& k : order)
        if (orc::pod_fits_node_replayed(replay, *p, s->nodes[k.idx], filt_mask).fit) {
          best = static_cast<int>(k.idx);
          break;
        }
    } else if (early_exit) {
      for (const Key& k : order)
        if (orc::pod_fits_node(*s, *p, s->nodes[k.idx], pre_mask, filt_mask).fit) {
          best = static_cast<int>(k.idx);
          break;
        }
    } else {
      const Key* bk = nullptr;
      for (size_t j = 0; j < s->nodes.size(); ++j)
        if (orc::pod_fits_node(*s, *p, s->nodes[j], pre_mask, filt_mask).fit && (!bk || key_of[j] < *bk)) bk = &key_of[j];
      if (bk) best = static_cast<int>(bk->idx);
    }
    out[i] = best;
    if (best < 0) continue;
    orc::NodeInfo& ni = s->nodes[static_cast<size_t>(best)];
    p->node_name = ni.node.name;  // context.go:879
    ni.add_pod(p);                // scheduler_cache.go:363
    if (!p->pod_anti_affinity.empty() && anti.insert(static_cast<size_t>(best)).second) s->nodes_with_anti.push_back(static_cast<size_t>(best));
    order.erase(key_of[static_cast<size_t>(best)]);
    key_of[static_cast<size_t>(best)].score = orc::binpack_score(ni);
    order.insert(key_of[static_cast<size_t>(best)]);
  }
  return 0;
}

int64_t orc_quantity_value(const char* s) { return orc::quantity_value(s); }
int64_t orc_quantity_milli(const char* s) { return orc::quantity_milli(s); }

}  // extern "C"
