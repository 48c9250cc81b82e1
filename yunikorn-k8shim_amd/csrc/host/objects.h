// objects.h — the slice of v1.Pod / v1.Node / framework.NodeInfo that the predicate path reads, as held by
// the host library (the C++ stand-in for the Go shim side; Go is not available in this image).
//
// Mirrors what /root/reference/pkg/cache/external/scheduler_cache.go keeps per node (NodeInfo: Allocatable,
// Requested, Pods, Node()) and per pod (*v1.Pod). Pods share an interned, immutable PodTemplate: everything
// of the pod except its identity (uid/name) and spec.nodeName, so a million replicas of a few thousand
// Deployments / gang task groups cost one template each (placeholder.go:113-157 builds exactly such clones).
#pragma once
#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <string>
#include <unordered_map>
#include <vector>

#include "minijson.h"
#include "quantity.h"

namespace ykh {

using StrMap = std::map<std::string, std::string>;
using ResMap = std::map<std::string, int64_t>;

struct Taint {
  std::string key, value, effect;
};
struct Toleration {
  std::string key, op, value, effect;
};
struct Requirement {
  std::string key, op;
  std::vector<std::string> values;
};
struct SelectorTerm {
  std::vector<Requirement> exprs, fields;
};
struct LabelSelector {
  bool present = false;
  StrMap match_labels;
  std::vector<Requirement> match_exprs;
};
struct SpreadConstraint {
  int32_t max_skew = 1;
  std::string topology_key, when_unsatisfiable = "DoNotSchedule";
  LabelSelector selector;
  bool has_min_domains = false;
  int32_t min_domains = 1;
  std::string node_affinity_policy = "Honor", node_taints_policy = "Ignore";
  std::vector<std::string> match_label_keys;
  // `selector` is the selector the plugin matches with: labelSelector with the matchLabelKeys folded in (for every listed key
  // the pod carries: key In [the pod's value] — podtopologyspread mergeLabelSetWithSelector, or the API server's merge of the
  // same requirement at admission; folding twice is idempotent). `raw_selector` is what the object said (snapshot dumps).
  LabelSelector raw_selector;
};
struct PodAffinityTerm {  // v1.PodAffinityTerm of a required (anti)affinity rule
  LabelSelector selector;
  std::vector<std::string> namespaces;  // empty = the owning pod's namespace
  std::string topology_key;
  // namespaceSelector: {} selects every namespace (evaluated); a non-empty one needs Namespace labels the mirror does not
  // hold (routed). matchLabelKeys / mismatchLabelKeys are merged into labelSelector by the API server at admission and never
  // read by the scheduler (framework.newAffinityTerm): ignored here as well.
  bool all_namespaces = false, namespace_selector_unsupported = false;
};
struct HostPort {  // v1.ContainerPort with hostPort > 0; "" hostIP = 0.0.0.0, "" protocol = TCP (HostPortInfo.sanitize)
  std::string protocol, ip;
  int64_t port = 0;
};
struct Container {
  std::string name;
  StrMap requests;
  bool sidecar = false;
  std::vector<HostPort> host_ports;
};

// Everything of a pod except uid / name / nodeName.
struct Resource {  // framework.Resource
  int64_t milli_cpu = 0, memory = 0, ephemeral = 0, allowed_pods = 0;
  std::map<std::string, int64_t> scalar;
};

struct PodTemplate {
  std::string ns;
  StrMap labels;
  bool has_node_selector = false;
  StrMap node_selector;
  bool has_required = false;
  std::vector<SelectorTerm> terms;
  std::vector<Toleration> tolerations;
  std::vector<Container> containers, init_containers;
  bool has_overhead = false;
  StrMap overhead;
  StrMap pod_level_requests;
  std::vector<SpreadConstraint> spread;
  std::vector<PodAffinityTerm> pod_affinity, pod_anti_affinity;  // requiredDuringSchedulingIgnoredDuringExecution
  bool pod_affinity_unsupported = false;  // some required (anti)affinity term carries a non-empty namespaceSelector
  // spec.volumes entries of a kind one of the Volume* Filters inspects (VolumeBinding, VolumeZone, VolumeRestrictions,
  // NodeVolumeLimits: everything except the node-local kinds below) and the number of spec.resourceClaims (DynamicResources).
  // Those plugins need PV / PVC / StorageClass / CSINode / ResourceSlice state the engine does not hold: such an ask is
  // marked unsupported and routed to the CPU predicate manager, never answered "fits" by omission.
  std::vector<std::string> volume_kinds;
  int32_t resource_claims = 0;
  std::string canonical;      // interning key (canonical JSON of the fields above)
  // What the encoder's DICTIONARIES can learn from this template: everything except its labels and the VALUES of its requests —
  // namespace, host ports, the names of the scalar resources it requests, and the spec behind the containers (selectors, affinity,
  // tolerations, spread constraints, volumes ...). Two templates of one shape register the same entries and are refused for the
  // same reasons, so a full encode visits one of them (10^6 asks that differ only in their cpu request are a few 10^4 shapes).
  // The pool numbers the shapes; the text is dropped after interning.
  std::string dict_shape;
  int32_t shape_id = -1;
  // derived once at interning time
  ResMap requests;            // upstream PodRequests (resource.go:56-109 minus the "pods" entry)
  Resource res;               // the same as a framework.Resource: what NodeInfo.AddPod / RemovePod add to and take from Requested
  int32_t spec_id = -1;       // engine spec index (assigned by the encoder)
  int32_t first_row = 0;      // scratch of a full encode: the first pending row that uses the template (spec ids follow first use)
};

struct Pod {
  std::string uid, name;
  std::string node_name;  // spec.nodeName ("" = pending ask)
  bool terminating = false;
  // SchedulerCache membership (scheduler_cache.go:57-62): podsMap = the mirror's uid index; the other three maps are
  // per-pod fields here.
  std::string assigned_node;  // assignedPods[uid]: the NodeInfo that holds (and accounts) this pod, "" = none
  bool assumed = false;       // assumedPods has uid (set by AssumePod, cleared by ForgetPod / Running / terminated)
  bool orphan = false;        // orphanedPods has uid: spec.nodeName names a node that is not in the cache
  bool ask = false;           // holds a row of the ask table (bitmap row); rows stay put across AssumePod / ForgetPod
  int32_t row = -1;           // that row
  int32_t remote_node = -1;   // assumed by a round of a node-sharded cluster on ANOTHER shard's node: its index in the whole cluster
  const PodTemplate* tpl = nullptr;
};

struct Node {
  std::string name;
  StrMap labels;
  std::vector<Taint> taints;
  bool unschedulable = false;
  StrMap allocatable;
};

inline bool has_prefix(const std::string& s, const char* p) { return s.rfind(p, 0) == 0; }
// schedutil.IsScalarResourceName
inline bool is_scalar_resource_name(const std::string& n) {
  bool has_slash = n.find('/') != std::string::npos;
  bool k8s = n.find("kubernetes.io/") != std::string::npos;
  bool extended = has_slash && !k8s && !has_prefix(n, "requests.");
  return extended || k8s || has_prefix(n, "hugepages-") || has_prefix(n, "attachable-volumes-");
}
inline ResMap get_resource(const StrMap& rl) {  // resource.go:273-285
  ResMap out;
  for (auto& kv : rl) out[kv.first] = kv.first == "cpu" ? quantity_milli(kv.second) : quantity_value(kv.second);
  return out;
}
inline Resource to_resource(const ResMap& m) {
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

// schedutil.GetHostPorts: containers + init containers with restartPolicy Always
inline std::vector<HostPort> template_host_ports(const PodTemplate& t);

// Request vector of a template: containers summed, init containers / native sidecars folded in, pod-level
// override for cpu/memory/hugepages, overhead added (resource.go:56-109,164-182,287-301).
inline ResMap compute_requests(const PodTemplate& t) {
  auto add = [](ResMap& l, const ResMap& r) {
    for (auto& kv : r) l[kv.first] += kv.second;
  };
  auto upmax = [](ResMap& l, const ResMap& r) {
    for (auto& kv : r) {
      auto it = l.find(kv.first);
      if (it == l.end())
        l[kv.first] = kv.second;
      else if (kv.second > it->second)
        it->second = kv.second;
    }
  };
  ResMap total;
  for (auto& c : t.containers) add(total, get_resource(c.requests));
  if (!t.init_containers.empty()) {
    ResMap init_max, sidecars;
    for (auto& c : t.init_containers) {
      ResMap ic = get_resource(c.requests);
      ResMap cur = ic;
      add(cur, sidecars);
      if (c.sidecar) add(sidecars, ic);
      upmax(init_max, cur);
    }
    add(total, sidecars);
    upmax(total, init_max);
  }
  for (auto& kv : get_resource(t.pod_level_requests))
    if (kv.first == "cpu" || kv.first == "memory" || has_prefix(kv.first, "hugepages-")) total[kv.first] = kv.second;
  if (t.has_overhead) add(total, get_resource(t.overhead));
  return total;
}

inline std::vector<HostPort> template_host_ports(const PodTemplate& t) {
  std::vector<HostPort> out;
  for (auto& c : t.init_containers)
    if (c.sidecar) out.insert(out.end(), c.host_ports.begin(), c.host_ports.end());
  for (auto& c : t.containers) out.insert(out.end(), c.host_ports.begin(), c.host_ports.end());
  return out;
}
// HostPortInfo.CheckConflict between one wanted and one used port
inline bool host_ports_conflict(const HostPort& want, const HostPort& used) {
  return want.port > 0 && want.protocol == used.protocol && want.port == used.port &&
         (want.ip == "0.0.0.0" || used.ip == "0.0.0.0" || want.ip == used.ip);
}

// framework.NodeInfo
struct NodeInfo {
  Node node;
  std::vector<const Pod*> pods;
  Resource requested, allocatable;
  int32_t index = -1;  // engine node index

  void set_node(const Node& n) {
    node = n;
    allocatable = to_resource(get_resource(n.allocatable));
  }
  void account(const Pod* p, int sign) {
    const Resource& r = p->tpl->res;  // (computed once per template: the map walk per pod was a fifth of a cache pass)
    requested.milli_cpu += sign * r.milli_cpu;
    requested.memory += sign * r.memory;
    requested.ephemeral += sign * r.ephemeral;
    for (auto& kv : r.scalar) requested.scalar[kv.first] += sign * kv.second;
  }
  void add_pod(const Pod* p) {
    pods.push_back(p);
    account(p, +1);
  }
  // `known`: the cached version of the pod, when the caller holds it — found by address before any uid string is compared
  bool remove_pod(const std::string& uid, const Pod* known = nullptr) {
    if (known)
      for (size_t i = 0; i < pods.size(); ++i)
        if (pods[i] == known) {
          account(pods[i], -1);
          pods.erase(pods.begin() + (long)i);
          return true;
        }
    for (size_t i = 0; i < pods.size(); ++i)
      if (pods[i]->uid == uid) {
        account(pods[i], -1);
        pods.erase(pods.begin() + (long)i);
        return true;
      }
    return false;
  }
};

// ---------------------------------------------------------------------------------------------------
// JSON → objects (Kubernetes field names)
// ---------------------------------------------------------------------------------------------------
inline StrMap read_strmap(const mj::Value* v) {
  StrMap out;
  if (v && v->is_obj())
    for (auto& kv : v->obj)
      if (kv.second->is_str() || kv.second->is_num()) out[kv.first] = kv.second->s;
  return out;
}
inline std::vector<Requirement> read_requirements(const mj::Value* v) {
  std::vector<Requirement> out;
  if (v && v->is_arr())
    for (auto& e : v->arr) {
      Requirement r;
      r.key = e->str_or("key", "");
      r.op = e->str_or("operator", "");
      if (const mj::Value* vals = e->get_nn("values"))
        for (auto& x : vals->arr) r.values.push_back(x->s);
      out.push_back(std::move(r));
    }
  return out;
}
inline std::vector<Container> read_containers(const mj::Value* v, bool init) {
  std::vector<Container> out;
  if (v && v->is_arr())
    for (auto& e : v->arr) {
      Container c;
      c.name = e->str_or("name", "");
      if (const mj::Value* res = e->get_nn("resources")) c.requests = read_strmap(res->get_nn("requests"));
      if (init) c.sidecar = e->str_or("restartPolicy", "") == "Always";
      if (const mj::Value* ports = e->get_nn("ports"))
        for (auto& pt : ports->arr) {
          int64_t hp = pt->int_or("hostPort", 0);
          if (hp <= 0) continue;
          HostPort h;
          h.protocol = pt->str_or("protocol", "");
          h.ip = pt->str_or("hostIP", "");
          if (h.protocol.empty()) h.protocol = "TCP";
          if (h.ip.empty()) h.ip = "0.0.0.0";
          h.port = hp;
          c.host_ports.push_back(h);
        }
      out.push_back(std::move(c));
    }
  return out;
}

// ---------------------------------------------------------------------------------------------------
// objects → JSON (used for interning keys and for ykhost_dump_snapshot)
// ---------------------------------------------------------------------------------------------------
inline void js_str(std::string& o, const std::string& s) {
  o.push_back('"');
  for (unsigned char c : s) {
    if (c == '"' || c == '\\') {
      o.push_back('\\');
      o.push_back((char)c);
    } else if (c < 0x20) {
      char buf[8];
      snprintf(buf, sizeof buf, "\\u%04x", c);
      o += buf;
    } else {
      o.push_back((char)c);
    }
  }
  o.push_back('"');
}
inline void js_map(std::string& o, const StrMap& m) {
  o.push_back('{');
  bool first = true;
  for (auto& kv : m) {
    if (!first) o.push_back(',');
    first = false;
    js_str(o, kv.first);
    o.push_back(':');
    js_str(o, kv.second);
  }
  o.push_back('}');
}
inline void js_reqs(std::string& o, const std::vector<Requirement>& rs) {
  o.push_back('[');
  for (size_t i = 0; i < rs.size(); ++i) {
    if (i) o.push_back(',');
    o += "{\"key\":";
    js_str(o, rs[i].key);
    o += ",\"operator\":";
    js_str(o, rs[i].op);
    if (!rs[i].values.empty()) {
      o += ",\"values\":[";
      for (size_t j = 0; j < rs[i].values.size(); ++j) {
        if (j) o.push_back(',');
        js_str(o, rs[i].values[j]);
      }
      o.push_back(']');
    }
    o.push_back('}');
  }
  o.push_back(']');
}
inline void js_containers(std::string& o, const std::vector<Container>& cs) {
  o.push_back('[');
  for (size_t i = 0; i < cs.size(); ++i) {
    if (i) o.push_back(',');
    o += "{\"name\":";
    js_str(o, cs[i].name);
    o += ",\"resources\":{\"requests\":";
    js_map(o, cs[i].requests);
    o += "}";
    if (cs[i].sidecar) o += ",\"restartPolicy\":\"Always\"";
    if (!cs[i].host_ports.empty()) {
      o += ",\"ports\":[";
      for (size_t k = 0; k < cs[i].host_ports.size(); ++k) {
        const HostPort& h = cs[i].host_ports[k];
        if (k) o.push_back(',');
        o += "{\"hostIP\":";
        js_str(o, h.ip);
        o += ",\"hostPort\":" + std::to_string(h.port) + ",\"protocol\":";
        js_str(o, h.protocol);
        o.push_back('}');
      }
      o.push_back(']');
    }
    o.push_back('}');
  }
  o.push_back(']');
}

// "metadata" (namespace + labels only) and "spec" members of a pod, without identity and nodeName.
inline void template_json(const PodTemplate& t, std::string& meta, std::string& spec, size_t* rest_at = nullptr) {
  meta.clear();
  spec.clear();
  meta += "\"namespace\":";
  js_str(meta, t.ns);
  meta += ",\"labels\":";
  js_map(meta, t.labels);
  spec += "\"containers\":";
  js_containers(spec, t.containers);
  if (!t.init_containers.empty()) {
    spec += ",\"initContainers\":";
    js_containers(spec, t.init_containers);
  }
  if (rest_at) *rest_at = spec.size();  // (everything behind the containers: selectors, affinity, tolerations, constraints, volumes ...)
  if (t.has_node_selector) {
    spec += ",\"nodeSelector\":";
    js_map(spec, t.node_selector);
  }
  auto js_pod_terms = [&](const char* name, const std::vector<PodAffinityTerm>& terms, bool comma) {
    if (comma) spec.push_back(',');
    spec += std::string("\"") + name + "\":{\"requiredDuringSchedulingIgnoredDuringExecution\":[";
    for (size_t i = 0; i < terms.size(); ++i) {
      if (i) spec.push_back(',');
      spec += "{\"topologyKey\":";
      js_str(spec, terms[i].topology_key);
      if (terms[i].selector.present) {
        spec += ",\"labelSelector\":{\"matchLabels\":";
        js_map(spec, terms[i].selector.match_labels);
        spec += ",\"matchExpressions\":";
        js_reqs(spec, terms[i].selector.match_exprs);
        spec += "}";
      }
      if (!terms[i].namespaces.empty()) {
        spec += ",\"namespaces\":[";
        for (size_t j = 0; j < terms[i].namespaces.size(); ++j) {
          if (j) spec.push_back(',');
          js_str(spec, terms[i].namespaces[j]);
        }
        spec.push_back(']');
      }
      if (terms[i].all_namespaces) spec += ",\"namespaceSelector\":{}";
      if (terms[i].namespace_selector_unsupported) spec += ",\"namespaceSelector\":{\"matchLabels\":{\"x-not-mirrored\":\"true\"}}";
      spec.push_back('}');
    }
    spec += "]}";
  };
  const bool any_pod_aff = !t.pod_affinity.empty() || !t.pod_anti_affinity.empty();
  if (t.has_required || any_pod_aff) {
    spec += ",\"affinity\":{";
    if (t.has_required) {
      spec += "\"nodeAffinity\":{\"requiredDuringSchedulingIgnoredDuringExecution\":{\"nodeSelectorTerms\":[";
      for (size_t i = 0; i < t.terms.size(); ++i) {
        if (i) spec.push_back(',');
        spec += "{\"matchExpressions\":";
        js_reqs(spec, t.terms[i].exprs);
        spec += ",\"matchFields\":";
        js_reqs(spec, t.terms[i].fields);
        spec.push_back('}');
      }
      spec += "]}}";
    }
    bool comma = t.has_required;
    if (!t.pod_affinity.empty()) {
      js_pod_terms("podAffinity", t.pod_affinity, comma);
      comma = true;
    }
    if (!t.pod_anti_affinity.empty()) {
      js_pod_terms("podAntiAffinity", t.pod_anti_affinity, comma);
      comma = true;
    }
    spec += "}";
  }
  if (!t.tolerations.empty()) {
    spec += ",\"tolerations\":[";
    for (size_t i = 0; i < t.tolerations.size(); ++i) {
      const Toleration& x = t.tolerations[i];
      if (i) spec.push_back(',');
      spec += "{\"key\":";
      js_str(spec, x.key);
      spec += ",\"operator\":";
      js_str(spec, x.op);
      spec += ",\"value\":";
      js_str(spec, x.value);
      spec += ",\"effect\":";
      js_str(spec, x.effect);
      spec.push_back('}');
    }
    spec.push_back(']');
  }
  if (!t.volume_kinds.empty()) {
    spec += ",\"volumes\":[";
    for (size_t i = 0; i < t.volume_kinds.size(); ++i) {
      if (i) spec.push_back(',');
      spec += "{\"name\":\"v" + std::to_string(i) + "\",";
      js_str(spec, t.volume_kinds[i]);
      spec += ":{}}";
    }
    spec.push_back(']');
  }
  if (t.resource_claims > 0) {
    spec += ",\"resourceClaims\":[";
    for (int32_t i = 0; i < t.resource_claims; ++i) spec += std::string(i ? "," : "") + "{\"name\":\"c" + std::to_string(i) + "\"}";
    spec.push_back(']');
  }
  if (t.has_overhead) {
    spec += ",\"overhead\":";
    js_map(spec, t.overhead);
  }
  if (!t.pod_level_requests.empty()) {
    spec += ",\"resources\":{\"requests\":";
    js_map(spec, t.pod_level_requests);
    spec += "}";
  }
  if (!t.spread.empty()) {
    spec += ",\"topologySpreadConstraints\":[";
    for (size_t i = 0; i < t.spread.size(); ++i) {
      const SpreadConstraint& c = t.spread[i];
      if (i) spec.push_back(',');
      spec += "{\"maxSkew\":" + std::to_string(c.max_skew) + ",\"topologyKey\":";
      js_str(spec, c.topology_key);
      spec += ",\"whenUnsatisfiable\":";
      js_str(spec, c.when_unsatisfiable);
      if (c.raw_selector.present) {
        spec += ",\"labelSelector\":{\"matchLabels\":";
        js_map(spec, c.raw_selector.match_labels);
        spec += ",\"matchExpressions\":";
        js_reqs(spec, c.raw_selector.match_exprs);
        spec += "}";
      }
      if (c.has_min_domains) spec += ",\"minDomains\":" + std::to_string(c.min_domains);
      spec += ",\"nodeAffinityPolicy\":";
      js_str(spec, c.node_affinity_policy);
      spec += ",\"nodeTaintsPolicy\":";
      js_str(spec, c.node_taints_policy);
      if (!c.match_label_keys.empty()) {
        spec += ",\"matchLabelKeys\":[";
        for (size_t j = 0; j < c.match_label_keys.size(); ++j) {
          if (j) spec.push_back(',');
          js_str(spec, c.match_label_keys[j]);
        }
        spec.push_back(']');
      }
      spec.push_back('}');
    }
    spec.push_back(']');
  }
}

inline void pod_json(const Pod& p, std::string& o, int replicas = 1) {
  std::string meta, spec;
  template_json(*p.tpl, meta, spec);
  o += "{\"metadata\":{\"name\":";
  js_str(o, p.name);
  o += ",\"uid\":";
  js_str(o, p.uid);
  o += ",";
  o += meta;
  if (p.terminating) o += ",\"deletionTimestamp\":\"2026-01-01T00:00:00Z\"";
  o += "},\"spec\":{";
  if (!p.node_name.empty()) {
    o += "\"nodeName\":";
    js_str(o, p.node_name);
    o += ",";
  }
  o += spec;
  o += "}";
  if (replicas > 1) o += ",\"replicas\":" + std::to_string(replicas);
  o += "}";
}

// compact: runs of pods that share a template (and are not terminating) are written once with "replicas": k — the loaders
// expand them again (uids get a "#r" suffix). Keeps the snapshot of a 50 000-node cluster at tens of MB.
inline void node_json(const NodeInfo& ni, std::string& o, bool compact = false) {
  const Node& n = ni.node;
  o += "{\"metadata\":{\"name\":";
  js_str(o, n.name);
  o += ",\"labels\":";
  js_map(o, n.labels);
  o += "},\"spec\":{\"unschedulable\":";
  o += n.unschedulable ? "true" : "false";
  o += ",\"taints\":[";
  for (size_t i = 0; i < n.taints.size(); ++i) {
    if (i) o.push_back(',');
    o += "{\"key\":";
    js_str(o, n.taints[i].key);
    o += ",\"value\":";
    js_str(o, n.taints[i].value);
    o += ",\"effect\":";
    js_str(o, n.taints[i].effect);
    o.push_back('}');
  }
  o += "]},\"status\":{\"allocatable\":";
  js_map(o, n.allocatable);
  o += "},\"pods\":[";
  for (size_t i = 0; i < ni.pods.size();) {
    if (i) o.push_back(',');
    size_t run = 1;
    if (compact && !ni.pods[i]->terminating)
      while (i + run < ni.pods.size() && ni.pods[i + run]->tpl == ni.pods[i]->tpl && !ni.pods[i + run]->terminating) ++run;
    pod_json(*ni.pods[i], o, (int)run);
    i += run;
  }
  o += "]}";
}

// ---------------------------------------------------------------------------------------------------
// template interning
// ---------------------------------------------------------------------------------------------------
class TemplatePool {
 public:
  // the part of interning that touches no shared state (the scanning threads of a batch do it for the templates they parse)
  static void prepare(PodTemplate& t) {
    std::string meta, spec;
    size_t rest_at = 0;
    template_json(t, meta, spec, &rest_at);
    t.canonical = meta + "|" + spec;
    t.requests = compute_requests(t);
    t.res = to_resource(t.requests);
    std::string& sh = t.dict_shape;
    sh.clear();
    sh += t.ns;
    sh.push_back('\x1f');
    for (const HostPort& hp : template_host_ports(t)) {
      sh += hp.protocol;
      sh.push_back(',');
      sh += hp.ip;
      sh.push_back(',');
      sh += std::to_string(hp.port);
      sh.push_back(';');
    }
    sh.push_back('\x1f');
    for (auto& kv : t.requests)
      if (kv.second > 0 && is_scalar_resource_name(kv.first)) {
        sh += kv.first;
        sh.push_back(';');
      }
    sh.push_back('\x1f');
    sh.append(spec, rest_at, std::string::npos);
    // matchLabelKeys fold the POD'S label values into the selector that names the spread count class (read_template): two
    // templates of one Deployment that differ in pod-template-hash register different classes — the values belong to the shape
    for (const SpreadConstraint& c : t.spread) {
      if (c.match_label_keys.empty() || !c.selector.present) continue;
      sh.push_back('\x1f');
      for (const std::string& key : c.match_label_keys) {
        auto lv = t.labels.find(key);
        if (lv == t.labels.end()) continue;
        sh += key;
        sh.push_back('=');
        sh += lv->second;
        sh.push_back('\x1e');
      }
    }
  }
  const PodTemplate* intern(PodTemplate&& t) {
    prepare(t);
    return intern_prepared(std::move(t));
  }
  const PodTemplate* intern_prepared(PodTemplate&& t) {
    auto it = by_key_.find(t.canonical);
    if (it != by_key_.end()) return it->second.get();
    std::string key = t.canonical;
    t.shape_id = shapes_.emplace(std::move(t.dict_shape), (int32_t)shapes_.size()).first->second;
    t.dict_shape = std::string();
    auto up = std::make_unique<PodTemplate>(std::move(t));
    const PodTemplate* raw = up.get();
    by_key_.emplace(std::move(key), std::move(up));
    order_.push_back(const_cast<PodTemplate*>(raw));
    if (!raw->pod_anti_affinity.empty()) ++with_anti_;
    return raw;
  }
  const std::vector<PodTemplate*>& all() const { return order_; }
  bool any_anti_affinity() const { return with_anti_ > 0; }  // (templates are never dropped from the pool: conservative)
  size_t num_shapes() const { return shapes_.size(); }
  void clear() {
    by_key_.clear();
    order_.clear();
    shapes_.clear();
    with_anti_ = 0;
  }

 private:
  std::unordered_map<std::string, std::unique_ptr<PodTemplate>> by_key_;
  std::vector<PodTemplate*> order_;
  size_t with_anti_ = 0;  // templates with required pod anti-affinity terms
  std::unordered_map<std::string, int32_t> shapes_;  // dictionary shape → id (PodTemplate::shape_id)
};

// KEP-1287 in-place resize (resource.go:110-142, computeContainerResource / isResizeInfeasible): a container that has a
// status entry counts with max(spec requests, status.allocatedResources, status.resources.requests) per resource, or —
// when the pending resize is infeasible — with the status requests alone. The template stores that effective request
// (cpu as "<milli>m", everything else as its integer Value()), so interning, dumps and the encoder see one plain spec.
inline void fold_container_statuses(const mj::Value& pod, PodTemplate* t) {
  const mj::Value* st = pod.get_nn("status");
  if (!st) return;
  bool infeasible = st->str_or("resize", "") == "Infeasible";
  if (const mj::Value* conds = st->get_nn("conditions"))
    for (auto& c : conds->arr)
      if (c->str_or("type", "") == "PodResizePending" && c->str_or("reason", "") == "Infeasible") infeasible = true;
  std::map<std::string, const mj::Value*> by_name;  // containerStatuses first, initContainerStatuses override (:65-71)
  for (const char* field : {"containerStatuses", "initContainerStatuses"})
    if (const mj::Value* css = st->get_nn(field))
      for (auto& cs : css->arr) by_name[cs->str_or("name", "")] = cs.get();
  if (by_name.empty()) return;
  auto upmax = [](ResMap& l, const ResMap& r) {
    for (auto& kv : r) {
      auto it = l.find(kv.first);
      if (it == l.end())
        l[kv.first] = kv.second;
      else if (kv.second > it->second)
        it->second = kv.second;
    }
  };
  for (auto* list : {&t->containers, &t->init_containers})
    for (Container& c : *list) {
      auto it = by_name.find(c.name);
      if (it == by_name.end()) continue;
      const mj::Value* res = it->second->get_nn("resources");
      ResMap combined;
      if (infeasible && res) {
        combined = get_resource(read_strmap(res->get_nn("requests")));
      } else {
        upmax(combined, get_resource(c.requests));
        upmax(combined, get_resource(read_strmap(it->second->get_nn("allocatedResources"))));
        if (res) upmax(combined, get_resource(read_strmap(res->get_nn("requests"))));
      }
      c.requests.clear();
      for (auto& kv : combined) c.requests[kv.first] = std::to_string(kv.second) + (kv.first == "cpu" ? "m" : "");
    }
}

inline PodTemplate read_template(const mj::Value& v) {
  PodTemplate t;
  if (const mj::Value* md = v.get_nn("metadata")) {
    t.ns = md->str_or("namespace", "");
    t.labels = read_strmap(md->get_nn("labels"));
  }
  const mj::Value* spec = v.get_nn("spec");
  if (!spec) return t;
  if (const mj::Value* ns = spec->get_nn("nodeSelector")) {
    t.has_node_selector = true;
    t.node_selector = read_strmap(ns);
  }
  if (const mj::Value* aff = spec->get_nn("affinity")) {
    if (const mj::Value* na = aff->get_nn("nodeAffinity"))
      if (const mj::Value* req = na->get_nn("requiredDuringSchedulingIgnoredDuringExecution")) {
        t.has_required = true;
        if (const mj::Value* terms = req->get_nn("nodeSelectorTerms"))
          for (auto& x : terms->arr) {
            SelectorTerm term;
            term.exprs = read_requirements(x->get_nn("matchExpressions"));
            term.fields = read_requirements(x->get_nn("matchFields"));
            t.terms.push_back(std::move(term));
          }
      }
    auto read_pod_terms = [&](const mj::Value* pa, std::vector<PodAffinityTerm>* out) {
      if (!pa) return;
      const mj::Value* req = pa->get_nn("requiredDuringSchedulingIgnoredDuringExecution");
      if (!req || !req->is_arr()) return;
      for (auto& x : req->arr) {
        PodAffinityTerm term;
        if (const mj::Value* ls = x->get_nn("labelSelector")) {
          term.selector.present = true;
          term.selector.match_labels = read_strmap(ls->get_nn("matchLabels"));
          term.selector.match_exprs = read_requirements(ls->get_nn("matchExpressions"));
        }
        if (const mj::Value* nss = x->get_nn("namespaces"))
          for (auto& y : nss->arr) term.namespaces.push_back(y->s);
        term.topology_key = x->str_or("topologyKey", "");
        if (const mj::Value* nsel = x->get_nn("namespaceSelector")) {
          const mj::Value* ml = nsel->get_nn("matchLabels");
          const mj::Value* me = nsel->get_nn("matchExpressions");
          const bool empty = (!ml || ml->obj.empty()) && (!me || me->arr.empty());
          term.all_namespaces = empty;  // metav1.LabelSelectorAsSelector({}) = Everything
          term.namespace_selector_unsupported = !empty;
          if (!empty) t.pod_affinity_unsupported = true;
        }
        out->push_back(std::move(term));
      }
    };
    read_pod_terms(aff->get_nn("podAffinity"), &t.pod_affinity);
    read_pod_terms(aff->get_nn("podAntiAffinity"), &t.pod_anti_affinity);
  }
  if (const mj::Value* tols = spec->get_nn("tolerations"))
    for (auto& x : tols->arr)
      t.tolerations.push_back({x->str_or("key", ""), x->str_or("operator", ""), x->str_or("value", ""), x->str_or("effect", "")});
  t.containers = read_containers(spec->get_nn("containers"), false);
  t.init_containers = read_containers(spec->get_nn("initContainers"), true);
  fold_container_statuses(v, &t);
  if (const mj::Value* oh = spec->get_nn("overhead")) {
    t.has_overhead = true;
    t.overhead = read_strmap(oh);
  }
  if (const mj::Value* res = spec->get_nn("resources")) t.pod_level_requests = read_strmap(res->get_nn("requests"));
  if (const mj::Value* vols = spec->get_nn("volumes"))
    for (auto& vol : vols->arr) {
      if (!vol->is_obj()) continue;
      for (auto& kv : vol->obj) {
        if (kv.first == "name" || kv.second->is_null()) continue;
        // node-local volume sources no Volume* Filter looks at
        static const char* kLocal[] = {"emptyDir", "configMap", "secret", "downwardAPI", "projected", "hostPath", "image"};
        bool local = false;
        for (const char* k : kLocal) local = local || kv.first == k;
        if (!local && std::find(t.volume_kinds.begin(), t.volume_kinds.end(), kv.first) == t.volume_kinds.end()) t.volume_kinds.push_back(kv.first);
      }
    }
  std::sort(t.volume_kinds.begin(), t.volume_kinds.end());
  if (const mj::Value* rc = spec->get_nn("resourceClaims"))
    if (rc->is_arr()) t.resource_claims = (int32_t)rc->arr.size();
  if (const mj::Value* tsc = spec->get_nn("topologySpreadConstraints"))
    for (auto& c : tsc->arr) {
      SpreadConstraint sc;
      sc.max_skew = (int32_t)c->int_or("maxSkew", 1);
      sc.topology_key = c->str_or("topologyKey", "");
      sc.when_unsatisfiable = c->str_or("whenUnsatisfiable", "DoNotSchedule");
      if (const mj::Value* ls = c->get_nn("labelSelector")) {
        sc.selector.present = true;
        sc.selector.match_labels = read_strmap(ls->get_nn("matchLabels"));
        sc.selector.match_exprs = read_requirements(ls->get_nn("matchExpressions"));
      }
      if (c->get_nn("minDomains")) {
        sc.has_min_domains = true;
        sc.min_domains = (int32_t)c->int_or("minDomains", 1);
      }
      sc.node_affinity_policy = c->str_or("nodeAffinityPolicy", "Honor");
      sc.node_taints_policy = c->str_or("nodeTaintsPolicy", "Ignore");
      if (const mj::Value* mk = c->get_nn("matchLabelKeys"))
        for (auto& x : mk->arr) sc.match_label_keys.push_back(x->s);
      sc.raw_selector = sc.selector;
      if (sc.selector.present)  // a nil selector stays labels.Nothing() (mergeLabelSetWithSelector returns it unchanged)
        for (auto& key : sc.match_label_keys) {
          auto lv = t.labels.find(key);
          if (lv == t.labels.end()) continue;  // keys the pod does not carry are ignored
          Requirement r;
          r.key = key;
          r.op = "In";
          r.values.push_back(lv->second);
          sc.selector.match_exprs.push_back(std::move(r));
        }
      t.spread.push_back(std::move(sc));
    }
  return t;
}

inline Node read_node(const mj::Value& v) {
  Node n;
  if (const mj::Value* md = v.get_nn("metadata")) {
    n.name = md->str_or("name", "");
    n.labels = read_strmap(md->get_nn("labels"));
  }
  if (const mj::Value* spec = v.get_nn("spec")) {
    n.unschedulable = spec->bool_or("unschedulable", false);
    if (const mj::Value* ts = spec->get_nn("taints"))
      for (auto& t : ts->arr) n.taints.push_back({t->str_or("key", ""), t->str_or("value", ""), t->str_or("effect", "")});
  }
  if (const mj::Value* st = v.get_nn("status")) n.allocatable = read_strmap(st->get_nn("allocatable"));
  return n;
}

}  // namespace ykh
