// encoder.h — object model → the structure-of-arrays tables of include/ykpred.h.
//
// This is the work the Go side does ONCE per pod / node update instead of once per (pod,node) pair:
//   * request vectors (pkg/common/resource.go:56-109 → int64 per resource dimension),
//   * a taint dictionary: every distinct NoSchedule/NoExecute (key,value,effect) on any node gets a bit; a node
//     is the OR of its taints' bits, a pod spec the set of dictionary taints its tolerations tolerate
//     (v1.Toleration.ToleratesTaint) — TaintToleration.Filter becomes (taints & ~tolerated) == 0,
//   * a requirement dictionary: every distinct node-selector requirement that some pod uses (label
//     expressions, nodeSelector pairs, matchFields, PreFilter node names) gets a bit, evaluated once per node
//     with the full apimachinery semantics (validation, In/NotIn/Exists/DoesNotExist/Gt/Lt); a selector
//     term becomes a mask and NodeAffinity.Filter becomes "some term's mask ⊆ node bits".
//   * a host-port dictionary (NodePorts): every distinct (protocol, hostIP, hostPort) some ask requests gets a bit; a
//     node's bit says "a pod already here conflicts with it" (HostPortInfo.CheckConflict incl. the 0.0.0.0 wildcard),
//   * topology-domain ids and per-selector matching-pod counts for PodTopologySpread.
// Asks carrying features the engine does not evaluate (volumes / DRA claims, namespaceSelector, matchLabelKeys, specs the
// API server would have rejected) or whose dictionary entries do not fit the engine's limits are marked UNSUPPORTED one by
// one (Encoder::unsupported: template → reason): their spec row carries YKPRED_SPEC_UNSUPPORTED, every pair of their bitmap
// row is 0 with plugin code YKPRED_CODE_UNSUPPORTED, and the host routes exactly those asks to the CPU predicate manager
//  — never silently passed, and every other ask keeps evaluating on the device.
#pragma once
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <functional>
#include <limits>
#include <set>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../../include/ykpred.h"
#include "objects.h"

namespace ykh {

// ---- apimachinery validation (labels.NewRequirement); pin: predicate_manager_test.go:793-820 -------------
inline bool is_alnum(char c) { return (c >= 'a' && c <= 'z') || (c >= 'A' && c <= 'Z') || (c >= '0' && c <= '9'); }
inline bool is_name_part(const std::string& s) {
  if (s.empty() || s.size() > 63) return false;
  if (!is_alnum(s.front()) || !is_alnum(s.back())) return false;
  for (char c : s)
    if (!is_alnum(c) && c != '-' && c != '_' && c != '.') return false;
  return true;
}
inline bool is_dns1123_subdomain(const std::string& s) {
  if (s.empty() || s.size() > 253) return false;
  auto ok = [](char c) { return (c >= 'a' && c <= 'z') || (c >= '0' && c <= '9'); };
  size_t start = 0;
  for (;;) {
    size_t dot = s.find('.', start);
    size_t end = dot == std::string::npos ? s.size() : dot;
    if (end == start) return false;
    if (!ok(s[start]) || !ok(s[end - 1])) return false;
    for (size_t i = start; i < end; ++i)
      if (!ok(s[i]) && s[i] != '-') return false;
    if (dot == std::string::npos) return true;
    start = dot + 1;
  }
}
inline bool is_qualified_name(const std::string& k) {
  size_t slash = k.find('/');
  if (slash == std::string::npos) return is_name_part(k);
  if (k.find('/', slash + 1) != std::string::npos) return false;
  return is_dns1123_subdomain(k.substr(0, slash)) && is_name_part(k.substr(slash + 1));
}
inline bool is_valid_label_value(const std::string& v) { return v.empty() || is_name_part(v); }
inline bool parse_int64(const std::string& s, int64_t* out) {  // strconv.ParseInt(s, 10, 64)
  size_t p = 0;
  bool neg = false;
  if (p < s.size() && (s[p] == '+' || s[p] == '-')) neg = s[p++] == '-';
  if (p >= s.size()) return false;
  unsigned __int128 v = 0;
  const unsigned __int128 lim = (unsigned __int128)1 << 63;
  for (; p < s.size(); ++p) {
    if (s[p] < '0' || s[p] > '9') return false;
    v = v * 10 + (unsigned)(s[p] - '0');
    if (v > lim) return false;
  }
  if (!neg && v >= lim) return false;
  *out = neg ? (int64_t)(-(__int128)v) : (int64_t)v;
  return true;
}
inline bool valid_label_requirement(const Requirement& r) {
  if (!is_qualified_name(r.key)) return false;
  if (r.op == "In" || r.op == "NotIn") {
    if (r.values.empty()) return false;
  } else if (r.op == "Exists" || r.op == "DoesNotExist") {
    if (!r.values.empty()) return false;
  } else if (r.op == "Gt" || r.op == "Lt") {
    int64_t tmp;
    if (r.values.size() != 1 || !parse_int64(r.values[0], &tmp)) return false;
  } else {
    return false;
  }
  for (auto& v : r.values)
    if (!is_valid_label_value(v)) return false;
  return true;
}
inline bool label_requirement_matches(const Requirement& r, const StrMap& labels) {  // labels.Requirement.Matches
  auto it = labels.find(r.key);
  bool has = it != labels.end();
  auto in = [&](const std::string& v) { return std::find(r.values.begin(), r.values.end(), v) != r.values.end(); };
  if (r.op == "In") return has && in(it->second);
  if (r.op == "NotIn") return !has || !in(it->second);
  if (r.op == "Exists") return has;
  if (r.op == "DoesNotExist") return !has;
  if (r.op == "Gt" || r.op == "Lt") {
    int64_t lv, rv;
    if (!has || !parse_int64(it->second, &lv) || r.values.size() != 1 || !parse_int64(r.values[0], &rv)) return false;
    return r.op == "Gt" ? lv > rv : lv < rv;
  }
  return false;
}
inline bool toleration_tolerates(const Toleration& t, const Taint& taint) {  // v1.Toleration.ToleratesTaint
  if (!t.effect.empty() && t.effect != taint.effect) return false;
  if (!t.key.empty() && t.key != taint.key) return false;
  if (t.op.empty() || t.op == "Equal") return t.value == taint.value;
  return t.op == "Exists";
}

// metav1.LabelSelectorAsSelector(...).Matches(labels). `*invalid` is set when the selector cannot be built
// (PodTopologySpread.PreFilter would return an Error status for such a pod).
inline bool selector_matches(const LabelSelector& s, const StrMap& labels, bool* invalid) {
  if (!s.present) return false;  // nil selector = labels.Nothing()
  for (auto& kv : s.match_labels) {
    if (!is_qualified_name(kv.first) || !is_valid_label_value(kv.second)) {
      *invalid = true;
      return false;
    }
    auto it = labels.find(kv.first);
    if (it == labels.end() || it->second != kv.second) return false;
  }
  for (auto& r : s.match_exprs) {
    if ((r.op != "In" && r.op != "NotIn" && r.op != "Exists" && r.op != "DoesNotExist") || !valid_label_requirement(r)) {
      *invalid = true;
      return false;
    }
    if (!label_requirement_matches(r, labels)) return false;
  }
  return true;
}
inline bool selector_counts_nothing(const LabelSelector& s) {  // nil (Nothing) or empty (Everything ⇒ countPodsMatchSelector returns 0)
  return !s.present || (s.match_labels.empty() && s.match_exprs.empty());
}
inline std::string selector_key(const std::string& ns, const LabelSelector& s) {
  std::string k = ns + '\x1f';
  for (auto& kv : s.match_labels) k += kv.first + '=' + kv.second + '\x1e';
  k += '\x1f';
  for (auto& r : s.match_exprs) {
    k += r.key + '\x1d' + r.op;
    for (auto& v : r.values) k += '\x1d' + v;
    k += '\x1e';
  }
  return k;
}

// ---- dictionary entries -------------------------------------------------------------------------------
struct DictReq {
  enum Kind { kLabel, kEquals, kField, kNameIn } kind;
  Requirement req;  // kLabel: validated requirement; kEquals: key + values[0]; kField: key/op/values[0]; kNameIn: values[0]
  bool eval(const Node& n) const {
    switch (kind) {
      case kLabel: return label_requirement_matches(req, n.labels);
      case kEquals: {
        auto it = n.labels.find(req.key);
        return it != n.labels.end() && it->second == req.values[0];
      }
      case kField: {
        // nodeSelectorTerm.match consults matchFields only when the node has a name (extractNodeFields)
        if (n.name.empty()) return true;
        std::string fv = req.key == "metadata.name" ? n.name : std::string();
        bool eq = fv == req.values[0];
        return req.op == "In" ? eq : !eq;
      }
      case kNameIn: return n.name == req.values[0];
    }
    return false;
  }
};

struct EncodedSpec {
  std::vector<int64_t> req;          // [R]
  std::vector<uint64_t> tol;         // [KT]
  uint32_t flags = 0;
  std::vector<std::vector<uint64_t>> terms, pre_terms;  // each [W]
  std::vector<ykpred_spread_t> spread;                  // hard (DoNotSchedule) topology spread constraints
};
// A "count class": how many pods (or pod/term pairs) on a node match something. One column of selector_count[KS][N].
struct SelectorClass {
  enum Kind { kSpread, kAffinityAll, kAntiTerm, kExistingAnti } kind = kSpread;
  std::string ns;                      // kSpread: the constraint's namespace; others: the owning (incoming) pod's namespace
  LabelSelector selector;              // kSpread
  std::vector<PodAffinityTerm> terms;  // kAffinityAll: all required affinity terms; kAntiTerm: one anti-affinity term
  StrMap labels;                       // kExistingAnti: the incoming pod's labels
  std::string topology_key;            // kExistingAnti
};
// framework.AffinityTerm.Matches(pod, nil): namespace rule (default = owner's namespace) and label selector
inline bool pod_term_matches(const PodAffinityTerm& t, const std::string& owner_ns, const std::string& target_ns, const StrMap& target_labels) {
  // namespaces ∪ namespaceSelector; both empty/nil = the owner's namespace (framework.newAffinityTerm / getNamespacesFromPodAffinityTerm)
  bool ns_ok = t.all_namespaces ||
               (t.namespaces.empty() ? target_ns == owner_ns : std::find(t.namespaces.begin(), t.namespaces.end(), target_ns) != t.namespaces.end());
  if (!ns_ok) return false;
  bool invalid = false;
  return selector_matches(t.selector, target_labels, &invalid);
}
inline std::string pod_terms_key(const std::vector<PodAffinityTerm>& terms) {
  std::string k;
  for (auto& t : terms) {
    k += selector_key(t.selector.present ? "+" : "-", t.selector);
    for (auto& n : t.namespaces) k += '\x1c' + n;
    if (t.all_namespaces) k += "\x1c*";
    k += '\x1b';
  }
  return k;
}

class Encoder {
 public:
  std::string error;
  int R = 3, KT = 1, W = 1;
  std::vector<std::string> scalar_names;  // resource dimension 3+i
  // Taints. Every distinct NoSchedule / NoExecute (key, value, effect) on any node is an entry of taint_dict — unbounded: a
  // cluster autoscaler stamps a unique ToBeDeletedByClusterAutoscaler=<timestamp> on every node it drains. What is bounded
  // is the number of BITS: taints that no pending ask can tell apart (the same set of toleration lists tolerates them)
  // share one bit, taint_bit[i]; a node's word is the OR over its taints' bits, a spec's word says which bits it tolerates.
  std::vector<Taint> taint_dict;
  std::vector<int32_t> taint_bit;                    // [taint] → bit
  std::vector<std::vector<int32_t>> taint_members;   // [bit] → taints
  int overflow_taint_bit = -1;                       // the bit that took every group beyond the engine's 256 (-1: none had to)
  std::vector<DictReq> req_dict;
  int KD = 0, KS = 0, KP = 0;
  std::vector<HostPort> port_dict;                                    // requested host port k (NodePorts)
  std::vector<std::string> topo_keys;                                 // topology key k
  std::vector<std::unordered_map<std::string, int>> domain_ids;       // value → id, per key (ids follow sorted values)
  std::vector<SelectorClass> sel_classes;                             // selector class s
  std::unordered_map<const PodTemplate*, std::string> unsupported;    // asks' templates the engine does not evaluate → why
  // engine limits (kernels.hip.h: kMaxR / kMaxKT / kMaxW / kMaxKP / kMaxKD; selector classes: ykpred_create)
  static constexpr int kLimitR = 8, kLimitTaintBits = 256, kLimitRequirements = 2048, kLimitPorts = 256, kLimitTopoKeys = 8, kLimitClasses = 4096;

  // Builds every dictionary from the current objects. Returns false (error set) on unsupported input.
  // may_have_existing_anti = false: the caller knows that NO pod template of the cluster carries a required anti-affinity term
  // (the common case) — the pass over every pod on every node that collects such templates is skipped (2.76 M pointer chases,
  // 45 of the 165 ms of a full encode at configs[2] size).
  // What a template asks of the dictionaries that depends on nothing but the template (its refusal, if any; the host ports it
  // requests; its selector requirements with their dictionary keys): prepared on the caller's threads (parallel_for) for the shape
  // representatives, consumed by the ordered loop — which then does hash lookups only.
  struct ReqItem {
    DictReq::Kind kind;
    std::string key, op;
    std::vector<std::string> values;
    std::string rk;
  };
  struct Prepared {
    std::string error;
    std::vector<HostPort> ports;
    std::vector<ReqItem> reqs;
    std::string tol_key;  // the toleration list's identity (assign_taint_bits)
    std::vector<std::string> scalars;  // names of the scalar resources the template requests (dimensions exist because an ask requests them)
    bool plain = false;   // no pod (anti)affinity term, no spread constraint: the ordered loop has nothing to read in the template itself
  };
  static std::string toleration_list_key(const PodTemplate& t) {
    std::string k;
    for (auto& tol : t.tolerations) k += tol.key + '\x1f' + tol.op + '\x1f' + tol.value + '\x1f' + tol.effect + '\x1e';
    return k;
  }
  using ParallelFor = std::function<void(int, const std::function<void(int)>&)>;
  bool trace = false;
  bool use_shapes = true;  // (encode_tables clears it for one rebuild when a shape turned out not to stand for its templates)
  // shape_reps (optional): the indices into `templates` of the first template of every dictionary shape, ascending — the caller
  // found them on its threads; without it the loop below finds them itself, one template after the other.
  bool build_dictionaries(const std::vector<NodeInfo*>& nodes, const std::vector<PodTemplate*>& templates, bool may_have_existing_anti = true,
                          const std::vector<int32_t>* shape_reps = nullptr, const ParallelFor* parallel_for = nullptr) {
    error.clear();
    scalar_names.clear();
    taint_dict.clear();
    taint_bit.clear();
    taint_members.clear();
    overflow_taint_bit = -1;
    wild_anti_terms_.clear();
    req_dict.clear();
    scalar_ix_.clear();
    taint_ix_.clear();
    req_ix_.clear();
    topo_keys.clear();
    topo_ix_.clear();
    domain_ids.clear();
    sel_classes.clear();
    sel_ix_.clear();
    sel_keys_.clear();
    req_keys_.clear();
    unsupported.clear();
    sel_memo_.clear();
    port_dict.clear();
    port_ix_.clear();
    // InterPodAffinity: topology keys of the anti-affinity terms carried by pods that are already on nodes — and by the pending
    // asks themselves: an allocation round ASSUMES asks one after the other (context.go:828-885), and an assumed pod is an
    // existing pod for every ask behind it (satisfyExistingPodsAntiAffinity). Their count classes exist from the start (all
    // zero until somebody is assumed), so a round never has to stop for a dictionary rebuild and the device can run it.
    existing_anti_templates_.clear();
    if (may_have_existing_anti) {
      std::set<const PodTemplate*> seen;
      auto consider = [&](const PodTemplate* tpl) {
        if (tpl->pod_anti_affinity.empty() || !seen.insert(tpl).second) return;
        // An anti-affinity term of a pod that already runs constrains the asks its selector matches. A term the mirror
        // cannot decide (non-empty namespaceSelector: needs Namespace labels; a selector that does not parse) costs
        // exactly THOSE asks their place on the engine — whoever it cannot match (labels) is unaffected whatever the
        // namespaces turn out to be. Topology keys beyond the engine's 8 are handled the same way.
        bool any_evaluable = false;
        for (auto& term : tpl->pod_anti_affinity) {
          bool invalid = false;
          selector_matches(term.selector, tpl->labels, &invalid);
          const bool new_key = !topo_ix_.count(term.topology_key);
          if (invalid || term.namespace_selector_unsupported || (new_key && (int)topo_keys.size() >= kLimitTopoKeys)) {
            wild_anti_terms_.push_back(&term);
            continue;
          }
          topo_key(term.topology_key);
          any_evaluable = true;
        }
        if (any_evaluable) existing_anti_templates_.push_back(tpl);
      };
      // (pending templates first: an ask that is assumed during a round moves its template from "pending" to "on a node", and
      // the numbering of topology keys and count classes must not depend on which of the two lists met it first. A pending
      // template the engine cannot evaluate is never assumed by a device round: it stays out)
      for (const PodTemplate* t : templates)
        if (!t->pod_anti_affinity.empty() && template_error(*t).empty()) consider(t);
      for (const NodeInfo* ni : nodes)
        for (const Pod* p : ni->pods) consider(p->tpl);
    }
    // Node-driven entries. Nothing here can cost the cluster its engine: taints are unbounded (their BITS are assigned below,
    // once the asks' toleration lists are known), and a scalar resource only becomes a dimension when an ask requests it —
    // NodeResourcesFit never looks at a resource the pod does not ask for, however many device plugins the nodes advertise.
    auto tq0 = std::chrono::steady_clock::now();
    auto qlap = [&](const char* w) {  // (trace: the phases of the build on stderr — the host sets it from YKHOST_TRACE_ENCODE)
      if (!trace) return;
      auto n = std::chrono::steady_clock::now();
      fprintf(stderr, "  dict %s %.1f ms\n", w, std::chrono::duration<double, std::milli>(n - tq0).count());
      tq0 = n;
    };
    for (const NodeInfo* ni : nodes)
      for (auto& t : ni->node.taints)
        if (t.effect == "NoSchedule" || t.effect == "NoExecute") taint(t);
    qlap("node taints");
    // ask-driven entries, template by template: a template whose entries would not fit is rolled back and marked
    // unsupported on its own — the asks before and after it keep their place in the dictionaries
    // Templates of one dictionary SHAPE (PodTemplate::shape_id: everything but labels and request values) register the same
    // entries and are refused for the same reasons — one of them is visited, the others inherit its verdict. Only while no
    // anti-affinity term of an on-node or pending pod exists: those are matched against every ask's LABELS.
    const bool by_shape = use_shapes && existing_anti_templates_.empty() && wild_anti_terms_.empty();
    std::vector<const PodTemplate*> shape_rep;
    std::vector<PodTemplate*> visited;  // the templates that went through the loop body, in order (assign_taint_bits reads their toleration lists)
    const bool listed = by_shape && shape_reps != nullptr;
    if (by_shape && !listed) {
      int32_t max_shape = -1;
      for (const PodTemplate* t : templates) max_shape = std::max(max_shape, t->shape_id);
      shape_rep.assign((size_t)(max_shape + 1), nullptr);
    }
    std::vector<Prepared> prepared;
    if (listed && parallel_for && shape_reps->size() >= 4096) {
      // (by_shape: no anti-affinity term of an on-node pod is in play — template_error reads nothing this loop changes)
      prepared.resize(shape_reps->size());
      (*parallel_for)((int)shape_reps->size(), [&](int i) {
        const PodTemplate& t = *templates[(size_t)(*shape_reps)[(size_t)i]];
        Prepared& p = prepared[(size_t)i];
        p.error = template_error(t);
        p.tol_key = toleration_list_key(t);
        if (!p.error.empty()) return;
        p.ports = template_host_ports(t);
        p.reqs = requirement_items(t);
        for (auto& kv : t.requests)
          if (kv.second > 0 && is_scalar_resource_name(kv.first)) p.scalars.push_back(kv.first);
        p.plain = t.pod_affinity.empty() && t.pod_anti_affinity.empty() && t.spread.empty();
      });
    }
    auto visit = [&](PodTemplate* t, Prepared* pre) {
      visited.push_back(t);
      {
        std::string why = pre ? pre->error : template_error(*t);
        if (!why.empty()) {
          unsupported[t] = why;
          return;
        }
      }
      const Mark mark = mark_now();
      if (pre && pre->plain) {
        // (everything the template asks of the dictionaries was prepared: the loop reads one contiguous record, not the template)
        for (const HostPort& hp : pre->ports)
          if (port_ix_.emplace(port_key(hp), (int)port_dict.size()).second) port_dict.push_back(hp);
        for (const std::string& name : pre->scalars) scalar(name);
        for (ReqItem& it : pre->reqs) add_req_item(std::move(it));
        const char* full = nullptr;
        if (3 + (int)scalar_names.size() > kLimitR) full = "scalar resource names (engine limit: 5 besides cpu, memory, ephemeral-storage)";
        else if ((int)req_dict.size() > kLimitRequirements) full = "distinct node-selector requirements (engine limit: 2048)";
        else if ((int)port_dict.size() > kLimitPorts) full = "distinct requested host ports (engine limit: 256)";
        if (full) {
          rollback(mark);
          unsupported[t] = std::string("the ask needs more ") + full + " than the dictionaries can still take";
        }
        return;
      }
      for (auto* terms : {&t->pod_affinity, &t->pod_anti_affinity})
        for (auto& term : *terms) topo_key(term.topology_key);
      if (!t->pod_affinity.empty())
        count_class("A|" + t->ns + '\x1f' + pod_terms_key(t->pod_affinity), SelectorClass{SelectorClass::kAffinityAll, t->ns, {}, t->pod_affinity, {}, ""});
      for (auto& term : t->pod_anti_affinity)
        count_class("B|" + t->ns + '\x1f' + pod_terms_key({term}), SelectorClass{SelectorClass::kAntiTerm, t->ns, {}, {term}, {}, ""});
      for (const PodTemplate* et : existing_anti_templates_)
        for (auto& term : et->pod_anti_affinity)
          if (!is_wild(&term) && pod_term_matches(term, et->ns, t->ns, t->labels))
            count_class("E|" + t->ns + '\x1f' + labels_key(t->labels) + '\x1f' + term.topology_key,
                        SelectorClass{SelectorClass::kExistingAnti, t->ns, {}, {}, t->labels, term.topology_key});
      for (const HostPort& hp : pre ? pre->ports : template_host_ports(*t))
        if (port_ix_.emplace(port_key(hp), (int)port_dict.size()).second) port_dict.push_back(hp);
      for (auto& c : t->spread) {
        if (c.when_unsatisfiable != "DoNotSchedule") continue;  // ScheduleAnyway constraints only score
        topo_key(c.topology_key);
        if (!selector_counts_nothing(c.selector))
          count_class("S|" + selector_key(t->ns, c.selector), SelectorClass{SelectorClass::kSpread, t->ns, c.selector, {}, {}, ""});
      }
      for (auto& kv : t->requests)
        if (kv.second > 0 && is_scalar_resource_name(kv.first)) scalar(kv.first);  // a dimension exists because an ask requests it
      if (pre) {
        for (ReqItem& it : pre->reqs) add_req_item(std::move(it));
      } else {
        collect_requirements(*t);
      }
      const char* over = nullptr;
      if (3 + (int)scalar_names.size() > kLimitR) over = "scalar resource names (engine limit: 5 besides cpu, memory, ephemeral-storage)";
      else if ((int)req_dict.size() > kLimitRequirements) over = "distinct node-selector requirements (engine limit: 2048)";
      else if ((int)port_dict.size() > kLimitPorts) over = "distinct requested host ports (engine limit: 256)";
      else if ((int)topo_keys.size() > kLimitTopoKeys) over = "topology keys in spread / pod-affinity constraints (engine limit: 8)";
      else if ((int)sel_classes.size() > kLimitClasses) over = "distinct pod-selector classes (engine limit: 4096)";
      if (over) {
        rollback(mark);
        unsupported[t] = std::string("the ask needs more ") + over + " than the dictionaries can still take";
      }
    };
    if (listed) {
      for (size_t i = 0; i < shape_reps->size(); ++i) visit(templates[(size_t)(*shape_reps)[i]], prepared.empty() ? nullptr : &prepared[i]);
      if (!unsupported.empty()) {  // the other templates of a refused shape inherit the verdict
        std::unordered_map<int32_t, const std::string*> refused;
        for (const PodTemplate* t : visited) {
          auto un = unsupported.find(t);
          if (un != unsupported.end()) refused.emplace(t->shape_id, &un->second);
        }
        std::vector<std::pair<const PodTemplate*, const std::string*>> heirs;
        for (const PodTemplate* t : templates) {
          auto r = refused.find(t->shape_id);
          if (r != refused.end() && !unsupported.count(t)) heirs.emplace_back(t, r->second);
        }
        for (auto& hr : heirs) unsupported.emplace(hr.first, *hr.second);  // (references into an unordered_map outlive its rehashes)
      }
    } else {
      for (PodTemplate* t : templates) {
        if (by_shape && t->shape_id >= 0) {
          const PodTemplate*& rep = shape_rep[(size_t)t->shape_id];
          if (rep) {
            auto un = unsupported.find(rep);
            if (un != unsupported.end()) unsupported[t] = un->second;
            continue;
          }
          rep = t;
        }
        visit(t, nullptr);
      }
    }
    qlap("templates");
    if (trace) fprintf(stderr, "  dict visited %zu of %zu templates, by_shape %d\n", visited.size(), templates.size(), (int)by_shape);
    assign_taint_bits(visited, templates, prepared.empty() ? nullptr : &prepared);
    qlap("taint bits");
    KD = (int)topo_keys.size();
    KS = (int)sel_classes.size();
    KP = ((int)port_dict.size() + 63) / 64;
    domain_ids.assign((size_t)KD, {});
    for (int k = 0; k < KD; ++k) {
      std::set<std::string> values;
      for (const NodeInfo* ni : nodes) {
        auto it = ni->node.labels.find(topo_keys[(size_t)k]);
        if (it != ni->node.labels.end()) values.insert(it->second);
      }
      int id = 0;
      for (auto& v : values) domain_ids[(size_t)k][v] = id++;
    }
    R = 3 + (int)scalar_names.size();
    KT = std::max(1, ((int)taint_members.size() + 63) / 64);
    // at least 32 spare requirement bits: an ask that arrives later with a selector nobody used before gets its bit(s)
    // without re-encoding the cluster (extend_requirements)
    W = std::min(kLimitRequirements / 64, std::max(1, ((int)req_dict.size() + 32 + 63) / 64));
    qlap("domains");
    rebuild_requirement_index();
    qlap("requirement index");
    return true;
  }

  // Dictionary growth: the node-selector requirements of template `t` that have no bit yet get the next free bits — if the
  // allocated words (W) still hold them and the template needs nothing else that is missing (scalar resource, host port,
  // topology key, count class). `new_bits` receives their indices: the caller evaluates each on every node
  // (req_dict[q].eval) and patches the label words. False = not possible this way (the caller re-encodes everything).
  bool extend_requirements(const PodTemplate& t, std::vector<int>* new_bits) {
    new_bits->clear();
    if (!template_error(t).empty() || unsupported.count(&t)) return false;
    const Mark mark = mark_now();
    collect_requirements(t);
    bool ok = (int)req_dict.size() <= 64 * W;
    if (ok) {
      EncodedSpec es;
      std::vector<uint64_t> wanted;
      ok = encode_spec_if_covered(t, &es, &wanted);
    }
    if (!ok) {
      rollback(mark);
      return false;
    }
    for (size_t q = mark.reqs; q < req_dict.size(); ++q) new_bits->push_back((int)q);
    rebuild_requirement_index();
    return true;
  }

  void rebuild_requirement_index() {
    label_keys_.clear();
    label_reqs_.clear();
    name_reqs_.clear();
    name_other_.clear();
    name_equals_.clear();
    label_memo_.clear();
    std::set<std::string> keys;
    for (size_t q = 0; q < req_dict.size(); ++q) {
      if (req_dict[q].kind == DictReq::kLabel || req_dict[q].kind == DictReq::kEquals) {
        label_reqs_.push_back((int)q);
        keys.insert(req_dict[q].req.key);
      } else {
        name_reqs_.push_back((int)q);
        const DictReq& d = req_dict[q];
        const bool equals_name = d.kind == DictReq::kNameIn || (d.kind == DictReq::kField && d.req.key == "metadata.name" && d.req.op == "In");
        if (equals_name)
          name_equals_[d.req.values[0]].push_back((int)q);  // true exactly on the node of that name
        else
          name_other_.push_back((int)q);
      }
    }
    label_keys_.assign(keys.begin(), keys.end());
  }

  // What build_dictionaries refuses in a pending ask's template ("" = acceptable): features the engine does not evaluate
  // and inputs the upstream PreFilters reject.
  std::string template_error(const PodTemplate& t) const {
    if (!t.volume_kinds.empty()) {
      std::string kinds;
      for (auto& k : t.volume_kinds) kinds += (kinds.empty() ? "" : ", ") + k;
      return "the pod mounts volumes (" + kinds + ") that VolumeBinding / VolumeZone / VolumeRestrictions / NodeVolumeLimits must check against PV, PVC "
             "and CSINode state the engine does not hold";
    }
    if (t.resource_claims > 0) return "the pod carries spec.resourceClaims (DynamicResources needs ResourceSlice / ResourceClaim state the engine does not hold)";
    if (t.pod_affinity_unsupported) return "a pod (anti)affinity term carries a non-empty namespaceSelector (needs Namespace labels the engine does not hold)";
    for (const PodAffinityTerm* wt : wild_anti_terms_) {
      bool invalid = false;
      if (selector_matches(wt->selector, t.labels, &invalid) || invalid)
        return "a pod already on a node carries an anti-affinity term the engine cannot decide (non-empty namespaceSelector, unparsable selector or "
               "a topology key beyond the engine's 8) whose labelSelector matches this ask";
    }
    for (auto* terms : {&t.pod_affinity, &t.pod_anti_affinity})
      for (auto& term : *terms) {
        bool invalid = false;
        selector_matches(term.selector, t.labels, &invalid);
        if (invalid) return "invalid labelSelector in a pod (anti)affinity term (InterPodAffinity.PreFilter would reject the pod)";
      }
    std::set<std::string> keys_seen;
    for (auto& c : t.spread) {
      if (c.when_unsatisfiable != "DoNotSchedule") continue;
      if (!keys_seen.insert(c.topology_key).second) return "duplicate topologyKey among DoNotSchedule constraints (rejected by API validation)";
      bool invalid = false;
      selector_matches(c.selector, t.labels, &invalid);
      if (invalid) return "invalid labelSelector in topologySpreadConstraints (PodTopologySpread.PreFilter would error)";
    }
    return "";
  }
  // Encodes a template against the EXISTING dictionaries. Returns false when it needs an entry that is not there yet
  // (a new selector requirement, scalar resource, host port, topology key or count class) or is not acceptable at all:
  // the caller then rebuilds the dictionaries.
  bool encode_spec_if_covered(const PodTemplate& t, EncodedSpec* spec, std::vector<uint64_t>* wanted) const {
    if (!template_error(t).empty() || unsupported.count(&t)) return false;
    // (a template with required anti-affinity terms is a future EXISTING pod: the asks it matches need its count classes)
    if (!node_pod_known(t)) return false;
    missing_ = false;
    try {
      *spec = encode_spec(t);
    } catch (const std::out_of_range&) {
      return false;  // topology key / count class not in the dictionaries
    }
    for (auto& kv : t.requests)
      if (kv.second > 0 && is_scalar_resource_name(kv.first) && !scalar_ix_.count(kv.first)) missing_ = true;
    for (const HostPort& hp : template_host_ports(t))
      if (!port_ix_.count(port_key(hp))) missing_ = true;
    // a toleration list that tells two taints of one bit apart needs new bits: the dictionaries are rebuilt
    for (auto& members : taint_members) {
      bool any = false, all = true;
      for (int32_t i : members) {
        const bool tol = tolerates(t.tolerations, taint_dict[(size_t)i]);
        any = any || tol;
        all = all && tol;
      }
      if (any && !all) missing_ = true;
    }
    wanted->assign((size_t)std::max(KP, 1), 0);
    encode_wanted_ports(t, wanted->data());
    return !missing_;
  }

  // What NodeInfo.AddPod of a pod of each template adds to its node besides resources (ykpred_spec_effects_t): its contribution
  // to every count class — class_contribution, the function encode_node_spread sums over the pods of a node — and the dictionary
  // host ports it occupies (encode_ports of the pod alone). A class is only tried against the templates that can match it: most
  // classes are a matchLabels selector, so they hang under one of their (key, value) pairs and a template looks under its own
  // labels (plus the few classes without such a pair, and — for templates with required anti-affinity — the symmetric classes).
  void spec_effects(const std::vector<PodTemplate*>& templates, std::vector<int32_t>* off, std::vector<int32_t>* cls, std::vector<int32_t>* cnt,
                    std::vector<uint64_t>* occupied) const {
    std::unordered_map<std::string, std::vector<int>> by_pair;
    std::vector<int> general, symmetric;
    for (int c = 0; c < KS; ++c) {
      const SelectorClass& sc = sel_classes[(size_t)c];
      if (sc.kind == SelectorClass::kExistingAnti) {
        symmetric.push_back(c);
        continue;
      }
      const LabelSelector* sel = sc.kind == SelectorClass::kSpread ? &sc.selector : (sc.terms.empty() ? nullptr : &sc.terms[0].selector);
      if (sel && sel->present && !sel->match_labels.empty()) {
        auto& kv = *sel->match_labels.begin();
        by_pair[kv.first + '\x1f' + kv.second].push_back(c);
      } else {
        general.push_back(c);
      }
    }
    off->assign(1, 0);
    cls->clear();
    cnt->clear();
    occupied->assign(templates.size() * (size_t)std::max(KP, 1), 0);
    std::vector<int> cand;
    for (size_t i = 0; i < templates.size(); ++i) {
      const PodTemplate& t = *templates[i];
      if (KS > 0) {
        cand = general;
        for (auto& kv : t.labels) {
          auto it = by_pair.find(kv.first + '\x1f' + kv.second);
          if (it != by_pair.end()) cand.insert(cand.end(), it->second.begin(), it->second.end());
        }
        if (!t.pod_anti_affinity.empty()) cand.insert(cand.end(), symmetric.begin(), symmetric.end());
        std::sort(cand.begin(), cand.end());
        for (int c : cand) {
          const int v = class_contribution(sel_classes[(size_t)c], t);
          if (v > 0) {
            cls->push_back(c);
            cnt->push_back(v);
          }
        }
      }
      off->push_back((int32_t)cls->size());
      if (KP > 0) {
        Pod probe;
        probe.tpl = &t;
        encode_ports({&probe}, occupied->data() + i * (size_t)KP);
      }
    }
  }

  // NodePorts: bit k = some pod in `pods` uses a host port that conflicts with dictionary port k
  // (HostPortInfo.CheckConflict: same protocol and port, equal or wildcard host IP).
  void encode_ports(const std::vector<const Pod*>& pods, uint64_t* bits) const {
    std::fill(bits, bits + KP, 0);
    if (port_dict.empty()) return;
    for (const Pod* p : pods) {
      std::vector<HostPort> used = template_host_ports(*p->tpl);
      if (used.empty()) continue;
      for (size_t k = 0; k < port_dict.size(); ++k)
        for (const HostPort& u : used)
          if (host_ports_conflict(port_dict[k], u)) {
            bits[k >> 6] |= 1ull << (k & 63);
            break;
          }
    }
  }
  void encode_wanted_ports(const PodTemplate& t, uint64_t* bits) const {
    std::fill(bits, bits + KP, 0);
    for (const HostPort& hp : template_host_ports(t)) {
      auto it = port_ix_.find(port_key(hp));
      if (it != port_ix_.end()) bits[it->second >> 6] |= 1ull << (it->second & 63);
    }
  }

  // True when every taint and scalar resource of the node object already has its dictionary entry, i.e. the node's row
  // can be re-encoded alone (label requirements are evaluated per node, topology values are checked by
  // encode_node_spread).
  bool node_known(const NodeInfo& ni) const {
    for (auto& t : ni.node.taints)
      if ((t.effect == "NoSchedule" || t.effect == "NoExecute") && !taint_ix_.count(taint_key(t))) return false;
    return true;  // (scalar resources are dimensions only when an ask requests them: a new one on a node changes nothing)
  }
  // True when accounting a pod of this template on a node needs no dictionary that does not exist yet (scalar resource
  // names, topology keys / count classes of its required anti-affinity terms). False → rebuild the dictionaries.
  bool node_pod_known(const PodTemplate& t) const {
    if (!t.pod_anti_affinity.empty() &&
        std::find(existing_anti_templates_.begin(), existing_anti_templates_.end(), &t) == existing_anti_templates_.end())
      return false;
    return true;
  }

  std::vector<int32_t> domain_sizes() const {
    std::vector<int32_t> out;
    for (auto& m : domain_ids) out.push_back((int32_t)m.size());
    return out;
  }
  // PodTopologySpread inputs of one node: domain id per topology key (-1 = label missing) and
  // countPodsMatchSelector per selector class. Returns false when a label value is not in the dictionary yet
  // (the caller must rebuild the dictionaries).
  bool encode_node_spread(const NodeInfo& ni, int32_t* domain, int32_t* selcount) {
    bool known = true;
    for (int k = 0; k < KD; ++k) {
      auto it = ni.node.labels.find(topo_keys[(size_t)k]);
      if (it == ni.node.labels.end()) {
        domain[k] = -1;
      } else {
        auto d = domain_ids[(size_t)k].find(it->second);
        if (d == domain_ids[(size_t)k].end()) {
          known = false;
          domain[k] = -1;
        } else {
          domain[k] = d->second;
        }
      }
    }
    for (int s = 0; s < KS; ++s) {
      const SelectorClass& sc = sel_classes[(size_t)s];
      int32_t c = 0;
      for (const Pod* p : ni.pods) {
        if (sc.kind == SelectorClass::kSpread && p->terminating) continue;  // countPodsMatchSelector skips terminating pods
        auto key = std::make_pair(p->tpl, s);
        auto m = sel_memo_.find(key);
        int contrib;
        if (m == sel_memo_.end()) {
          contrib = class_contribution(sc, *p->tpl);
          sel_memo_.emplace(key, contrib);
        } else {
          contrib = m->second;
        }
        c += contrib;
      }
      selcount[s] = c;
    }
    return known;
  }

  // ---- node rows -----------------------------------------------------------------------------------
  typedef std::unordered_map<std::string, std::vector<uint64_t>> LabelMemo;
  // memo: the label-tuple cache to use — null = the encoder's own (single-threaded callers); the parallel node loop of a full
  // encode hands every thread its own (the cache only saves work, the words are the same)
  void encode_node(const NodeInfo& ni, int64_t* alloc, int64_t* requested, int32_t* allowed, int32_t* count, uint32_t* flags,
                   uint64_t* taints, uint64_t* labels, LabelMemo* memo_in = nullptr) const {
    LabelMemo& label_memo = memo_in ? *memo_in : label_memo_;
    std::fill(alloc, alloc + R, 0);
    std::fill(requested, requested + R, 0);
    alloc[0] = ni.allocatable.milli_cpu;
    alloc[1] = ni.allocatable.memory;
    alloc[2] = ni.allocatable.ephemeral;
    requested[0] = ni.requested.milli_cpu;
    requested[1] = ni.requested.memory;
    requested[2] = ni.requested.ephemeral;
    for (size_t i = 0; i < scalar_names.size(); ++i) {
      auto a = ni.allocatable.scalar.find(scalar_names[i]);
      auto u = ni.requested.scalar.find(scalar_names[i]);
      alloc[3 + i] = a == ni.allocatable.scalar.end() ? 0 : a->second;
      requested[3 + i] = u == ni.requested.scalar.end() ? 0 : u->second;
    }
    *allowed = clamp32(ni.allocatable.allowed_pods);
    *count = (int32_t)ni.pods.size();
    *flags = ni.node.unschedulable ? YKPRED_NODE_UNSCHEDULABLE : 0u;
    std::fill(taints, taints + KT, 0);
    for (auto& t : ni.node.taints) {
      if (t.effect != "NoSchedule" && t.effect != "NoExecute") continue;  // PreferNoSchedule is ignored by the Filter
      auto it = taint_ix_.find(taint_key(t));
      if (it == taint_ix_.end() || (size_t)it->second >= taint_bit.size()) continue;
      const int b = taint_bit[(size_t)it->second];
      taints[b >> 6] |= 1ull << (b & 63);
    }
    // Requirement bits. Label requirements depend only on the node's values for the few label keys the dictionary
    // mentions, and most nodes share those values (zones, instance types ...): the words are computed once per distinct
    // value tuple; the requirements on the node NAME (matchFields) are evaluated per node.
    std::string sig;
    for (const std::string& key : label_keys_) {
      auto it = ni.node.labels.find(key);
      if (it == ni.node.labels.end()) {
        sig.push_back('\x01');
      } else {
        sig.push_back('\x02');
        sig += it->second;
      }
      sig.push_back('\x1f');
    }
    auto memo = label_memo.find(sig);
    if (memo == label_memo.end()) {
      std::vector<uint64_t> words((size_t)W, 0);
      for (int q : label_reqs_)
        if (req_dict[(size_t)q].eval(ni.node)) words[(size_t)q >> 6] |= 1ull << (q & 63);
      memo = label_memo.emplace(std::move(sig), std::move(words)).first;
    }
    std::copy(memo->second.begin(), memo->second.end(), labels);
    if (ni.node.name.empty()) {  // matchFields are not consulted for a nameless node (DictReq::eval)
      for (int q : name_reqs_)
        if (req_dict[(size_t)q].eval(ni.node)) labels[q >> 6] |= 1ull << (q & 63);
    } else {
      for (int q : name_other_)
        if (req_dict[(size_t)q].eval(ni.node)) labels[q >> 6] |= 1ull << (q & 63);
      auto hit = name_equals_.find(ni.node.name);  // "metadata.name In [x]" / NodeNames entries naming this node
      if (hit != name_equals_.end())
        for (int q : hit->second) labels[q >> 6] |= 1ull << (q & 63);
    }
  }

  // ---- spec rows -----------------------------------------------------------------------------------
  EncodedSpec encode_spec(const PodTemplate& t) const {
    EncodedSpec s;
    s.req.assign((size_t)R, 0);
    if (unsupported.count(&t)) {  // a row that fits nowhere, flagged so that the host routes the ask to the CPU manager
      s.tol.assign((size_t)KT, 0);
      s.flags = YKPRED_SPEC_UNSUPPORTED | YKPRED_SPEC_AFFINITY_SKIP;
      s.terms.push_back(std::vector<uint64_t>((size_t)W, 0));
      return s;
    }
    for (auto& kv : t.requests) {
      if (kv.first == "cpu")
        s.req[0] = kv.second;
      else if (kv.first == "memory")
        s.req[1] = kv.second;
      else if (kv.first == "ephemeral-storage")
        s.req[2] = kv.second;
      else {
        auto it = scalar_ix_.find(kv.first);
        if (it != scalar_ix_.end()) s.req[3 + (size_t)it->second] = kv.second;
      }
    }
    s.tol.assign((size_t)KT, 0);
    // one representative per bit: the members of a bit are told apart by no toleration list the dictionaries were built for
    // (assign_taint_bits), and encode_spec_if_covered refuses a later list that would
    for (size_t b = 0; b < taint_members.size(); ++b)
      if (!taint_members[b].empty() && tolerates(t.tolerations, taint_dict[(size_t)taint_members[b][0]])) s.tol[b >> 6] |= 1ull << (b & 63);
    Taint unsched{"node.kubernetes.io/unschedulable", "", "NoSchedule"};
    for (auto& tol : t.tolerations)
      if (toleration_tolerates(tol, unsched)) s.flags |= YKPRED_SPEC_TOLERATES_UNSCHEDULABLE;

    // NodeAffinity.PreFilter: Skip when there is neither a nodeSelector nor required affinity
    if (!t.has_required && !t.has_node_selector) s.flags |= YKPRED_SPEC_AFFINITY_SKIP;
    // Filter DNF: nodeSelector pairs are ANDed into every term
    std::vector<uint64_t> base((size_t)W, 0);
    for (auto& kv : t.node_selector) set_bit(base, find_req(DictReq::kEquals, kv.first, "", {kv.second}));
    if (!t.has_required) {
      s.terms.push_back(base);
    } else {
      for (auto& term : t.terms) {
        if (term.exprs.empty() && term.fields.empty()) continue;  // empty term selects no objects
        bool valid = true;
        std::vector<uint64_t> m = base;
        for (auto& e : term.exprs) {
          if (!valid_label_requirement(e)) {
            valid = false;
            break;
          }
          set_bit(m, find_req(DictReq::kLabel, e.key, e.op, canonical_values(e)));
        }
        for (auto& f : term.fields) {
          if (!valid) break;
          if ((f.op != "In" && f.op != "NotIn") || f.values.size() != 1) {
            valid = false;
            break;
          }
          set_bit(m, find_req(DictReq::kField, f.key, f.op, f.values));
        }
        if (valid) s.terms.push_back(std::move(m));
      }
    }
    // PreFilter NodeNames: union over terms of the intersection of their metadata.name In sets,
    // only when EVERY term carries such a requirement.
    if (t.has_required && !t.terms.empty()) {
      std::set<std::string> names;
      bool all_terms = true;
      for (auto& term : t.terms) {
        bool term_set = false;
        std::set<std::string> tn;
        for (auto& f : term.fields) {
          if (f.key != "metadata.name" || f.op != "In") continue;
          std::set<std::string> v(f.values.begin(), f.values.end());
          if (!term_set) {
            tn = v;
            term_set = true;
          } else {
            std::set<std::string> inter;
            for (auto& x : tn)
              if (v.count(x)) inter.insert(x);
            tn.swap(inter);
          }
        }
        if (!term_set) {
          all_terms = false;
          break;
        }
        names.insert(tn.begin(), tn.end());
      }
      if (all_terms) {
        if (names.empty()) {
          s.flags |= YKPRED_SPEC_PREFILTER_REJECT;
        } else {
          s.flags |= YKPRED_SPEC_PREFILTER_NAMES;
          for (auto& nm : names) {
            std::vector<uint64_t> m((size_t)W, 0);
            set_bit(m, find_req(DictReq::kNameIn, "", "", {nm}));
            s.pre_terms.push_back(std::move(m));
          }
        }
      }
    }
    for (auto& c : t.spread) {
      if (c.when_unsatisfiable != "DoNotSchedule") continue;
      ykpred_spread_t r{};
      r.topology_key = topo_ix_.at(c.topology_key);
      r.selector_class = -1;
      if (!selector_counts_nothing(c.selector)) r.selector_class = sel_ix_.at("S|" + selector_key(t.ns, c.selector));
      r.max_skew = c.max_skew;
      r.min_domains = c.has_min_domains ? c.min_domains : 1;
      bool invalid = false;
      r.self_match = selector_matches(c.selector, t.labels, &invalid) ? 1 : 0;
      r.flags = (c.node_affinity_policy == "Honor" ? YKPRED_SPREAD_HONOR_AFFINITY : 0u) |
                (c.node_taints_policy == "Honor" ? YKPRED_SPREAD_HONOR_TAINTS : 0u);
      r.kind = YKPRED_CONSTRAINT_SPREAD;
      s.spread.push_back(r);
    }
    // InterPodAffinity rules (after the spread constraints: Filter order)
    if (!t.pod_affinity.empty()) {
      int self = 1;  // podMatchesAllAffinityTerms(terms, the pod itself)
      for (auto& term : t.pod_affinity)
        if (!pod_term_matches(term, t.ns, t.ns, t.labels)) self = 0;
      int cls = sel_ix_.at("A|" + t.ns + '\x1f' + pod_terms_key(t.pod_affinity));
      for (auto& term : t.pod_affinity) {
        ykpred_spread_t r{};
        r.kind = YKPRED_CONSTRAINT_POD_AFFINITY;
        r.topology_key = topo_ix_.at(term.topology_key);
        r.selector_class = cls;
        r.self_match = self;
        s.spread.push_back(r);
      }
    }
    for (auto& term : t.pod_anti_affinity) {
      ykpred_spread_t r{};
      r.kind = YKPRED_CONSTRAINT_POD_ANTI_AFFINITY;
      r.topology_key = topo_ix_.at(term.topology_key);
      r.selector_class = sel_ix_.at("B|" + t.ns + '\x1f' + pod_terms_key({term}));
      s.spread.push_back(r);
    }
    {
      std::set<std::string> keys;  // one symmetry rule per topology key on which some existing term matches this pod
      for (const PodTemplate* et : existing_anti_templates_)
        for (auto& term : et->pod_anti_affinity)
          if (pod_term_matches(term, et->ns, t.ns, t.labels)) keys.insert(term.topology_key);
      for (auto& k : keys) {
        ykpred_spread_t r{};
        r.kind = YKPRED_CONSTRAINT_EXISTING_ANTI_AFFINITY;
        r.topology_key = topo_ix_.at(k);
        r.selector_class = sel_ix_.at("E|" + t.ns + '\x1f' + labels_key(t.labels) + '\x1f' + k);
        s.spread.push_back(r);
      }
    }
    return s;
  }

 private:
  struct MemoHash {
    size_t operator()(const std::pair<const PodTemplate*, int>& k) const {
      return std::hash<const void*>()(k.first) * 31u + (size_t)k.second;
    }
  };
  std::unordered_map<std::string, int> scalar_ix_, taint_ix_, req_ix_, topo_ix_, sel_ix_, port_ix_;
  mutable bool missing_ = false;  // set by a dictionary lookup that found nothing (see encode_spec_if_covered)
  std::vector<std::string> label_keys_;  // distinct label keys of the label requirements (sorted)
  std::vector<int> label_reqs_, name_reqs_;  // dictionary indices: requirements on labels / on the node name
  std::vector<int> name_other_;              // name requirements that are not "name == x" (NotIn, other field keys)
  std::unordered_map<std::string, std::vector<int>> name_equals_;  // node name → requirements that hold exactly there
  mutable LabelMemo label_memo_;  // label-value tuple → requirement words
  std::vector<const PodTemplate*> existing_anti_templates_;  // distinct templates of on-node pods that carry anti-affinity terms
  std::vector<const PodAffinityTerm*> wild_anti_terms_;  // anti-affinity terms of on-node pods the engine cannot decide
  bool is_wild(const PodAffinityTerm* t) const { return std::find(wild_anti_terms_.begin(), wild_anti_terms_.end(), t) != wild_anti_terms_.end(); }
  static bool tolerates(const std::vector<Toleration>& tols, const Taint& taint) {
    for (auto& tol : tols)
      if (toleration_tolerates(tol, taint)) return true;
    return false;
  }
  // Bits for the taint dictionary: taints tolerated by exactly the same toleration lists (over every pending template) share a
  // bit. Keyed tolerations are matched through an index by taint key, so the cost is (#lists x their tolerations + #taints),
  // not #lists x #taints; a toleration without a key (operator Exists) covers every taint of its effect and never tells two
  // taints of one effect apart — the effect is part of the group key instead.
  // lists_of: the templates whose toleration lists are read (one per dictionary shape suffices); templates: all of them
  // prepared (optional): lists_of[i]'s toleration-list key sits in (*prepared)[i].tol_key
  void assign_taint_bits(const std::vector<PodTemplate*>& lists_of, const std::vector<PodTemplate*>& templates, std::vector<Prepared>* prepared = nullptr) {
    const size_t T = taint_dict.size();
    taint_bit.assign(T, 0);
    taint_members.clear();
    overflow_taint_bit = -1;
    if (T == 0) return;
    std::unordered_map<std::string, std::vector<int32_t>> by_key;
    for (size_t i = 0; i < T; ++i) by_key[taint_dict[i].key].push_back((int32_t)i);
    std::unordered_map<std::string, int32_t> list_ids;
    std::vector<const PodTemplate*> list_owner;
    std::vector<std::vector<int32_t>> tolerated_by(T);  // [taint] → ids of the lists that tolerate it through a keyed toleration
    for (size_t li = 0; li < lists_of.size(); ++li) {
      const PodTemplate* t = lists_of[li];
      if (t->tolerations.empty() || (!unsupported.empty() && unsupported.count(t))) continue;
      std::string k = prepared ? std::move((*prepared)[li].tol_key) : toleration_list_key(*t);
      auto ins = list_ids.emplace(std::move(k), (int32_t)list_ids.size());
      if (!ins.second) continue;
      list_owner.push_back(t);
      std::set<int32_t> hit;
      for (auto& tol : t->tolerations) {
        if (tol.key.empty()) {
          if (tol.op == "Exists") continue;  // every taint of its effect: tells no two taints of one effect apart
          // (API validation wants Exists with an empty key; ToleratesTaint itself would compare values — follow the function)
          for (size_t i = 0; i < T; ++i)
            if (toleration_tolerates(tol, taint_dict[i])) hit.insert((int32_t)i);
          continue;
        }
        auto it = by_key.find(tol.key);
        if (it == by_key.end()) continue;
        for (int32_t i : it->second)
          if (toleration_tolerates(tol, taint_dict[(size_t)i])) hit.insert(i);
      }
      for (int32_t i : hit) tolerated_by[(size_t)i].push_back(ins.first->second);
    }
    std::unordered_map<std::string, int32_t> groups;
    for (size_t i = 0; i < T; ++i) {
      std::string g = taint_dict[i].effect + '\x1f';
      for (int32_t l : tolerated_by[i]) g += std::to_string(l) + ',';
      auto ins = groups.emplace(std::move(g), (int32_t)taint_members.size());
      int32_t bit = ins.first->second;
      if (ins.second) {
        if (overflow_taint_bit < 0 && (int)taint_members.size() < kLimitTaintBits - 1) {
          taint_members.emplace_back();  // a bit of its own
        } else {
          if (overflow_taint_bit < 0) {  // the engine's last bit takes this and every further group
            overflow_taint_bit = (int32_t)taint_members.size();
            taint_members.emplace_back();
          }
          bit = overflow_taint_bit;
          ins.first->second = bit;
        }
      }
      taint_bit[i] = bit;
      taint_members[(size_t)bit].push_back((int32_t)i);
    }
    if (overflow_taint_bit >= 0)  // mixed groups share the last bit: a list that tolerates some but not all of them cannot be encoded
      for (const PodTemplate* t : templates) {
        if (unsupported.count(t)) continue;
        bool any = false, all = true;
        for (int32_t i : taint_members[(size_t)overflow_taint_bit]) {
          const bool tol = tolerates(t->tolerations, taint_dict[(size_t)i]);
          any = any || tol;
          all = all && tol;
        }
        if (any && !all) unsupported[t] = "the ask tolerates some but not all of the taints that share the dictionary's overflow bit (more than 256 distinguishable taint groups)";
      }
  }
  static std::string port_key(const HostPort& h) { return h.protocol + '\x1f' + h.ip + '\x1f' + std::to_string(h.port); }
  std::unordered_map<std::pair<const PodTemplate*, int>, int, MemoHash> sel_memo_;
  // how much one pod with template `t` on a node adds to count class `sc`
  static int class_contribution(const SelectorClass& sc, const PodTemplate& t) {
    switch (sc.kind) {
      case SelectorClass::kSpread: {
        if (t.ns != sc.ns) return 0;
        bool invalid = false;
        return selector_matches(sc.selector, t.labels, &invalid) ? 1 : 0;
      }
      case SelectorClass::kAffinityAll:  // podMatchesAllAffinityTerms(incoming terms, existing pod)
        for (auto& term : sc.terms)
          if (!pod_term_matches(term, sc.ns, t.ns, t.labels)) return 0;
        return sc.terms.empty() ? 0 : 1;
      case SelectorClass::kAntiTerm: return pod_term_matches(sc.terms[0], sc.ns, t.ns, t.labels) ? 1 : 0;
      case SelectorClass::kExistingAnti: {  // the EXISTING pod's anti-affinity terms on this key that match the incoming pod
        int c = 0;
        for (auto& term : t.pod_anti_affinity)
          if (term.topology_key == sc.topology_key && pod_term_matches(term, t.ns, sc.ns, sc.labels)) ++c;
        return c;
      }
    }
    return 0;
  }
  int count_class(const std::string& key, SelectorClass&& sc) {
    auto it = sel_ix_.find(key);
    if (it != sel_ix_.end()) return it->second;
    int id = (int)sel_classes.size();
    sel_ix_.emplace(key, id);
    sel_keys_.push_back(key);
    sel_classes.push_back(std::move(sc));
    return id;
  }
  // dictionary sizes before one template's entries are added, and their removal again
  struct Mark {
    size_t scalars, reqs, topo, sel, ports;
  };
  Mark mark_now() const { return Mark{scalar_names.size(), req_dict.size(), topo_keys.size(), sel_classes.size(), port_dict.size()}; }
  void rollback(const Mark& m) {
    while (scalar_names.size() > m.scalars) {
      scalar_ix_.erase(scalar_names.back());
      scalar_names.pop_back();
    }
    while (req_dict.size() > m.reqs) {
      req_ix_.erase(req_keys_.back());
      req_keys_.pop_back();
      req_dict.pop_back();
    }
    while (topo_keys.size() > m.topo) {
      topo_ix_.erase(topo_keys.back());
      topo_keys.pop_back();
    }
    while (sel_classes.size() > m.sel) {
      sel_ix_.erase(sel_keys_.back());
      sel_keys_.pop_back();
      sel_classes.pop_back();
    }
    while (port_dict.size() > m.ports) {
      port_ix_.erase(port_key(port_dict.back()));
      port_dict.pop_back();
    }
  }
  std::vector<std::string> sel_keys_, req_keys_;  // keys of sel_classes / req_dict entries, in order (for rollback)
  void topo_key(const std::string& k) {
    if (topo_ix_.emplace(k, (int)topo_keys.size()).second) topo_keys.push_back(k);
  }
  static std::string labels_key(const StrMap& m) {
    std::string k;
    for (auto& kv : m) k += kv.first + '=' + kv.second + '\x1e';
    return k;
  }

  bool fail(const std::string& m) {
    error = m;
    return false;
  }
  static int32_t clamp32(int64_t v) {
    if (v > std::numeric_limits<int32_t>::max()) return std::numeric_limits<int32_t>::max();
    if (v < std::numeric_limits<int32_t>::min()) return std::numeric_limits<int32_t>::min();
    return (int32_t)v;
  }
  void scalar(const std::string& n) {
    if (scalar_ix_.emplace(n, (int)scalar_names.size()).second) scalar_names.push_back(n);
  }
  static std::string taint_key(const Taint& t) { return t.key + '\x1f' + t.value + '\x1f' + t.effect; }
  void taint(const Taint& t) {
    // (a cluster's nodes carry the same handful of taints: while the dictionary is that small, compare before building a key)
    if (taint_dict.size() <= 16)
      for (const Taint& d : taint_dict)
        if (d.key == t.key && d.value == t.value && d.effect == t.effect) return;
    if (taint_ix_.emplace(taint_key(t), (int)taint_dict.size()).second) taint_dict.push_back(t);
  }
  static std::vector<std::string> canonical_values(const Requirement& r) {
    std::vector<std::string> v = r.values;
    if (r.op == "In" || r.op == "NotIn") {
      std::sort(v.begin(), v.end());
      v.erase(std::unique(v.begin(), v.end()), v.end());
    }
    return v;
  }
  static std::string req_key(DictReq::Kind k, const std::string& key, const std::string& op, const std::vector<std::string>& values) {
    std::string s = std::to_string((int)k) + '\x1f' + key + '\x1f' + op;
    for (auto& v : values) s += '\x1f' + v;
    return s;
  }
  void add_req(DictReq::Kind k, const std::string& key, const std::string& op, const std::vector<std::string>& values) {
    add_req_item(ReqItem{k, key, op, values, req_key(k, key, op, values)});
  }
  void add_req_item(ReqItem&& it) {
    if (req_ix_.count(it.rk)) return;
    req_keys_.push_back(it.rk);
    req_ix_.emplace(std::move(it.rk), (int)req_dict.size());
    DictReq d;
    d.kind = it.kind;
    d.req.key = std::move(it.key);
    d.req.op = std::move(it.op);
    d.req.values = std::move(it.values);
    req_dict.push_back(std::move(d));
  }
  int find_req(DictReq::Kind k, const std::string& key, const std::string& op, const std::vector<std::string>& values) const {
    auto it = req_ix_.find(req_key(k, key, op, values));
    if (it == req_ix_.end()) missing_ = true;
    return it == req_ix_.end() ? -1 : it->second;
  }
  static void set_bit(std::vector<uint64_t>& m, int q) {
    if (q >= 0 && (size_t)(q >> 6) < m.size()) m[(size_t)(q >> 6)] |= 1ull << (q & 63);
  }
  // the selector requirements of a template in the order collect_requirements registers them (a pure function of the template)
  static std::vector<ReqItem> requirement_items(const PodTemplate& t) {
    std::vector<ReqItem> out;
    auto item = [&](DictReq::Kind k, const std::string& key, const std::string& op, std::vector<std::string> values) {
      std::string rk = req_key(k, key, op, values);
      out.push_back(ReqItem{k, key, op, std::move(values), std::move(rk)});
    };
    for (auto& kv : t.node_selector) item(DictReq::kEquals, kv.first, "", {kv.second});
    if (!t.has_required) return out;
    for (auto& term : t.terms) {
      for (auto& e : term.exprs)
        if (valid_label_requirement(e)) item(DictReq::kLabel, e.key, e.op, canonical_values(e));
      for (auto& f : term.fields) {
        if ((f.op == "In" || f.op == "NotIn") && f.values.size() == 1) item(DictReq::kField, f.key, f.op, f.values);
        if (f.key == "metadata.name" && f.op == "In")
          for (auto& v : f.values) item(DictReq::kNameIn, "", "", {v});
      }
    }
    return out;
  }
  void collect_requirements(const PodTemplate& t) {
    for (ReqItem& it : requirement_items(t)) add_req_item(std::move(it));
  }
};

}  // namespace ykh
