// jsonscan.h — allocation-free walking of JSON text, for the ingest path of the mirror.
//
// The Go manager hands every v1.Pod / v1.Node over as the text encoding/json produces. Most of such a
// document is of no interest to the predicates (a Node's status.images, conditions, addresses; a Pod's managedFields,
// ownerReferences, status.conditions ...), and the pods of one Deployment / task group differ in name and uid only. This
// scanner finds the members the mirror reads WITHOUT building a tree:
//   * skip(): steps over one JSON value (string-aware bracket matching);
//   * members(): iterates the (key, raw value range) pairs of an object;
//   * scan_pod(): name / uid / nodeName / phase / deletionTimestamp as raw ranges plus a TEMPLATE KEY — the raw text of
//     namespace, labels and every spec member except nodeName. Two pods with the same key have the same PodTemplate, so the
//     second one needs no parse at all (host.cpp keeps key -> template); a document that carries anything the key does not
//     cover (container statuses of an in-place resize, escaped strings in the captured fields) says so and takes the full parser;
//   * reduce_node(): a Node document cut down to what read_node reads (name, labels, spec.taints / unschedulable,
//     status.allocatable) — the full parser then sees a few hundred bytes instead of tens of kilobytes.
// Malformed input makes the scanner give up (false); the caller falls back to the full parser, which reports the error.
#pragma once
#include <cstddef>
#include <cstring>
#include <string>

#if defined(__SSE2__)
#include <emmintrin.h>
#endif

namespace js {

struct Range {
  const char* b = nullptr;
  const char* e = nullptr;
  bool present() const { return b != nullptr; }
  size_t size() const { return (size_t)(e - b); }
  bool is(const char* lit) const { return present() && size() == strlen(lit) && memcmp(b, lit, size()) == 0; }
  bool is_null() const { return is("null"); }
  bool is_string() const { return present() && size() >= 2 && *b == '"'; }
  // contents of a string value without the quotes; only valid when has_escape() is false
  std::string str() const { return is_string() ? std::string(b + 1, e - 1) : std::string(); }
  bool has_escape() const { return present() && memchr(b, '\\', size()) != nullptr; }
};

inline const char* ws(const char* p, const char* end) {
  while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p;
  return p;
}
// end of the string that starts at p (p points at the opening quote); nullptr = unterminated
inline const char* skip_string(const char* p, const char* end) {
#if defined(__SSE2__)
  // sixteen bytes at a time while the buffer allows it: uids, quantities and label values are the long runs of a Pod document
  {
    const __m128i quote = _mm_set1_epi8('"'), backslash = _mm_set1_epi8('\\');
    const char* q = p + 1;
    while (q + 16 <= end) {
      const __m128i v = _mm_loadu_si128((const __m128i*)q);
      const unsigned m = (unsigned)_mm_movemask_epi8(_mm_or_si128(_mm_cmpeq_epi8(v, quote), _mm_cmpeq_epi8(v, backslash)));
      if (!m) {
        q += 16;
        continue;
      }
      q += __builtin_ctz(m);
      if (*q == '"') return q + 1;
      q += 2;  // an escape: the next character is part of the string whatever it is
    }
    p = q - 1;
  }
#endif
  for (++p; p < end; ++p) {
    if (*p == '\\') {
      ++p;
      continue;
    }
    if (*p == '"') return p + 1;
  }
  return nullptr;
}
// end of the value that starts at p; nullptr = malformed
inline const char* skip(const char* p, const char* end) {
  p = ws(p, end);
  if (p >= end) return nullptr;
  if (*p == '"') return skip_string(p, end);
  if (*p == '{' || *p == '[') {
    int depth = 0;
    while (p < end) {
      const char c = *p;
      if (c == '"') {
        p = skip_string(p, end);
        if (!p) return nullptr;
        continue;
      }
      if (c == '{' || c == '[') ++depth;
      if (c == '}' || c == ']') {
        if (--depth == 0) return p + 1;
      }
      ++p;
    }
    return nullptr;
  }
  const char* s = p;
  while (p < end && *p != ',' && *p != '}' && *p != ']' && *p != ' ' && *p != '\n' && *p != '\t' && *p != '\r') ++p;
  return p > s ? p : nullptr;
}

// Iterates the members of the object at [b, e): f(key range WITHOUT quotes, value range). Returns false on malformed input
// or when f returns false.
template <class F>
inline bool members(Range obj, F&& f) {
  const char* p = ws(obj.b, obj.e);
  if (p >= obj.e || *p != '{') return false;
  p = ws(p + 1, obj.e);
  if (p < obj.e && *p == '}') return true;
  for (;;) {
    p = ws(p, obj.e);
    if (p >= obj.e || *p != '"') return false;
    const char* ke = skip_string(p, obj.e);
    if (!ke) return false;
    Range key{p + 1, ke - 1};
    p = ws(ke, obj.e);
    if (p >= obj.e || *p != ':') return false;
    p = ws(p + 1, obj.e);
    const char* ve = skip(p, obj.e);
    if (!ve) return false;
    if (!f(key, Range{p, ve})) return false;
    p = ws(ve, obj.e);
    if (p < obj.e && *p == ',') {
      ++p;
      continue;
    }
    return p < obj.e && *p == '}';
  }
}
inline bool key_is(Range k, const char* lit) { return k.size() == strlen(lit) && memcmp(k.b, lit, k.size()) == 0; }

// Like members(), for an object that starts at p (p points at '{') and whose end is not known yet: returns the position behind
// its closing brace, nullptr on malformed input or when f returns false. One pass: the walk over the members IS the skip.
template <class F>
inline const char* members_from(const char* p, const char* end, F&& f) {
  if (p >= end || *p != '{') return nullptr;
  p = ws(p + 1, end);
  if (p < end && *p == '}') return p + 1;
  for (;;) {
    p = ws(p, end);
    if (p >= end || *p != '"') return nullptr;
    const char* ke = skip_string(p, end);
    if (!ke) return nullptr;
    Range key{p + 1, ke - 1};
    p = ws(ke, end);
    if (p >= end || *p != ':') return nullptr;
    p = ws(p + 1, end);
    const char* ve = skip(p, end);
    if (!ve) return nullptr;
    if (!f(key, Range{p, ve})) return nullptr;
    p = ws(ve, end);
    if (p < end && *p == ',') {
      ++p;
      continue;
    }
    return p < end && *p == '}' ? p + 1 : nullptr;
  }
}

struct PodScan {
  Range name, uid, node_name, phase;
  bool terminating = false;
  bool needs_full_parse = false;  // something the template key does not cover
  std::string key;                // namespace | labels | the text of spec without its nodeName member (raw text)
};

// One pass over the document: the top-level loop walks into metadata / spec / status instead of skipping them first and
// iterating them afterwards (the first form read every byte of the document three times: 2.9 ns per byte on one core).
inline bool scan_pod(Range doc, PodScan* out) {
  const char* p = ws(doc.b, doc.e);
  const char* const end = doc.e;
  if (p >= end || *p != '{') return false;
  Range ns, labels, spec;
  Range node_member{};  // spec's nodeName member with its key and, when it is not the last member, its comma
  bool ok = true;
  auto meta_f = [&](Range k, Range v) {
    if (key_is(k, "name")) out->name = v;
    else if (key_is(k, "uid")) out->uid = v;
    else if (key_is(k, "namespace")) ns = v;
    else if (key_is(k, "labels")) labels = v;
    else if (key_is(k, "deletionTimestamp")) out->terminating = !v.is_null();
    return true;
  };
  auto spec_f = [&](Range k, Range v) {
    if (key_is(k, "nodeName")) {
      out->node_name = v;
      const char* e = ws(v.e, end);
      node_member = Range{k.b - 1, (e < end && *e == ',') ? e + 1 : v.e};
    }
    return true;
  };
  auto status_f = [&](Range k, Range v) {
    if (key_is(k, "phase")) out->phase = v;
    // in-place resize inputs change the request vector (fold_container_statuses): full parser
    else if ((key_is(k, "containerStatuses") || key_is(k, "initContainerStatuses") || key_is(k, "resize")) && !v.is_null())
      out->needs_full_parse = true;
    else if (key_is(k, "conditions") && v.size() > 16 && memmem(v.b, v.size(), "PodResizePending", 16) != nullptr)
      out->needs_full_parse = true;
    return true;
  };
  // value of a top-level member: an object of one of the three kinds is walked, anything else skipped
  // A key that occurs twice is legal JSON with last-wins meaning in the tree parser; the one-pass form would merge the fields of
  // two `metadata` objects and cut a `nodeName` member found in an EARLIER `spec` out of the LAST one: such a
  // document goes to the full parser, and the nodeName state always belongs to the spec object being walked.
  int seen[3] = {0, 0, 0};
  auto value = [&](const char* q, int kind) -> const char* {
    if (kind < 3 && ++seen[kind] > 1) out->needs_full_parse = true;
    if (q < end && *q == '{') {
      if (kind == 0) return members_from(q, end, meta_f);
      if (kind == 1) {
        node_member = Range{};
        out->node_name = Range{};
        const char* e = members_from(q, end, spec_f);
        if (e) spec = Range{q, e};
        return e;
      }
      if (kind == 2) return members_from(q, end, status_f);
    }
    return skip(q, end);
  };
  p = ws(p + 1, end);
  if (!(p < end && *p == '}')) {
    for (;;) {
      p = ws(p, end);
      if (p >= end || *p != '"') return false;
      const char* ke = skip_string(p, end);
      if (!ke) return false;
      const Range k{p + 1, ke - 1};
      p = ws(ke, end);
      if (p >= end || *p != ':') return false;
      p = ws(p + 1, end);
      const int kind = key_is(k, "metadata") ? 0 : key_is(k, "spec") ? 1 : key_is(k, "status") ? 2 : 3;
      const char* ve = value(p, kind);
      if (!ve) return false;
      p = ws(ve, end);
      if (p < end && *p == ',') {
        ++p;
        continue;
      }
      if (!(p < end && *p == '}')) return false;
      break;
    }
  }
  (void)ok;
  out->key.clear();
  out->key.reserve(ns.size() + labels.size() + spec.size() + 4);
  if (ns.present()) out->key.append(ns.b, ns.size());
  out->key.push_back('\x1f');
  if (labels.present()) out->key.append(labels.b, labels.size());
  out->key.push_back('\x1f');
  if (spec.present()) {
    if (node_member.present() && node_member.b >= spec.b && node_member.e <= spec.e) {
      out->key.append(spec.b, (size_t)(node_member.b - spec.b));
      out->key.append(node_member.e, (size_t)(spec.e - node_member.e));
    } else {
      out->key.append(spec.b, spec.size());
    }
  }
  for (const Range* r : {&out->name, &out->uid, &out->node_name, &out->phase})
    if (r->present() && !r->is_null() && (!r->is_string() || r->has_escape())) out->needs_full_parse = true;
  return true;
}

// {"metadata":{"name":..,"labels":..},"spec":{"taints":..,"unschedulable":..},"status":{"allocatable":..}} from a Node document
inline bool reduce_node(Range doc, std::string* out) {
  Range meta, spec, status;
  if (!members(doc, [&](Range k, Range v) {
        if (key_is(k, "metadata")) meta = v;
        else if (key_is(k, "spec")) spec = v;
        else if (key_is(k, "status")) status = v;
        return true;
      }))
    return false;
  auto pick = [&](std::string& dst, Range obj, const char* a, const char* b) {
    dst.push_back('{');
    bool first = true, ok = true;
    if (obj.present() && !obj.is_null())
      ok = members(obj, [&](Range k, Range v) {
        if (key_is(k, a) || (b && key_is(k, b))) {
          if (!first) dst.push_back(',');
          first = false;
          dst.push_back('"');
          dst.append(k.b, k.size());
          dst.append("\":");
          dst.append(v.b, v.size());
        }
        return true;
      });
    dst.push_back('}');
    return ok;
  };
  std::string acc;
  acc.reserve(1024);
  acc = "{\"metadata\":";
  if (!pick(acc, meta, "name", "labels")) return false;
  acc += ",\"spec\":";
  if (!pick(acc, spec, "taints", "unschedulable")) return false;
  acc += ",\"status\":";
  if (!pick(acc, status, "allocatable", nullptr)) return false;
  acc += "}";
  out->swap(acc);
  return true;
}

// Splits a buffer of concatenated / newline-separated JSON documents: f(document range). Returns the number of documents,
// or -1 - (documents before the malformed one).
template <class F>
inline long documents(const char* p, const char* end, F&& f) {
  long n = 0;
  for (;;) {
    p = ws(p, end);
    while (p < end && *p == ',') p = ws(p + 1, end);  // (a JSON array body works as well)
    if (p >= end) return n;
    const char* e = skip(p, end);
    if (!e) return -1 - n;
    f(Range{p, e});
    ++n;
    p = e;
  }
}

}  // namespace js
