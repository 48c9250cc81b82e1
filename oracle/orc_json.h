// orc_json.h — the parity ORACLE's own JSON reader (test infrastructure only).
//
// Written independently of the product's reader (yunikorn-k8shim_amd/csrc/host/minijson.h) on purpose: both sides of every
// parity test parse the same snapshot text, and a defect shared by one common parser would be invisible to all of them.
// Different construction here: an iterative parser with an explicit container stack over raw character pointers, objects
// held in an ordered map (a repeated key keeps the LAST value, like encoding/json), numbers kept as their literal text
// after a grammar check, strings decoded by a table-driven unescaper.
//
// An absent key and an explicit null stay distinguishable (the reference distinguishes a nil slice / pointer from an
// empty one, predicate_manager_test.go:550-607): find() returns nullptr for absent, at() for absent-or-null.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace oj {

struct Node;
using NodePtr = std::unique_ptr<Node>;

struct Node {
  enum Type : unsigned char { kNull, kTrue, kFalse, kNumber, kText, kList, kDict };
  Type type = kNull;
  std::string s;                        // kText payload / literal text of a kNumber
  std::vector<NodePtr> arr;             // kList
  std::map<std::string, NodePtr> obj;   // kDict

  bool is_null() const { return type == kNull; }
  bool is_obj() const { return type == kDict; }
  bool is_arr() const { return type == kList; }
  bool is_str() const { return type == kText; }
  bool is_num() const { return type == kNumber; }

  const Node* find(const std::string& key) const {
    if (type != kDict) return nullptr;
    auto it = obj.find(key);
    return it == obj.end() ? nullptr : it->second.get();
  }
  // absent or null → nullptr
  const Node* get_nn(const std::string& key) const {
    const Node* n = find(key);
    return n && n->type != kNull ? n : nullptr;
  }
  std::string str_or(const std::string& key, const std::string& fallback) const {
    const Node* n = get_nn(key);
    return n && (n->type == kText || n->type == kNumber) ? n->s : fallback;
  }
  bool bool_or(const std::string& key, bool fallback) const {
    const Node* n = get_nn(key);
    if (!n || (n->type != kTrue && n->type != kFalse)) return fallback;
    return n->type == kTrue;
  }
  int64_t int_or(const std::string& key, int64_t fallback) const {
    const Node* n = get_nn(key);
    if (!n || (n->type != kText && n->type != kNumber)) return fallback;
    return std::strtoll(n->s.c_str(), nullptr, 10);
  }
};

namespace detail {

[[noreturn]] inline void bail(const char* what, const char* begin, const char* at) {
  throw std::runtime_error(std::string("orc_json: ") + what + " (byte " + std::to_string(at - begin) + ")");
}
inline const char* skip_blank(const char* p, const char* end) {
  while (p != end && (*p == ' ' || *p == '\t' || *p == '\n' || *p == '\r')) ++p;
  return p;
}
inline int hex_digit(char c) {
  if (c >= '0' && c <= '9') return c - '0';
  c |= 0x20;
  return c >= 'a' && c <= 'f' ? c - 'a' + 10 : -1;
}
inline void append_code_point(std::string* out, uint32_t cp) {
  static const struct { uint32_t below; int extra; unsigned char lead; } kForm[] = {
      {0x80, 0, 0x00}, {0x800, 1, 0xC0}, {0x10000, 2, 0xE0}, {0x110000, 3, 0xF0}};
  for (const auto& f : kForm)
    if (cp < f.below) {
      out->push_back(static_cast<char>(f.lead | (cp >> (6 * f.extra))));
      for (int i = f.extra - 1; i >= 0; --i) out->push_back(static_cast<char>(0x80 | ((cp >> (6 * i)) & 0x3F)));
      return;
    }
}
// p points just past the opening quote; returns just past the closing quote
inline const char* read_text(const char* begin, const char* p, const char* end, std::string* out) {
  static const char kSimple[][2] = {{'"', '"'}, {'\\', '\\'}, {'/', '/'}, {'b', '\b'}, {'f', '\f'}, {'n', '\n'}, {'r', '\r'}, {'t', '\t'}};
  for (;;) {
    const char* run = p;
    while (p != end && *p != '"' && *p != '\\') ++p;
    out->append(run, p);
    if (p == end) bail("unterminated string", begin, p);
    if (*p++ == '"') return p;
    if (p == end) bail("dangling backslash", begin, p);
    const char esc = *p++;
    bool done = false;
    for (const auto& s : kSimple)
      if (s[0] == esc) {
        out->push_back(s[1]);
        done = true;
      }
    if (done) continue;
    if (esc != 'u') bail("unknown escape", begin, p);
    auto four = [&]() {
      uint32_t v = 0;
      for (int i = 0; i < 4; ++i) {
        const int h = p != end ? hex_digit(*p) : -1;
        if (h < 0) bail("bad \\u escape", begin, p);
        v = v * 16 + static_cast<uint32_t>(h);
        ++p;
      }
      return v;
    };
    uint32_t cp = four();
    if (cp >= 0xD800 && cp < 0xDC00 && end - p >= 6 && p[0] == '\\' && p[1] == 'u') {
      p += 2;
      const uint32_t low = four();
      cp = 0x10000 + ((cp - 0xD800) << 10) + (low - 0xDC00);
    }
    append_code_point(out, cp);
  }
}
// JSON number grammar: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?   (a leading '+' is tolerated: quantities)
inline const char* read_number(const char* begin, const char* p, const char* end, std::string* out) {
  const char* start = p;
  if (p != end && (*p == '-' || *p == '+')) ++p;
  auto digits = [&]() {
    const char* d = p;
    while (p != end && *p >= '0' && *p <= '9') ++p;
    return p != d;
  };
  if (!digits()) bail("malformed number", begin, p);
  if (p != end && *p == '.') {
    ++p;
    if (!digits()) bail("malformed fraction", begin, p);
  }
  if (p != end && (*p == 'e' || *p == 'E')) {
    ++p;
    if (p != end && (*p == '-' || *p == '+')) ++p;
    if (!digits()) bail("malformed exponent", begin, p);
  }
  out->assign(start, p);
  return p;
}

}  // namespace detail

inline NodePtr parse(const std::string& text) {
  using namespace detail;
  const char* const begin = text.data();
  const char* const end = begin + text.size();
  const char* p = skip_blank(begin, end);
  NodePtr root;
  struct Frame {
    Node* container;
    std::string pending_key;
    bool have_key;
  };
  std::vector<Frame> stack;
  // attaches a finished value to the innermost open container (or makes it the root)
  auto attach = [&](NodePtr v) -> Node* {
    Node* raw = v.get();
    if (stack.empty()) {
      if (root) bail("more than one top-level value", begin, p);
      root = std::move(v);
    } else if (stack.back().container->type == Node::kList) {
      stack.back().container->arr.push_back(std::move(v));
    } else {
      if (!stack.back().have_key) bail("value without a key", begin, p);
      stack.back().container->obj[stack.back().pending_key] = std::move(v);
      stack.back().have_key = false;
    }
    return raw;
  };
  bool expect_value = true;  // false: a ',' or a closer must come next
  while (true) {
    p = skip_blank(p, end);
    if (p == end) break;
    const char c = *p;
    if (!expect_value) {
      if (stack.empty()) bail("trailing characters", begin, p);
      const bool list = stack.back().container->type == Node::kList;
      if (c == ',') {
        ++p;
        expect_value = true;
        if (!list) {  // a dict wants its next key right away
          p = skip_blank(p, end);
          if (p == end || *p != '"') bail("expected a key", begin, p);
          stack.back().pending_key.clear();
          p = read_text(begin, p + 1, end, &stack.back().pending_key);
          p = skip_blank(p, end);
          if (p == end || *p != ':') bail("expected ':'", begin, p);
          ++p;
          stack.back().have_key = true;
        }
        continue;
      }
      if ((list && c == ']') || (!list && c == '}')) {
        ++p;
        stack.pop_back();
        continue;
      }
      bail("expected ',' or a closing bracket", begin, p);
    }
    // a value starts here
    if (c == '[' || c == '{') {
      NodePtr v(new Node);
      v->type = c == '[' ? Node::kList : Node::kDict;
      Node* raw = attach(std::move(v));
      stack.push_back(Frame{raw, std::string(), false});
      ++p;
      p = skip_blank(p, end);
      if (p != end && ((c == '[' && *p == ']') || (c == '{' && *p == '}'))) {  // empty container
        ++p;
        stack.pop_back();
        expect_value = false;
        continue;
      }
      if (c == '{') {
        if (p == end || *p != '"') bail("expected a key", begin, p);
        p = read_text(begin, p + 1, end, &stack.back().pending_key);
        p = skip_blank(p, end);
        if (p == end || *p != ':') bail("expected ':'", begin, p);
        ++p;
        stack.back().have_key = true;
      }
      continue;  // still expecting a value (the first element / the first member's value)
    }
    NodePtr v(new Node);
    if (c == '"') {
      v->type = Node::kText;
      p = read_text(begin, p + 1, end, &v->s);
    } else if (end - p >= 4 && std::string(p, 4) == "true") {
      v->type = Node::kTrue;
      p += 4;
    } else if (end - p >= 5 && std::string(p, 5) == "false") {
      v->type = Node::kFalse;
      p += 5;
    } else if (end - p >= 4 && std::string(p, 4) == "null") {
      p += 4;
    } else {
      v->type = Node::kNumber;
      p = read_number(begin, p, end, &v->s);
    }
    attach(std::move(v));
    expect_value = false;
  }
  if (!stack.empty()) bail("unexpected end inside a container", begin, p);
  if (!root) bail("empty document", begin, p);
  return root;
}

}  // namespace oj
