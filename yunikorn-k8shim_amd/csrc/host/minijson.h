// minijson.h — tiny JSON reader used by the host library (libykhost).
//
// Reads the "cluster snapshot" documents described in INTEGRATION.md. Deliberately small:
// objects keep insertion order, numbers keep their source text (so 64-bit integers survive),
// and a missing key is distinguishable from an explicit null (the reference distinguishes a nil
// slice/pointer from an empty one, predicate_manager_test.go:550-607).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <memory>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

namespace mj {

struct Value;
using ValuePtr = std::shared_ptr<Value>;

enum class Kind { Null, Bool, Number, String, Array, Object };

struct Value {
  Kind kind = Kind::Null;
  bool b = false;
  std::string s;  // String payload, or the literal text of a Number
  std::vector<ValuePtr> arr;
  std::vector<std::pair<std::string, ValuePtr>> obj;

  bool is_null() const { return kind == Kind::Null; }
  bool is_obj() const { return kind == Kind::Object; }
  bool is_arr() const { return kind == Kind::Array; }
  bool is_str() const { return kind == Kind::String; }
  bool is_num() const { return kind == Kind::Number; }

  // nullptr when the key is absent (an explicit null yields a Value of Kind::Null).
  const Value* get(const std::string& key) const {
    if (kind != Kind::Object) return nullptr;
    // (a key that occurs twice: the LAST occurrence counts, as in every mainstream reader; the one-pass scanner of jsonscan.h
    // hands such documents to this parser)
    for (auto it = obj.rbegin(); it != obj.rend(); ++it)
      if (it->first == key) return it->second.get();
    return nullptr;
  }
  // Absent or null → nullptr.
  const Value* get_nn(const std::string& key) const {
    const Value* v = get(key);
    return (v && !v->is_null()) ? v : nullptr;
  }
  std::string str_or(const std::string& key, const std::string& dflt) const {
    const Value* v = get_nn(key);
    if (!v) return dflt;
    if (v->kind == Kind::String || v->kind == Kind::Number) return v->s;
    return dflt;
  }
  bool bool_or(const std::string& key, bool dflt) const {
    const Value* v = get_nn(key);
    return (v && v->kind == Kind::Bool) ? v->b : dflt;
  }
  int64_t int_or(const std::string& key, int64_t dflt) const {
    const Value* v = get_nn(key);
    if (!v) return dflt;
    if (v->kind == Kind::Number || v->kind == Kind::String) return std::strtoll(v->s.c_str(), nullptr, 10);
    return dflt;
  }
};

class Parser {
 public:
  explicit Parser(const std::string& text) : t_(text) {}
  ValuePtr parse() {
    ws();
    ValuePtr v = value();
    ws();
    if (p_ != t_.size()) fail("trailing characters");
    return v;
  }

 private:
  const std::string& t_;
  size_t p_ = 0;

  [[noreturn]] void fail(const char* why) const {
    throw std::runtime_error(std::string("minijson: ") + why + " at offset " + std::to_string(p_));
  }
  void ws() {
    while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) ++p_;
  }
  bool lit(const char* w) {
    size_t n = 0;
    while (w[n]) ++n;
    if (t_.compare(p_, n, w) == 0) {
      p_ += n;
      return true;
    }
    return false;
  }
  ValuePtr value() {
    if (p_ >= t_.size()) fail("unexpected end");
    auto v = std::make_shared<Value>();
    char c = t_[p_];
    if (c == '{') {
      v->kind = Kind::Object;
      ++p_;
      ws();
      if (p_ < t_.size() && t_[p_] == '}') {
        ++p_;
        return v;
      }
      for (;;) {
        ws();
        if (p_ >= t_.size() || t_[p_] != '"') fail("expected object key");
        std::string k = string();
        ws();
        if (p_ >= t_.size() || t_[p_] != ':') fail("expected ':'");
        ++p_;
        ws();
        v->obj.emplace_back(std::move(k), value());
        ws();
        if (p_ < t_.size() && t_[p_] == ',') {
          ++p_;
          continue;
        }
        if (p_ < t_.size() && t_[p_] == '}') {
          ++p_;
          break;
        }
        fail("expected ',' or '}'");
      }
      return v;
    }
    if (c == '[') {
      v->kind = Kind::Array;
      ++p_;
      ws();
      if (p_ < t_.size() && t_[p_] == ']') {
        ++p_;
        return v;
      }
      for (;;) {
        ws();
        v->arr.push_back(value());
        ws();
        if (p_ < t_.size() && t_[p_] == ',') {
          ++p_;
          continue;
        }
        if (p_ < t_.size() && t_[p_] == ']') {
          ++p_;
          break;
        }
        fail("expected ',' or ']'");
      }
      return v;
    }
    if (c == '"') {
      v->kind = Kind::String;
      v->s = string();
      return v;
    }
    if (lit("true")) {
      v->kind = Kind::Bool;
      v->b = true;
      return v;
    }
    if (lit("false")) {
      v->kind = Kind::Bool;
      v->b = false;
      return v;
    }
    if (lit("null")) return v;
    // number: keep literal text
    size_t s = p_;
    while (p_ < t_.size()) {
      char d = t_[p_];
      if ((d >= '0' && d <= '9') || d == '-' || d == '+' || d == '.' || d == 'e' || d == 'E')
        ++p_;
      else
        break;
    }
    if (s == p_) fail("unexpected character");
    v->kind = Kind::Number;
    v->s = t_.substr(s, p_ - s);
    return v;
  }
  static void put_utf8(std::string& out, unsigned cp) {
    if (cp < 0x80)
      out.push_back(static_cast<char>(cp));
    else if (cp < 0x800) {
      out.push_back(static_cast<char>(0xC0 | (cp >> 6)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    } else if (cp < 0x10000) {
      out.push_back(static_cast<char>(0xE0 | (cp >> 12)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    } else {
      out.push_back(static_cast<char>(0xF0 | (cp >> 18)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 12) & 0x3F)));
      out.push_back(static_cast<char>(0x80 | ((cp >> 6) & 0x3F)));
      out.push_back(static_cast<char>(0x80 | (cp & 0x3F)));
    }
  }
  unsigned hex4() {
    if (p_ + 4 > t_.size()) fail("bad \\u escape");
    unsigned v = 0;
    for (int i = 0; i < 4; ++i) {
      char h = t_[p_++];
      v <<= 4;
      if (h >= '0' && h <= '9')
        v |= static_cast<unsigned>(h - '0');
      else if (h >= 'a' && h <= 'f')
        v |= static_cast<unsigned>(h - 'a' + 10);
      else if (h >= 'A' && h <= 'F')
        v |= static_cast<unsigned>(h - 'A' + 10);
      else
        fail("bad hex digit");
    }
    return v;
  }
  std::string string() {
    std::string out;
    ++p_;  // opening quote
    for (;;) {
      if (p_ >= t_.size()) fail("unterminated string");
      char c = t_[p_++];
      if (c == '"') break;
      if (c != '\\') {
        out.push_back(c);
        continue;
      }
      if (p_ >= t_.size()) fail("bad escape");
      char e = t_[p_++];
      switch (e) {
        case '"': out.push_back('"'); break;
        case '\\': out.push_back('\\'); break;
        case '/': out.push_back('/'); break;
        case 'b': out.push_back('\b'); break;
        case 'f': out.push_back('\f'); break;
        case 'n': out.push_back('\n'); break;
        case 'r': out.push_back('\r'); break;
        case 't': out.push_back('\t'); break;
        case 'u': {
          unsigned cp = hex4();
          if (cp >= 0xD800 && cp <= 0xDBFF && p_ + 1 < t_.size() && t_[p_] == '\\' && t_[p_ + 1] == 'u') {
            p_ += 2;
            unsigned lo = hex4();
            cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          }
          put_utf8(out, cp);
          break;
        }
        default: fail("unknown escape");
      }
    }
    return out;
  }
};

inline ValuePtr parse(const std::string& text) { return Parser(text).parse(); }

}  // namespace mj
