// orc_quantity.h — Kubernetes resource.Quantity text → int64 for the parity ORACLE (test infrastructure only).
//
// Written independently of the product's converter (yunikorn-k8shim_amd/csrc/host/quantity.h, 128-bit mantissa arithmetic):
// this one works on DECIMAL DIGIT STRINGS — exact at any magnitude — so that a defect in one of them cannot hide behind
// the other. tests/test_quantity_fuzz.py drives both against Python's exact rationals.
//
// Semantics restated (k8s.io/apimachinery v0.36.1 resource.Quantity, not vendored under /root/reference; used by
// /root/reference/pkg/common/resource.go:273-285 — cpu → MilliValue(), everything else → Value()):
//   <quantity> ::= <signedNumber><suffix>,  <suffix> ::= Ki|Mi|Gi|Ti|Pi|Ei | n|u|m|""|k|M|G|T|P|E | e<int> | E<int>
//   Value() / MilliValue() round AWAY from zero when digits are dropped and saturate at ±MaxInt64.
// Known answers pinned by the reference's tests: "500M"+"1024M" = 1524e6, cpu "1"+"2" = 3000 milli, "0.5" = 500 milli,
// "5.12" = 5120 milli (resource_test.go:153-260).
#pragma once
#include <cstdint>
#include <limits>
#include <string>

namespace orc {
namespace qty {

// non-negative integers as decimal strings without leading zeros ("0" is the empty string)
inline void strip(std::string* d) {
  size_t z = 0;
  while (z < d->size() && (*d)[z] == '0') ++z;
  d->erase(0, z);
}
inline std::string doubled(const std::string& d) {
  std::string out(d.size() + 1, '0');
  int carry = 0;
  for (size_t i = d.size(); i-- > 0;) {
    int v = (d[i] - '0') * 2 + carry;
    out[i + 1] = static_cast<char>('0' + v % 10);
    carry = v / 10;
  }
  out[0] = static_cast<char>('0' + carry);
  strip(&out);
  return out;
}
inline std::string plus_one(std::string d) {
  for (size_t i = d.size(); i-- > 0;) {
    if (d[i] != '9') {
      ++d[i];
      return d;
    }
    d[i] = '0';
  }
  return "1" + d;
}
// |text| as (digits, power of ten): value = digits * 10^exp10. Returns false when the text is not a quantity.
inline bool split(const std::string& text, bool* negative, std::string* digits, long* exp10, int* pow2) {
  size_t i = 0;
  *negative = false;
  if (i < text.size() && (text[i] == '-' || text[i] == '+')) *negative = text[i++] == '-';
  std::string whole, frac;
  while (i < text.size() && text[i] >= '0' && text[i] <= '9') whole.push_back(text[i++]);
  if (i < text.size() && text[i] == '.') {
    ++i;
    while (i < text.size() && text[i] >= '0' && text[i] <= '9') frac.push_back(text[i++]);
  }
  if (whole.empty() && frac.empty()) return false;
  *digits = whole + frac;
  *exp10 = -static_cast<long>(frac.size());
  *pow2 = 0;
  const std::string suffix = text.substr(i);
  if (suffix.empty()) return true;
  static const struct { const char* s; int p2; int p10; } kSuffix[] = {
      {"Ki", 10, 0}, {"Mi", 20, 0}, {"Gi", 30, 0}, {"Ti", 40, 0}, {"Pi", 50, 0}, {"Ei", 60, 0},
      {"n", 0, -9},  {"u", 0, -6},  {"m", 0, -3},  {"k", 0, 3},   {"M", 0, 6},   {"G", 0, 9},
      {"T", 0, 12},  {"P", 0, 15},  {"E", 0, 18}};
  for (const auto& k : kSuffix)
    if (suffix == k.s) {
      *pow2 = k.p2;
      *exp10 += k.p10;
      return true;
    }
  if ((suffix[0] == 'e' || suffix[0] == 'E') && suffix.size() > 1) {  // decimal exponent
    size_t j = 1;
    bool eneg = false;
    if (suffix[j] == '-' || suffix[j] == '+') eneg = suffix[j++] == '-';
    if (j >= suffix.size()) return false;
    long e = 0;
    for (; j < suffix.size(); ++j) {
      if (suffix[j] < '0' || suffix[j] > '9') return false;
      if (e < 1000000) e = e * 10 + (suffix[j] - '0');
    }
    *exp10 += eneg ? -e : e;
    return true;
  }
  return false;
}

// ceil(|value| * 10^shift) with sign restored, saturating at ±MaxInt64; 0 for text that is not a quantity
inline int64_t scaled(const std::string& text, int shift) {
  bool negative;
  std::string digits;
  long exp10;
  int pow2;
  if (!split(text, &negative, &digits, &exp10, &pow2)) return 0;
  strip(&digits);
  if (digits.empty()) return 0;
  for (int i = 0; i < pow2; ++i) digits = doubled(digits);
  exp10 += shift;
  const int64_t kMax = std::numeric_limits<int64_t>::max();
  const std::string kMaxText = std::to_string(kMax);
  if (exp10 > 0) {
    if (static_cast<long>(digits.size()) + exp10 > static_cast<long>(kMaxText.size())) return negative ? -kMax : kMax;
    digits.append(static_cast<size_t>(exp10), '0');
  } else if (exp10 < 0) {
    const size_t drop = static_cast<size_t>(-exp10);
    bool lost = false;
    if (drop >= digits.size()) {
      lost = true;  // digits is non-zero
      digits.clear();
    } else {
      for (size_t k = digits.size() - drop; k < digits.size(); ++k) lost = lost || digits[k] != '0';
      digits.erase(digits.size() - drop);
    }
    if (lost) digits = plus_one(digits.empty() ? std::string("0") : digits);
    strip(&digits);
    if (digits.empty()) return 0;
  }
  if (digits.size() > kMaxText.size() || (digits.size() == kMaxText.size() && digits > kMaxText)) return negative ? -kMax : kMax;
  int64_t v = 0;
  for (char c : digits) v = v * 10 + (c - '0');
  return negative ? -v : v;
}

}  // namespace qty

inline int64_t quantity_value(const std::string& s) { return qty::scaled(s, 0); }
inline int64_t quantity_milli(const std::string& s) { return qty::scaled(s, 3); }

}  // namespace orc
