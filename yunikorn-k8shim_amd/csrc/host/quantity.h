// quantity.h — Kubernetes resource.Quantity text → int64 (host-library copy).
//
// The reference turns every request / allocatable entry into an int64 with `MilliValue()` for cpu
// and `Value()` for everything else (/root/reference/pkg/common/resource.go:273-285). Those two
// methods live in k8s.io/apimachinery (pinned v0.36.1 in go.mod, not vendored under
// /root/reference); this restates their published behaviour:
//   <quantity> ::= <signedNumber><suffix>
//   <suffix>   ::= Ki|Mi|Gi|Ti|Pi|Ei | n|u|m|""|k|M|G|T|P|E | e<signedNumber> | E<signedNumber>
// Value()/MilliValue() round AWAY from zero when the scale drops digits and saturate at ±MaxInt64.
// Known answers pinned by the reference's tests: "500M"+"1024M" = 1524e6, cpu "1"+"2" = 3000 milli,
// "0.5" = 500 milli, "5.12" = 5120 milli (resource_test.go:153-260).
#pragma once
#include <cstdint>
#include <limits>
#include <string>

namespace ykh {

struct Quantity {
  bool ok = false;
  __int128 mant = 0;  // signed mantissa
  int bin_shift = 0;  // multiply by 2^bin_shift
  int dec_exp = 0;    // multiply by 10^dec_exp
};

inline Quantity parse_quantity(const std::string& s) {
  Quantity q;
  size_t p = 0, n = s.size();
  bool neg = false;
  if (p < n && (s[p] == '+' || s[p] == '-')) {
    neg = s[p] == '-';
    ++p;
  }
  int digits = 0, frac = 0;
  bool seen_dot = false;
  __int128 m = 0;
  int dropped = 0;
  while (p < n) {
    char c = s[p];
    if (c >= '0' && c <= '9') {
      if (m < (static_cast<__int128>(1) << 100)) {
        m = m * 10 + (c - '0');
        if (seen_dot) ++frac;
      } else if (!seen_dot) {
        ++dropped;  // absurdly long integer part: keep magnitude (will saturate)
      }
      ++digits;
      ++p;
    } else if (c == '.' && !seen_dot) {
      seen_dot = true;
      ++p;
    } else {
      break;
    }
  }
  if (digits == 0) return q;
  q.mant = neg ? -m : m;
  q.dec_exp = dropped - frac;
  std::string suf = s.substr(p);
  if (suf.empty()) {
    q.ok = true;
    return q;
  }
  if ((suf[0] == 'e' || suf[0] == 'E') && suf.size() > 1) {
    // decimal exponent (a bare "E" is the exa suffix, handled below)
    size_t k = 1;
    bool eneg = false;
    if (suf[k] == '+' || suf[k] == '-') {
      eneg = suf[k] == '-';
      ++k;
    }
    if (k < suf.size() && suf[k] >= '0' && suf[k] <= '9') {
      int e = 0;
      for (; k < suf.size(); ++k) {
        if (suf[k] < '0' || suf[k] > '9') return q;
        if (e < 100000) e = e * 10 + (suf[k] - '0');
      }
      q.dec_exp += eneg ? -e : e;
      q.ok = true;
      return q;
    }
    if (suf != "Ei") return q;
  }
  struct Suf {
    const char* name;
    int bin;
    int dec;
  };
  static const Suf table[] = {{"Ki", 10, 0}, {"Mi", 20, 0}, {"Gi", 30, 0}, {"Ti", 40, 0}, {"Pi", 50, 0},
                              {"Ei", 60, 0}, {"n", 0, -9},  {"u", 0, -6},  {"m", 0, -3},  {"k", 0, 3},
                              {"M", 0, 6},   {"G", 0, 9},   {"T", 0, 12},  {"P", 0, 15},  {"E", 0, 18}};
  for (const Suf& t : table) {
    if (suf == t.name) {
      q.bin_shift = t.bin;
      q.dec_exp += t.dec;
      q.ok = true;
      return q;
    }
  }
  return q;
}

// value * 10^(-scale10) as int64, rounding away from zero, saturating.
inline int64_t scaled_value(const Quantity& q, int scale10) {
  const int64_t kMax = std::numeric_limits<int64_t>::max();
  if (!q.ok || q.mant == 0) return 0;
  bool neg = q.mant < 0;
  __int128 m = neg ? -q.mant : q.mant;
  const __int128 lim = static_cast<__int128>(1) << 120;
  for (int i = 0; i < q.bin_shift; ++i) {
    if (m >= lim / 2) return neg ? -kMax : kMax;
    m *= 2;
  }
  int e = q.dec_exp - scale10;
  bool inexact = false;
  while (e > 0) {
    if (m >= lim / 10) return neg ? -kMax : kMax;
    m *= 10;
    --e;
  }
  while (e < 0 && m != 0) {
    if (m % 10 != 0) inexact = true;
    m /= 10;
    ++e;
  }
  if (inexact) m += 1;
  if (m > static_cast<__int128>(kMax)) return neg ? -kMax : kMax;
  int64_t v = static_cast<int64_t>(m);
  return neg ? -v : v;
}

// Exact sign of (q − bound): resource.Quantity.CmpInt64 (no rounding, no saturation).
inline int quantity_cmp_int64(const Quantity& q, int64_t bound) {
  if (!q.ok) return 0;
  if (q.mant == 0) return bound > 0 ? -1 : (bound < 0 ? 1 : 0);
  const bool neg = q.mant < 0;
  if (neg != (bound < 0)) return neg ? -1 : 1;  // different signs (bound == 0 counts as non-negative)
  __int128 m = neg ? -q.mant : q.mant;
  __int128 b = bound < 0 ? -static_cast<__int128>(bound) : static_cast<__int128>(bound);
  const __int128 lim = static_cast<__int128>(1) << 120;
  int mag = 0;  // sign of (|q| − |bound|)
  bool decided = false;
  for (int i = 0; i < q.bin_shift && !decided; ++i) {
    if (m >= lim / 2) {
      mag = 1;
      decided = true;
    } else {
      m *= 2;
    }
  }
  for (int e = q.dec_exp; e > 0 && !decided; --e) {
    if (m >= lim / 10) {
      mag = 1;
      decided = true;
    } else {
      m *= 10;
    }
  }
  for (int e = q.dec_exp; e < 0 && !decided; ++e) {
    if (b >= lim / 10) {
      mag = -1;
      decided = true;
    } else {
      b *= 10;
    }
  }
  if (!decided) mag = m > b ? 1 : (m < b ? -1 : 0);
  return neg ? -mag : mag;
}

inline int64_t quantity_value(const std::string& s) { return scaled_value(parse_quantity(s), 0); }
inline int64_t quantity_milli(const std::string& s) { return scaled_value(parse_quantity(s), -3); }

}  // namespace ykh
