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

// |q| * 10^(-scale10) split EXACTLY into an integer part (saturating at 2^64) and "a non-zero fraction remains".
// The value is mant * 2^bin_shift * 10^dec_exp. With a negative decimal exponent the binary shift is applied to the pair
// (quotient, remainder) of the division by 10^k, one bit at a time: nothing is ever rounded or saturated before the final
// magnitude is known (a 25-digit fraction with an Ei suffix needs ~160 bits as a plain product).
struct Magnitude {
  unsigned __int128 whole = 0;
  bool fraction = false, huge = false;  // huge: whole part >= 2^64 (saturates every int64 use)
};
inline Magnitude quantity_magnitude(const Quantity& q, int scale10) {
  Magnitude out;
  unsigned __int128 m = static_cast<unsigned __int128>(q.mant < 0 ? -q.mant : q.mant);
  const unsigned __int128 cap = static_cast<unsigned __int128>(1) << 64;
  const int e = q.dec_exp - scale10;
  if (m == 0) return out;
  if (e >= 0) {
    for (int i = 0; i < e && !out.huge; ++i) {
      m *= 10;  // m < 2^101 here: no wrap before the cap test
      if (m >= cap) out.huge = true;
    }
    for (int i = 0; i < q.bin_shift && !out.huge; ++i) {
      m *= 2;
      if (m >= cap) out.huge = true;
    }
    out.whole = m;
    return out;
  }
  const int k = -e;
  if (k > 38) {  // only reachable with a decimal suffix (no binary shift): 0 < value < 1
    out.fraction = true;
    return out;
  }
  unsigned __int128 p = 1;
  for (int i = 0; i < k; ++i) p *= 10;  // <= 10^38 < 2^127
  unsigned __int128 whole = m / p, rem = m % p;
  for (int i = 0; i < q.bin_shift && !out.huge; ++i) {
    rem *= 2;  // < 2^128
    const bool carry = rem >= p;
    if (carry) rem -= p;
    whole = whole * 2 + (carry ? 1 : 0);
    if (whole >= cap) out.huge = true;
  }
  out.whole = whole;
  out.fraction = rem != 0;
  return out;
}

// value * 10^(-scale10) as int64, rounding away from zero, saturating.
inline int64_t scaled_value(const Quantity& q, int scale10) {
  const int64_t kMax = std::numeric_limits<int64_t>::max();
  if (!q.ok || q.mant == 0) return 0;
  const bool neg = q.mant < 0;
  const Magnitude g = quantity_magnitude(q, scale10);
  if (g.huge) return neg ? -kMax : kMax;
  const unsigned __int128 v = g.whole + (g.fraction ? 1 : 0);
  if (v > static_cast<unsigned __int128>(kMax)) return neg ? -kMax : kMax;
  return neg ? -static_cast<int64_t>(v) : static_cast<int64_t>(v);
}

// Exact sign of (q − bound): resource.Quantity.CmpInt64 (no rounding, no saturation).
inline int quantity_cmp_int64(const Quantity& q, int64_t bound) {
  if (!q.ok) return 0;
  if (q.mant == 0) return bound > 0 ? -1 : (bound < 0 ? 1 : 0);
  const bool neg = q.mant < 0;
  if (neg != (bound < 0)) return neg ? -1 : 1;  // different signs (bound == 0 counts as non-negative)
  const unsigned __int128 b = bound < 0 ? static_cast<unsigned __int128>(-static_cast<__int128>(bound)) : static_cast<unsigned __int128>(bound);
  const Magnitude g = quantity_magnitude(q, 0);
  int mag;  // sign of (|q| − |bound|)
  if (g.huge || g.whole > b)
    mag = 1;
  else if (g.whole < b)
    mag = -1;
  else
    mag = g.fraction ? 1 : 0;
  return neg ? -mag : mag;
}

inline int64_t quantity_value(const std::string& s) { return scaled_value(parse_quantity(s), 0); }
inline int64_t quantity_milli(const std::string& s) { return scaled_value(parse_quantity(s), -3); }

}  // namespace ykh
