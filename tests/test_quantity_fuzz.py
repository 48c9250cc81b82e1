"""resource.Quantity → int64 (Value() / MilliValue(), pkg/common/resource.go:273-285) — the oracle's digit-string converter
(oracle/orc_quantity.h) and the product's 128-bit one (yunikorn-k8shim_amd/csrc/host/quantity.h) were written independently;
both are driven here against exact rational arithmetic (fractions.Fraction): round away from zero, saturate at ±MaxInt64."""
import importlib
import json
import random
from fractions import Fraction

import pytest

import _oracle as orc

pkg = importlib.import_module("yunikorn-k8shim_amd")
MAX = 2**63 - 1
SUFFIX = {"": (0, 0), "Ki": (10, 0), "Mi": (20, 0), "Gi": (30, 0), "Ti": (40, 0), "Pi": (50, 0), "Ei": (60, 0), "n": (0, -9), "u": (0, -6),
          "m": (0, -3), "k": (0, 3), "M": (0, 6), "G": (0, 9), "T": (0, 12), "P": (0, 15), "E": (0, 18)}


def exact(text, milli):
    """Expected int64 of a VALID quantity text, by exact arithmetic."""
    sign = -1 if text.startswith("-") else 1
    body = text.lstrip("+-")
    i = 0
    while i < len(body) and (body[i].isdigit() or body[i] == "."):
        i += 1
    number, suffix = body[:i], body[i:]
    value = Fraction(number if not number.endswith(".") else number + "0") if number not in (".",) else Fraction(0)
    if suffix in SUFFIX:
        p2, p10 = SUFFIX[suffix]
        value *= Fraction(2) ** p2 * Fraction(10) ** p10
    else:  # e<int> / E<int>
        value *= Fraction(10) ** int(suffix[1:])
    if milli:
        value *= 1000
    whole = -(-value.numerator // value.denominator)  # ceil of the magnitude = away from zero
    return sign * min(whole, MAX)


def random_quantity(rng):
    digits = "".join(rng.choice("0123456789") for _ in range(rng.choice([1, 1, 2, 3, 5, 9, 18, 19, 20, 25])))
    if rng.random() < 0.5:
        digits += "." + "".join(rng.choice("0123456789") for _ in range(rng.choice([0, 1, 2, 3, 4, 9, 13])))
    elif rng.random() < 0.1:
        digits = "." + digits
    r = rng.random()
    if r < 0.7:
        suffix = rng.choice(list(SUFFIX))
    else:
        suffix = rng.choice("eE") + rng.choice(["", "+", "-"]) + str(rng.choice([0, 1, 2, 3, 6, 9, 12, 17, 18, 19, 25]))
    return rng.choice(["", "", "", "+", "-"]) + digits + suffix


def product_values(texts):
    """(Value(), MilliValue()) of every text through the product's host library: a pod whose only container requests
    memory = text (→ Value()) and cpu = text (→ MilliValue()); mirror-only handle, no device needed."""
    m = pkg.GpuPredicateManager(device=-1)
    try:
        pods = [{"metadata": {"name": f"p{i}", "uid": f"p{i}"},
                 "spec": {"containers": [{"name": "c", "resources": {"requests": {"memory": t, "cpu": t}}}]}} for i, t in enumerate(texts)]
        m.load_snapshot({"nodes": [], "pods": pods})
        out = []
        for i in range(len(texts)):
            r = m.pod_request(i)
            out.append((r.get("memory", 0), r.get("cpu", 0)))
        return out
    finally:
        m.close()


def test_both_converters_against_exact_arithmetic():
    rng = random.Random(20260921)
    texts = [random_quantity(rng) for _ in range(6000)]
    texts += ["0", "0.0", "1", "1.5", "500M", "1024M", "0.5", "5.12", "100m", "1e3", "1E3", "1E", "1Ei", "9223372036854775807",
              "9223372036854775808", "-9223372036854775808", "8Ei", "7.9999999999Ei", "0.0000000001n", "-0.1m", "1e-30", "12345678901234567890123",
              "1.0000000000000000000000001", "999999999999999999999m", "0.9999999999999999999", "16000000000", "256Gi", "32", "110"]
    L = orc.lib()
    got_product = product_values(texts)
    bad = []
    for t, (pv, pm) in zip(texts, got_product):
        want_v, want_m = exact(t, False), exact(t, True)
        ov, om = L.orc_quantity_value(t.encode()), L.orc_quantity_milli(t.encode())
        if (ov, om) != (want_v, want_m) or (pv, pm) != (want_v, want_m):
            bad.append((t, (want_v, want_m), (ov, om), (pv, pm)))
    assert not bad, f"{len(bad)} of {len(texts)} differ, e.g. {bad[:5]}  (text, exact, oracle, product)"


@pytest.mark.parametrize("text", ["", "abc", "1x", "1KiB", "--1", "1e", "1e+", "e3", ".", "1..2", "1 Gi", "0x10"])
def test_malformed_quantities_convert_to_zero_on_both_sides(text):
    """apimachinery's parser rejects these; the snapshot readers map an unparsable quantity to 0 (a request that is not
    there) — identically in the oracle and in the product."""
    L = orc.lib()
    assert (L.orc_quantity_value(text.encode()), L.orc_quantity_milli(text.encode())) == (0, 0)
    assert product_values([text]) == [(0, 0)]
