"""The second CPU baseline of bench.py (oracle/soa_cpu.c: encoded tables, per pair, OpenMP) must itself be right: same
bits as the object-model oracle on random clusters, both phases."""
import importlib

import numpy as np
import pytest

import _gen
import _oracle as orc
import _soa_cpu

pkg = importlib.import_module("yunikorn-k8shim_amd")


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("allocate", [True, False])
def test_table_driven_cpu_evaluator_matches_the_oracle(seed, allocate):
    snap = _gen.random_snapshot(8800 + seed, n_nodes=50 + 31 * seed, n_pods=70)
    m = pkg.GpuPredicateManager(device=-1)
    try:
        m.load_snapshot(snap)
        t = m.encoded_tables()
    finally:
        m.close()
    pre, filt = (orc.ALL, orc.ALL) if allocate else (orc.RESERVE_PRE, orc.RESERVE_FILT)
    want = orc.Oracle(snap).eval_grid(pre_mask=pre, filt_mask=filt, threads=4)
    for threads in (1, 4):
        got = _soa_cpu.evaluate(t, pre, filt, threads=threads)
        assert np.array_equal(got, orc.pack_bits(want))
