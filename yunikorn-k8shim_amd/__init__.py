"""yunikorn-k8shim_amd — MI355X-native batched predicate engine for the yunikorn-k8shim hot path.

Only what the path needs lives here: csrc/engine (HIP kernels + the C ABI of include/ykpred.h), csrc/host (the
C++ stand-in for the Go shim side, include/ykhost.h) and thin ctypes bindings. See DESIGN.md.
The package name contains a hyphen; import it with importlib.import_module("yunikorn-k8shim_amd").
"""
from . import build  # noqa: F401
from .predicate_manager import (ALL_PLUGINS, PLUGIN_BITS, GpuPredicateManager, PredicateError,  # noqa: F401
                                UnsupportedAsk, plugin_mask)

build_all = build.build_all
