"""CPU-side checks of the drop-in boundary: the C-ABI libraries build, load, and export every entry point that
include/*.h declares; and the product fails loudly (no CPU fallback) when no HIP device is present."""
import ctypes
import importlib
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pkg = importlib.import_module("yunikorn-k8shim_amd")


def declared_functions(header):
    text = open(os.path.join(ROOT, "include", header)).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(yk(?:pred|host)_[a-z0-9_]+)\s*\(", text)))


def test_libraries_build_and_export_every_declared_symbol():
    pred_path, host_path = pkg.build_all()
    pred = ctypes.CDLL(pred_path, mode=ctypes.RTLD_GLOBAL)
    host = ctypes.CDLL(host_path)
    names = declared_functions("ykpred.h")
    assert len(names) >= 18
    for fn in names:
        assert hasattr(pred, fn), f"libykpred.so does not export {fn}"
    hnames = [n for n in declared_functions("ykhost.h") if n.startswith("ykhost_")]
    assert len(hnames) >= 20
    for fn in hnames:
        assert hasattr(host, fn), f"libykhost.so does not export {fn}"
    pred.ykpred_abi_version.restype = ctypes.c_int32
    assert pred.ykpred_abi_version() == 4


def test_no_cpu_fallback_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        pkg.GpuPredicateManager()


def test_product_does_not_reference_the_oracle():
    """The oracle is test infrastructure: nothing under the package may include, link or load it."""
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "yunikorn-k8shim_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f), errors="ignore").read()
                if re.search(r"ykoracle|oracle/|import _oracle|orc_", text):
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_go_side_only_uses_the_declared_abi():
    """integration/gpu_predicate_manager.go (the Go drop-in; no Go toolchain here) may only call what include/*.h declares,
    with the declared argument counts, and contains no elisions."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "scripts", "check_go_bindings.py")], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "checked" in r.stdout and "PROBLEM" not in r.stdout


def test_python_binding_matches_the_headers():
    """yunikorn-k8shim_amd/_ffi.py declares ctypes argtypes by hand: every function it types must exist in include/*.h with
    that many parameters (a drifted binding would pass garbage through the C ABI without any compile error)."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "scripts"))
    import check_go_bindings as cg
    headers = ""
    for h in ("ykpred.h", "ykhost.h"):
        headers += cg.strip_comments(open(os.path.join(ROOT, "include", h)).read()) + "\n"
    declared = {}
    for m in re.finditer(r"\b(yk(?:pred|host)_\w+)\s*\(", headers):
        args = cg.call_args(headers, m.end() - 1)
        declared[m.group(1)] = 0 if args.strip() in ("", "void") else len(cg.split_args(args))
    ffi = importlib.import_module("yunikorn-k8shim_amd._ffi")
    checked = 0
    for lib in (ffi.load_ykpred(), ffi.load_ykhost()):
        for name, want in declared.items():
            fn = getattr(lib, name, None)
            if fn is None or fn.argtypes is None:
                continue
            assert len(fn.argtypes) == want, f"{name}: _ffi.py declares {len(fn.argtypes)} parameters, the header {want}"
            checked += 1
    assert checked >= 40, checked


def _build_replay(tmp_path, sanitize):
    """tests/c/replay_cgo_sequence.c + the host library's SOURCES in one binary (so that the sanitizer instruments the
    library code that could retain a caller's pointer); the engine library is linked as built."""
    import subprocess
    lib = os.path.join(ROOT, "yunikorn-k8shim_amd", "lib")
    importlib.import_module("yunikorn-k8shim_amd").build_all()
    san = ["-fsanitize=address,undefined", "-fno-omit-frame-pointer", "-fno-sanitize-recover=undefined"] if sanitize else []
    obj, exe = str(tmp_path / "replay.o"), str(tmp_path / "replay")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-g", "-Wall", "-Wextra"] + san + ["-I" + os.path.join(ROOT, "include"), "-c",
                           os.path.join(ROOT, "tests", "c", "replay_cgo_sequence.c"), "-o", obj])
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g"] + san + ["-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "yunikorn-k8shim_amd", "csrc", "host", "host.cpp"), obj, "-o", exe, "-L" + lib, "-lykpred",
                           "-Wl,-rpath," + lib])
    return exe


def test_cgo_call_sequence_replays_under_the_sanitizers(tmp_path):
    """The Go manager's call sequence and string ownership (every C.CString freed right after its call) replayed from C on a
    mirror-only handle, libykhost's sources built with AddressSanitizer + UBSan: the library retains no caller pointer."""
    import subprocess
    exe = _build_replay(tmp_path, sanitize=True)
    out = subprocess.run([exe, "-1"], capture_output=True, text=True, timeout=300, env=dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0"))
    assert out.returncode == 0 and "replay ok" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


@pytest.mark.gpu
def test_cgo_call_sequence_replays_on_the_device(tmp_path):
    """The same sequence against the engine: the verdicts the Go file would hand back to the core."""
    import subprocess
    exe = _build_replay(tmp_path, sanitize=False)
    out = subprocess.run([exe, "0"], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "replay ok (device 0)" in out.stdout, (out.stdout[-500:], out.stderr[-3000:])


@pytest.mark.skipif(not os.path.isdir("/root/reference"), reason="the reference tree is only present in the build container")
def test_go_patches_apply_to_the_reference():
    """integration/patches/*.diff (scripts/make_go_patches.py): the install line of context.go:130, the observer hooks of
    scheduler_cache.go:148-484 and the service.predicateEngine key of schedulerconf.go:59-81 apply cleanly to the reference,
    and are what the generator produces from it today."""
    import shutil
    import subprocess
    if not shutil.which("git"):
        pytest.skip("git not installed")
    pdir = os.path.join(ROOT, "integration", "patches")
    names = sorted(n for n in os.listdir(pdir) if n.endswith(".diff"))
    assert names == ["context.go.diff", "scheduler_cache.go.diff", "schedulerconf.go.diff"]
    for n in names:
        r = subprocess.run(["git", "apply", "--check", "--verbose", os.path.join(pdir, n)], cwd="/root/reference", capture_output=True, text=True)
        assert r.returncode == 0, (n, r.stderr)
    # the Go file and the patches agree on the names they share
    go = open(os.path.join(ROOT, "integration", "gpu_predicate_manager.go")).read()
    cache_patch = open(os.path.join(pdir, "scheduler_cache.go.diff")).read()
    for hook in ("OnUpdateNode", "OnRemoveNode", "OnUpdatePod", "OnRemovePod", "OnAssumePod", "OnForgetPod"):
        assert f"func (m *gpuPredicateManager) {hook}(" in go and f"cache.observer.{hook}(" in cache_patch
    assert "func NewConfiguredPredicateManager(handle fwk.Handle, engine string, device int) PredicateManager" in go
    assert "predicates.NewConfiguredPredicateManager(" in open(os.path.join(pdir, "context.go.diff")).read()
