"""Debug helper: find (pod, word) cells where k_direct and the plane/class path disagree at full size."""
import importlib, sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import _oracle as orc
pkg = importlib.import_module("yunikorn-k8shim_amd")
pm = pkg.GpuPredicateManager()
N, P = int(sys.argv[1]), int(sys.argv[2])
pm.generate_kwok(seed=0x59554E49 + 11, num_nodes=N, num_pods=P, num_templates=2000, node_affinity=1)
pm.evaluate()
step = 20000
bad = []
planes = []
for f in range(0, P, step):
    planes.append(pm.read_bitmap(f, min(step, P - f)))
pm.evaluate(direct=True, counts=False, decisions=False)
for i, f in enumerate(range(0, P, step)):
    d = pm.read_bitmap(f, min(step, P - f))
    diff = np.argwhere(d != planes[i])
    for (r, w) in diff[:5]:
        bad.append((f + int(r), int(w), int(planes[i][r, w]), int(d[r, w])))
    if len(diff):
        print("slab", f, "differing words", len(diff))
print("total examples", len(bad))
for (p, w, a, b) in bad[:10]:
    x = a ^ b
    bits = [j for j in range(64) if (x >> j) & 1]
    nodes = [w * 64 + j for j in bits]
    print(f"pod {p} word {w}: planes={a:016x} direct={b:016x} differing nodes {nodes[:8]}")
    fit, code, reason = pm.query([p] * len(nodes[:8]), nodes[:8])
    o = orc.Oracle(pm.dump_snapshot(pods=[p], nodes=nodes[:8]))
    print("   k_query fit", fit.tolist(), "code", code.tolist(), " oracle", o.eval_grid().tolist())
    print("   pod json", pm.dump_snapshot(pods=[p], nodes=[])[:1500])
