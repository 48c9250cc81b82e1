import importlib, time, os, sys, json
sys.path.insert(0, os.getcwd())
pkg = importlib.import_module('yunikorn-k8shim_amd')
out = {"affinity": len(os.sched_getaffinity(0)), "cpu_count": os.cpu_count()}
try:
    out["cpu_max"] = open('/sys/fs/cgroup/cpu.max').read().strip()
except Exception as e:
    out["cpu_max"] = str(e)
src = pkg.GpuPredicateManager(device=-1)
src.generate_kwok(seed=0x59554E49 + 2, num_nodes=50000, num_pods=1000000, num_templates=2000, node_affinity=1, spread=0)
docs = [src.dump_documents(k) for k in (0, 1, 2)]
src.close()
for thr in ("64", "16", "1"):
    os.environ["YKHOST_INGEST_THREADS"] = thr
    break
m = pkg.GpuPredicateManager(device=-1)
t0 = time.perf_counter(); m.update_documents(0, docs[0]); t1 = time.perf_counter(); m.update_documents(1, docs[1]); m.update_documents(2, docs[2]); t2 = time.perf_counter()
out["nodes_ms"] = round((t1 - t0) * 1e3, 1); out["pods_ms"] = round((t2 - t1) * 1e3, 1); out["timing"] = m.ingest_timing()
m.close()
print(json.dumps(out))
