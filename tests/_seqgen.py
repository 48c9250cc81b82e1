"""Clusters for the sequential-allocation tests: unlike tests/_gen.py (edge cases, mostly infeasible asks) these are made so that
MANY asks fit and compete for the same nodes — the interesting case for conflict-resolved decisions. Pure Python, tests only."""
import random


def perf_shape(n_nodes, n_pods, apps=5):
    """The reference's only perf artefact, /root/reference/pkg/shim/scheduler_perf_test.go:62-66,283-352: empty nodes of 16 000 m /
    16 G / 110 pods, asks of 10 m / 1 M ("UID-app000<i>-...-task000<j>" names shortened)."""
    nodes = [{"metadata": {"name": f"test.host.{i:04d}", "labels": {}}, "spec": {},
              "status": {"allocatable": {"cpu": "16000m", "memory": "16G", "pods": "110"}}, "pods": []} for i in range(n_nodes)]
    pods = []
    for k in range(n_pods):
        app = k % apps
        pods.append({"metadata": {"name": f"app{app:04d}-task{k:06d}", "uid": f"UID-app{app:04d}-task{k:06d}", "namespace": "default"},
                     "spec": {"containers": [{"name": "container-01", "resources": {"requests": {"cpu": "10m", "memory": "1M"}}}]}})
    return {"nodes": nodes, "pods": pods}


def competing(seed, n_nodes=40, n_pods=120, taints=True, selectors=True, pins=True, scalars=False, spread=False, ports=False, ipa=False):
    """Small nodes, asks from a handful of templates that mostly fit somewhere: resources and pod slots run out during the round."""
    rng = random.Random(seed)
    zones = ["a", "b", "c"]
    nodes = []
    for i in range(n_nodes):
        alloc = {"cpu": rng.choice(["2", "4", "8"]), "memory": rng.choice(["4Gi", "8Gi", "16Gi"]), "pods": rng.choice(["3", "5", "8", "110"])}
        if scalars and rng.random() < 0.5:
            alloc["example.com/gpu"] = rng.choice(["1", "2", "4"])
        node = {"metadata": {"name": f"n{rng.randrange(10**6):06d}-{i}", "labels": {"zone": rng.choice(zones), "kubernetes.io/hostname": f"h{i}"}},
                "spec": {"taints": [], "unschedulable": rng.random() < 0.05}, "status": {"allocatable": alloc}, "pods": []}
        if taints and rng.random() < 0.3:
            node["spec"]["taints"].append({"key": "dedicated", "value": rng.choice(["x", "y"]), "effect": "NoSchedule"})
        for j in range(rng.choice([0, 0, 1, 2])):
            node["pods"].append({"metadata": {"name": f"r{i}-{j}", "uid": f"r{i}-{j}", "namespace": "default", "labels": {"app": rng.choice(["w", "v"])}},
                                 "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": rng.choice(["100m", "500m", "1"]), "memory": rng.choice(["256Mi", "1Gi"])}}}]}})
        nodes.append(node)
    templates = []
    for t in range(rng.randint(3, 7)):
        req = {"cpu": rng.choice(["100m", "250m", "500m", "1", "0"]), "memory": rng.choice(["128Mi", "512Mi", "1Gi", "2Gi"])}
        if scalars and rng.random() < 0.3:
            req["example.com/gpu"] = "1"
        spec = {"containers": [{"name": "c", "resources": {"requests": {k: v for k, v in req.items() if v != "0"}}}]}
        if taints and rng.random() < 0.5:
            spec["tolerations"] = [{"key": "dedicated", "operator": "Equal", "value": rng.choice(["x", "y"]), "effect": "NoSchedule"}]
        if selectors and rng.random() < 0.4:
            spec["nodeSelector"] = {"zone": rng.choice(zones)}
        if spread and rng.random() < 0.6:
            spec["topologySpreadConstraints"] = [{"maxSkew": rng.choice([1, 2]), "topologyKey": "zone", "whenUnsatisfiable": "DoNotSchedule",
                                                  "labelSelector": {"matchLabels": {"app": f"t{t}"}}}]
        if ports and rng.random() < 0.5:
            spec["containers"][0]["ports"] = [{"hostPort": rng.choice([80, 443, 8080]), "containerPort": 80}]
        if ipa and rng.random() < 0.7:
            # required inter-pod (anti)affinity between the asks themselves and towards the pods already running ("w" / "v"): an ask
            # assumed earlier in the round is an EXISTING pod for every ask behind it, in both directions of the anti-affinity rule
            who = lambda: rng.choice([f"t{t}", f"t{rng.randrange(0, t + 1)}", "w", "v"])
            term = lambda: {"labelSelector": {"matchLabels": {"app": who()}}, "topologyKey": rng.choice(["kubernetes.io/hostname", "zone"])}
            aff = {}
            if rng.random() < 0.7:
                aff["podAntiAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": [term() for _ in range(rng.choice([1, 1, 2]))]}
            if rng.random() < 0.4:
                aff["podAffinity"] = {"requiredDuringSchedulingIgnoredDuringExecution": [term()]}
            if aff:
                spec["affinity"] = aff
        templates.append((f"t{t}", spec))
    pods = []
    for k in range(n_pods):
        name, spec = rng.choice(templates)
        spec = {kk: (list(vv) if isinstance(vv, list) else vv) for kk, vv in spec.items()}
        if pins and rng.random() < 0.03:
            spec["nodeName"] = rng.choice(nodes)["metadata"]["name"] if rng.random() < 0.8 else "no-such-node"
        pods.append({"metadata": {"name": f"ask-{k}", "uid": f"ask-{k}", "namespace": "default", "labels": {"app": name}}, "spec": spec})
    return {"nodes": nodes, "pods": pods}


def small_slots(seed, n_nodes=3000, n_pods=12000, n_templates=24, spread=False, ports=False):
    """Thousands of MOVED nodes in one round: nodes with 3 to 6 pod slots and a few cores, asks of a couple of dozen templates in
    random order — a node takes a handful of asks and is full, so a 12 000-ask round moves well over 2 000 nodes and the scan over
    the moved-node slots (512 slots per step in k_allocate_round) runs over many steps. `spread`: a third of the templates carry a
    hard zone constraint on their own label; `ports`: a few templates want a host port (one such pod per node)."""
    rng = random.Random(seed)
    zones = [f"z{i}" for i in range(8)]
    nodes = []
    for i in range(n_nodes):
        alloc = {"cpu": rng.choice(["2", "3", "4", "6"]), "memory": rng.choice(["4Gi", "6Gi", "8Gi"]), "pods": rng.choice(["3", "4", "5", "6"])}
        node = {"metadata": {"name": f"s{rng.randrange(10**7):07d}-{i}", "labels": {"zone": rng.choice(zones), "kubernetes.io/hostname": f"h{i}",
                                                                                     "pool": rng.choice(["a", "b", "c"])}},
                "spec": {"taints": [], "unschedulable": rng.random() < 0.01}, "status": {"allocatable": alloc}, "pods": []}
        if rng.random() < 0.15:
            node["spec"]["taints"].append({"key": "dedicated", "value": rng.choice(["x", "y"]), "effect": "NoSchedule"})
        if rng.random() < 0.3:
            node["pods"].append({"metadata": {"name": f"r{i}", "uid": f"r{i}", "namespace": "default", "labels": {"app": rng.choice(["w", "v"])}},
                                 "spec": {"containers": [{"name": "c", "resources": {"requests": {"cpu": rng.choice(["100m", "500m", "1"]), "memory": rng.choice(["256Mi", "1Gi"])}}}]}})
        nodes.append(node)
    templates = []
    for t in range(n_templates):
        req = {"cpu": rng.choice(["100m", "250m", "500m", "750m", "1"]), "memory": rng.choice(["128Mi", "512Mi", "1Gi", "1536Mi"])}
        spec = {"containers": [{"name": "c", "resources": {"requests": req}}]}
        if rng.random() < 0.4:
            spec["tolerations"] = [{"key": "dedicated", "operator": "Equal", "value": rng.choice(["x", "y"]), "effect": "NoSchedule"}]
        if rng.random() < 0.3:
            spec["nodeSelector"] = {"pool": rng.choice(["a", "b", "c"])}
        if spread and t % 3 == 0:
            spec["topologySpreadConstraints"] = [{"maxSkew": rng.choice([1, 2, 5]), "topologyKey": "zone", "whenUnsatisfiable": "DoNotSchedule",
                                                  "labelSelector": {"matchLabels": {"app": f"t{t}"}}}]
        if ports and t % 6 == 1:
            spec["containers"][0]["ports"] = [{"hostPort": rng.choice([80, 443, 8080]), "containerPort": 80}]
        templates.append((f"t{t}", spec))
    pods = []
    for k in range(n_pods):
        name, spec = rng.choice(templates)
        pods.append({"metadata": {"name": f"ask-{k}", "uid": f"ask-{k}", "namespace": "default", "labels": {"app": name}}, "spec": dict(spec)})
    return {"nodes": nodes, "pods": pods}
