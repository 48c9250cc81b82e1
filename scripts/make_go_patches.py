#!/usr/bin/env python3
"""Generates integration/patches/*.diff — the three edits of the reference that install the GPU predicate manager — as
`git apply`-able unified diffs against /root/reference (read-only here: the files are copied to a scratch directory, edited
there and diffed). tests/test_abi_symbols.py::test_go_patches_apply_to_the_reference runs `git apply --check` on the result.

  context.go          construct the manager through the configuration switch, hand it to the cache as observer   (:130)
  scheduler_cache.go  an Observer interface (defined in package external: pkg/plugin/support imports this package, so the
                      interface cannot live next to the manager) + one call at the end of each of the six critical sections
  schedulerconf.go    service.predicateEngine / service.predicateDevice next to the other service.* keys           (:59-81)
"""
import os
import shutil
import subprocess
import sys
import tempfile

REF = "/root/reference"
HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "integration", "patches")


def edit(text, old, new, count=1):
    assert text.count(old) == count, (text.count(old), old[:70])
    return text.replace(old, new)


def context_go(t):
    t = edit(t, """	ctx.predManager = predicates.NewPredicateManager(support.NewFrameworkHandle(sharedLister, informerFactory, clientSet, csiManager, sharedDRAManager))
""", """	// service.predicateEngine selects the implementation: "cpu" (default) is predicates.NewPredicateManager unchanged, "gpu" the
	// MI355X engine behind the same interface (falls back to the CPU manager when no device can be opened)
	ctx.predManager = predicates.NewConfiguredPredicateManager(
		support.NewFrameworkHandle(sharedLister, informerFactory, clientSet, csiManager, sharedDRAManager),
		schedulerconf.GetSchedulerConf().PredicateEngine, schedulerconf.GetSchedulerConf().PredicateDevice)
	if observer, ok := ctx.predManager.(schedulercache.Observer); ok {
		ctx.schedulerCache.SetObserver(observer) // the engine mirrors the cache from inside the cache's own critical sections
	}
""")
    return t


def scheduler_cache_go(t):
    t = edit(t, """	lock         locking.RWMutex
	clients      *client.Clients // client APIs
""", """	lock         locking.RWMutex
	observer     Observer        // optional: told about every change from inside the critical section that made it
	clients      *client.Clients // client APIs
""")
    t = edit(t, """func NewSchedulerCache(clients *client.Clients) *SchedulerCache {""", """// Observer is told about every mutation of the cache while the cache write lock is held, so an observer sees the
// changes in exactly the order the cache applied them. The GPU predicate manager (pkg/plugin/predicates) implements it to
// keep its encoded mirror of nodes and pods; it is defined here because pkg/plugin/support imports this package.
type Observer interface {
	OnUpdateNode(node *v1.Node)
	OnRemoveNode(node *v1.Node)
	OnUpdatePod(pod *v1.Pod)
	OnRemovePod(pod *v1.Pod)
	OnAssumePod(pod *v1.Pod)
	OnForgetPod(pod *v1.Pod)
}

// SetObserver installs the observer; call it before the informers start delivering events.
func (cache *SchedulerCache) SetObserver(observer Observer) {
	cache.lock.Lock()
	defer cache.lock.Unlock()
	cache.observer = observer
}

func NewSchedulerCache(clients *client.Clients) *SchedulerCache {""")
    t = edit(t, """	defer cache.dumpState("UpdateNode.Post")
	return cache.updateNode(node)
""", """	defer cache.dumpState("UpdateNode.Post")
	prev, adopted := cache.updateNode(node)
	if cache.observer != nil {
		cache.observer.OnUpdateNode(node)
	}
	return prev, adopted
""")
    t = edit(t, """	defer cache.dumpState("RemoveNode.Post")

	return cache.removeNode(node)
""", """	defer cache.dumpState("RemoveNode.Post")

	prev, orphans := cache.removeNode(node)
	if cache.observer != nil {
		cache.observer.OnRemoveNode(node)
	}
	return prev, orphans
""")
    t = edit(t, """	defer cache.dumpState("UpdatePod.Post")
	return cache.updatePod(newPod)
""", """	defer cache.dumpState("UpdatePod.Post")
	ok := cache.updatePod(newPod)
	if cache.observer != nil {
		cache.observer.OnUpdatePod(newPod)
	}
	return ok
""")
    t = edit(t, """	defer cache.dumpState("RemovePod.Post")
	cache.removePod(pod)
""", """	defer cache.dumpState("RemovePod.Post")
	cache.removePod(pod)
	if cache.observer != nil {
		cache.observer.OnRemovePod(pod)
	}
""")
    t = edit(t, """	defer cache.dumpState("AssumePod.Post")
	cache.assumePod(pod, allBound)
""", """	defer cache.dumpState("AssumePod.Post")
	cache.assumePod(pod, allBound)
	if cache.observer != nil {
		cache.observer.OnAssumePod(pod)
	}
""")
    t = edit(t, """	defer cache.dumpState("ForgetPod.Post")

	cache.forgetPod(pod)
""", """	defer cache.dumpState("ForgetPod.Post")

	cache.forgetPod(pod)
	if cache.observer != nil {
		cache.observer.OnForgetPod(pod)
	}
""")
    return t


def schedulerconf_go(t):
    t = edit(t, """	CMSvcNodeInstanceTypeNodeLabelKey = PrefixService + "nodeInstanceTypeNodeLabelKey"
""", """	CMSvcNodeInstanceTypeNodeLabelKey = PrefixService + "nodeInstanceTypeNodeLabelKey"
	CMSvcPredicateEngine              = PrefixService + "predicateEngine" // "cpu" (default) | "gpu"
	CMSvcPredicateDevice              = PrefixService + "predicateDevice" // HIP device ordinal of the GPU engine
""")
    t = edit(t, """	DefaultDisableGangScheduling           = false
""", """	DefaultDisableGangScheduling           = false
	DefaultPredicateEngine                 = "cpu"
	DefaultPredicateDevice                 = 0
""")
    t = edit(t, """	InstanceTypeNodeLabelKey string             `json:"instanceTypeNodeLabelKey"`
""", """	InstanceTypeNodeLabelKey string             `json:"instanceTypeNodeLabelKey"`
	PredicateEngine          string             `json:"predicateEngine"`
	PredicateDevice          int                `json:"predicateDevice"`
""")
    t = edit(t, """		InstanceTypeNodeLabelKey: conf.InstanceTypeNodeLabelKey,
""", """		InstanceTypeNodeLabelKey: conf.InstanceTypeNodeLabelKey,
		PredicateEngine:          conf.PredicateEngine,
		PredicateDevice:          conf.PredicateDevice,
""")
    t = edit(t, """	checkNonReloadableString(CMSvcNodeInstanceTypeNodeLabelKey, &old.InstanceTypeNodeLabelKey, &new.InstanceTypeNodeLabelKey)
""", """	checkNonReloadableString(CMSvcNodeInstanceTypeNodeLabelKey, &old.InstanceTypeNodeLabelKey, &new.InstanceTypeNodeLabelKey)
	checkNonReloadableString(CMSvcPredicateEngine, &old.PredicateEngine, &new.PredicateEngine)
	checkNonReloadableInt(CMSvcPredicateDevice, &old.PredicateDevice, &new.PredicateDevice)
""")
    t = edit(t, """		InstanceTypeNodeLabelKey: constants.DefaultNodeInstanceTypeNodeLabelKey,
""", """		InstanceTypeNodeLabelKey: constants.DefaultNodeInstanceTypeNodeLabelKey,
		PredicateEngine:          DefaultPredicateEngine,
		PredicateDevice:          DefaultPredicateDevice,
""")
    t = edit(t, """	parser.stringVar(&conf.InstanceTypeNodeLabelKey, CMSvcNodeInstanceTypeNodeLabelKey)
""", """	parser.stringVar(&conf.InstanceTypeNodeLabelKey, CMSvcNodeInstanceTypeNodeLabelKey)
	parser.stringVar(&conf.PredicateEngine, CMSvcPredicateEngine)
	parser.intVar(&conf.PredicateDevice, CMSvcPredicateDevice)
""")
    return t


FILES = {"pkg/cache/context.go": context_go, "pkg/cache/external/scheduler_cache.go": scheduler_cache_go, "pkg/conf/schedulerconf.go": schedulerconf_go}


def main():
    if not os.path.isdir(REF):
        sys.exit("no /root/reference here: the committed patches stand")
    os.makedirs(OUT, exist_ok=True)
    with tempfile.TemporaryDirectory() as tmp:
        for rel, fn in FILES.items():
            for side in ("a", "b"):
                os.makedirs(os.path.join(tmp, side, os.path.dirname(rel)), exist_ok=True)
                shutil.copy(os.path.join(REF, rel), os.path.join(tmp, side, rel))
            path = os.path.join(tmp, "b", rel)
            os.chmod(path, 0o644)
            with open(path) as f:
                text = f.read()
            with open(path, "w") as f:
                f.write(fn(text))
            r = subprocess.run(["diff", "-u", os.path.join("a", rel), os.path.join("b", rel)], cwd=tmp, capture_output=True, text=True)
            assert r.returncode == 1, r.stderr
            lines = r.stdout.splitlines(keepends=True)
            lines[0] = f"--- a/{rel}\n"  # (drop diff's timestamps)
            lines[1] = f"+++ b/{rel}\n"
            name = os.path.basename(rel) + ".diff"
            with open(os.path.join(OUT, name), "w") as f:
                f.write("".join(lines))
            print(name, sum(1 for ln in lines if ln.startswith("+") and not ln.startswith("+++")), "lines added")


if __name__ == "__main__":
    main()
