/* replay_cgo_sequence.c — the call sequence and STRING OWNERSHIP of integration/gpu_predicate_manager.go, replayed from C.
 *
 * cgo rules the Go file lives by: every string it passes is a C.CString (malloc) that is freed right after the call
 * returns; every buffer it receives into is its own; nothing it passed may be read by the library afterwards. This harness
 * makes the same calls in the same order with exactly that ownership — each string is malloc'ed, used for ONE call, poisoned
 * and freed — so that, built together with libykhost's sources under AddressSanitizer (tests/test_abi_symbols.py), any
 * pointer the library retains shows up as a heap-use-after-free on a later call.
 *
 *   replay_cgo_sequence <device>     device < 0: mirror-only handle (CPU suite: evaluations fail, which is the path on which
 *                                    the Go manager routes the call to the CPU predicate manager — RoutedOnError);
 *                                    device >= 0: the engine answers (GPU suite).
 * Exit code 0 = the sequence behaved as the Go file expects.
 */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ykhost.h"

static int failures = 0;
#define EXPECT(cond, what)                                             \
  do {                                                                 \
    if (!(cond)) {                                                     \
      fprintf(stderr, "FAIL %s:%d %s\n", __FILE__, __LINE__, what);    \
      failures++;                                                      \
    }                                                                  \
  } while (0)

/* C.CString: a malloc'ed copy the caller owns */
static char* cstring(const char* s) {
  size_t n = strlen(s) + 1;
  char* p = (char*)malloc(n);
  memcpy(p, s, n);
  return p;
}
/* C.free right after the call; the bytes are overwritten first so that a stale read cannot accidentally still look right */
static void cfree(char* p) {
  memset(p, '#', strlen(p));
  free(p);
}

static char* node_json(const char* name, const char* zone, const char* cpu, int tainted) {
  char buf[1024];
  snprintf(buf, sizeof buf,
           "{\"kind\":\"Node\",\"apiVersion\":\"v1\",\"metadata\":{\"name\":\"%s\",\"creationTimestamp\":null,\"labels\":{\"kubernetes.io/hostname\":\"%s\","
           "\"topology.kubernetes.io/zone\":\"%s\"}},\"spec\":{%s},\"status\":{\"allocatable\":{\"cpu\":\"%s\",\"memory\":\"8Gi\",\"pods\":\"10\"},"
           "\"daemonEndpoints\":{\"kubeletEndpoint\":{\"Port\":0}},\"nodeInfo\":{\"machineID\":\"\"}}}",
           name, name, zone, tainted ? "\"taints\":[{\"key\":\"dedicated\",\"value\":\"infra\",\"effect\":\"NoSchedule\"}]" : "", cpu);
  return cstring(buf);
}
static char* pod_json(const char* uid, const char* cpu, const char* node_name, const char* phase) {
  char buf[1024];
  snprintf(buf, sizeof buf,
           "{\"kind\":\"Pod\",\"apiVersion\":\"v1\",\"metadata\":{\"name\":\"%s\",\"namespace\":\"default\",\"uid\":\"%s\",\"creationTimestamp\":null,"
           "\"labels\":{\"applicationId\":\"app-1\",\"queue\":\"root.default\"}},\"spec\":{\"containers\":[{\"name\":\"c\",\"image\":\"pause\","
           "\"resources\":{\"requests\":{\"cpu\":\"%s\",\"memory\":\"1Gi\"}}}],\"schedulerName\":\"yunikorn\"%s%s%s},\"status\":{%s%s%s}}",
           uid, uid, cpu, node_name[0] ? ",\"nodeName\":\"" : "", node_name, node_name[0] ? "\"" : "", phase[0] ? "\"phase\":\"" : "", phase,
           phase[0] ? "\"" : "");
  return cstring(buf);
}

/* gpuPredicateManager.Predicates: pod_index + node_index by freshly allocated names, then ykhost_predicates into own buffers */
static int predicates(ykhost_t* h, const char* uid_s, const char* node_s, int allocate, char* plugin_out, size_t plugin_len) {
  char* uid = cstring(uid_s);
  char* name = cstring(node_s);
  int32_t pod = ykhost_pod_index(h, uid), node = ykhost_node_index(h, name);
  int rc = -100; /* "RoutedNotMirrored" */
  if (pod >= 0 && node >= 0) {
    char plugin[64], message[1024];
    rc = ykhost_predicates(h, pod, node, allocate, plugin, 64, message, 1024);
    if (rc == 0) snprintf(plugin_out, plugin_len, "%s", plugin);
  }
  cfree(name);
  cfree(uid);
  return rc;
}

int main(int argc, char** argv) {
  const int device = argc > 1 ? atoi(argv[1]) : -1;
  char* err = (char*)malloc(512); /* errBuf := C.malloc(512); defer C.free */
  ykhost_t* h = ykhost_create(device, err, 512);
  if (!h) {
    fprintf(stderr, "ykhost_create(%d): %s\n", device, err);
    free(err);
    return device >= 0 ? 2 : 1;
  }
  free(err);
  EXPECT(ykpred_abi_version() == YKPRED_ABI_VERSION, "ABI version of the loaded engine");

  /* ---- InitializeState-style replay (context.go:1411-1484): nodes first, then pods, one OnUpdate* per object */
  char* t;
  t = node_json("node-a", "z1", "4", 0);
  EXPECT(ykhost_update_node(h, t) == 0, "OnUpdateNode(node-a): no orphan adopted");
  cfree(t);
  t = node_json("node-b", "z2", "4", 1);
  EXPECT(ykhost_update_node(h, t) == 0, "OnUpdateNode(node-b)");
  cfree(t);
  t = pod_json("running-1", "3", "node-a", "Running"); /* a pod the cluster already bound: accounted on node-a */
  EXPECT(ykhost_update_pod(h, t) == 1, "OnUpdatePod(running-1)");
  cfree(t);
  t = pod_json("orphan-1", "1", "node-c", "Running"); /* names a node that is not known yet: stored as an orphan (→ 0) */
  EXPECT(ykhost_update_pod(h, t) == 0, "OnUpdatePod(orphan-1) is an orphan");
  cfree(t);
  t = pod_json("ask-1", "2", "", "Pending");
  EXPECT(ykhost_update_pod(h, t) == 1, "OnUpdatePod(ask-1)");
  cfree(t);
  t = pod_json("ask-2", "500m", "", "Pending");
  EXPECT(ykhost_update_pod(h, t) == 1, "OnUpdatePod(ask-2)");
  cfree(t);
  t = node_json("node-c", "z1", "8", 0); /* the orphan's node arrives: adopted */
  EXPECT(ykhost_update_node(h, t) == 1, "OnUpdateNode(node-c) adopts orphan-1");
  cfree(t);

  char plugin[64] = "";
  if (device >= 0) {
    /* ---- Predicates(): ask-1 wants 2 cpu; node-a has 4 - 3 = 1 free → NodeResourcesFit; node-b is tainted; node-c fits */
    EXPECT(predicates(h, "ask-1", "node-a", 1, plugin, sizeof plugin) == 0 && strcmp(plugin, "NodeResourcesFit") == 0, "ask-1 on node-a: NodeResourcesFit");
    EXPECT(predicates(h, "ask-1", "node-b", 1, plugin, sizeof plugin) == 0 && strcmp(plugin, "TaintToleration") == 0, "ask-1 on node-b: TaintToleration");
    EXPECT(predicates(h, "ask-1", "node-c", 1, plugin, sizeof plugin) == 1, "ask-1 on node-c fits");
    EXPECT(predicates(h, "ask-2", "node-a", 1, plugin, sizeof plugin) == 1, "ask-2 on node-a fits");
    EXPECT(predicates(h, "ask-1", "node-c", 0, plugin, sizeof plugin) == 1, "reservation phase");
  } else {
    /* mirror-only handle: the engine call fails, the Go manager counts RoutedOnError and asks the CPU manager */
    EXPECT(predicates(h, "ask-1", "node-c", 1, plugin, sizeof plugin) < 0, "no device: Predicates() reports an engine error");
    EXPECT(strlen(ykhost_last_error(h)) > 0, "ykhost_last_error carries the reason");
  }
  EXPECT(predicates(h, "no-such-pod", "node-a", 1, plugin, sizeof plugin) == -100, "unknown pod: RoutedNotMirrored");

  /* ---- PreemptionPredicates(): victims as a calloc'ed array of C strings, a nil victim stays NULL */
  {
    char* uid = cstring("ask-1");
    char* name = cstring("node-a");
    int32_t pod = ykhost_pod_index(h, uid), node = ykhost_node_index(h, name);
    char* reason = (char*)calloc(600, 1);
    int supported = device >= 0 ? ykhost_ask_supported(h, pod, reason, 600) : 1;
    EXPECT(supported == 1, "ask-1 is evaluated by the engine");
    char** victims = (char**)calloc(2, sizeof(char*));
    victims[0] = NULL;
    victims[1] = cstring("running-1");
    int32_t index = ykhost_preemption_predicates(h, pod, node, (const char* const*)victims, 2, 0);
    if (device >= 0)
      EXPECT(index == 1, "removing running-1 (victim 1) makes room on node-a");
    else
      EXPECT(index < -1, "no device: engine error");
    cfree(victims[1]);
    free(victims);
    free(reason);
    cfree(name);
    cfree(uid);
  }

  /* ---- a scheduling decision: AssumePod (the pod carries the node), then the answers of the next ask move */
  {
    char* uid = cstring("ask-1");
    char* name = cstring("node-c");
    EXPECT(ykhost_assume_pod(h, uid, name) == 0, "OnAssumePod(ask-1 → node-c)");
    cfree(name);
    cfree(uid);
  }
  if (device >= 0) {
    int32_t patched = -2;
    EXPECT(ykhost_evaluate_dirty(h, 1, YKPRED_OUT_BITMAP | YKPRED_OUT_COUNTS | YKPRED_OUT_DECISIONS, &patched) == 0, "Refresh()");
    EXPECT(predicates(h, "ask-2", "node-c", 1, plugin, sizeof plugin) == 1, "ask-2 still fits node-c (8 - 1 - 2 = 5 cpu free)");
    int32_t cand[2] = {-1, -1};
    char* uid = cstring("ask-2");
    int32_t pod = ykhost_pod_index(h, uid);
    cfree(uid);
    EXPECT(ykhost_candidates(h, pod, 1, 2, cand) == 2, "two feasible nodes for ask-2 in bin-pack order");
    int64_t st[5];
    ykhost_resident_stats(h, st);
    EXPECT(st[0] + st[1] >= 1, "the callback after the refresh was served from the resident answer");
  }
  {
    char* uid = cstring("ask-1");
    EXPECT(ykhost_forget_pod(h, uid) == 1, "OnForgetPod(ask-1)");
    cfree(uid);
    uid = cstring("ask-2");
    EXPECT(ykhost_remove_pod(h, uid) == 1, "OnRemovePod(ask-2)");
    cfree(uid);
    uid = cstring("ask-2");
    EXPECT(ykhost_remove_pod(h, uid) == 0, "OnRemovePod of an unknown pod is ignored");
    cfree(uid);
    t = pod_json("running-1", "3", "node-a", "Succeeded"); /* terminated: leaves every map */
    EXPECT(ykhost_update_pod(h, t) == 1, "OnUpdatePod(running-1 Succeeded)");
    cfree(t);
    char* name = cstring("node-b");
    EXPECT(ykhost_remove_node(h, name) == 0, "OnRemoveNode(node-b): nothing orphaned");
    cfree(name);
  }
  int64_t routing[3];
  ykhost_routing_stats(h, routing);
  ykhost_destroy(h); /* Close() */
  if (failures == 0) printf("replay ok (device %d)\n", device);
  return failures ? 3 : 0;
}
