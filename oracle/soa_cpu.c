/*
 * soa_cpu.c — a SECOND, stronger CPU baseline for bench.py's `cpu_baseline` leg ("a batched-SoA CPU
 * variant may be shown as a second, stronger baseline"). TEST / MEASUREMENT INFRASTRUCTURE ONLY: nothing in the product
 * links or loads it.
 *
 * It evaluates the ENCODED tables (the structure-of-arrays form documented in include/ykpred.h, produced by the host
 * encoder) per (pod, node) pair on the host cores with OpenMP: the plugin order and early exit of runFilterPlugins
 * (/root/reference/pkg/plugin/predicates/predicate_manager.go:206-283) for the plugins that need no cluster-wide
 * histogram — NodeAffinity PreFilter, NodeUnschedulable, NodeName, TaintToleration, NodeAffinity, NodePorts,
 * NodeResourcesFit — with none of the GPU path's signature planes or pod classes: every pair is computed.
 * Checked against the object-model oracle in tests/test_soa_cpu.py.
 */
#include <stdint.h>
#include <string.h>

typedef struct soa_tables {
  int32_t N, S, P, R, KT, W, KP;
  const int64_t *allocatable, *requested;     /* [R][N] */
  const int32_t *allowed_pods, *pod_count;    /* [N] */
  const uint32_t* node_flags;                 /* [N] bit0 unschedulable */
  const uint64_t *taint_bits, *label_bits, *port_bits; /* [KT][N], [W][N], [KP][N] */
  const int64_t* requests;                    /* [S][R] */
  const uint64_t* tolerated;                  /* [S][KT] */
  const uint32_t* spec_flags;                 /* [S] */
  const int32_t* aff_term_off;                /* [S+1] */
  const uint64_t* aff_terms;                  /* [..][W] */
  const int32_t* pre_term_off;                /* [S+1] */
  const uint64_t* pre_terms;                  /* [..][W] */
  const uint64_t* wanted_ports;               /* [S][KP] */
  const int32_t *pod_spec, *pod_node_name;    /* [P] */
} soa_tables_t;

enum { UNSCHED = 1, NODE_NAME = 2, TAINT = 4, AFFINITY = 8, PORTS = 16, FIT = 32, SPREAD = 64, INTERPOD = 128 };
enum { SPEC_TOL_UNSCHED = 1, SPEC_AFF_SKIP = 2, SPEC_PRE_REJECT = 4, SPEC_PRE_NAMES = 8 };

static int dnf(const soa_tables_t* t, const uint64_t* terms, int t0, int t1, int n) {
  for (int k = t0; k < t1; ++k) {
    int all = 1;
    for (int w = 0; w < t->W && all; ++w) {
      uint64_t m = terms[(size_t)k * t->W + w];
      all = (t->label_bits[(size_t)w * t->N + n] & m) == m;
    }
    if (all) return 1;
  }
  return 0;
}

static int fits(const soa_tables_t* t, int s, int pin, int n, uint32_t pre, uint32_t filt) {
  const uint32_t f = t->spec_flags[s];
  if ((pre & AFFINITY) && !(f & SPEC_AFF_SKIP)) {
    if (f & SPEC_PRE_REJECT) return 0;
    if ((f & SPEC_PRE_NAMES) && !dnf(t, t->pre_terms, t->pre_term_off[s], t->pre_term_off[s + 1], n)) return 0;
  }
  if ((filt & UNSCHED) && (t->node_flags[n] & 1u) && !(f & SPEC_TOL_UNSCHED)) return 0;
  if ((filt & NODE_NAME) && pin != -1 && pin != n) return 0;
  if (filt & TAINT)
    for (int k = 0; k < t->KT; ++k)
      if (t->taint_bits[(size_t)k * t->N + n] & ~t->tolerated[(size_t)s * t->KT + k]) return 0;
  if (filt & AFFINITY) {
    int skip = (pre & AFFINITY) && (f & SPEC_AFF_SKIP);
    if (!skip && !dnf(t, t->aff_terms, t->aff_term_off[s], t->aff_term_off[s + 1], n)) return 0;
  }
  if (filt & PORTS) {
    if (!(pre & PORTS)) return 0;
    for (int k = 0; k < t->KP; ++k)
      if (t->port_bits[(size_t)k * t->N + n] & t->wanted_ports[(size_t)s * t->KP + k]) return 0;
  }
  if (filt & FIT) {
    if (!(pre & FIT)) return 0;
    if ((int64_t)t->pod_count[n] + 1 > (int64_t)t->allowed_pods[n]) return 0;
    for (int r = 0; r < t->R; ++r) {
      int64_t q = t->requests[(size_t)s * t->R + r];
      if (q > 0 && q > t->allocatable[(size_t)r * t->N + n] - t->requested[(size_t)r * t->N + n]) return 0;
    }
  }
  if ((filt & SPREAD) && !(pre & SPREAD)) return 0;
  if ((filt & INTERPOD) && !(pre & INTERPOD)) return 0;
  return 1;
}

/* bitmap[P][words] (words = ceil(N/64)), bit n&63 of word n>>6 = pod p fits node n. Returns the number of pairs. */
int64_t soa_eval(const soa_tables_t* t, uint32_t pre, uint32_t filt, int32_t threads, uint64_t* bitmap) {
  const int words = (t->N + 63) / 64;
#pragma omp parallel for schedule(dynamic, 16) num_threads(threads > 0 ? threads : 1)
  for (int p = 0; p < t->P; ++p) {
    uint64_t* row = bitmap + (size_t)p * words;
    memset(row, 0, (size_t)words * sizeof(uint64_t));
    const int s = t->pod_spec[p], pin = t->pod_node_name[p];
    for (int n = 0; n < t->N; ++n)
      if (fits(t, s, pin, n, pre, filt)) row[n >> 6] |= 1ull << (n & 63);
  }
  return (int64_t)t->P * t->N;
}
