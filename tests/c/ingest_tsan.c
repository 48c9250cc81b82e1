/* ingest_tsan.c — the batched ingest of the mirror (ykhost_update_nodes_batch / ykhost_update_pods_batch: scanning threads
 * inside, cache bookkeeping on the caller's thread) with libykhost's sources built under ThreadSanitizer, while two reader
 * threads hammer the entry points the core's goroutines use concurrently (Context.IsPodFitNode holds read locks only,
 * /root/reference/pkg/cache/context.go:697,709): any unsynchronised access inside the library is a TSAN report.
 *
 *   ingest_tsan <nodes.ndjson> <pods_on_nodes.ndjson> <asks.ndjson>      (mirror-only handle: device -1)
 * Prints "ingest ok: <asks> asks, <fast> memo hits, <full> parses"; exit code 0. */
#include <pthread.h>
#include <stdatomic.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "ykhost.h"

static ykhost_t* H;
static atomic_int stop_readers;

static char* slurp(const char* path, long* len) {
  FILE* f = fopen(path, "rb");
  if (!f) return NULL;
  fseek(f, 0, SEEK_END);
  *len = ftell(f);
  fseek(f, 0, SEEK_SET);
  char* p = (char*)malloc((size_t)*len + 1);
  if (fread(p, 1, (size_t)*len, f) != (size_t)*len) *len = -1;
  p[*len > 0 ? *len : 0] = 0;
  fclose(f);
  return p;
}
static void* reader(void* arg) {
  long calls = 0;
  char node[256];
  while (!atomic_load(&stop_readers)) {
    (void)ykhost_num_pods(H);
    (void)ykhost_num_nodes(H);
    (void)ykhost_pod_index(H, "pod-0000001");
    (void)ykhost_pod_state(H, "pod-0000002", node, sizeof node);
    ++calls;
  }
  *(long*)arg = calls;
  return NULL;
}

int main(int argc, char** argv) {
  if (argc != 4) return 2;
  char err[256];
  H = ykhost_create(-1, err, sizeof err);
  if (!H) {
    fprintf(stderr, "ykhost_create: %s\n", err);
    return 1;
  }
  long calls[2] = {0, 0};
  pthread_t th[2];
  for (int i = 0; i < 2; ++i) pthread_create(&th[i], NULL, reader, &calls[i]);
  int rc = 0;
  for (int k = 1; k <= 3 && rc == 0; ++k) {
    long len = 0;
    char* text = slurp(argv[k], &len);
    if (!text || len < 0) {
      fprintf(stderr, "cannot read %s\n", argv[k]);
      rc = 1;
      break;
    }
    const int32_t n = k == 1 ? ykhost_update_nodes_batch(H, text, len) : ykhost_update_pods_batch(H, text, len);
    if (n < 0) {
      fprintf(stderr, "batch %d: %s\n", k, ykhost_last_error(H));
      rc = 1;
    }
    memset(text, '#', (size_t)len); /* cgo ownership: the buffer is the caller's again */
    free(text);
  }
  atomic_store(&stop_readers, 1);
  for (int i = 0; i < 2; ++i) pthread_join(th[i], NULL);
  int64_t st[2] = {0, 0};
  {
    /* a full encode of the ingested mirror (mirror-only handle: no device): its node loop runs on the scanning threads when the
     * cluster has >= 4096 nodes */
    char reason[256];
    const int32_t sup = ykhost_ask_supported(H, 0, reason, sizeof reason);
    if (sup < 0) {
      fprintf(stderr, "encode: %s\n", ykhost_last_error(H));
      return 1;
    }
  }
  {
    /* informer resync on the loaded mirror: the same pod documents again — every uid is cached, so the scanning threads look the
     * uids up in the uid index (read-only, several threads) and the ordered pass applies the batch on their hints */
    long len = 0;
    char* text = slurp(argv[2], &len);
    const int32_t n = (text && len >= 0) ? ykhost_update_pods_batch(H, text, len) : -1;
    if (n < 0) {
      fprintf(stderr, "resync batch: %s\n", ykhost_last_error(H));
      rc = 1;
    }
    free(text);
  }
  ykhost_ingest_stats(H, st);
  int64_t tm[5] = {0, 0, 0, 0, 0};
  ykhost_ingest_timing(H, tm);
  printf("ingest ok: %d asks, %lld memo hits, %lld parses, %ld reader calls, bulk batches %lld\n", ykhost_num_pods(H), (long long)st[0], (long long)st[1],
         calls[0] + calls[1], (long long)tm[4]);
  ykhost_destroy(H);
  return rc;
}
