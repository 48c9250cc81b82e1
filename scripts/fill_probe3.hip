// fill_probe3.hip — can a CLASS-ORDERED bitmap be expanded at the hipMemset rate?
// fill_probe2 showed: only the grid-stride walk over 4 KiB tiles by <= 256 workgroups reaches 6.4 TB/s on this part;
// contiguous per-workgroup regions, row-shaped writes and LDS-staged class rows all plateau at 5.4 TB/s or far below.
// This probe keeps that walk and sources every 16-byte group from the REPRESENTATIVE ROW of the row's class, which lives
// in the bitmap itself (the first row of the class run, written by a first pass): src_row[row] → 16 B load (L1/L2 hit:
// all workgroups of a 1 MiB window read the same 1-2 class rows) → 16 B store. The two dependent loads are software-
// pipelined in batches of U tiles so that a workgroup keeps U KiB-sized loads in flight.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe3.hip -o /tmp/fill_probe3 && /tmp/fill_probe3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void fill_linear(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = val;
}

// first pass stand-in: representative rows get their pattern (tab row of the class)
__global__ __launch_bounds__(256) void write_reps(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ class_first,
                                                  int n_classes, int W) {
  const int c = blockIdx.x;
  if (c >= n_classes) return;
  u64* row = out + (size_t)class_first[c] * W;
  for (int w = threadIdx.x * 2; w < W; w += 512) *(u64x2*)(row + w) = *(const u64x2*)(tab + (size_t)c * W + w);
}

// Ring pipeline, branch-free and workgroup-uniform: step k of a workgroup is tile blockIdx.x + k*gridDim.x. Slot j = k % D:
//   store v[j] (data of step k, loaded D steps ago)  →  v[j] = load data of step k+D (needs s[j], loaded 2D steps ago)
//   →  s[j] = load src_row of step k+2D.
// gfx9 counts loads AND stores in one in-order vmcnt, so a load issued behind a store completes only after that store has
// been acknowledged: the ring has to be deep enough to cover the store latency (a shallow pipeline stalls on it).
// The buffer is padded to whole tiles; cursors past the last step repeat the last step (idempotent).
struct Cursor {
  int row, col;
  long off;
};
template <int T, int D>
__global__ __launch_bounds__(T) void expand_pipe(u64* __restrict__ out, const int* __restrict__ src_row, long n_rows, int W) {
  const int row_b = W * 8;
  const long total_b = n_rows * (long)row_b;
  constexpr long kTileB = (long)T * 16;
  const long n_tiles = (total_b + kTileB - 1) / kTileB;
  const long stride_b = (long)gridDim.x * kTileB;
  const int drow = (int)(stride_b / row_b), dcol = (int)(stride_b - (long)drow * row_b);
  if ((long)blockIdx.x >= n_tiles) return;
  const int n_steps = (int)((n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x);  // uniform per workgroup
  const int last_row = (int)n_rows - 1;
  Cursor a, b, c;
  a.off = (long)blockIdx.x * kTileB + threadIdx.x * 16;
  a.row = (int)(a.off / row_b);
  a.col = (int)(a.off - (long)a.row * row_b);
  b = c = a;
  auto advance = [&](Cursor& q, bool go) {  // `go` is workgroup-uniform
    int nr = q.row + drow, nc = q.col + dcol;
    if (nc >= row_b) { nc -= row_b; ++nr; }
    q.row = go ? nr : q.row;
    q.col = go ? nc : q.col;
    q.off = go ? q.off + stride_b : q.off;
  };
  int ka = 0, kb = 0;  // step index of cursors a, b (clamped to n_steps - 1)
  int s[D];
  u64x2 v[D];
  // prologue: src of steps 0..D-1, then data of steps 0..D-1 and src of steps D..2D-1
#pragma unroll
  for (int j = 0; j < D; ++j) {
    s[j] = src_row[min(a.row, last_row)];
    advance(a, ka < n_steps - 1);
    ka = min(ka + 1, n_steps - 1);
  }
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const int rb = min(b.row, last_row);
    const int sr = s[j] < 0 ? rb : s[j];
    v[j] = *(const u64x2*)((const char*)out + (size_t)sr * row_b + b.col);
    advance(b, kb < n_steps - 1);
    kb = min(kb + 1, n_steps - 1);
    s[j] = src_row[min(a.row, last_row)];
    advance(a, ka < n_steps - 1);
    ka = min(ka + 1, n_steps - 1);
  }
  for (int k0 = 0; k0 < n_steps; k0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (k0 + j < n_steps) *(u64x2*)((char*)out + c.off) = v[j];  // uniform predicate (only the last round is partial)
      advance(c, true);
      const int rb = min(b.row, last_row);
      const int sr = s[j] < 0 ? rb : s[j];
      v[j] = *(const u64x2*)((const char*)out + (size_t)sr * row_b + b.col);
      advance(b, kb < n_steps - 1);
      kb = min(kb + 1, n_steps - 1);
      s[j] = src_row[min(a.row, last_row)];
      advance(a, ka < n_steps - 1);
      ka = min(ka + 1, n_steps - 1);
    }
  }
}

__global__ void verify(const u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ row_class, long n_rows, int W,
                       unsigned long long* bad) {
  const long total = n_rows * W;
  unsigned long long b = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / W;
    int col = (int)(i - row * W);
    if (out[i] != tab[(size_t)row_class[row] * W + col]) ++b;
  }
  if (b) atomicAdd(bad, b);
}

int main() {
  const long rows = 1000000;
  const int W = 784;
  const size_t bytes = (size_t)rows * W * 8;
  u64* d; CK(hipMalloc(&d, bytes + (1 << 20)));
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  for (int C : {2061, 10000, 126418}) {
    std::vector<int> h_rc(rows), h_first(C), h_src(rows);
    {
      std::vector<double> wgt(C); double tot = 0; unsigned long long s = 88172645463325252ull + C;
      for (int c = 0; c < C; ++c) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; wgt[c] = 0.2 + (double)(s % 1000) / 500.0; tot += wgt[c]; }
      long r = 0;
      for (int c = 0; c < C; ++c) {
        long n = c == C - 1 ? rows - r : (long)(wgt[c] / tot * rows);
        if (n < 1) n = 1;
        if (r + n > rows - (C - 1 - c)) n = rows - (C - 1 - c) - r;
        h_first[c] = (int)r;
        for (long k = 0; k < n; ++k) { h_rc[r] = c; h_src[r] = h_first[c]; ++r; }
      }
    }
    std::vector<u64> h_tab((size_t)C * W);
    { unsigned long long s = 1234567; for (auto& x : h_tab) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = s; } }
    u64* tab; int *rc, *cf, *sr;
    CK(hipMalloc(&tab, h_tab.size() * 8)); CK(hipMemcpy(tab, h_tab.data(), h_tab.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&rc, rows * 4)); CK(hipMemcpy(rc, h_rc.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sr, rows * 4)); CK(hipMemcpy(sr, h_src.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&cf, C * 4)); CK(hipMemcpy(cf, h_first.data(), C * 4, hipMemcpyHostToDevice));
    printf("---- %d classes (avg %.1f rows)\n", C, (double)rows / C);
    auto run = [&](const char* name, bool check, auto launch) {
      if (check) {
        (void)hipMemset(d, 0, bytes); (void)hipMemset(bad, 0, 8);
        launch();
        verify<<<4096, 256>>>(d, tab, rc, rows, W, bad);
        unsigned long long hb = 0; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
        if (hb) printf("  !! %s: %llu wrong words\n", name, hb);
      }
      for (int i = 0; i < 2; ++i) launch();
      (void)hipEventRecord(ev0);
      for (int i = 0; i < 8; ++i) launch();
      (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
      float ms; (void)hipEventElapsedTime(&ms, ev0, ev1); ms /= 8;
      printf("%-60s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
      fflush(stdout);
    };
    char nm[128];
    run("hipMemsetAsync", false, [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
    run("linear grid-stride fill, 256 blocks", false, [&] { fill_linear<<<256, 256>>>((u64x2*)d, bytes / 16, 7); });
    run("reps only (first pass stand-in)", false, [&] { write_reps<<<C, 256>>>(d, tab, cf, C, W); });
#define RUN(T, U, G)                                                                                       \
    snprintf(nm, 128, "reps + expand_pipe T=%d D=%d G=%d", T, U, G);                                       \
    run(nm, true, [&] { write_reps<<<C, 256>>>(d, tab, cf, C, W); expand_pipe<T, U><<<G, T>>>(d, sr, rows, W); });
    RUN(256, 2, 256) RUN(256, 4, 256) RUN(256, 8, 256) RUN(256, 12, 256) RUN(256, 16, 256) RUN(256, 20, 256)
    RUN(256, 8, 128) RUN(256, 16, 128) RUN(256, 20, 128)
    RUN(256, 4, 512) RUN(256, 8, 512) RUN(256, 16, 512)
    RUN(512, 4, 128) RUN(512, 8, 128) RUN(512, 16, 128) RUN(512, 8, 256)
    RUN(128, 16, 256) RUN(128, 20, 512)
    (void)hipFree(tab); (void)hipFree(rc); (void)hipFree(cf); (void)hipFree(sr);
  }
  return 0;
}
