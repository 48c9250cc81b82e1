// fill_probe4.hip — wave-granular variant of fill_probe3's expansion walk: the position of a wave's 1 KiB tile (byte
// offset, first row, column) is wave-uniform and lives in SGPRs; src_row[] of the (at most two) rows a tile touches comes
// through the scalar cache (s_load: its own counter, never queued behind stores); per lane only the row-wrap select and
// the address of the 16-byte load remain. MODE: 0 = full, 1 = no data load (stores the src index: ALU/SMEM/store cost only),
// 2 = data always from row 0 (load path hot in L1).
// Build+run: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe4.hip -o /tmp/fill_probe4 && /tmp/fill_probe4
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
__global__ __launch_bounds__(256) void write_reps(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ class_first,
                                                  int n_classes, int W) {
  const int c = blockIdx.x;
  if (c >= n_classes) return;
  u64* row = out + (size_t)class_first[c] * W;
  for (int w = threadIdx.x * 2; w < W; w += 512) *(u64x2*)(row + w) = *(const u64x2*)(tab + (size_t)c * W + w);
}

struct SCursor {  // wave-uniform
  long off;
  int row, col;
};
template <int T, int D, int MODE>
__global__ __launch_bounds__(T) void expand_wave(u64* __restrict__ out, const int* __restrict__ src_row /* [n_rows + 1] */, long n_rows,
                                                 int W) {
  const int row_b = W * 8;  // >= 1024: a 1 KiB wave tile touches at most two rows
  const long total_b = n_rows * (long)row_b;
  constexpr int kWaves = T / 64;
  const long n_tiles = (total_b + 1023) / 1024;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int lane16 = (threadIdx.x & 63) * 16;
  const long gw = (long)blockIdx.x * kWaves + wave;
  const long n_waves = (long)gridDim.x * kWaves;
  if (gw >= n_tiles) return;
  const int n_steps = (int)((n_tiles - gw + n_waves - 1) / n_waves);
  const long stride_b = n_waves * 1024;
  const int drow = (int)(stride_b / row_b), dcol = (int)(stride_b - (long)drow * row_b);
  const int last_row = (int)n_rows - 1;
  SCursor a, b, c;
  a.off = gw * 1024;
  a.row = (int)(a.off / row_b);
  a.col = (int)(a.off - (long)a.row * row_b);
  b = c = a;
  auto advance = [&](SCursor& q, bool go) {
    int nr = q.row + drow, nc = q.col + dcol;
    if (nc >= row_b) { nc -= row_b; ++nr; }
    q.row = go ? nr : q.row;
    q.col = go ? nc : q.col;
    q.off = go ? q.off + stride_b : q.off;
  };
  int ka = 0, kb = 0;
  int s0[D], s1[D];
  u64x2 v[D];
  auto load_src = [&](int j) {
    const int r = min(a.row, last_row);
    s0[j] = src_row[r];
    s1[j] = src_row[r + 1];
    advance(a, ka < n_steps - 1);
    ka = min(ka + 1, n_steps - 1);
  };
  auto load_data = [&](int j) {
    int colL = b.col + lane16;
    const bool over = colL >= row_b;
    colL -= over ? row_b : 0;
    const int own = min(b.row + (over ? 1 : 0), last_row);
    int sr = over ? s1[j] : s0[j];
    sr = sr < 0 ? own : sr;
    if (MODE == 0) v[j] = *(const u64x2*)((const char*)out + (size_t)sr * row_b + colL);
    if (MODE == 1) v[j] = u64x2{(u64)sr, (u64)colL};
    if (MODE == 2) v[j] = *(const u64x2*)((const char*)out + colL);
    advance(b, kb < n_steps - 1);
    kb = min(kb + 1, n_steps - 1);
  };
#pragma unroll
  for (int j = 0; j < D; ++j) load_src(j);
#pragma unroll
  for (int j = 0; j < D; ++j) {
    load_data(j);
    load_src(j);
  }
  for (int k0 = 0; k0 < n_steps; k0 += D) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      if (k0 + j < n_steps) *(u64x2*)((char*)out + c.off + lane16) = v[j];
      advance(c, true);
      load_data(j);
      load_src(j);
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
  for (int C : {2061, 126418}) {
    std::vector<int> h_rc(rows), h_first(C), h_src(rows + 1);
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
      h_src[rows] = -1;
    }
    std::vector<u64> h_tab((size_t)C * W);
    { unsigned long long s = 1234567; for (auto& x : h_tab) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = s; } }
    u64* tab; int *rc, *cf, *sr;
    CK(hipMalloc(&tab, h_tab.size() * 8)); CK(hipMemcpy(tab, h_tab.data(), h_tab.size() * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&rc, rows * 4)); CK(hipMemcpy(rc, h_rc.data(), rows * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&sr, (rows + 1) * 4)); CK(hipMemcpy(sr, h_src.data(), (rows + 1) * 4, hipMemcpyHostToDevice));
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
#define RUN(T, D, G, MODE)                                                                                 \
    snprintf(nm, 128, "reps + expand_wave T=%d D=%d G=%d mode=%d", T, D, G, MODE);                         \
    run(nm, MODE == 0, [&] { write_reps<<<C, 256>>>(d, tab, cf, C, W); expand_wave<T, D, MODE><<<G, T>>>(d, sr, rows, W); });
    RUN(256, 8, 256, 1) RUN(256, 8, 256, 2) RUN(256, 16, 256, 2)
    RUN(256, 4, 256, 0) RUN(256, 8, 256, 0) RUN(256, 12, 256, 0) RUN(256, 16, 256, 0) RUN(256, 24, 256, 0)
    RUN(256, 8, 128, 0) RUN(256, 16, 128, 0)
    RUN(256, 8, 512, 0) RUN(256, 16, 512, 0)
    RUN(512, 8, 128, 0) RUN(512, 16, 128, 0) RUN(512, 8, 256, 0)
    RUN(128, 16, 256, 0) RUN(128, 16, 512, 0) RUN(64, 16, 1024, 0) RUN(64, 16, 512, 0)
    (void)hipFree(tab); (void)hipFree(rc); (void)hipFree(cf); (void)hipFree(sr);
  }
  return 0;
}
