// fill_probe5.hip — expansion of a class-ordered bitmap with the EXACT store pattern of the linear fill and (almost) no loads.
// Geometry: G workgroups x T threads, tile = T*16 bytes, and the window G*tile is a whole number R of bitmap rows
// (784-word rows: 245 x 4 KiB = 160 rows). Then every thread owns a FIXED 16-byte column group and a fixed row offset inside
// the window: step k writes row R*k + ro at that column — the thread keeps the 16 bytes of "its" column of the current class
// in a register and reloads them (from the class's representative row, written by the first pass) only when the class of
// its row changes. The class lookups are fetched in bulk: one vector load brings the src_row[] entries of 64 future steps
// (lane i = step i), the step loop reads them with v_readlane. Steady state per step: one dwordx4 store, a few VALU ops.
// Build+run: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe5.hip -o /tmp/fill_probe5 && /tmp/fill_probe5
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int T>
__global__ __launch_bounds__(T) void fill_linear(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  for (size_t i = (size_t)blockIdx.x * T + threadIdx.x; i < n16; i += (size_t)gridDim.x * T) p[i] = val;
}
__global__ __launch_bounds__(256) void write_reps(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ class_first,
                                                  int n_classes, int W) {
  const int c = blockIdx.x;
  if (c >= n_classes) return;
  u64* row = out + (size_t)class_first[c] * W;
  for (int w = threadIdx.x * 2; w < W; w += 512) *(u64x2*)(row + w) = *(const u64x2*)(tab + (size_t)c * W + w);
}

// src_row has (n_steps + 128) * R + 2 entries; entries past n_rows are -1. The bitmap holds n_steps * R rows; n_steps is a
// multiple of D. Per lane: `val` = the 16 bytes of its column in the class at the STORE cursor; ld[j] / fresh bit j = value
// loaded for the step in ring slot j when the class changed there (prefetch cursor, D steps ahead). No value is copied
// between ring slots, so nothing waits on a load that was just issued, and the steady-state stores are unconditional.
template <int T, int D>
__global__ __launch_bounds__(T) void expand_fixed(u64* __restrict__ out, const int* __restrict__ src_row, int n_steps, int R, int row_b) {
  const int p = blockIdx.x * (T * 16) + threadIdx.x * 16;  // byte offset inside the window
  const int ro = p / row_b, col = p - ro * row_b;
  const int lane = threadIdx.x & 63;
  const int ro0 = __builtin_amdgcn_readfirstlane(ro);  // lanes of a wave sit in row ro0 or ro0 + 1
  const bool second = ro != ro0;
  const long window_b = (long)R * row_b;
  const char* base = (const char*)out;
  u64x2 ld[D];
#pragma unroll
  for (int j = 0; j < D; ++j) ld[j] = u64x2{0, 0};
  u64x2 val = {0, 0};
  unsigned fresh = 0;
  int cur_src = -2;
  char* wr = (char*)out + p;  // store cursor
  int cur0 = src_row[(long)R * lane + ro0], cur1 = src_row[(long)R * lane + ro0 + 1];
#define PREFETCH(j, sel)                                                                       \
  {                                                                                            \
    const int s0 = __builtin_amdgcn_readlane(cur0, (sel)), s1 = __builtin_amdgcn_readlane(cur1, (sel)); \
    const int src = second ? s1 : s0;                                                          \
    const bool need = src >= 0 && src != cur_src;                                              \
    if (need) ld[j] = *(const u64x2*)(base + (size_t)src * row_b + col);                       \
    fresh = need ? (fresh | (1u << (j))) : (fresh & ~(1u << (j)));                             \
    cur_src = need ? src : cur_src;                                                            \
  }
#define STORE(j)                                   \
  {                                                \
    if (fresh & (1u << (j))) val = ld[j];          \
    *(u64x2*)wr = val;                             \
    wr += window_b;                                \
  }
#pragma unroll
  for (int j = 0; j < D; ++j) PREFETCH(j, j)
  for (int kb0 = 0; kb0 < n_steps; kb0 += 64) {
    const int nxt0 = src_row[(long)R * (kb0 + 64 + lane) + ro0], nxt1 = src_row[(long)R * (kb0 + 64 + lane) + ro0 + 1];
    for (int i0 = kb0 == 0 ? D : 0; i0 < 64 && kb0 + i0 < n_steps; i0 += D) {
#pragma unroll
      for (int j = 0; j < D; ++j) {
        STORE(j)
        PREFETCH(j, i0 + j)
      }
    }
    cur0 = nxt0;
    cur1 = nxt1;
  }
#pragma unroll
  for (int j = 0; j < D; ++j) STORE(j)
#undef PREFETCH
#undef STORE
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
  u64* d; CK(hipMalloc(&d, bytes + (64 << 20)));
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  for (int C : {2061, 10000, 126418}) {
    const long pad_rows = rows + 1024 * 600;
    std::vector<int> h_rc(rows), h_first(C), h_src(pad_rows, -1);
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
    CK(hipMalloc(&sr, pad_rows * 4)); CK(hipMemcpy(sr, h_src.data(), pad_rows * 4, hipMemcpyHostToDevice));
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
    run("linear grid-stride fill, 256 x 256", false, [&] { fill_linear<256><<<256, 256>>>((u64x2*)d, bytes / 16, 7); });
    run("linear grid-stride fill, 245 x 256", false, [&] { fill_linear<256><<<245, 256>>>((u64x2*)d, bytes / 16, 7); });
    run("linear grid-stride fill, 196 x 256", false, [&] { fill_linear<256><<<196, 256>>>((u64x2*)d, bytes / 16, 7); });
    run("linear grid-stride fill, 147 x 512", false, [&] { fill_linear<512><<<147, 512>>>((u64x2*)d, bytes / 16, 7); });
    // G * T * 16 = R * 6272
#define RUN(T, D, G, R)                                                                                    \
    snprintf(nm, 128, "reps + expand_fixed T=%d D=%d G=%d R=%d", T, D, G, R);                              \
    run(nm, true, [&] { write_reps<<<C, 256>>>(d, tab, cf, C, W);                                         \
                        expand_fixed<T, D><<<G, T>>>(d, sr, (int)(((rows + R - 1) / R + D - 1) / D * D), R, W * 8); });
    RUN(256, 4, 245, 160) RUN(256, 8, 245, 160) RUN(256, 16, 245, 160) RUN(256, 32, 245, 160)
    RUN(256, 8, 196, 128) RUN(256, 16, 196, 128) RUN(256, 16, 147, 96)
    RUN(512, 8, 147, 192) RUN(512, 16, 147, 192) RUN(512, 16, 98, 128) RUN(512, 16, 196, 256)
    RUN(128, 16, 490, 160) RUN(128, 32, 245, 80)
    (void)hipFree(tab); (void)hipFree(rc); (void)hipFree(cf); (void)hipFree(sr);
  }
  return 0;
}
