// fill_probe2.hip — write-pattern probes for a CLASS-ORDERED bitmap (rows of one pod class are contiguous), configs[2]
// shape: 1M rows x 784 words (6272 B), 2061 classes of random size. Question: which walk of the 6.27 GB output reaches the
// hipMemset rate when every row must carry its class's 6272-byte pattern?
//   regs   : G persistent workgroups, each owns a contiguous range of rows; the class row is held in registers and
//            written row after row (2 dwordx4 stores per thread and row at 256 threads).
//   lds/S  : the output is cut into 4 KiB tiles; spans of S consecutive tiles are dealt round-robin to G workgroups
//            (S = 1: the grid-stride walk of the plain linear fill; S = all: one contiguous region per workgroup). The
//            class row lives in LDS and is re-read at (byte offset mod 6272); it is reloaded when the class changes.
// Every variant is verified against the expected pattern once.
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe2.hip -o /tmp/fill_probe2 && /tmp/fill_probe2
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kW = 784;            // words per row
constexpr long kRowB = kW * 8;     // 6272

__global__ __launch_bounds__(256) void fill_linear(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = val;
}

// regs: rows [r0, r1) of workgroup g; classes are contiguous row runs: class_end[c] = first row after class c
template <int T>
__global__ __launch_bounds__(T) void expand_regs(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ row_class,
                                                 const int* __restrict__ class_end, long n_rows) {
  const long per = (n_rows + gridDim.x - 1) / gridDim.x;
  long r = (long)blockIdx.x * per;
  const long rend = min(n_rows, r + per);
  constexpr int U = (kW / 2 + T - 1) / T;
  while (r < rend) {
    const int c = row_class[r];
    const long e = min((long)class_end[c], rend);
    u64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      int w = (u * T + threadIdx.x) * 2;
      v[u] = w < kW ? *(const u64x2*)(tab + (size_t)c * kW + w) : u64x2{0, 0};
    }
    for (; r < e; ++r) {
      u64* row = out + (size_t)r * kW;
#pragma unroll
      for (int u = 0; u < U; ++u) {
        int w = (u * T + threadIdx.x) * 2;
        if (w < kW) *(u64x2*)(row + w) = v[u];
      }
    }
  }
}

// lds: spans of S tiles (tile = T*16 bytes) dealt round-robin; class row in LDS
template <int T>
__global__ __launch_bounds__(T) void expand_lds(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ row_class,
                                                const int* __restrict__ class_end, long n_rows, long span_tiles) {
  __shared__ u64 lrow[kW];
  constexpr long kTileB = (long)T * 16;
  const long total_b = n_rows * kRowB;
  const long total_tiles = (total_b + kTileB - 1) / kTileB;
  const long n_spans = (total_tiles + span_tiles - 1) / span_tiles;
  long cur_end_b = -1, cur_begin_b = 0;  // byte range of the class whose row is in LDS
  for (long s = blockIdx.x; s < n_spans; s += gridDim.x) {
    const long t1 = min(total_tiles, (s + 1) * span_tiles);
    for (long t = s * span_tiles; t < t1; ++t) {
      const long tb = t * kTileB;
      if (tb >= cur_end_b || tb < cur_begin_b) {  // class of the tile's first byte is not the cached one
        const long row = tb / kRowB;
        const int c = row_class[row];
        __syncthreads();
        for (int w = threadIdx.x * 2; w < kW; w += 2 * T) *(u64x2*)(lrow + w) = *(const u64x2*)(tab + (size_t)c * kW + w);
        __syncthreads();
        cur_end_b = (long)class_end[c] * kRowB;
        cur_begin_b = (c ? (long)class_end[c - 1] : 0) * kRowB;
      }
      const long off = tb + threadIdx.x * 16;
      if (off >= total_b) continue;
      u64x2 v;
      if (off < cur_end_b) {
        const long row = off / kRowB;
        const int col = (int)(off - row * kRowB) >> 3;
        v = *(const u64x2*)(lrow + col);
      } else {  // tile straddles into the next class: fetch directly
        const long row = off / kRowB;
        const int col = (int)(off - row * kRowB) >> 3;
        v = *(const u64x2*)(tab + (size_t)row_class[row] * kW + col);
      }
      *(u64x2*)((char*)out + off) = v;
    }
  }
}

__global__ void verify(const u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ row_class, long n_rows,
                       unsigned long long* bad) {
  const long total = n_rows * kW;
  unsigned long long b = 0;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    long row = i / kW;
    int col = (int)(i - row * kW);
    if (out[i] != tab[(size_t)row_class[row] * kW + col]) ++b;
  }
  if (b) atomicAdd(bad, b);
}

int main() {
  const long rows = 1000000;
  const int C = 2061;
  const size_t bytes = (size_t)rows * kW * 8;
  u64* d; CK(hipMalloc(&d, bytes));
  // classes of random size, contiguous
  std::vector<int> h_rc(rows), h_end(C);
  {
    std::vector<double> wgt(C); double tot = 0; unsigned long long s = 88172645463325252ull;
    for (int c = 0; c < C; ++c) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; wgt[c] = 0.2 + (double)(s % 1000) / 500.0; tot += wgt[c]; }
    long r = 0;
    for (int c = 0; c < C; ++c) { long n = c == C - 1 ? rows - r : (long)(wgt[c] / tot * rows); if (n < 1) n = 1; for (long k = 0; k < n && r < rows; ++k) h_rc[r++] = c; h_end[c] = (int)r; }
    h_end[C - 1] = (int)rows;
  }
  std::vector<u64> h_tab((size_t)C * kW);
  { unsigned long long s = 1234567; for (auto& x : h_tab) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = s; } }
  u64* tab; int *rc, *ce; unsigned long long* bad;
  CK(hipMalloc(&tab, h_tab.size() * 8)); CK(hipMemcpy(tab, h_tab.data(), h_tab.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&rc, rows * 4)); CK(hipMemcpy(rc, h_rc.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&ce, C * 4)); CK(hipMemcpy(ce, h_end.data(), C * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&bad, 8));
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  auto run = [&](const char* name, bool check, auto launch) {
    if (check) {
      (void)hipMemset(d, 0, bytes); (void)hipMemset(bad, 0, 8);
      launch();
      verify<<<4096, 256>>>(d, tab, rc, rows, bad);
      unsigned long long hb = 0; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
      if (hb) printf("  !! %s: %llu wrong words\n", name, hb);
    }
    for (int i = 0; i < 2; ++i) launch();
    (void)hipEventRecord(ev0);
    for (int i = 0; i < 8; ++i) launch();
    (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
    float ms; (void)hipEventElapsedTime(&ms, ev0, ev1); ms /= 8;
    printf("%-56s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
    fflush(stdout);
  };
  char nm[128];
  for (int rep = 0; rep < 2; ++rep) {
    run("hipMemsetAsync", false, [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
    for (int g : {128, 256, 512}) { snprintf(nm, 128, "linear grid-stride fill, %d blocks", g); run(nm, false, [&] { fill_linear<<<g, 256>>>((u64x2*)d, bytes / 16, 7); }); }
  }
  for (int g : {128, 192, 256, 384, 512, 1024, 2048}) {
    snprintf(nm, 128, "regs  T=256 G=%d (contiguous row ranges)", g); run(nm, g == 256, [&] { expand_regs<256><<<g, 256>>>(d, tab, rc, ce, rows); });
  }
  for (int g : {128, 256, 512}) {
    snprintf(nm, 128, "regs  T=512 G=%d (contiguous row ranges)", g); run(nm, g == 256, [&] { expand_regs<512><<<g, 512>>>(d, tab, rc, ce, rows); });
  }
  for (int g : {128, 256}) {
    snprintf(nm, 128, "regs  T=128 G=%d (contiguous row ranges)", g); run(nm, g == 256, [&] { expand_regs<128><<<g, 128>>>(d, tab, rc, ce, rows); });
  }
  const long tiles256 = (long)(bytes / 4096) + 1;
  for (int g : {128, 256, 512}) {
    for (long S : {1L, 4L, 16L, 64L, 256L, 1024L, 0L}) {
      long span = S ? S : (tiles256 + g - 1) / g;
      snprintf(nm, 128, "lds   T=256 G=%d span=%ld tiles%s", g, span, S ? "" : " (one region per WG)");
      run(nm, g == 256, [&] { expand_lds<256><<<g, 256>>>(d, tab, rc, ce, rows, span); });
    }
  }
  const long tiles512 = (long)(bytes / 8192) + 1;
  for (int g : {128, 256}) {
    for (long S : {1L, 16L, 256L, 0L}) {
      long span = S ? S : (tiles512 + g - 1) / g;
      snprintf(nm, 128, "lds   T=512 G=%d span=%ld tiles%s", g, span, S ? "" : " (one region per WG)");
      run(nm, g == 256, [&] { expand_lds<512><<<g, 512>>>(d, tab, rc, ce, rows, span); });
    }
  }
  run("hipMemsetAsync (again)", false, [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
  return 0;
}
