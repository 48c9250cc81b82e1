// fill_probe7.hip — fill_probe6 + THE ENGINE'S OWN k_expand_bands (included from kernels.hip.h) on the same tables: same-box A/B.
// VAR bits perturb the probe's step arithmetic (1: shorter store predicate, 2: single-compare class key, 4: byte-offset LDS
// address, 8: min() column wrap): every one of these semantically neutral changes costs 3-13 % on MI355X — the kernel sits
// at the store-issue limit of 4 waves per CU and its speed moves with instruction scheduling, not with instruction count.
// fill_probe6.hip — "band" layout probe: can the bitmap be written with EXACTLY the store pattern of the linear fill
// (256 workgroups, 4 KiB aligned tiles, a 1 MiB window that advances) while every row carries its class's pattern and
// the steady state issues no global loads?
//
// Layout idea: the physical rows of the bitmap are permuted (row_of_pod indirection) so that, inside a BAND of S
// consecutive windows, a pod class owns the rows whose START OFFSET INSIDE THEIR WINDOW, X(r) = (r * row_bytes) mod 2^20,
// falls into one interval. Workgroup b always writes window bytes [4096 b, 4096 b + 4096): for the S steps of a band it
// therefore needs the rows of at most two classes (class width in X >> 4 KiB), which it keeps in LDS; the rows of the next
// band are prefetched into registers mid-band and committed to the second LDS buffer at the band boundary. Per step: a few
// integer ops, one ds_read_b128, one global_store_dwordx4.
// Build+run: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe6.hip -o /tmp/fill_probe6 && /tmp/fill_probe6
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <climits>
#include "../yunikorn-k8shim_amd/csrc/engine/kernels.hip.h"
using ykk::u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
constexpr int kW = 784;
constexpr long kRowB = kW * 8;

__global__ __launch_bounds__(256) void fill_linear(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = val;
}

struct WgBand {  // what workgroup b needs in one band
  int cls_a, cls_b;  // class of the rows before / from the boundary (cls_b == cls_a: one class only)
  int xb, sb;        // first row of class B: (X, step) key; xb = 1 << 30 when there is no B
};

template <int G, int S, int VAR>
__global__ __launch_bounds__(256) void expand_bands(u64* __restrict__ out, const u64* __restrict__ class_rows, const WgBand* __restrict__ wb,
                                                    int n_bands, int n_steps, long total_b) {
  __shared__ u64 lds[2][2][kW];  // [buffer][A/B][word]
  constexpr long kWin = (long)G * 4096;
  const int b = blockIdx.x, tid = threadIdx.x;
  auto fetch = [&](const WgBand& w, u64x2 (&ra)[2], u64x2 (&rb)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int word = (i * 256 + tid) * 2;
      ra[i] = word < kW ? *(const u64x2*)(class_rows + (size_t)w.cls_a * kW + word) : u64x2{0, 0};
      rb[i] = word < kW ? *(const u64x2*)(class_rows + (size_t)w.cls_b * kW + word) : u64x2{0, 0};
    }
  };
  auto commit = [&](int buf, const u64x2 (&ra)[2], const u64x2 (&rb)[2]) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int word = (i * 256 + tid) * 2;
      if (word < kW) {
        *(u64x2*)(&lds[buf][0][word]) = ra[i];
        *(u64x2*)(&lds[buf][1][word]) = rb[i];
      }
    }
  };
  u64x2 na[2], nb[2];
  WgBand cur = wb[(size_t)0 * G + b];
  fetch(cur, na, nb);
  commit(0, na, nb);
  __syncthreads();
  int buf = 0;
  // Everything a step needs follows from the thread's byte offset p inside the window and the column of its 16 bytes in
  // their row: off = s * win + p, col advances by (win mod row) per step, the row started in this window iff col <= p, and
  // then its start offset in the window is x = p - col. No division in the loop.
  const int p = b * 4096 + tid * 16;
  constexpr int kDcol = (int)(kWin % kRowB);
  int col = (int)(((long)p) % kRowB);  // step 0
  char* wr = (char*)out + p;
  for (int band = 0; band < n_bands; ++band) {
    const int s0 = band * S, s1 = min(n_steps, s0 + S);
    WgBand nxt = cur;
    for (int s = s0; s < s1; s += 4) {
      if (s == s0 + S / 2 && band + 1 < n_bands) {  // prefetch the next band's rows: they land while this band finishes
        nxt = wb[(size_t)(band + 1) * G + b];
        fetch(nxt, na, nb);
      }
      u64x2 v[4];
      bool live[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int x = p - col;
        if (VAR & 1) live[u] = x >= 0;
        else live[u] = s + u < s1 && x >= 0 && (long)(s + u) * kWin + p < total_b;
        bool use_b;
        if (VAR & 2) use_b = (((unsigned)x << 8) | (unsigned)(s + u - s0)) >= (cur.xb == (1 << 30) ? 0xffffffffu : (((unsigned)cur.xb << 8) | (unsigned)(cur.sb - s0)));
        else use_b = x > cur.xb || (x == cur.xb && s + u >= cur.sb);
        if (VAR & 4) v[u] = *(const u64x2*)((const char*)&lds[buf][0][0] + (use_b ? (int)kRowB : 0) + col);
        else v[u] = *(const u64x2*)(&lds[buf][use_b ? 1 : 0][col >> 3]);
        if (VAR & 8) { const unsigned t = (unsigned)col + (unsigned)kDcol; col = (int)min(t, t - (unsigned)kRowB); }
        else { col += kDcol; col -= col >= (int)kRowB ? (int)kRowB : 0; }
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        if (live[u]) *(u64x2*)wr = v[u];
        wr += kWin;
      }
    }
    if (band + 1 < n_bands) {
      commit(buf ^ 1, na, nb);
      __syncthreads();
      buf ^= 1;
      cur = nxt;
    }
  }
}
// rows straddling a window boundary (one per window): rewritten whole
__global__ __launch_bounds__(256) void fixup_rows(u64* __restrict__ out, const u64* __restrict__ class_rows, const int* __restrict__ rows,
                                                  const int* __restrict__ row_class, int n) {
  if ((int)blockIdx.x >= n) return;
  const int r = rows[blockIdx.x];
  for (int w = threadIdx.x * 2; w < kW; w += 512) *(u64x2*)(out + (size_t)r * kW + w) = *(const u64x2*)(class_rows + (size_t)row_class[r] * kW + w);
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

template <int G, int S, int VAR>
int run_case(u64* d, size_t bytes, long rows, const std::vector<int>& sizes, unsigned long long* bad, hipEvent_t ev0, hipEvent_t ev1) {
  const int C = (int)sizes.size();
  constexpr long kWin = (long)G * 4096;
  const long total_b = rows * kRowB;
  const int n_steps = (int)((total_b + kWin - 1) / kWin);
  const int n_bands = (n_steps + S - 1) / S;
  // physical rows sorted by (band, X, step); classes take consecutive runs of that order
  struct Key { int band, x, s; int r; };
  std::vector<Key> keys(rows);
  for (long r = 0; r < rows; ++r) {
    long start = r * kRowB;
    int s = (int)(start / kWin);
    keys[r] = {s / S, (int)(start % kWin), s, (int)r};
  }
  std::sort(keys.begin(), keys.end(), [](const Key& a, const Key& b) { return a.band != b.band ? a.band < b.band : (a.x != b.x ? a.x < b.x : a.s < b.s); });
  std::vector<int> row_class(rows);
  struct First { int band, x, s; };
  std::vector<First> first(C + 1);
  {
    long i = 0;
    for (int c = 0; c < C; ++c) {
      first[c] = {keys[i].band, keys[i].x, keys[i].s};
      for (int k = 0; k < sizes[c] && i < rows; ++k, ++i) row_class[keys[i].r] = c;
    }
    first[C] = {1 << 30, 0, 0};
  }
  // a class that straddles a band boundary simply continues in the next band (sorted order restarts at X = 0 there)
  std::vector<WgBand> wb((size_t)n_bands * G);
  // per band: the classes present, in X order, with the key of their first row inside the band
  std::vector<std::vector<std::pair<std::pair<int, int>, int>>> per_band(n_bands);  // ((x, s), class)
  {
    int prev_band = -1, prev_cls = -1;
    for (long i = 0; i < rows; ++i) {
      const int c = row_class[keys[i].r];
      if (keys[i].band != prev_band || c != prev_cls) per_band[keys[i].band].push_back({{keys[i].x, keys[i].s}, c});
      prev_band = keys[i].band;
      prev_cls = c;
    }
  }
  int complex_wgs = 0;
  for (int band = 0; band < n_bands; ++band) {
    auto& pb = per_band[band];
    for (int b = 0; b < G; ++b) {
      const int x_lo = std::max(0, b * 4096 - (int)kRowB + 1), x_hi = (b + 1) * 4096;  // rows starting in [x_lo, x_hi) touch the tile
      // class containing key (x_lo, -inf): last entry with first key <= (x_lo, -inf)
      int ia = 0;
      for (int i = 0; i < (int)pb.size(); ++i)
        if (pb[i].first.first < x_lo) ia = i;
      WgBand w{pb[ia].second, pb[ia].second, 1 << 30, 0};
      if (ia + 1 < (int)pb.size() && pb[ia + 1].first.first < x_hi) {
        w.cls_b = pb[ia + 1].second;
        w.xb = pb[ia + 1].first.first;
        w.sb = pb[ia + 1].first.second;
        if (ia + 2 < (int)pb.size() && pb[ia + 2].first.first < x_hi) ++complex_wgs;
      }
      wb[(size_t)band * G + b] = w;
    }
  }
  std::vector<int> fix;
  for (long r = 0; r < rows; ++r) {
    long start = r * kRowB, end = start + kRowB - 1;
    if (start / kWin != end / kWin) fix.push_back((int)r);
  }
  std::vector<u64> h_tab((size_t)C * kW);
  { unsigned long long s = 1234567; for (auto& x : h_tab) { s = s * 6364136223846793005ull + 1442695040888963407ull; x = s; } }
  u64* tab; int *rc, *fx; WgBand* dwb;
  CK(hipMalloc(&tab, h_tab.size() * 8)); CK(hipMemcpy(tab, h_tab.data(), h_tab.size() * 8, hipMemcpyHostToDevice));
  CK(hipMalloc(&rc, rows * 4)); CK(hipMemcpy(rc, row_class.data(), rows * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&fx, (fix.size() + 1) * 4)); CK(hipMemcpy(fx, fix.data(), fix.size() * 4, hipMemcpyHostToDevice));
  CK(hipMalloc(&dwb, wb.size() * sizeof(WgBand))); CK(hipMemcpy(dwb, wb.data(), wb.size() * sizeof(WgBand), hipMemcpyHostToDevice));
  auto launch = [&] {
    expand_bands<G, S, VAR><<<G, 256>>>(d, tab, dwb, n_bands, n_steps, total_b);
    fixup_rows<<<(unsigned)fix.size(), 256>>>(d, tab, fx, rc, (int)fix.size());
  };
  (void)hipMemset(d, 0, bytes); (void)hipMemset(bad, 0, 8);
  launch();
  verify<<<4096, 256>>>(d, tab, rc, rows, bad);
  unsigned long long hb = 0; (void)hipMemcpy(&hb, bad, 8, hipMemcpyDeviceToHost);
  for (int i = 0; i < 2; ++i) launch();
  (void)hipEventRecord(ev0);
  for (int i = 0; i < 8; ++i) launch();
  (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
  float ms; (void)hipEventElapsedTime(&ms, ev0, ev1); ms /= 8;
  printf("bands VAR=%d G=%d S=%d: %d classes, %d bands, %zu fix-up rows, %d complex wg-bands, wrong words %llu   %.3f ms  %.0f GB/s\n", VAR, G, S, C, n_bands,
         fix.size(), complex_wgs, hb, ms, bytes / ms / 1e6);
  fflush(stdout);
  if (G == 256 && (VAR == 0)) {
    std::vector<ykk::BandEntry> et(wb.size());
    for (int band = 0; band < n_bands; ++band)
      for (int b = 0; b < G; ++b) {
        const WgBand& w = wb[(size_t)band * G + b];
        ykk::BandEntry be{};
        be.slot[0] = w.cls_a;
        for (int i = 1; i < ykk::kBandClasses; ++i) be.slot[i] = w.cls_b;
        for (int i = 0; i < ykk::kBandClasses - 1; ++i) { be.xb[i] = INT_MAX; be.sb[i] = 0; }
        be.xb[0] = w.cls_b == w.cls_a ? INT_MAX : w.xb; be.sb[0] = w.sb;
        be.first_step = band * S; be.steps = std::min(S, (n_steps - band * S + 3) / 4 * 4);
        be.n = w.cls_b == w.cls_a ? 1 : 2;
        et[(size_t)band * G + b] = be;
      }
    ykk::BandEntry* det; CK(hipMalloc(&det, et.size() * sizeof(ykk::BandEntry)));
    CK(hipMemcpy(det, et.data(), et.size() * sizeof(ykk::BandEntry), hipMemcpyHostToDevice));
    auto run_ws = [&](auto kern, int sets) -> int {
      const size_t lds = (size_t)2 * ykk::kBandClasses * kW * 8;
      CK(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
      auto launch2 = [&] {
        hipLaunchKernelGGL(kern, dim3(256), dim3(sets), lds, 0, d, tab, det, n_bands, kW);
        fixup_rows<<<(unsigned)fix.size(), 256>>>(d, tab, fx, rc, (int)fix.size());
      };
      (void)hipMemset(d, 0, bytes); (void)hipMemset(bad, 0, 8);
      launch2();
      verify<<<4096, 256>>>(d, tab, rc, rows, bad);
      unsigned long long hb2 = 0; (void)hipMemcpy(&hb2, bad, 8, hipMemcpyDeviceToHost);
      for (int i = 0; i < 2; ++i) launch2();
      (void)hipEventRecord(ev0);
      for (int i = 0; i < 8; ++i) launch2();
      (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
      float ms2; (void)hipEventElapsedTime(&ms2, ev0, ev1); ms2 /= 8;
      printf("  ENGINE k_expand_bands (%d threads), same tables S=%d: wrong words %llu   %.3f ms  %.0f GB/s\n", sets, S, hb2, ms2, bytes / ms2 / 1e6);
      return 0;
    };
    if (run_ws(ykk::k_expand_bands, ykk::kBandBlock)) return 1;
    (void)hipFree(det);
  }
  (void)hipFree(tab); (void)hipFree(rc); (void)hipFree(fx); (void)hipFree(dwb);
  return 0;
}

int main() {
  const long rows = 1000000;
  const size_t bytes = (size_t)rows * kW * 8;
  u64* d; CK(hipMalloc(&d, bytes + (8 << 20)));
  unsigned long long* bad; CK(hipMalloc(&bad, 8));
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  auto timeit = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    (void)hipEventRecord(ev0);
    for (int i = 0; i < 8; ++i) launch();
    (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
    float ms; (void)hipEventElapsedTime(&ms, ev0, ev1); ms /= 8;
    printf("%-40s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  timeit("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
  timeit("linear grid-stride fill 256x256", [&] { fill_linear<<<256, 256>>>((u64x2*)d, bytes / 16, 7); });
  timeit("linear grid-stride fill 128x256", [&] { fill_linear<<<128, 256>>>((u64x2*)d, bytes / 16, 7); });
  for (int avg : {485}) {
    std::vector<int> sizes;
    long left = rows; unsigned long long s = 88172645463325252ull + avg;
    while (left > 0) {
      s ^= s << 13; s ^= s >> 7; s ^= s << 17;
      int n = avg / 2 + (int)(s % (unsigned)avg);
      if (n > left) n = (int)left;
      sizes.push_back(n);
      left -= n;
    }
    printf("---- classes of %d..%d rows\n", avg / 2, avg / 2 + avg - 1);
    if (run_case<256, 128, 0>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
    if (run_case<256, 128, 1>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
    if (run_case<256, 128, 2>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
    if (run_case<256, 128, 4>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
    if (run_case<256, 128, 8>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
    if (run_case<256, 128, 15>(d, bytes, rows, sizes, bad, ev0, ev1)) return 1;
  }
  timeit("hipMemsetAsync (again)", [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
  return 0;
}
