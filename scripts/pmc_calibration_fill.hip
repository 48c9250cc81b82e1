// pmc_calibration_fill.hip — fills of KNOWN size in the bitmap shape of configs[2] (1M rows x 6272 B): scripts/pmc_passes.sh runs it under
// `rocprofv3 --pmc WRITE_SIZE` / `FETCH_SIZE` to calibrate the counters (ratio 1.000 on the linear fill) before they price the engine's
// kernels; it is also the write-ceiling probe of round 1 (linear fills and row-shaped writes at several launch shapes).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/pmc_calibration_fill.hip -o /tmp/fill_probe && /tmp/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

// (a) linear grid-stride fill, 16 B per thread per iteration
__global__ __launch_bounds__(256) void fill_linear(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) p[i] = val;
}
// (b) same with a launch-time block size and U independent 16 B stores per thread per iteration
template <int U>
__global__ void fill_linear_u(u64x2* p, size_t n16, u64 v) {
  u64x2 val = {v, v};
  const size_t step = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * step < n16; i += U * step) {
#pragma unroll
    for (int u = 0; u < U; ++u) p[i + u * step] = val;
  }
  for (; i < n16; i += step) p[i] = val;
}
// (c/d) row pattern: block handles `rows_per_block` rows; row r of block b = (b + r * row_step) (scatter) or b*rpb + r (contiguous)
template <bool NT>
__global__ __launch_bounds__(256) void fill_rows(u64* p, int stride_words, int rows_per_block, long row_step, long n_rows, u64 v) {
  u64x2 val = {v, v};
  for (int r = 0; r < rows_per_block; ++r) {
    long row = row_step ? ((long)blockIdx.x % row_step) + (long)r * row_step + ((long)blockIdx.x / row_step) * row_step * rows_per_block
                        : (long)blockIdx.x * rows_per_block + r;
    if (row >= n_rows) continue;
    u64* base = p + (size_t)row * stride_words;
    for (int w = threadIdx.x * 2; w < stride_words; w += 512) {
      if (NT) __builtin_nontemporal_store(val, (u64x2*)(base + w));
      else *(u64x2*)(base + w) = val;
    }
  }
}
// (g) persistent rows: grid of G blocks, block b writes rows b, b+G, b+2G, ... (window of G consecutive rows in flight)
__global__ __launch_bounds__(256) void fill_rows_persistent(u64* p, int stride_words, long n_rows, u64 v) {
  u64x2 val = {v, v};
  for (long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    u64* base = p + (size_t)row * stride_words;
    for (int w = threadIdx.x * 2; w < stride_words; w += 512) *(u64x2*)(base + w) = val;
  }
}
// (h) persistent, scattered: block b writes rows (b*K + j*step) style: emulate class members far apart
__global__ __launch_bounds__(256) void fill_rows_persistent_scatter(u64* p, int stride_words, long n_rows, long step, u64 v) {
  u64x2 val = {v, v};
  // bijection row = (i % step_count) * step + i / step_count over i = blockIdx.x + k*gridDim.x
  const long step_count = n_rows / step;
  for (long i = blockIdx.x; i < n_rows; i += gridDim.x) {
    long row = (i % step_count) * step + i / step_count;
    u64* base = p + (size_t)row * stride_words;
    for (int w = threadIdx.x * 2; w < stride_words; w += 512) *(u64x2*)(base + w) = val;
  }
}
// (i) channel-aligned gather-expand: G blocks walk 4 KiB tiles of the output in linear order (tile t → block t % G);
// every 16-B group is gathered from K source tables: src_k[cls_k(row)][col]. Emulates "bitmap row = AND of K plane rows".
template <int K, int U>
__global__ __launch_bounds__(256) void expand_tiles(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ cls,
                                                    int tab_rows, long n_rows) {
  const long total_tiles = n_rows * 6272 / 4096;
  for (long t0 = blockIdx.x; t0 < total_tiles; t0 += (long)gridDim.x * U) {
    u64x2 v[U];
    long offs[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long t = t0 + (long)u * gridDim.x;
      offs[u] = -1;
      if (t < total_tiles) {
        long off = t * 4096 + threadIdx.x * 16;
        long row = off / 6272;
        int col = (int)(off - row * 6272) / 8;  // word
        u64x2 acc = {~0ull, ~0ull};
#pragma unroll
        for (int k = 0; k < K; ++k) {
          int c = cls[row * K + k];
          acc &= *(const u64x2*)(tab + (size_t)c * 784 + col);
        }
        v[u] = acc;
        offs[u] = off;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
      if (offs[u] >= 0) *(u64x2*)((char*)out + offs[u]) = v[u];
  }
}
// (j) rows with a padded stride: write `used_words` of every `stride_words` row (row starts 4 KiB aligned when stride = 1024 words)
__global__ __launch_bounds__(256) void fill_rows_padded(u64* p, int stride_words, int used_words, int rows_per_block, long n_rows, u64 v) {
  u64x2 val = {v, v};
  for (int r = 0; r < rows_per_block; ++r) {
    long row = (long)blockIdx.x * rows_per_block + r;
    if (row >= n_rows) return;
    u64* base = p + (size_t)row * stride_words;
    for (int w = threadIdx.x * 2; w < used_words; w += 512) *(u64x2*)(base + w) = val;
  }
}
__global__ __launch_bounds__(256) void fill_rows_padded_persistent(u64* p, int stride_words, int used_words, long n_rows, u64 v) {
  u64x2 val = {v, v};
  for (long row = blockIdx.x; row < n_rows; row += gridDim.x) {
    u64* base = p + (size_t)row * stride_words;
    for (int w = threadIdx.x * 2; w < used_words; w += 512) *(u64x2*)(base + w) = val;
  }
}
// (k) tile-linear gather with a launch-time block size: block b walks 16-B groups (b*T + tid) + k*G*T, T = blockDim.x
template <int K, int U>
__global__ void expand_linear(u64* __restrict__ out, const u64* __restrict__ tab, const int* __restrict__ cls, long n_rows) {
  const long n16 = n_rows * 6272 / 16;
  const long step = (long)gridDim.x * blockDim.x;
  for (long i0 = (long)blockIdx.x * blockDim.x + threadIdx.x; i0 < n16; i0 += step * U) {
    u64x2 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long i = i0 + u * step;
      if (i < n16) {
        long off = i * 16;
        long row = off / 6272;
        int col = (int)(off - row * 6272) / 8;
        u64x2 acc = {~0ull, ~0ull};
#pragma unroll
        for (int k = 0; k < K; ++k) acc &= *(const u64x2*)(tab + (size_t)cls[row * K + k] * 784 + col);
        v[u] = acc;
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      long i = i0 + u * step;
      if (i < n16) *(u64x2*)((char*)out + i * 16) = v[u];
    }
  }
}
// (f) one block per row (no inner row loop): 1M blocks
__global__ __launch_bounds__(256) void fill_row_per_block(u64* p, int stride_words, u64 v) {
  u64x2 val = {v, v};
  u64* base = p + (size_t)blockIdx.x * stride_words;
  for (int w = threadIdx.x * 2; w < stride_words; w += 512) *(u64x2*)(base + w) = val;
}

int main() {
  const long rows = 1000000; const int stride = 784;
  const size_t bytes = (size_t)rows * stride * 8;
  u64* d; CK(hipMalloc(&d, bytes));
  hipEvent_t ev0, ev1; CK(hipEventCreate(&ev0)); CK(hipEventCreate(&ev1));
  auto run = [&](const char* name, auto launch) {
    for (int i = 0; i < 2; ++i) launch();
    (void)hipEventRecord(ev0);
    for (int i = 0; i < 10; ++i) launch();
    (void)hipEventRecord(ev1); (void)hipEventSynchronize(ev1);
    float ms; (void)hipEventElapsedTime(&ms, ev0, ev1); ms /= 10;
    printf("%-52s %.3f ms  %.0f GB/s\n", name, ms, bytes / ms / 1e6);
  };
  run("hipMemsetAsync", [&] { (void)hipMemsetAsync(d, 1, bytes, 0); });
  for (int g : {1024, 2048, 4096, 8192, 16384})
    { char nm[64]; snprintf(nm, 64, "linear grid-stride, %d blocks", g); run(nm, [&] { fill_linear<<<g, 256>>>((u64x2*)d, bytes / 16, 7); }); }
  for (int g : {128, 256, 512, 768})
    { char nm[64]; snprintf(nm, 64, "linear grid-stride, %d blocks", g); run(nm, [&] { fill_linear<<<g, 256>>>((u64x2*)d, bytes / 16, 7); }); }
  for (int g : {256, 512, 1024, 2048})
    { char nm[64]; snprintf(nm, 64, "rows persistent, %d blocks", g); run(nm, [&] { fill_rows_persistent<<<g, 256>>>(d, stride, rows, 7); }); }
  for (int g : {256, 512, 1024})
    { char nm[64]; snprintf(nm, 64, "rows persistent scatter(2000), %d blocks", g); run(nm, [&] { fill_rows_persistent_scatter<<<g, 256>>>(d, stride, rows, 2000, 7); }); }
  {
    // class table: 2061 rows (12.9 MB) for K=1; plane table: 375 rows (2.3 MB) for K=3
    std::vector<int> h1(rows), h3(rows * 3);
    for (long r = 0; r < rows; ++r) { h1[r] = (int)((r * 2654435761ul) % 2061); h3[3*r] = (int)((r * 2654435761ul) % 63); h3[3*r+1] = 63 + (int)((r * 40503ul) % 18); h3[3*r+2] = 81 + (int)((r * 2246822519ul) % 294); }
    u64* tab; int *c1, *c3;
    CK(hipMalloc(&tab, (size_t)2061 * 784 * 8)); CK(hipMemset(tab, 0xff, (size_t)2061 * 784 * 8));
    CK(hipMalloc(&c1, rows * 4)); CK(hipMalloc(&c3, rows * 12));
    CK(hipMemcpy(c1, h1.data(), rows * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(c3, h3.data(), rows * 12, hipMemcpyHostToDevice));
    for (int bs : {256, 512, 1024})
      for (int g : {256, 512}) {
        char nm[96];
        snprintf(nm, 96, "linear fill, %d blocks x %d threads", g, bs); run(nm, [&] { fill_linear_u<1><<<g, bs>>>((u64x2*)d, bytes / 16, 7); });
        snprintf(nm, 96, "expand linear K=1 U=4, %d blocks x %d threads", g, bs); run(nm, [&] { expand_linear<1, 4><<<g, bs>>>(d, tab, c1, rows); });
        snprintf(nm, 96, "expand linear K=1 U=8, %d blocks x %d threads", g, bs); run(nm, [&] { expand_linear<1, 8><<<g, bs>>>(d, tab, c1, rows); });
        snprintf(nm, 96, "expand linear K=3 U=4, %d blocks x %d threads", g, bs); run(nm, [&] { expand_linear<3, 4><<<g, bs>>>(d, tab, c3, rows); });
      }
    for (int g : {128, 256, 512})
    { char nm[80];
      snprintf(nm, 80, "expand tiles K=1 (class rows 12.9MB) U=2, %d blocks", g); run(nm, [&] { expand_tiles<1, 2><<<g, 256>>>(d, tab, c1, 2061, rows); });
      snprintf(nm, 80, "expand tiles K=1 (class rows 12.9MB) U=4, %d blocks", g); run(nm, [&] { expand_tiles<1, 4><<<g, 256>>>(d, tab, c1, 2061, rows); });
      snprintf(nm, 80, "expand tiles K=3 (planes 2.3MB) U=2, %d blocks", g); run(nm, [&] { expand_tiles<3, 2><<<g, 256>>>(d, tab, c3, 375, rows); });
      snprintf(nm, 80, "expand tiles K=3 (planes 2.3MB) U=4, %d blocks", g); run(nm, [&] { expand_tiles<3, 4><<<g, 256>>>(d, tab, c3, 375, rows); });
    }
  }
  {
    u64* d2; CK(hipMalloc(&d2, (size_t)rows * 1024 * 8));
    CK(hipMemset(d2, 0, (size_t)rows * 1024 * 8));
    for (int used : {784, 1024})
      for (int rpb : {64, 8}) { char nm[96]; snprintf(nm, 96, "rows stride 8KiB, %d words written, %d rows/block", used, rpb);
        run(nm, [&] { fill_rows_padded<<<(rows + rpb - 1) / rpb, 256>>>(d2, 1024, used, rpb, rows, 7); }); }
    for (int g : {256, 512, 2048}) { char nm[96]; snprintf(nm, 96, "rows stride 8KiB, 784 words, persistent %d blocks", g);
      run(nm, [&] { fill_rows_padded_persistent<<<g, 256>>>(d2, 1024, 784, rows, 7); }); }
    for (int g : {256, 512}) { char nm[96]; snprintf(nm, 96, "rows stride 8KiB, 1024 words, persistent %d blocks", g);
      run(nm, [&] { fill_rows_padded_persistent<<<g, 256>>>(d2, 1024, 1024, rows, 7); }); }
    (void)hipFree(d2);
  }
  for (int bs : {1024})
    for (int g : {256 * 1024 / bs * 2, 256 * 1024 / bs * 8})
      { char nm[80]; snprintf(nm, 80, "linear U=4, block %d, %d blocks", bs, g); run(nm, [&] { fill_linear_u<4><<<g, bs>>>((u64x2*)d, bytes / 16, 7); }); }
  for (int rpb : {64, 16, 4})
    { char nm[80]; snprintf(nm, 80, "rows contiguous, %d rows/block", rpb); run(nm, [&] { fill_rows<false><<<(rows + rpb - 1) / rpb, 256>>>(d, stride, rpb, 0, rows, 7); }); }
  run("rows contiguous 64/block, nontemporal", [&] { fill_rows<true><<<(rows + 63) / 64, 256>>>(d, stride, 64, 0, rows, 7); });
  run("rows scattered (step 2000), 64 rows/block", [&] { fill_rows<false><<<(rows + 63) / 64, 256>>>(d, stride, 64, 2000, rows, 7); });
  run("rows scattered (step 15625), 64 rows/block", [&] { fill_rows<false><<<(rows + 63) / 64, 256>>>(d, stride, 64, 15625, rows, 7); });
  run("one row per block (1M blocks)", [&] { fill_row_per_block<<<rows, 256>>>(d, stride, 7); });
  return 0;
}
