// fill_probe.hip — write-bandwidth probes on MI355X for the bitmap shape of configs[2] (1M rows x 6272 B).
// Build+run on the GPU box: hipcc --offload-arch=gfx950 -O3 scripts/fill_probe.hip -o /tmp/fill_probe && /tmp/fill_probe
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
