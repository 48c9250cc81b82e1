// Round 6 probe: what a ROW-SHAPED store stream can reach on this part, apart from any kernel logic. The zone-B writers (k_walk_rows,
// k_combine_wave, k_sweep_rows) all sit near 3.3 TB/s for single-member classes whatever their arithmetic; the band writer's linear
// pattern reaches 5.7. This probe writes the 6.27 GB bitmap of configs[2] (10^6 rows of 784 words) from registers with the
// wave-to-row assignments and store widths the writers could use, nothing else in the kernel.
//   hipcc --offload-arch=gfx950 -O3 scripts/r06_rowstore_probe.hip -o /tmp/rowstore_probe && /tmp/rowstore_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kStride = 784;  // words per row

// mode 0: a wave owns a contiguous range of rows and writes the row's SEGMENT [w0, w0 + nit*64) word per lane (dwordx2), row after row
// mode 1: the same rows handed out round robin: wave k writes rows k, k + W, k + 2W, ... (all waves inside one moving window)
// mode 2: round robin in batches of `batch` rows
template <int NIT>
__global__ __launch_bounds__(1024) void k_seg_x2(u64* out, int n_rows, int w0, int mode, int batch) {
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64, waves = blockDim.x / 64;
  const long gw = (long)blockIdx.x * waves + wave, W = (long)gridDim.x * waves;
  u64 v[NIT];
  for (int it = 0; it < NIT; ++it) v[it] = 0x0101010101010101ull * (unsigned)(lane + it);
  auto put = [&](long r) {
    u64* dst = out + r * kStride + w0 + lane;
#pragma unroll
    for (int it = 0; it < NIT; ++it) dst[it * 64] = v[it];
  };
  if (mode == 0) {
    const long per = (n_rows + W - 1) / W, r0 = gw * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    for (long r = r0; r < r1; ++r) put(r);
  } else {
    for (long b = gw * batch; b < n_rows; b += W * batch)
      for (long r = b; r < b + batch && r < n_rows; ++r) put(r);
  }
}
// dwordx4: a lane owns two adjacent words; a wave store covers 128 words (1 KiB)
template <int NIT>  // NIT pieces of 128 words
__global__ __launch_bounds__(1024) void k_seg_x4(u64* out, int n_rows, int w0, int mode, int batch) {
  const int lane = threadIdx.x % 64, wave = threadIdx.x / 64, waves = blockDim.x / 64;
  const long gw = (long)blockIdx.x * waves + wave, W = (long)gridDim.x * waves;
  u64x2 v[NIT];
  for (int it = 0; it < NIT; ++it) v[it] = u64x2{0x0101010101010101ull * (unsigned)(lane + it), 7};
  auto put = [&](long r) {
    u64* dst = out + r * kStride + w0 + 2 * lane;
#pragma unroll
    for (int it = 0; it < NIT; ++it) *(u64x2*)(dst + it * 128) = v[it];
  };
  if (mode == 0) {
    const long per = (n_rows + W - 1) / W, r0 = gw * per, r1 = r0 + per < n_rows ? r0 + per : n_rows;
    for (long r = r0; r < r1; ++r) put(r);
  } else {
    for (long b = gw * batch; b < n_rows; b += W * batch)
      for (long r = b; r < b + batch && r < n_rows; ++r) put(r);
  }
}

template <class K>
float run(K kernel, dim3 grid, dim3 block, u64* out, int n_rows, int w0, int mode, int batch) {
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a));
  CHECK(hipEventCreate(&b));
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, out, n_rows, w0, mode, batch);
  CHECK(hipEventRecord(a, 0));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL(kernel, grid, block, 0, 0, out, n_rows, w0, mode, batch);
  CHECK(hipEventRecord(b, 0));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  return ms / 5;
}

int main() {
  const int n_rows = 1000000;
  u64* out = nullptr;
  CHECK(hipMalloc(&out, (size_t)n_rows * kStride * 8 + 4096));
  CHECK(hipMemset(out, 0, (size_t)n_rows * kStride * 8));
  const char* mode_name[] = {"contiguous range per wave", "round robin rows", "round robin batches"};
  for (int waves : {4, 8, 16}) {
    for (int groups : {128, 256, 512, 1024}) {
      struct { int mode, batch; } forms[] = {{0, 1}, {1, 1}, {2, 4}, {2, 16}};
      for (auto f : forms) {
        // segment of 7 word groups (448 words) per lane word: the 7-group segment of the walk / sweep writers
        const float a = run(k_seg_x2<7>, dim3(groups), dim3(waves * 64), out, n_rows, 0, f.mode, f.batch);
        // 3 pieces of 128 words as dwordx4 (384 words)
        const float b = run(k_seg_x4<3>, dim3(groups), dim3(waves * 64), out, n_rows, 0, f.mode, f.batch);
        // whole rows: 6 pieces of 128 words as dwordx4 (768 of the 784 words)
        const float c = run(k_seg_x4<6>, dim3(groups), dim3(waves * 64), out, n_rows, 0, f.mode, f.batch);
        printf("waves/wg %2d groups %4d %-26s batch %2d | 7x64 words dwordx2: %.3f ms %.2f TB/s | 3x128 dwordx4: %.3f ms %.2f TB/s | whole row 6x128 dwordx4: %.3f ms %.2f TB/s\n",
               waves, groups, mode_name[f.mode], f.batch, a, n_rows * 448.0 * 8 / a / 1e9, b, n_rows * 384.0 * 8 / b / 1e9, c,
               n_rows * 768.0 * 8 / c / 1e9);
        fflush(stdout);
      }
    }
  }
  return 0;
}
