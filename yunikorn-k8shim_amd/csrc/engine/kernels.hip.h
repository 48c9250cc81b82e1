// kernels.hip.h — hand-written HIP kernels of the ykpred engine, gfx950 (MI355X / CDNA4) only.
//
// The hot path replaces the reference's per-(pod,node) Predicates() loop
// (/root/reference/pkg/plugin/predicates/predicate_manager.go:206-283) with three device stages:
//
//   1. SIGNATURE PLANES (k_plane_*): pods are grouped by what each plugin can see of them. Every distinct
//      request vector, toleration mask and node-selector DNF ("signature") is evaluated ONCE against all N
//      nodes: lane = node, one wave = 64 nodes, `__ballot` packs the 64 verdicts into one u64 word of the
//      signature's N-bit plane. Node columns are read coalesced from the structure-of-arrays tables and held
//      in registers while the wave walks the signatures (signature data is wave-uniform → scalar loads).
//   2. COMBINE (k_combine): a pod class = tuple of signatures (+ pinned node). Its bitmap row is the AND of
//      its planes; the block holds that row in registers and streams it to the bitmap row of every member
//      pod — sequential 6 KB row writes, no per-pair arithmetic left. `__popcll` of the row gives the class's
//      feasible-node count. This kernel writes the P×N/8-byte bitmap and is HBM-write bound.
//   3. DECIDE (k_decide): planes are also produced in bin-pack rank order, so the best feasible node of a
//      class is the first set bit of the AND of its rank-ordered planes (early exit).
//
// k_direct is the per-pair formulation (lane = node, wave walks pods) kept as the on-device cross-check and
// as the ablation baseline; k_query answers single Predicates() calls including the first failing plugin.
//
// No MFMA anywhere: the path is integer compare / bitmask / popcount, not a contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <type_traits>

namespace ykk {

typedef unsigned long long u64;
typedef long long i64;

// compile-time ceilings of the register-resident node columns (checked in ykpred_create)
constexpr int kMaxR = 8;   // resource dimensions
constexpr int kMaxKT = 4;  // taint dictionary words (256 taints)
constexpr int kMaxW = 8;   // requirement dictionary words held in REGISTERS by the per-pair kernels (512 requirements) ...
constexpr int kMaxWTotal = 32;  // ... words beyond them are read from the node table when a term references them (2048)
constexpr int kMaxKD = 8;  // topology keys used by spread / inter-pod-affinity constraints
constexpr int kMaxKP = 4;  // host-port dictionary words (256 distinct requested host ports)

constexpr int kWave = 64;
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kSigsPerBlock = 16;   // signatures walked by one plane block (lane i keeps the ballot word of signature i):
                                    // the walk is latency-bound, so short walks in many blocks beat long ones ...
constexpr int kSigsPerBlockMany = 64;  // ... until there are so many signatures that per-block overhead dominates
constexpr int kManySigs = 4096;
__host__ __device__ inline int sigs_per_block(int D) { return D > kManySigs ? kSigsPerBlockMany : kSigsPerBlock; }
constexpr int kChunkMembers = 64;   // member pods per combine chunk (member ids live in the lanes of a wave)
constexpr int kCombineUnroll = 4;   // row words per thread held in registers by k_combine

// plugin bits (mirror include/ykpred.h)
constexpr unsigned kSpreadHonorAffinity = 1u << 0, kSpreadHonorTaints = 1u << 1;
constexpr unsigned kPlugUnsched = 1u << 0, kPlugNodeName = 1u << 1, kPlugTaint = 1u << 2, kPlugAffinity = 1u << 3,
                   kPlugPorts = 1u << 4, kPlugFit = 1u << 5, kPlugSpread = 1u << 6, kPlugInterPod = 1u << 7;
constexpr int kKindSpread = 0, kKindPodAffinity = 1, kKindPodAntiAffinity = 2, kKindExistingAnti = 3;
constexpr unsigned kSpecToleratesUnsched = 1u << 0, kSpecAffSkip = 1u << 1, kSpecPreReject = 1u << 2, kSpecPreNames = 1u << 3,
                   kSpecUnsupported = 1u << 4;
constexpr unsigned kNodeUnschedulable = 1u << 0;

struct NodeTable {
  int n;               // nodes
  int R, KT, W;        // dims
  const i64* alloc;    // [R][n]
  const i64* req;      // [R][n]
  const int* allowed;  // [n]
  const int* count;    // [n]
  const unsigned* flags;  // [n]
  const u64* taints;   // [KT][n]
  const u64* labels;   // [W][n]
  int KD, KS;          // PodTopologySpread: topology keys, selector classes
  const int* domain;   // [KD][n] id of the node's value for topology key k, -1 = label missing
  const int* selcount; // [KS][n] pods on the node matching selector class s
  int KP;              // NodePorts: host-port dictionary words
  const u64* ports;    // [KP][n] bit k: a pod on the node conflicts with dictionary host port k
};

// State that a running kernel changes and re-reads (ykpred_allocate_round: the scratch copies of Requested / pod counts / topology
// histograms / host-port words). The round is ONE workgroup, so WORKGROUP scope is all the coherence it needs: its waves share
// the compute unit's L1, which sees the unit's own stores — plain loads and stores, ordered by the loop's barriers. (Agent scope
// was what the loop cost most: on this part every such load bypasses the L2 of its XCD and every release fence writes it back.)
template <class T>
__device__ __forceinline__ T ld_live(const T* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
template <class T>
__device__ __forceinline__ void st_live(T* p, T v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void or_live(u64* p, u64 v) { (void)__hip_atomic_fetch_or(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
template <bool LIVE, class T>
__device__ __forceinline__ T ld_maybe_live(const T* p) {
  if constexpr (LIVE) return ld_live(p);
  else return *p;
}

// ---------------------------------------------------------------------------------------------------
// bin-pack score + rank
// ---------------------------------------------------------------------------------------------------
// score(n) = 1 - (Σ_{r∈{cpu,mem}, total_r>0} (1 - avail_r/total_r)) / #r   in float64, the exact operation
// order written in DESIGN.md §"bin-pack score contract" (no multiplies ⇒ nothing for FMA contraction to fuse).
__device__ __forceinline__ double node_score_of(const i64 (&total)[2], const i64 (&used)[2]) {
  double sum = 0.0, wsum = 0.0;
#pragma unroll
  for (int r = 0; r < 2; ++r) {
    if (total[r] <= 0) continue;
    double avail = (double)(total[r] - used[r]);
    double share = 1.0 - avail / (double)total[r];
    sum = sum + share;
    wsum = wsum + 1.0;
  }
  if (wsum == 0.0) return 1.0;
  return 1.0 - sum / wsum;
}
__device__ __forceinline__ double node_score(const NodeTable& t, int n) {
  const i64 total[2] = {t.alloc[n], t.alloc[(size_t)t.n + n]}, used[2] = {t.req[n], t.req[(size_t)t.n + n]};
  return node_score_of(total, used);
}
__device__ __forceinline__ u64 sortable_key(double s) {
  u64 b = (u64)__double_as_longlong(s);
  return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__global__ __launch_bounds__(kBlock) void k_score(NodeTable t, double* __restrict__ score, u64* __restrict__ key) {
  int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= t.n) return;
  double s = node_score(t, n);
  score[n] = s;
  key[n] = sortable_key(s);
}
// Bin-pack ORDER: rank[n] = #{m : (key_m, tie_m) < (key_n, tie_n)}, perm[rank[n]] = n — an exact sort by (score, tie) with
// tie = the node's position in NodeID order (ykpred_nodes_t.name_rank) or its index.
// Nodes are first bucketed by a monotone function of the score (kRankBuckets equal-width bins of [0,1], clamped), so
// the bucket sequence already agrees with the key order; the exact rank is the bucket's start plus the rank among
// the ~N/1024 members of the same bucket. Four tiny kernels instead of N² compares (50k nodes: 2.5e9 → ~3e6).
// Degenerate inputs (all scores in one bucket) fall back to O(N²) work inside that bucket but stay exact.
constexpr int kRankBuckets = 1024;
__device__ __forceinline__ int rank_bucket(double score) {
  double x = score * (double)kRankBuckets;
  int b = (x >= (double)(kRankBuckets - 1)) ? (kRankBuckets - 1) : ((x > 0.0) ? (int)x : 0);  // NaN-free: score is finite
  return b;
}
__global__ __launch_bounds__(kBlock) void k_rank_hist(int n_nodes, const double* __restrict__ score, int* __restrict__ hist) {
  int n = blockIdx.x * kBlock + threadIdx.x;
  if (n < n_nodes) atomicAdd(&hist[rank_bucket(score[n])], 1);
}
// one block of kRankBuckets threads: exclusive scan of the histogram → bucket_off[kRankBuckets + 1]; cursor = copy
__global__ __launch_bounds__(kRankBuckets) void k_rank_scan(const int* __restrict__ hist, int* __restrict__ bucket_off,
                                                            int* __restrict__ cursor) {
  __shared__ int tmp[kRankBuckets];
  int t = threadIdx.x;
  int v = hist[t];
  tmp[t] = v;
  __syncthreads();
  for (int off = 1; off < kRankBuckets; off <<= 1) {
    int add = t >= off ? tmp[t - off] : 0;
    __syncthreads();
    tmp[t] += add;
    __syncthreads();
  }
  int excl = tmp[t] - v;
  bucket_off[t] = excl;
  cursor[t] = excl;
  if (t == kRankBuckets - 1) bucket_off[kRankBuckets] = tmp[t];
}
__global__ __launch_bounds__(kBlock) void k_rank_fill(int n_nodes, const double* __restrict__ score, const u64* __restrict__ key,
                                                      const int* __restrict__ name_rank, int* __restrict__ cursor, int* __restrict__ members,
                                                      int* __restrict__ member_tie, u64* __restrict__ member_key) {
  int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= n_nodes) return;
  int pos = atomicAdd(&cursor[rank_bucket(score[n])], 1);
  members[pos] = n;
  member_tie[pos] = name_rank ? name_rank[n] : n;  // tie-break between equal scores: NodeID order (or node index)
  member_key[pos] = key[n];  // keys travel with the ids: the final pass streams them instead of chasing key[members[i]]
}
// One block per bucket: the bucket's (key, node) pairs are staged through LDS in tiles and every member counts the
// entries that sort before it. LDS keeps this independent of global-memory latency (the kernel runs beside the
// bandwidth-saturating k_combine) and of bucket skew (e.g. all idle nodes share score 1.0 and one bucket).
constexpr int kRankTile = 1024;
__global__ __launch_bounds__(kBlock) void k_rank_final(const int* __restrict__ bucket_off, const int* __restrict__ members,
                                                       const int* __restrict__ member_tie, const u64* __restrict__ member_key,
                                                       int* __restrict__ rank, int* __restrict__ perm) {
  __shared__ u64 t_key[kRankTile];
  __shared__ int t_idx[kRankTile];
  const int lo = bucket_off[blockIdx.x], hi = bucket_off[blockIdx.x + 1];
  for (int base = lo; base < hi; base += kBlock) {  // this pass ranks members base + tid
    const int i = base + threadIdx.x;
    const bool live = i < hi;
    const u64 mine = live ? member_key[i] : 0ull;
    const int me = live ? member_tie[i] : 0, me_node = live ? members[i] : 0;
    int r = lo;
    for (int tb = lo; tb < hi; tb += kRankTile) {
      __syncthreads();
      for (int j = threadIdx.x; j < kRankTile && tb + j < hi; j += kBlock) {
        t_key[j] = member_key[tb + j];
        t_idx[j] = member_tie[tb + j];
      }
      __syncthreads();
      const int lim = min(kRankTile, hi - tb);
      if (live)
        for (int j = 0; j < lim; ++j) r += (t_key[j] < mine) || (t_key[j] == mine && t_idx[j] < me);
    }
    if (live) {
      rank[me_node] = r;
      perm[r] = me_node;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// signature planes
// ---------------------------------------------------------------------------------------------------
struct PlaneOut {
  u64* canon;   // [D][stride]
  u64* ranked;  // [D][stride] (node axis permuted by bin-pack rank), may be null
  int stride;   // words per plane row
  int D;        // signatures
  int* first;   // rank-ordered planes only (may be null): first[d] = index of the first non-zero word of row d (atomicMin; preset
                // to kNoWord): k_decide starts its scan there and gives up at once on an empty row
};
constexpr int kNoWord = 0x7f7f7f7f;

// blockIdx.x: chunk of kSigsPerBlock signatures (the unbounded axis: up to 2^31 blocks); blockIdx.y: group of 4 node
// words (≤ 65 535 groups = 16.7 M nodes). `perm` != null selects the
// rank-ordered plane (position i holds node perm[i]); the canonical and the ranked planes are separate launches so
// that the ranked ones can run on the decision stream. Returns the node of this lane (-1 = past the end).
__device__ __forceinline__ int plane_node(int n_nodes, const int* __restrict__ perm, int* word) {
  int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  int w = blockIdx.y * kWavesPerBlock + wave;
  *word = w;
  int pos = w * kWave + lane;
  if (pos >= n_nodes) return -1;
  return perm ? perm[pos] : pos;
}
// lane i holds the ballot word of signature d0 + i, for i < nsig
__device__ __forceinline__ void plane_store(const PlaneOut& o, bool ranked, int word, int d0, int nsig, u64 keep) {
  int lane = threadIdx.x % kWave;
  int d = d0 + lane;
  if (word < o.stride && lane < nsig && d < o.D) {
    u64* base = ranked ? o.ranked : o.canon;
    base[(size_t)d * o.stride + word] = keep;
    if (ranked && o.first && keep) atomicMin(&o.first[d], word);
  }
}

// NodeResourcesFit.Filter: fail ⇔ count+1 > allowed ∨ ∃r: req_r > 0 ∧ req_r > alloc_r − requested_r.
// The Filter is a conjunction over resource dimensions, and every conjunct depends on ONE number of the pod — so planes are
// kept per (dimension, distinct request value), not per request vector: plane(r, v) = {n : alloc_r − requested_r ≥ v}, and
// row 0 of the family is the pod-independent part (a pod slot is free; the PreFilter state exists). A request vector is
// the AND of row 0 and one row per dimension it asks for (res_rows table, applied in class_rows). The number of planes is
// Σ_r #distinct values — at most, and usually far below, #distinct vectors × R: 10^6 asks with distinct cpu requests cost
// 10^6 ONE-compare rows instead of 10^6 R-compare rows, and a realistic population (dozens of cpu values × dozens of
// memory values) costs dozens of rows instead of thousands.
// Rows are numbered in first-use order (stable when spec tables only append); `order` lists them grouped by dimension and
// a chunk (dim, begin, len <= kDimRowsPerBlock) never mixes dimensions, so a block reads a single column of the node table.
constexpr int kDimRowsPerBlock = 64;
struct DimPlanes {
  const i64* val;       // [rows] request value of the row (row 0: unused)
  const int* order;     // [rows] row ids grouped by dimension (row 0 first, as the pseudo-dimension -1)
  const int* chunk_dim;    // [chunks] dimension of the chunk, -1 = the base row
  const int* chunk_begin;  // [chunks] first entry of `order`
  const int* chunk_len;    // [chunks] 1..kDimRowsPerBlock
  int n_chunks;
};
__device__ __forceinline__ i64 readlane_i64(i64 x, int lane) {
  const unsigned lo = __builtin_amdgcn_readlane((unsigned)(u64)x, lane), hi = __builtin_amdgcn_readlane((unsigned)((u64)x >> 32), lane);
  return (i64)(((u64)hi << 32) | lo);
}
__device__ __forceinline__ void plane_dim(const NodeTable& t, const int* __restrict__ perm, const DimPlanes& dp, const PlaneOut& o,
                                          int fit_error, int n_words) {
  int word;
  const int n = plane_node(t.n, perm, &word);
  if (word >= n_words) return;
  const int chunk = blockIdx.x;
  const int dim = dp.chunk_dim[chunk], begin = dp.chunk_begin[chunk], len = dp.chunk_len[chunk];
  const int lane = threadIdx.x % kWave;
  // lane i fetches (row id, value) of the chunk's i-th row once; the walk below reads them with v_readlane — no memory
  // latency inside the loop
  const int my_row = lane < len ? dp.order[begin + lane] : 0;
  const i64 my_val = (lane < len && dim >= 0) ? dp.val[my_row] : 0;
  unsigned keep_lo = 0, keep_hi = 0;
  if (dim < 0) {
    const bool ok = n >= 0 && !fit_error && (i64)t.count[n] + 1 <= (i64)t.allowed[n];
    const u64 b = __ballot(ok);
    keep_lo = (unsigned)b;
    keep_hi = (unsigned)(b >> 32);
  } else if (!fit_error) {
    // invalid positions (past N) never fit: every row value is > 0 > INT64_MIN
    const i64 fr = n >= 0 ? t.alloc[(size_t)dim * t.n + n] - t.req[(size_t)dim * t.n + n] : (i64)0x8000000000000000ull;
    for (int i = 0; i < len; ++i) {
      const i64 v = readlane_i64(my_val, i);
      const u64 b = __ballot(fr >= v);
      // lane i keeps the word of the chunk's i-th row: two v_writelane instead of a compare + two selects per row
      const unsigned blo = __builtin_amdgcn_readfirstlane((unsigned)b), bhi = __builtin_amdgcn_readfirstlane((unsigned)(b >> 32));
      // (VOP3 takes one SGPR operand besides M0: the lane select travels in M0, which is saved and restored)
      unsigned saved_m0;
      asm volatile("s_mov_b32 %2, m0\n\ts_mov_b32 m0, %4\n\tv_writelane_b32 %0, %3, m0\n\tv_writelane_b32 %1, %5, m0\n\ts_mov_b32 m0, %2"
                   : "+v"(keep_lo), "+v"(keep_hi), "=&s"(saved_m0)
                   : "s"(blo), "s"(i), "s"(bhi));
    }
  }
  if (lane < len) {
    u64* base = perm ? o.ranked : o.canon;
    base[(size_t)my_row * o.stride + word] = ((u64)keep_hi << 32) | keep_lo;
    if (perm && o.first && (keep_lo | keep_hi)) atomicMin(&o.first[my_row], word);
  }
}

// Dimensions with MANY distinct request values (10^3 … 10^6: every ask its own cpu request) use the monotone structure
// instead of one compare per (value, node): plane(r, v) shrinks as v grows, and inside one 64-node word it changes at most
// 64 times. k_dim_sort orders the 64 free values of every word once per evaluation (sfree ascending, pmask[j] = the nodes of
// the entries j..63); k_dim_walk then gives a THREAD one word: it walks the dimension's rows in ascending value order,
// advances its position in the sorted list while the free value there is below the row's value, and stores pmask[position]
// — lanes are consecutive words, so a wave writes 512 contiguous bytes of a row. Cost: the plane bytes written once
// (HBM-write bound) instead of rows × N compares (ALU bound).
constexpr int kWalkRows = 512;  // rows per walk chunk (one binary search per thread and chunk)
constexpr int kRankBits = 7;    // rank planes of a word: bit k of r' = valid ? position in the word's ascending free list + 1 : 0 (0..64)
struct DimWalk {
  const i64* val;          // [rows]
  const int* order;        // [rows] row ids grouped by dimension; inside a walked dimension ascending by value
  const int* big_dim;      // [n_big] walked dimensions
  const int* chunk_big;    // [chunks] index into big_dim
  const int* chunk_begin;  // [chunks] first entry of `order`
  const int* chunk_len;    // [chunks] 1..kWalkRows
  i64* sfree;              // [n_big][n_words][64]
  u64* pmask;              // [n_big][n_words][65]
  int n_big, n_chunks, n_words;
  u64* rbits;              // [n_big][kRankBits][n_words] the same sets bit-sliced (k_walk_rows): pmask[j] = { node : r' > j }; null = not kept
  // cursor lists of k_sweep_rows (null = not kept): ent[(big * 65 + j) * n_words + word] = (g << 6) | node position, for the j-th
  // smallest free value f of the word, g = how many of the dimension's sorted request values are <= f; row 64 = 0xffffffff
  unsigned* ent;
  const i64* sorted;       // the walked dimensions' request values, ascending, dimension after dimension
  const int* sorted_off;   // [n_big + 1] into `sorted`
  unsigned* glin;          // rank order, k_run_decide (null = not kept): glin[(big * n_words + word) * 64 + lane] = g of that node's free value;
                           // behind them [n_big][n_words]: the largest g of every word
};
// blockIdx.x: walked dimension, blockIdx.y: group of 4 words; wave = word, lane = node position
__global__ __launch_bounds__(kBlock) void k_dim_sort(NodeTable t, const int* __restrict__ perm, DimWalk a) {
  int word;
  const int n = plane_node(t.n, perm, &word);
  if (word >= a.n_words) return;
  const int lane = threadIdx.x % kWave;
  const int dim = a.big_dim[blockIdx.x];
  const bool valid = n >= 0;
  const i64 fr = valid ? t.alloc[(size_t)dim * t.n + n] - t.req[(size_t)dim * t.n + n] : (i64)0x8000000000000000ull;
  int rank = 0;  // position in ascending (free, lane) order
  for (int j = 0; j < kWave; ++j) {
    const i64 o = readlane_i64(fr, j);
    rank += (o < fr || (o == fr && j < lane)) ? 1 : 0;
  }
  const size_t cell = (size_t)blockIdx.x * a.n_words + word;
  a.sfree[cell * 64 + rank] = fr;
  u64 keep = 0;
  for (int j = 0; j < kWave; ++j) {
    const u64 b = __ballot(valid && rank >= j);
    if (j == lane) keep = b;
  }
  a.pmask[cell * 65 + lane] = keep;
  if (lane == 0) a.pmask[cell * 65 + 64] = 0;
  if (a.rbits) {
    const int r1 = valid ? rank + 1 : 0;
    u64 plane = 0;
#pragma unroll
    for (int k = 0; k < kRankBits; ++k) {
      const u64 b = __ballot((r1 >> k) & 1);
      if (k == lane) plane = b;
    }
    if (lane < kRankBits) a.rbits[((size_t)blockIdx.x * kRankBits + lane) * a.n_words + word] = plane;
  }
  if (a.ent || a.glin) {
    const i64* sv = a.sorted + a.sorted_off[blockIdx.x];
    int lo = 0, hi = a.sorted_off[blockIdx.x + 1] - a.sorted_off[blockIdx.x];  // upper bound: values [0, lo) are <= fr
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sv[mid] <= fr) lo = mid + 1; else hi = mid;
    }
    if (a.ent) {
      unsigned* col = a.ent + (size_t)blockIdx.x * 65 * a.n_words + word;
      col[(size_t)rank * a.n_words] = ((unsigned)lo << 6) | (unsigned)lane;
      if (lane == 0) col[(size_t)64 * a.n_words] = 0xffffffffu;
    }
    if (a.glin) {
      const unsigned g = valid ? (unsigned)lo : 0u;
      a.glin[((size_t)blockIdx.x * a.n_words + word) * 64 + lane] = g;
      unsigned mx = g;  // the word's largest g, behind the per-node values: k_run_decide looks at a word's nodes only where that can matter
#pragma unroll
      for (int off = 32; off > 0; off >>= 1) mx = max(mx, (unsigned)__shfl_xor((int)mx, off, kWave));
      if (lane == 0) a.glin[(size_t)a.n_big * a.n_words * 64 + (size_t)blockIdx.x * a.n_words + word] = mx;
    }
  }
}
// blockIdx.x: walk chunk, blockIdx.y: block of 256 * kWalkWords words; thread = kWalkWords ADJACENT words.
// The rows it produces are INDEX rows: one BYTE per 64-node word — the position `ptr` in the word's sorted free list — instead of
// the 8-byte plane word pmask[word][ptr] itself. A walked dimension has up to 10^6 rows; as u64 planes they are 6.5 GB written
// here and read again by the combine / decide kernels (half of that population's traffic), as index rows 0.8 GB. Consumers
// decode with one lookup into the 65-entry mask table of the word (Planes::pmask; class_word); k_walk_rows
// decodes with the rank planes of k_dim_sort instead.
// Four words per thread, one dword store per row: with a byte per thread the kernel sat on its 13 M byte-store instructions per
// pass (64 bytes per wave store; SQ counters: 73 % of the wave cycles waiting, 0.55–0.9 ms for 0.78 GB). EIGHT words per thread
// (8-byte stores, half the threads) are slower: 0.35 -> 0.52 ms, the rank-ordered walk 0.45 -> 0.71 ms (round 4, session 21).
constexpr int kWalkWords = 4;
constexpr int kWalkBatch = 32;  // rows whose positions are staged in LDS before they are stored (32 KB per workgroup)
__global__ __launch_bounds__(kBlock) void k_dim_walk(DimWalk a, unsigned char* __restrict__ out, int stride) {
  const int chunk = blockIdx.x;
  const int big = a.chunk_big[chunk], begin = a.chunk_begin[chunk], len = a.chunk_len[chunk];
  const int lane = threadIdx.x % kWave;
  const int w0 = (blockIdx.y * kBlock + threadIdx.x) * kWalkWords;
  if (w0 - lane * kWalkWords >= a.n_words) return;  // whole wave beyond the row
  const bool live = w0 < a.n_words;                 // (stride is a multiple of 64: the dword of a live thread lies inside the row)
  const i64 kMax = 0x7fffffffffffffffll;
  const i64* sf[kWalkWords];
  int ptr[kWalkWords];
  i64 next[kWalkWords];
  const i64 v0 = a.val[a.order[begin]];
#pragma unroll
  for (int j = 0; j < kWalkWords; ++j) {
    const int w = min(w0 + j, a.n_words - 1);  // every lane takes part in the v_readlane exchanges; bytes past the row are never read
    sf[j] = a.sfree + ((size_t)big * a.n_words + w) * 64;
    int lo = 0, hi = 64;  // lower bound: entries [0, ptr) are below the chunk's first value
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (sf[j][mid] < v0) lo = mid + 1; else hi = mid;
    }
    ptr[j] = lo;
    next[j] = lo < 64 ? sf[j][lo] : kMax;
  }
  // Rows go in batches of kWalkBatch: the positions of a batch are computed into LDS first (a column per thread: no barrier), then stored
  // back to back. A store inside the walking loop puts a wait for ALL outstanding memory operations in front of the next
  // position load (gfx9 counts loads and stores in one in-order vmcnt, and across a loop back-edge the compiler waits for zero) —
  // a store round trip per row, 73 % of the kernel's wave cycles in round 3's counters.
  __shared__ unsigned stage[kWalkBatch * kBlock];
  for (int i0 = 0; i0 < len; i0 += kWave) {
    const int my_row = i0 + lane < len ? a.order[begin + i0 + lane] : 0;
    const i64 my_val = i0 + lane < len ? a.val[my_row] : 0;
    const int m = min(kWave, len - i0);
    for (int h = 0; h < m; h += kWalkBatch) {
      const int he = min(h + kWalkBatch, m);
      for (int i = h; i < he; ++i) {
        const i64 v = readlane_i64(my_val, i);
        unsigned packed = 0;
#pragma unroll
        for (int j = 0; j < kWalkWords; ++j) {
          while (next[j] < v) {
            ++ptr[j];
            next[j] = ptr[j] < 64 ? sf[j][ptr[j]] : kMax;
          }
          packed |= (unsigned)ptr[j] << (8 * j);
        }
        stage[(i - h) * kBlock + threadIdx.x] = packed;
      }
      if (live)
        for (int i = h; i < he; ++i) {
          const int row = __builtin_amdgcn_readlane(my_row, i);
          *(unsigned*)(out + (size_t)row * stride + w0) = stage[(i - h) * kBlock + threadIdx.x];
        }
    }
  }
}

// Rank order only. pfx[big][w] = the largest free value among the nodes of words 0..w of the bin-pack order (the last entry of a
// word's sorted list is its maximum; positions past N carry INT64_MIN). One block per walked dimension, chunked inclusive
// max-scan with a carry. k_decide starts a class's scan at the first word whose running maximum reaches the class's value.
constexpr int kPfxBlock = 1024;
__global__ __launch_bounds__(kPfxBlock) void k_dim_prefix_max(DimWalk a, i64* __restrict__ pfx) {
  __shared__ i64 tmp[kPfxBlock];
  const int big = blockIdx.x, t = threadIdx.x;
  const i64 kMin = (i64)0x8000000000000000ull;
  i64 carry = kMin;
  for (int base = 0; base < a.n_words; base += kPfxBlock) {
    const int w = base + t;
    i64 v = w < a.n_words ? a.sfree[((size_t)big * a.n_words + w) * 64 + 63] : kMin;
    tmp[t] = v;
    __syncthreads();
    for (int off = 1; off < kPfxBlock; off <<= 1) {
      const i64 o = t >= off ? tmp[t - off] : kMin;
      __syncthreads();
      v = v > o ? v : o;
      tmp[t] = v;
      __syncthreads();
    }
    v = v > carry ? v : carry;
    if (w < a.n_words) pfx[(size_t)big * a.n_words + w] = v;
    carry = tmp[kPfxBlock - 1] > carry ? tmp[kPfxBlock - 1] : carry;
    __syncthreads();
  }
}

// Rank order: the WINDOW of every index row instead of the row (Planes::res_win). The decision scan of a class begins at the
// first word its rows can have a bit in — for an index row of value v the first word whose running maximum of the free values
// reaches v (pfx) — and finds its node within a few words, so of the 784 bytes of a rank-ordered row it read a dozen; the rows
// were written whole only for that (0.45 ms and 0.8 GB per pass at 10^6 rows). A thread owns four adjacent words as in
// k_dim_walk and stores a row's dword only when its word group lies in the row's window [start & ~3, +64): start <= w0 + 3 and
// start >= w0 - 60, i.e. pfx[w0 - 61] < v <= pfx[w0 + 3] — a contiguous run of the ascending values of a chunk, and no run at
// all for most (chunk, thread) pairs: a twelfth of the walk's work.
// (BATCH = rows staged in LDS before they are stored: kWalkBatch for the walk over every row; 8 — 8 KB of LDS — for the short lists of
// a pass whose sweep runs need no window: that launch runs BESIDE k_sweep_rows, whose workgroups leave 16 KB of every CU's LDS)
template <int BATCH>
__global__ __launch_bounds__(kBlock) void k_dim_walk_window(DimWalk a, const i64* __restrict__ pfx, unsigned char* __restrict__ win) {
  const int chunk = blockIdx.x;
  const int big = a.chunk_big[chunk], begin = a.chunk_begin[chunk], len = a.chunk_len[chunk];
  const int lane = threadIdx.x % kWave;
  const int w0 = (blockIdx.y * kBlock + threadIdx.x) * kWalkWords;
  if (w0 - lane * kWalkWords >= a.n_words) return;  // whole wave beyond the row
  const bool live = w0 < a.n_words;
  const i64 kMax = 0x7fffffffffffffffll, kMin = (i64)0x8000000000000000ull;
  const i64* px = pfx + (size_t)big * a.n_words;
  const i64 hi = live ? px[min(w0 + 3, a.n_words - 1)] : kMin;  // v <= hi: the row can have a bit at or before this word group
  const i64 lo = (live && w0 >= 61) ? px[w0 - 61] : kMin;         // v > lo:  ... and not before word w0 - 60
  const i64 v_first = a.val[a.order[begin]], v_last = a.val[a.order[begin + len - 1]];
  const bool mine = live && hi > lo && v_last > lo && v_first <= hi;
  if (__ballot(mine) == 0) return;  // (wave-uniform: the value exchanges below need every lane)
  const i64* sf[kWalkWords];
  int ptr[kWalkWords];
  i64 next[kWalkWords];
#pragma unroll
  for (int j = 0; j < kWalkWords; ++j) {
    const int w = min(w0 + j, a.n_words - 1);
    sf[j] = a.sfree + ((size_t)big * a.n_words + w) * 64;
    ptr[j] = 0;
    next[j] = mine ? sf[j][0] : kMax;
  }
  bool started = false;
  // (positions of a batch of 64 rows into LDS, then the stores back to back — see k_dim_walk; 0xffffffff = the row's window does not
  // hold this thread's words)
  __shared__ unsigned stage[BATCH * kBlock];
  for (int i0 = 0; i0 < len; i0 += kWave) {
    const int my_row = i0 + lane < len ? a.order[begin + i0 + lane] : 0;
    const i64 my_val = i0 + lane < len ? a.val[my_row] : 0;
    const int m = min(kWave, len - i0);
   for (int h = 0; h < m; h += BATCH) {
    const int he = min(h + BATCH, m);
    // (a batch whose values all lie outside this wave's windows is skipped whole)
    const i64 b_first = readlane_i64(my_val, h), b_last = readlane_i64(my_val, he - 1);
    if (__ballot(mine && b_last > lo && b_first <= hi) == 0) continue;
    for (int i = h; i < he; ++i) {
      const i64 v = readlane_i64(my_val, i);
      unsigned packed = 0xffffffffu;
      if (mine && v > lo && v <= hi) {
        if (!started) {  // the first row of this thread's run: one binary search per word, then the walk only advances
          started = true;
#pragma unroll
          for (int j = 0; j < kWalkWords; ++j) {
            int l = 0, h = 64;
            while (l < h) {
              const int mid = (l + h) >> 1;
              if (sf[j][mid] < v) l = mid + 1; else h = mid;
            }
            ptr[j] = l;
            next[j] = l < 64 ? sf[j][l] : kMax;
          }
        }
        packed = 0;
#pragma unroll
        for (int j = 0; j < kWalkWords; ++j) {
          while (next[j] < v) {
            ++ptr[j];
            next[j] = ptr[j] < 64 ? sf[j][ptr[j]] : kMax;
          }
          packed |= (unsigned)ptr[j] << (8 * j);
        }
      }
      stage[(i - h) * kBlock + threadIdx.x] = packed;
    }
    for (int i = h; i < he; ++i) {
      const int row = __builtin_amdgcn_readlane(my_row, i);
      const unsigned packed = stage[(i - h) * kBlock + threadIdx.x];
      if (packed != 0xffffffffu) *(unsigned*)(win + (size_t)row * 64 + (w0 & 63)) = packed;
    }
   }
  }
}

// NodeAffinity PreFilter + Filter (A.5). Signature = (flags, Filter DNF, PreFilter node-name DNF).
struct AffSigs {
  const unsigned* flags;  // [D]
  const int* term_off;    // [D+1]
  const u64* terms;       // [T][W]
  const int* pre_off;     // [D+1]
  const u64* pre_terms;   // [M][W]
};
// `more` = the node's label words from kMaxW on (stride = nodes), consulted only where a term has bits there: large
// dictionaries (one requirement per hostname ...) are sparse per term.
__device__ __forceinline__ bool dnf_match(const u64* __restrict__ terms, int t0, int t1, const u64 (&lb)[kMaxW], int W,
                                          const u64* __restrict__ more, size_t more_stride) {
  bool any = false;
  for (int t = t0; t < t1; ++t) {
    const u64* m = terms + (size_t)t * W;
    bool all = true;
#pragma unroll
    for (int w = 0; w < kMaxW; ++w)
      if (w < W) all = all && (lb[w] & m[w]) == m[w];
    for (int w = kMaxW; w < W; ++w) {
      const u64 mw = m[w];  // wave-uniform
      if (mw) all = all && more && (more[(size_t)(w - kMaxW) * more_stride] & mw) == mw;
    }
    any = any || all;
  }
  return any;
}
// ---------------------------------------------------------------------------------------------------
// Bit-sliced signature planes for the two dictionary-based families (TaintToleration+NodeUnschedulable, NodeAffinity).
// Every dictionary bit (taint t, requirement q) first becomes its own N-bit BASE PLANE (k_base_planes: lane = node,
// 64 ballots per dictionary word). A signature's plane is then pure bitwise algebra on 64-node words — lane = word:
//   tol plane = exists & ~(OR of the base planes of the dictionary taints the spec does NOT tolerate) [& ~unschedulable]
//   aff plane = exists & prefilter-names-DNF & filter-DNF,  DNF = OR over terms of (AND of the term's requirement planes)
// i.e. per 64 (signature,node) verdicts a handful of loads and ANDs instead of 64 lane-wise mask compares.
// ---------------------------------------------------------------------------------------------------
struct BasePlanes {
  u64* req;      // [64*W][stride]  requirement q
  u64* taint;    // [64*KT][stride] dictionary taint t
  u64* port;     // [64*KP][stride] dictionary host port k in conflict on the node
  u64* unsched;  // [stride] node.Spec.Unschedulable
  u64* exists;   // [stride] bit set for positions < N (zero padding of the last word)
  int stride;
};
// blockIdx.x: dictionary word (0..W-1 labels, then KT taint words, then KP port words, last = flags); blockIdx.y: group of 4 node words.
__device__ __forceinline__ void plane_base(const NodeTable& t, const int* __restrict__ perm, const BasePlanes& o, int n_words) {
  int word;
  const int n = plane_node(t.n, perm, &word);
  if (word >= n_words) return;
  const int lane = threadIdx.x % kWave;
  const int c = blockIdx.x;
  if (c < t.W + t.KT + t.KP) {
    const bool is_label = c < t.W, is_taint = !is_label && c < t.W + t.KT;
    const u64 v = n < 0 ? 0ull
                        : (is_label ? t.labels[(size_t)c * t.n + n]
                                    : (is_taint ? t.taints[(size_t)(c - t.W) * t.n + n] : t.ports[(size_t)(c - t.W - t.KT) * t.n + n]));
    u64 keep = 0;
#pragma unroll 8
    for (int b = 0; b < 64; ++b) {
      u64 m = __ballot((v >> b) & 1ull);
      if (b == lane) keep = m;
    }
    u64* dst = is_label ? o.req + (size_t)(c * 64 + lane) * o.stride
                        : (is_taint ? o.taint + (size_t)((c - t.W) * 64 + lane) * o.stride
                                    : o.port + (size_t)((c - t.W - t.KT) * 64 + lane) * o.stride);
    dst[word] = keep;
  } else {
    u64 un = __ballot(n >= 0 && (t.flags[n >= 0 ? n : 0] & kNodeUnschedulable));
    u64 ex = __ballot(n >= 0);
    if (lane == 0) {
      o.unsched[word] = un;
      o.exists[word] = ex;
    }
  }
}

__global__ __launch_bounds__(kBlock) void k_base_planes(NodeTable t, const int* __restrict__ perm, BasePlanes o, int n_words) {
  if ((int)blockIdx.x < t.W + t.KT + t.KP + 1) plane_base(t, perm, o, n_words);
}

struct SigPlaneArgs {
  BasePlanes base;
  PlaneOut tol, aff;                 // outputs (canon or ranked pointer pre-selected in `.canon`)
  const u64* sig_tol;                // [Dtol][KT]
  const unsigned* sig_tolflags;      // [Dtol]
  const u64* sig_ports;              // [Dtol][KP] requested host ports of the signature
  int KP;
  u64 taint_used[kMaxKT];            // dictionary taints that occur on some node (other base planes are all zero)
  AffSigs affs;
  int KT, W;
  unsigned pre_mask, filt_mask;
  int n_words;
};
constexpr int kBitSigsPerBlock = 8;  // signatures per block; thread = one 64-node word of the row

// (Round 3 tried loading the plane words of a term four at a time — unconditional loads, empty slots pointed at a neutral
// row — to break the "next set bit → load → combine" dependency: k_sig_planes got SLOWER, 0.50 → 0.65 ms on 46 k signatures;
// the extra loads of the neutral row cost more than the shorter chain saved. profiles/r03_session5_*.txt)
//
// What the kernel waits on is LATENCY, signature after signature: the loads of signature i + 1 sit behind the store of
// signature i in the one in-order vmcnt of gfx9, and every signature is a couple of loads and one store. WPL > 1 gives a lane
// WPL words of the row, 64 apart (each load / store of a wave is still 512 contiguous bytes): the same chain per signature
// moves WPL times the bytes, with WPL independent loads in flight per base plane and no extra traffic.
template <int WPL>
__device__ __forceinline__ void dnf_words(const u64* __restrict__ terms, int t0, int t1, int W, const u64* __restrict__ req, int stride,
                                          const int (&w)[WPL], u64 (&any)[WPL]) {
#pragma unroll
  for (int j = 0; j < WPL; ++j) any[j] = 0;
  for (int t = t0; t < t1; ++t) {
    u64 all[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) all[j] = ~0ull;
    for (int k = 0; k < W; ++k) {
      u64 m = terms[(size_t)t * W + k];  // wave-uniform
      while (m) {
        int q = __ffsll((long long)m) - 1;
        m &= m - 1;
        const u64* row = req + (size_t)(k * 64 + q) * stride;
#pragma unroll
        for (int j = 0; j < WPL; ++j) all[j] &= row[w[j]];
      }
    }
#pragma unroll
    for (int j = 0; j < WPL; ++j) any[j] |= all[j];
  }
}
// The same DNF with the term masks of the block's signatures held in registers (lane j of tw[r] = word r * 64 + j of the
// block's slice of `terms`): j0 / j1 index that slice. One v_readlane pair per mask word instead of a scalar load whose
// address depends on the previous one.
constexpr int kSigTermRegs = 4;  // 256 mask words per block of signatures; larger slices take the scalar path
template <int WPL>
__device__ __forceinline__ void dnf_words_regs(const u64 (&tw)[kSigTermRegs], int j0, int j1, int W, const u64* __restrict__ req, int stride,
                                               const int (&w)[WPL], u64 (&any)[WPL]) {
#pragma unroll
  for (int j = 0; j < WPL; ++j) any[j] = 0;
  for (int jj = j0; jj < j1;) {
    u64 all[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) all[j] = ~0ull;
    for (int k = 0; k < W; ++k, ++jj) {
      const int r = jj >> 6, l = jj & 63;  // wave-uniform
      u64 m = (u64)readlane_i64((i64)(r == 0 ? tw[0] : (r == 1 ? tw[1] : (r == 2 ? tw[2] : tw[3]))), l);
      while (m) {
        int q = __ffsll((long long)m) - 1;
        m &= m - 1;
        const u64* row = req + (size_t)(k * 64 + q) * stride;
#pragma unroll
        for (int j = 0; j < WPL; ++j) all[j] &= row[w[j]];
      }
    }
#pragma unroll
    for (int j = 0; j < WPL; ++j) any[j] |= all[j];
  }
}
// blockIdx.x: chunk of kBitSigsPerBlock signatures, blockIdx.y: block of 256 * WPL row words, blockIdx.z: 0 = tol family, 1 = aff family.
// Wave v of the block owns words [(blockIdx.y * 4 + v) * 64 * WPL, +64 * WPL); lane l its words base + j * 64 + l, j < WPL.
// The signature tables of a block (flags, offsets, tolerated-taint words, term masks) are fetched ONCE with vector loads —
// lane i holds signature d0 + i's entry — and broadcast with v_readlane: the per-signature chain of dependent scalar loads
// (flags → offsets → masks, each a cold miss when every ask has its own template) was what the kernel waited on.
template <int WPL>
__global__ __launch_bounds__(kBlock) void k_sig_planes(SigPlaneArgs a) {
  const int lane = threadIdx.x % kWave;
  const int wbase = (blockIdx.y * kWavesPerBlock + threadIdx.x / kWave) * (kWave * WPL);
  if (wbase >= a.n_words) return;  // whole wave beyond the row
  int w[WPL], w_raw[WPL];
  bool live[WPL];
  u64 exists[WPL];
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    w_raw[j] = wbase + j * kWave + lane;
    live[j] = w_raw[j] < a.n_words;
    w[j] = live[j] ? w_raw[j] : a.n_words - 1;  // every lane of a live wave takes part in the v_readlane exchanges
    exists[j] = a.base.exists[w[j]];
  }
  // first non-zero word of the signature's row among this wave's words (rank-ordered planes only)
  auto note_first = [&](int* first, int d, const u64 (&val)[WPL]) {
    if (!first) return;
    int fw = kNoWord;
#pragma unroll
    for (int j = WPL - 1; j >= 0; --j) {
      const u64 nz = __ballot(live[j] && val[j] != 0);
      if (nz) fw = wbase + j * kWave + __ffsll((long long)nz) - 1;
    }
    if (fw != kNoWord && lane == 0) atomicMin(&first[d], fw);
  };
  const int d0 = blockIdx.x * kBitSigsPerBlock;
  if (blockIdx.z == 0) {
    const int nd = min(kBitSigsPerBlock, a.tol.D - d0);
    if (nd <= 0) return;
    const bool taint_en = a.filt_mask & kPlugTaint, unsched_en = a.filt_mask & kPlugUnsched;
    const bool ports_en = (a.filt_mask & kPlugPorts) && (a.pre_mask & kPlugPorts);
    u64 unsched[WPL];
#pragma unroll
    for (int j = 0; j < WPL; ++j) unsched[j] = unsched_en ? a.base.unsched[w[j]] : 0ull;
    const bool batch = nd * a.KT <= kWave && nd * a.KP <= kWave;
    const unsigned fl_l = lane < nd ? a.sig_tolflags[d0 + lane] : 0u;
    u64 tol_l = 0, port_l = 0;
    if (batch) {
      if (lane < nd * a.KT) tol_l = a.sig_tol[(size_t)d0 * a.KT + lane];
      if (ports_en && lane < nd * a.KP) port_l = a.sig_ports[(size_t)d0 * a.KP + lane];
    }
    for (int i = 0; i < nd; ++i) {
      const int d = d0 + i;
      const unsigned fl = (unsigned)__builtin_amdgcn_readlane((int)fl_l, i);
      u64 bad[WPL];  // nodes carrying a taint this signature does not tolerate
#pragma unroll
      for (int j = 0; j < WPL; ++j) bad[j] = 0;
      if (taint_en)
        for (int k = 0; k < a.KT; ++k) {
          const u64 tolerated = batch ? (u64)readlane_i64((i64)tol_l, i * a.KT + k) : a.sig_tol[(size_t)d * a.KT + k];
          u64 m = ~tolerated & a.taint_used[k];
          while (m) {
            int tt = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64* row = a.base.taint + (size_t)(k * 64 + tt) * a.base.stride;
#pragma unroll
            for (int j = 0; j < WPL; ++j) bad[j] |= row[w[j]];
          }
        }
      if (ports_en)  // NodePorts: a requested host port that is in conflict on the node
        for (int k = 0; k < a.KP; ++k) {
          u64 m = batch ? (u64)readlane_i64((i64)port_l, i * a.KP + k) : a.sig_ports[(size_t)d * a.KP + k];
          while (m) {
            int pp = __ffsll((long long)m) - 1;
            m &= m - 1;
            const u64* row = a.base.port + (size_t)(k * 64 + pp) * a.base.stride;
#pragma unroll
            for (int j = 0; j < WPL; ++j) bad[j] |= row[w[j]];
          }
        }
      u64 val[WPL];
#pragma unroll
      for (int j = 0; j < WPL; ++j) {
        u64 b = bad[j];
        if (!(fl & kSpecToleratesUnsched)) b |= unsched[j];
        if (fl & kSpecUnsupported) b = ~0ull;  // not evaluated by the engine: fits nowhere (this family is always on)
        val[j] = exists[j] & ~b;
        if (live[j]) a.tol.canon[(size_t)d * a.tol.stride + w[j]] = val[j];
      }
      note_first(a.tol.first, d, val);
    }
  } else {
    const int nd = min(kBitSigsPerBlock, a.aff.D - d0);
    if (nd <= 0) return;
    const bool pre_en = a.pre_mask & kPlugAffinity, filt_en = a.filt_mask & kPlugAffinity;
    const unsigned f_l = lane < nd ? a.affs.flags[d0 + lane] : 0u;
    const int to_l = lane <= nd ? a.affs.term_off[d0 + lane] : 0;
    const int tbase = __builtin_amdgcn_readlane(to_l, 0);
    const int n_mask_words = (__builtin_amdgcn_readlane(to_l, nd) - tbase) * a.W;
    const bool batch = filt_en && n_mask_words <= kSigTermRegs * kWave;
    u64 tw[kSigTermRegs];
#pragma unroll
    for (int r = 0; r < kSigTermRegs; ++r) {
      const int j = r * kWave + lane;
      tw[r] = batch && j < n_mask_words ? a.affs.terms[(size_t)tbase * a.W + j] : 0ull;
    }
    for (int i = 0; i < nd; ++i) {
      const int d = d0 + i;
      const unsigned f = (unsigned)__builtin_amdgcn_readlane((int)f_l, i);
      const bool skip = pre_en && (f & kSpecAffSkip);  // predicate_manager.go:233-234,264-266
      u64 ok[WPL];
#pragma unroll
      for (int j = 0; j < WPL; ++j) ok[j] = exists[j];
      if (pre_en && !skip) {
        if (f & kSpecPreReject) {  // PreFilter rejected the pod (:236-238)
#pragma unroll
          for (int j = 0; j < WPL; ++j) ok[j] = 0;
        }
        if (f & kSpecPreNames) {  // "node not eligible" (:248-250)
          u64 any[WPL];
          dnf_words<WPL>(a.affs.pre_terms, a.affs.pre_off[d], a.affs.pre_off[d + 1], a.W, a.base.req, a.base.stride, w, any);
#pragma unroll
          for (int j = 0; j < WPL; ++j) ok[j] &= any[j];
        }
      }
      if (filt_en && !skip) {
        const int t0 = __builtin_amdgcn_readlane(to_l, i), t1 = __builtin_amdgcn_readlane(to_l, i + 1);
        u64 any[WPL];
        if (batch)
          dnf_words_regs<WPL>(tw, (t0 - tbase) * a.W, (t1 - tbase) * a.W, a.W, a.base.req, a.base.stride, w, any);
        else
          dnf_words<WPL>(a.affs.terms, t0, t1, a.W, a.base.req, a.base.stride, w, any);
#pragma unroll
        for (int j = 0; j < WPL; ++j) ok[j] &= any[j];
      }
#pragma unroll
      for (int j = 0; j < WPL; ++j)
        if (live[j]) a.aff.canon[(size_t)d * a.aff.stride + w[j]] = ok[j];
      note_first(a.aff.first, d, ok);
    }
  }
}

// Rank-ordered planes by bit permutation: ranked[d] bit i = canonical[d] bit perm[i]. One launch covers every family
// (their planes are rows of one buffer), replacing a second evaluation of all predicates in permuted node order.
__global__ __launch_bounds__(kBlock) void k_permute_planes(int n_nodes, const int* __restrict__ perm, const u64* __restrict__ canon,
                                                           u64* __restrict__ ranked, int stride, int n_rows, int n_words,
                                                           int* __restrict__ first /* may be null */) {
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int w = blockIdx.y * kWavesPerBlock + wave;
  if (w >= n_words) return;
  const int pos = w * kWave + lane;
  const int src = pos < n_nodes ? perm[pos] : -1;
  const int sw = src >> 6, sb = src & 63;
  const int spb = sigs_per_block(n_rows);
  const int d0 = blockIdx.x * spb;
  const int dend = min(d0 + spb, n_rows);
  u64 keep = 0;
  for (int d = d0; d < dend; ++d) {
    bool bit = src >= 0 && ((canon[(size_t)d * stride + sw] >> sb) & 1ull);
    u64 b = __ballot(bit);
    if (d - d0 == lane) keep = b;
  }
  if (lane < dend - d0) {
    ranked[(size_t)(d0 + lane) * stride + w] = keep;
    if (first && keep) atomicMin(&first[d0 + lane], w);
  }
}

// ---------------------------------------------------------------------------------------------------
// PodTopologySpread. PreFilter = a histogram over nodes per hard constraint, Filter = a skew test.
// The reference recomputes the histogram for EVERY (pod,node) call (predicate_manager.go:221-254 lists all nodes per
// pair); here it is built once per distinct (constraints, eligibility) signature.
// ---------------------------------------------------------------------------------------------------
struct SpreadC {  // one constraint of one topology signature (PodTopologySpread constraints first, then InterPodAffinity)
  int kd, ks, max_skew, min_domains, self_match;
  unsigned flags;
  int cnt_off, dom_size;  // cells [cnt_off, cnt_off + dom_size) of cnt/present belong to this constraint
  int kind;               // kKind*
};
struct SpreadSigs {
  int D;                // signatures
  const int* c_off;     // [D+1] → rows of c
  const SpreadC* c;     // [G]
  const int* aff_sig;   // [D] row of the AffSigs tables whose Filter DNF decides nodeAffinityPolicy=Honor eligibility
  const int* tol_sig;   // [D] row of the toleration table for nodeTaintsPolicy=Honor
  int* cnt;             // [cells] matching pods per (constraint, domain) over eligible nodes   (TpPairToMatchNum)
  int* present;         // [cells] 1 = some eligible node carries the domain                       (TpKeyToDomainsNum)
  int* minv;            // [G] global minimum after the minDomains rule                           (criticalPaths / minMatchNum)
};

// Is node n one of the nodes whose pods COUNT for the PodTopologySpread constraints of signature d? It must carry ALL topology
// keys of the signature's spread constraints (nodeLabelsMatchSpreadConstraints) and pass the inclusion policies a constraint asks
// for (matchNodeInclusionPolicies: nodeAffinityPolicy / nodeTaintsPolicy == Honor). Node properties only — pods do not move it.
struct SpreadElig {
  bool keys, aff_ok, tol_ok;
};
__device__ __forceinline__ SpreadElig spread_eligibility(const NodeTable& t, const SpreadSigs& sp, const AffSigs& aff, const u64* __restrict__ sig_tol,
                                                         int d, int n) {
  const int c0 = sp.c_off[d], c1 = sp.c_off[d + 1];
  SpreadElig el{true, true, true};
  bool need_aff = false, need_tol = false;
  for (int g = c0; g < c1; ++g) {
    if (sp.c[g].kind != kKindSpread) continue;
    if (t.domain[(size_t)sp.c[g].kd * t.n + n] < 0) el.keys = false;  // nodeLabelsMatchSpreadConstraints
    need_aff |= (sp.c[g].flags & kSpreadHonorAffinity) != 0;
    need_tol |= (sp.c[g].flags & kSpreadHonorTaints) != 0;
  }
  if (el.keys && need_aff) {
    u64 lb[kMaxW];
#pragma unroll
    for (int w = 0; w < kMaxW; ++w) lb[w] = w < t.W ? t.labels[(size_t)w * t.n + n] : 0;
    const int a = sp.aff_sig[d];
    el.aff_ok = dnf_match(aff.terms, aff.term_off[a], aff.term_off[a + 1], lb, t.W, t.labels + (size_t)kMaxW * t.n + n, (size_t)t.n);
  }
  if (el.keys && need_tol) {
    const u64* tol = sig_tol + (size_t)sp.tol_sig[d] * t.KT;
    for (int k = 0; k < t.KT; ++k) el.tol_ok = el.tol_ok && (t.taints[(size_t)k * t.n + n] & ~tol[k]) == 0;
  }
  return el;
}
// does a pod on node n (domain `dom` of the constraint's key, already known to be a valid id) count for constraint c?
__device__ __forceinline__ bool spread_counts_here(const SpreadC& c, const SpreadElig& el) {
  if (c.kind != kKindSpread) return true;  // InterPodAffinity: the node only needs the constraint's own key (topologyToMatchedTermCount.update)
  if (!el.keys) return false;
  if ((c.flags & kSpreadHonorAffinity) && !el.aff_ok) return false;  // matchNodeInclusionPolicies
  if ((c.flags & kSpreadHonorTaints) && !el.tol_ok) return false;
  return true;
}
// thread = node; blockIdx.x = signature, blockIdx.y = block of 256 nodes. Eligible nodes add their match counts to their
// domain's cell.
__global__ __launch_bounds__(kBlock) void k_spread_count(NodeTable t, SpreadSigs sp, AffSigs aff, const u64* __restrict__ sig_tol) {
  const int n = blockIdx.y * kBlock + threadIdx.x;
  if (n >= t.n) return;
  const int d = blockIdx.x;
  const int c0 = sp.c_off[d], c1 = sp.c_off[d + 1];
  const SpreadElig el = spread_eligibility(t, sp, aff, sig_tol, d, n);
  for (int g = c0; g < c1; ++g) {
    const SpreadC c = sp.c[g];
    const int dom = t.domain[(size_t)c.kd * t.n + n];
    if (dom < 0 || dom >= c.dom_size) continue;
    if (!spread_counts_here(c, el)) continue;
    if (c.kind == kKindSpread) sp.present[c.cnt_off + dom] = 1;
    const int v = c.ks >= 0 ? t.selcount[(size_t)c.ks * t.n + n] : 0;
    if (v) atomicAdd(&sp.cnt[c.cnt_off + dom], v);
  }
}
// one wave per constraint: minimum over present domains, 0 when fewer domains than minDomains
__global__ __launch_bounds__(kWave) void k_spread_min(SpreadSigs sp, int n_constraints) {
  const int g = blockIdx.x;
  if (g >= n_constraints) return;
  const SpreadC c = sp.c[g];
  int mn = 0x7fffffff, nd = 0;
  for (int i = threadIdx.x; i < c.dom_size; i += kWave)
    if (sp.present[c.cnt_off + i]) {
      mn = min(mn, sp.cnt[c.cnt_off + i]);
      ++nd;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, __shfl_down(mn, off, kWave));
    nd += __shfl_down(nd, off, kWave);
  }
  if (c.kind != kKindSpread) {
    // InterPodAffinity: minv holds the total number of matches over all domains (len(affinityCounts) == 0 test)
    int tot = 0;
    for (int i = threadIdx.x; i < c.dom_size; i += kWave) tot += sp.cnt[c.cnt_off + i] > 0 ? 1 : 0;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) tot += __shfl_down(tot, off, kWave);
    if (threadIdx.x == 0) sp.minv[g] = tot;
    return;
  }
  if (threadIdx.x == 0) sp.minv[g] = nd < c.min_domains ? 0 : mn;
}
// Incremental path (ykpred_eval_nodes): which topology signatures see different PreFilter state than before the node
// change? One wave per signature compares its cells of (cnt, present) and its minv entries with the copies taken before the
// histograms were rebuilt. A pod class whose signature changed gets its whole row rewritten; every other class only the
// columns of the touched nodes.
__global__ __launch_bounds__(kWave) void k_spread_diff(SpreadSigs sp, const int* __restrict__ cnt_prev, const int* __restrict__ present_prev,
                                                       const int* __restrict__ minv_prev, int* __restrict__ sig_changed) {
  const int d = blockIdx.x;
  if (d >= sp.D) return;
  bool diff = false;
  for (int g = sp.c_off[d]; g < sp.c_off[d + 1]; ++g) {
    const SpreadC c = sp.c[g];
    if (sp.minv[g] != minv_prev[g]) diff = true;
    for (int i = threadIdx.x; i < c.dom_size; i += kWave)
      diff = diff || sp.cnt[c.cnt_off + i] != cnt_prev[c.cnt_off + i] || sp.present[c.cnt_off + i] != present_prev[c.cnt_off + i];
  }
  const u64 any = __ballot(diff);
  if (threadIdx.x == 0) sig_changed[d] = any != 0 ? 1 : 0;
}
// thread = class: dirty ⇔ its topology signature changed; a dirty class's feasible count is rebuilt by the combine pass
__global__ __launch_bounds__(kBlock) void k_mark_dirty_classes(int n_classes, const int* __restrict__ class_sig, const int* __restrict__ sig_changed,
                                                               int* __restrict__ class_dirty, int* __restrict__ class_count) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= n_classes) return;
  const int ss = class_sig[c * 4 + 3];
  const int dirty = ss >= 0 && sig_changed[ss] ? 1 : 0;
  class_dirty[c] = dirty;
  if (dirty) class_count[c] = 0;
}

// Filters of the topology constraints of signature d, PodTopologySpread first (Filter order of predicate_manager.go:339-352):
//   spread:  fail ⇔ the topology label is missing, or matchNum + selfMatch − min > maxSkew
//   InterPodAffinity (satisfyPodAffinity / satisfyPodAntiAffinity / satisfyExistingPodsAntiAffinity):
//     every required-affinity key must be on the node; each needs a matching pod in the node's domain unless NO pod
//     matches anywhere and the pod matches its own terms; an anti-affinity / existing-anti-affinity match in the node's
//     domain fails.
// Returns 0 = ok, 7 = PodTopologySpread failed, 8 = InterPodAffinity failed.
// LIVE: the histograms are being changed by the running kernel (ykpred_allocate_round) — read them past the L1.
template <bool LIVE = false>
__device__ __forceinline__ int constraints_fail(const SpreadSigs& sp, int d, const int (&dom)[kMaxKD], bool spread_en, bool ipa_en,
                                                unsigned* missing) {
  bool pods_exist = true, any_affinity = false, self_match = false;
  int aff_domains = 0;
  int ipa_fail = 0;
  for (int g = sp.c_off[d]; g < sp.c_off[d + 1]; ++g) {
    const SpreadC c = sp.c[g];
    int dm = -1;
#pragma unroll
    for (int k = 0; k < kMaxKD; ++k)
      if (k == c.kd) dm = dom[k];
    const i64 match = (dm >= 0 && dm < c.dom_size) ? ld_maybe_live<LIVE>(sp.cnt + c.cnt_off + dm) : 0;
    if (c.kind == kKindSpread) {
      if (!spread_en) continue;
      if (dm < 0) {
        if (missing) *missing = 1;
        return 7;
      }
      const i64 m = (dm < c.dom_size && sp.present[c.cnt_off + dm]) ? match : 0;
      if (m + c.self_match - (i64)ld_maybe_live<LIVE>(sp.minv + g) > (i64)c.max_skew) return 7;
    } else if (ipa_en && !ipa_fail) {
      if (c.kind == kKindPodAffinity) {
        any_affinity = true;
        self_match = c.self_match != 0;
        aff_domains += ld_maybe_live<LIVE>(sp.minv + g);
        if (dm < 0) ipa_fail = 8;  // all topology labels must exist on the node
        if (match <= 0) pods_exist = false;
      } else if (dm >= 0 && match > 0) {
        ipa_fail = 8;
      }
    }
  }
  if (ipa_en && !ipa_fail && any_affinity && !pods_exist && !(aff_domains == 0 && self_match)) ipa_fail = 8;
  return ipa_fail;
}
__device__ __forceinline__ void plane_spread(const NodeTable& t, const int* __restrict__ perm, const SpreadSigs& sp, const PlaneOut& o,
                                             int n_words, bool spread_en, bool ipa_en) {
  int word;
  int n = plane_node(t.n, perm, &word);
  if (word >= n_words) return;
  int dom[kMaxKD];
#pragma unroll
  for (int k = 0; k < kMaxKD; ++k) dom[k] = (n >= 0 && k < t.KD) ? t.domain[(size_t)k * t.n + n] : -1;
  const int spb = sigs_per_block(o.D);
  int d0 = blockIdx.x * spb;
  int dend = min(d0 + spb, o.D);
  u64 keep = 0;
  for (int d = d0; d < dend; ++d) {
    bool ok = n >= 0 && constraints_fail(sp, d, dom, spread_en, ipa_en, nullptr) == 0;
    u64 b = __ballot(ok);
    if ((d - d0) == (int)(threadIdx.x % kWave)) keep = b;
  }
  plane_store(o, perm != nullptr, word, d0, dend - d0, keep);
}

// Ballot-based families (NodeResourcesFit value planes, PodTopologySpread) in one launch: blockIdx.z selects the
// family so their (latency-bound, cache-cold) walks overlap. A disabled family has no chunks / D = 0.
struct PlaneArgs {
  const int* perm;
  PlaneOut res, spread;
  DimPlanes dims;  // the ballot-evaluated dimensions (few distinct values); the others go through k_dim_sort / k_dim_walk
  SpreadSigs spreads;
  int fit_error, n_words;
  int spread_en, ipa_en;
};
__global__ __launch_bounds__(kBlock) void k_planes(NodeTable t, PlaneArgs a) {
  if (blockIdx.z == 0) {
    if ((int)blockIdx.x < a.dims.n_chunks) plane_dim(t, a.perm, a.dims, a.res, a.fit_error, a.n_words);
  } else {
    if ((int)blockIdx.x * sigs_per_block(a.spread.D) < a.spread.D) plane_spread(t, a.perm, a.spreads, a.spread, a.n_words, a.spread_en != 0, a.ipa_en != 0);
  }
}
// The three node-reading plane kernels of one node order in ONE launch (they are independent of each other and each is a few
// microseconds of latency: launched one after the other they are three boundaries on the step's critical path). blockIdx.z:
// 0 = request-value planes, 1 = topology planes, 2 = bit-sliced dictionaries (k_base_planes).
__global__ __launch_bounds__(kBlock) void k_node_planes(NodeTable t, PlaneArgs a, BasePlanes base) {
  if (blockIdx.z == 0) {
    if ((int)blockIdx.x < a.dims.n_chunks) plane_dim(t, a.perm, a.dims, a.res, a.fit_error, a.n_words);
  } else if (blockIdx.z == 1) {
    if ((int)blockIdx.x * sigs_per_block(a.spread.D) < a.spread.D) plane_spread(t, a.perm, a.spreads, a.spread, a.n_words, a.spread_en != 0, a.ipa_en != 0);
  } else {
    if ((int)blockIdx.x < t.W + t.KT + t.KP + 1) plane_base(t, a.perm, base, a.n_words);
  }
}

// ---------------------------------------------------------------------------------------------------
// combine: class row = AND of its planes; stream it into every member pod's bitmap row
// ---------------------------------------------------------------------------------------------------
struct ClassTable {
  const int* sig;       // [C][4]: res, tol, aff, spread plane row (-1 = family not evaluated ⇒ all ones)
  const int* pin;       // [C]: pinned node (NodeName), -1 none, -2 unknown name
  const int* chunk_class;  // [NC]
  const int* chunk_begin;  // [NC] offset into members
  const int* chunk_len;    // [NC] 1..kChunkMembers
  const int* chunk_first;  // [NC] 1 = first chunk of its class (owns the popcount)
  const int* members;      // chunk row lists: PHYSICAL bitmap rows (ykpred_layout_t.row_of_pod), -1 = unused entry
  const int* chunk_zone;   // [NC] 1 = the chunk's class lives in zone A (the full pass writes it with k_expand_bands); 2 = zone B,
                           // a class of a sweep run: written by k_sweep_rows in the full passes that run it (sweep_on), else like zone B
  int sweep_on;
  const int* decide_list;  // or null: the classes k_decide scans (those whose decision does not come from k_run_decide), n_classes = their number
};
// full pass: is the chunk left to another writer (the band writer, k_sweep_rows)?
// (zone 3: a class with a FuseRec — k_fused_rows writes its rows in the passes that run it: sweep_on bit 1)
__device__ __forceinline__ bool chunk_elsewhere(const ClassTable& ct, int chunk) {
  const int z = ct.chunk_zone[chunk];
  return z == 1 || (z == 2 && (ct.sweep_on & 1)) || (z == 3 && (ct.sweep_on & 2));
}
struct Planes {
  const u64* res;         // value planes of NodeResourcesFit (row 0 = pod-independent part); null = family disabled
  const u64* tol;
  const u64* aff;
  const u64* spread;
  int stride;
  const int* res_rows;    // [Dvec][res_slots]: rows of `res` a request vector ANDs together, -1 = unused slot; a row of a
                          // sorted-walk dimension carries (walked dimension + 1) << kRowBigShift: it is an INDEX row
  int res_slots;          // 1 + R
  const unsigned char* res_idx;  // [rows][idx_stride] index rows of the sorted-walk dimensions (k_dim_walk), indexed by plane row id
  int idx_stride;
  const u64* pmask;       // [walked dimensions][n_words][65] mask tables of k_dim_sort (entry 64 = no node)
  int n_words;
  const int* first;       // rank-ordered planes: first non-zero word per plane row of the whole buffer (null: not kept) ...
  int base_res, base_tol, base_aff, base_spread;  // ... indexed by family base + signature
  int n_big;              // walked dimensions with a mask table in `pmask` and rank planes in `rbits`
  const i64* res_val;     // [rows] request value of a plane / index row of `res` (rank-ordered planes: with `pfx`, else null)
  const i64* pfx;         // [walked dimensions][n_words] rank order only: largest free value among the nodes of words 0..w
                          // (k_dim_prefix_max) — a row of value v has no bit before the first word with pfx >= v
  const u64* rbits;       // [walked dimensions][kRankBits][n_words] the same tables bit-sliced (k_walk_rows); null: not kept (rank order)
  // Rank order keeps only a WINDOW of every index row (res_idx is null then): the 64 bytes of the words [start & ~3, +64), start =
  // the first word the row can have a bit in (pfx) — the decision scan begins there and rarely gets further; byte of word w at
  // res_win[row * 64 + (w & 63)]. A word outside the window is decoded from the word's sorted free list instead (sfree).
  const unsigned char* res_win;  // [rows][64]
  const i64* sfree;              // [walked dimensions][n_words][64] ascending free values of every word (k_dim_sort)
};
constexpr int kMaxClassRows = 3 + 1 + kMaxR;
constexpr int kMaxIdxRows = 2;     // sorted-walk dimensions (further many-valued dimensions stay on ballot planes)
constexpr int kRowBigShift = 28;   // plane row ids stay below 2^28

// The plane rows whose AND is the bitmap row of a class: pointers to their first words, families disabled for this
// evaluation left out (a null family pointer; wave-uniform).
struct ClassRows {
  const u64* row[kMaxClassRows];
  int n;
  const unsigned char* irow[kMaxIdxRows];  // index rows (sorted-walk dimensions) ...
  const u64* ipm[kMaxIdxRows];             // ... and the mask table of their dimension
  // windowed index rows (rank order): irow = the row's 64-byte window; the value, the window's first word group (words / 4) and the
  // sorted free lists of the dimension for the words outside it. isf == null: irow is a whole row.
  const i64* isf[kMaxIdxRows];
  i64 ival[kMaxIdxRows];
  int ig4[kMaxIdxRows];
  int ni;
  int start;                               // no row of the class has a bit before this word (kNoWord: some row is empty)
};
__device__ __forceinline__ ClassRows class_rows(const Planes& pl, int sr, int st, int sa, int ss) {
  ClassRows cr;
  cr.n = 0;
  cr.ni = 0;
  cr.start = 0;
#pragma unroll
  for (int i = 0; i < kMaxClassRows; ++i) cr.row[i] = nullptr;
#pragma unroll
  for (int i = 0; i < kMaxIdxRows; ++i) {
    cr.irow[i] = nullptr;
    cr.ipm[i] = nullptr;
    cr.isf[i] = nullptr;
    cr.ival[i] = 0;
    cr.ig4[i] = 0x3fffffff;
  }
  auto add = [&](const u64* p, int global_row) {
#pragma unroll
    for (int i = 0; i < kMaxClassRows; ++i)
      if (i == cr.n) cr.row[i] = p;
    ++cr.n;
    if (pl.first) cr.start = max(cr.start, pl.first[global_row]);
  };
  if (pl.tol && st >= 0) add(pl.tol + (size_t)st * pl.stride, pl.base_tol + st);
  if (pl.aff && sa >= 0) add(pl.aff + (size_t)sa * pl.stride, pl.base_aff + sa);
  if (pl.spread && ss >= 0) add(pl.spread + (size_t)ss * pl.stride, pl.base_spread + ss);
  if (pl.res && sr >= 0) {
    const int* rr = pl.res_rows + (size_t)sr * pl.res_slots;
    for (int k = 0; k < pl.res_slots; ++k) {
      const int r = rr[k];
      if (r < 0) continue;
      const int big = r >> kRowBigShift, rid = r & ((1 << kRowBigShift) - 1);
      if (big) {
        int first_word = -1;  // the first word the row can have a bit in (-1: not known)
        if (pl.pfx) {
          // index rows keep no first-word table (10^6 rows); the walked dimension is monotone instead: nodes with free >= v
          // first appear in the first word whose running maximum reaches v — one binary search in an L1-resident array
          const i64 v = pl.res_val[rid];
          const i64* px = pl.pfx + (size_t)(big - 1) * pl.n_words;
          int lo = 0, hi = pl.n_words;
          while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (px[mid] < v) lo = mid + 1; else hi = mid;
          }
          first_word = lo < pl.n_words ? lo : kNoWord;
          if (pl.first) cr.start = max(cr.start, first_word);
        }
#pragma unroll
        for (int i = 0; i < kMaxIdxRows; ++i)
          if (i == cr.ni) {
            cr.ipm[i] = pl.pmask + (size_t)(big - 1) * pl.n_words * 65;
            if (pl.res_win) {
              cr.irow[i] = pl.res_win + (size_t)rid * 64;
              cr.isf[i] = pl.sfree + (size_t)(big - 1) * pl.n_words * 64;
              cr.ival[i] = pl.res_val[rid];
              cr.ig4[i] = (first_word >= 0 && first_word != kNoWord) ? (first_word >> 2) : 0x3fffffff;  // (no window: every word from the lists)
            } else {
              cr.irow[i] = pl.res_idx + (size_t)rid * pl.idx_stride;
            }
          }
        ++cr.ni;
      } else {
        add(pl.res + (size_t)rid * pl.stride, pl.base_res + rid);
      }
    }
  }
  return cr;
}
// the words of the class's INDEX rows at word w (w < n_words): position byte -> entry of the word's mask table
__device__ __forceinline__ u64 class_idx_word(const ClassRows& cr, int w) {
  u64 v = ~0ull;
#pragma unroll
  for (int i = 0; i < kMaxIdxRows; ++i)
    if (i < cr.ni) {
      int pos;
      if (!cr.isf[i]) {
        pos = cr.irow[i][w];
      } else if ((unsigned)((w >> 2) - cr.ig4[i]) < 16u) {
        pos = cr.irow[i][w & 63];  // inside the row's window
      } else {
        // outside it: the number of the word's free values below the row's value (what k_dim_walk would have stored)
        const i64* sf = cr.isf[i] + (size_t)w * 64;
        int lo = 0, hi = 64;
        while (lo < hi) {
          const int mid = (lo + hi) >> 1;
          if (sf[mid] < cr.ival[i]) lo = mid + 1; else hi = mid;
        }
        pos = lo;
      }
      v &= cr.ipm[i][(size_t)w * 65 + min(pos, 64)];  // (a byte is 0..64; the clamp keeps a stray one inside the table)
    }
  return v;
}
__device__ __forceinline__ u64 class_word(const ClassRows& cr, int w) {
  u64 v = ~0ull;
#pragma unroll
  for (int i = 0; i < kMaxClassRows; ++i)
    if (i < cr.n) v &= cr.row[i][w];
  if (cr.ni) v &= class_idx_word(cr, w);
  return v;
}

// The rows of the band layout that straddle a window boundary are rewritten whole from their class's row (see k_expand_bands).
struct FixRows {
  const u64* class_rows;  // zone-A class-row table
  const int* rows;        // [n] physical bitmap rows
  const int* slots;       // [n] their class's slot in the table
  int n;
};
__device__ __forceinline__ void fix_row(const FixRows& f, int i, u64* __restrict__ out, int row_stride) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  if (i >= f.n) return;
  const u64* src = f.class_rows + (size_t)f.slots[i] * row_stride;
  u64* dst = out + (size_t)f.rows[i] * row_stride;
  for (int w = threadIdx.x * 2; w < row_stride; w += 2 * kBlock) *(u64x2*)(dst + w) = *(const u64x2*)(src + w);
}
// grid.x = chunks, grid.y = row super-segments of tpg*kCombineUnroll*WPL words. The block is split into kBlock/tpg
// thread groups (tpg = threads per group: 64, 128 or 256, a whole number of waves) that write DIFFERENT member rows
// concurrently: wide rows (≥ 512 words, e.g. 50 k nodes) use one group of 256 threads, narrow rows of a node shard
// (e.g. 200 words for 12.5 k nodes) are covered by 128 or 64 threads so that no lane idles. Each thread owns
// kCombineUnroll groups of WPL adjacent words of the class row (word = seg_base + (u*tpg + t)*WPL ...), so one wave
// store writes 64*WPL*8 contiguous bytes (512 B at WPL=1, 1 KiB = dwordx4 per lane at WPL=2).
// (Round-1 experiments with one word per lane and with non-temporal stores were not faster; only this form is kept.)
__global__ __launch_bounds__(kBlock) void k_combine(ClassTable ct, Planes pl, u64* __restrict__ bitmap, int row_words, int row_stride,
                                                    int pin_enabled, int* __restrict__ class_count, int tpg,
                                                    const int* __restrict__ class_dirty /* null = every class */,
                                                    const int* __restrict__ chunk_list /* null: grid.x = every chunk; else the chunks
                                                    to run (the full pass lists the zone-B chunks: no workgroup for the others) */,
                                                    FixRows fix /* workgroups n_run .. n_run + fix.n - 1 rewrite the band layout's
                                                    straddling rows (k_fix_rows' work in this launch); fix.n = 0: none */,
                                                    int n_run) {
  // pin_enabled bit 0: NodeName filter on; bit 1: a Filter has no PreFilter state ⇒ every pair fails
  constexpr int WPL = 2;  // adjacent words per thread: one wave store writes 64 x 16 contiguous bytes
  if ((int)blockIdx.x >= n_run) {
    if (blockIdx.y == 0) fix_row(fix, (int)blockIdx.x - n_run, bitmap, row_stride);
    return;
  }
  const bool all_fail = pin_enabled & 2;
  pin_enabled &= 1;
  const int chunk = chunk_list ? chunk_list[blockIdx.x] : (int)blockIdx.x;
  const int cls = ct.chunk_class[chunk];
  if (class_dirty ? !class_dirty[cls] : chunk_elsewhere(ct, chunk)) return;  // full pass: zone B only; incremental pass: the dirty classes
  const int begin = ct.chunk_begin[chunk];
  const int len = ct.chunk_len[chunk];
  const int sr = ct.sig[cls * 4 + 0], st = ct.sig[cls * 4 + 1], sa = ct.sig[cls * 4 + 2], ss = ct.sig[cls * 4 + 3];
  const int pin = pin_enabled ? ct.pin[cls] : -1;
  const int lane = threadIdx.x % kWave;
  const int group = threadIdx.x / tpg, groups = kBlock / tpg, t = threadIdx.x % tpg;
  const int seg_base = blockIdx.y * (tpg * kCombineUnroll * WPL);
  const ClassRows cr = class_rows(pl, sr, st, sa, ss);

  u64 v[kCombineUnroll][WPL];
  int pc = 0;
#pragma unroll
  for (int u = 0; u < kCombineUnroll; ++u) {
#pragma unroll
    for (int j = 0; j < WPL; ++j) {
      int w = seg_base + (u * tpg + t) * WPL + j;
      u64 x = 0;
      if (w < row_words) {
        x = class_word(cr, w);
        if (pin == -2 || all_fail)
          x = 0;
        else if (pin >= 0)
          x &= (w == (pin >> 6)) ? (1ull << (pin & 63)) : 0ull;
      }
      v[u][j] = x;
      pc += __popcll(x);
    }
  }
  if (ct.chunk_first[chunk] && group == 0) {
    // feasible-node count of the class: wave reduce, one atomic per wave (thread group 0 holds the whole segment)
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off, kWave);
    if (lane == 0 && pc && class_count) atomicAdd(&class_count[cls], pc);  // (null: k_class_rows counted this class already)
  }
  // member rows: lane i of every wave holds the bitmap row of member i (one coalesced 256 B load), broadcast by v_readlane
  int mine = lane < len ? ct.members[begin + lane] : 0;
  for (int i = group; i < len; i += groups) {  // `group` is wave-uniform: tpg is a multiple of the wave size
    int p = __builtin_amdgcn_readlane(mine, i);
    if (p < 0) continue;  // unused entry (wave-uniform)
    u64* row = bitmap + (size_t)p * row_stride;
#pragma unroll
    for (int u = 0; u < kCombineUnroll; ++u) {
      int w = seg_base + (u * tpg + t) * WPL;
      if (w < row_stride) {  // row_stride is a multiple of 16 ⇒ a pair never straddles the end
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        *(u64x2*)(row + w) = u64x2{v[u][0], v[u][1]};
      }
    }
  }
}

// Wave-per-chunk form of k_combine for ask populations whose classes have few members (every ask its own template or its
// own request vector): a chunk is then a handful of rows, and what limits the block-per-chunk kernel is the number of
// independent (class → signature rows → plane words → store) chains in flight, not bandwidth. Here every WAVE owns a chunk —
// 4x as many chains per workgroup and no occupancy cap; lane = a pair of adjacent row words (dwordx4 loads and stores).
struct SliceDesc;
__device__ __forceinline__ bool slice_desc_general(const SliceDesc* desc, int chunk);
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(5, 8))) void k_combine_wave(ClassTable ct, Planes pl, u64* __restrict__ bitmap, int row_words, int row_stride,
                                                         int pin_enabled, int* __restrict__ class_count, int n_chunks,
                                                         const int* __restrict__ class_dirty /* null = every class */,
                                                         const SliceDesc* __restrict__ only_general /* non-null: only the chunks whose
                                                         descriptor says `general` (the rest belongs to k_walk_rows) */,
                                                         const int* __restrict__ n_general /* with only_general: their number */,
                                                         const int* __restrict__ chunk_list /* null: n_chunks = every chunk; else the
                                                         n_chunks chunks to run */) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const bool all_fail = pin_enabled & 2;
  pin_enabled &= 1;
  const int slot = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (slot >= n_chunks) return;
  const int chunk = chunk_list ? chunk_list[slot] : slot;
  if (only_general && (*n_general == 0 || !slice_desc_general(only_general, chunk))) return;
  const int lane = threadIdx.x % kWave;
  const int cls = ct.chunk_class[chunk];
  if (class_dirty ? !class_dirty[cls] : chunk_elsewhere(ct, chunk)) return;
  const int begin = ct.chunk_begin[chunk], len = ct.chunk_len[chunk];
  const int sr = ct.sig[cls * 4 + 0], st = ct.sig[cls * 4 + 1], sa = ct.sig[cls * 4 + 2], ss = ct.sig[cls * 4 + 3];
  const int pin = pin_enabled ? ct.pin[cls] : -1;
  const ClassRows cr = class_rows(pl, sr, st, sa, ss);
  const int mine = lane < len ? ct.members[begin + lane] : -1;
  int pc = 0;
  // the pair of words [w, w+1] of the class row (row_stride is a multiple of 16: a pair never straddles the end)
  auto class_pair = [&](int w) {
    u64x2 x = {0, 0};
    if (w < row_words && pin != -2 && !all_fail) {
      x = u64x2{~0ull, ~0ull};
#pragma unroll
      for (int i = 0; i < kMaxClassRows; ++i)
        if (i < cr.n) x &= *(const u64x2*)(cr.row[i] + w);
      if (cr.ni) {
        x.x &= class_idx_word(cr, w);
        if (w + 1 < row_words) x.y &= class_idx_word(cr, w + 1);
      }
      if (w + 1 >= row_words) x.y = 0;  // planes are zero there anyway (padding); keep the contract explicit
      if (pin >= 0) {
        x.x &= (w == (pin >> 6)) ? (1ull << (pin & 63)) : 0ull;
        x.y &= (w + 1 == (pin >> 6)) ? (1ull << (pin & 63)) : 0ull;
      }
    }
    return x;
  };
  // two-stage pipeline: the plane words of the next 1 KiB piece are in flight while this piece is stored
  u64x2 next = class_pair(2 * lane);
  for (int w0 = 0; w0 < row_stride; w0 += 2 * kWave) {
    const int w = w0 + 2 * lane;
    const u64x2 x = next;
    if (w0 + 2 * kWave < row_stride) next = class_pair(w + 2 * kWave);
    pc += __popcll(x.x) + __popcll(x.y);
    if (w < row_stride)
      for (int i = 0; i < len; ++i) {
        const int p = __builtin_amdgcn_readlane(mine, i);
        if (p >= 0) *(u64x2*)(bitmap + (size_t)p * row_stride + w) = x;
      }
  }
  if (ct.chunk_first[chunk]) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off, kWave);
    if (lane == 0 && pc) atomicAdd(&class_count[cls], pc);
  }
}

// Sum of an int over the 64 lanes of the wave, valid in LANE 63: the DPP row-shift / row-broadcast ladder (VALU only — six
// ds_bpermute round trips per reduction were a visible part of the slice writer's per-chunk latency in round 3).
__device__ __forceinline__ int wave_sum_lane63(int v) {
  int t = v + __builtin_amdgcn_update_dpp(0, v, 0x111, 0xf, 0xf, true);  // row_shr:1
  t += __builtin_amdgcn_update_dpp(0, v, 0x112, 0xf, 0xf, true);         // row_shr:2
  t += __builtin_amdgcn_update_dpp(0, v, 0x113, 0xf, 0xf, true);         // row_shr:3
  t += __builtin_amdgcn_update_dpp(0, t, 0x114, 0xf, 0xe, true);         // row_shr:4, lanes 4..15 of every row
  t += __builtin_amdgcn_update_dpp(0, t, 0x118, 0xf, 0xc, true);         // row_shr:8, lanes 8..15: lane 15 holds the row sum
  t += __builtin_amdgcn_update_dpp(0, t, 0x142, 0xa, 0xf, true);         // row_bcast:15 into rows 1 and 3
  t += __builtin_amdgcn_update_dpp(0, t, 0x143, 0xc, 0xf, true);         // row_bcast:31 into rows 2 and 3
  return t;
}
// Populations with INDEX rows (every ask its own request value: 10^6 single-member classes). Decoding an index byte through the
// word's 65-entry mask table is a gather with a 520-byte lane stride: in k_combine_wave (lane = word) every lane of a wave load
// hits its own cache line, ≈ 100 KB of L2 → L1 line traffic per 6 KB row written, and that — not HBM — sets the pace (4.5 ms alone
// for a 6.27 GB bitmap). Round 3 kept the tables of a <= 128-word slice of the row in LDS (k_combine_slices, 3.15 ms; its history
// is in profiles/r03_sessions7_20_small_class_paths.txt); round 4 decodes with rank planes instead: k_walk_rows below.
typedef u64 u64x2_t __attribute__((ext_vector_type(2)));

// Chunk descriptors of k_walk_rows: what a wave needs to know about a chunk, resolved ONCE per pass by one thread per chunk
// (k_slice_desc) instead of by the writer — the walk chunk → class → signatures → request-value rows is three levels of dependent,
// uncoalesced loads from tables that do not fit the L2 (27 M random 64-byte reads per pass when the slice writer of round 3 did it
// per (chunk, slice): profiles/r03_session11_pmc.txt); the writer reads them as wave-uniform scalar loads.
// A class's rows on the fast path: toleration, affinity, spread and the FIRST request-value plane row (row 0 of the family, the
// pod-independent part — the same row for every class) are "cached" rows, folded into the lane's base words while they do not
// change from chunk to chunk; what remains per chunk is at most one more plane row and exactly one index row.
struct SliceDesc {   // 48 bytes = three 16-byte loads
  int cls, meta, pin, mem0;   // meta: len | first << 8 | general path << 9 | per-chunk plane row << 10 | live << 15
  int st, sa, ss, p0;         // signature rows + the cached plane row (-1: none)
  int prow, irow, begin, slot; // per-chunk plane row, index row (walked dimension << kRowBigShift | row), first member slot, the plane
                               // row's slot among the rows k_walk_rows stages in LDS (WalkStage::n: none — the all-ones row)
};
constexpr int kWalkMaxStage = 16;    // ballot rows of the request family k_walk_rows keeps in LDS
struct WalkStage {
  int n;
  int row[kWalkMaxStage];  // their row ids in the request family
};
constexpr int kSliceFirst = 1 << 8, kSliceGeneral = 1 << 9, kSlicePlane = 1 << 10, kSliceLive = 1 << 15;
__device__ __forceinline__ bool slice_desc_general(const SliceDesc* desc, int chunk) {
  return (desc[chunk].meta & (kSliceLive | kSliceGeneral)) == (kSliceLive | kSliceGeneral);
}
// n_general (zeroed by the caller): the number of live chunks left to k_combine_wave
__global__ __launch_bounds__(kBlock) void k_slice_desc(ClassTable ct, Planes pl, int n_chunks, const int* __restrict__ class_dirty, int pin_enabled,
                                                       SliceDesc* __restrict__ out, int* __restrict__ n_general, WalkStage stage) {
  const int chunk = blockIdx.x * kBlock + threadIdx.x;
  if (chunk >= n_chunks) return;
  SliceDesc d{};
  d.st = d.sa = d.ss = d.p0 = d.pin = d.mem0 = -1;
  d.irow = 1 << kRowBigShift;
  d.cls = ct.chunk_class[chunk];
  const bool act = class_dirty ? class_dirty[d.cls] != 0 : !chunk_elsewhere(ct, chunk);
  if (act) {
    const int len = ct.chunk_len[chunk];
    d.meta = kSliceLive | len | (ct.chunk_first[chunk] ? kSliceFirst : 0);
    const int4 sg = *(const int4*)(ct.sig + (size_t)d.cls * 4);
    d.st = sg.y, d.sa = sg.z, d.ss = sg.w;
    d.pin = (pin_enabled & 1) ? ct.pin[d.cls] : -1;
    d.begin = ct.chunk_begin[chunk];
    d.mem0 = ct.members[d.begin];
    int np = 0, ni = 0;
    if (pl.res && sg.x >= 0)
      for (int k = 0; k < pl.res_slots; ++k) {
        const int r = pl.res_rows[(size_t)sg.x * pl.res_slots + k];
        if (r < 0) continue;
        if (r >> kRowBigShift) {
          if (ni == 0) d.irow = r;
          ++ni;
        } else {
          if (np == 0) d.p0 = r;
          if (np == 1) d.prow = r;
          ++np;
        }
      }
    // the general path: several member rows, the tail chunk of a longer class (it adds no count), a freed member slot, a
    // pinned node (NodeName), a plane row that is not staged, or a shape the fast path has no code for — so the fast path needs no test for any of them
    d.slot = stage.n;
    if (np >= 2) {
      d.meta |= kSlicePlane;
      d.slot = -1;
      for (int k = 0; k < stage.n; ++k)
        if (stage.row[k] == d.prow) d.slot = k;
    }
    if (len != 1 || !(d.meta & kSliceFirst) || d.mem0 < 0 || d.pin != -1 || np > 2 || ni != 1 || d.slot < 0) d.meta |= kSliceGeneral;
    if (d.meta & kSliceGeneral) atomicAdd(n_general, 1);
  }
  out[chunk] = d;
}

// ---------------------------------------------------------------------------------------------------
// k_walk_rows: the zone-B writer of populations with INDEX rows (10^6 single-member classes). A wave writes whole row segments.
//
// History. Round 3's k_combine_slices and the first k_walk_rows of round 4 decoded an index byte with a lookup into the word's
// 65-entry mask table (520 bytes per word): a row's tables (408 KB) do not fit the LDS, so a workgroup owned a SLICE of <= 128
// words of the row — 7 (row, slice) pairs per row at 50 k nodes, ~134 instructions per pair, one workgroup per CU (157 KB of LDS),
// 2.8–3.2 ms for 6.27 GB whatever was tuned (profiles/r03_sessions7_20_*, profiles/r04_walk_rows.txt: latency at 3.5 waves per SIMD).
// The table is the wrong data structure for the LDS. mask(word, j) = { node : valid and rank(node) >= j } with rank = the node's
// position in the word's ascending free list: a COMPARE of 64 small numbers with j. Bit-sliced — plane k of a word holds bit k of
// r' = valid ? rank + 1 : 0 for its 64 nodes, mask = r' > j — the compare is a ripple of 7 majority steps on whole words
// (gt = maj(gt, r_k, ~j_k): ONE v_bitop3_b32 per step and 32-bit half on gfx950), and the "table" is 56 bytes per word instead of
// 520: a whole row of 50 k nodes is 44 KB. So:
//   * a workgroup stages, for a segment of NIT x 64 words of the row, the rank planes of the walked dimensions ([dimension][7][words])
//     and the ballot rows of the request family (the few-valued dimensions' planes, <= kWalkMaxStage rows + an all-ones row) —
//     everything a row ANDs per word except its own index bytes; read with lane = word: conflict-free ds_read_b64;
//   * a WAVE owns a run of consecutive chunks and writes each row's segment whole, lane = word, NIT words per lane in registers: the
//     AND of the cached rows (toleration / affinity / spread / pod-independent request row) stays in registers while the signature
//     does not change (zone-B chunks come in signature order); per row the only global loads are NIT index bytes per lane;
//   * those are prefetched a group of 3–4 rows ahead (two register banks, the loop body unrolled twice): under the store stream
//     of this kernel a load takes several microseconds, and with one row of look-ahead a wave did one row per round trip
//     (second form: 2.99 ms). Loads are never predicated and never sit under a data-dependent branch: a conditional load into a
//     register the compiler also writes costs an s_waitcnt vmcnt(0) in front of it (first form: 5.4 ms);
//   * chunk descriptors (k_slice_desc) are wave-uniform scalar loads from the constant address space: they stay out of the
//     in-order vmcnt.
// Lanes past the row read inside the buffers (index rows are padded to 64 words with the empty position, the plane buffers carry
// 64 words of slack) and their words are masked by `base`; an absent cached row (no toleration / affinity / spread signature) is
// read as row 0 of the request family, which is part of every class anyway. Chunks without this fast path (several member rows,
// a pinned node, two index rows, more than two plane rows, a plane row that is not staged) are left to k_combine_wave through the
// descriptor filter, as before.
constexpr int kRowsWaves = 8;
constexpr int kRowsThreads = kRowsWaves * kWave;
constexpr int kWalkMaxIt = 7;         // words per lane and segment (register budget): wider rows are written segment by segment (grid.y)
constexpr int kWalkChunksPerWave = 64;
// gfx950's three-operand bit operation (truth table in the immediate: a = 0xF0, b = 0xCC, c = 0xAA)
__device__ __forceinline__ unsigned maj32(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0xE8); }
__device__ __forceinline__ unsigned and3(unsigned a, unsigned b, unsigned c) { return __builtin_amdgcn_bitop3_b32(a, b, c, 0x80); }
__host__ __device__ inline size_t walk_lds_bytes(int n_big, int n_stage, int nit) {
  return (size_t)(n_big * kRankBits + n_stage + 1) * (size_t)nit * kWave * sizeof(u64);
}
// NIT: words per lane (the segment is NIT x 64 words, every one of them inside the row rounded up to 64 words); TAIL: the segment
// holds the row's last word group, whose lanes past row_stride must not store.
template <int NIT, bool TAIL>
__global__ __launch_bounds__(kRowsThreads) __attribute__((amdgpu_waves_per_eu(4))) void k_walk_rows(
    Planes pl, const SliceDesc* __restrict__ desc, u64* __restrict__ bitmap, int row_words, int row_stride, int pin_enabled,
    int* __restrict__ class_count, int n_chunks, int w_base, WalkStage stage) {
  extern __shared__ u64 walk_lds[];  // [n_big][kRankBits][SW] rank planes, then [stage.n + 1][SW] ballot rows (the last one all ones)
  constexpr int SW = NIT * kWave;
  const bool all_fail = pin_enabled & 2;
  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int w_first = w_base + blockIdx.y * SW;
  u64* s_stage = walk_lds + (size_t)pl.n_big * kRankBits * SW;
  for (int i = (int)threadIdx.x; i < pl.n_big * kRankBits * SW; i += kRowsThreads) {
    const int plane = i / SW, w = w_first + i % SW;
    walk_lds[i] = w < pl.n_words ? pl.rbits[(size_t)plane * pl.n_words + w] : 0ull;
  }
  for (int i = (int)threadIdx.x; i < (stage.n + 1) * SW; i += kRowsThreads) {
    const int k = i / SW, w = w_first + i % SW;
    u64 v = ~0ull;
    if (k < stage.n) {
      int row = stage.row[0];
#pragma unroll
      for (int q = 1; q < kWalkMaxStage; ++q) row = k == q ? stage.row[q] : row;
      v = w < row_words ? pl.res[(size_t)row * pl.stride + w] : 0ull;
    }
    s_stage[i] = v;
  }
  __syncthreads();
  const bool tail_store = w_first + (NIT - 1) * kWave + lane < row_stride;  // (TAIL: this lane's last word lies inside the row)
  const int c0 = ((int)blockIdx.x * kRowsWaves + wave) * kWalkChunksPerWave, c1 = min(c0 + kWalkChunksPerWave, n_chunks);
  struct Row {
    int cls, dest, st, sa, ss, slot, irow;
  };
  // the next chunk of the wave's range that is on the fast path (wave-uniform: scalar loads); false: none left, r unchanged
  int c = c0;
  auto fetch = [&](Row& r) -> bool {
    for (; c < c1; ++c) {
      typedef const SliceDesc __attribute__((address_space(4))) ConstDesc;
      const ConstDesc* d = (const ConstDesc*)(unsigned long long)(desc + c);
      const int meta = d->meta;
      if ((meta & (kSliceLive | kSliceGeneral)) != kSliceLive) continue;
      r.cls = d->cls, r.dest = d->mem0;
      r.st = d->st, r.sa = d->sa, r.ss = d->ss;
      r.slot = d->slot;
      r.irow = d->irow;
      ++c;
      return true;
    }
    return false;
  };
  auto issue = [&](const Row& r, unsigned (&idx)[NIT]) {
    const unsigned char* isrc = pl.res_idx + (size_t)(r.irow & ((1 << kRowBigShift) - 1)) * pl.idx_stride + w_first + lane;
#pragma unroll
    for (int it = 0; it < NIT; ++it) idx[it] = isrc[it * kWave];
  };
  int cst = -2, csa = -2, css = -2;  // the signature rows folded into `base`
  u64 base[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) base[it] = 0;
  auto write_row = [&](const Row& r, const unsigned (&idx)[NIT]) {
    if (r.st != cst || r.sa != csa || r.ss != css) {  // (wave-uniform; rare: chunks come in signature order)
      cst = r.st, csa = r.sa, css = r.ss;
      int rw = all_fail ? 0 : row_words;
      asm volatile("" : "+s"(rw));  // (keeps the masks below out of the loop-invariant registers)
      const u64* r0 = pl.res + w_first + lane;  // row 0 of the request family: the pod-independent part, in every class
      const u64* rt = (pl.tol && cst >= 0) ? pl.tol + (size_t)cst * pl.stride + w_first + lane : r0;
      const u64* ra_ = (pl.aff && csa >= 0) ? pl.aff + (size_t)csa * pl.stride + w_first + lane : r0;
      const u64* rs = (pl.spread && css >= 0) ? pl.spread + (size_t)css * pl.stride + w_first + lane : r0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const u64 v = r0[it * kWave] & rt[it * kWave] & ra_[it * kWave] & rs[it * kWave];
        base[it] = w_first + it * kWave + lane < rw ? v : 0ull;
        if (it % 2 == 1) __builtin_amdgcn_sched_barrier(0);  // (register budget: two words' loads in flight)
      }
    }
    const u64* planes = walk_lds + (size_t)max((r.irow >> kRowBigShift) - 1, 0) * kRankBits * SW + lane;
    const u64* srow = s_stage + r.slot * SW + lane;
    u64* dst = bitmap + (size_t)r.dest * row_stride + w_first + lane;
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned nj = ~idx[it];
      u64 rk[kRankBits];
#pragma unroll
      for (int k = 0; k < kRankBits; ++k) rk[k] = planes[k * SW + it * kWave];
      const u64 pw = srow[it * kWave];
      unsigned m = (unsigned)((int)(nj << 31) >> 31);
      unsigned glo = (unsigned)rk[0] & m, ghi = (unsigned)(rk[0] >> 32) & m;
#pragma unroll
      for (int k = 1; k < kRankBits; ++k) {
        m = (unsigned)((int)(nj << (31 - k)) >> 31);
        glo = maj32(glo, (unsigned)rk[k], m);
        ghi = maj32(ghi, (unsigned)(rk[k] >> 32), m);
      }
      const unsigned xlo = and3((unsigned)base[it], (unsigned)pw, glo), xhi = and3((unsigned)(base[it] >> 32), (unsigned)(pw >> 32), ghi);
      cnt += __popc(xlo) + __popc(xhi);
      const u64 x = (u64)xlo | ((u64)xhi << 32);
      if (TAIL && it == NIT - 1) {
        if (tail_store) dst[it * kWave] = x;
      } else {
        dst[it * kWave] = x;
      }
      if (it % 2 == 1) __builtin_amdgcn_sched_barrier(0);  // two words' LDS reads in flight, not all of them (register budget)
    }
    const int pc = wave_sum_lane63(cnt);
    if (lane == 63 && pc) atomicAdd(&class_count[r.cls], pc);
  };
  // Two register banks of D rows each, the loop body unrolled twice: the index bytes of the next group are issued, then the
  // current group is written. The loads are unconditional — when the wave's range is exhausted a slot repeats the row before it
  // (`live` off) — so the compiler's vmcnt bookkeeping never has to assume a path without them (a branch around them cost an
  // s_waitcnt for the stores of the row before), and no register set travels around the loop through copies (a rotation of four
  // single-row sets did: every copy of a value still in flight is a wait).
  constexpr int D = NIT <= 5 ? 4 : 3;  // rows per bank (register budget: 2 x D x NIT index registers beside base, planes and temporaries)
  Row ra[D], rb[D];
  bool la[D], lb[D];
  unsigned ia[D][NIT], ib[D][NIT];
  auto fetch_group = [&](Row (&r)[D], bool (&l)[D], const Row& before) {
#pragma unroll
    for (int k = 0; k < D; ++k) {
      r[k] = k ? r[k - 1] : before;
      l[k] = fetch(r[k]);
    }
  };
  auto issue_group = [&](const Row (&r)[D], unsigned (&idx)[D][NIT]) {
#pragma unroll
    for (int k = 0; k < D; ++k) issue(r[k], idx[k]);
  };
  auto write_group = [&](const Row (&r)[D], const bool (&l)[D], const unsigned (&idx)[D][NIT]) {
#pragma unroll
    for (int k = 0; k < D; ++k)
      if (l[k]) write_row(r[k], idx[k]);
  };
  fetch_group(ra, la, Row{0, 0, -1, -1, -1, stage.n, 1 << kRowBigShift});
  if (!la[0]) return;
  issue_group(ra, ia);
  for (;;) {
    fetch_group(rb, lb, ra[D - 1]);
    issue_group(rb, ib);
    write_group(ra, la, ia);
    if (!lb[0]) break;
    fetch_group(ra, la, rb[D - 1]);
    issue_group(ra, ia);
    write_group(rb, lb, ib);
    if (!la[0]) break;
  }
}

// ---------------------------------------------------------------------------------------------------
// k_sweep_rows: the zone-B writer of RUNS of index-row classes (round 6). What a row is: predicate_manager.go:260-283 for one
// ask against every node — here for asks that differ from their neighbours ONLY in the value of one walked dimension.
//
// k_walk_rows decodes every (row, word) from scratch: 8 LDS reads and 16 bit operations per 8 bytes written, although two rows of
// one signature differ only in a threshold. Here the zone-B classes of one (toleration, affinity, spread, staged plane row) are laid
// out in ascending order of their walked value (build_classes; row_of_pod is a free permutation) — a RUN — and a lane (= word)
// keeps the word's current mask in registers: going from value v to the next larger one only CLEARS the bits of the nodes whose
// free value lies in [v, v'), found with a cursor into the word's ascending free list. Nothing is decoded per row, no index row
// is read (k_dim_walk need not write it), and the kernel is what it should be: a store stream with an occasional LDS lookup.
//   * k_dim_sort leaves, per (walked dimension, word), the list in the form the cursor wants: entry j = (g << 6) | node with
//     g = the number of the dimension's sorted request values that are <= the j-th smallest free value of the word. A row at
//     position `vi` of the sorted values fails node j  <=>  value > free  <=>  vi >= g  <=>  entry <= (vi << 6 | 63): one unsigned
//     compare per word and row; entry 64 is a sentinel (0xffffffff) that no row reaches.
//   * a workgroup stages the lists of its segment of NIT x 64 words ([65][SW] dwords, position-major: lanes are consecutive words,
//     so every cursor read is conflict-free whatever the positions) and the staged ballot rows of the request family;
//   * a wave owns a contiguous range of the sweep rows. At the start of a run (or of its range) a lane finds its position by
//     binary search in LDS and takes mask = (AND of the run's plane rows) & pmask[word][position] from the mask table of k_dim_sort;
//     from then on: compare, (rarely) advance the cursor and clear a bit, store. The feasible count of a row changes only when a
//     bit was cleared in this segment; the popcount reduction runs only then.
//   * row descriptors {class, bitmap row, position, run} are wave-uniform scalar loads (constant address space): the vector memory
//     counter holds nothing but stores in the steady state, and nothing ever waits for them.
struct SweepRow {   // 16 bytes
  int cls, dest, vi, run;   // class (its feasible count), bitmap row, position of the value in the dimension's sorted order, run
};
struct SweepRun {   // 16 bytes: one scalar load
  int st, sa, ss, prow;   // signature rows (-1: none) and the second ballot row of the request family (0: none — row 0 is in every class)
};
constexpr int kSweepWaves = 8;    // one workgroup per compute unit (the lists of a segment fill the LDS): two waves per SIMD, 256 VGPRs each
constexpr int kSweepThreads = kSweepWaves * kWave;
constexpr int kSweepBatch = 16;   // row descriptors a wave keeps in LDS at a time
constexpr int kSweepRunCost = 4;  // what a run start costs a wave, in rows of a long run (a drain + a load round trip against 3.5 KB of stores)
constexpr int kSweepMaxSegs = 16;
__host__ __device__ inline size_t sweep_lds_bytes(int nit) {
  return (size_t)65 * nit * kWave * sizeof(unsigned) + (size_t)kRankBits * nit * kWave * sizeof(u64) + (size_t)kSweepWaves * kSweepBatch * 16;
}
// One launch per walked dimension: grid.x persistent workgroups (one per compute unit: the lists of a segment fill the LDS), grid.y =
// the row's segments — `n_long` segments of NIT word groups, then segments of NIT - 1, the last one holding the row's tail.
template <int NIT>
__global__ __launch_bounds__(kSweepThreads) void k_sweep_rows(
    Planes pl, const unsigned* __restrict__ ent /* [65][n_words] of this walked dimension */, const u64* __restrict__ rbits /* [kRankBits][n_words] */,
    const SweepRow* __restrict__ rows, const SweepRun* __restrict__ runs, int n_rows, const int* __restrict__ units /* [n_units + 1] */,
    int n_units, u64* __restrict__ bitmap,
    int row_words, int row_stride, int pin_enabled, int* __restrict__ class_count, int n_long) {
  extern __shared__ u64 sweep_lds[];  // [65][SW] cursor lists, [kRankBits][SW] rank planes, [waves][kSweepBatch] row descriptors
  constexpr int SW = NIT * kWave;
  const bool all_fail = pin_enabled & 2;
  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int seg = (int)blockIdx.y;
  const int nit = seg < n_long ? NIT : NIT - 1;  // word groups of this segment (wave-uniform)
  const int w_first = (seg < n_long ? seg * NIT : n_long * NIT + (seg - n_long) * (NIT - 1)) * kWave;
  const bool last_seg = seg == (int)gridDim.y - 1;
  unsigned* s_ent = (unsigned*)sweep_lds;
  u64* s_rk = (u64*)(s_ent + 65 * SW);
  int4* s_desc = (int4*)(s_rk + kRankBits * SW) + wave * kSweepBatch;
  for (int i = (int)threadIdx.x; i < 65 * SW; i += kSweepThreads) {
    const int j = i / SW, w = w_first + i % SW;
    s_ent[i] = (j < 64 && i % SW < nit * kWave && w < pl.n_words) ? ent[(size_t)j * pl.n_words + w] : 0xffffffffu;
  }
  // the rank planes of the segment (k_dim_sort: bit k of r' = valid ? position in the word's ascending free list + 1 : 0): the word of
  // a row that starts at position j is { node : r' > j } — seven majority steps on LDS words. (The 65-entry mask table gives the
  // same word with one load — a gather with a 520-byte lane stride: 448 cache lines per run start, which is what a run start cost.)
  for (int i = (int)threadIdx.x; i < kRankBits * SW; i += kSweepThreads) {
    const int k = i / SW, w = w_first + i % SW;
    s_rk[i] = w < pl.n_words ? rbits[(size_t)k * pl.n_words + w] : 0ull;
  }
  __syncthreads();
  typedef const SweepRun __attribute__((address_space(4))) ConstRun;
  // Row descriptors reach the loop through LDS, a batch at a time (lane i fetches row rb + i; the wave then reads a row's 16 bytes with
  // a uniform address, one row ahead), the next batch's load issued a batch ahead. The vector memory counter of the steady state holds
  // nothing but stores: gfx9 counts loads and stores in ONE in-order counter, so a wait for any load is a wait for the stores in
  // front of it — a descriptor that arrived by a load per row put such a wait in front of every row (first forms: 2.8 TB/s).
  // The rows are dealt in UNITS of equal estimated cost (build_classes: a row = one, a run start = kSweepRunCost rows — a binary
  // search in LDS, one round of global loads and the wait for them, which drains the wave's stores), round robin over all waves of the
  // segment. Inside a unit a wave walks consecutive rows: a run is started once per unit, not once per batch, and every wave gets
  // the same share of the short runs (the asks with a selector of their own, at the end of the list). What did not work: a contiguous
  // split by row count (most of the chip waiting for the workgroups that drew the short runs), batches of rows round robin (a run
  // start per batch in every run below 32 k rows), batches claimed from a counter (a returning atomic per batch, whose wait is a drain
  // as well: 3.2 TB/s on the long runs against 4.8). profiles/r06_rowstore_probe.txt: the store pattern itself reaches 5.4-5.6 TB/s
  // however the rows are dealt.
  const int4* vrow = (const int4*)rows;
  typedef const int __attribute__((address_space(4))) ConstInt;
  const ConstInt* ub = (const ConstInt*)(unsigned long long)units;
  const int unit_step = (int)gridDim.x * kSweepWaves;
  int u = (int)blockIdx.x * kSweepWaves + wave;
  if (u >= n_units) return;
  auto fetch = [&](int rb) { return vrow[min(rb + min(lane, kSweepBatch - 1), n_rows - 1)]; };
  int rb = ub[u], unit_end = ub[u + 1];
  int4 d_nx = fetch(rb);
  if (lane < kSweepBatch) s_desc[lane] = d_nx;
  int r = rb, r1 = min(rb + kSweepBatch, unit_end);
  d_nx = fetch(r1);
  int4 dd = s_desc[0];
  u64 mask[NIT];
  unsigned cur[NIT];   // the list entry under the cursor
  int at[NIT];         // its dword index in s_ent
  const int rw = all_fail ? 0 : row_words;
  // row 0 of the request family — the pod-independent part, in every class — stays in registers, with the words past the row (and
  // past this segment's word groups) cleared: every row is ANDed with it
  u64 row0[NIT];
#pragma unroll
  for (int it = 0; it < NIT; ++it) {
    const u64 v = pl.res[w_first + it * kWave + lane];
    row0[it] = (w_first + it * kWave + lane < rw && it < nit) ? v : 0ull;
  }
  // the first list entry a row of threshold T does not reach, per word group: entries [0, lo) are <= T. The NIT searches step together.
  auto positions = [&](unsigned T, int (&lo)[NIT]) {
    int hi[NIT];
#pragma unroll
    for (int it = 0; it < NIT; ++it) lo[it] = 0, hi[it] = 64;
#pragma unroll
    for (int k = 0; k < 7; ++k) {
      unsigned e[NIT];
#pragma unroll
      for (int it = 0; it < NIT; ++it) e[it] = s_ent[((lo[it] + hi[it]) >> 1) * SW + it * kWave + lane];  // (lo == hi == 64 reads the sentinel)
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const int mid = (lo[it] + hi[it]) >> 1;
        const bool below = e[it] <= T;
        lo[it] = below ? mid + 1 : lo[it];
        hi[it] = below ? hi[it] : mid;
      }
    }
  };
  // the words of a run's row at the positions `lo`: the AND of its plane rows and { node : rank >= position } — every global load
  // of a run start in one round (a wait for a load is a wait for every store this wave has in flight: one such wait per run start)
  auto run_words = [&](int run, const int (&lo)[NIT], u64 (&pm)[NIT]) {
    const ConstRun* cr = (const ConstRun*)(unsigned long long)(runs + run);
    const int st = cr->st, sa = cr->sa, ss = cr->ss, prow = cr->prow;
    const u64* p0 = pl.res + w_first + lane;  // (an absent signature row reads row 0 again)
    const u64* pp = pl.res + (size_t)prow * pl.stride + w_first + lane;
    const u64* pt = (pl.tol && st >= 0) ? pl.tol + (size_t)st * pl.stride + w_first + lane : p0;
    const u64* pa = (pl.aff && sa >= 0) ? pl.aff + (size_t)sa * pl.stride + w_first + lane : p0;
    const u64* ps = (pl.spread && ss >= 0) ? pl.spread + (size_t)ss * pl.stride + w_first + lane : p0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      const unsigned nj = ~(unsigned)lo[it];
      u64 rk[kRankBits];
#pragma unroll
      for (int k = 0; k < kRankBits; ++k) rk[k] = s_rk[k * SW + it * kWave + lane];
      unsigned m = (unsigned)((int)(nj << 31) >> 31);
      unsigned glo = (unsigned)rk[0] & m, ghi = (unsigned)(rk[0] >> 32) & m;
#pragma unroll
      for (int k = 1; k < kRankBits; ++k) {
        m = (unsigned)((int)(nj << (31 - k)) >> 31);
        glo = maj32(glo, (unsigned)rk[k], m);
        ghi = maj32(ghi, (unsigned)(rk[k] >> 32), m);
      }
      pm[it] = (u64)glo | ((u64)ghi << 32);
    }
#pragma unroll
    for (int it = 0; it < NIT; ++it) pm[it] &= row0[it] & pp[it * kWave];
#pragma unroll
    for (int it = 0; it < NIT; ++it) pm[it] &= pt[it * kWave] & pa[it * kWave] & ps[it * kWave];
  };
  auto popcount = [&](const u64 (&m)[NIT]) {
    int cnt = 0;
#pragma unroll
    for (int it = 0; it < NIT; ++it) cnt += __popcll(m[it]);
    return __builtin_amdgcn_readlane(wave_sum_lane63(cnt), 63);
  };
  auto emit = [&](int cls, int dest, const u64 (&m)[NIT], int pc) {
    u64* dst = bitmap + (size_t)dest * row_stride + w_first + lane;
#pragma unroll
    for (int it = 0; it < NIT; ++it) {
      if (it < NIT - 2) {
        dst[it * kWave] = m[it];
      } else if (it < nit) {  // (wave-uniform: the segment's last one or two word groups)
        if (!last_seg || it < nit - 1 || w_first + it * kWave + lane < row_stride) dst[it * kWave] = m[it];
      }
    }
    if (lane == 63 && pc) atomicAdd(&class_count[cls], pc);
  };
  for (;;) {
    // ---- a run begins — or this wave's first row of one: positions by binary search, masks from the planes and the mask table.
    // (The rows a wave takes ascend: inside a run its cursors only ever move forward.)
    const int run = __builtin_amdgcn_readfirstlane(dd.w);
    int pc;
    {
      int lo[NIT];
      positions(((unsigned)__builtin_amdgcn_readfirstlane(dd.z) << 6) | 63u, lo);
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        at[it] = lo[it] * SW + it * kWave + lane;
        cur[it] = s_ent[at[it]];
      }
      run_words(run, lo, mask);
      pc = popcount(mask);
    }
    // ---- the rows of the run that are this wave's
    for (;;) {
      const int cls = __builtin_amdgcn_readfirstlane(dd.x), dest = __builtin_amdgcn_readfirstlane(dd.y);
      const unsigned T = ((unsigned)__builtin_amdgcn_readfirstlane(dd.z) << 6) | 63u;
      // Some word of the segment loses nodes between the previous value and this one: all cursors step together — the NIT list
      // reads of a step are in flight at once — and a cursor that has nothing to clear re-reads its entry (no branch per word: the
      // loop carries the state in place).
      bool stepped = false;
      for (;;) {
        bool any = false;
#pragma unroll
        for (int it = 0; it < NIT; ++it) any |= cur[it] <= T;
        if (!__builtin_amdgcn_ballot_w64(any)) break;
        stepped = true;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          const bool go = cur[it] <= T;
          const u64 bit = 1ull << (cur[it] & 63u);
          mask[it] &= go ? ~bit : ~0ull;
          at[it] += go ? SW : 0;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) cur[it] = s_ent[at[it]];
      }
      if (stepped) pc = popcount(mask);
      // the next row's descriptor: in flight while this row is stored (behind the cursor check: the loop above waits for every LDS
      // read that is outstanding when it is entered); a batch that is used up is replaced by the one fetched a batch ago
      ++r;
      bool more = true;
      if (r == r1) {
        if (r < unit_end) {  // the unit's next batch: its descriptors were fetched a batch ago
          rb = r;
          if (lane < kSweepBatch) s_desc[lane] = d_nx;
          r1 = min(rb + kSweepBatch, unit_end);
          d_nx = fetch(r1);
        } else {
          u += unit_step;
          if (u >= n_units) {
            more = false;
          } else {  // the wave's next unit (a handful per wave: this fetch is waited for at once)
            rb = ub[u], unit_end = ub[u + 1];
            d_nx = fetch(rb);
            if (lane < kSweepBatch) s_desc[lane] = d_nx;
            r = rb;
            r1 = min(rb + kSweepBatch, unit_end);
            d_nx = fetch(r1);
          }
        }
      }
      if (more) dd = s_desc[r - rb];
      emit(cls, dest, mask, pc);
      if (!more) return;
      if (__builtin_amdgcn_readfirstlane(dd.w) != run) break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_fused_rows: the zone-B writer of the classes NO run kernel takes (round 6) — a signature with a handful of rows, an ask with a
// node selector of its own: unpinned classes whose rows are plain plane rows (no index row, no topology signature). 77 k of them
// took k_combine_wave 0.34 ms on the own-template population (1.4 TB/s): a wave there resolves its class through three levels of
// dependent table loads and then alternates loads and stores, and gfx9 counts both in ONE in-order counter — every wait for a
// plane word is a wait for the kilobyte stored before it. Here the class is a RECORD the host resolved at class-build time (the
// plane rows to AND, the bitmap rows to write), and the roles are split the way the band writer splits them: waves 0..2 of a
// workgroup COMPUTE (they only ever load: a record's rows, WPL words per lane), wave 3 STORES (it only ever reads LDS): a class's
// words go through a double-buffered LDS row, one s_barrier per class with LDS-scope fences only — no vmcnt wait stands between a
// load and an earlier store anywhere.
//
// What the first form of it showed (profiles/r06_fused_rows_experiments.txt): with five rows per record — affinity, toleration,
// three request-value rows — the kernel was no faster than k_combine_wave, and its time was the SUM of three parts: 0.12 ms the
// affinity row + the per-record step (LDS, barrier), 0.11 ms the four shared rows (L2 hits, but every word of them through the
// compute unit's texture path: 31 KB per 6 KB written), 0.06 ms the stores; prefetching the affinity row further ahead changed
// nothing. So the shared rows are COMBINED first: classes that differ only in their node-affinity signature share the AND of their
// other rows (toleration & request values: as many combinations as there are distinct (toleration, request vector) pairs among
// these classes), the same kernel writes those combination rows into a small table (MAXR 8, a launch of a few hundred records), and
// a class is affinity row & combination row (MAXR 2): two loads per 6 KB written, half the registers, twice the workgroups per CU.
struct FuseRec {
  int cls, dest, len, n;  // class (count slot), its first output row, member rows (consecutive), plane rows to AND
  int row[8];             // family << kFuseFamShift | row of the family: 0 = request values, 1 = toleration, 2 = node affinity (first),
                          // 3 = the combination table
};
constexpr int kFuseFamShift = 28;
constexpr int kFuseComputeWaves = 3;
constexpr int kFuseRecInts = sizeof(FuseRec) / sizeof(int);
constexpr int kFuseRecsPerBlock = 16;  // 192 ints: three registers of a wave hold the block's records
__device__ __forceinline__ void wg_barrier_lds() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup", "local");
  __builtin_amdgcn_s_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup", "local");
}
// int j (wave-uniform) of the block's records, held by lanes of three registers
__device__ __forceinline__ int fuse_rec_field(int l0, int l1, int l2, int j) {
  const int v = j < kWave ? l0 : (j < 2 * kWave ? l1 : l2);
  return __builtin_amdgcn_readlane(v, j & (kWave - 1));
}
struct FuseSrc {
  const u64* planes;  // the canonical plane buffer
  const u64* combos;  // the combination table (family 3), or null
  int base[3];        // first rows of the request-value, toleration and node-affinity families in `planes`
};
__device__ __forceinline__ const u64* fuse_row(const FuseSrc& src, int enc, int stride) {
  const int fam = enc >> kFuseFamShift, id = enc & ((1 << kFuseFamShift) - 1);
  const int b0 = src.base[0], b1 = src.base[1], b2 = src.base[2];
  const u64* p = fam == 3 ? src.combos : src.planes;
  const int base = fam == 0 ? b0 : (fam == 1 ? b1 : (fam == 2 ? b2 : 0));
  return p + (size_t)(id + base) * stride;
}
// grid.x = blocks of kFuseRecsPerBlock records, grid.y = groups of 3 * 64 * WPL row words; MAXR = most rows a record of the launch has
template <int WPL, int MAXR>
__global__ __launch_bounds__(kBlock) void k_fused_rows(const FuseRec* __restrict__ recs, int n_recs, FuseSrc src, int n_words, int stride,
                                                       u64* __restrict__ out, int* __restrict__ class_count /* null: not counted */) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  constexpr int GW = kFuseComputeWaves * kWave * WPL;  // words of a group (a multiple of 16)
  __shared__ __attribute__((aligned(16))) u64 buf[2][GW];
  __shared__ __attribute__((aligned(16))) int hdr[2][4];
  const int lane = threadIdx.x % kWave, wave = threadIdx.x / kWave;
  const int gbase = blockIdx.y * GW;
  const int r0 = blockIdx.x * kFuseRecsPerBlock;
  const int nr = min(kFuseRecsPerBlock, n_recs - r0);
  if (nr <= 0 || gbase >= n_words) return;  // (workgroup-uniform)
  if (wave == kFuseComputeWaves) {
    // ---- the store wave: lane l owns the word pairs l, l + 64, .. of the group
    constexpr int NP = (GW / 2 + kWave - 1) / kWave;
    for (int q = 0; q < nr; ++q) {
      wg_barrier_lds();
      const int b = q & 1;
      const int dest = hdr[b][0], len = hdr[b][1], cls = hdr[b][2];
      u64x2 v[NP];
      int pc = 0;
#pragma unroll
      for (int k = 0; k < NP; ++k) {
        const int p = k * kWave + lane;
        v[k] = p < GW / 2 ? *(const u64x2*)&buf[b][2 * p] : u64x2{0, 0};
        pc += __popcll(v[k].x) + __popcll(v[k].y);
      }
      u64* dst = out + (size_t)dest * stride + gbase;
      for (int r = 0; r < len; ++r, dst += stride) {
#pragma unroll
        for (int k = 0; k < NP; ++k) {
          const int p = k * kWave + lane;
          if (p < GW / 2 && gbase + 2 * p < stride) *(u64x2*)(dst + 2 * p) = v[k];  // (stride is a multiple of 16: a pair never straddles the end)
        }
      }
      if (class_count) {
        pc = wave_sum_lane63(pc);
        if (lane == kWave - 1 && pc) atomicAdd(&class_count[cls], pc);
      }
    }
    return;
  }
  // ---- compute waves: lane j of rec_l[r] = int (r * 64 + j) of the block's records
  int rec_l[3];
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int j = r * kWave + lane;
    rec_l[r] = j < nr * kFuseRecInts ? ((const int*)(recs + r0))[j] : 0;
  }
  const int rl0 = rec_l[0], rl1 = rec_l[1], rl2 = rec_l[2];
#define YK_REC_FIELD(q, k) fuse_rec_field(rl0, rl1, rl2, (q) * kFuseRecInts + (k))
#define YK_REC_ROW(q, k) fuse_row(src, YK_REC_FIELD(q, 4 + (k)), stride)
  const int lbase = wave * (kWave * WPL);
  int w[WPL];
  u64 keep[WPL];  // all ones for the words of the row, 0 past its end
#pragma unroll
  for (int j = 0; j < WPL; ++j) {
    const int w_raw = gbase + lbase + j * kWave + lane;
    w[j] = min(w_raw, n_words - 1);
    keep[j] = w_raw < n_words ? ~0ull : 0ull;
  }
  // A record's first row (the node-affinity plane: read once, from HBM) is fetched a record ahead; its other rows are all in flight at once.
  u64 head[WPL];
  {
    const u64* row = YK_REC_ROW(0, 0);
#pragma unroll
    for (int j = 0; j < WPL; ++j) head[j] = row[w[j]];
  }
  for (int q = 0; q < nr; ++q) {
    const int n = YK_REC_FIELD(q, 3);
    u64 x[WPL], r[MAXR - 1][WPL];
#pragma unroll
    for (int k = 1; k < MAXR; ++k)
      if (k < n) {  // (wave-uniform)
        const u64* row = YK_REC_ROW(q, k);
#pragma unroll
        for (int j = 0; j < WPL; ++j) r[k - 1][j] = row[w[j]];
      }
#pragma unroll
    for (int j = 0; j < WPL; ++j) x[j] = keep[j] & head[j];
    if (q + 1 < nr) {
      const u64* row = YK_REC_ROW(q + 1, 0);
#pragma unroll
      for (int j = 0; j < WPL; ++j) head[j] = row[w[j]];
    }
#pragma unroll
    for (int k = 1; k < MAXR; ++k)
      if (k < n) {
#pragma unroll
        for (int j = 0; j < WPL; ++j) x[j] &= r[k - 1][j];
      }
#pragma unroll
    for (int j = 0; j < WPL; ++j) buf[q & 1][lbase + j * kWave + lane] = x[j];
    if (threadIdx.x == 0) *(int4*)hdr[q & 1] = int4{YK_REC_FIELD(q, 1), YK_REC_FIELD(q, 2), YK_REC_FIELD(q, 0), 0};
    wg_barrier_lds();
  }
#undef YK_REC_ROW
#undef YK_REC_FIELD
}

// ---------------------------------------------------------------------------------------------------
// k_class_runs: the zone-B writer of RUNS of ballot-row classes (round 6) — k_sweep_rows' sibling for classes without an index row.
// The zone-B classes of one (toleration, affinity, spread) signature lie side by side (build_classes); they differ in nothing but
// their request-value rows, and those are few (a palette of cpu and memory values): the ballot rows of the request family are staged
// in LDS once per workgroup (WalkStage: at most kWalkMaxStage rows + an all-ones row), the AND of the signature's rows and row 0 is
// fetched once per RUN and stays in registers, and a class is `base & staged rows` — no global load — stored to its member rows
// (consecutive bitmap rows). k_combine_wave resolves every class on its own: three levels of dependent table loads, then four
// plane rows per class from L2, a wait in front of every kilobyte it stores.
struct RunClass {   // 32 bytes
  int cls, dest0, len, slots;   // class, its first bitmap row, member rows, four staged-row slots (a byte each; WalkStage::n: none)
  int run, pad0, pad1, pad2;
};
constexpr int kRunsWaves = 8;
constexpr int kRunsThreads = kRunsWaves * kWave;
constexpr int kRunsBatch = 16;    // class descriptors a wave keeps in LDS at a time
constexpr int kRunsClassCost = 1; // what a class costs beside its rows, in rows (unit balance)
__host__ __device__ inline size_t runs_lds_bytes(int n_stage, int nit) {
  return (size_t)(n_stage + 1) * nit * kWave * sizeof(u64) + (size_t)kRunsWaves * kRunsBatch * 32;
}
template <int NIT>
__global__ __launch_bounds__(kRunsThreads) void k_class_runs(
    Planes pl, const RunClass* __restrict__ classes, const SweepRun* __restrict__ runs, int n_classes, const int* __restrict__ units /* [n_units + 1] */,
    int n_units, u64* __restrict__ bitmap, int row_words, int row_stride, int pin_enabled, int* __restrict__ class_count, int n_long, WalkStage stage) {
  extern __shared__ u64 runs_lds[];  // [stage.n + 1][SW] ballot rows of the request family (the last one all ones), [waves][kRunsBatch] descriptors
  constexpr int SW = NIT * kWave;
  const bool all_fail = pin_enabled & 2;
  const int lane = threadIdx.x % kWave;
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x / kWave);
  const int seg = (int)blockIdx.y;
  const int nit = seg < n_long ? NIT : NIT - 1;
  const int w_first = (seg < n_long ? seg * NIT : n_long * NIT + (seg - n_long) * (NIT - 1)) * kWave;
  const bool last_seg = seg == (int)gridDim.y - 1;
  u64* s_stage = runs_lds;
  int4* s_desc = (int4*)(s_stage + (size_t)(stage.n + 1) * SW) + wave * kRunsBatch * 2;
  for (int i = (int)threadIdx.x; i < (stage.n + 1) * SW; i += kRunsThreads) {
    const int k = i / SW, w = w_first + i % SW;
    u64 v = ~0ull;
    if (k < stage.n) {
      int row = stage.row[0];
#pragma unroll
      for (int q = 1; q < kWalkMaxStage; ++q) row = k == q ? stage.row[q] : row;
      v = w < row_words ? pl.res[(size_t)row * pl.stride + w] : 0ull;
    }
    s_stage[i] = v;
  }
  __syncthreads();
  typedef const SweepRun __attribute__((address_space(4))) ConstRun;
  typedef const int __attribute__((address_space(4))) ConstInt;
  const ConstInt* ub = (const ConstInt*)(unsigned long long)units;
  const int4* vcls = (const int4*)classes;  // two int4 per class
  const int unit_step = (int)gridDim.x * kRunsWaves;
  int u = (int)blockIdx.x * kRunsWaves + wave;
  if (u >= n_units) return;
  // (descriptors reach the loop through LDS, a batch at a time, the next batch's load issued a batch ahead — see k_sweep_rows)
  auto fetch = [&](int cb) { return vcls[2 * (size_t)min(cb + (min(lane, 2 * kRunsBatch - 1) >> 1), n_classes - 1) + (lane & 1)]; };
  int cb = ub[u], unit_end = ub[u + 1];
  int4 d_nx = fetch(cb);
  if (lane < 2 * kRunsBatch) s_desc[lane] = d_nx;
  int c = cb, c1 = min(cb + kRunsBatch, unit_end);
  d_nx = fetch(c1);
  const int rw = all_fail ? 0 : row_words;
  int4 da = s_desc[0], db = s_desc[1];
  for (;;) {
    // ---- a run begins: the AND of row 0 and the signature's rows, once
    const int run = __builtin_amdgcn_readfirstlane(db.x);
    u64 base[NIT];
    {
      const ConstRun* cr = (const ConstRun*)(unsigned long long)(runs + run);
      const int st = cr->st, sa = cr->sa, ss = cr->ss;
      const u64* p0 = pl.res + w_first + lane;
      const u64* pt = (pl.tol && st >= 0) ? pl.tol + (size_t)st * pl.stride + w_first + lane : p0;
      const u64* pa = (pl.aff && sa >= 0) ? pl.aff + (size_t)sa * pl.stride + w_first + lane : p0;
      const u64* ps = (pl.spread && ss >= 0) ? pl.spread + (size_t)ss * pl.stride + w_first + lane : p0;
#pragma unroll
      for (int it = 0; it < NIT; ++it) {
        const u64 v = p0[it * kWave] & pt[it * kWave] & pa[it * kWave] & ps[it * kWave];
        base[it] = (w_first + it * kWave + lane < rw && it < nit) ? v : 0ull;
      }
    }
    // ---- the classes of the run that are this wave's
    for (;;) {
      const int cls = __builtin_amdgcn_readfirstlane(da.x), dest0 = __builtin_amdgcn_readfirstlane(da.y);
      const int len = __builtin_amdgcn_readfirstlane(da.z), slots = __builtin_amdgcn_readfirstlane(da.w);
      u64 m[NIT];
      int cnt = 0;
      {
        const u64* r0 = s_stage + (slots & 255) * SW + lane;
        const u64* r1 = s_stage + ((slots >> 8) & 255) * SW + lane;
        const u64* r2 = s_stage + ((slots >> 16) & 255) * SW + lane;
        const u64* r3 = s_stage + ((slots >> 24) & 255) * SW + lane;
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          m[it] = base[it] & r0[it * kWave] & r1[it * kWave] & r2[it * kWave] & r3[it * kWave];
          cnt += __popcll(m[it]);
        }
      }
      const int pc = __builtin_amdgcn_readlane(wave_sum_lane63(cnt), 63);
      // the next class's descriptor: in flight while this class is stored
      ++c;
      bool more = true;
      if (c == c1) {
        if (c < unit_end) {
          cb = c;
          if (lane < 2 * kRunsBatch) s_desc[lane] = d_nx;
          c1 = min(cb + kRunsBatch, unit_end);
          d_nx = fetch(c1);
        } else {
          u += unit_step;
          if (u >= n_units) {
            more = false;
          } else {
            cb = ub[u], unit_end = ub[u + 1];
            d_nx = fetch(cb);
            if (lane < 2 * kRunsBatch) s_desc[lane] = d_nx;
            c = cb;
            c1 = min(cb + kRunsBatch, unit_end);
            d_nx = fetch(c1);
          }
        }
      }
      if (more) da = s_desc[2 * (c - cb)], db = s_desc[2 * (c - cb) + 1];
      u64* dst = bitmap + (size_t)dest0 * row_stride + w_first + lane;
      for (int i = 0; i < len; ++i, dst += row_stride) {
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
          if (it < NIT - 2) {
            dst[it * kWave] = m[it];
          } else if (it < nit) {
            if (!last_seg || it < nit - 1 || w_first + it * kWave + lane < row_stride) dst[it * kWave] = m[it];
          }
        }
      }
      if (lane == 63 && pc) atomicAdd(&class_count[cls], pc);
      if (!more) return;
      if (__builtin_amdgcn_readfirstlane(db.x) != run) break;
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// zone A of the bitmap: written with the store pattern of a linear fill (scripts/fill_probe*.hip)
// ---------------------------------------------------------------------------------------------------
// Measured on MI355X: of all ways to write the 6.27 GB bitmap only ONE reaches the hipMemset rate (6.4 TB/s against 5.4 for
// anything row-shaped): 256 (or 128) workgroups, each storing one 4 KiB-aligned tile per step, all tiles of a step forming
// one contiguous 1 MiB window that advances. k_expand_bands writes exactly that pattern and sources the data from LDS:
//   * the physical rows are permuted (row_of_pod) so that inside a band of S windows a class owns the rows whose start
//     offset inside their window, X = (row * row_bytes) mod window, lies in one interval (engine.hip: build_classes);
//   * workgroup b always writes window bytes [4096 b, 4096 b + 4096), so for the S steps of a band it needs the rows of at
//     most kBandClasses classes: they sit in LDS; the next band's rows are fetched into registers mid-band and committed
//     to the second LDS buffer at the band boundary, so the steady state issues no global load at all;
//   * per step and thread: the column of its 16 bytes inside their row advances by (window mod row_bytes); the row started
//     in this window iff col <= p (p = the thread's offset in the window) and then X = p - col; X against the band's class
//     boundaries selects the LDS row; one ds_read_b128, one global_store_dwordx4.
// The head of every window belongs to a row that started in the previous window: k_fix_rows rewrites those rows (one per
// window) whole. Class rows come from k_class_rows (AND of the class's planes, popcount = the class's feasible count).
constexpr int kBandGroups = 256;   // workgroups = 4 KiB tiles per window (power of two: the HBM channel interleave)
constexpr int kBandClasses = 4;    // class rows a workgroup can hold per band
constexpr int kBandMaxLds = 150 * 1024;
struct BandEntry {  // what one workgroup needs during one band
  int slot[kBandClasses];    // class-row table slots, in X order (unused entries repeat the last one)
  int xb[kBandClasses - 1];  // first row of class i + 1: its X ...
  int sb[kBandClasses - 1];  // ... and step; INT_MAX X = no further class
  int steps, first_step;     // height of the band, its first window
  int n, pad;                // classes this workgroup really needs (1..kBandClasses): selects the loop variant
};

// one wave per zone-A class: class row = AND of its planes (+ NodeName pin), written to the class-row table
// (class_list_b / n_b: further classes whose feasible COUNT is wanted here and whose row is not — the zone-B classes of a pass
// whose zone B is small: every class count is then known before a single bitmap row is written, see ykpred_eval)
__global__ __launch_bounds__(kBlock) void k_class_rows(ClassTable ct, Planes pl, const int* __restrict__ class_list, int n_list, int row_words,
                                                       int row_stride, int pin_enabled, u64* __restrict__ row_table,
                                                       int* __restrict__ class_count, const int* __restrict__ class_dirty,
                                                       const int* __restrict__ class_list_b, int n_b) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const bool all_fail = pin_enabled & 2;
  pin_enabled &= 1;
  const int k = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (k >= n_list + n_b) return;
  const bool store = k < n_list;
  const int cls = store ? class_list[k] : class_list_b[k - n_list];
  if (class_dirty && !class_dirty[cls]) return;
  const int lane = threadIdx.x % kWave;
  const int sr = ct.sig[cls * 4 + 0], st = ct.sig[cls * 4 + 1], sa = ct.sig[cls * 4 + 2], ss = ct.sig[cls * 4 + 3];
  const int pin = pin_enabled ? ct.pin[cls] : -1;
  const ClassRows cr = class_rows(pl, sr, st, sa, ss);
  u64* dst = row_table + (size_t)k * row_stride;
  int pc = 0;
  for (int w = 2 * lane; w < row_stride; w += 2 * kWave) {
    u64x2 x = {0, 0};
    if (w < row_words && pin != -2 && !all_fail) {
      x = u64x2{~0ull, ~0ull};
#pragma unroll
      for (int i = 0; i < kMaxClassRows; ++i)
        if (i < cr.n) x &= *(const u64x2*)(cr.row[i] + w);
      if (cr.ni) {
        x.x &= class_idx_word(cr, w);
        if (w + 1 < row_words) x.y &= class_idx_word(cr, w + 1);
      }
      if (w + 1 >= row_words) x.y = 0;
      if (pin >= 0) {
        x.x &= (w == (pin >> 6)) ? (1ull << (pin & 63)) : 0ull;
        x.y &= (w + 1 == (pin >> 6)) ? (1ull << (pin & 63)) : 0ull;
      }
    }
    pc += __popcll(x.x) + __popcll(x.y);
    if (store) *(u64x2*)(dst + w) = x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) pc += __shfl_down(pc, off, kWave);
  if (lane == 0) class_count[cls] = pc;
}

// grid = kBandGroups workgroups of kBandBlock = 256 + 64 threads; dynamic LDS = 2 * kBandClasses * row_stride * 8 bytes.
// Waves 0-3 are the STORE waves: thread t of workgroup b owns bytes [4096 b + 16 t, +16) of every window, reads its 16 bytes
// of the right class row from LDS and stores them; they never load from global memory, so their vmcnt is never waited on
// (gfx9 counts loads and stores in ONE in-order counter: a wave that both prefetches class rows and streams stores has to
// drain its whole store queue before it can use the prefetch). Wave 4 is the LOADER: during band k it copies the class rows
// of band k + 1 into the other LDS buffer. One s_barrier per band joins them.
constexpr int kBandBlock = kBlock + kWave;
__global__ __launch_bounds__(kBandBlock) void k_expand_bands(u64* __restrict__ out, const u64* __restrict__ class_rows,
                                                             const BandEntry* __restrict__ tab, int n_bands, int row_stride) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  extern __shared__ u64 band_lds[];  // [2][kBandClasses][row_stride]
  const int b = blockIdx.x, tid = threadIdx.x;
  const int row_b = row_stride * 8;
  constexpr long kWin = (long)kBandGroups * 4096;
  if (tid >= kBlock) {  // ---- loader wave
    const int lane = tid - kBlock;
    const int pieces = row_stride / 2;  // 16-byte pieces per row
    auto load_rows = [&](int band) {
      const BandEntry be = tab[(size_t)band * kBandGroups + b];
#pragma unroll
      for (int c = 0; c < kBandClasses; ++c) {
        if (c >= be.n && c > 0) break;
        const u64x2* src = (const u64x2*)(class_rows + (size_t)be.slot[c] * row_stride);
        u64x2* dst = (u64x2*)(band_lds + (size_t)((band & 1) * kBandClasses + c) * row_stride);
        for (int i0 = 0; i0 < pieces; i0 += 4 * kWave) {
          u64x2 r[4];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (i0 + k * kWave + lane < pieces) r[k] = src[i0 + k * kWave + lane];
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (i0 + k * kWave + lane < pieces) dst[i0 + k * kWave + lane] = r[k];
        }
      }
    };
    load_rows(0);
    __syncthreads();
    for (int band = 0; band < n_bands; ++band) {
      if (band + 1 < n_bands) load_rows(band + 1);
      __syncthreads();
    }
    return;
  }
  // ---- store waves
  __syncthreads();
  const int p = b * 4096 + tid * 16;
  const unsigned urow_b = (unsigned)row_b;
  const unsigned dcol = (unsigned)(kWin % row_b);  // column advance per step
  unsigned ucol = (unsigned)(p % row_b);           // column (byte) of the thread's 16 bytes in their row, step 0
  for (int band = 0; band < n_bands; ++band) {
    const BandEntry cur = tab[(size_t)band * kBandGroups + b];
    const int s0 = cur.first_step, s1 = s0 + cur.steps;
    const char* lds = (const char*)(band_lds + (size_t)(band & 1) * kBandClasses * row_stride);
    // class boundaries as ONE comparable number: (X << 8) | step-in-band — a row belongs to class i + 1.. iff its key >= kb[i]
    unsigned kb[kBandClasses - 1];
#pragma unroll
    for (int i = 0; i < kBandClasses - 1; ++i)
      kb[i] = cur.xb[i] == 0x7fffffff ? 0xffffffffu : (((unsigned)cur.xb[i] << 8) | (unsigned)(cur.sb[i] - s0));
    char* win_base = (char*)out + (long)s0 * kWin;
    // One loop body per number of classes the workgroup has to tell apart in this band: more than half of the (workgroup,
    // band) pairs see a single class — no compare at all — most of the rest two.
    auto run_band = [&](auto ncls_tag) {
      constexpr int kN = decltype(ncls_tag)::value;
      constexpr int kU = 4;
      for (int s = s0; s < s1; s += kU) {
        u64x2 v[kU];
        bool live[kU];
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          unsigned off = ucol;       // LDS byte offset: class index * row bytes + column
          live[u] = p >= (int)ucol;  // x = p - col >= 0: the row started in this window (else k_fix_rows writes it whole)
          if (kN > 1) {
            const unsigned key = ((unsigned)(p - (int)ucol) << 8) | (unsigned)(s + u - s0);
#pragma unroll
            for (int i = 0; i < kN - 1; ++i) off += key >= kb[i] ? urow_b : 0u;
          }
          v[u] = *(const u64x2*)(lds + off);
          const unsigned t = ucol + dcol;
          ucol = min(t, t - urow_b);  // t < row_b: t - row_b wraps to a huge value
        }
#pragma unroll
        for (int u = 0; u < kU; ++u) {
          if (live[u]) *(u64x2*)(win_base + p) = v[u];
          win_base += kWin;
        }
      }
    };
    if (cur.n <= 1)
      run_band(std::integral_constant<int, 1>{});
    else if (cur.n == 2)
      run_band(std::integral_constant<int, 2>{});
    else if (cur.n == 3)
      run_band(std::integral_constant<int, 3>{});
    else
      run_band(std::integral_constant<int, kBandClasses>{});
    __syncthreads();
  }
}
// one block per row that straddles a window boundary: the whole row from its class row
__global__ __launch_bounds__(kBlock) void k_fix_rows(u64* __restrict__ out, const u64* __restrict__ class_rows, const int* __restrict__ rows,
                                                     const int* __restrict__ slots, int n, int row_stride) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  if ((int)blockIdx.x >= n) return;
  const u64* src = class_rows + (size_t)slots[blockIdx.x] * row_stride;
  u64* dst = out + (size_t)rows[blockIdx.x] * row_stride;
  for (int w = threadIdx.x * 2; w < row_stride; w += 2 * kBlock) *(u64x2*)(dst + w) = *(const u64x2*)(src + w);
}

// ---------------------------------------------------------------------------------------------------
// decide: best feasible node of a class = first set bit in bin-pack rank order
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void k_decide(ClassTable ct, Planes ranked, int n_classes, int row_words, const int* __restrict__ perm,
                                                   const int* __restrict__ rank, int pin_enabled, int* __restrict__ class_best, int eager) {
  int cls = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  if (cls >= n_classes) return;
  if (ct.decide_list) cls = ct.decide_list[cls];
  const int lane = threadIdx.x % kWave;
  const int sr = ct.sig[cls * 4 + 0], st = ct.sig[cls * 4 + 1], sa = ct.sig[cls * 4 + 2], ss = ct.sig[cls * 4 + 3];
  const bool all_fail = pin_enabled & 2;
  const int pin = (pin_enabled & 1) ? ct.pin[cls] : -1;
  int best = -1;
  const ClassRows cr = class_rows(ranked, sr, st, sa, ss);
  if (pin == -2 || all_fail) {
    best = -1;
  } else if (pin >= 0) {
    int pos = rank[pin];
    u64 x = class_word(cr, pos >> 6);
    best = ((x >> (pos & 63)) & 1ull) ? pin : -1;
  } else {
    // (cr.start = kNoWord ≥ row_words: some row of the class has no node at all — nothing to scan)
    for (int base = cr.start & ~(kWave - 1); base < row_words; base += kWave) {
      int w = base + lane;
      // Lane-predicated AND, request-value planes first (they come last in `cr`): bin-pack order tries the fullest nodes
      // first, where those planes are mostly zero — a lane whose word is already zero loads nothing more, so the scan to the
      // first feasible node reads little beyond one plane.
      u64 x = w < row_words ? ~0ull : 0ull;
      if (eager) {
        // few classes (< one wave per SIMD): nothing to save by skipping loads, everything to gain from issuing a step's plane
        // words together instead of one after the other — the scan is a chain of memory latencies beside a saturated HBM
        const int ws = w < row_words ? w : 0;
        u64 y = ~0ull;
#pragma unroll
        for (int i = 0; i < kMaxClassRows; ++i)
          if (i < cr.n) y &= cr.row[i][ws];
        if (cr.ni) y &= class_idx_word(cr, ws);
        x &= y;
      } else {
        if (cr.ni && x) x &= class_idx_word(cr, w);  // (the request-value rows first: see above)
#pragma unroll
        for (int i = kMaxClassRows - 1; i >= 0; --i)
          if (i < cr.n && x) x &= cr.row[i][w];
      }
      u64 any = __ballot(x != 0);
      if (any) {
        int first_lane = __ffsll((long long)any) - 1;
        u64 xw = __shfl(x, first_lane, kWave);
        int pos = (base + first_lane) * kWave + (__ffsll((long long)xw) - 1);
        best = perm[pos];
        break;
      }
    }
  }
  if (lane == 0) class_best[cls] = best;
}

// Sub-wave form for many small classes (every ask its own template / request vector: 10^5 ... 10^6 classes). The whole-wave
// form above spends one wave — a chain of dependent loads class -> signatures -> row ids -> plane words, and 5 x 512 bytes of
// plane words per step — on a class whose first feasible node usually sits in the first few hundred positions of the bin-pack
// order. Here a wave serves kDecideGroups classes at once: group g = 16 lanes = 16 words = 1 024 positions per step; the
// classes' table entries are fetched by the lanes in parallel (one load round for four classes), plane words are loaded
// lane-predicated as above, and a group that has found its node idles while the others go on.
constexpr int kDecideGroups = 4, kDecideLanes = kWave / kDecideGroups;
__global__ __launch_bounds__(kBlock) void k_decide_groups(ClassTable ct, Planes ranked, int n_classes, int row_words, const int* __restrict__ perm,
                                                          const int* __restrict__ rank, int pin_enabled, int* __restrict__ class_best) {
  const int wave = blockIdx.x * kWavesPerBlock + threadIdx.x / kWave;
  const int lane = threadIdx.x % kWave, g = lane / kDecideLanes, l = lane % kDecideLanes;
  const int cls_raw = wave * kDecideGroups + g;
  const bool live = cls_raw < n_classes;
  const int cls = ct.decide_list ? ct.decide_list[live ? cls_raw : n_classes - 1] : (live ? cls_raw : n_classes - 1);
  const bool all_fail = pin_enabled & 2;
  // every lane of the group reads the same table entries (one broadcast load each): no cross-lane traffic needed afterwards
  const int sr = ct.sig[cls * 4 + 0], st = ct.sig[cls * 4 + 1], sa = ct.sig[cls * 4 + 2], ss = ct.sig[cls * 4 + 3];
  const int pin = (pin_enabled & 1) ? ct.pin[cls] : -1;
  const ClassRows cr = class_rows(ranked, sr, st, sa, ss);  // (per lane group here, not wave-uniform: the pointers live in VGPRs)
  auto word_at = [&](int w) {
    u64 x = w < row_words ? ~0ull : 0ull;
    if (cr.ni && x) x &= class_idx_word(cr, w);
#pragma unroll
    for (int i = kMaxClassRows - 1; i >= 0; --i)
      if (i < cr.n && x) x &= cr.row[i][w];
    return x;
  };
  int best = -1;
  bool done = !live || pin == -2 || all_fail;
  if (!done && pin >= 0) {
    const int pos = rank[pin];
    const u64 x = word_at(pos >> 6);
    best = ((x >> (pos & 63)) & 1ull) ? pin : -1;
    done = true;
  }
  // every group starts where its class's rows first have a bit at all (cr.start; kNoWord = an empty row: nothing to scan)
  const int w0 = cr.start >= row_words ? row_words : (cr.start & ~(kDecideLanes - 1));
  for (int t = 0;; ++t) {
    const int base = w0 + t * kDecideLanes;
    if (base >= row_words) done = true;
    if (__ballot(!done) == 0) break;
    const u64 x = done ? 0ull : word_at(base + l);
    const u64 any = __ballot(x != 0);
    const unsigned mine = (unsigned)((any >> (g * kDecideLanes)) & ((1u << kDecideLanes) - 1u));
    const int first = mine ? __ffs((int)mine) - 1 : 0;  // first lane of MY group with a feasible position
    const u64 xw = __shfl(x, g * kDecideLanes + first, kWave);  // (executed by every lane: the exchange stays convergent)
    if (!done && mine) {
      best = perm[(base + first) * kWave + (__ffsll((long long)xw) - 1)];
      done = true;
    }
  }
  if (live && l == 0) class_best[cls] = best;
}

// decision key = order-preserving signed image of the node's sortable score key (smaller = earlier in bin-pack order)
// The decisions of the SWEEP RUNS (round 6). A run's rows share every plane row and ascend in the value of one walked dimension, so
// their first feasible nodes in bin-pack order are one monotone staircase: walking the rank order once, a node enters the answer only
// where it holds MORE of the dimension than every feasible node before it (a record of g = its free value's position among the sorted
// request values, k_dim_sort), and it is the decision of every row of the run whose position lies below that record and above the
// record before. One wave per run: the AND of the run's rank-ordered plane rows, 64 words at a time, the records found lane by lane,
// the rows handed their node in order — no window of an index row is written or read for them (k_dim_walk_window), no scan per class
// (k_decide). The scan starts where the smallest value of the run can first have a node (pfx: running maximum of the free values
// along the order) and ends with the run's last row.
constexpr int kRunDecideRows = 512;  // rows of a run one wave decides (a longer run is cut: every piece scans from where ITS smallest value can
                                     // first have a node to its own last row — the scan is cheap, a wave walking 65 000 rows is not)
struct RunRange {
  int run, big, begin, count;  // run (SweepRun), walked dimension, first row in the sweep row list, rows
};
__global__ __launch_bounds__(kBlock) void k_run_decide(Planes ranked, const SweepRun* __restrict__ runs, const RunRange* __restrict__ ranges, int n_ranges,
                                                       const SweepRow* __restrict__ rows, const unsigned* __restrict__ glin, const i64* __restrict__ sorted,
                                                       const int* __restrict__ sorted_off, const int* __restrict__ perm, int n_words, int n_nodes,
                                                       int pin_enabled, int* __restrict__ class_best) {
  const int k = __builtin_amdgcn_readfirstlane((int)(blockIdx.x * kWavesPerBlock + threadIdx.x / kWave));
  if (k >= n_ranges) return;
  const int lane = threadIdx.x % kWave;
  const RunRange rr = ranges[k];
  const SweepRun run = runs[rr.run];
  const int4* vrow = (const int4*)(rows + rr.begin);
  const bool all_fail = pin_enabled & 2;
  const u64* p0 = ranked.res;
  const u64* pp = ranked.res ? ranked.res + (size_t)run.prow * ranked.stride : nullptr;
  const u64* pt = (ranked.tol && run.st >= 0) ? ranked.tol + (size_t)run.st * ranked.stride : p0;
  const u64* pa = (ranked.aff && run.sa >= 0) ? ranked.aff + (size_t)run.sa * ranked.stride : p0;
  const u64* ps = (ranked.spread && run.ss >= 0) ? ranked.spread + (size_t)run.ss * ranked.stride : p0;
  const unsigned* gl = glin + (size_t)rr.big * n_words * 64;
  const unsigned* gmax = glin + (size_t)ranked.n_big * n_words * 64 + (size_t)rr.big * n_words;
  int i = 0, ib = 0;  // the next row without a node; the batch of 64 rows the lanes hold
  int4 d = vrow[min(lane, rr.count - 1)];
  int start = 0;
  if (ranked.pfx) {  // the first word the run's smallest value can have a node in
    const i64 v = sorted[sorted_off[rr.big] + __builtin_amdgcn_readfirstlane(d.z)];
    const i64* px = ranked.pfx + (size_t)rr.big * n_words;
    int lo = 0, hi = n_words;
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (px[mid] < v) lo = mid + 1; else hi = mid;
    }
    start = lo;
  }
  // the largest g among the feasible nodes so far — from the range's first position on: a node below it decides no row of the range
  unsigned cur = (unsigned)__builtin_amdgcn_readfirstlane(d.z);
  if (!all_fail && p0)
    for (int wb = start & ~(kWave - 1); wb < n_words && i < rr.count; wb += kWave) {
      const int w = wb + lane;
      const u64 S = w < n_words ? (p0[w] & pp[w] & pt[w] & pa[w] & ps[w]) : 0ull;
      const unsigned wmx = w < n_words ? gmax[w] : 0u;
      u64 done_mask = 0;  // the words of the block looked at so far (in ascending order: a word passed over stays passed over — cur only grows)
      for (;;) {
        // the words of the block that have a feasible node AND a node that can set a record (their nodes are looked at one word at
        // a time: a load per word, in order — so as few words as possible)
        const u64 todo = __ballot(S != 0 && wmx > cur) & ~done_mask;
        if (!todo || i >= rr.count) break;
        const int wk = __ffsll((long long)todo) - 1;
        done_mask |= (2ull << wk) - 1;
        const u64 Sk = (u64)readlane_i64((i64)S, wk);
        const unsigned g = gl[(size_t)(wb + wk) * 64 + lane];
        u64 m = __ballot(((Sk >> lane) & 1ull) && g > cur);
        while (m && i < rr.count) {
          const int f = __ffsll((long long)m) - 1;
          cur = (unsigned)__builtin_amdgcn_readlane((int)g, f);
          const int pos = (wb + wk) * kWave + f;
          const int node = pos < n_nodes ? perm[pos] : -1;
          for (;;) {  // the rows below this record, in order (the lanes hold a batch of them)
            const int idx = ib + lane;
            const bool take = idx >= i && idx < rr.count && (unsigned)d.z < cur;
            if (take) class_best[d.x] = node;
            i += __popcll(__ballot(take));
            if (i < ib + kWave || i >= rr.count) break;
            ib += kWave;
            d = vrow[min(ib + lane, rr.count - 1)];
          }
          m = __ballot(((Sk >> lane) & 1ull) && g > cur);
        }
      }
    }
  // rows no node of the run's planes holds enough for
  while (i < rr.count) {
    const int idx = ib + lane;
    if (idx >= i && idx < rr.count) class_best[d.x] = -1;
    i = min(ib + kWave, rr.count);
    if (i < rr.count) {
      ib += kWave;
      d = vrow[min(ib + lane, rr.count - 1)];
    }
  }
}
__global__ __launch_bounds__(kBlock) void k_scatter(int n_pods, const int* __restrict__ pod_class, const int* __restrict__ class_count,
                                                    const int* __restrict__ class_best, const u64* __restrict__ node_key,
                                                    int* __restrict__ counts, int* __restrict__ decisions, i64* __restrict__ keys) {
  int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_pods) return;
  int c = pod_class[p];
  if (counts) counts[p] = class_count[c];
  int b = (decisions || keys) ? class_best[c] : -1;
  if (decisions) decisions[p] = b;
  if (keys) keys[p] = b >= 0 ? (i64)(node_key[b] ^ 0x8000000000000000ull) : 0x7fffffffffffffffll;
}

// ---------------------------------------------------------------------------------------------------
// per-pair evaluation straight from the tables (k_direct: whole grid; k_query: listed pairs)
// ---------------------------------------------------------------------------------------------------
struct SpecTable {
  int R, KT, W;
  const i64* req;        // [S][R]
  const u64* tol;        // [S][KT]
  const unsigned* flags; // [S]
  AffSigs aff;           // indexed by spec
  int KP;
  const u64* wanted_ports; // [S][KP]
  const int* spread_sig; // [S] spread signature of the spec, -1 = no hard constraints (PreFilter Skip)
  SpreadSigs spread;
};

struct NodeRegs {
  i64 fr[kMaxR];
  u64 tn[kMaxKT];
  u64 lb[kMaxW];
  const u64* lb_more;  // label words kMaxW.. of this node (stride lb_stride), null past the table
  size_t lb_stride;
  u64 pt[kMaxKP];
  int dom[kMaxKD];
  bool slots_ok, unsched;
};
__device__ __forceinline__ void load_node(const NodeTable& t, int n, NodeRegs* r) {
#pragma unroll
  for (int i = 0; i < kMaxR; ++i) r->fr[i] = (i < t.R && n >= 0) ? t.alloc[(size_t)i * t.n + n] - t.req[(size_t)i * t.n + n] : 0;
#pragma unroll
  for (int i = 0; i < kMaxKT; ++i) r->tn[i] = (i < t.KT && n >= 0) ? t.taints[(size_t)i * t.n + n] : 0;
#pragma unroll
  for (int i = 0; i < kMaxW; ++i) r->lb[i] = (i < t.W && n >= 0) ? t.labels[(size_t)i * t.n + n] : 0;
  r->lb_more = (t.W > kMaxW && n >= 0) ? t.labels + (size_t)kMaxW * t.n + n : nullptr;
  r->lb_stride = (size_t)t.n;
#pragma unroll
  for (int i = 0; i < kMaxKP; ++i) r->pt[i] = (i < t.KP && n >= 0) ? t.ports[(size_t)i * t.n + n] : 0;
#pragma unroll
  for (int i = 0; i < kMaxKD; ++i) r->dom[i] = (i < t.KD && n >= 0) ? t.domain[(size_t)i * t.n + n] : -1;
  r->slots_ok = n >= 0 && (i64)t.count[n] + 1 <= (i64)t.allowed[n];
  r->unsched = n >= 0 && (t.flags[n] & kNodeUnschedulable);
}

// One Predicates() call: PreFilter pass, then the ordered Filter list with early exit
// (predicate_manager.go:206-283). Returns fit; *code / *reason describe the first failure.
template <bool LIVE = false>
__device__ __forceinline__ bool eval_pair(const SpecTable& s, int spec, int pin, int node, const NodeRegs& nr, unsigned pre_mask,
                                          unsigned filt_mask, int* code, unsigned* reason) {
  unsigned f = s.flags[spec];
  *code = 0;
  *reason = 0;
  if (f & kSpecUnsupported) {  // the host routes this ask to the CPU manager
    *code = 255;
    return false;
  }
  // --- PreFilter plugins, MultiPoint order: NodeAffinity, NodeResourcesFit (never rejects), PodTopologySpread
  if ((pre_mask & kPlugAffinity) && !(f & kSpecAffSkip)) {
    if (f & kSpecPreReject) {
      *code = 0;
      *reason = 1u << 2;
      return false;
    }
    if ((f & kSpecPreNames) && !dnf_match(s.aff.pre_terms, s.aff.pre_off[spec], s.aff.pre_off[spec + 1], nr.lb, s.W, nr.lb_more, nr.lb_stride)) {
      *code = 4;
      *reason = 1u << 1;
      return false;
    }
  }
  // --- Filter plugins
  if ((filt_mask & kPlugUnsched) && nr.unsched && !(f & kSpecToleratesUnsched)) {
    *code = 1;
    return false;
  }
  if ((filt_mask & kPlugNodeName) && pin != -1 && pin != node) {
    *code = 2;
    return false;
  }
  if (filt_mask & kPlugTaint) {
    const u64* tol = s.tol + (size_t)spec * s.KT;
    bool ok = true;
#pragma unroll
    for (int k = 0; k < kMaxKT; ++k)
      if (k < s.KT) ok = ok && (nr.tn[k] & ~tol[k]) == 0;
    if (!ok) {
      *code = 3;
      return false;
    }
  }
  if (filt_mask & kPlugAffinity) {
    bool skip = (pre_mask & kPlugAffinity) && (f & kSpecAffSkip);
    if (!skip && !dnf_match(s.aff.terms, s.aff.term_off[spec], s.aff.term_off[spec + 1], nr.lb, s.W, nr.lb_more, nr.lb_stride)) {
      *code = 4;
      return false;
    }
  }
  if (filt_mask & kPlugPorts) {
    const u64* want = s.wanted_ports + (size_t)spec * s.KP;
    bool any = false, conflict = false;
#pragma unroll
    for (int k = 0; k < kMaxKP; ++k)
      if (k < s.KP) {
        any = any || want[k] != 0;
        conflict = conflict || (nr.pt[k] & want[k]) != 0;
      }
    if (!(pre_mask & kPlugPorts)) {
      *code = 5;  // Filter without PreFilter state: Error status
      return false;
    }
    if (any && conflict) {  // no requested host port ⇒ PreFilter Skip ⇒ Filter not run
      *code = 5;
      return false;
    }
  }
  if (filt_mask & kPlugFit) {
    unsigned why = 0;
    if (!(pre_mask & kPlugFit)) {
      *code = 6;  // Filter without PreFilter state: Error status (:269-277)
      return false;
    }
    if (!nr.slots_ok) why |= 1u;
    const i64* q = s.req + (size_t)spec * s.R;
#pragma unroll
    for (int r = 0; r < kMaxR; ++r)
      if (r < s.R && q[r] > 0 && q[r] > nr.fr[r]) why |= 1u << (8 + r);
    if (why) {
      *code = 6;
      *reason = why;
      return false;
    }
  }
  if ((filt_mask & kPlugSpread) && !(pre_mask & kPlugSpread)) {
    *code = 7;  // Filter without PreFilter state: Error status
    return false;
  }
  {
    const int d = s.spread_sig ? s.spread_sig[spec] : -1;
    const bool spread_en = filt_mask & kPlugSpread, ipa_en = (filt_mask & kPlugInterPod) && (pre_mask & kPlugInterPod);
    if (d >= 0 && (spread_en || ipa_en)) {
      unsigned missing = 0;
      int f = constraints_fail<LIVE>(s.spread, d, nr.dom, spread_en, ipa_en, &missing);
      if (f == 7) {
        *code = 7;
        *reason = missing ? (1u << 3) : 0u;
        return false;
      }
      if ((filt_mask & kPlugInterPod) && !(pre_mask & kPlugInterPod)) {
        *code = 8;
        return false;
      }
      if (f) {
        *code = f;
        return false;
      }
    }
  }
  if ((filt_mask & kPlugInterPod) && !(pre_mask & kPlugInterPod)) {
    *code = 8;  // Filter without PreFilter state: Error status
    return false;
  }
  return true;
}

__global__ __launch_bounds__(kBlock) void k_query(NodeTable t, SpecTable s, int n_pairs, const int* __restrict__ pods,
                                                  const int* __restrict__ nodes, const int* __restrict__ pod_spec,
                                                  const int* __restrict__ pod_pin, unsigned pre_mask, unsigned filt_mask,
                                                  unsigned char* __restrict__ fit, unsigned char* __restrict__ code_out,
                                                  unsigned* __restrict__ reason_out) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_pairs) return;
  int p = pods[i], n = nodes[i];
  NodeRegs nr;
  load_node(t, n, &nr);
  int code;
  unsigned reason;
  bool ok = eval_pair(s, pod_spec[p], pod_pin[p], n, nr, pre_mask, filt_mask, &code, &reason);
  fit[i] = ok ? 1 : 0;
  if (code_out) code_out[i] = (unsigned char)code;
  if (reason_out) reason_out[i] = reason;
}

// one pod against every node (thread = node): the answers of all Predicates(pod, ·) callbacks of a scheduling attempt
__global__ __launch_bounds__(kBlock) void k_query_pod(NodeTable t, SpecTable s, int spec, int pin, unsigned pre_mask, unsigned filt_mask,
                                                      unsigned char* __restrict__ fit, unsigned char* __restrict__ code_out,
                                                      unsigned* __restrict__ reason_out) {
  int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= t.n) return;
  NodeRegs nr;
  load_node(t, n, &nr);
  int code;
  unsigned reason;
  bool ok = eval_pair(s, spec, pin, n, nr, pre_mask, filt_mask, &code, &reason);
  fit[n] = ok ? 1 : 0;
  code_out[n] = (unsigned char)code;
  reason_out[n] = reason;
}

// the same answers packed into one word per node: code | fit << 8 | (reason & 15) << 9 | (reason >> 8) << 13
__global__ __launch_bounds__(kBlock) void k_query_pod_packed(NodeTable t, SpecTable s, int spec, int pin, unsigned pre_mask, unsigned filt_mask,
                                                             unsigned* __restrict__ out) {
  int n = blockIdx.x * kBlock + threadIdx.x;
  if (n >= t.n) return;
  NodeRegs nr;
  load_node(t, n, &nr);
  int code;
  unsigned reason;
  bool ok = eval_pair(s, spec, pin, n, nr, pre_mask, filt_mask, &code, &reason);
  out[n] = (unsigned)(code & 0xff) | (ok ? 0x100u : 0u) | ((reason & 0xfu) << 9) | ((reason >> 8) << 13);
}

// Per-pair grid: blockIdx.x = chunk of 64 pods (the unbounded axis), blockIdx.y = group of 4 node words. lane = node, the wave walks the
// 64 pods of its chunk (pod data wave-uniform), ballot → lane (i) keeps pod i's word → 64 row stores.
__global__ __launch_bounds__(kBlock) void k_direct(NodeTable t, SpecTable s, int n_pods, const int* __restrict__ pod_spec,
                                                   const int* __restrict__ pod_pin, const int* __restrict__ pod_row, unsigned pre_mask,
                                                   unsigned filt_mask, u64* __restrict__ bitmap, int row_words, int row_stride) {
  int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  int w = blockIdx.y * kWavesPerBlock + wave;
  if (w >= row_stride) return;
  int n = w * kWave + lane;
  if (n >= t.n) n = -1;
  NodeRegs nr;
  load_node(t, n, &nr);
  int p0 = blockIdx.x * kWave;
  int pend = min(p0 + kWave, n_pods);
  u64 keep = 0;
  for (int p = p0; p < pend; ++p) {
    int code;
    unsigned reason;
    bool ok = n >= 0 && eval_pair(s, pod_spec[p], pod_pin[p], n, nr, pre_mask, filt_mask, &code, &reason);
    u64 b = __ballot(ok);
    if (p - p0 == lane) keep = b;
  }
  int p = p0 + lane;
  if (p < n_pods) bitmap[(size_t)pod_row[p] * row_stride + w] = (w < row_words) ? keep : 0ull;
}

// ---------------------------------------------------------------------------------------------------
// incremental column patch: only the bitmap columns of the listed (updated) nodes are re-evaluated
// ---------------------------------------------------------------------------------------------------
struct ColumnGroups {     // updated nodes grouped by bitmap word (host-built, <= kMaxColGroups per launch)
  int n_groups;
  int word[64];           // bitmap word index of the group
  int first[65];          // nodes of group g = nodes[first[g] .. first[g+1])
  int nodes[64];          // node indices
};
constexpr int kMaxColGroups = 64;
// thread = class: every member row of a class is identical, so the old word is read from the first member's row, the
// touched bits are re-evaluated per pair straight from the tables (same routine as k_query / k_direct), and the class's
// feasible count moves by the popcount difference.
__global__ __launch_bounds__(kBlock) void k_column_class(NodeTable t, SpecTable s, ColumnGroups cg, int n_classes,
                                                         const int* __restrict__ class_first, const int* __restrict__ class_pin,
                                                         const int* __restrict__ pod_spec, const int* __restrict__ pod_row, unsigned pre_mask,
                                                         unsigned filt_mask, const u64* __restrict__ bitmap, int row_stride,
                                                         u64* __restrict__ class_word, int* __restrict__ class_count) {
  int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= n_classes) return;
  const int p0 = class_first[c];
  if (p0 < 0) return;  // no live member (ykpred_update_pods moved them all away)
  const int spec = pod_spec[p0];
  const int pin = (filt_mask & kPlugNodeName) ? class_pin[c] : -1;
  int delta = 0;
  for (int g = 0; g < cg.n_groups; ++g) {
    const u64 old = bitmap[(size_t)pod_row[p0] * row_stride + cg.word[g]];
    u64 neu = old;
    for (int i = cg.first[g]; i < cg.first[g + 1]; ++i) {
      const int n = cg.nodes[i];
      NodeRegs nr;
      load_node(t, n, &nr);
      int code;
      unsigned reason;
      bool ok = eval_pair(s, spec, pin, n, nr, pre_mask, filt_mask, &code, &reason);
      const u64 m = 1ull << (n & 63);
      neu = ok ? (neu | m) : (neu & ~m);
    }
    class_word[(size_t)c * kMaxColGroups + g] = neu;
    delta += __popcll(neu) - __popcll(old);
  }
  if (delta) class_count[c] += delta;
}
// thread = pod: store the class's new words into the pod's row, refresh its count
__global__ __launch_bounds__(kBlock) void k_column_patch(ColumnGroups cg, int n_pods, const int* __restrict__ pod_class,
                                                         const int* __restrict__ pod_row, const u64* __restrict__ class_word,
                                                         const int* __restrict__ class_count, u64* __restrict__ bitmap, int row_stride,
                                                         int* __restrict__ counts) {
  int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_pods) return;
  const int c = pod_class[p];
  const size_t row = (size_t)pod_row[p] * row_stride;
  for (int g = 0; g < cg.n_groups; ++g) bitmap[row + cg.word[g]] = class_word[(size_t)c * kMaxColGroups + g];
  if (counts) counts[p] = class_count[c];
}

// ---------------------------------------------------------------------------------------------------
// incremental row patch: ask-table rows rewritten by ykpred_update_pods are re-evaluated per pair
// ---------------------------------------------------------------------------------------------------
// Small integer patches of the engine's int32 tables in one launch: {table, index, value}.
struct TablePatch {
  int table, index, value, pad;
};
constexpr int kPatchTables = 14;
constexpr int kNoRank = 0x7f7f7f7f;  // row_best is initialised with memset(0x7f); ranks are < 2^24
struct TablePtrs {
  int* t[kPatchTables];
};
__global__ __launch_bounds__(kBlock) void k_apply_patches(TablePtrs tp, int n, const TablePatch* __restrict__ patches) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n) return;
  TablePatch q = patches[i];
  tp.t[q.table][q.index] = q.value;
}

// blockIdx.x = listed row, blockIdx.y = group of 4 node words, lane = node: the whole bitmap row of the pod is rebuilt
// with the per-pair routine (same as k_direct), its popcount and the smallest bin-pack rank among its feasible nodes
// are accumulated in row_count / row_best (zeroed / set to kNoRank by the caller).
__global__ __launch_bounds__(kBlock) void k_rows(NodeTable t, SpecTable s, int n_rows, const int* __restrict__ rows,
                                                 const int* __restrict__ pod_spec, const int* __restrict__ pod_pin,
                                                 const int* __restrict__ pod_row, unsigned pre_mask, unsigned filt_mask,
                                                 u64* __restrict__ bitmap, int row_words, int row_stride,
                                                 const int* __restrict__ rank /* null: no decisions */, int* __restrict__ row_count,
                                                 int* __restrict__ row_best) {
  const int wave = threadIdx.x / kWave, lane = threadIdx.x % kWave;
  const int w = blockIdx.y * kWavesPerBlock + wave;
  if (w >= row_stride) return;
  const int p = rows[blockIdx.x];
  int n = w * kWave + lane;
  if (n >= t.n) n = -1;
  NodeRegs nr;
  load_node(t, n, &nr);
  int code;
  unsigned reason;
  const bool ok = n >= 0 && eval_pair(s, pod_spec[p], pod_pin[p], n, nr, pre_mask, filt_mask, &code, &reason);
  const u64 b = __ballot(ok);
  int best = (ok && rank) ? rank[n] : kNoRank;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) best = min(best, __shfl_down(best, off, kWave));
  if (lane == 0) {
    bitmap[(size_t)pod_row[p] * row_stride + w] = (w < row_words) ? b : 0ull;
    if (b) {
      atomicAdd(&row_count[blockIdx.x], __popcll(b));
      if (rank) atomicMin(&row_best[blockIdx.x], best);
    }
  }
}
// thread = listed row: publish count / decision of the row and of its class (every member row of a class is identical)
__global__ __launch_bounds__(kBlock) void k_rows_finish(int n_rows, const int* __restrict__ rows, const int* __restrict__ pod_class,
                                                        const int* __restrict__ row_count, const int* __restrict__ row_best,
                                                        const int* __restrict__ perm, const u64* __restrict__ node_key,
                                                        int* __restrict__ class_count, int* __restrict__ class_best,
                                                        int* __restrict__ counts, int* __restrict__ decisions, i64* __restrict__ keys) {
  int i = blockIdx.x * kBlock + threadIdx.x;
  if (i >= n_rows) return;
  const int p = rows[i], c = pod_class[p];
  class_count[c] = row_count[i];
  if (counts) counts[p] = row_count[i];
  if (perm) {
    const int b = row_best[i] >= kNoRank ? -1 : perm[row_best[i]];
    class_best[c] = b;
    if (decisions) decisions[p] = b;
    if (keys) keys[p] = b >= 0 ? (i64)(node_key[b] ^ 0x8000000000000000ull) : 0x7fffffffffffffffll;
  }
}

// PreemptionPredicates (predicate_manager.go:141-179) for one (pod,node): victims are removed in order; returns the first
// index >= start at which the pod fits, or -1.
__device__ __forceinline__ int preempt_one(const NodeTable& t, const SpecTable& s, int spec, int pin, int node, int n_victims,
                                           const i64* __restrict__ vreq, const unsigned char* __restrict__ vpresent,
                                           const u64* __restrict__ ports_after /*[n_victims][KP] or null*/, int start, unsigned pre_mask,
                                           unsigned filt_mask) {
  NodeRegs nr;
  load_node(t, node, &nr);
  int code;
  unsigned reason;
  // PreFilter check first (:146-155): evaluate with filters disabled to see only PreFilter failures
  if (!eval_pair(s, spec, pin, node, nr, pre_mask, 0u, &code, &reason)) return -1;
  i64 pods_on_node = t.count[node];
  const i64 allowed = t.allowed[node];
  for (int i = 0; i < n_victims; ++i) {
    if (vpresent[i]) {  // removePodFromNodeNoFail (:181-192)
      // (compile-time indices: a register array indexed by a loop counter the compiler cannot unroll lives in scratch)
#pragma unroll
      for (int r = 0; r < kMaxR; ++r)
        if (r < s.R) nr.fr[r] += vreq[(size_t)i * s.R + r];
      pods_on_node -= 1;
    }
    if (ports_after) {
#pragma unroll
      for (int k = 0; k < kMaxKP; ++k)
        if (k < t.KP) nr.pt[k] = ports_after[(size_t)i * t.KP + k];
    }
    if (i < start) continue;  // :161-163
    nr.slots_ok = pods_on_node + 1 <= allowed;
    // PreFilter outcomes were accepted above; eval_pair re-checks them (idempotent) and runs the filters (:168)
    if (eval_pair(s, spec, pin, node, nr, pre_mask, filt_mask, &code, &reason)) return i;
  }
  return -1;
}
// thread = query; query q owns victims [voff[q], voff[q+1]) of the flattened victim arrays
__global__ __launch_bounds__(kWave) void k_preempt(NodeTable t, SpecTable s, int n_queries, const int* __restrict__ q_pod,
                                                   const int* __restrict__ q_node, const int* __restrict__ voff,
                                                   const int* __restrict__ pod_spec, const int* __restrict__ pod_pin,
                                                   const i64* __restrict__ vreq, const unsigned char* __restrict__ vpresent,
                                                   const u64* __restrict__ ports_after, const int* __restrict__ q_start,
                                                   unsigned pre_mask, unsigned filt_mask, int* __restrict__ out) {
  int q = blockIdx.x * kWave + threadIdx.x;
  if (q >= n_queries) return;
  const int v0 = voff[q], nv = voff[q + 1] - v0, p = q_pod[q];
  out[q] = preempt_one(t, s, pod_spec[p], pod_pin[p], q_node[q], nv, vreq + (size_t)v0 * s.R, vpresent + v0,
                       ports_after ? ports_after + (size_t)v0 * t.KP : nullptr, q_start[q], pre_mask, filt_mask);
}

// ---------------------------------------------------------------------------------------------------
// Conflict-resolved decisions: the SEQUENTIAL loop yunikorn-core drives through the shim. For the asks of a
// round, in order: the first node of the bin-pack order — under the state the EARLIER asks of the round left behind — that
// passes Predicates() (scheduler_callback.go:203-205 → context.go:696-716); then AssumePod (context.go:828-885 →
// scheduler_cache.go:443-461 → NodeInfo.AddPod): the node's Requested grows by the ask's request vector, len(Pods) by one, its
// bin-pack score moves, its used host ports and the match counts of the topology plugins grow — and the next ask sees that.
//
// ONE WORKGROUP of kRoundWaves waves runs the loop (the asks are a dependency chain; the parallelism is inside an ask). The work
// per ask stays small because a snapshot evaluation with decisions is current when the round starts:
//   * UNMOVED nodes (no allocation in this round yet) keep their state and their relative order: the first feasible one of a
//     class is the first set bit of (AND of the class's rank-ordered planes) & ~moved — thread = 64-node word, the whole row of
//     a 50 k-node cluster in two steps, from a per-class cursor that only ever advances (bits only leave);
//   * MOVED nodes (bin-packing piles the asks onto few nodes, and a node without a free pod slot leaves the list for good) are
//     evaluated per pair from the LIVE tables (eval_pair: the routine of k_query / k_direct), thread = node, with their current
//     score key — kRoundThreads of them per step;
//   * the winner is the smaller (score key, NodeID rank) of the two candidates — what a walk over the re-sorted node list
//     would have found first.
// What an assumed pod changes BESIDES the node's resources (SpecEffects, uploaded by the host per spec): the host-port words of
// the node (NodePorts) and the match counts behind PodTopologySpread / InterPodAffinity. The round keeps the PreFilter state of
// the topology plugins LIVE: one histogram cell per assumed pod and matching constraint, the running minimum of a spread
// constraint through a count of the domains that sit at the minimum (a full pass over the constraint's domains only when that
// count reaches zero). A class with a topology signature cannot use its snapshot plane for those plugins — an assumed pod
// moves the verdict of every node of its domain, and of every node when the minimum rises — so its unmoved candidates (AND of
// the OTHER planes) are checked against the live histograms, a wave per 64-node word, lane = node.
// Mutable state is read with agent-scope atomics (or after the per-ask fence) and written by one thread per item.
constexpr int kRoundWaves = 8, kRoundThreads = kRoundWaves * kWave;
constexpr int kRoundDirty = 16;  // spread constraints whose minimum has to be recomputed after one assume (more: the owner thread does it alone)
struct SpecEffects {       // what NodeInfo.AddPod of a pod of spec s adds to its node besides the request vector (null: none uploaded)
  const int* off;          // [S+1] rows of cls / cnt
  const int* cls;          // selector class (column of selector_count)
  const int* cnt;          // what one pod of the spec adds to that column
  const u64* occupied;     // [S][KP] dictionary host ports a pod of the spec conflicts with once it is on a node
};
constexpr int kRoundDecide = 0, kRoundAssume = 2;
struct RoundProposal {      // a shard's best node for an ask, and what the other ranks need to order it and to re-key it after assumes
  u64 key;                  // score key of the node as it stands (smaller = earlier)
  int node;                 // node index in this shard, -1 = no node of the shard fits
  int fits;                 // how many pods of the ask's spec the node still holds when they couple through resources only, else 1
  i64 alloc[kMaxR], req[kMaxR];  // Allocatable and Requested of every resource dimension ([0], [1] = cpu, memory: the key after k more pods
                            // of a spec is arithmetic on these; all of them: whether a LATER ask of the batch still fits the node is too)
  int gnode, room;          // node index in the whole cluster (the shard's node offset + node): the tie-break between equal keys;
                            // pod slots left (AllowedPodNumber - len(Pods))
  int didx, pad;            // batched rounds: the node's number among the distinct nodes its shard proposed in this batch (k_round_distinct);
                            // pad = its NodeID rank inside the shard (the tie-break on one GPU)
};
struct RoundArgs {
  int first, n_asks;        // this launch decides asks [first, first + n_asks) of the round
  const int* asks;          // [round] ask (pod) indices in decision order
  const int* pod_spec;
  const int* pod_pin;
  const int* pod_class;
  const int* perm;          // bin-pack order of the snapshot ...
  const int* rank;
  const u64* key0;          // ... and its score keys
  const u64* rkey;          // [N] key0 in bin-pack order (rkey[pos] = key0[perm[pos]]) and the NodeID ranks beside it: what candidate A
  const int* rtie;          //     needs of its node, addressed by the POSITION the scan found (k_round_ranked_keys, once per round)
  const u64* cdesc;         // [C][kDescWords] the rank-ordered plane rows of every class, resolved once per round (k_round_class_desc)
  const int* name_rank;     // NodeID order (null: node index)
  unsigned pre, filt;
  int row_words, all_fail;
  i64* req;                 // [R][N] scratch copy of Requested — grows with the round (t.req points here too)
  int* count;               // [N] scratch copy of len(Pods)
  u64* ports;               // [KP][N] scratch copy of the host-port words (t.ports points here too); null: no ask of the round has host ports
  u64* moved_bits;          // [row_words], rank order
  int* cursor;              // [C] -1 = not started
  int* n_moved;             // slots in use, carried between the launches of one round
  int* out;                 // [round] node index, -1 = no node fits
  // MOVED nodes live in SLOTS (the order in which they first received a pod of this round; a slot is never reused): their columns
  // side by side, so that the per-ask scan over them is coalesced loads by slot instead of twenty gathers by node index
  int cap, cap64;           // slots (= N), 64-bit words of a bitset over them
  int* slot_of;             // [N] node -> slot, -1 = unmoved
  int* m_node;              // [cap] slot -> node
  int* m_tie;               // [cap] NodeID rank (tie-break)
  u64* m_key;               // [cap] current score key                          (live)
  i64* m_free;              // [R][cap] Allocatable - Requested                   (live)
  int* m_room;              // [cap] AllowedPodNumber - len(Pods)                 (live)
  u64* m_ports;             // [KP][cap] host-port words                          (live when a.ports)
  u64* m_taint;             // [KT][cap]
  u64* m_label;             // [min(W, kMaxW)][cap]
  int* m_dom;               // [KD][cap]
  unsigned* m_flags;        // [cap] node flags
  u64* dead;                // [cap64] no pod slot left: out of every later ask's reach
  u64* failed;              // [C][cap64] bit = the class did not fit the slot's node when it last looked. Without a topology signature a
                            // verdict only ever turns from fit to fail during a round (Requested, pod counts and port sets only grow; taints,
                            // labels and names do not move), so a set bit is final. null: too many classes to keep the bits
  // topology plugins (s.spread.cnt / .minv point at the round's scratch copies); topo_on = 0: no signature is active
  int topo_on, G;           // G = constraints of all signatures
  AffSigs sig_aff;          // eligibility tables of the signatures (nodeAffinityPolicy / nodeTaintsPolicy == Honor)
  const u64* sig_tol;
  const int* sig_of;        // [G] signature of constraint g
  int* mn;                  // [G] spread: minimum over the present domains (before the minDomains rule)
  int* at_min;              // [G] spread: present domains whose count equals mn
  const int* nd;            // [G] spread: present domains
  SpecEffects fx;
  // Batched rounds (engine.hip, ykpred_allocate_round: sharded engines always, one GPU by the ask list): the proposals of a batch
  // come from k_round_propose / k_round_cross, the host replays the loop, and the accepted pods are assumed node by node
  // (k_round_assume_nodes) or — batches that move host ports or topology histograms — by this kernel in assume mode: the loop's assume
  // for given nodes, no scans.
  int mode;                 // kRoundDecide (0) or kRoundAssume (the proposals of a batched round are k_round_propose's)
  int node_offset;          // index of this shard's first node in the whole cluster
  const int* forced;        // kRoundAssume: [round] the node (of this shard) an ask goes to, -1 = none of this shard's
  const int* run_len;       // kRoundAssume: [round] asks from this one on that go to the same node with the same spec (a run the host accepted
                            // at once: assumed in one step; inside one 64-ask header window); null: 1
  i64* prof;                // YKPRED_TUNE round_prof=1: thread 0's 100 MHz ticks per phase of the loop (null: off)
  // kRoundAssume on a sharded engine with topology signatures: what the assume added to the histograms, for the OTHER shards (they
  // hold the same cluster-wide histograms and cannot see this shard's node): [round][kDeltaStride] ints — word 0 = entries, then
  // (constraint, domain, count) triples. null: not recorded.
  int* delta;
};
constexpr int kDeltaMax = 10;                   // histogram cells one assumed pod can move (constraints whose selector class it adds to)
constexpr int kDeltaStride = 2 + 3 * kDeltaMax; // ints per ask of the delta record (word 1: pad)
// Per-phase ticks of the loop (thread 0, 100 MHz): compiled in with -DYK_ROUND_PROF only (build.py: YK_ROUND_PROF=1) — the kernel
// has no scalar register to spare for a pointer it does not use.
#ifdef YK_ROUND_PROF
#define YK_RP(k)                                  \
  if (a.prof && tid == 0) {                       \
    const i64 now_ = (i64)wall_clock64();         \
    a.prof[k] += now_ - rp_prev;                  \
    rp_prev = now_;                               \
  }
#define YK_RP_COUNT(k) \
  if (a.prof && tid == 0) a.prof[k] += 1;
#else
#define YK_RP(k)
#define YK_RP_COUNT(k)
#endif
__device__ __forceinline__ void load_node_live(const NodeTable& t, const RoundArgs& a, int n, NodeRegs* r) {
  load_node(t, n, r);  // immutable columns (the stale-prone ones are overwritten below)
#pragma unroll
  for (int i = 0; i < kMaxR; ++i)
    if (i < t.R) r->fr[i] = t.alloc[(size_t)i * t.n + n] - ld_live(t.req + (size_t)i * t.n + n);
  r->slots_ok = (i64)ld_live(t.count + n) + 1 <= (i64)t.allowed[n];
  if (a.ports) {
#pragma unroll
    for (int i = 0; i < kMaxKP; ++i)
      if (i < t.KP) r->pt[i] = ld_live(a.ports + (size_t)i * t.n + n);
  }
}
// the same registers from the slot columns of a moved node
__device__ __forceinline__ void load_slot_live(const NodeTable& t, const RoundArgs& a, int slot, bool with_domains, NodeRegs* r) {
  const size_t c = (size_t)a.cap;
#pragma unroll
  for (int i = 0; i < kMaxR; ++i) r->fr[i] = i < t.R ? ld_live(a.m_free + (size_t)i * c + slot) : 0;
#pragma unroll
  for (int i = 0; i < kMaxKT; ++i) r->tn[i] = i < t.KT ? a.m_taint[(size_t)i * c + slot] : 0;
#pragma unroll
  for (int i = 0; i < kMaxW; ++i) r->lb[i] = i < t.W ? a.m_label[(size_t)i * c + slot] : 0;
  r->lb_more = t.W > kMaxW ? t.labels + (size_t)kMaxW * t.n + a.m_node[slot] : nullptr;
  r->lb_stride = (size_t)t.n;
#pragma unroll
  for (int i = 0; i < kMaxKP; ++i) r->pt[i] = i < t.KP ? ld_live(a.m_ports + (size_t)i * c + slot) : 0;
#pragma unroll
  for (int i = 0; i < kMaxKD; ++i) r->dom[i] = (with_domains && i < t.KD) ? a.m_dom[(size_t)i * c + slot] : -1;
  r->slots_ok = ld_live(a.m_room + slot) >= 1;
  r->unsched = a.m_flags[slot] & kNodeUnschedulable;
}
// The FLAT form of the scan over the moved nodes — tables of the usual width (kFlat* below), spec without topology signature:
// the same registers WITHOUT a branch between the loads. A column past the table's width re-reads the last real column and is
// masked afterwards, so that every load of a slot sits in one basic block — one round trip per step of the scan instead of one
// per column group (the compiler waits at every branch that guards a load). Wider tables take load_slot_live + eval_pair.
constexpr int kFlatR = 4, kFlatKT = 1, kFlatW = 4, kFlatKP = 1;
struct SlotFlat {
  i64 fr[kFlatR];
  u64 tn[kFlatKT], lb[kFlatW], pt[kFlatKP];
  bool slots_ok, unsched;
};
__device__ __forceinline__ void load_slot_flat(const NodeTable& t, const RoundArgs& a, int slot, SlotFlat* r, int* m, u64* key, int* tie) {
  const size_t c = (size_t)a.cap;
  const int R1 = t.R - 1, KT1 = max(t.KT, 1) - 1, W1 = max(t.W, 1) - 1, KP1 = max(t.KP, 1) - 1;
  i64 fr[kFlatR];
  u64 tn[kFlatKT], lb[kFlatW], pt[kFlatKP];
#pragma unroll
  for (int i = 0; i < kFlatR; ++i) fr[i] = ld_live(a.m_free + (size_t)min(i, R1) * c + slot);
#pragma unroll
  for (int i = 0; i < kFlatKT; ++i) tn[i] = a.m_taint[(size_t)min(i, KT1) * c + slot];
#pragma unroll
  for (int i = 0; i < kFlatW; ++i) lb[i] = a.m_label[(size_t)min(i, W1) * c + slot];
#pragma unroll
  for (int i = 0; i < kFlatKP; ++i) pt[i] = ld_live(a.m_ports + (size_t)min(i, KP1) * c + slot);
  const int room = ld_live(a.m_room + slot);
  const unsigned fl = a.m_flags[slot];
  *m = a.m_node[slot];
  *key = ld_live(a.m_key + slot);
  *tie = a.m_tie[slot];
#pragma unroll
  for (int i = 0; i < kFlatR; ++i) r->fr[i] = i < t.R ? fr[i] : 0;
#pragma unroll
  for (int i = 0; i < kFlatKT; ++i) r->tn[i] = i < t.KT ? tn[i] : 0;
#pragma unroll
  for (int i = 0; i < kFlatW; ++i) r->lb[i] = i < t.W ? lb[i] : 0;
#pragma unroll
  for (int i = 0; i < kFlatKP; ++i) r->pt[i] = i < t.KP ? pt[i] : 0;
  r->slots_ok = room >= 1;
  r->unsched = fl & kNodeUnschedulable;
}
__device__ __forceinline__ u64 readlane64(u64 v, int l) {
  return ((u64)(unsigned)__builtin_amdgcn_readlane((int)(v >> 32), l) << 32) | (u64)(unsigned)__builtin_amdgcn_readlane((int)v, l);
}
// Minimum of an unsigned over the 64 lanes of the wave, in every lane: the DPP row-shift / row-broadcast ladder of wave_sum_to_lane63
// with min for + (lanes without a source keep the identity), then one readlane — VALU only, no trip through the LDS crossbar.
__device__ __forceinline__ unsigned wave_umin(unsigned v) {
  const int id = -1;  // 0xffffffff
  unsigned m = min(v, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x111, 0xf, 0xf, false));  // row_shr:1
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x112, 0xf, 0xf, false));           // row_shr:2
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)v, 0x113, 0xf, 0xf, false));           // row_shr:3
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)m, 0x114, 0xf, 0xe, false));           // row_shr:4, lanes 4..15 of every row
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)m, 0x118, 0xf, 0xc, false));           // row_shr:8, lanes 8..15: lane 15 holds the row's
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)m, 0x142, 0xa, 0xf, false));           // row_bcast:15 into rows 1 and 3
  m = min(m, (unsigned)__builtin_amdgcn_update_dpp(id, (int)m, 0x143, 0xc, 0xf, false));           // row_bcast:31 into rows 2 and 3
  return (unsigned)__builtin_amdgcn_readlane((int)m, 63);
}
// dnf_match over terms held in LANE registers: lane j of `tv` = word j of the spec's term rows (term t, word w at lane t * W + w;
// nt * W <= 64, W <= kFlatW). No memory behind the slot loads: the mask words come out of the register file.
__device__ __forceinline__ bool dnf_match_lanes(u64 tv, int nt, const u64 (&lb)[kFlatW], int W) {
  bool any = false;
  for (int tt = 0; tt < nt; ++tt) {
    bool all = true;
#pragma unroll
    for (int w = 0; w < kFlatW; ++w)
      if (w < W) {
        const u64 mw = readlane64(tv, tt * W + w);
        all = all && (lb[w] & mw) == mw;
      }
    any = any || all;
  }
  return any;
}
// The plane rows of a class as the round's scan reads them: kDescRows row pointers (the last real one repeated — an AND with a
// duplicate changes nothing, and the loads need no branch), then [kDescRows] = rows | first word << 32. rows = 0xffffffff: the class
// has index rows or more plane rows than fit — it takes class_rows() per ask.
constexpr int kDescRows = 8, kDescWords = kDescRows + 2;
__global__ __launch_bounds__(kBlock) void k_round_class_desc(ClassTable ct, Planes ranked, int n_classes, u64* __restrict__ desc) {
  const int cls = blockIdx.x * kBlock + threadIdx.x;
  if (cls >= n_classes) return;
  const ClassRows cr = class_rows(ranked, ct.sig[cls * 4 + 0], ct.sig[cls * 4 + 1], ct.sig[cls * 4 + 2], -1);
  u64* d = desc + (size_t)cls * kDescWords;
  const bool slow = cr.ni > 0 || cr.n > kDescRows;
  const u64* last = nullptr;
#pragma unroll
  for (int i = 0; i < kMaxClassRows; ++i)
    if (i < cr.n) last = cr.row[i];
#pragma unroll
  for (int i = 0; i < kDescRows; ++i) d[i] = (u64)(i < cr.n ? cr.row[i] : last);
  d[kDescRows] = (u64)(slow ? 0xffffffffu : (unsigned)cr.n) | ((u64)(unsigned)cr.start << 32);
  d[kDescRows + 1] = 0;
}
// What the Filter list reads of a SPEC, loaded once per ask: the scan over the moved nodes evaluates one spec against many
// slots, and eval_pair's loads sit behind its early exits — a chain of dependent round trips per pair.
struct SpecRegs {
  unsigned f;
  u64 tol[kFlatKT];
  i64 req[kFlatR];
  u64 want[kFlatKP];
  int pre_b, pre_e, term_b, term_e;
};
__device__ __forceinline__ void load_spec_regs(const SpecTable& s, int spec, SpecRegs* q) {
  q->f = s.flags[spec];
#pragma unroll
  for (int k = 0; k < kFlatKT; ++k) q->tol[k] = k < s.KT ? s.tol[(size_t)spec * s.KT + k] : ~0ull;
#pragma unroll
  for (int k = 0; k < kFlatR; ++k) q->req[k] = k < s.R ? s.req[(size_t)spec * s.R + k] : 0;
#pragma unroll
  for (int k = 0; k < kFlatKP; ++k) q->want[k] = k < s.KP ? s.wanted_ports[(size_t)spec * s.KP + k] : 0ull;
  q->pre_b = s.aff.pre_off[spec];
  q->pre_e = s.aff.pre_off[spec + 1];
  q->term_b = s.aff.term_off[spec];
  q->term_e = s.aff.term_off[spec + 1];
}
// eval_pair's VERDICT (fit or not; no failing plugin) for an ask without pin whose spec has no topology signature, from the
// registers above and without early exits: the same conditions in the same plugin order (predicate_manager.go:221-283).
// tv / pv: the spec's Filter terms / PreFilter names in lane registers, see dnf_match_lanes.
__device__ __forceinline__ bool spec_fits_node(int W, const SpecRegs& q, const SlotFlat& nr, unsigned pre_mask, unsigned filt_mask, u64 tv, u64 pv) {
  const unsigned f = q.f;
  bool ok = !(f & kSpecUnsupported);
  if ((pre_mask & kPlugAffinity) && !(f & kSpecAffSkip)) {
    if (f & kSpecPreReject) ok = false;
    else if (f & kSpecPreNames) ok = dnf_match_lanes(pv, q.pre_e - q.pre_b, nr.lb, W) && ok;
  }
  if ((filt_mask & kPlugUnsched) && nr.unsched && !(f & kSpecToleratesUnsched)) ok = false;
  if (filt_mask & kPlugTaint) {
#pragma unroll
    for (int k = 0; k < kFlatKT; ++k) ok = ok && (nr.tn[k] & ~q.tol[k]) == 0;
  }
  if (filt_mask & kPlugAffinity) {
    const bool skip = (pre_mask & kPlugAffinity) && (f & kSpecAffSkip);
    if (!skip) ok = dnf_match_lanes(tv, q.term_e - q.term_b, nr.lb, W) && ok;
  }
  if (filt_mask & kPlugPorts) {
    bool any = false, conflict = false;
#pragma unroll
    for (int k = 0; k < kFlatKP; ++k) {
      any = any || q.want[k] != 0;
      conflict = conflict || (nr.pt[k] & q.want[k]) != 0;
    }
    if (!(pre_mask & kPlugPorts) || (any && conflict)) ok = false;
  }
  if (filt_mask & kPlugFit) {
    if (!(pre_mask & kPlugFit) || !nr.slots_ok) ok = false;
#pragma unroll
    for (int r = 0; r < kFlatR; ++r)
      if (q.req[r] > 0 && q.req[r] > nr.fr[r]) ok = false;
  }
  if ((filt_mask & kPlugSpread) && !(pre_mask & kPlugSpread)) ok = false;
  if ((filt_mask & kPlugInterPod) && !(pre_mask & kPlugInterPod)) ok = false;
  return ok;
}
// score keys and NodeID ranks in bin-pack order (see RoundArgs::rkey)
__global__ __launch_bounds__(kBlock) void k_round_ranked_keys(int n_nodes, const int* __restrict__ perm, const u64* __restrict__ key0,
                                                              const int* __restrict__ name_rank, u64* __restrict__ rkey, int* __restrict__ rtie) {
  const int pos = blockIdx.x * kBlock + threadIdx.x;
  if (pos >= n_nodes) return;
  const int n = perm[pos];
  rkey[pos] = key0[n];
  rtie[pos] = name_rank ? name_rank[n] : n;
}
// one wave per constraint: what the round needs to keep a spread constraint's minimum current (k_spread_min's numbers, unfolded)
__global__ __launch_bounds__(kWave) void k_round_topo_init(SpreadSigs sp, int n_constraints, int* __restrict__ mn_out, int* __restrict__ at_min_out,
                                                           int* __restrict__ nd_out) {
  const int g = blockIdx.x;
  if (g >= n_constraints) return;
  const SpreadC c = sp.c[g];
  int mn = 0x7fffffff, nd = 0;
  for (int i = threadIdx.x; i < c.dom_size; i += kWave)
    if (sp.present[c.cnt_off + i]) {
      mn = min(mn, sp.cnt[c.cnt_off + i]);
      ++nd;
    }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    mn = min(mn, __shfl_xor(mn, off, kWave));
    nd += __shfl_xor(nd, off, kWave);
  }
  int at = 0;
  for (int i = threadIdx.x; i < c.dom_size; i += kWave) at += (sp.present[c.cnt_off + i] && sp.cnt[c.cnt_off + i] == mn) ? 1 : 0;
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) at += __shfl_xor(at, off, kWave);
  if (threadIdx.x == 0) {
    mn_out[g] = mn;
    at_min_out[g] = at;
    nd_out[g] = nd;
  }
}
// The kernel's tables are a hundred pointers. Taken as ordinary arguments they are all loaded at entry and stay live through the
// loop: twice the scalar register file, spilled to vector lanes and those to scratch — every reload a full wait in the middle of a
// load round. So the loop reads them where it uses them, from the kernel-argument segment itself (scalar loads, constant cache),
// through a pointer the compiler has to take as new at the head of every phase (YK_CTX_FRESH): nothing is carried across phases.
struct RoundCtx {
  NodeTable nodes;
  SpecTable specs;
  ClassTable classes;
  Planes planes;
  RoundArgs args;
};
static_assert(sizeof(RoundCtx) <= 3584, "the round's tables travel in the kernel-argument segment");
typedef const RoundCtx __attribute__((address_space(4))) * RoundCtxPtr;
#define YK_CTX_FRESH() asm volatile("" : "+s"(cx))
#define t (*(const NodeTable*)&cx->nodes)
#define s (*(const SpecTable*)&cx->specs)
#define ct (*(const ClassTable*)&cx->classes)
#define ranked (*(const Planes*)&cx->planes)
#define a (*(const RoundArgs*)&cx->args)
#define sp (*(const SpreadSigs*)&cx->specs.spread)
// A sharded round with topology signatures: the histogram steps of the asks OTHER shards assumed in this batch (their delta records,
// all-gathered), applied to this shard's copy of the cluster-wide histograms. One workgroup, ask after ask, entry after entry: the
// cell, then — for a spread constraint — its minimum over the present domains, how many domains sit there and the global minimum
// after the minDomains rule, recomputed whole (the same numbers k_allocate_round's assume maintains incrementally; a step of an
// InterPodAffinity term counts a domain that got its first match).
__global__ __launch_bounds__(kBlock) void k_round_apply_deltas(SpreadSigs hist, int* __restrict__ mn_out, int* __restrict__ at_min_out,
                                                               const int* __restrict__ nd, const int* __restrict__ recs, int n_asks) {
  __shared__ int sh_mn[kWavesPerBlock], sh_at[kWavesPerBlock];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  for (int i = 0; i < n_asks; ++i) {
    const int* rec = recs + (size_t)i * kDeltaStride;
    const int n = min(rec[0], kDeltaMax);
    for (int k = 0; k < n; ++k) {
      const int g = rec[2 + 3 * k], dom = rec[3 + 3 * k], v = rec[4 + 3 * k];
      const SpreadC c = hist.c[g];
      const int old = hist.cnt[c.cnt_off + dom];
      __syncthreads();  // (everybody has read the old count)
      if (tid == 0) hist.cnt[c.cnt_off + dom] = old + v;
      __syncthreads();
      if (c.kind == kKindSpread) {
        int mn = 0x7fffffff;
        for (int d = tid; d < c.dom_size; d += kBlock)
          if (hist.present[c.cnt_off + d]) mn = min(mn, hist.cnt[c.cnt_off + d]);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, __shfl_xor(mn, off, kWave));
        if (lane == 0) sh_mn[wave] = mn;
        __syncthreads();
        mn = sh_mn[0];
#pragma unroll
        for (int q = 1; q < kWavesPerBlock; ++q) mn = min(mn, sh_mn[q]);
        int at = 0;
        for (int d = tid; d < c.dom_size; d += kBlock) at += (hist.present[c.cnt_off + d] && hist.cnt[c.cnt_off + d] == mn) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) at += __shfl_xor(at, off, kWave);
        if (lane == 0) sh_at[wave] = at;
        __syncthreads();
        if (tid == 0) {
          at = 0;
          for (int q = 0; q < kWavesPerBlock; ++q) at += sh_at[q];
          mn_out[g] = mn;
          at_min_out[g] = at;
          hist.minv[g] = nd[g] < c.min_domains ? 0 : mn;
        }
        __syncthreads();
      } else if (tid == 0 && old <= 0 && old + v > 0) {
        hist.minv[g] = hist.minv[g] + 1;
      }
      __syncthreads();
    }
  }
}
__global__ __launch_bounds__(kRoundThreads) void k_allocate_round(RoundCtx by_value) {
  RoundCtxPtr cx = (RoundCtxPtr)__builtin_amdgcn_kernarg_segment_ptr();
  __shared__ int sh_aw[kRoundWaves], sh_cw[kRoundWaves], sh_bt[kRoundWaves], sh_bn[kRoundWaves], sh_an[kRoundWaves], sh_at[kRoundWaves];
  __shared__ u64 sh_ak[kRoundWaves], sh_bk[kRoundWaves];
  __shared__ int sh_step, sh_was_moved;  // what wave 0's assume decided: asks of the run, the node had a slot already
  // (flags of one ask; two sets, used alternately: thread 0 re-arms the set of ask i + 1 while ask i runs — nobody touches it then)
  __shared__ int sh_stop_[2], sh_ndirty_[2], sh_dirty[kRoundDirty];
  __shared__ int sh_rmn[kRoundWaves], sh_rat[kRoundWaves];
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  int n_moved = ld_live(a.n_moved);
  const bool name_on = a.filt & kPlugNodeName;
  const bool fit_on = (a.filt & kPlugFit) && (a.pre & kPlugFit);
  const bool spread_en = a.filt & kPlugSpread, ipa_en = (a.filt & kPlugInterPod) && (a.pre & kPlugInterPod);
  const size_t cap = (size_t)a.cap;
  // (what every ask reads of the launch: once, into registers — behind the per-phase refresh each would be a trip to the constant cache)
  const int n_asks = a.n_asks, first_ask = a.first, launch_mode = a.mode;
  const bool all_fail = a.all_fail != 0;
  const int* const spread_sig = a.topo_on ? s.spread_sig : nullptr;
  int p_l = 0, spec_l = 0, pin_l = -1, cls_l = 0;
  int last_spec = -1, last_win = -1;  // (per launch: the first ask of a launch takes the scans)
  if (tid == 0) {
    sh_stop_[0] = 0x7fffffff;
    sh_ndirty_[0] = 0;
  }
  __syncthreads();
  int step = 0;  // (asks decided by an iteration: runs of one spec that land on one node are decided together)
#ifdef YK_ROUND_PROF
  i64 rp_prev = a.prof ? (i64)wall_clock64() : 0;
  const i64 rp_c0 = (i64)clock64(), rp_w0 = (i64)wall_clock64();
#endif
  for (int i = 0; i < n_asks; i += step) {
    YK_CTX_FRESH();
    YK_RP_COUNT(8)
    int& sh_stop = sh_stop_[i & 1];
    int& sh_ndirty = sh_ndirty_[i & 1];
    step = 1;
    if ((i & (kWave - 1)) == 0) {  // the headers of the next 64 asks, one load round per wave
      const int j = i + lane;
      if (j < n_asks) {
        p_l = a.asks[first_ask + j];
        spec_l = a.pod_spec[p_l];
        pin_l = name_on ? a.pod_pin[p_l] : -1;
        cls_l = a.pod_class[p_l];
      } else {
        spec_l = -1;
      }
    }
    const int hl = i & (kWave - 1);
    const int spec = __builtin_amdgcn_readlane(spec_l, hl), cls = __builtin_amdgcn_readlane(cls_l, hl);
    const int pin = __builtin_amdgcn_readlane(pin_l, hl);
    const int tsig = spread_sig ? spread_sig[spec] : -1;  // the spec's topology signature: its verdicts move with every assume
    // (the flag set of the NEXT ask, whichever it will be: an iteration advances by `step`, and both parities may follow)
    if (tid == 0) {
      sh_stop_[(i + 1) & 1] = 0x7fffffff;
      sh_ndirty_[(i + 1) & 1] = 0;
    }
    int win = -1;
    bool again = false;
    YK_RP(0)
    const int mode = launch_mode;
    if (mode == kRoundDecide && !all_fail && pin == -1 && spec == last_spec && last_win >= 0 && tsig < 0) {
      // The same spec as the ask before, which went to node last_win: that node is AT LEAST as early in the bin-pack order now
      // (an allocation only raises a node's utilisation, i.e. lowers its score; every other node stands where it stood), so it
      // is this ask's node too as long as it still fits — one pair instead of the two scans. Bin-packing sends long runs of a
      // Deployment's or task group's asks to one node: this is the common case. (Not for a spec with a topology signature: the
      // assume moved the verdicts of OTHER nodes — an earlier node may have become feasible.)
      NodeRegs nr;
      load_node_live(t, a, last_win, &nr);
      int code;
      unsigned reason;
      again = eval_pair<true>(s, spec, -1, last_win, nr, a.pre, a.filt, &code, &reason);
    }
    YK_RP(1)
    if (again) {
      win = last_win;
      YK_RP_COUNT(9)
    } else if (mode == kRoundAssume) {
      win = a.forced[first_ask + i];  // (decided by the exchange: the scans are somebody else's)
    } else if (all_fail || pin == -2) {
      // a Filter without its PreFilter state / spec.nodeName names no node of the table: nothing fits
    } else if (pin >= 0) {
      NodeRegs nr;
      load_node_live(t, a, pin, &nr);
      int code;
      unsigned reason;
      if (eval_pair<true>(s, spec, pin, pin, nr, a.pre, a.filt, &code, &reason)) win = pin;
      YK_RP(2)
    } else {
      YK_CTX_FRESH();
      // ---- candidate A: the first unmoved feasible node in snapshot order. aw / ax: the first word of THIS wave with a feasible
      // node and its bits; cw: the first word of this wave with a candidate of the snapshot planes (the class's cursor)
      int aw = 0x7fffffff, cw = 0x7fffffff;
      u64 ax = 0;
      // (requested before the scan, used behind it: the spec's registers and the bitset words of the first 64 steps of candidate B)
      SpecRegs q;
      u64 tv_l = 0, pv_l = 0;
      bool flat = tsig < 0 && t.R <= kFlatR && t.KT <= kFlatKT && t.W <= kFlatW && t.KP <= kFlatKP;  // (the tables are of the usual width)
      // (the class's descriptor: every lane reads the same words — they stay in vector registers, the scalar file is full of table pointers)
      const u64* dsc = a.cdesc + (size_t)cls * kDescWords;
      typedef const u64 __attribute__((address_space(1))) * GlobalRow;  // (a pointer that comes out of memory: said to be global, else the loads are flat)
      GlobalRow dr[kDescRows];
#pragma unroll
      for (int k = 0; k < kDescRows; ++k) dr[k] = (GlobalRow)dsc[k];
      const u64 meta = dsc[kDescRows];
      int w0 = ld_live(a.cursor + cls);
      if (flat) {
        load_spec_regs(s, spec, &q);
        const int nt = (q.term_e - q.term_b) * s.W, np = (q.pre_e - q.pre_b) * s.W;
        flat = nt <= kWave && np <= kWave;  // (the spec's term rows fit the lanes of a wave)
        if (flat) {  // (a lane past the spec's words re-reads its last one — never consulted)
          tv_l = nt > 0 ? s.aff.terms[(size_t)q.term_b * s.W + min(lane, nt - 1)] : 0ull;
          pv_l = np > 0 ? s.aff.pre_terms[(size_t)q.pre_b * s.W + min(lane, np - 1)] : 0ull;
        }
      }
      u64* fw = (a.failed && tsig < 0) ? a.failed + (size_t)cls * a.cap64 : nullptr;
      u64 fb_l = 0, dead_l = ~0ull;  // lane k: this wave's bitset words of step (j0 / kRoundThreads) % 64 == k, one load round for 64 steps
      {
        const int wl = lane * kRoundWaves + wave;
        const bool in = wl * kWave < n_moved;
        fb_l = (in && fw) ? ld_live(fw + wl) : 0ull;
        dead_l = in ? ld_live(a.dead + wl) : ~0ull;
      }
      {
        // (ranked.spread is null: the topology family is checked live)
        const int dn = (int)(unsigned)meta;  // plane rows of the class (-1: resolved here, per ask)
        ClassRows cr;
        cr.n = 0;
        cr.ni = 0;
        cr.start = (int)(meta >> 32);
        if (dn < 0) cr = class_rows(ranked, ct.sig[cls * 4 + 0], ct.sig[cls * 4 + 1], ct.sig[cls * 4 + 2], -1);
        if (w0 < 0) w0 = cr.start;  // (kNoWord: some row of the class is empty)
        const int row_words = a.row_words;
        const u64* const moved_bits = a.moved_bits;
        for (int base = w0 < row_words ? (w0 & ~(kWave - 1)) : row_words; base < row_words && aw == 0x7fffffff; base += kRoundThreads) {
          const int w = base + tid;
          u64 x = 0;
          if (w >= w0 && w < row_words) {
            const u64 mv = ld_live(moved_bits + w);
            u64 v = ~0ull;
            if (dn > 0) {  // (one load round: the moved bits and every row of the class)
#pragma unroll
              for (int k = 0; k < kDescRows; ++k) v &= dr[k][w];
            } else if (dn < 0) {
              v = class_word(cr, w);
            }
            x = ~mv & v;
          }
          u64 todo = __ballot(x != 0);
          if (!todo) continue;  // (wave-uniform)
          const int wave_w = base + wave * kWave;
          if (cw == 0x7fffffff) cw = wave_w + __ffsll((long long)todo) - 1;
          if (tsig < 0) {
            const int fl = __ffsll((long long)todo) - 1;
            aw = wave_w + fl;
            ax = readlane64(x, fl);
          } else {
            // the class's topology constraints against the live histograms, word after word, lane = node; a wave gives up once
            // another one has found a node in an earlier word
            while (todo) {
              const int fl = __ffsll((long long)todo) - 1;
              todo &= todo - 1;
              const int ww = wave_w + fl;
              if (__builtin_amdgcn_readfirstlane(*(volatile int*)&sh_stop) < ww) break;
              const u64 xw = readlane64(x, fl);
              bool ok = false;
              if ((xw >> lane) & 1ull) {
                const int n = a.perm[ww * kWave + lane];
                int dom[kMaxKD];
#pragma unroll
                for (int k = 0; k < kMaxKD; ++k) dom[k] = k < t.KD ? t.domain[(size_t)k * t.n + n] : -1;
                ok = constraints_fail<true>(sp, tsig, dom, spread_en, ipa_en, nullptr) == 0;
              }
              const u64 y = __ballot(ok);
              if (y) {
                aw = ww;
                ax = y;
                if (lane == 0) atomicMin(&sh_stop, ww);
                break;
              }
            }
          }
        }
      }
      // the node behind the found position, its key and NodeID rank: in flight while this wave scans the moved nodes
      int an_w = -1, at_w = 0;
      u64 ak_w = 0;
      if (aw != 0x7fffffff) {
        const int pos = aw * kWave + (__ffsll((long long)ax) - 1);
        an_w = a.perm[pos];
        ak_w = a.rkey[pos];
        at_w = a.rtie[pos];
      }
      YK_RP(3)
      YK_CTX_FRESH();
      // ---- candidate B: the best moved node, per pair from the live slot columns. A wave's 64 slots are one word of the bitsets:
      // slots of dead nodes and — for a class without topology signature — slots the class failed on before are skipped without
      // a load; what it fails on now is remembered.
      u64 bk = ~0ull;
      int bt = 0x7fffffff, bn = -1;
      // One step of the scan: 512 slots, a wave per bitset word. EVAL(slot) sets fit / m / k / tie of an active lane.
#define YK_SCAN_MOVED(DEAD, EVAL)                                                                              \
      for (int j0 = 0; j0 < n_moved; j0 += kRoundThreads) {                                                        \
        const int st_k = (j0 / kRoundThreads) & (kWave - 1);                                                       \
        if (st_k == 0 && j0 > 0) {                                                                                 \
          const int wl = ((j0 + lane * kRoundThreads) >> 6) + wave;                                                \
          const bool in = wl * kWave < n_moved;                                                                    \
          fb_l = (in && fw) ? ld_live(fw + wl) : 0ull;                                                             \
          dead_l = in ? ld_live((DEAD) + wl) : ~0ull;                                                              \
        }                                                                                                          \
        const int wi = (j0 >> 6) + wave;                                                                           \
        const int slot = j0 + tid;                                                                                 \
        if (wi * kWave >= n_moved) continue; /* (wave-uniform) */                                                  \
        const u64 fbits = readlane64(fb_l, st_k);                                                                  \
        const u64 skip = fbits | readlane64(dead_l, st_k);                                                         \
        const bool act = slot < n_moved && !((skip >> lane) & 1ull);                                               \
        if (__ballot(act) == 0) continue;                                                                          \
        \
        bool fit = false;                                                                                          \
        if (act) {                                                                                                 \
          int m, tie;                                                                                              \
          u64 k; /* (the key with the columns, not behind the verdict: one round trip per step) */                 \
          EVAL                                                                                                     \
          if (fit && (k < bk || (k == bk && tie < bt))) {                                                          \
            bk = k;                                                                                                \
            bt = tie;                                                                                              \
            bn = m;                                                                                                \
          }                                                                                                        \
        }                                                                                                          \
        if (fw) {                                                                                                  \
          const u64 nf = __ballot(act && !fit);                                                                    \
          if (nf && lane == 0) st_live(fw + wi, fbits | nf); /* (this wave is the only writer of the word during this ask) */ \
        }                                                                                                          \
      }
      if (flat) {
        // (what the scan reads of the tables, fetched here in one batch of scalar loads — inside the loop every pointer would be its
        // own trip to the constant cache in front of the load it addresses)
        RoundArgs ab;
        ab.cap = a.cap;
        ab.m_free = a.m_free;
        ab.m_taint = a.m_taint;
        ab.m_label = a.m_label;
        ab.m_ports = a.m_ports;
        ab.m_room = a.m_room;
        ab.m_flags = a.m_flags;
        ab.m_node = a.m_node;
        ab.m_key = a.m_key;
        ab.m_tie = a.m_tie;
        NodeTable tb;
        tb.R = t.R;
        tb.KT = t.KT;
        tb.W = t.W;
        tb.KP = t.KP;
        const u64* const dead_p = a.dead;
        const unsigned pre_m = a.pre, filt_m = a.filt;
        YK_SCAN_MOVED(dead_p, {
          SlotFlat nr;
          load_slot_flat(tb, ab, slot, &nr, &m, &k, &tie);
          fit = spec_fits_node(tb.W, q, nr, pre_m, filt_m, tv_l, pv_l);
        })
      } else {
        YK_SCAN_MOVED(a.dead, {
          NodeRegs nr;
          load_slot_live(t, a, slot, tsig >= 0, &nr);
          m = a.m_node[slot];
          k = ld_live(a.m_key + slot);
          tie = a.m_tie[slot];
          int code;
          unsigned reason;
          fit = eval_pair<true>(s, spec, -1, m, nr, a.pre, a.filt, &code, &reason);
        })
      }
#undef YK_SCAN_MOVED
      {  // the wave's best: the lexicographic minimum of (key, NodeID rank) over the lanes that hold a candidate
        bool cand = bn >= 0;
        if (__ballot(cand)) {
          const unsigned hi = wave_umin(cand ? (unsigned)(bk >> 32) : 0xffffffffu);
          cand = cand && (unsigned)(bk >> 32) == hi;
          const unsigned lo = wave_umin(cand ? (unsigned)bk : 0xffffffffu);
          cand = cand && (unsigned)bk == lo;
          const unsigned ti = wave_umin(cand ? (unsigned)bt : 0xffffffffu);
          cand = cand && (unsigned)bt == ti;
          const int src = __ffsll((long long)__ballot(cand)) - 1;
          bk = ((u64)hi << 32) | lo;
          bt = (int)ti;
          bn = __builtin_amdgcn_readlane(bn, src);
        }
      }
      YK_RP(4)
      YK_CTX_FRESH();
      // ---- the waves' candidates meet in LDS; every thread reduces them (the result is workgroup-uniform)
      if (lane == 0) {
        sh_aw[wave] = aw;
        sh_an[wave] = an_w;
        sh_ak[wave] = ak_w;
        sh_at[wave] = at_w;
        sh_cw[wave] = cw;
        sh_bk[wave] = bk;
        sh_bt[wave] = bt;
        sh_bn[wave] = bn;
      }
      __syncthreads();
      int gaw = 0x7fffffff, gcw = 0x7fffffff;
      int an = -1, at = 0;
      u64 ak = 0;
      bk = ~0ull;
      bt = 0x7fffffff;
      bn = -1;
#pragma unroll
      for (int k = 0; k < kRoundWaves; ++k) {
        if (sh_aw[k] < gaw) {
          gaw = sh_aw[k];
          an = sh_an[k];
          ak = sh_ak[k];
          at = sh_at[k];
        }
        gcw = min(gcw, sh_cw[k]);
        const u64 ok = sh_bk[k];
        const int ot = sh_bt[k], on = sh_bn[k];
        if (on >= 0 && (bn < 0 || ok < bk || (ok == bk && ot < bt))) {
          bk = ok;
          bt = ot;
          bn = on;
        }
      }
      // (a wave that stopped early had nothing in front of the word another wave found: the minimum over the waves is exact)
      if (tid == 0) st_live(a.cursor + cls, gcw < a.row_words ? gcw : a.row_words);
      win = (an >= 0 && (bn < 0 || ak < bk || (ak == bk && at < bt))) ? an : bn;
      YK_RP(5)
    }
    last_spec = pin == -1 ? spec : -1;
    last_win = win;
    YK_CTX_FRESH();
    if (win >= 0) {  // (workgroup-uniform)
      // ---- AssumePod on the scratch state: Requested += the ask's request vector, len(Pods) += 1 (NodeInfo.AddPod).
      // A RUN of asks with this spec lands on this node as long as it fits (the argument of the `again` path), and for a spec whose
      // pods couple through nothing but resources "fits k more times" is arithmetic: k = what the free resources and pod slots
      // hold. The run is decided here in one step (the asks of a Deployment or task group: 110 per node in the reference's perf
      // shape) — the headers in this wave's registers bound it to the asks up to the next multiple of 64.
      // The assume is WAVE 0's: everything it reads of the node, the spec and the round's bookkeeping is requested in ONE load round
      // (lane r = resource r; lanes 8.. = the static columns a node brings along when it takes a slot), then stored; the other
      // waves go on to the topology counts and meet it at the barrier.
      const bool contributes = a.topo_on && a.fx.off && a.fx.off[spec + 1] > a.fx.off[spec];
      if (wave == 0) {
        const bool lr = lane < t.R, lp = lane < t.KP;
        const int rl = min(lane, t.R - 1);  // (lanes past the resources re-read the last one: no branch around the loads)
        const i64 rq_raw = s.req[(size_t)spec * s.R + rl], al_raw = t.alloc[(size_t)rl * t.n + win], old_raw = ld_live(a.req + (size_t)rl * t.n + win);
        const i64 rq_l = lr ? rq_raw : 0, al_l = lr ? al_raw : 0, old_l = lr ? old_raw : 0;
        const int cnt0 = ld_live(a.count + win), allowed = t.allowed[win];
        const int slot_prev = ld_live(a.slot_of + win);
        const int rk = a.rank[win], tie_w = a.name_rank ? a.name_rank[win] : win;
        const unsigned flags_w = t.flags[win];
        const u64 occ_l = (lp && a.ports && a.fx.occupied) ? a.fx.occupied[(size_t)spec * t.KP + lane] : 0ull;
        const u64 port_l = lp ? (a.ports ? ld_live(a.ports + (size_t)lane * t.n + win) : t.ports[(size_t)lane * t.n + win]) : 0ull;
        const int c0 = 8, Wc = min(t.W, kMaxW), col = lane - c0;  // (lanes 0..7 hold the resources)
        u64 stat_l = 0;
        int dom_l = -1;
        if (col >= 0 && col < t.KT) stat_l = t.taints[(size_t)col * t.n + win];
        else if (col >= t.KT && col < t.KT + Wc) stat_l = t.labels[(size_t)(col - t.KT) * t.n + win];
        else if (col >= t.KT + Wc && col < t.KT + Wc + t.KD) dom_l = t.domain[(size_t)(col - t.KT - Wc) * t.n + win];
        const bool was_moved = slot_prev >= 0;
        const int slot = was_moved ? slot_prev : n_moved;
        // A RUN of asks with this spec lands on this node as long as it fits (the argument of the `again` path), and for a spec whose
        // pods couple through nothing but resources "fits k more times" is arithmetic: k = what the free resources and pod slots
        // hold. The run is decided here in one step (the asks of a Deployment or task group: 110 per node in the reference's perf
        // shape) — the headers in this wave's registers bound it to the asks up to the next multiple of 64.
        int k_run = 1;
        const bool any_occ = __ballot(occ_l != 0) != 0;  // (a pod that occupies host ports conflicts with its own twin: no run)
        if (mode == kRoundDecide && pin == -1 && tsig < 0 && !any_occ && !contributes && fit_on) {
          // asks hl+1 .. of the header window with the same spec and no pin, consecutively
          const u64 same = __ballot(spec_l == spec && pin_l == -1);
          const u64 behind = hl < 63 ? (~same) >> (hl + 1) : ~0ull;  // first 0 of `same` behind lane hl ends the run
          const int run = behind ? (int)(__ffsll((long long)behind) - 1) : (63 - hl);
          if (run > 0) {  // (the next ask is another spec: nothing to divide)
            i64 fits_l = (lr && rq_l > 0) ? (al_l - old_l) / rq_l : 0x7fffffffffffffffll;
#pragma unroll
            for (int off = kMaxR / 2; off > 0; off >>= 1) fits_l = min(fits_l, (i64)__shfl_xor((long long)fits_l, off, kWave));
            const i64 fits = min((i64)allowed - (i64)cnt0, (i64)__shfl((long long)fits_l, 0, kWave));  // pod slots left (>= 1: the ask fits)
            k_run = (int)max((i64)1, min(fits, (i64)(1 + min(run, 63 - hl))));
            k_run = min(k_run, n_asks - i);
          }
        }
        if (mode == kRoundAssume && a.run_len) k_run = max(1, min(min(a.run_len[first_ask + i], kWave - hl), n_asks - i));
        if (lane < k_run) a.out[first_ask + i + lane] = win;
        const int cnt = cnt0 + k_run;
        const bool dead = fit_on && (i64)cnt + 1 > (i64)allowed;  // no pod slot left: no ask of this phase fits it any more
        const i64 v_l = old_l + rq_l * (i64)k_run;
        if (lr) {
          st_live(a.req + (size_t)lane * t.n + win, v_l);
          st_live(a.m_free + (size_t)lane * cap + slot, al_l - v_l);
        }
        const i64 used[2] = {(i64)__shfl((long long)v_l, 0, kWave), (i64)__shfl((long long)v_l, 1, kWave)};
        const i64 total[2] = {(i64)__shfl((long long)al_l, 0, kWave), (i64)__shfl((long long)al_l, 1, kWave)};
        if (lane == 0) {
          st_live(a.count + win, cnt);
          st_live(a.m_room + slot, allowed - cnt);
          st_live(a.m_key + slot, sortable_key(node_score_of(total, used)));
          if (!was_moved) {
            or_live(a.moved_bits + (rk >> 6), 1ull << (rk & 63));
            st_live(a.slot_of + win, slot);
            a.m_node[slot] = win;
            a.m_tie[slot] = tie_w;
            a.m_flags[slot] = flags_w;
          }
          if (dead) or_live(a.dead + (slot >> 6), 1ull << (slot & 63));
          sh_step = k_run;
          sh_was_moved = was_moved ? 1 : 0;
        }
        // host ports the pod occupies from now on (NodeInfo.UsedPorts)
        if (lp) {
          if (a.ports && occ_l) or_live(a.ports + (size_t)lane * t.n + win, occ_l);
          if (occ_l || !was_moved) st_live(a.m_ports + (size_t)lane * cap + slot, port_l | occ_l);
        }
        // the static columns of a node that has just taken a slot: one lane per column
        if (!was_moved) {
          if (col >= 0 && col < t.KT) a.m_taint[(size_t)col * cap + slot] = stat_l;
          else if (col >= t.KT && col < t.KT + Wc) a.m_label[(size_t)(col - t.KT) * cap + slot] = stat_l;
          else if (col >= t.KT + Wc && col < t.KT + Wc + t.KD) a.m_dom[(size_t)(col - t.KT - Wc) * cap + slot] = dom_l;
        }
      }
      // match counts of the topology plugins: thread = constraint (of any signature) whose selector class the pod adds to
      if (contributes) {
        const int f0 = a.fx.off[spec], f1 = a.fx.off[spec + 1];
        for (int g = tid; g < a.G; g += kRoundThreads) {
          const SpreadC c = sp.c[g];
          int v = 0;
          for (int k = f0; k < f1; ++k)
            if (a.fx.cls[k] == c.ks) v = a.fx.cnt[k];
          if (v == 0) continue;
          const int dom = t.domain[(size_t)c.kd * t.n + win];
          if (dom < 0 || dom >= c.dom_size) continue;
          if (c.kind == kKindSpread && !spread_counts_here(c, spread_eligibility(t, sp, a.sig_aff, a.sig_tol, a.sig_of[g], win))) continue;
          const int old = __hip_atomic_fetch_add(sp.cnt + c.cnt_off + dom, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
          if (a.delta) {  // (sharded round: the other shards apply the same step to their copy — k_round_apply_deltas)
            int* rec = a.delta + (size_t)(first_ask + i) * kDeltaStride;
            const int k = atomicAdd(rec, 1);
            if (k < kDeltaMax) {
              rec[2 + 3 * k] = g;
              rec[3 + 3 * k] = dom;
              rec[4 + 3 * k] = v;
            }
          }
          if (c.kind == kKindSpread) {
            // (an eligible node carries the domain: it is a present one. Its count leaves the minimum; the minimum itself moves
            // only when no present domain is left there)
            if (old == ld_live(a.mn + g)) {
              const int left = ld_live(a.at_min + g) - 1;
              st_live(a.at_min + g, left);
              if (left == 0) {
                const int dslot = atomicAdd(&sh_ndirty, 1);
                if (dslot < kRoundDirty) {
                  sh_dirty[dslot] = g;
                } else {  // (more than a handful at once: this thread recomputes its constraint alone)
                  int mn = 0x7fffffff, at = 0;
                  for (int q = 0; q < c.dom_size; ++q)
                    if (sp.present[c.cnt_off + q]) mn = min(mn, ld_live(sp.cnt + c.cnt_off + q));
                  for (int q = 0; q < c.dom_size; ++q) at += (sp.present[c.cnt_off + q] && ld_live(sp.cnt + c.cnt_off + q) == mn) ? 1 : 0;
                  st_live(a.mn + g, mn);
                  st_live(a.at_min + g, at);
                  st_live(sp.minv + g, a.nd[g] < c.min_domains ? 0 : mn);
                }
              }
            }
          } else if (old <= 0 && old + v > 0) {
            st_live(sp.minv + g, ld_live(sp.minv + g) + 1);  // InterPodAffinity: domains with a match (k_spread_min's `tot`)
          }
        }
      }
      __threadfence_block();
      __syncthreads();
      YK_RP(6)
      step = sh_step;
      if (!sh_was_moved) ++n_moved;
      // spread constraints whose last domain left the minimum: the new minimum, the whole workgroup over the constraint's domains
      const int ndirty = min(sh_ndirty, kRoundDirty);
      for (int q = 0; q < ndirty; ++q) {
        const int g = sh_dirty[q];
        const SpreadC c = sp.c[g];
        int mn = 0x7fffffff;
        for (int d = tid; d < c.dom_size; d += kRoundThreads)
          if (sp.present[c.cnt_off + d]) mn = min(mn, ld_live(sp.cnt + c.cnt_off + d));
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) mn = min(mn, __shfl_xor(mn, off, kWave));
        if (lane == 0) sh_rmn[wave] = mn;
        __syncthreads();
        mn = sh_rmn[0];
#pragma unroll
        for (int k = 1; k < kRoundWaves; ++k) mn = min(mn, sh_rmn[k]);
        int at = 0;
        for (int d = tid; d < c.dom_size; d += kRoundThreads) at += (sp.present[c.cnt_off + d] && ld_live(sp.cnt + c.cnt_off + d) == mn) ? 1 : 0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) at += __shfl_xor(at, off, kWave);
        if (lane == 0) sh_rat[wave] = at;
        __syncthreads();
        if (tid == 0) {
          at = 0;
          for (int k = 0; k < kRoundWaves; ++k) at += sh_rat[k];
          st_live(a.mn + g, mn);
          st_live(a.at_min + g, at);
          st_live(sp.minv + g, a.nd[g] < c.min_domains ? 0 : mn);
        }
      }
      if (ndirty) __threadfence_block();
    } else {
      if (tid == 0) a.out[first_ask + i] = -1;
      if (mode == kRoundAssume && a.run_len) step = max(1, min(min(a.run_len[first_ask + i], kWave - hl), n_asks - i));  // (another shard's run)
    }
    // (the exchange slots are rewritten by the next ask only after this barrier; every wave has read them by now)
    __syncthreads();
    YK_RP(7)
  }
  if (tid == 0) st_live(a.n_moved, n_moved);
#ifdef YK_ROUND_PROF
  if (a.prof && tid == 0) {  // shader clock against the 100 MHz counter
    a.prof[12] += (i64)clock64() - rp_c0;
    a.prof[13] += (i64)wall_clock64() - rp_w0;
  }
#endif
}
#undef t
#undef s
#undef ct
#undef ranked
#undef a
#undef sp
#undef YK_CTX_FRESH

// ---------------------------------------------------------------------------------------------------
// BATCHED rounds (round 6): the asks of a batch proposed IN PARALLEL. k_allocate_round decides ask after ask in one workgroup — the
// asks are a dependency chain. A batch of asks evaluated against ONE frozen state is not: k_round_propose gives every ask of the
// batch its own workgroup (the whole device instead of one compute unit) and finds its kPropK best feasible nodes in
// (score key, NodeID rank) order — unmoved nodes from the class's rank-ordered plane rows, moved nodes per pair from the live slot
// columns, exactly the two candidate scans of k_allocate_round, continued past the first hit. k_round_distinct numbers the distinct
// nodes the batch proposed; k_round_cross evaluates EVERY ask of the batch against every one of them (a bit per pair). With that the
// host can replay the sequential loop over the batch exactly (engine.hip, allocate_round_batched): a node that took pods earlier in
// the batch is one of the numbered nodes — whether a later ask passes its other Filters is its bit, whether it still fits is
// arithmetic on the columns the proposals carry — and a candidate that is full is followed by the next entry of the ask's list.
// Of the round's state only the classes' failed-before bits are written here (set, never cleared); cursors are read, not advanced.
// ---------------------------------------------------------------------------------------------------
#ifndef YK_PROP_K
#define YK_PROP_K 8
#endif
constexpr int kPropK = YK_PROP_K;            // candidates per ask (and shard)
#ifndef YK_PROPOSE_WAVES
#define YK_PROPOSE_WAVES 8
#endif
constexpr int kProposeWaves = YK_PROPOSE_WAVES, kProposeThreads = kProposeWaves * kWave;
constexpr int kDistinctSlots = 4096;         // hash slots of k_round_distinct: at least twice the proposals of a batch
struct PropCand {
  u64 key;
  int tie, node;
};
__device__ __forceinline__ bool cand_less(u64 ka, int ta, u64 kb, int tb) { return ka < kb || (ka == kb && ta < tb); }
// one proposal entry: what the ranks need of node `win` to order it, to re-key it after assumes and to re-check NodeResourcesFit
// (one wave; every lane takes part)
__device__ __forceinline__ void write_proposal(const NodeTable& t, const SpecTable& s, const RoundArgs& a, int spec, int pin, int tsig, bool fit_on,
                                               int win, RoundProposal* out) {
  const int lane = threadIdx.x % kWave;
  if (win < 0) {
    if (lane == 0) {
      RoundProposal pr{};
      pr.key = ~0ull;
      pr.node = -1;
      pr.gnode = -1;
      pr.didx = -1;
      *out = pr;
    }
    return;
  }
  const bool lr = lane < t.R;
  const int rl = min(lane, t.R - 1);
  const i64 rq_raw = s.req[(size_t)spec * s.R + rl], al_raw = t.alloc[(size_t)rl * t.n + win], old_raw = ld_live(a.req + (size_t)rl * t.n + win);
  const int cnt0 = ld_live(a.count + win), allowed = t.allowed[win];
  const u64 occ_l = (lane < t.KP && a.ports && a.fx.occupied) ? a.fx.occupied[(size_t)spec * t.KP + lane] : 0ull;
  const bool contributes = a.topo_on && a.fx.off && a.fx.off[spec + 1] > a.fx.off[spec];
  const i64 rq_l = lr ? rq_raw : 0, al_l = lr ? al_raw : 0, old_l = lr ? old_raw : 0;
  const bool any_occ = __ballot(occ_l != 0) != 0;
  i64 fits_l = (lr && rq_l > 0) ? (al_l - old_l) / rq_l : 0x7fffffffffffffffll;
#pragma unroll
  for (int off = kMaxR / 2; off > 0; off >>= 1) fits_l = min(fits_l, (i64)__shfl_xor((long long)fits_l, off, kWave));
  i64 fits = min((i64)allowed - (i64)cnt0, (i64)__shfl((long long)fits_l, 0, kWave));
  if (pin != -1 || tsig >= 0 || any_occ || contributes || !fit_on) fits = 1;
  const i64 used[2] = {(i64)__shfl((long long)old_l, 0, kWave), (i64)__shfl((long long)old_l, 1, kWave)};
  const i64 total[2] = {(i64)__shfl((long long)al_l, 0, kWave), (i64)__shfl((long long)al_l, 1, kWave)};
  if (lane < kMaxR) {  // (lane r = resource r; lanes past the table's dimensions hold zeros)
    out->alloc[lane] = al_l;
    out->req[lane] = old_l;
  }
  if (lane == 0) {
    out->key = sortable_key(node_score_of(total, used));
    out->node = win;
    out->fits = (int)max((i64)1, min(fits, (i64)0x7fffffff));
    out->gnode = a.node_offset + win;
    out->room = allowed - cnt0;
    out->didx = -1;
    out->pad = a.name_rank ? a.name_rank[win] : win;
  }
}
// grid.x = asks of the batch (a.first .. a.first + gridDim.x of the round's list); out: [asks][kPropK], best first, node -1 = no further node
__global__ __launch_bounds__(kProposeThreads) void k_round_propose(RoundCtx c, RoundProposal* __restrict__ out) {
  const NodeTable& t = c.nodes;
  const SpecTable& s = c.specs;
  const ClassTable& ct = c.classes;
  const Planes& ranked = c.planes;
  const RoundArgs& a = c.args;
  const SpreadSigs& sp = s.spread;
  __shared__ int sh_apos[kProposeWaves][kPropK], sh_an[kProposeWaves], sh_alist[kPropK];
  __shared__ u64 sh_wk[kProposeWaves];
  __shared__ int sh_wt[kProposeWaves], sh_wn[kProposeWaves];
  __shared__ PropCand sh_a[kPropK], sh_b[kPropK];
  __shared__ int sh_win[kPropK], sh_nwin;
  const int tid = threadIdx.x, lane = tid % kWave, wave = tid / kWave;
  const int j = blockIdx.x;
  const bool name_on = a.filt & kPlugNodeName;
  const bool fit_on = (a.filt & kPlugFit) && (a.pre & kPlugFit);
  const bool spread_en = a.filt & kPlugSpread, ipa_en = (a.filt & kPlugInterPod) && (a.pre & kPlugInterPod);
  const int p = a.asks[a.first + j];
  const int spec = a.pod_spec[p], cls = a.pod_class[p];
  const int pin = name_on ? a.pod_pin[p] : -1;
  const int tsig = (a.topo_on && s.spread_sig) ? s.spread_sig[spec] : -1;
  RoundProposal* o = out + (size_t)j * kPropK;
  if (tid == 0) sh_nwin = 0;
  __syncthreads();
  if (a.all_fail || pin == -2) {
    // a Filter without its PreFilter state / spec.nodeName names no node of the table: nothing fits
  } else if (pin >= 0) {
    if (tid == 0) {
      NodeRegs nr;
      load_node_live(t, a, pin, &nr);
      int code;
      unsigned reason;
      if (eval_pair<true>(s, spec, pin, pin, nr, a.pre, a.filt, &code, &reason)) {
        sh_win[0] = pin;
        sh_nwin = 1;
      }
    }
  } else {
    // ---- unmoved nodes, snapshot order: the first kPropK set bits of (AND of the class's rank-ordered rows) & ~moved — for a class with
    // a topology signature checked against the live histograms, word after word, lane = node
    int na = 0;
    {
      const u64* dsc = a.cdesc + (size_t)cls * kDescWords;
      const u64* dr[kDescRows];
#pragma unroll
      for (int k = 0; k < kDescRows; ++k) dr[k] = (const u64*)dsc[k];
      const u64 meta = dsc[kDescRows];
      const int dn = (int)(unsigned)meta;
      ClassRows cr;
      cr.n = 0;
      cr.ni = 0;
      cr.start = (int)(meta >> 32);
      if (dn < 0) cr = class_rows(ranked, ct.sig[cls * 4 + 0], ct.sig[cls * 4 + 1], ct.sig[cls * 4 + 2], -1);
      int w0 = ld_live(a.cursor + cls);
      if (w0 < 0) w0 = cr.start;  // (kNoWord: some row of the class is empty)
      const int row_words = a.row_words;
      for (int base = w0 < row_words ? (w0 & ~(kWave - 1)) : row_words; base < row_words && na < kPropK; base += kProposeThreads) {
        const int w = base + tid;
        u64 x = 0;
        if (w >= w0 && w < row_words) {
          const u64 mv = ld_live(a.moved_bits + w);
          u64 v = ~0ull;
          if (dn > 0) {
#pragma unroll
            for (int k = 0; k < kDescRows; ++k) v &= dr[k][w];
          } else if (dn < 0) {
            v = class_word(cr, w);
          }
          x = ~mv & v;
        }
        const int wave_w = base + wave * kWave;
        if (tsig >= 0) {
          u64 todo = __ballot(x != 0), kept = 0;
          int found = 0;
          while (todo && found < kPropK) {
            const int fl = __ffsll((long long)todo) - 1;
            todo &= todo - 1;
            const int ww = wave_w + fl;
            const u64 xw = readlane64(x, fl);
            bool ok = false;
            if ((xw >> lane) & 1ull) {
              const int n = a.perm[ww * kWave + lane];
              int dom[kMaxKD];
#pragma unroll
              for (int k = 0; k < kMaxKD; ++k) dom[k] = k < t.KD ? t.domain[(size_t)k * t.n + n] : -1;
              ok = constraints_fail<true>(sp, tsig, dom, spread_en, ipa_en, nullptr) == 0;
            }
            const u64 y = __ballot(ok);
            if (lane == fl) kept = y;
            found += __popcll(y);
          }
          x = kept;  // (words behind the wave's kPropK-th node are not looked at: they cannot be among the first kPropK)
        }
        u64 todo = __ballot(x != 0);
        int cnt = 0;
        while (todo && cnt < kPropK) {
          const int fl = __ffsll((long long)todo) - 1;
          todo &= todo - 1;
          u64 xw = readlane64(x, fl);
          while (xw && cnt < kPropK) {
            const int b = __ffsll((long long)xw) - 1;
            xw &= xw - 1;
            if (lane == 0) sh_apos[wave][cnt] = (wave_w + fl) * kWave + b;
            ++cnt;
          }
        }
        if (lane == 0) sh_an[wave] = cnt;
        __syncthreads();
        for (int wv = 0; wv < kProposeWaves; ++wv)  // (wave wv holds the words in front of wave wv + 1's)
          for (int q = 0; q < sh_an[wv] && na < kPropK; ++q, ++na)
            if (tid == 0) sh_alist[na] = sh_apos[wv][q];
        __syncthreads();
      }
    }
    // ---- moved nodes: every live slot per pair from the slot columns, the thread's kPropK best in registers (ascending)
    u64 lk[kPropK];
    int lt[kPropK], ln[kPropK];
#pragma unroll
    for (int q = 0; q < kPropK; ++q) {
      lk[q] = ~0ull;
      lt[q] = 0x7fffffff;
      ln[q] = -1;
    }
    {
      const int n_moved = ld_live(a.n_moved);
      // (the class's "failed before" bits: a verdict without topology signature only ever turns from fit to fail, so a set bit is final —
      // set here too, by whichever workgroup of the class meets the slot first; later batches skip it without a load)
      u64* fw = (a.failed && tsig < 0) ? a.failed + (size_t)cls * a.cap64 : nullptr;
      for (int slot = tid; slot < n_moved; slot += kProposeThreads) {
        const u64 skip = ld_live(a.dead + (slot >> 6)) | (fw ? ld_live(fw + (slot >> 6)) : 0ull);
        if ((skip >> (slot & 63)) & 1ull) continue;
        NodeRegs nr;
        load_slot_live(t, a, slot, tsig >= 0, &nr);
        const int m = a.m_node[slot];
        int code;
        unsigned reason;
        if (!eval_pair<true>(s, spec, -1, m, nr, a.pre, a.filt, &code, &reason)) {
          if (fw) atomicOr((unsigned long long*)(fw + (slot >> 6)), 1ull << (slot & 63));
          continue;
        }
        u64 k = ld_live(a.m_key + slot);
        int tie = a.m_tie[slot], nn = m;
#pragma unroll
        for (int q = 0; q < kPropK; ++q)  // (insertion: the entry sinks to its place, the last one falls off)
          if (cand_less(k, tie, lk[q], lt[q])) {
            const u64 k2 = lk[q];
            const int t2 = lt[q], n2 = ln[q];
            lk[q] = k;
            lt[q] = tie;
            ln[q] = nn;
            k = k2;
            tie = t2;
            nn = n2;
          }
      }
    }
    // the workgroup's kPropK best: kPropK rounds of "smallest head", the owner of the winner moves on to its next entry
    int nb = 0;
    for (int r = 0; r < kPropK; ++r) {
      bool cand = ln[0] >= 0;
      u64 wk = ~0ull;
      int wt = 0x7fffffff, wn = -1;
      if (__ballot(cand)) {
        const unsigned hi = wave_umin(cand ? (unsigned)(lk[0] >> 32) : 0xffffffffu);
        cand = cand && (unsigned)(lk[0] >> 32) == hi;
        const unsigned lo = wave_umin(cand ? (unsigned)lk[0] : 0xffffffffu);
        cand = cand && (unsigned)lk[0] == lo;
        const unsigned ti = wave_umin(cand ? (unsigned)lt[0] : 0xffffffffu);
        cand = cand && (unsigned)lt[0] == ti;
        const int src = __ffsll((long long)__ballot(cand)) - 1;
        wk = ((u64)hi << 32) | lo;
        wt = (int)ti;
        wn = __builtin_amdgcn_readlane(ln[0], src);
      }
      if (lane == 0) {
        sh_wk[wave] = wk;
        sh_wt[wave] = wt;
        sh_wn[wave] = wn;
      }
      __syncthreads();
      u64 gk = ~0ull;
      int gt = 0x7fffffff, gn = -1;
#pragma unroll
      for (int wv = 0; wv < kProposeWaves; ++wv)
        if (sh_wn[wv] >= 0 && (gn < 0 || cand_less(sh_wk[wv], sh_wt[wv], gk, gt))) {
          gk = sh_wk[wv];
          gt = sh_wt[wv];
          gn = sh_wn[wv];
        }
      __syncthreads();
      if (gn < 0) break;  // (workgroup-uniform)
      if (tid == 0) sh_b[nb] = PropCand{gk, gt, gn};
      ++nb;
      if (ln[0] == gn) {  // (a node sits in one slot: one thread holds it)
#pragma unroll
        for (int q = 0; q + 1 < kPropK; ++q) {
          lk[q] = lk[q + 1];
          lt[q] = lt[q + 1];
          ln[q] = ln[q + 1];
        }
        lk[kPropK - 1] = ~0ull;
        lt[kPropK - 1] = 0x7fffffff;
        ln[kPropK - 1] = -1;
      }
    }
    __syncthreads();
    // ---- the two ascending lists merged (the unmoved candidates' keys, NodeID ranks and nodes fetched side by side first)
    if (tid < na) {
      const int pos = sh_alist[tid];
      sh_a[tid] = PropCand{a.rkey[pos], a.rtie[pos], a.perm[pos]};
    }
    __syncthreads();
    if (tid == 0) {
      int ia = 0, ib = 0, n = 0;
      while (n < kPropK && (ia < na || ib < nb)) {
        bool take_a = ib >= nb;
        if (ia < na && ib < nb) take_a = cand_less(sh_a[ia].key, sh_a[ia].tie, sh_b[ib].key, sh_b[ib].tie);
        sh_win[n++] = take_a ? sh_a[ia++].node : sh_b[ib++].node;
      }
      sh_nwin = n;
    }
  }
  __syncthreads();
  {  // (a wave per entry)
    const int nwin = sh_nwin;
    for (int r = wave; r < kPropK; r += kProposeWaves) write_proposal(t, s, a, spec, pin, tsig, fit_on, r < nwin ? sh_win[r] : -1, o + r);
  }
}
// The distinct nodes among a batch's n proposals (one workgroup): list[0 .. *n_out) and every proposal's index into it (didx).
__global__ __launch_bounds__(kBlock) void k_round_distinct(RoundProposal* __restrict__ props, int n, int* __restrict__ list, int* __restrict__ n_out) {
  __shared__ int keys[kDistinctSlots], idx[kDistinctSlots];
  __shared__ int cnt;
  const int tid = threadIdx.x;
  for (int h = tid; h < kDistinctSlots; h += kBlock) keys[h] = -1;
  if (tid == 0) cnt = 0;
  __syncthreads();
  for (int i = tid; i < n; i += kBlock) {
    const int node = props[i].node;
    if (node < 0) continue;
    unsigned h = ((unsigned)node * 2654435761u) & (kDistinctSlots - 1);
    for (;;) {
      const int old = atomicCAS(&keys[h], -1, node);
      if (old == -1 || old == node) break;
      h = (h + 1) & (kDistinctSlots - 1);
    }
  }
  __syncthreads();
  for (int h = tid; h < kDistinctSlots; h += kBlock)
    if (keys[h] >= 0) {
      const int my = atomicAdd(&cnt, 1);
      idx[h] = my;
      list[my] = keys[h];
    }
  __syncthreads();
  for (int i = tid; i < n; i += kBlock) {
    const int node = props[i].node;
    if (node < 0) continue;
    unsigned h = ((unsigned)node * 2654435761u) & (kDistinctSlots - 1);
    while (keys[h] != node) h = (h + 1) & (kDistinctSlots - 1);
    props[i].didx = idx[h];
  }
  if (tid == 0) *n_out = cnt;
}
// What a batch's accepted asks add to ONE node, summed by the host's replay: the assume of a batch whose pods move nothing but
// resources and pod counts (no host port, no contribution to a topology histogram) is independent node by node — a wave per node
// instead of k_allocate_round's ask-after-ask assume mode (2.9 us per ask: half the time of a batched round on one GPU).
struct RoundNodeDelta {
  int node, pods;
  i64 add[kMaxR];
};
__global__ __launch_bounds__(kWave) void k_round_assume_nodes(RoundCtx c, const RoundNodeDelta* __restrict__ list, int n) {
  const NodeTable& t = c.nodes;
  const RoundArgs& a = c.args;
  const int g = blockIdx.x, lane = threadIdx.x;
  if (g >= n) return;
  const int win = list[g].node, k_run = list[g].pods;
  const bool fit_on = (a.filt & kPlugFit) && (a.pre & kPlugFit);
  const size_t cap = (size_t)a.cap;
  const bool lr = lane < t.R, lp = lane < t.KP;
  const int rl = min(lane, t.R - 1);
  const i64 add_raw = list[g].add[min(rl, kMaxR - 1)], al_raw = t.alloc[(size_t)rl * t.n + win], old_raw = a.req[(size_t)rl * t.n + win];
  const i64 add_l = lr ? add_raw : 0, al_l = lr ? al_raw : 0, old_l = lr ? old_raw : 0;
  const int cnt0 = a.count[win], allowed = t.allowed[win];
  const int slot_prev = a.slot_of[win];
  const int rk = a.rank[win], tie_w = a.name_rank ? a.name_rank[win] : win;
  const unsigned flags_w = t.flags[win];
  const u64 port_l = lp ? (a.ports ? a.ports[(size_t)lane * t.n + win] : t.ports[(size_t)lane * t.n + win]) : 0ull;
  const int c0 = 8, Wc = min(t.W, kMaxW), col = lane - c0;  // (lanes 0..7 hold the resources, the others a static column each)
  u64 stat_l = 0;
  int dom_l = -1;
  if (col >= 0 && col < t.KT) stat_l = t.taints[(size_t)col * t.n + win];
  else if (col >= t.KT && col < t.KT + Wc) stat_l = t.labels[(size_t)(col - t.KT) * t.n + win];
  else if (col >= t.KT + Wc && col < t.KT + Wc + t.KD) dom_l = t.domain[(size_t)(col - t.KT - Wc) * t.n + win];
  const bool was_moved = slot_prev >= 0;
  int slot = slot_prev;
  if (!was_moved) {  // (the order in which nodes take their slots is free: every scan over them takes a minimum)
    if (lane == 0) slot = atomicAdd(a.n_moved, 1);
    slot = __builtin_amdgcn_readfirstlane(slot);
  }
  const int cnt = cnt0 + k_run;
  const bool dead = fit_on && (i64)cnt + 1 > (i64)allowed;
  const i64 v_l = old_l + add_l;
  if (lr) {
    a.req[(size_t)lane * t.n + win] = v_l;
    a.m_free[(size_t)lane * cap + slot] = al_l - v_l;
  }
  const i64 used[2] = {(i64)__shfl((long long)v_l, 0, kWave), (i64)__shfl((long long)v_l, 1, kWave)};
  const i64 total[2] = {(i64)__shfl((long long)al_l, 0, kWave), (i64)__shfl((long long)al_l, 1, kWave)};
  if (lane == 0) {
    a.count[win] = cnt;
    a.m_room[slot] = allowed - cnt;
    a.m_key[slot] = sortable_key(node_score_of(total, used));
    if (!was_moved) {
      atomicOr((unsigned long long*)(a.moved_bits + (rk >> 6)), 1ull << (rk & 63));
      a.slot_of[win] = slot;
      a.m_node[slot] = win;
      a.m_tie[slot] = tie_w;
      a.m_flags[slot] = flags_w;
    }
    if (dead) atomicOr((unsigned long long*)(a.dead + (slot >> 6)), 1ull << (slot & 63));
  }
  if (!was_moved) {
    if (lp) a.m_ports[(size_t)lane * cap + slot] = port_l;
    if (col >= 0 && col < t.KT) a.m_taint[(size_t)col * cap + slot] = stat_l;
    else if (col >= t.KT && col < t.KT + Wc) a.m_label[(size_t)(col - t.KT) * cap + slot] = stat_l;
    else if (col >= t.KT + Wc && col < t.KT + Wc + t.KD) a.m_dom[(size_t)(col - t.KT - Wc) * cap + slot] = dom_l;
  }
}
// grid.x = asks of the batch; cross[ask][cw] bit d = the ask passes Predicates() on node list[d] as the round's state stands
__global__ __launch_bounds__(kBlock) void k_round_cross(RoundCtx c, const int* __restrict__ list, const int* __restrict__ n_list, u64* __restrict__ cross, int cw) {
  const NodeTable& t = c.nodes;
  const SpecTable& s = c.specs;
  const RoundArgs& a = c.args;
  const int tid = threadIdx.x, lane = tid % kWave;
  const int j = blockIdx.x;
  const int p = a.asks[a.first + j];
  const int spec = a.pod_spec[p];
  const int pin = (a.filt & kPlugNodeName) ? a.pod_pin[p] : -1;
  const int nd = *n_list;
  const bool never = a.all_fail || pin == -2;
  for (int d0 = 0; d0 < cw * kWave; d0 += kBlock) {
    const int d = d0 + tid;
    bool fit = false;
    if (d < nd && !never) {
      const int node = list[d];
      NodeRegs nr;
      load_node_live(t, a, node, &nr);
      int code;
      unsigned reason;
      fit = eval_pair<true>(s, spec, pin, node, nr, a.pre, a.filt, &code, &reason);
    }
    const u64 bits = __ballot(fit);
    if (lane == 0 && (d >> 6) < cw) cross[(size_t)j * cw + (d >> 6)] = bits;
  }
}

// order-independent checksum of the bitmap: Σ mix64(word ⊕ position-salt) over the meaningful words
__device__ __forceinline__ u64 mix64(u64 z) {
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
__global__ __launch_bounds__(kBlock) void k_checksum(const u64* __restrict__ bitmap, int n_pods, int row_words, int row_stride,
                                                     const int* __restrict__ pod_row, u64* __restrict__ out) {
  size_t total = (size_t)n_pods * row_words;
  u64 acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    size_t p = i / row_words, w = i % row_words;
    u64 x = bitmap[(size_t)pod_row[p] * row_stride + w];
    acc += mix64(x ^ ((i + 1) * 0x9e3779b97f4a7c15ull));
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
  if (threadIdx.x % kWave == 0) atomicAdd(out, acc);
}


// Decision exchange of a node-sharded cluster (ykpred_exchange_decisions): after the MIN all-reduce of the order keys, the
// shards that hold a node with the winning key offer its GLOBAL index; a second MIN picks the smallest (ties by node index).
__global__ __launch_bounds__(kBlock) void k_decision_candidates(int n_pods, const int* __restrict__ decisions, const i64* __restrict__ keys,
                                                                const i64* __restrict__ best_key, int node_offset, int* __restrict__ cand) {
  int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_pods) return;
  const int d = decisions[p];
  cand[p] = (d >= 0 && keys[p] == best_key[p]) ? d + node_offset : 0x7fffffff;
}
__global__ __launch_bounds__(kBlock) void k_decision_finalize(int n_pods, const int* __restrict__ cand, const i64* __restrict__ best_key,
                                                              int* __restrict__ decisions, i64* __restrict__ keys) {
  int p = blockIdx.x * kBlock + threadIdx.x;
  if (p >= n_pods) return;
  decisions[p] = cand[p] == 0x7fffffff ? -1 : cand[p];
  keys[p] = best_key[p];
}

// Parity support: (a) every member row must equal the row of its class's representative pod, padding words must be zero —
// counts the offending words; (b) gather of listed rows into a dense [n][row_words] buffer for readback.
__global__ __launch_bounds__(kBlock) void k_check_class_rows(const u64* __restrict__ bitmap, int n_pods, int row_words, int row_stride,
                                                             const int* __restrict__ pod_class, const int* __restrict__ class_first,
                                                             const int* __restrict__ pod_row, u64* __restrict__ bad) {
  const size_t total = (size_t)n_pods * row_stride;
  u64 acc = 0;
  for (size_t i = (size_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (size_t)gridDim.x * kBlock) {
    const size_t p = i / row_stride, w = i % row_stride;
    const u64 x = bitmap[(size_t)pod_row[p] * row_stride + w];
    if (w >= (size_t)row_words) {
      acc += x != 0;
    } else {
      const int rep = class_first[pod_class[p]];
      acc += rep < 0 || x != bitmap[(size_t)pod_row[rep] * row_stride + w];
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) acc += __shfl_down(acc, off, kWave);
  if (threadIdx.x % kWave == 0 && acc) atomicAdd(bad, acc);
}
// Class rows out of an evaluated bitmap: out[c] = the row of class c's representative ask (zeros for a class without asks),
// full stride. Member rows of a class are identical, so [C][row_stride] is the whole bitmap in class-compressed form —
// what a node shard sends over xGMI instead of its P rows (ykpred_gather_bitmap_compressed).
__global__ __launch_bounds__(kBlock) void k_collect_class_rows(const u64* __restrict__ bitmap, int n_classes, const int* __restrict__ class_first,
                                                               const int* __restrict__ pod_row, int row_stride, u64* __restrict__ out) {
  const int c = blockIdx.x;
  if (c >= n_classes) return;
  const int rep = class_first[c];
  const u64* src = rep >= 0 ? bitmap + (size_t)pod_row[rep] * row_stride : nullptr;
  for (int w = threadIdx.x; w < row_stride; w += kBlock) out[(size_t)c * row_stride + w] = src ? src[w] : 0ull;
}
// rows of the listed classes in list order (the band writer addresses class rows by zone-A slot)
__global__ __launch_bounds__(kBlock) void k_pick_class_rows(const u64* __restrict__ class_rows, const int* __restrict__ list, int n, int row_stride,
                                                            u64* __restrict__ out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const u64* src = class_rows + (size_t)list[i] * row_stride;
  for (int w = threadIdx.x; w < row_stride; w += kBlock) out[(size_t)i * row_stride + w] = src[w];
}
// Expansion of a PEER shard's class rows when the peer partitions the asks into classes differently (signatures are merged
// relative to the shard's own taint / node-name dictionaries): one wave per PHYSICAL row — consecutive waves write consecutive
// rows, the class rows come out of L2 — through the inverse row map (row -> ask, -1 = no ask owns the row) and the peer's
// ask -> class map.
__global__ __launch_bounds__(kBlock) void k_invert_row_map(const int* __restrict__ pod_row, int n_pods, int* __restrict__ row_pod) {
  const int p = blockIdx.x * kBlock + threadIdx.x;
  if (p < n_pods) row_pod[pod_row[p]] = p;
}
// One wave covers kExpandRows consecutive rows as ONE run of 16-byte pieces (narrow rows of a node shard are < 1 KiB: a wave
// per row would issue a single store per lane behind a chain of three dependent loads).
constexpr int kExpandRows = 8;
__global__ __launch_bounds__(kBlock) void k_expand_by_row(u64* __restrict__ out, const u64* __restrict__ class_rows, const int* __restrict__ pod_class,
                                                          const int* __restrict__ row_pod, int n_rows, int row_stride) {
  typedef u64 u64x2 __attribute__((ext_vector_type(2)));
  const int r0 = (blockIdx.x * kWavesPerBlock + threadIdx.x / kWave) * kExpandRows;
  if (r0 >= n_rows) return;
  const int lane = threadIdx.x % kWave;
  const int ppr = row_stride / 2;  // 16-byte pieces per row
  const int rows = min(kExpandRows, n_rows - r0);
  // lane i < rows: the class row of row r0 + i (or -1), fetched once and broadcast
  int src_l = -1;
  if (lane < rows) {
    const int p = row_pod[r0 + lane];
    src_l = p >= 0 ? pod_class[p] : -1;
  }
  const int total = rows * ppr;
  u64x2* dst = (u64x2*)(out + (size_t)r0 * row_stride);
  int rr = lane / ppr, cc = lane - rr * ppr;        // piece `lane`: row in the run, piece in the row
  const int drr = kWave / ppr, dcc = kWave - drr * ppr;  // advance of 64 pieces
  for (int j = lane; j < total; j += kWave) {
    const int cls = __shfl(src_l, rr, kWave);
    if (cls >= 0) dst[j] = ((const u64x2*)(class_rows + (size_t)cls * row_stride))[cc];
    rr += drr;
    cc += dcc;
    if (cc >= ppr) {
      cc -= ppr;
      ++rr;
    }
  }
}
// sig table that makes class c's only plane row c of the `tol` family: k_combine / k_combine_wave then expand a class-row table
__global__ __launch_bounds__(kBlock) void k_identity_sigs(int n_classes, int* __restrict__ sig) {
  const int c = blockIdx.x * kBlock + threadIdx.x;
  if (c >= n_classes) return;
  sig[c * 4 + 0] = -1;
  sig[c * 4 + 1] = c;
  sig[c * 4 + 2] = -1;
  sig[c * 4 + 3] = -1;
}
// pods == null: pods first, first + 1, ...
__global__ __launch_bounds__(kBlock) void k_gather_rows(const u64* __restrict__ bitmap, int n, const int* __restrict__ pods, int first,
                                                        const int* __restrict__ pod_row, int row_words, int row_stride, u64* __restrict__ out) {
  const int i = blockIdx.x;
  if (i >= n) return;
  const u64* src = bitmap + (size_t)pod_row[pods ? pods[i] : first + i] * row_stride;
  for (int w = threadIdx.x; w < row_words; w += kBlock) out[(size_t)i * row_words + w] = src[w];
}

}  // namespace ykk
