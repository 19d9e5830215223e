// sort_mid.hip -- stable LSD radix sort of (u32 key, i32 index) pairs for
// "mid-size" inputs (64 k .. 4 M items), gfx950.
//
// Why it exists: the depth ordering of the N ~ 1e6 Gaussians is the only sort of
// that size in the pipeline, and rocPRIM serves it badly on MI355X whatever the
// configuration (tools/exp/sortbench.hip: default / merge-path / Onesweep
// variants all 143-190 us for 1 M pairs -- the work itself is 32 MB of traffic).
// This is the textbook three-kernel pass, sized for the job:
//   histogram  one workgroup per 2048-item chunk, 256-bin LDS histogram
//   scan       one workgroup per digit: exclusive scan of its chunk counts + total
//   scatter    same chunks; stable in-chunk ranking with wave-level digit
//              matching (8 ballots) + wave-private running counts in LDS
// 4 passes x 3 launches for 31-bit keys.  The first pass reads the index as the
// lane's position (no iota buffer).
//
// Round 3: the depth ordering itself is built by sort_bucket.hip (one bucket pass + one in-LDS pass); what follows
// serves it as the fallback its hint selects after a view whose buckets overflowed (binning_fast.hip:
// use_bucket_sort), as the look-back scan of the gathered counts (gsr_sort_mid_scan_inplace), and the other
// mid-size sorts of the library (GSR_TILE_SORT=m).
// gsr_sort_mid_depth is the depth ordering's own entry: the first pass makes its keys
// from (depth, radius) on the fly (no key-building launch), the last pass also moves the
// per-Gaussian tile counts into depth order (the gather rides along with the scatter's own
// random accesses), and one decoupled look-back kernel turns them into the inclusive
// prefix: 13 launches for what were 1 + 12 + rocPRIM's 2 (and a 1 M-element gather).
#include "gsr_common.h"

namespace gsr_sort {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kChunk = kThreads * kItems;  // 2048
constexpr int kRadix = 256;

__device__ __forceinline__ unsigned digit_of(unsigned key, int shift) { return (key >> shift) & 255u; }

// visible splats have depth > 0 (bit pattern orders like the value); culled ones emit
// nothing, park them at the front
__device__ __forceinline__ unsigned depth_key(const float *__restrict__ depths, const int *__restrict__ radii,
                                              const int i) {
  return radii[i] > 0 ? __float_as_uint(depths[i]) : 0u;
}

// kDepth: the keys are depth_key(depths, radii, i) (keys == nullptr); the launch also zeroes
// `zero_words` words at `zero` (the look-back scan's state, used after the sort).
template <bool kDepth>
__global__ __launch_bounds__(kThreads) void hist_kernel(const int n, const unsigned *__restrict__ keys,
                                                        const float *__restrict__ depths,
                                                        const int *__restrict__ radii, const int shift,
                                                        const int num_chunks, unsigned *__restrict__ hist,
                                                        unsigned *__restrict__ zero, const int zero_words) {
  __shared__ unsigned h[kRadix];
  const int tid = threadIdx.x, b = blockIdx.x;
  h[tid] = 0;
  if (kDepth) {
    const int z = b * kThreads + tid;
    if (z < zero_words) zero[z] = 0u;
  }
  __syncthreads();
  const int base = b * kChunk;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int idx = base + i * kThreads + tid;
    if (idx < n) atomicAdd(&h[digit_of(kDepth ? depth_key(depths, radii, idx) : keys[idx], shift)], 1u);
  }
  __syncthreads();
  hist[(size_t)tid * num_chunks + b] = h[tid];  // digit-major
}

// One workgroup per digit: exclusive scan of that digit's per-chunk counts (in place)
// and the digit's total.  (A single-workgroup scan of all 256 x chunks counters took
// 90 us -- one CU chasing 60 k dependent loads.)
__global__ __launch_bounds__(kThreads) void digit_scan_kernel(const int num_chunks,
                                                              unsigned *__restrict__ hist,
                                                              unsigned *__restrict__ totals) {
  __shared__ unsigned wave_sums[4];
  __shared__ unsigned carry;
  const int tid = threadIdx.x, d = blockIdx.x;
  unsigned *row = hist + (size_t)d * num_chunks;
  if (tid == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < num_chunks; base += kThreads) {
    const int i = base + tid;
    const unsigned c = i < num_chunks ? row[i] : 0u;
    unsigned v = c;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(v, o);
      if ((tid & 63) >= o) v += t;
    }
    if ((tid & 63) == 63) wave_sums[tid >> 6] = v;
    __syncthreads();
    unsigned before = carry;
    for (int k = 0; k < (tid >> 6); ++k) before += wave_sums[k];
    if (i < num_chunks) row[i] = before + v - c;
    __syncthreads();
    if (tid == 0) carry += wave_sums[0] + wave_sums[1] + wave_sums[2] + wave_sums[3];
    __syncthreads();
  }
  if (tid == 0) totals[d] = carry;
}

// `vals_in == nullptr`: the value of item i is i.
// Stable in-chunk ranking without a barrier inside the loop: every wave owns a contiguous
// quarter of the chunk (kItems rounds of 64 items); the lanes of a round that share a digit are
// found by wave-wide matching (8 ballots) and ranked by lane, a WAVE-PRIVATE counter per digit
// carries the rank across the rounds; one barrier later the four waves' counts are prefixed per
// digit.  (The first version synchronised the workgroup three times per round: 15-17 us per
// pass at 1 M keys.)
// kDepth: keys from (depths, radii) as in hist_kernel.  kGather: the item that lands at `pos`
// also brings its rows of a side table along: gather_dst[b n + pos] = gather_src[b n + val]
// for b < gather_rows (keys_out may then be nullptr: the last pass).
template <bool kDepth, bool kGather>
__global__ __launch_bounds__(kThreads) void scatter_kernel(
    const int n, const unsigned *__restrict__ keys_in, const float *__restrict__ depths,
    const int *__restrict__ radii, const int *__restrict__ vals_in, const int shift,
    const int num_chunks, const unsigned *__restrict__ offsets, const unsigned *__restrict__ totals,
    unsigned *__restrict__ keys_out, int *__restrict__ vals_out, const int *__restrict__ gather_src,
    int *__restrict__ gather_dst, const int gather_rows) {
  __shared__ unsigned wave_cnt[4][kRadix];  // per-wave digit counts, then per-wave output bases
  __shared__ unsigned wsum[4];
  const int tid = threadIdx.x, b = blockIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
#pragma unroll
  for (int k = 0; k < 4; ++k) wave_cnt[k][tid] = 0;
  // global base of digit `tid` = exclusive scan of the 256 digit totals + this chunk's offset
  unsigned digit_base;
  {
    const unsigned t = totals[tid];
    unsigned v = t;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(v, o);
      if (lane >= o) v += u;
    }
    if (lane == 63) wsum[w] = v;
    __syncthreads();
    unsigned before = 0;
    for (int k = 0; k < w; ++k) before += wsum[k];
    digit_base = before + v - t + offsets[(size_t)tid * num_chunks + b];
  }
  const int base = b * kChunk + w * (kChunk / 4);
  unsigned key[kItems], rank[kItems];
  int val[kItems];
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int idx = base + i * 64 + lane;
    const bool live = idx < n;
    key[i] = live ? (kDepth ? depth_key(depths, radii, idx) : keys_in[idx]) : 0xffffffffu;
    val[i] = live ? (vals_in ? vals_in[idx] : idx) : -1;
  }
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const bool live = val[i] >= 0 || (base + i * 64 + lane) < n;
    const unsigned d = digit_of(key[i], shift);
    unsigned long long peers = __ballot(live);
#pragma unroll
    for (int bit = 0; bit < 8; ++bit) {
      const unsigned long long set = __ballot((d >> bit) & 1u);
      peers &= ((d >> bit) & 1u) ? set : ~set;
    }
    const unsigned below = (unsigned)__popcll(peers & lt);
    const unsigned prev = live ? wave_cnt[w][d] : 0u;  // all peers read before the group's first lane writes
    rank[i] = prev + below;
    if (live && below == 0) wave_cnt[w][d] = prev + (unsigned)__popcll(peers);
  }
  __syncthreads();
  {  // digit `tid`: turn the four waves' counts into their output bases
    unsigned run = digit_base;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const unsigned c = wave_cnt[k][tid];
      wave_cnt[k][tid] = run;
      run += c;
    }
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int idx = base + i * 64 + lane;
    if (idx < n) {
      const unsigned pos = wave_cnt[w][digit_of(key[i], shift)] + rank[i];
      if (!kGather || keys_out) keys_out[pos] = key[i];
      vals_out[pos] = val[i];
      if (kGather)
        for (int r = 0; r < gather_rows; ++r)
          gather_dst[(size_t)r * n + pos] = gather_src[(size_t)r * n + val[i]];
    }
  }
}

// ---- in-place inclusive scan, one pass with decoupled look-back ------------------------
// state[0]: ticket counter (workgroups take their tile in ARRIVAL order, so every predecessor
// of a tile is already running); state[2 + 2 t], [3 + 2 t]: tile t's (flag << 32 | value) as one
// 64-bit word: flag 1 = the tile's own sum, 2 = its inclusive prefix.  All zero at launch.
constexpr int kScanItems = 16;
constexpr int kScanTile = kThreads * kScanItems;  // 4096

__global__ __launch_bounds__(kThreads) void scan_lookback_kernel(const int total, int *__restrict__ data,
                                                                unsigned *__restrict__ state) {
  __shared__ int wsum[4];
  __shared__ int s_tile, s_prefix;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned long long *status = reinterpret_cast<unsigned long long *>(state + 2);
  if (tid == 0) s_tile = (int)atomicAdd(state, 1u);
  __syncthreads();
  const int tile = s_tile;
  const int base = tile * kScanTile + tid * kScanItems;
  int v[kScanItems];
  const bool vec = base + kScanItems <= total && (reinterpret_cast<uintptr_t>(data) & 15) == 0;
  if (vec) {
    const int4 *p = reinterpret_cast<const int4 *>(data + base);  // base % 16 == 0
#pragma unroll
    for (int q = 0; q < kScanItems / 4; ++q) {
      const int4 t = p[q];
      v[4 * q] = t.x, v[4 * q + 1] = t.y, v[4 * q + 2] = t.z, v[4 * q + 3] = t.w;
    }
  } else {
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) v[q] = base + q < total ? data[base + q] : 0;
  }
  int sum = 0;
#pragma unroll
  for (int q = 0; q < kScanItems; ++q) sum += v[q];
  int incl = sum;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int before = 0;
  for (int k = 0; k < w; ++k) before += wsum[k];
  const int tile_sum = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  if (w == 0) {  // wave 0 publishes the tile's sum and looks back
    int prefix = 0;
    if (tile > 0) {
      if (lane == 0)
        __hip_atomic_store(&status[tile], (1ull << 32) | (unsigned)tile_sum, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_AGENT);
      int hi = tile - 1;  // nearest predecessor not yet accounted for
      while (true) {
        const int t = hi - lane;
        unsigned long long st = 2ull << 32;  // tiles before the first count as "prefix 0"
        if (t >= 0) st = __hip_atomic_load(&status[t], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned flag = (unsigned)(st >> 32);
        const unsigned long long done = __ballot(flag == 2u), none = __ballot(flag == 0u);
        // lanes nearer than the first finished predecessor must all have published their sums
        const int first = done ? __ffsll((long long)done) - 1 : 64;
        const unsigned long long nearer = first >= 64 ? ~0ull : ((1ull << first) - 1ull);
        if (none & (nearer | (first < 64 ? (1ull << first) : 0ull))) continue;  // not there yet: poll again
        int add = (lane <= first) ? (int)(unsigned)st : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) add += __shfl_xor(add, o);
        prefix += add;
        if (first < 64) break;
        hi -= 64;
      }
    }
    if (lane == 0) {
      __hip_atomic_store(&status[tile], (2ull << 32) | (unsigned)(prefix + tile_sum), __ATOMIC_RELAXED,
                         __HIP_MEMORY_SCOPE_AGENT);
      s_prefix = prefix;
    }
  }
  __syncthreads();
  int run = s_prefix + before + incl - sum;
  if (vec) {
    int4 *p = reinterpret_cast<int4 *>(data + base);
#pragma unroll
    for (int q = 0; q < kScanItems / 4; ++q) {
      int4 t;
      run += v[4 * q], t.x = run;
      run += v[4 * q + 1], t.y = run;
      run += v[4 * q + 2], t.z = run;
      run += v[4 * q + 3], t.w = run;
      p[q] = t;
    }
  } else {
#pragma unroll
    for (int q = 0; q < kScanItems; ++q) {
      run += v[q];
      if (base + q < total) data[base + q] = run;
    }
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace gsr_sort

// ---- internal interface used by binning_fast.hip ---------------------------
// workspace: 2 key buffers + 2 value buffers + histogram
size_t gsr_sort_mid_workspace_bytes(int n) {
  using namespace gsr_sort;
  const size_t chunks = (size_t)gsr_cdiv((unsigned)n, kChunk);
  return 2 * align_up(4 * (size_t)n) + align_up(4 * (size_t)n) + align_up(4 * kRadix * (chunks + 1));
}

// Sorts by the low `key_bits` bits, stably.  keys_in / vals_in are left untouched.
// vals_in == nullptr: the value of item i is i.  keys_out may be nullptr.
int gsr_sort_mid_pairs(int n, const unsigned *keys_in, const int *vals_in, unsigned *keys_out,
                       int *vals_out, int key_bits, void *workspace, size_t workspace_bytes,
                       hipStream_t s) {
  using namespace gsr_sort;
  if (n <= 0) return GSR_OK;
  if (workspace_bytes < gsr_sort_mid_workspace_bytes(n)) {
    gsr_set_error("sort_mid: workspace too small");
    return GSR_ENOMEM;
  }
  const int chunks = (int)gsr_cdiv((unsigned)n, kChunk);
  char *ws = static_cast<char *>(workspace);
  const size_t nb = align_up(4 * (size_t)n);
  unsigned *kbuf[2] = {reinterpret_cast<unsigned *>(ws), reinterpret_cast<unsigned *>(ws + nb)};
  int *vtmp = reinterpret_cast<int *>(ws + 2 * nb);
  unsigned *hist = reinterpret_cast<unsigned *>(ws + 3 * nb);
  unsigned *totals = hist + (size_t)kRadix * chunks;
  const int passes = (key_bits + 7) / 8;
  // ping-pong so that the LAST pass writes into the caller's output buffers
  const unsigned *kin = keys_in;
  const int *vin = vals_in;
  for (int p = 0; p < passes; ++p) {
    const bool last = p == passes - 1;
    unsigned *kout = (last && keys_out) ? keys_out : kbuf[p & 1];
    int *vout = ((passes - 1 - p) & 1) ? vtmp : vals_out;
    hipLaunchKernelGGL(hist_kernel<false>, dim3(chunks), dim3(kThreads), 0, s, n, kin, (const float *)nullptr,
                       (const int *)nullptr, 8 * p, chunks, hist, (unsigned *)nullptr, 0);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(kRadix), dim3(kThreads), 0, s, chunks, hist, totals);
    hipLaunchKernelGGL((scatter_kernel<false, false>), dim3(chunks), dim3(kThreads), 0, s, n, kin,
                       (const float *)nullptr, (const int *)nullptr, vin, 8 * p, chunks, (const unsigned *)hist,
                       (const unsigned *)totals, kout, vout, (const int *)nullptr, (int *)nullptr, 0);
    kin = kout;
    vin = vout;
  }
  GSR_CHECK_LAUNCH("sort_mid");
  return GSR_OK;
}

int gsr_sort_mid(int n, const unsigned *keys_in, int *vals_out, int key_bits, void *workspace,
                 size_t workspace_bytes, hipStream_t s) {
  return gsr_sort_mid_pairs(n, keys_in, nullptr, nullptr, vals_out, key_bits, workspace, workspace_bytes, s);
}

// the look-back scan on its own (sort_bucket.hip gathers the counts itself)
int gsr_sort_mid_scan_state_words(long long total) {
  using namespace gsr_sort;
  const long long tiles = (total + kScanTile - 1) / kScanTile;
  return (int)(2 * (tiles + 1) + 2);
}
int gsr_sort_mid_scan_inplace(long long total, int *data, unsigned *state, hipStream_t s) {
  using namespace gsr_sort;
  if (total <= 0) return GSR_OK;
  const int tiles = (int)((total + kScanTile - 1) / kScanTile);
  hipLaunchKernelGGL(scan_lookback_kernel, dim3(tiles), dim3(kThreads), 0, s, (int)total, data, state);
  GSR_CHECK_LAUNCH("sort_mid_scan_inplace");
  return GSR_OK;
}

// ---- the depth ordering: order + inclusive prefix of the tile counts in that order -------
// counts[rows][n] (index order) -> cum[rows * n]: inclusive scan of counts[r][order[i]] over
// (r, i); counts == cum == nullptr: the order only.  Needs the workspace of gsr_sort_mid_workspace_bytes(n) + gsr_sort_mid_depth_extra(n, rows).
size_t gsr_sort_mid_depth_extra(int n, int rows) {
  using namespace gsr_sort;
  const size_t tiles = ((size_t)n * rows + kScanTile - 1) / kScanTile;
  return align_up(8 * (tiles + 1) + 8);
}

int gsr_sort_mid_depth(int n, const float *depths, const int *radii, const int *counts, int rows, int *order,
                       int *cum, void *workspace, size_t workspace_bytes, hipStream_t s) {
  using namespace gsr_sort;
  if (n <= 0) return GSR_OK;
  const size_t sort_bytes = gsr_sort_mid_workspace_bytes(n);
  if (workspace_bytes < sort_bytes + gsr_sort_mid_depth_extra(n, rows)) {
    gsr_set_error("sort_mid_depth: workspace too small");
    return GSR_ENOMEM;
  }
  const int chunks = (int)gsr_cdiv((unsigned)n, kChunk);
  char *ws = static_cast<char *>(workspace);
  const size_t nb = align_up(4 * (size_t)n);
  unsigned *kbuf[2] = {reinterpret_cast<unsigned *>(ws), reinterpret_cast<unsigned *>(ws + nb)};
  int *vtmp = reinterpret_cast<int *>(ws + 2 * nb);
  unsigned *hist = reinterpret_cast<unsigned *>(ws + 3 * nb);
  unsigned *totals = hist + (size_t)kRadix * chunks;
  unsigned *state = reinterpret_cast<unsigned *>(ws + sort_bytes);
  const long long total = (long long)n * rows;
  const int tiles = (int)((total + kScanTile - 1) / kScanTile);
  const int state_words = 2 * (tiles + 1) + 2;
  if (state_words > chunks * kThreads) {  // (16 rows of counts at most: never more words than threads)
    gsr_set_error("sort_mid_depth: scan state does not fit the zeroing launch");
    return GSR_EINVAL;
  }
  constexpr int passes = 4;  // 31-bit keys
  const unsigned *kin = nullptr;
  const int *vin = nullptr;
  for (int p = 0; p < passes; ++p) {
    const bool first = p == 0, last = p == passes - 1;
    unsigned *kout = (last && counts) ? nullptr : kbuf[p & 1];  // (the gather variant needs no sorted keys)
    int *vout = ((passes - 1 - p) & 1) ? vtmp : order;
    if (first)
      hipLaunchKernelGGL(hist_kernel<true>, dim3(chunks), dim3(kThreads), 0, s, n, kin, depths, radii, 0, chunks,
                         hist, state, state_words);
    else
      hipLaunchKernelGGL(hist_kernel<false>, dim3(chunks), dim3(kThreads), 0, s, n, kin, (const float *)nullptr,
                         (const int *)nullptr, 8 * p, chunks, hist, (unsigned *)nullptr, 0);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(kRadix), dim3(kThreads), 0, s, chunks, hist, totals);
    if (first)
      hipLaunchKernelGGL((scatter_kernel<true, false>), dim3(chunks), dim3(kThreads), 0, s, n, kin, depths, radii,
                         vin, 0, chunks, (const unsigned *)hist, (const unsigned *)totals, kout, vout,
                         (const int *)nullptr, (int *)nullptr, 0);
    else if (last && counts)
      hipLaunchKernelGGL((scatter_kernel<false, true>), dim3(chunks), dim3(kThreads), 0, s, n, kin,
                         (const float *)nullptr, (const int *)nullptr, vin, 8 * p, chunks, (const unsigned *)hist,
                         (const unsigned *)totals, kout, vout, counts, cum, rows);
    else
      hipLaunchKernelGGL((scatter_kernel<false, false>), dim3(chunks), dim3(kThreads), 0, s, n, kin,
                         (const float *)nullptr, (const int *)nullptr, vin, 8 * p, chunks, (const unsigned *)hist,
                         (const unsigned *)totals, kout, vout, (const int *)nullptr, (int *)nullptr, 0);
    kin = kout;
    vin = vout;
  }
  if (counts)
    hipLaunchKernelGGL(scan_lookback_kernel, dim3(tiles), dim3(kThreads), 0, s, (int)total, cum, state);
  GSR_CHECK_LAUNCH("sort_mid_depth");
  return GSR_OK;
}
