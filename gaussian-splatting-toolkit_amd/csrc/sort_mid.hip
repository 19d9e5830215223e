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
#include "gsr_common.h"

namespace gsr_sort {

constexpr int kThreads = 256;
constexpr int kItems = 8;
constexpr int kChunk = kThreads * kItems;  // 2048
constexpr int kRadix = 256;

__device__ __forceinline__ unsigned digit_of(unsigned key, int shift) { return (key >> shift) & 255u; }

__global__ __launch_bounds__(kThreads) void hist_kernel(const int n, const unsigned *__restrict__ keys,
                                                        const int shift, const int num_chunks,
                                                        unsigned *__restrict__ hist) {
  __shared__ unsigned h[kRadix];
  const int tid = threadIdx.x, b = blockIdx.x;
  h[tid] = 0;
  __syncthreads();
  const int base = b * kChunk;
#pragma unroll
  for (int i = 0; i < kItems; ++i) {
    const int idx = base + i * kThreads + tid;
    if (idx < n) atomicAdd(&h[digit_of(keys[idx], shift)], 1u);
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
__global__ __launch_bounds__(kThreads) void scatter_kernel(
    const int n, const unsigned *__restrict__ keys_in, const int *__restrict__ vals_in, const int shift,
    const int num_chunks, const unsigned *__restrict__ offsets, const unsigned *__restrict__ totals,
    unsigned *__restrict__ keys_out, int *__restrict__ vals_out) {
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
    key[i] = live ? keys_in[idx] : 0xffffffffu;
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
      keys_out[pos] = key[i];
      vals_out[pos] = val[i];
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
    hipLaunchKernelGGL(hist_kernel, dim3(chunks), dim3(kThreads), 0, s, n, kin, 8 * p, chunks, hist);
    hipLaunchKernelGGL(digit_scan_kernel, dim3(kRadix), dim3(kThreads), 0, s, chunks, hist, totals);
    hipLaunchKernelGGL(scatter_kernel, dim3(chunks), dim3(kThreads), 0, s, n, kin, vin, 8 * p, chunks,
                       (const unsigned *)hist, (const unsigned *)totals, kout, vout);
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
