// sort_bucket.hip -- the depth ordering as ONE bucket pass + ONE in-LDS pass (gfx950).
//
// sort_mid.hip orders the N ~ 1e6 Gaussians with four 8-bit LSD passes: 12 launches that move every
// (key, index) pair through HBM four times and spend most of their 100 us on launch/latency chains.
// Depths of one view are not arbitrary 31-bit numbers, so four launches do (a fifth scans the tile counts when the
// caller wants them: they are gathered wherever an index is written into the order):
//   hist      every workgroup first derives the SAME monotone bucket map from the same 4096 sampled keys: the
//             sample's population per float octave (exponent) decides how many of the B buckets the octave gets,
//             spread linearly over its mantissas (over the part of them the sample's smallest / largest key bound,
//             in the first / last octave): linear in depth inside an octave, population-proportional across
//             octaves; octaves outside the sampled window fall into an underflow / overflow bucket; bucket 0 =
//             culled.  Then an LDS histogram of its 4096-item chunk over all B buckets -> table[chunk][B]
//   scan      every bucket's column of the table, exclusive, in place, + the bucket's total
//   scatter   items in registers; slot = bucket's first slot + its items in earlier chunks + in earlier waves of
//             the chunk + rank in the wave (wave-level matching on the bucket bits, wave-private counters):
//             stable without sorting the chunk; writes (key, index) pairs, bucket 0 straight into the order
//   sort      four adjacent buckets per workgroup, one WAVE each (<= 1024 items, no barrier), stable, by
//             (key - the bucket's smallest key), whose few bits the wave takes from the data; larger buckets by the
//             whole workgroup (<= 2048 in LDS, above that -- and the under/overflow buckets -- LSD rounds
//             through HBM)
// Monotone bucket map + stable passes = the same permutation as the LSD sort (and as a stable sort of the
// keys): ties keep ascending index.  The sample only shapes the map: any map built this way is monotone, so
// the result never depends on it, only the balance of the buckets does.  gsr_depth_order looks at the largest
// bucket of EARLIER calls (published to pinned memory) and goes back to sort_mid.hip while it is above what the
// LDS paths hold.
#include "gsr_common.h"
#include "raster_common.h"
#include "tile_rows.h"

#include <algorithm>

namespace gsr_bsort {

constexpr int kChunk = 4096;      // items per workgroup of hist / scatter (one row of the table)
constexpr int kCap = 2048;        // largest bucket sorted in LDS by a workgroup (16 items per thread: 264 VGPRs)
constexpr int kWaveCap = 1024;    // largest bucket sorted by a single wave
constexpr int kElemBits = 12;     // an item's position in its chunk / bucket (< 4096)
constexpr int kHistThreads = 1024;
constexpr int kScatterWaves = 4, kScatterThreads = 64 * kScatterWaves, kScatterItems = kChunk / kScatterThreads;
constexpr int kSortThreads = 256;  // 4 waves: 4 buckets, or one large bucket

__device__ __forceinline__ unsigned depth_key(const float *__restrict__ depths, const int *__restrict__ radii,
                                              const int i) {
  // as sort_mid.hip: the low 31 bits order the (positive) depths; culled splats (key 0) first.  Both loads are
  // unconditional (a load under the select becomes a branch with its own wait: one round trip per item)
  const int r = radii[i];
  const unsigned bits = __float_as_uint(depths[i]);
  return r > 0 ? (bits & 0x7fffffffu) : 0u;
}

__device__ __forceinline__ void wave_minmax(unsigned &lo, unsigned &hi) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    lo = min(lo, (unsigned)__shfl_xor((int)lo, o));
    hi = max(hi, (unsigned)__shfl_xor((int)hi, o));
  }
}

// lanes (among `live` ones) whose low `bits` bits of d equal this lane's
__device__ __forceinline__ unsigned long long match_digit(const unsigned d, const int bits, const bool live) {
  const unsigned long long all = __ballot(live);
  unsigned lo = (unsigned)all, hi = (unsigned)(all >> 32);
#pragma unroll
  for (int bit = 0; bit < 8; ++bit) {
    if (bit < bits) {  // (uniform)
      const unsigned one = (d >> bit) & 1u;
      const unsigned long long set = __ballot(one != 0u);
      const unsigned flip = one - 1u;  // lanes with the bit clear match the complement
      lo &= (unsigned)set ^ flip, hi &= (unsigned)(set >> 32) ^ flip;
    }
  }
  return ((unsigned long long)hi << 32) | lo;
}

// ---- the bucket map -----------------------------------------------------------------------------------
// One entry per float octave (exponent): W buckets from `base` on, spread linearly over the mantissas [mlo, mlo + span)
// (the whole octave, except in the sample's first and last octave, which the depths fill only partly: there the
// sample's smallest / largest key bound the range, padded by 1/32; keys beyond it fall into the octave's first /
// last bucket):  bucket = base + min(W - 1, (unsigned)(float(clamp(mantissa - mlo, 0, span - 1)) * c)),  c = W / span.
// Every step is monotone in the key, whatever the table holds; octaves outside the sampled window (+- 1) have
// W = 1 and base = the underflow bucket 1 / the overflow bucket B - 1.
struct OctaveMap {
  unsigned base_w;  // base | W << 16
  unsigned mlo, span;
  float c;
};
constexpr unsigned kMinWidth = 16;  // (a bucket spans at most 2^23 / 16 + 1 mantissas: 20 bits)
constexpr int kSample = 4096;

__device__ __forceinline__ unsigned bucket_of(const unsigned key, const OctaveMap *__restrict__ map) {
  if (!key) return 0u;
  const OctaveMap t = map[key >> 23];
  const unsigned mant = key & 0x7fffffu, W = t.base_w >> 16, base = t.base_w & 0xffffu;
  const unsigned d = min(mant > t.mlo ? mant - t.mlo : 0u, t.span - 1u);
  return base + min(W - 1u, (unsigned)((float)d * t.c));
}

// Called by all threads of a workgroup (blockDim.x >= 256, whole waves); afterwards map[0..256) (LDS) holds the
// table.  Integer arithmetic on the same samples: every workgroup derives the same table.
// the sample: 256 runs of 16 adjacent items, evenly spaced (n >= 4096).  Every workgroup reads the same 4096 items:
// as 4096 single items spread over the array that was 2 x 4096 cache lines per workgroup, 130 MB of L2 traffic
// for a 1 M-item sort and 12 us; runs of 16 are 64 bytes each
template <int kT>
__device__ __forceinline__ void load_samples(const int n, const float *__restrict__ depths,
                                             const int *__restrict__ radii, unsigned (&sample)[kSample / kT]) {
  const int spacing = n / (kSample / 16);
#pragma unroll
  for (int q = 0; q < kSample / kT; ++q) {
    const int j = q * kT + (int)threadIdx.x;
    sample[q] = depth_key(depths, radii, (j >> 4) * spacing + (j & 15));
  }
}

// Called by all threads of a workgroup (kT threads, whole waves); afterwards map[0..256) (LDS) holds the
// table.  Integer arithmetic on the same samples: every workgroup derives the same table.
template <int kT>
__device__ __forceinline__ void build_bucket_map(const unsigned (&sample)[kSample / kT], const int log2_buckets,
                                                 OctaveMap *__restrict__ map, unsigned *__restrict__ oct,
                                                 int *__restrict__ sh /* >= 24 words */) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned B = 1u << log2_buckets;
  if (tid < 256) oct[tid] = 0u;
  if (tid == 0) sh[16] = -1, sh[17] = 0;  // (unsigned min / max of the sampled keys)
  __syncthreads();
#pragma unroll
  for (int q = 0; q < kSample / kT; ++q) {  // one LDS atomic per (wave, octave), not per sample
    const unsigned long long peers = match_digit(sample[q] >> 23, 8, sample[q] != 0u);
    if (sample[q] && (peers & ((1ull << lane) - 1ull)) == 0) atomicAdd(&oct[sample[q] >> 23], (unsigned)__popcll(peers));
  }
  __syncthreads();
  {  // the sample's smallest and largest visible key: what the first and the last octave really hold
    unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
    for (int q = 0; q < kSample / kT; ++q)
      if (sample[q]) kmin = min(kmin, sample[q]), kmax = max(kmax, sample[q]);
    wave_minmax(kmin, kmax);
    if (lane == 0) atomicMin(reinterpret_cast<unsigned *>(&sh[16]), kmin), atomicMax(reinterpret_cast<unsigned *>(&sh[17]), kmax);
  }
  unsigned pop = 0;
  int first = 256, last = -1;
  if (tid < 256) {
    pop = oct[tid];
    if (pop) first = last = tid;
    unsigned sum = pop;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      first = min(first, __shfl_xor(first, o));
      last = max(last, __shfl_xor(last, o));
      sum += __shfl_xor(sum, o);
    }
    if (lane == 0) sh[w] = first, sh[4 + w] = last, sh[8 + w] = (int)sum;
  }
  __syncthreads();
  first = min(min(sh[0], sh[1]), min(sh[2], sh[3]));
  last = max(max(sh[4], sh[5]), max(sh[6], sh[7]));
  const unsigned visible = (unsigned)max(sh[8] + sh[9] + sh[10] + sh[11], 1);
  int lo = max(first - 1, 0), hi = min(last + 1, 255);
  if (last < 0) lo = 1, hi = 0;  // nothing visible in the sample: everything visible -> overflow bucket
  const unsigned budget = B - 3u - kMinWidth * (unsigned)(hi - lo + 1);  // (kMinWidth buckets per octave set aside)
  const bool inside = tid >= lo && tid <= hi;
  // the octave's share of the budget, rounded down (+ the buckets set aside for it): the widths sum to < B - 2
  const unsigned width =
      (tid < 256 && inside) ? kMinWidth + (unsigned)(((unsigned long long)pop * budget) / visible) : 0u;
  unsigned incl = width;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const unsigned t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();
  if (tid < 256 && lane == 63) sh[w] = (int)incl;
  __syncthreads();
  if (tid < 256) {
    unsigned base = 2u + incl - width;
    for (int q = 0; q < w; ++q) base += (unsigned)sh[q];
    OctaveMap t{(1u << 16) | (tid > hi ? B - 1u : 1u), 0u, 1u, 0.f};
    if (inside) {
      const unsigned kmin = (unsigned)sh[16], kmax = (unsigned)sh[17];
      unsigned m0 = 0u, m1 = 0x7fffffu;
      if (tid == first) m0 = kmin & 0x7fffffu;
      if (tid == last) m1 = kmax & 0x7fffffu;
      const unsigned pad = (m1 - m0 + 1u) >> 5;
      if (tid == first) m0 = m0 > pad ? m0 - pad : 0u;
      if (tid == last) m1 = min(m1 + pad, 0x7fffffu);
      t.base_w = (width << 16) | base;
      t.mlo = m0, t.span = m1 - m0 + 1u;
      t.c = (float)width / (float)t.span;
    }
    map[tid] = t;
  }
  __syncthreads();
}

// ---- histogram (+ the reach records of the list builders, when asked) ---------------------------------
// kRecords: the launch also writes the per-Gaussian record gsr_count_reach writes (same function, same inputs): its
// 60 B per Gaussian stream while the map is being built, instead of in a launch of its own.
struct RecordArgs {
  const float *xys, *conics, *opacities;
  int tiles_x, tiles_y;
  SplatRec *recs;
};

template <bool kRecords>
__global__ __launch_bounds__(kHistThreads) void hist_kernel(const int n, const float *__restrict__ depths,
                                                            const int *__restrict__ radii, const int log2_buckets,
                                                            unsigned *__restrict__ table,
                                                            OctaveMap *__restrict__ map_out, const RecordArgs ra,
                                                            unsigned *__restrict__ zero, const int zero_words) {
  extern __shared__ unsigned h[];  // B counters
  __shared__ OctaveMap map[256];
  __shared__ unsigned oct[256];
  __shared__ int sh[24];
  constexpr int kI = kChunk / kHistThreads;
  const int tid = threadIdx.x, B = 1 << log2_buckets;
  const int base = blockIdx.x * kChunk;
  for (int z = blockIdx.x * kHistThreads + tid; z < zero_words; z += gridDim.x * kHistThreads) zero[z] = 0u;  // (the look-back scan's state)
  // the chunk's keys (and record inputs) are in flight while the map is built
  unsigned key[kI];
  SplatIn in[kRecords ? kI : 1];
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int idx = base + i * kHistThreads + tid;
    const unsigned k = depth_key(depths, radii, min(idx, n - 1));
    key[i] = idx < n ? k : 0xffffffffu;
    if (kRecords) in[i] = load_splat_in(min(idx, n - 1), ra.xys, radii, ra.conics, ra.opacities);
  }
  unsigned sample[kSample / kHistThreads];
  load_samples<kHistThreads>(n, depths, radii, sample);
  for (int j = tid; j < B; j += kHistThreads) h[j] = 0u;
  if (kRecords) {  // (their stores drain while the map is built)
#pragma unroll
    for (int i = 0; i < kI; ++i) {
      const int idx = base + i * kHistThreads + tid;
      if (idx < n) ra.recs[idx] = splat_record_from(in[i], ra.conics != nullptr, ra.tiles_x, ra.tiles_y, 16);
    }
  }
  build_bucket_map<kHistThreads>(sample, log2_buckets, map, oct, sh);
  if (blockIdx.x == 0 && tid < 256) map_out[tid] = map[tid];
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const bool culled = key[i] == 0u;  // (often a third of the chunk: one atomic per wave instead of one each)
    const unsigned long long c = __ballot(culled);
    if (c && (tid & 63) == 0) atomicAdd(&h[0], (unsigned)__popcll(c));
    if (!culled && key[i] != 0xffffffffu) atomicAdd(&h[bucket_of(key[i], map)], 1u);
  }
  __syncthreads();
  unsigned *row = table + (size_t)blockIdx.x * B;
  for (int j = tid; j < B; j += kHistThreads) row[j] = h[j];
}

// ---- column scan -------------------------------------------------------------------------------------
// A workgroup owns 16 adjacent buckets (one 64-byte segment of every table row); its 16 thread groups
// split the chunks.  table[c][b] becomes the number of items of bucket b in the chunks before c.
__global__ __launch_bounds__(256) void scan_kernel(const int chunks, const int B, unsigned *__restrict__ table,
                                                   unsigned *__restrict__ totals) {
  __shared__ unsigned part[16][16];
  const int tid = threadIdx.x, bl = tid & 15, g = tid >> 4;
  const int b = blockIdx.x * 16 + bl;
  const int per = (chunks + 15) / 16;
  const int c0 = g * per, c1 = min(chunks, c0 + per);
  if (per <= 16) {  // (up to 1 M items: the group's counts stay in registers between the two passes)
    unsigned t[16], sum = 0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      const unsigned v = table[(size_t)min(c0 + k, chunks - 1) * B + b];  // (clamped: all sixteen loads go out together)
      t[k] = c0 + k < c1 ? v : 0u;
      sum += t[k];
    }
    part[g][bl] = sum;
    __syncthreads();
    unsigned run = 0;
    for (int k = 0; k < g; ++k) run += part[k][bl];
    if (g == 15) totals[b] = run + sum;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
      if (c0 + k < c1) table[(size_t)(c0 + k) * B + b] = run;
      run += t[k];
    }
    return;
  }
  unsigned sum = 0;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) sum += table[(size_t)c * B + b];
  part[g][bl] = sum;
  __syncthreads();
  unsigned run = 0;
  for (int k = 0; k < g; ++k) run += part[k][bl];
  if (g == 15) totals[b] = run + sum;
#pragma unroll 8
  for (int c = c0; c < c1; ++c) {
    unsigned *q = table + (size_t)c * B + b;
    const unsigned t = *q;
    *q = run;
    run += t;
  }
}

// ---- the stable in-LDS digit round ---------------------------------------------------------------------

template <int kW>
struct SortShared {
  unsigned cnt[kW][256];
  unsigned wsum[4];
};

// Workgroup of kW waves.  The item of (wave w, round i, lane) is element w * seg + i * 64 + lane of the
// sequence (live below m).  rank[i]: the item's rank among the wave's earlier items with its digit.  On return
// S.cnt[w][d] is the number of the block's items that precede wave w's first item with digit d in digit-major
// order (the exclusive digit base included); the return value (threads below 256) is digit `tid`'s item count.
template <int kW, int kI>
__device__ __forceinline__ unsigned rank_round(const unsigned (&v)[kI], unsigned (&rank)[kI], const int R,
                                               const int seg, const int m, const int shf, const int bits,
                                               SortShared<kW> &S) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const unsigned mask = (1u << bits) - 1u;
#pragma unroll
  for (int k = 0; k < 4; ++k) (&S.cnt[0][0])[k * (kW * 64) + tid] = 0u;
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    if (i < R) {
      const bool live = w * seg + i * 64 + lane < m;
      unsigned d = (v[i] >> shf) & mask;
      asm volatile("" : "+v"(d)::"memory");  // (one item's ballots at a time, see scatter_kernel)
      const unsigned long long peers = match_digit(d, bits, live);
      const unsigned below = (unsigned)__popcll(peers & lt);
      const unsigned prev = S.cnt[w][d];  // every peer reads before the group's first lane writes
      rank[i] = prev + below;
      if (live && below == 0) S.cnt[w][d] = prev + (unsigned)__popcll(peers);
    }
  }
  __syncthreads();
  unsigned c[kW], tot = 0, incl = 0;
  if (tid < 256) {  // (whole waves)
#pragma unroll
    for (int k = 0; k < kW; ++k) c[k] = S.cnt[k][tid], tot += c[k];
    incl = tot;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) S.wsum[w] = incl;
  }
  __syncthreads();
  if (tid < 256) {
    unsigned run = incl - tot;
    for (int k = 0; k < w; ++k) run += S.wsum[k];
#pragma unroll
    for (int k = 0; k < kW; ++k) S.cnt[k][tid] = run, run += c[k];
  }
  __syncthreads();
  return tot;
}

// Stable sort of the m (<= 256 kI) words in v[] by bits [lo, lo + nbits), in rounds of <= 8 bits; buf: that many
// words of LDS, left holding the sorted sequence; v[] holds it too, in the same arrangement.
template <int kW, int kI>
__device__ __forceinline__ void block_sort(unsigned (&v)[kI], const int R, const int seg, const int m, const int lo,
                                           const int nbits, unsigned *__restrict__ buf, SortShared<kW> &S) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int rounds = (nbits + 7) / 8;
  for (int done = 0; done < nbits;) {
    const int bits = (nbits - done + rounds - 1) / rounds;
    const int shf = lo + done;
    unsigned rank[kI];
    rank_round<kW, kI>(v, rank, R, seg, m, shf, bits, S);
    const unsigned mask = (1u << bits) - 1u;
#pragma unroll
    for (int i = 0; i < kI; ++i)
      if (i < R && w * seg + i * 64 + lane < m) buf[S.cnt[w][(v[i] >> shf) & mask] + rank[i]] = v[i];
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kI; ++i)
      if (i < R && w * seg + i * 64 + lane < m) v[i] = buf[w * seg + i * 64 + lane];
    done += bits, --rounds;
  }
}

// Where an item lands in the order, its tile counts go along: cum[r n + pos] = counts[r n + id] (rows r: one count
// per Gaussian, or one per tile-row band); the inclusive scan over (r, pos) follows the sort (sort_mid.hip's
// look-back kernel).  counts == nullptr: the order only.
struct Gather {
  const int *counts;
  int *cum;
  int rows, n;
};
__device__ __forceinline__ void place(int *__restrict__ order, const Gather &ga, const unsigned pos, const int id) {
  order[pos] = id;
  if (ga.counts)
    for (int r = 0; r < ga.rows; ++r) ga.cum[(size_t)r * ga.n + pos] = ga.counts[(size_t)r * ga.n + id];
}

// ---- scatter ---------------------------------------------------------------------------------------
// One workgroup per chunk, items in registers in (wave, round, lane) = index order.  An item's slot is
//   (bucket's first slot) + (bucket's items in earlier chunks) + (in earlier waves of this chunk) + (rank in its wave),
// the last two from WAVE-PRIVATE 16-bit counters over a sweep of 4096 buckets (wave-level matching on the 12 bucket
// bits gives the rank among the lanes of a round; the counter carries it across rounds): stable without sorting the
// chunk.  B > 4096: one sweep per 4096 buckets over the same registers.
constexpr int kSweep = 4096;

__device__ __forceinline__ unsigned long long match12(const unsigned d, const bool in) {
  const unsigned long long all = __ballot(in);
  unsigned lo = (unsigned)all, hi = (unsigned)(all >> 32);
#pragma unroll
  for (int bit = 0; bit < 12; ++bit) {
    const unsigned one = (d >> bit) & 1u;
    const unsigned long long set = __ballot(one != 0u);
    const unsigned flip = one - 1u;
    lo &= (unsigned)set ^ flip, hi &= (unsigned)(set >> 32) ^ flip;
  }
  return ((unsigned long long)hi << 32) | lo;
}

__global__ __launch_bounds__(kScatterThreads) void scatter_kernel(
    const int n, const float *__restrict__ depths, const int *__restrict__ radii,
    const OctaveMap *__restrict__ map_in, const int log2_buckets, const unsigned *__restrict__ table,
    const unsigned *__restrict__ totals,
    uint2 *__restrict__ pairs, int *__restrict__ order, unsigned *__restrict__ bucket_base,
    int *__restrict__ stats, const Gather ga) {
  __shared__ unsigned arr[kSweep];
  __shared__ unsigned short cnt[kScatterWaves][kSweep];
  __shared__ unsigned s_sum[kScatterWaves], s_max[kScatterWaves];
  __shared__ OctaveMap map[256];
  constexpr int kI = kScatterItems, seg = kChunk / kScatterWaves, kPer = kSweep / kScatterThreads;
  const int B = 1 << log2_buckets;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  const unsigned long long lt = (1ull << lane) - 1ull;
  map[tid] = map_in[tid];
  const int base = blockIdx.x * kChunk;
  const int m = min(kChunk, n - base);
  unsigned key[kI], bkt[kI], rank[kI];
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = w * seg + i * 64 + lane;
    const unsigned k = depth_key(depths, radii, base + min(e, m - 1));
    key[i] = e < m ? k : 0u;
  }
  __syncthreads();
#pragma unroll
  for (int i = 0; i < kI; ++i) bkt[i] = w * seg + i * 64 + lane < m ? bucket_of(key[i], map) : 0xffffffffu;
  unsigned first = 0, biggest = 0;  // first slot of the sweep's first bucket
  for (int sweep = 0; sweep < B / kSweep; ++sweep) {
    // this thread's kPer adjacent buckets of the sweep: totals and this chunk's column entries (in flight during the ranking)
    unsigned tot[kPer], col[kPer];
    {
      const uint4 *tp = reinterpret_cast<const uint4 *>(totals + sweep * kSweep + tid * kPer);
      const uint4 *cp = reinterpret_cast<const uint4 *>(table + (size_t)blockIdx.x * B + sweep * kSweep + tid * kPer);
#pragma unroll
      for (int q = 0; q < kPer / 4; ++q) {
        const uint4 t = tp[q], c = cp[q];
        tot[4 * q] = t.x, tot[4 * q + 1] = t.y, tot[4 * q + 2] = t.z, tot[4 * q + 3] = t.w;
        col[4 * q] = c.x, col[4 * q + 1] = c.y, col[4 * q + 2] = c.z, col[4 * q + 3] = c.w;
      }
    }
    {
      unsigned *z = reinterpret_cast<unsigned *>(&cnt[0][0]);
#pragma unroll
      for (int q = 0; q < kScatterWaves * kSweep / 2 / kScatterThreads; ++q) z[q * kScatterThreads + tid] = 0u;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kI; ++i) {
      const bool in = (bkt[i] >> 12) == (unsigned)sweep;
      unsigned d = bkt[i] & (kSweep - 1);
      asm volatile("" : "+v"(d)::"memory");  // one item's twelve ballots at a time: hoisted above the loop they
                                             // take 200 SGPR pairs (512 VGPRs of spills)
      const unsigned long long peers = match12(d, in);
      const unsigned below = (unsigned)__popcll(peers & lt);
      const unsigned prev = cnt[w][d];  // every peer reads before the group's first lane writes
      rank[i] = in ? prev + below : rank[i];
      if (in && below == 0) cnt[w][d] = (unsigned short)(prev + (unsigned)__popcll(peers));
    }
    __syncthreads();
    {  // arr[b] = first slot of bucket b + its items in earlier chunks; cnt[w][b] -> items in earlier waves
      unsigned sum = 0;
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        sum += tot[q];
        if (sweep * kSweep + tid * kPer + q > 0) biggest = max(biggest, tot[q]);
      }
      unsigned incl = sum;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane == 63) s_sum[w] = incl;
#pragma unroll
      for (int q = 0; q < kSweep / kScatterThreads; ++q) {  // (bucket q * threads + tid: conflict-free)
        const int d = q * kScatterThreads + tid;
        unsigned run = 0;
#pragma unroll
        for (int k = 0; k < kScatterWaves; ++k) {
          const unsigned c = cnt[k][d];
          cnt[k][d] = (unsigned short)run;
          run += c;
        }
      }
      __syncthreads();
      unsigned run = first + incl - sum;
      for (int k = 0; k < w; ++k) run += s_sum[k];
#pragma unroll
      for (int q = 0; q < kPer; ++q) {
        if (blockIdx.x == 0) bucket_base[sweep * kSweep + tid * kPer + q] = run;
        arr[tid * kPer + q] = run + col[q];
        run += tot[q];
      }
      unsigned whole = 0;
      for (int k = 0; k < kScatterWaves; ++k) whole += s_sum[k];
      first += whole;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < kI; ++i) {
      if ((bkt[i] >> 12) == (unsigned)sweep) {
        const unsigned d = bkt[i] & (kSweep - 1);
        const unsigned pos = arr[d] + cnt[w][d] + rank[i];
        const int id = base + w * seg + i * 64 + lane;
        if (bkt[i] == 0)
          place(order, ga, pos, id);
        else
          pairs[pos] = make_uint2(key[i], (unsigned)id);
      }
    }
    __syncthreads();  // (the next sweep clears the counters)
  }
  if (blockIdx.x == 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) biggest = max(biggest, (unsigned)__shfl_xor((int)biggest, o));
    if (lane == 0) s_max[w] = biggest;
    __syncthreads();
    if (tid == 0) {
      unsigned mx = 0;
      for (int k = 0; k < kScatterWaves; ++k) mx = max(mx, s_max[k]);
      stats[0] = (int)mx;
      bucket_base[B] = first;
    }
  }
}

// ---- per-bucket sort ---------------------------------------------------------------------------------
struct WaveShared {
  unsigned buf[kWaveCap], ids[kWaveCap], cnt[256];
};
struct BlockShared {
  unsigned buf[kCap], idL[kCap];
  SortShared<4> S;
};

// One wave, m <= 64 kI items, no barrier: the rounds of rank_round / block_sort with wave-private state.  The keys
// of a bucket differ in few bits: they are sorted as (key - smallest key of the bucket), whose width the wave takes
// from the data.  -> false when those bits and the item's position do not fit one word (a bucket that caught keys from
// beyond the sampled range): the workgroup then takes it through the slow path.  (The item count per lane is a
// template parameter: with a run-time count the sixteen predicated copies of the ranking step cost 216 VGPRs and
// spill the SGPRs.)
template <int kI>
__device__ __forceinline__ bool wave_sort(const unsigned s, const int m, const uint2 *__restrict__ pairs,
                                          int *__restrict__ order, const Gather &ga, WaveShared &W) {
  constexpr int kEBits = kI == 1 ? 6 : kI == 2 ? 7 : kI == 4 ? 8 : kI == 8 ? 9 : 10;
  const int lane = threadIdx.x & 63;
  const unsigned long long lt = (1ull << lane) - 1ull;
  unsigned v[kI], rank[kI];
  unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = i * 64 + lane;
    const uint2 pr = pairs[s + min(e, m - 1)];  // (unconditional: the loads of all rounds go out together)
    if (e < m) W.ids[e] = pr.y;
    v[i] = pr.x;
    kmin = min(kmin, pr.x), kmax = max(kmax, pr.x);  // (a clamped duplicate changes neither)
  }
  wave_minmax(kmin, kmax);
  kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
  const int lowbits = kmax > kmin ? 32 - __clz((int)(kmax - kmin)) : 0;
  if (lowbits + kEBits > 32) return false;
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = i * 64 + lane;
    v[i] = e < m ? ((v[i] - kmin) << kEBits) | (unsigned)e : 0u;
  }
  int rounds = (lowbits + 7) / 8;
  for (int done = 0; done < lowbits;) {
    const int bits = (lowbits - done + rounds - 1) / rounds;
    const int shf = kEBits + done;
    const unsigned mask = (1u << bits) - 1u;
#pragma unroll
    for (int q = 0; q < 4; ++q) W.cnt[q * 64 + lane] = 0u;
#pragma unroll
    for (int i = 0; i < kI; ++i) {
      {
        const bool live = i * 64 + lane < m;
        unsigned d = (v[i] >> shf) & mask;
        asm volatile("" : "+v"(d)::"memory");  // (one item's ballots at a time, see scatter_kernel)
        const unsigned long long peers = match_digit(d, bits, live);
        const unsigned below = (unsigned)__popcll(peers & lt);
        const unsigned prev = W.cnt[d];
        rank[i] = prev + below;
        if (live && below == 0) W.cnt[d] = prev + (unsigned)__popcll(peers);
      }
    }
    {  // exclusive scan over the 256 digit counts: lane owns digits 4 lane .. 4 lane + 3
      const unsigned c0 = W.cnt[4 * lane], c1 = W.cnt[4 * lane + 1], c2 = W.cnt[4 * lane + 2],
                     c3 = W.cnt[4 * lane + 3];
      const unsigned tot = c0 + c1 + c2 + c3;
      unsigned incl = tot;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      const unsigned ex = incl - tot;
      W.cnt[4 * lane] = ex, W.cnt[4 * lane + 1] = ex + c0, W.cnt[4 * lane + 2] = ex + c0 + c1,
                   W.cnt[4 * lane + 3] = ex + c0 + c1 + c2;
    }
#pragma unroll
    for (int i = 0; i < kI; ++i)
      if (i * 64 + lane < m) W.buf[W.cnt[(v[i] >> shf) & mask] + rank[i]] = v[i];
#pragma unroll
    for (int i = 0; i < kI; ++i)
      if (i * 64 + lane < m) v[i] = W.buf[i * 64 + lane];
    done += bits, --rounds;
  }
#pragma unroll
  for (int i = 0; i < kI; ++i)
    if (i * 64 + lane < m) place(order, ga, s + i * 64 + lane, (int)W.ids[v[i] & ((1u << kEBits) - 1u)]);
  return true;
}

// The slow path of a bucket above kCap items: LSD rounds through HBM (pairs <-> pairs2), one workgroup.
__device__ __noinline__ void sort_large_bucket(const unsigned s, const unsigned m, uint2 *pairs, uint2 *pairs2,
                                               int *__restrict__ order, const Gather &ga, unsigned *run,
                                               SortShared<4> &S) {
  constexpr int kI = 4, kTile = kI * kSortThreads;  // (small tiles: few registers; speed is not the point here)
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned sub;
  int lowbits;
  {  // the bucket's key range; all keys equal: the scatter's order is final
    unsigned lo = 0xffffffffu, hi = 0u;
    for (unsigned e = tid; e < m; e += kSortThreads) {
      const unsigned k = pairs[s + e].x;
      lo = min(lo, k), hi = max(hi, k);
    }
    wave_minmax(lo, hi);
    if (lane == 0) S.cnt[0][w] = lo, S.cnt[1][w] = hi;
    __syncthreads();
    lo = min(min(S.cnt[0][0], S.cnt[0][1]), min(S.cnt[0][2], S.cnt[0][3]));
    hi = max(max(S.cnt[1][0], S.cnt[1][1]), max(S.cnt[1][2], S.cnt[1][3]));
    __syncthreads();
    if (lo == hi) {
      for (unsigned e = tid; e < m; e += kSortThreads) place(order, ga, s + e, (int)pairs[s + e].y);
      return;
    }
    sub = lo, lowbits = 32 - __clz((int)(hi - lo));
  }
  uint2 *src = pairs, *dst = pairs2;
  int rounds = (lowbits + 7) / 8;
  for (int done = 0; done < lowbits;) {
    const int bits = (lowbits - done + rounds - 1) / rounds;
    const unsigned mask = (1u << bits) - 1u;
    run[tid] = 0u;  // histogram of the whole bucket -> run[d] = first slot of digit d
    __syncthreads();
    for (unsigned e = tid; e < m; e += kSortThreads)
      atomicAdd(&run[((src[s + e].x - sub) >> done) & mask], 1u);
    __syncthreads();
    {
      const unsigned c = run[tid];
      unsigned incl = c;
#pragma unroll
      for (int o = 1; o < 64; o <<= 1) {
        const unsigned t = __shfl_up(incl, o);
        if (lane >= o) incl += t;
      }
      if (lane == 63) S.wsum[w] = incl;
      __syncthreads();
      unsigned ex = incl - c;
      for (int k = 0; k < w; ++k) ex += S.wsum[k];
      run[tid] = ex;
      __syncthreads();
    }
    for (unsigned t0 = 0; t0 < m; t0 += kTile) {
      const int mt = (int)min((unsigned)kTile, m - t0);
      constexpr int R = kI, seg = R * 64;
      unsigned v[kI], rank[kI];
      uint2 pr[kI];
#pragma unroll
      for (int i = 0; i < kI; ++i) {
        const int e = w * seg + i * 64 + lane;
        pr[i] = src[s + t0 + min(e, mt - 1)];
        v[i] = e < mt ? (pr[i].x - sub) >> done : 0u;
      }
      const unsigned tot = rank_round<4, kI>(v, rank, R, seg, mt, 0, bits, S);
#pragma unroll
      for (int i = 0; i < kI; ++i) {
        const int e = w * seg + i * 64 + lane;
        if (i < R && e < mt) {
          const unsigned d = v[i] & mask;
          dst[s + run[d] + S.cnt[w][d] - S.cnt[0][d] + rank[i]] = pr[i];
        }
      }
      __syncthreads();
      run[tid] += tot;
      __syncthreads();
    }
    uint2 *t = src;
    src = dst, dst = t;
    done += bits, --rounds;
    __threadfence_block();
  }
  for (unsigned e = tid; e < m; e += kSortThreads) place(order, ga, s + e, (int)src[s + e].y);
}

// One workgroup, m <= 256 kI items (kI <= 8: positions take 11 bits).  -> false: key range too wide to pack.
template <int kI>
__device__ __forceinline__ bool block_bucket(const unsigned s, const int m, const uint2 *__restrict__ pairs,
                                             int *__restrict__ order, const Gather &ga, BlockShared &L) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  constexpr int seg = kI * 64;
  unsigned v[kI];
  unsigned kmin = 0xffffffffu, kmax = 0u;
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = w * seg + i * 64 + lane;
    const uint2 pr = pairs[s + min(e, m - 1)];  // (unconditional: the loads of all rounds go out together)
    if (e < m) L.idL[e] = pr.y;
    v[i] = pr.x;
    kmin = min(kmin, pr.x), kmax = max(kmax, pr.x);
  }
  wave_minmax(kmin, kmax);
  if (lane == 0) L.S.cnt[0][w] = kmin, L.S.cnt[1][w] = kmax;
  __syncthreads();
  kmin = min(min(L.S.cnt[0][0], L.S.cnt[0][1]), min(L.S.cnt[0][2], L.S.cnt[0][3]));
  kmax = max(max(L.S.cnt[1][0], L.S.cnt[1][1]), max(L.S.cnt[1][2], L.S.cnt[1][3]));
  kmin = __builtin_amdgcn_readfirstlane(kmin), kmax = __builtin_amdgcn_readfirstlane(kmax);
  __syncthreads();
  const int lowbits = kmax > kmin ? 32 - __clz((int)(kmax - kmin)) : 0;
  if (lowbits + kElemBits > 32) return false;
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = w * seg + i * 64 + lane;
    v[i] = e < m ? ((v[i] - kmin) << kElemBits) | (unsigned)e : 0u;
  }
  block_sort<4, kI>(v, kI, seg, m, kElemBits, lowbits, L.buf, L.S);
#pragma unroll
  for (int i = 0; i < kI; ++i) {
    const int e = w * seg + i * 64 + lane;
    if (e < m) place(order, ga, s + e, (int)L.idL[v[i] & ((1u << kElemBits) - 1u)]);
  }
  return true;
}

// A workgroup takes four adjacent buckets: each wave sorts one (<= 1024 items) on its own; larger ones are
// then taken one at a time by the whole workgroup.
__global__ __launch_bounds__(kSortThreads) void bucket_sort_kernel(const int B,
                                                                   const unsigned *__restrict__ bucket_base,
                                                                   uint2 *pairs, uint2 *pairs2,
                                                                   int *__restrict__ order, const Gather ga) {
  __shared__ union {
    WaveShared wave[4];
    BlockShared blk;
  } L;
  __shared__ unsigned s_start[5];
  __shared__ int s_left[4];  // bucket k still to be sorted by the whole workgroup
  const int tid = threadIdx.x, w = tid >> 6;
  const int b0 = 1 + 4 * blockIdx.x;
  if (tid < 5) s_start[tid] = bucket_base[min(b0 + tid, B)];
  __syncthreads();
  {
    // (wave-uniform values, told to the compiler: as vector values every `bit < bits` is a divergent branch)
    const unsigned sw = __builtin_amdgcn_readfirstlane(s_start[w]);
    const unsigned mw = __builtin_amdgcn_readfirstlane(s_start[w + 1]) - sw;
    const int b = b0 + __builtin_amdgcn_readfirstlane(w);
    // the under/overflow buckets hold keys of any octave: through the slow path, on full keys, whatever their size
    bool left = mw > 0 && b < B;
    if (left && mw <= (unsigned)kWaveCap && b != 1 && b != B - 1) {
      if (mw <= 64)
        left = !wave_sort<1>(sw, (int)mw, pairs, order, ga, L.wave[w]);
      else if (mw <= 128)
        left = !wave_sort<2>(sw, (int)mw, pairs, order, ga, L.wave[w]);
      else if (mw <= 256)
        left = !wave_sort<4>(sw, (int)mw, pairs, order, ga, L.wave[w]);
      else if (mw <= 512)
        left = !wave_sort<8>(sw, (int)mw, pairs, order, ga, L.wave[w]);
      else
        left = !wave_sort<16>(sw, (int)mw, pairs, order, ga, L.wave[w]);
    }
    if ((tid & 63) == 0) s_left[w] = left;
  }
  __syncthreads();
  if (!(s_left[0] | s_left[1] | s_left[2] | s_left[3])) return;  // (uniform)
  for (int k = 0; k < 4; ++k) {
    if (!s_left[k]) continue;
    const unsigned s = __builtin_amdgcn_readfirstlane(s_start[k]);
    const unsigned m = __builtin_amdgcn_readfirstlane(s_start[k + 1]) - s;
    const int b = b0 + k;
    __syncthreads();
    if (b == 1 || b == B - 1 || m > (unsigned)kCap || !block_bucket<8>(s, (int)m, pairs, order, ga, L.blk)) {
      __syncthreads();
      sort_large_bucket(s, m, pairs, pairs2, order, ga, L.blk.buf, L.blk.S);
    }
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }
inline int log2_buckets_for(int n) { return n <= (3 << 19) ? 12 : 13; }  // 4096 buckets up to 1.5 M items, 8192 above
}  // namespace gsr_bsort

// ---- internal interface used by binning_fast.hip ------------------------------------------------------
// workspace: pairs | pairs2 | table[chunks][B] | totals[B] | bucket_base[B + 1] | map[256] | stats | scan state
int gsr_sort_mid_scan_state_words(long long total);
size_t gsr_sort_bucket_workspace_bytes(int n, int rows) {
  using namespace gsr_bsort;
  const size_t chunks = gsr_cdiv((unsigned)n, kChunk), B = (size_t)1 << log2_buckets_for(n);
  const size_t state = rows > 0 ? align_up(4 * (size_t)gsr_sort_mid_scan_state_words((long long)n * rows)) : 0;
  return 2 * align_up(8 * (size_t)n) + align_up(4 * chunks * B) + align_up(4 * B) + align_up(4 * (B + 1)) + 4096 + 256 +
         state;
}

// Items a single wave of the per-bucket sort holds: larger buckets occupy a whole workgroup (<= 2048: in LDS, above:
// through HBM) while its other three buckets wait.
int gsr_sort_bucket_wave_cap(void) { return gsr_bsort::kWaveCap; }

// stats (device-writable, e.g. pinned host memory; or NULL): the call leaves the size of its largest visible bucket there
// sort_mid.hip: in-place inclusive scan of `total` ints, one launch; `state` = its words (gsr_sort_mid_scan_state_words),
// zeroed by the caller
int gsr_sort_mid_scan_state_words(long long total);
int gsr_sort_mid_scan_inplace(long long total, int *data, unsigned *state, hipStream_t s);

// xys != NULL: the first launch also writes the reach records of gsr_count_reach(counts == NULL) to `recs`.
// counts != NULL (rows x n, row-major): cum[rows * n] <- inclusive scan of counts[r][order[i]] over (r, i) -- the
// counts are gathered where the order is written, one look-back scan launch follows.
int gsr_sort_bucket_depth(int n, const float *depths, const int *radii, int *order, void *workspace,
                          size_t workspace_bytes, int *stats, const float *xys, const float *conics,
                          const float *opacities, int tiles_x, int tiles_y, void *recs, const int *counts, int rows,
                          int *cum, hipStream_t s) {
  using namespace gsr_bsort;
  if (n <= 0) return GSR_OK;
  if (workspace_bytes < gsr_sort_bucket_workspace_bytes(n, counts ? rows : 0)) {
    gsr_set_error("sort_bucket_depth: workspace too small");
    return GSR_ENOMEM;
  }
  const int chunks = (int)gsr_cdiv((unsigned)n, kChunk);
  const int lb = log2_buckets_for(n), B = 1 << lb;
  char *ws = static_cast<char *>(workspace);
  uint2 *pairs = reinterpret_cast<uint2 *>(ws);
  ws += align_up(8 * (size_t)n);
  uint2 *pairs2 = reinterpret_cast<uint2 *>(ws);
  ws += align_up(8 * (size_t)n);
  unsigned *table = reinterpret_cast<unsigned *>(ws);
  ws += align_up(4 * (size_t)chunks * B);
  unsigned *totals = reinterpret_cast<unsigned *>(ws);
  ws += align_up(4 * (size_t)B);
  unsigned *bucket_base = reinterpret_cast<unsigned *>(ws);
  ws += align_up(4 * ((size_t)B + 1));
  OctaveMap *map = reinterpret_cast<OctaveMap *>(ws);
  ws += 4096;
  if (!stats) stats = reinterpret_cast<int *>(ws);
  ws += 256;
  unsigned *scan_state = reinterpret_cast<unsigned *>(ws);
  const int state_words = counts ? gsr_sort_mid_scan_state_words((long long)n * rows) : 0;
  const RecordArgs ra{xys, conics, opacities, tiles_x, tiles_y, static_cast<SplatRec *>(recs)};
  const Gather ga{counts, cum, counts ? rows : 0, n};
  if (xys)
    hipLaunchKernelGGL(hist_kernel<true>, dim3(chunks), dim3(kHistThreads), sizeof(unsigned) * B, s, n, depths, radii,
                       lb, table, map, ra, scan_state, state_words);
  else
    hipLaunchKernelGGL(hist_kernel<false>, dim3(chunks), dim3(kHistThreads), sizeof(unsigned) * B, s, n, depths, radii,
                       lb, table, map, ra, scan_state, state_words);
  hipLaunchKernelGGL(scan_kernel, dim3(B / 16), dim3(256), 0, s, chunks, B, table, totals);
  hipLaunchKernelGGL(scatter_kernel, dim3(chunks), dim3(kScatterThreads), 0, s, n, depths, radii,
                     (const OctaveMap *)map, lb, (const unsigned *)table, (const unsigned *)totals, pairs, order,
                     bucket_base, stats, ga);
  hipLaunchKernelGGL(bucket_sort_kernel, dim3(gsr_cdiv((unsigned)(B - 1), 4)), dim3(kSortThreads), 0, s,
                     B, (const unsigned *)bucket_base, pairs, pairs2, order, ga);
  GSR_CHECK_LAUNCH("sort_bucket_depth");
  if (counts) return gsr_sort_mid_scan_inplace((long long)n * rows, cum, scan_state, s);
  return GSR_OK;
}
