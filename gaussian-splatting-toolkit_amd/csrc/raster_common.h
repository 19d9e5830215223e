// raster_common.h -- pieces shared by the two wave-per-tile compositing kernels.
#pragma once
#include "gsr_common.h"

namespace gsr {

constexpr int kChunk = 64;  // splats staged per LDS fill (one per lane)

// 1: the tile16 compositing loops keep the NEXT chunk's global loads in flight while they walk the current one
// (stage_load_id / stage_load_attrs / stage_commit below); 0: stage_chunk in front of every chunk (A/B builds)
#ifndef GSR_STAGE_AHEAD
#define GSR_STAGE_AHEAD 1
#endif

// LDS record of one staged splat; consumed by wave-uniform broadcast reads.
struct __align__(16) SplatA { float x, y, ha, b; };     // ha = a/2
struct __align__(16) SplatB { float hc, opac, r, g; };  // hc = c/2
struct __align__(16) SplatC { float blue; int sidx; int mask; float extra; };  // extra: optional 4th channel
// (0.5*(a dx^2 + c dy^2) == (a/2) dx^2 + (c/2) dy^2 exactly: scaling by a power
//  of two commutes with rounding)
// sidx = index in the sorted list; mask = which of the tile's four 8x8
// sub-tiles the splat can reach (bit p <-> sub-tile (p&1, p>>1)).

// Pixel ownership inside a 16x16 tile: lane l sits at (l&7, l>>3) of EVERY 8x8
// sub-tile, i.e. it owns the 4 pixels (l&7 + 8*(p&1), (l>>3) + 8*(p>>1)),
// p = 0..3.  One wave-instruction therefore covers a whole sub-tile, and a
// sub-tile a splat cannot reach is skipped with a scalar branch.

// Can ANY pixel centre of the (w+1) x (w+1) pixel square whose first pixel is
// (rx0, ry0) get alpha = opac*exp(-sigma) >= 1/255 from this splat?  The tile
// lists of the reference are built from the 3-sigma *square* bounding box
// (forward.cu:73): about half of a tile's entries never pass the alpha test at
// any of its pixels, and of the rest ~1/3 of the (splat, sub-tile) pairs don't
// either.  Dropping them cannot change a result: the compositing rule skips them
// pixel by pixel (alpha < 1/255 -> continue).  The test is conservative: sigma is
// minimised over the continuous rectangle (<= its minimum over the pixel
// centres) and `smax` carries a 1 % margin in alpha, far above fp32 rounding.
struct Reach {
  float x, y, a, b, c;
  float nb_c, nb_a;  // -b/c, -b/a
  float smax;        // sigma bound: +inf = always keep, < 0 = can never reach
};

__device__ __forceinline__ Reach make_reach(float x, float y, float a, float b, float c, float opac) {
  Reach r{x, y, a, b, c, 0.f, 0.f, 0.f};
  // alpha >= 1/255  <=>  sigma <= log(255*opac)
  r.smax = __logf(255.f * opac) + 0.01f;
  if (!(r.smax >= 0.f)) r.smax = (opac == opac) ? -1.f : INFINITY;  // too faint anywhere (NaN: keep)
  if (!(a > 0.f && c > 0.f && a * c - b * b > 0.f)) {
    r.smax = INFINITY;  // not positive definite: no culling
  } else {
    r.nb_c = -b / c;
    r.nb_a = -b / a;
  }
  return r;
}

__device__ __forceinline__ bool reaches_rect(const Reach &r, float rx0, float ry0, float w) {
  if (r.smax == INFINITY) return true;
  if (r.smax < 0.f) return false;
  const float u0 = rx0 - r.x, u1 = u0 + w, v0 = ry0 - r.y, v1 = v0 + w;
  if (u0 <= 0.f && u1 >= 0.f && v0 <= 0.f && v1 >= 0.f) return true;  // centre inside
  // convex quadratic, unconstrained minimum (0,0) outside the rectangle ->
  // the minimum over the rectangle lies on one of its four edges
  auto edge_u = [&](float ue) {
    const float v = fminf(fmaxf(r.nb_c * ue, v0), v1);
    return 0.5f * (r.a * ue * ue + r.c * v * v) + r.b * ue * v;
  };
  auto edge_v = [&](float ve) {
    const float u = fminf(fmaxf(r.nb_a * ve, u0), u1);
    return 0.5f * (r.a * u * u + r.c * ve * ve) + r.b * u * ve;
  };
  const float smin = fminf(fminf(edge_u(u0), edge_u(u1)), fminf(edge_v(v0), edge_v(v1)));
  return smin <= r.smax;
}

// 4-bit reach mask over the sub-tiles of the tile at (tx0, ty0); 0 = drop.
__device__ __forceinline__ int splat_reach_mask(float x, float y, float a, float b, float c,
                                                float opac, float tx0, float ty0) {
  const Reach r = make_reach(x, y, a, b, c, opac);
  int m = 0;
#pragma unroll
  for (int p = 0; p < 4; ++p)
    m |= reaches_rect(r, tx0 + 8.f * (p & 1), ty0 + 8.f * (p >> 1), 7.f) ? (1 << p) : 0;
  return m;
}

// ---- deep tiles ---------------------------------------------------------------
// One wave per tile makes a kernel as slow as its deepest tiles: a tile whose list is
// several times the average keeps one wave busy long after the rest of the chip has
// drained (clustered scenes: 10 % of the tiles with 10x the depth).  With
// deep_threshold > 0 the launch carries three more blocks per tile slot; for a tile
// whose list is longer than the threshold the four blocks each take ONE 8x8 sub-tile
// (the regular block sub-tile 0), for every other tile the extra blocks exit at once.
// A sub-tile wave runs the very same per-pixel instructions on its 64 pixels (the other
// three sub-tiles are masked off wave-uniformly) and stages only the splats that reach
// its sub-tile, so results are bit-identical to the one-wave-per-tile walk.
// Block b of the extra range [base_grid, 4 base_grid) sits on XCD b % 8 like the regular
// block of the same tile (the four waves share the splat records in one L2).
// ORDER (round 5).  The hardware starts workgroups in block order, and a launch lasts as long as its last wave: with a
// static block -> tile map the waves that start last are whatever tiles sit at the end of the map, and on a scene whose
// lists differ in length (every trained model) the second half of the launch is a few long walks on an emptying chip
// (tools/exp/wave_trace.py, profiles/r05_wave_trace.txt: 50 % of the waves done after 40 % of the span, the last 1 %
// take the final 20 %).  With GSR_DEEP_ORDERED in deep_tile_threshold the entry point first builds a JOB ORDER in the
// 4 base_grid ints behind tile_bins (tile_jobs_kernel): per XCD -- a block's XCD is b % 8, and a tile keeps the XCD
// the static map gives it, so the L2 that holds a neighbourhood's splats stays the same -- its tiles' jobs (the whole
// tile, or four sub-tile jobs above the threshold) sorted by estimated work, LONGEST FIRST; block b runs job[b].
// Without the flag: the static map, regular blocks first, the extra sub-tile blocks behind them.
struct TileJob {
  int tile;      // < 0: nothing to do
  int allowed;   // sub-tile mask this wave owns (15 = the whole tile)
  float *stats;  // ordered launches: where the waves leave what the NEXT launch's order learns from (job_stats_end)
};
constexpr int kJobTileBits = 27;  // job = tile | allowed << 27; -1 = none
// What a sub-tile wave costs per list entry relative to a whole-tile wave is a property of the scene (0.2-0.25 on a
// trained model, 0.04 on the long-tail cloud, 0.03 on the trainer's ball of translucent Gaussians whose 10 000-entry
// tiles saturate after a few hundred entries) -- and it decides where a split tile's jobs belong in a longest-first
// order: with a fixed length / 4 the ball's rim tiles (whole, 1 100 entries, 310 us: the launch's critical path)
// started behind the centre's sub-tile jobs and the launch grew 6 %; with length / 16 the trained model's and the
// long-tail cloud's gains halved.  So the launches MEASURE it: every wave of an ordered launch adds its duration and its
// tile's list length to four floats per direction (whole / sub-tile; a per-device buffer of the library whose address
// the order kernel leaves behind the job arrays), and the next order kernel of that direction keys split tiles'
// jobs by  length x (sub-tile time per entry) / (whole-tile time per entry),  then halves the sums (an exponential
// average over launches).  No host involvement; a benign race at worst between streams (the order only schedules).
struct JobStats {     // one per (direction, XCD): 64 bytes, a cache line of its own
  float dur[2];       // [0] whole-tile waves, [1] sub-tile waves: summed wave durations (100-MHz ticks)
  float len[2];       // ... and the summed list lengths of their tiles
  float pad[12];
};
constexpr int kJobStatsFloats = 2 * 8 * 16;
// Every 8th job of an XCD's order reports, to its XCD's own line: 16 k atomics per launch on ONE line cost 150 us
// (they serialise at ~6 ns each across the eight L2s) -- measured, the first version of this.
__device__ __forceinline__ void job_stats_end(const TileJob &j, const int dir, const unsigned long long t0, const int len) {
  if (j.stats && threadIdx.x == 0 && len > 0 && ((blockIdx.x >> 3) & 7u) == 0u) {
    const int cls = j.allowed == 15 ? 0 : 1;
    JobStats *st = reinterpret_cast<JobStats *>(j.stats) + dir * 8 + (blockIdx.x & 7u);
    unsafeAtomicAdd(&st->dur[cls], (float)(wall_clock64() - t0));
    unsafeAtomicAdd(&st->len[cls], (float)len);
  }
}
__device__ __forceinline__ TileJob tile_job(const unsigned b, const unsigned base_grid, const int tiles_x,
                                            const int tiles_y, const int2 *__restrict__ tile_bins,
                                            const int deep_arg, int2 &range) {
  TileJob j{-1, 15, nullptr};
  const int deep_threshold = gsr_deep_threshold(deep_arg);
  if (gsr_deep_ordered(deep_arg)) {
    const int *tail = reinterpret_cast<const int *>(tile_bins + (size_t)tiles_x * tiles_y);
    const int code = tail[(size_t)gsr_deep_second(deep_arg) * 4u * base_grid + b];
    j.stats = *reinterpret_cast<float *const *>(tail + 8u * (size_t)base_grid);
    if (code < 0 || (code & ((1 << kJobTileBits) - 1)) >= tiles_x * tiles_y) return j;  // (none / not a job order)
    j.tile = code & ((1 << kJobTileBits) - 1);
    j.allowed = code >> kJobTileBits;
    range = tile_bins[j.tile];
    return j;
  }
  if (b < base_grid) {
    j.tile = gsr_xcd_remap(b, tiles_x, tiles_y);
    if (j.tile < 0) return j;
    range = tile_bins[j.tile];
    if (deep_threshold > 0 && range.y - range.x > deep_threshold) j.allowed = 1;
    return j;
  }
  const unsigned e = b - base_grid, x = e & 7u, q = e >> 3;
  const int tile = gsr_xcd_remap((q / 3u) * 8u + x, tiles_x, tiles_y);
  if (tile < 0) return j;
  range = tile_bins[tile];
  if (range.y - range.x <= deep_threshold) return j;
  j.tile = tile;
  j.allowed = 2 << (q % 3u);
  return j;
}

// (s_setprio tiers by estimated work were built and measured -- no gain: tools/exp/wave_prio.patch,
//  profiles/r05_wave_priority_ab.txt)

// The job order of one launch: one workgroup of 1024 lanes per XCD.  Items: every tile the static map gives this XCD
// -- one job (tile, 15) keyed by its list length, or, above the threshold, four jobs (tile, 1 << p) keyed by
// length x the measured cost ratio of a sub-tile wave (JobStats above).  Counting sort, descending, on HALF-OCTAVE buckets of the
// key (bucket = floor(2 log2 key)): jobs inside a bucket differ by < 1.42 x in length and keep -- up to the order in
// which the waves' LDS atomics land, i.e. in runs of 64 slots -- the static map's SPATIAL order.  That matters: with a
// fine-grained sort (2 048 linear buckets, the first version) a scene whose lists are all alike was shuffled into a
// random spatial order for nothing, the tiles running together no longer shared their splats in the XCD's L2, and
// the uniform bench scene lost 4 % (profiles/r05_lpt_ab.txt); with half-octaves its tiles fall into one or two
// buckets and the launch is the static one.
constexpr int kJobBuckets = 64;
__device__ __forceinline__ int job_bucket(const int key) {  // larger keys -> smaller bucket index (sorted first)
  if (key <= 0) return kJobBuckets - 1;
  const int k = 31 - __clz(key);                                          // floor(log2 key)
  const int half = ((unsigned long long)key * (unsigned)key) >> (2 * k + 1);  // key^2 >= 2^(2k+1)  <=>  key >= 2^k sqrt 2
  return max(0, kJobBuckets - 2 - (2 * k + (half ? 1 : 0)));
}
// tail64 > 0: the last tail64 / 64 of this XCD's whole-tile jobs (the shortest ones; on a scene whose lists are all
// alike simply the last ones) are cut into four sub-tile jobs each and run behind everything else: a launch of N
// equal jobs on S slots ends with a drain of one job's length over which the chip empties (wave_trace.py: the last
// 20 % of the uniform scene's launches run below 1.6 waves per SIMD, where a SIMD no longer saturates its VALU);
// quarter-length jobs at the end shorten it.  (Forward only by default: a split tile costs the backward 1.7 x the
// instructions -- the butterfly per sub-tile wave -- and it loses what the drain gains.)
// The sort is STABLE and deterministic: inside a bucket the jobs keep the static map's order (ranks by wave-wide key
// matching, a per-(wave chunk, bucket) table, one prefix down the chunks) -- spatial neighbours stay neighbours in time.
constexpr int kJobChunks = 64;  // wave chunks of 64 slots per XCD: up to 4 096 tile slots per XCD (32 768 in all: 3840 x 2160 is 32 400 tiles)
static __global__ __launch_bounds__(1024) void tile_jobs_kernel(const int tiles_x, const int tiles_y,
                                                                const unsigned base_grid,
                                                                const int2 *__restrict__ tile_bins,
                                                                const int deep_arg0, const int deep_arg1,
                                                                int *__restrict__ jobs_base, float *stats,
                                                                const float fixed_ratio) {
  // workgroups 0-7: the order deep_arg0 describes (threshold, tail, which array); 8-15: deep_arg1's, if launched
  const int deep_arg = blockIdx.x < 8 ? deep_arg0 : deep_arg1;
  int deep_threshold = gsr_deep_threshold(deep_arg);
  const int tail64 = gsr_deep_tail64(deep_arg);
  int *const jobs = jobs_base + (size_t)gsr_deep_second(deep_arg) * 4u * base_grid;
  // a split tile's jobs are keyed by length x ratio: measured by the previous launches of this direction (JobStats),
  // 1/8 until there is a measurement, `fixed_ratio` > 0 when the caller pins it (GSR_DEEP_SPLIT_KEY)
  float ratio = 0.125f;
  if (stats) {
    const JobStats *st = reinterpret_cast<const JobStats *>(stats) + gsr_deep_second(deep_arg) * 8;
    float d0 = 0.f, d1 = 0.f, l0 = 0.f, l1 = 0.f;
#pragma unroll
    for (int x = 0; x < 8; ++x) d0 += st[x].dur[0], d1 += st[x].dur[1], l0 += st[x].len[0], l1 += st[x].len[1];
    if (l0 > 0.f && l1 > 0.f && d0 > 0.f && d1 > 0.f) ratio = fminf(0.5f, fmaxf(1.f / 64.f, (d1 / l1) / (d0 / l0)));
  }
  if (fixed_ratio > 0.f) ratio = fixed_ratio;
  // bucket-major tables: a wave scans one bucket's chunks with its lanes (conflict-free rows)
  __shared__ int cnt[kJobBuckets][kJobChunks];   // slots taken by chunk c's jobs of bucket q -> their offset in the bucket
  __shared__ int cntw[kJobBuckets][kJobChunks];  // the same for whole-tile jobs only (their rank among themselves)
  __shared__ int tot[kJobBuckets], totw[kJobBuckets];
  __shared__ int total_s, whole_s;
  __shared__ int longest_s, filled_s;
  __shared__ unsigned long long entries_s;
  const unsigned xcd = blockIdx.x & 7u, slots = base_grid / 8u;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nchunks = (int)((slots + 63u) >> 6);
  const unsigned long long below = (1ull << lane) - 1ull;
  // Lists that are ALL ALIKE (the longest within 1.5 x the mean of the non-empty ones: a random cloud, never a trained
  // model) leave nothing to balance: their tiles keep the static map's order (one bucket) and the BACKWARD splits none
  // of them, whatever the threshold says -- the grid-scaled backward threshold of small grids (rasterizer/cuda:
  // deep_tile_threshold) is below such a scene's mean, and four sub-tile waves per tile would cost it 1.7 x the
  // instructions for no shorter launch (1 M uniform Gaussians at 960 x 540: backward 0.33 -> 0.41 ms).
  if (tid == 0) longest_s = 0, filled_s = 0, entries_s = 0ull;
  __syncthreads();
  {
    int longest = 0, filled = 0;
    unsigned long long entries = 0ull;
    for (unsigned s = tid; s < slots; s += 1024u) {
      const int tile = gsr_xcd_remap(s * 8u + xcd, tiles_x, tiles_y);
      if (tile < 0) continue;
      const int2 r = tile_bins[tile];
      const int len = r.y - r.x;
      longest = max(longest, len);
      filled += len > 0;
      entries += (unsigned)max(len, 0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      longest = max(longest, __shfl_xor(longest, o));
      filled += __shfl_xor(filled, o);
      entries += __shfl_xor(entries, o);
    }
    if (lane == 0) {
      atomicMax(&longest_s, longest);
      atomicAdd(&filled_s, filled);
      atomicAdd(&entries_s, entries);
    }
  }
  __syncthreads();
  const bool alike = filled_s > 0 && 2ull * (unsigned long long)longest_s * (unsigned)filled_s <= 3ull * entries_s;
  if (alike && gsr_deep_second(deep_arg)) deep_threshold = max(deep_threshold, longest_s);
  for (int i = tid; i < kJobBuckets * nchunks; i += 1024) {
    cnt[i / nchunks][i % nchunks] = 0;
    cntw[i / nchunks][i % nchunks] = 0;
  }
  // one item per slot: bucket, weight (4 = a split tile's four jobs), and the lanes of its wave chunk that share its bucket
  auto item = [&](const unsigned s, int &tile, int &len, int &q, int &weight, unsigned long long &same,
                  unsigned long long &four) {
    tile = s < slots ? gsr_xcd_remap(s * 8u + xcd, tiles_x, tiles_y) : -1;
    len = 0;
    if (tile >= 0) {
      const int2 r = tile_bins[tile];
      len = r.y - r.x;
    }
    const bool split = tile >= 0 && deep_threshold > 0 && len > deep_threshold;
    weight = tile < 0 ? 0 : (split ? 4 : 1);
    q = tile < 0 ? kJobBuckets - 1 : alike ? 0 : job_bucket(split ? max(1, (int)((float)len * ratio)) : len);
    same = __ballot(tile >= 0);
#pragma unroll
    for (int bit = 0; bit < 6; ++bit) {
      const unsigned long long b = __ballot((q >> bit) & 1);
      same &= ((q >> bit) & 1) ? b : ~b;
    }
    four = __ballot(weight == 4) & same;
  };
  const unsigned rounds = (slots + 1023u) / 1024u;
  int tile0, len0, q0, weight0;  // round 0's item stays in registers (the only round up to 8 192 tiles)
  unsigned long long same0, four0;
  item(tid, tile0, len0, q0, weight0, same0, four0);
  __syncthreads();
  // pass 1: per (bucket, chunk) slot counts
  for (unsigned it = 0; it < rounds; ++it) {
    const unsigned s = it * 1024u + tid;
    const int c = (int)(s >> 6);
    int tile = tile0, len = len0, q = q0, weight = weight0;
    unsigned long long same = same0, four = four0;
    if (it) item(s, tile, len, q, weight, same, four);
    if (tile >= 0 && (same & below) == 0) {  // the first lane of its bucket in this chunk
      cnt[q][c] = __popcll(same) + 3 * __popcll(four);
      cntw[q][c] = __popcll(same & ~four);
    }
  }
  __syncthreads();
  // exclusive prefix down the chunks of every bucket: wave w takes buckets w, w + 16, ...; lane = chunk
  for (int q = wave; q < kJobBuckets; q += 16) {
    const int v = lane < nchunks ? cnt[q][lane] : 0, w = lane < nchunks ? cntw[q][lane] : 0;
    int incl = v, inclw = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o), u = __shfl_up(inclw, o);
      if (lane >= o) incl += t, inclw += u;
    }
    if (lane < nchunks) {
      cnt[q][lane] = incl - v;
      cntw[q][lane] = inclw - w;
    }
    if (lane == 63) tot[q] = incl, totw[q] = inclw;
  }
  __syncthreads();
  if (tid < kJobBuckets) {  // ... and over the buckets (one wave; lane = bucket)
    const int v = tot[tid], w = totw[tid];
    int incl = v, inclw = w;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o), u = __shfl_up(inclw, o);
      if (tid >= o) incl += t, inclw += u;
    }
    tot[tid] = incl - v;
    totw[tid] = inclw - w;
    if (tid == kJobBuckets - 1) total_s = incl, whole_s = inclw;
  }
  __syncthreads();
  const int total = total_s, whole = whole_s;
  // the whole-tile jobs ranked [whole - n_tail, whole) among themselves leave the order and come back as four sub-tile
  // jobs each behind it (as many as the 4 base_grid slots leave room for)
  const int n_tail = min((int)(((long long)whole * tail64) >> 6), max(0, (int)(4u * slots) - total) / 4);
  const int tail_first = whole - n_tail;
  // pass 2: every job to its slot
  for (unsigned it = 0; it < rounds; ++it) {
    const unsigned s = it * 1024u + tid;
    const int c = (int)(s >> 6);
    int tile = tile0, len = len0, q = q0, weight = weight0;
    unsigned long long same = same0, four = four0;
    if (it) item(s, tile, len, q, weight, same, four);
    if (tile < 0) continue;
    const unsigned long long before = same & below;
    const int at = tot[q] + cnt[q][c] + __popcll(before) + 3 * __popcll(before & four);
    if (weight == 4) {
#pragma unroll
      for (int p = 0; p < 4; ++p) jobs[(size_t)(at + p) * 8u + xcd] = tile | ((1 << p) << kJobTileBits);
      continue;
    }
    const int rank = totw[q] + cntw[q][c] + __popcll(before & ~four);
    const bool in_tail = n_tail > 0 && rank >= tail_first;
    if (in_tail && len > 0) {
      jobs[(size_t)at * 8u + xcd] = -1;
#pragma unroll
      for (int p = 0; p < 4; ++p)
        jobs[(size_t)(total + 4 * (rank - tail_first) + p) * 8u + xcd] = tile | ((1 << p) << kJobTileBits);
    } else {
      jobs[(size_t)at * 8u + xcd] = tile | (15 << kJobTileBits);
      if (in_tail) {  // (an empty tile in the tail stays whole: its four tail slots stay empty)
#pragma unroll
        for (int p = 0; p < 4; ++p) jobs[(size_t)(total + 4 * (rank - tail_first) + p) * 8u + xcd] = -1;
      }
    }
  }
  for (unsigned s = total + 4 * n_tail + tid; s < 4u * slots; s += 1024) jobs[(size_t)s * 8u + xcd] = -1;
  if (xcd == 0 && tid == 0) {
    // where the compositing waves find the statistics (behind both job arrays), and the exponential average: what the
    // earlier launches of this direction measured counts half from now on (every workgroup of this launch has read
    // the sums by the time they shrink? not guaranteed -- a workgroup that reads them late keys by a slightly other
    // ratio: it orders, it does not compute)
    *reinterpret_cast<float **>(jobs_base + 8u * (size_t)base_grid) = stats;
    if (stats) {
      JobStats *st = reinterpret_cast<JobStats *>(stats) + gsr_deep_second(deep_arg) * 8;
      for (int x = 0; x < 8; ++x) st[x].dur[0] *= 0.5f, st[x].dur[1] *= 0.5f, st[x].len[0] *= 0.5f, st[x].len[1] *= 0.5f;
    }
  }
}

float *gsr_job_stats_buffer(hipStream_t s);  // capi.hip: this device's JobStats[2] (nullptr if it cannot be had)
float gsr_job_split_ratio();    // capi.hip: GSR_DEEP_SPLIT_KEY (0 = measured)

// (host) build the job order behind tile_bins when deep_arg asks for it (and gsr_tile_jobs_build has not already);
// -> the argument the kernels take
static inline int gsr_prepare_jobs(const int deep_arg, const int tiles_x, const int tiles_y, const int32_t *tile_bins,
                                   hipStream_t s) {
  if (!gsr_deep_ordered(deep_arg)) return gsr_deep_threshold(deep_arg);
  const unsigned base = gsr_xcd_grid(tiles_x, tiles_y);
  if (base / 8u > (unsigned)kJobChunks * 64u) return gsr_deep_threshold(deep_arg);  // (beyond the sort's tables: static map)
  if (deep_arg & GSR_DEEP_PREBUILT) return deep_arg;
  int *jobs = const_cast<int *>(tile_bins) + 2 * (size_t)tiles_x * tiles_y;
  hipLaunchKernelGGL(tile_jobs_kernel, dim3(8), dim3(1024), 0, s, tiles_x, tiles_y, base,
                     reinterpret_cast<const int2 *>(tile_bins), deep_arg, 0, jobs, gsr_job_stats_buffer(s),
                     gsr_job_split_ratio());
  return deep_arg;
}

// Depth segments (raster_fwd.hip / raster_bwd.hip): a split tile's list of `len` entries is cut into at most K runs of
// whole 64-entry chunks; run k is [k seg_len, (k + 1) seg_len) of the list.
__device__ __forceinline__ int seg_len_of(const int len, const int K) { return ((len + K * 64 - 1) / (K * 64)) * 64; }

// ---- measurement hook (gsr_debug_wave_trace): one 32-byte record per wave that ran a tile to its end ------------
// {constant 100-MHz clock at entry, at exit, tile | allowed << 32, list length | hw id << 32}: who ran when and for
// how long -- the tail / imbalance of a launch (tools/exp/wave_trace.py).  hw id = HW_REG_HW_ID (SIMD [5:4], CU [11:8],
// SH [12], SE [15:13]) | XCC_ID << 20.
struct WaveTrace {
  unsigned long long *buf;
  unsigned capacity;
};
__device__ __forceinline__ unsigned long long trace_begin(const WaveTrace &tr) {
  return tr.buf ? wall_clock64() : 0ull;
}
__device__ __forceinline__ void trace_end(const WaveTrace &tr, const unsigned long long t0, const int tile,
                                          const int allowed, const int len) {
  if (tr.buf && blockIdx.x < tr.capacity && threadIdx.x == 0) {
    const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xcc = __builtin_amdgcn_s_getreg((3 << 11) | 20);
    unsigned long long *r = tr.buf + 4ull * blockIdx.x;
    r[0] = t0;
    r[1] = wall_clock64();
    r[2] = (unsigned long long)(unsigned)tile | ((unsigned long long)(unsigned)allowed << 32);
    r[3] = (unsigned long long)(unsigned)len | ((unsigned long long)((hw & 0xffffu) | (xcc << 20)) << 32);
  }
}

// Stage up to 64 splats (one per `live` lane, sorted index `sidx`) into LDS,
// dropping the ones that cannot reach the tile; returns how many were kept.
// Kept splats stay in lane order.  Wave-synchronous (one wave per workgroup).
__device__ __forceinline__ int stage_chunk(
    const int lane, const bool live, const int sidx, const float tx0, const float ty0,
    const int *__restrict__ ids_sorted, const float2 *__restrict__ xys,
    const float *__restrict__ conics, const float *__restrict__ colors,
    const float *__restrict__ opacities, SplatA *sA, SplatB *sB, SplatC *sC, int *sId,
    const float *__restrict__ extra = nullptr, unsigned long long *staged_counter = nullptr,
    const int allowed = 15) {
  // measurement hook (gsr_debug_count_staged): list entries read by this launch
  if (staged_counter) {
    const int n = __popcll(__ballot(live));
    if (lane == 0) atomicAdd(staged_counter, (unsigned long long)n);
  }
  int mask = 0;
  int g = 0;
  float2 xy = make_float2(0.f, 0.f);
  float a = 0.f, b = 0.f, c = 0.f, opac = 0.f;
  if (live) {
    g = ids_sorted[sidx];
    xy = xys[g];
    a = conics[3 * g];
    b = conics[3 * g + 1];
    c = conics[3 * g + 2];
    opac = opacities[g];
    mask = splat_reach_mask(xy.x, xy.y, a, b, c, opac, tx0, ty0);
#ifdef GSR_NO_CULL
    mask = 15;  // experiment: keep every list entry
#endif
    mask &= allowed;  // a wave that owns one sub-tile of a deep tile keeps only what reaches it
  }
  const unsigned long long kept = __ballot(mask != 0);
  if (mask != 0) {
    const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(kept >> 32),
                                               __builtin_amdgcn_mbcnt_lo((unsigned)kept, 0u));
    sA[slot] = SplatA{xy.x, xy.y, 0.5f * a, b};
    sB[slot] = SplatB{0.5f * c, opac, colors[3 * g], colors[3 * g + 1]};
    sC[slot] = SplatC{colors[3 * g + 2], sidx, mask, extra ? extra[g] : 0.f};
    if (sId) sId[slot] = g;
  }
  return __popcll(kept);
}

// ---- staging one chunk AHEAD (GSR_STAGE_AHEAD) -------------------------------------------
// stage_chunk is a chain of three dependent global loads (list entry -> geometry -> colours) in front of every 64 list
// entries; a wave that is alone on its SIMD -- the long walks a launch ends with -- waits it out every time.  Split in
// three so that the compositing loops can keep the loads of the NEXT chunk in flight while they walk the current one:
//   stage_load_id     the chunk's list entries (one per lane; lanes outside the range read nothing and get Gaussian 0)
//   stage_load_attrs  everything a splat record holds, UNCONDITIONALLY (colours of splats the reach test will drop
//                     included: 12 bytes more per dropped entry, no third round trip)
//   stage_commit      the reach test and the LDS records -- the same arithmetic, slots and order as stage_chunk.
struct StageRegs {
  float2 xy;
  float a, b, c, opac, red, green, blue, extra;
};
__device__ __forceinline__ int stage_load_id(const bool live, const int sidx, const int *__restrict__ ids_sorted) {
  return live ? ids_sorted[sidx] : 0;
}
__device__ __forceinline__ StageRegs stage_load_attrs(const int g, const float2 *__restrict__ xys,
                                                      const float *__restrict__ conics,
                                                      const float *__restrict__ colors,
                                                      const float *__restrict__ opacities,
                                                      const float *__restrict__ extra) {
  StageRegs s;
  s.xy = xys[g];
  s.a = conics[3 * g];
  s.b = conics[3 * g + 1];
  s.c = conics[3 * g + 2];
  s.opac = opacities[g];
  s.red = colors ? colors[3 * g] : 0.f;  // (the forward's transmittance pre-pass composites no colour)
  s.green = colors ? colors[3 * g + 1] : 0.f;
  s.blue = colors ? colors[3 * g + 2] : 0.f;
  s.extra = extra ? extra[g] : 0.f;
  return s;
}
__device__ __forceinline__ int stage_commit(const int lane, const bool live, const int sidx, const int g,
                                            const StageRegs &s, const float tx0, const float ty0, SplatA *sA,
                                            SplatB *sB, SplatC *sC, int *sId,
                                            unsigned long long *staged_counter = nullptr, const int allowed = 15) {
  if (staged_counter) {
    const int n = __popcll(__ballot(live));
    if (lane == 0) atomicAdd(staged_counter, (unsigned long long)n);
  }
  int mask = splat_reach_mask(s.xy.x, s.xy.y, s.a, s.b, s.c, s.opac, tx0, ty0);
#ifdef GSR_NO_CULL
  mask = 15;
#endif
  mask = live ? (mask & allowed) : 0;
  const unsigned long long kept = __ballot(mask != 0);
  if (mask != 0) {
    const int slot = __builtin_amdgcn_mbcnt_hi((unsigned)(kept >> 32),
                                               __builtin_amdgcn_mbcnt_lo((unsigned)kept, 0u));
    sA[slot] = SplatA{s.xy.x, s.xy.y, 0.5f * s.a, s.b};
    sB[slot] = SplatB{0.5f * s.c, s.opac, s.red, s.green};
    sC[slot] = SplatC{s.blue, sidx, mask, s.extra};
    if (sId) sId[slot] = g;
  }
  return __popcll(kept);
}

// One chunk in one go through the same three pieces (the pre-pass kernels of the depth segments, whose runs are a
// chunk or two long): two round trips instead of stage_chunk's three.
__device__ __forceinline__ int stage_chunk_flat(
    const int lane, const bool live, const int sidx, const float tx0, const float ty0,
    const int *__restrict__ ids_sorted, const float2 *__restrict__ xys, const float *__restrict__ conics,
    const float *__restrict__ colors, const float *__restrict__ opacities, SplatA *sA, SplatB *sB, SplatC *sC,
    const float *__restrict__ extra, const int allowed) {
  const int g = stage_load_id(live, sidx, ids_sorted);
  const StageRegs regs = stage_load_attrs(g, xys, conics, colors, opacities, extra);
  return stage_commit(lane, live, sidx, g, regs, tx0, ty0, sA, sB, sC, nullptr, nullptr, allowed);
}

}  // namespace gsr
