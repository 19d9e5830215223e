// tile_scatter.hip -- stable single-pass counting sort of the depth-ordered
// (tile id, Gaussian id) stream by tile id, gfx950.
//
// The stream is already ordered by depth (binning_fast.hip), so all that is left
// of `bin_and_sort_gaussians` (rasterizer/utils.py:128-182) is a STABLE partition
// by tile.  A general radix sort does that in two 6/7-bit passes over 8-byte
// pairs plus histogram / fill launches (rocPRIM: 160 us for 4.5 M pairs, T = 8160
// on MI355X).  With T <= 16384 tiles one pass is enough, because a whole row of
// per-tile counters fits in LDS (4 T bytes):
//
//   S1 hist      one workgroup per chunk of the stream: LDS histogram over all T
//                tiles -> table[chunk][T]
//   S2a colscan  per (tile, group of chunks): exclusive prefix down the chunk axis
//   S2b groups   per tile: prefix over the groups, tile totals
//   S2c bases    one workgroup: prefix over the tile totals -> tile_bins (the
//                ranges the compositing kernels walk) and each tile's first slot
//   S3 scatter   ONE WAVE per chunk: loads its row of absolute offsets into LDS
//                and walks its elements in stream order; the slot of an element is
//                ds_add_rtn(off[tile], 1).  Within one 64-lane step the LDS unit
//                applies same-address adds in unspecified order, so lanes that
//                hit the same tile in the same step (detected by reading the
//                counter back) are re-ranked by lane index -- which is stream
//                order.  Everything else is ordered by the wave's program order.
//
// Only the Gaussian ids are written (4 B / pair); the sorted tile keys a radix
// sort would also produce are never materialised, and tile_bins falls out of S2b
// instead of a separate edge-detection pass.  The output (18 MB at 4.5 M pairs)
// sits in the 256 MB Infinity Cache, which is what makes the 4-byte scattered
// writes of S3 affordable.
#include "gsr_common.h"

#ifndef GSR_TS_CHUNK
#define GSR_TS_CHUNK 4096
#endif

namespace gsr_ts {

constexpr int kMaxTiles = 16384;  // 64 KB of LDS counters
constexpr int kMinChunk = GSR_TS_CHUNK;   // stream elements per chunk (per wave in S3)
constexpr int kMaxChunks = 1024;
constexpr int kMaxGroups = 32;

struct Plan {
  int chunk, chunks, groups, chunks_per_group;
};

inline Plan make_plan(int I) {
  Plan p;
  long long c = kMinChunk;
  const long long need = ((long long)I + kMaxChunks - 1) / kMaxChunks;
  if (need > c) c = (need + 511) / 512 * 512;
  p.chunk = (int)c;
  p.chunks = (int)(((long long)I + c - 1) / c);
  if (p.chunks < 1) p.chunks = 1;
  p.groups = p.chunks < kMaxGroups ? p.chunks : kMaxGroups;
  p.chunks_per_group = (p.chunks + p.groups - 1) / p.groups;
  p.groups = (p.chunks + p.chunks_per_group - 1) / p.chunks_per_group;
  return p;
}

// `I_dev` (nullable): the stream length lives on the device; `I` is then the
// capacity the buffers were sized for and the stream is cut there.
__device__ __forceinline__ int stream_length(const int I, const int *I_dev) {
  if (!I_dev) return I;
  const int v = *I_dev;
  return v < I ? v : I;
}

__global__ __launch_bounds__(256) void hist_kernel(const int I_cap, const int *__restrict__ I_dev,
                                                   const int chunk, const int T,
                                                   const unsigned *__restrict__ keys,
                                                   unsigned *__restrict__ table, int *__restrict__ count_out) {
  extern __shared__ unsigned h[];
  const int I = stream_length(I_cap, I_dev);
  // the uncut length, for the caller's capacity check (count_out may be mapped host memory)
  if (count_out && blockIdx.x == 0 && threadIdx.x == 0) *count_out = I_dev ? *I_dev : I;
  const int tid = threadIdx.x, c = blockIdx.x;
  for (int t = tid; t < T; t += 256) h[t] = 0;
  __syncthreads();
  const long long beg = (long long)c * chunk;
  const int end = (int)(beg + chunk < (long long)I ? beg + chunk : (long long)I);
  for (int e = (int)beg + tid; e < end; e += 256) atomicAdd(&h[keys[e]], 1u);
  __syncthreads();
  unsigned *row = table + (size_t)c * T;
  for (int t = tid; t < T; t += 256) row[t] = h[t];
}

// exclusive prefix down the chunks of one group, per tile; gsum[group][t] = group total
__global__ __launch_bounds__(256) void colscan_kernel(const int T, const int chunks, const int chunks_per_group,
                                                      unsigned *__restrict__ table,
                                                      unsigned *__restrict__ gsum) {
  const int t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y;
  if (t >= T) return;
  const int c0 = g * chunks_per_group;
  const int c1 = c0 + chunks_per_group < chunks ? c0 + chunks_per_group : chunks;
  unsigned run = 0;
  int c = c0;
  for (; c + 8 <= c1; c += 8) {
    unsigned *p = table + (size_t)c * T + t;
    unsigned v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[(size_t)j * T];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      p[(size_t)j * T] = run;
      run += v[j];
    }
  }
  for (; c < c1; ++c) {
    unsigned *p = table + (size_t)c * T + t;
    const unsigned v = *p;
    *p = run;
    run += v;
  }
  gsum[(size_t)g * T + t] = run;
}

// per tile: exclusive prefix over the groups (in place) and the tile's total.
// All loads are issued before the first store (the compiler cannot hoist them
// past stores into the same array itself).
__global__ __launch_bounds__(256) void group_scan_kernel(const int T, const int groups,
                                                         unsigned *__restrict__ gsum,
                                                         unsigned *__restrict__ totals) {
  const int t = blockIdx.x * 256 + threadIdx.x;
  if (t >= T) return;
  unsigned v[kMaxGroups];
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) v[g] = g < groups ? gsum[(size_t)g * T + t] : 0u;
  unsigned run = 0;
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) {
    if (g < groups) gsum[(size_t)g * T + t] = run;
    run += v[g];
  }
  totals[t] = run;
}

// One workgroup of 1024 threads; thread i owns tiles i, i + 1024, ... (coalesced):
// totals[t] <- first slot of tile t; tile_bins[t] = [first, last) or (0, 0) for an
// empty tile (what the reference's zero-initialised tile_bins holds,
// bindings.cu:258); *total_out = I.
__global__ __launch_bounds__(1024) void bases_kernel(const int T, unsigned *__restrict__ totals,
                                                     int *__restrict__ tile_bins, int *__restrict__ total_out) {
  constexpr int kPer = kMaxTiles / 1024;  // 16 batches of 1024 tiles
  __shared__ unsigned wsum[kPer * 16];    // [batch][wave] sums, then their exclusive prefix
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned v[kPer], incl[kPer];
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int t = j * 1024 + tid;
    v[j] = t < T ? totals[t] : 0u;
    incl[j] = v[j];
  }
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const unsigned u = __shfl_up(incl[j], o);
      if (lane >= o) incl[j] += u;
    }
  }
  if (lane == 63) {
#pragma unroll
    for (int j = 0; j < kPer; ++j) wsum[j * 16 + w] = incl[j];
  }
  __syncthreads();
  if (w == 0) {  // exclusive scan of the 256 (batch, wave) sums: 4 per lane
    unsigned a[4], s = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      a[k] = wsum[lane * 4 + k];
      s += a[k];
    }
    unsigned p = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const unsigned u = __shfl_up(p, o);
      if (lane >= o) p += u;
    }
    if (lane == 63 && total_out) *total_out = (int)p;
    p -= s;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      wsum[lane * 4 + k] = p;
      p += a[k];
    }
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < kPer; ++j) {
    const int t = j * 1024 + tid;
    if (t < T) {
      const unsigned base = wsum[j * 16 + w] + incl[j] - v[j];
      totals[t] = base;
      tile_bins[2 * t] = v[j] ? (int)base : 0;
      tile_bins[2 * t + 1] = v[j] ? (int)(base + v[j]) : 0;
    }
  }
}

constexpr int kUnroll = 8;

// T is a multiple of 4 here (padded row stride): the rows are staged with 16-byte loads.
__global__ __launch_bounds__(256) void scatter_kernel(const int I_cap, const int *__restrict__ I_dev,
                                                      const int chunk, const int T,
                                                     const int chunks_per_group,
                                                     const unsigned *__restrict__ keys,
                                                     const int *__restrict__ gids,
                                                     const unsigned *__restrict__ table,
                                                     const unsigned *__restrict__ gsum,
                                                     const unsigned *__restrict__ tile_base,
                                                     const int chunks, int *__restrict__ ids_out) {
  extern __shared__ unsigned off[];
  const int lane = threadIdx.x & 63;
  // Workgroup b runs on XCD b % 8.  Give each XCD a contiguous range of chunks:
  // the slots one tile's list receives from consecutive chunks are adjacent, so
  // each XCD fills its own ~1/8 of every list and its 4-byte writes merge into
  // full lines in that XCD's L2 instead of leaving 8 partially written copies.
  const int I = stream_length(I_cap, I_dev);
  // (over the chunks the stream really fills: with a device-side length the grid is
  // sized for the capacity and the tail chunks are empty)
  const int used = (int)(((long long)I + chunk - 1) / chunk);
  const int per_xcd = (used + 7) >> 3;
  const int slot = (int)blockIdx.x >> 3;
  const int c = ((int)blockIdx.x & 7) * per_xcd + slot;
  if (slot >= per_xcd || c >= used) return;
  const unsigned *row = table + (size_t)c * T;
  const unsigned *grow = gsum + (size_t)(c / chunks_per_group) * T;
  const long long beg = (long long)c * chunk;
  const int end = (int)(beg + chunk < (long long)I ? beg + chunk : (long long)I);
  const unsigned long long lt = (1ull << lane) - 1ull;
  int key_bits = 1;  // bits that distinguish tile ids
  while ((1 << key_bits) < T) ++key_bits;
  // One 64-lane step: slot = ds_add_rtn, conflicts re-ranked by lane (see the header).
  auto place = [&](const unsigned key, const int gid, const bool live) {
    unsigned old = 0, cur = 1;
    if (live) {
      old = atomicAdd(&off[key], 1u);
      // the wave's adds of this step are all applied before this load issues
      cur = __hip_atomic_load(&off[key], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    unsigned pos = old;
    // cur != old + 1: another lane hit this tile in this step (true for every lane
    // of such a group except the one whose add was applied last)
    const bool conf = cur != old + 1u;
    unsigned long long cm = __ballot(conf);
    if (__popcll(cm) >= 8) {
      // many lanes collided (spatially coherent scenes: consecutive Gaussians cover
      // the same tiles): one pass of wave-wide key matching, whose cost does not
      // depend on the number of distinct tiles involved
      unsigned long long peers = __ballot(live);
      for (int bit = 0; bit < key_bits; ++bit) {
        const bool b = (key >> bit) & 1u;
        const unsigned long long set = __ballot(b);
        peers &= b ? set : ~set;
      }
      if (live) pos = cur - (unsigned)__popcll(peers) + (unsigned)__popcll(peers & lt);
      cm = 0;
    }
    while (cm) {
      const int leader = __ffsll((long long)cm) - 1;
      const unsigned k = (unsigned)__shfl((int)key, leader);
      const bool mine = live && key == k;  // the whole group, flagged or not
      const unsigned long long same = __ballot(mine);
      if (mine) pos = cur - (unsigned)__popcll(same) + (unsigned)__popcll(same & lt);
      cm &= ~same;
    }
    if (live) ids_out[pos] = gid;
  };
  // Full batches of kUnroll x 64 elements, software-pipelined: the next batch is
  // loaded before the current one is placed.  The loop body is straight-line
  // (unconditional loads and stores) so that the wait for the prefetched batch is
  // `vmcnt(kUnroll)` -- it must not wait for the scattered stores issued after it.
  const int nfull = end > (int)beg ? (end - (int)beg) / (kUnroll * 64) : 0;
  unsigned nkey[kUnroll];
  int ngid[kUnroll];
  if (nfull > 0 && threadIdx.x < 64) {
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      nkey[u] = keys[(int)beg + u * 64 + lane];
      ngid[u] = gids[(int)beg + u * 64 + lane];
    }
  }
  // All four waves stage the chunk's row of absolute offsets (a single wave doing
  // this alone spent more time here than in the walk below); then wave 0 walks.
  {
    const uint4 *r4 = reinterpret_cast<const uint4 *>(row), *g4 = reinterpret_cast<const uint4 *>(grow),
                *b4 = reinterpret_cast<const uint4 *>(tile_base);
    uint4 *o4 = reinterpret_cast<uint4 *>(off);
#pragma unroll 4
    for (int q = threadIdx.x; q < (T >> 2); q += 256) {
      const uint4 a = r4[q], g = g4[q], t = b4[q];
      o4[q] = make_uint4(a.x + g.x + t.x, a.y + g.y + t.y, a.z + g.z + t.z, a.w + g.w + t.w);
    }
  }
  __syncthreads();
  if (threadIdx.x >= 64) return;
  for (int b = 0; b < nfull; ++b) {
    unsigned key[kUnroll];
    int gid[kUnroll];
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      key[u] = nkey[u];
      gid[u] = ngid[u];
    }
    const int nb = b + 1 < nfull ? b + 1 : b;  // the last batch is re-read, unused
    const int base = (int)beg + nb * (kUnroll * 64) + lane;
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) {
      nkey[u] = keys[base + u * 64];
      ngid[u] = gids[base + u * 64];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) place(key[u], gid[u], true);
  }
  for (int e0 = (int)beg + nfull * (kUnroll * 64); e0 < end; e0 += 64) {
    const int e = e0 + lane;
    const bool live = e < end;
    place(live ? keys[e] : 0u, live ? gids[e] : 0, live);
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace gsr_ts

// ---- internal interface used by binning_fast.hip ---------------------------
bool gsr_tile_scatter_supported(int num_tiles) { return num_tiles > 0 && num_tiles <= gsr_ts::kMaxTiles; }

size_t gsr_tile_scatter_workspace_bytes(int I, int num_tiles) {
  using namespace gsr_ts;
  const Plan p = make_plan(I);
  num_tiles = (num_tiles + 3) & ~3;
  return align_up(4 * (size_t)p.chunks * num_tiles) + align_up(4 * (size_t)p.groups * num_tiles) +
         align_up(4 * (size_t)num_tiles);
}

// keys[I] (tile ids < num_tiles) / gids[I] in stream order -> ids_sorted[I] stably
// ordered by tile, tile_bins[num_tiles][2].  With I_dev (device int) the stream length is min(*I_dev, I): the
// caller sized the buffers for I without knowing the length on the host; count_out
// (device-accessible int, may be null) then receives *I_dev, uncut.
int gsr_tile_scatter(int I, const int *I_dev, const unsigned *keys, const int *gids, int num_tiles,
                     int *ids_sorted, int *tile_bins, int *count_out, void *workspace, size_t workspace_bytes,
                     hipStream_t s) {
  using namespace gsr_ts;
  if (!gsr_tile_scatter_supported(num_tiles)) {
    gsr_set_error("tile_scatter: %d tiles > %d", num_tiles, kMaxTiles);
    return GSR_EINVAL;
  }
  if (workspace_bytes < gsr_tile_scatter_workspace_bytes(I, num_tiles)) {
    gsr_set_error("tile_scatter: workspace too small");
    return GSR_ENOMEM;
  }
  const Plan p = make_plan(I);
  const int real_tiles = num_tiles;
  num_tiles = (num_tiles + 3) & ~3;  // row stride; the padding columns stay zero
  unsigned *table = static_cast<unsigned *>(workspace);
  unsigned *gsum = reinterpret_cast<unsigned *>(static_cast<char *>(workspace) +
                                                align_up(4 * (size_t)p.chunks * num_tiles));
  unsigned *totals = reinterpret_cast<unsigned *>(reinterpret_cast<char *>(gsum) +
                                                  align_up(4 * (size_t)p.groups * num_tiles));
  const size_t lds = 4 * (size_t)num_tiles;
  hipLaunchKernelGGL(hist_kernel, dim3(p.chunks), dim3(256), lds, s, I, I_dev, p.chunk, num_tiles, keys, table,
                     count_out);
  hipLaunchKernelGGL(colscan_kernel, dim3(gsr_cdiv(num_tiles, 256), p.groups), dim3(256), 0, s, num_tiles,
                     p.chunks, p.chunks_per_group, table, gsum);
  hipLaunchKernelGGL(group_scan_kernel, dim3(gsr_cdiv(num_tiles, 256)), dim3(256), 0, s, num_tiles, p.groups,
                     gsum, totals);
  hipLaunchKernelGGL(bases_kernel, dim3(1), dim3(1024), 0, s, real_tiles, totals, tile_bins, (int *)nullptr);
  hipLaunchKernelGGL(scatter_kernel, dim3(8 * gsr_cdiv(p.chunks, 8)), dim3(256), lds, s, I, I_dev, p.chunk,
                     num_tiles, p.chunks_per_group, keys, gids, (const unsigned *)table, (const unsigned *)gsum,
                     (const unsigned *)totals, p.chunks, ids_sorted);
  GSR_CHECK_LAUNCH("tile_scatter");
  return GSR_OK;
}
