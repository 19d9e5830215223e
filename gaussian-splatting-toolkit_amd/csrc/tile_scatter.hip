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
// Tile-row BANDS (grids above 16384 tiles, e.g. 4K = 240 x 135 = 32400): the tile
// grid is cut into B bands of whole tile rows with <= 8192 tiles each, the stream
// arrives ordered by (band, depth) -- band b's segment is [cum[b n - 1], cum[(b+1) n - 1])
// of the scan binning_fast.hip makes over the per-band counts -- and every kernel
// works band by band (grid dimension y / z): the LDS counters stay at 32 KB, and
// the write frontier of the scatter (one partially filled line per tile) stays
// inside one XCD's 4-MB L2, so the 4-byte stores still leave the L2 as full lines
// (with all 32400 tiles in flight at once each store went out as its own 32-B
// granule: 2.95 ms for BASELINE config 5's 98 M entries).
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

constexpr int kMaxTiles = 16384;      // tiles per band: 64 KB of LDS counters
constexpr int kBandTiles = 8192;      // band size chosen for grids above kMaxTiles
constexpr int kMaxBands = 16;
constexpr int kMinChunk = GSR_TS_CHUNK;   // stream elements per chunk (per wave in S3)
constexpr int kMaxChunks = 3072;
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

// The stream segment of band b: [cum[b n - 1], cum[(b + 1) n - 1]) cut at the capacity
// `I` the buffers were sized for (`cum` = inclusive scan of the per-band tile counts in
// (band, depth) order, on the device; the host may not know the lengths).
struct Bands {
  const int *cum;
  int n, num, tiles_per_band, num_tiles;  // band b holds tiles [b tpb, min((b+1) tpb, num_tiles))
};
__device__ __forceinline__ void band_segment(const Bands &B, const int b, const int I, int &beg, int &end) {
  const int s = b ? B.cum[(size_t)b * B.n - 1] : 0, e = B.cum[(size_t)(b + 1) * B.n - 1];
  beg = s < I ? s : I;
  end = e < I ? e : I;
}

// grid (max chunks, bands); table[band][chunk][T] with T = padded tiles per band
__global__ __launch_bounds__(256) void hist_kernel(const int I_cap, const Bands B, const int chunk, const int T,
                                                   const int max_chunks, const unsigned *__restrict__ keys,
                                                   unsigned *__restrict__ table, int *__restrict__ count_out) {
  extern __shared__ unsigned h[];
  const int tid = threadIdx.x, c = blockIdx.x, b = blockIdx.y;
  // the uncut length, for the caller's capacity check (count_out may be mapped host memory)
  if (count_out && c == 0 && b == 0 && tid == 0) *count_out = B.cum[(size_t)B.num * B.n - 1];
  int sbeg, send;
  band_segment(B, b, I_cap, sbeg, send);
  const long long beg = (long long)sbeg + (long long)c * chunk;
  if (beg >= send) return;
  const int end = (int)(beg + chunk < (long long)send ? beg + chunk : (long long)send);
  for (int t = tid; t < T; t += 256) h[t] = 0;
  __syncthreads();
  const unsigned key0 = (unsigned)(b * B.tiles_per_band);
  for (int e = (int)beg + tid; e < end; e += 256) atomicAdd(&h[keys[e] - key0], 1u);
  __syncthreads();
  unsigned *row = table + ((size_t)b * max_chunks + c) * T;
  for (int t = tid; t < T; t += 256) row[t] = h[t];
}

// exclusive prefix down the chunks of one group, per tile; gsum[group][t] = group total
// grid (T / 256, groups, bands)
__global__ __launch_bounds__(256) void colscan_kernel(const int I_cap, const Bands B, const int chunk, const int T,
                                                      const int max_chunks, const int groups,
                                                      const int chunks_per_group, unsigned *__restrict__ table,
                                                      unsigned *__restrict__ gsum) {
  const int t = blockIdx.x * 256 + threadIdx.x, g = blockIdx.y, b = blockIdx.z;
  if (t >= T) return;
  int sbeg, send;
  band_segment(B, b, I_cap, sbeg, send);
  const int chunks = (int)(((long long)(send - sbeg) + chunk - 1) / chunk);  // chunks this band fills
  const int c0 = g * chunks_per_group;
  int c1 = c0 + chunks_per_group < chunks ? c0 + chunks_per_group : chunks;
  table += (size_t)b * max_chunks * T;
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
  gsum[((size_t)b * groups + g) * T + t] = run;
}

// per tile: exclusive prefix over the groups (in place) and the tile's total.
// All loads are issued before the first store (the compiler cannot hoist them
// past stores into the same array itself).
// grid (T / 256, bands); totals[band][T] (band-strided like the tables; padding stays 0)
__global__ __launch_bounds__(256) void group_scan_kernel(const Bands B, const int T, const int groups,
                                                         unsigned *__restrict__ gsum,
                                                         unsigned *__restrict__ totals) {
  const int t = blockIdx.x * 256 + threadIdx.x, b = blockIdx.y;
  if (t >= T) return;
  gsum += (size_t)b * groups * T;
  unsigned v[kMaxGroups];
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) v[g] = g < groups ? gsum[(size_t)g * T + t] : 0u;
  unsigned run = 0;
#pragma unroll
  for (int g = 0; g < kMaxGroups; ++g) {
    if (g < groups) gsum[(size_t)g * T + t] = run;
    run += v[g];
  }
  totals[(size_t)b * T + t] = run;
}

// One workgroup of 1024 threads; thread i owns entries i, i + 1024, ... (coalesced) of
// the band-strided totals[bands][stride], in rounds of 16384: totals <- first slot of
// that tile; tile_bins[tile] = [first, last) or (0, 0) for an empty tile (what the
// reference's zero-initialised tile_bins holds, bindings.cu:258).  Entry (b, t) is tile
// b * tiles_per_band + t when t < tiles_per_band and that is < num_tiles; the other
// entries are padding (count 0).
__global__ __launch_bounds__(1024) void bases_kernel(const int bands, const int stride, const int tiles_per_band,
                                                     const int num_tiles, unsigned *__restrict__ totals,
                                                     int *__restrict__ tile_bins) {
  const int T = bands * stride;
  constexpr int kPer = kMaxTiles / 1024;  // 16 batches of 1024 tiles per round
  __shared__ unsigned wsum[kPer * 16];    // [batch][wave] sums, then their exclusive prefix
  __shared__ unsigned round_total;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  unsigned carry = 0;
  for (int t0 = 0; t0 < T; t0 += kMaxTiles) {
    unsigned v[kPer], incl[kPer];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int t = t0 + j * 1024 + tid;
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
    __syncthreads();  // wsum / round_total of the previous round have been read
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
      if (lane == 63) round_total = p;
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
      const int t = t0 + j * 1024 + tid;
      if (t < T) {
        const unsigned base = carry + wsum[j * 16 + w] + incl[j] - v[j];
        totals[t] = base;
        const int b = t / stride, tl = t - b * stride, tile = b * tiles_per_band + tl;
        if (tl < tiles_per_band && tile < num_tiles) {
          tile_bins[2 * tile] = v[j] ? (int)base : 0;
          tile_bins[2 * tile + 1] = v[j] ? (int)(base + v[j]) : 0;
        }
      }
    }
    carry += round_total;
  }
}

constexpr int kUnroll = 8;

// T is a multiple of 4 here (padded row stride): the rows are staged with 16-byte loads.
// grid (8 * ceil(max chunks / 8), bands)
__global__ __launch_bounds__(256) void scatter_kernel(const int I_cap, const Bands B, const int chunk, const int T,
                                                      const int max_chunks, const int groups,
                                                      const int chunks_per_group,
                                                      const unsigned *__restrict__ keys,
                                                      const int *__restrict__ gids,
                                                      const unsigned *__restrict__ table,
                                                      const unsigned *__restrict__ gsum,
                                                      const unsigned *__restrict__ tile_base,
                                                      int *__restrict__ ids_out, int *__restrict__ slot_out) {
  extern __shared__ unsigned off[];
  const int lane = threadIdx.x & 63, b = blockIdx.y;
  // Workgroup x runs on XCD x % 8.  Give each XCD a contiguous range of chunks:
  // the slots one tile's list receives from consecutive chunks are adjacent, so
  // each XCD fills its own ~1/8 of every list and its 4-byte writes merge into
  // full lines in that XCD's L2 instead of leaving 8 partially written copies.
  int sbeg, send;
  band_segment(B, b, I_cap, sbeg, send);
  // (over the chunks the band's segment really fills: the grid is sized for the capacity
  // and the tail chunks are empty)
  const int used = (int)(((long long)(send - sbeg) + chunk - 1) / chunk);
  const int per_xcd = (used + 7) >> 3;
  const int slot = (int)blockIdx.x >> 3;
  const int c = ((int)blockIdx.x & 7) * per_xcd + slot;
  if (slot >= per_xcd || c >= used) return;
  const unsigned *row = table + ((size_t)b * max_chunks + c) * T;
  const unsigned *grow = gsum + ((size_t)b * groups + c / chunks_per_group) * T;
  const unsigned key0 = (unsigned)(b * B.tiles_per_band);
  tile_base += (size_t)b * T;
  const long long beg = (long long)sbeg + (long long)c * chunk;
  const int end = (int)(beg + chunk < (long long)send ? beg + chunk : (long long)send);
  const unsigned long long lt = (1ull << lane) - 1ull;
  int key_bits = 1;  // bits that distinguish tile ids
  while ((1 << key_bits) < T) ++key_bits;
  // One 64-lane step: slot = ds_add_rtn, conflicts re-ranked by lane (see the header).
  // slot_out (nullable): the final slot of stream element `eidx`, i.e. the inverse of the
  // scatter (deterministic backward: raster_bwd.hip reduces per Gaussian in stream order)
  auto place = [&](const unsigned key, const int gid, const bool live, const int eidx) {
    unsigned old = 0, cur = 1;
    if (live) {
      old = atomicAdd(&off[key], 1u);  // key: band-local tile id
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
    if (live) {
      ids_out[pos] = gid;
      if (slot_out) slot_out[eidx] = (int)pos;
    }
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
      nkey[u] = keys[(int)beg + u * 64 + lane] - key0;
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
      nkey[u] = keys[base + u * 64] - key0;
      ngid[u] = gids[base + u * 64];
    }
#pragma unroll
    for (int u = 0; u < kUnroll; ++u) place(key[u], gid[u], true, (int)beg + b * (kUnroll * 64) + u * 64 + lane);
  }
  for (int e0 = (int)beg + nfull * (kUnroll * 64); e0 < end; e0 += 64) {
    const int e = e0 + lane;
    const bool live = e < end;
    place(live ? keys[e] - key0 : 0u, live ? gids[e] : 0, live, e);
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

}  // namespace gsr_ts

// exclusive prefix over per-tile counts (in place) + tile_bins, for tile_partition2.hip
int gsr_tile_bases(int num_tiles, unsigned *totals, int *tile_bins, hipStream_t s) {
  hipLaunchKernelGGL(gsr_ts::bases_kernel, dim3(1), dim3(1024), 0, s, 1, num_tiles, num_tiles, num_tiles, totals,
                     tile_bins);
  GSR_CHECK_LAUNCH("tile_bases");
  return GSR_OK;
}

// ---- internal interface used by binning_fast.hip ---------------------------
// Number of tile-row bands for a tile grid, and the tile rows per band.  One band up to
// 16384 tiles; above, bands of whole tile rows holding <= 8192 tiles.  0 = unsupported
// (a single tile row above 8192 tiles, or more than 16 bands).
int gsr_tile_band_rows(int tiles_x, int tiles_y, int *rows_per_band) {
  using namespace gsr_ts;
  const long long T = (long long)tiles_x * tiles_y;
  if (tiles_x <= 0 || tiles_y <= 0) return 0;
  if (T <= kMaxTiles) {
    if (rows_per_band) *rows_per_band = tiles_y;
    return 1;
  }
  const int rpb = kBandTiles / tiles_x;
  if (rpb < 1) return 0;
  const int bands = (tiles_y + rpb - 1) / rpb;
  if (bands > kMaxBands) return 0;
  if (rows_per_band) *rows_per_band = rpb;
  return bands;
}

bool gsr_tile_scatter_supported(int tiles_x, int tiles_y) { return gsr_tile_band_rows(tiles_x, tiles_y, nullptr) > 0; }

size_t gsr_tile_scatter_workspace_bytes(int I, int tiles_per_band, int bands) {
  using namespace gsr_ts;
  const Plan p = make_plan(I);
  const size_t T = (size_t)((tiles_per_band + 3) & ~3);
  return align_up(4 * (size_t)bands * p.chunks * T) + align_up(4 * (size_t)bands * p.groups * T) +
         align_up(4 * (size_t)bands * T);
}

// keys[.] (global tile ids) / gids[.] in (band, stream) order -> ids_sorted stably ordered
// by tile, tile_bins[num_tiles][2].  The stream lengths live on the device: `cum` is the
// inclusive scan over the per-band counts of the n Gaussians in (band, depth) order, band
// b's segment of the stream is [cum[b n - 1], cum[(b + 1) n - 1]), cut at `I` (what the
// buffers were sized for); count_out (device-accessible int, may be null) receives the
// uncut total cum[bands n - 1]; slot_of_entry (may be null) the slot each stream element went to.
int gsr_tile_scatter(int I, const int *cum, int n, const unsigned *keys, const int *gids, int tiles_x, int tiles_y,
                     int *ids_sorted, int *tile_bins, int *count_out, int *slot_of_entry, void *workspace,
                     size_t workspace_bytes, hipStream_t s) {
  using namespace gsr_ts;
  int rpb = 0;
  const int bands = gsr_tile_band_rows(tiles_x, tiles_y, &rpb);
  if (bands < 1) {
    gsr_set_error("tile_scatter: a %d x %d tile grid is not supported", tiles_x, tiles_y);
    return GSR_EINVAL;
  }
  const int num_tiles = tiles_x * tiles_y, tpb = rpb * tiles_x;
  if (workspace_bytes < gsr_tile_scatter_workspace_bytes(I, tpb, bands)) {
    gsr_set_error("tile_scatter: workspace too small");
    return GSR_ENOMEM;
  }
  const Plan p = make_plan(I);
  const int T = (tpb + 3) & ~3;  // row stride; the padding columns stay zero
  const Bands B{cum, n, bands, tpb, num_tiles};
  char *ws = static_cast<char *>(workspace);
  unsigned *table = reinterpret_cast<unsigned *>(ws);
  ws += align_up(4 * (size_t)bands * p.chunks * T);
  unsigned *gsum = reinterpret_cast<unsigned *>(ws);
  ws += align_up(4 * (size_t)bands * p.groups * T);
  unsigned *totals = reinterpret_cast<unsigned *>(ws);
  const size_t lds = 4 * (size_t)T;
  hipLaunchKernelGGL(hist_kernel, dim3(p.chunks, bands), dim3(256), lds, s, I, B, p.chunk, T, p.chunks, keys, table,
                     count_out);
  hipLaunchKernelGGL(colscan_kernel, dim3(gsr_cdiv(T, 256), p.groups, bands), dim3(256), 0, s, I, B, p.chunk, T,
                     p.chunks, p.groups, p.chunks_per_group, table, gsum);
  hipLaunchKernelGGL(group_scan_kernel, dim3(gsr_cdiv(T, 256), bands), dim3(256), 0, s, B, T, p.groups, gsum, totals);
  hipLaunchKernelGGL(bases_kernel, dim3(1), dim3(1024), 0, s, bands, T, tpb, num_tiles, totals, tile_bins);
  hipLaunchKernelGGL(scatter_kernel, dim3(8 * gsr_cdiv(p.chunks, 8), bands), dim3(256), lds, s, I, B, p.chunk, T,
                     p.chunks, p.groups, p.chunks_per_group, keys, gids, (const unsigned *)table,
                     (const unsigned *)gsum, (const unsigned *)totals, ids_sorted, slot_of_entry);
  GSR_CHECK_LAUNCH("tile_scatter");
  return GSR_OK;
}
