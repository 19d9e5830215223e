// tile_partition2.hip -- the per-tile depth-sorted lists of LARGE tile grids (above 16384
// tiles: 4K at 16 px = 240 x 135) by a two-level partition whose every global store is
// coalesced, gfx950.
//
// Why: the single-pass tile scatter (tile_scatter.hip) writes each list entry as one 4-byte
// store to its own cache line.  The L2 retires such partial-line writes at ~70-85 M per ms
// chip-wide whatever the band size: 1.39 ms for BASELINE config 5's 97.7 M
// entries, with 8x write amplification.  Here no entry is ever written alone:
//
//   level A, by tile ROW, fused into the emission (no intermediate depth-ordered stream):
//     P1 rowcount   one wave per 64 Gaussians in depth order: entries per tile row ->
//                   tableC[wave][row]
//     P2 rowscan    exclusive prefix of every row's column of that table (two small
//                   kernels), row starts, per-row chunk starts, the total count
//     P3 emit       the same walk again; the wave sorts its (Gaussian, row) items by row in
//                   LDS and writes each row's entries -- (tile column, Gaussian id) -- as
//                   ONE contiguous run at row_start + prefix, lanes owning consecutive
//                   addresses.  Within a row the stream stays in depth order.
//   level B, by tile COLUMN inside each row (<= 1024 tiles: counters and tables are tiny):
//     P4 colhist    per 4096-entry chunk of a row's segment: histogram over the columns
//     P5 colscan    per (row, column): prefix down the row's chunks, tile totals
//     P6 bases      prefix over all tiles -> tile_bins (tile_scatter.hip's kernel)
//     P7 colscatter per chunk: entries are ranked by column in stream order inside LDS
//                   and leave as runs of ~17-34 ids per tile.
//
// HBM traffic per list entry: 8 B written by P3, 4 + 8 B read and 4 B written by P4 / P7
// (24 B) against 8 + 4 + 8 + 4x8 (amplified) before.  The lists are the same, bit for bit,
// as the reference pipeline's minus the dead pairs (tests/test_gpu_kernels.py,
// tests/test_gpu_fullsize.py).
#include "gsr_common.h"
#include "raster_common.h"
#include "tile_rows.h"

// tile_scatter.hip
int gsr_tile_bases(int num_tiles, unsigned *totals, int *tile_bins, hipStream_t s);

namespace gsr_p2 {

constexpr int kGroup = 256;    // waves per scan group
constexpr int kItems = 256;    // (Gaussian, row) items sorted per batch in P3
constexpr int kMaskBlocks = 128;  // batches of up to 32768 entries use the start masks (larger: binary search)
constexpr int kChunk = 4096;   // entries per level-B chunk
constexpr int kMaxDim = 1024;  // tile rows / columns supported

struct Dims {
  int n, waves, groups, tiles_x, tiles_y, txp;  // txp: padded tiles_x (row stride of tableB)
};

// the wave's 64 Gaussians (depth order) and the prefix of their box heights: shared by P1 / P3
struct WaveItems {
  int pref[64];  // inclusive prefix of the box heights (rows) over the lanes
  int gid[64];
  SplatRec rec[64];
  RowParams par[64];
};

__device__ __forceinline__ int load_wave(WaveItems &W, const int lane, const int i, const int n,
                                         const int *__restrict__ order, const SplatRec *__restrict__ recs) {
  SplatRec rec{0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 0u, 0u};
  int g = 0;
  if (i < n) {
    g = order[i];
    rec = recs[g];
  }
  int incl = (int)(rec.box1 >> 16);
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  W.pref[lane] = incl;
  W.gid[lane] = g;
  W.rec[lane] = rec;
  W.par[lane] = make_row_params(rec);
  return __shfl(incl, 63);
}

// item q of the wave -> owner lane k, tile row, tile range [t0, t1)
__device__ __forceinline__ void item_of(const WaveItems &W, const int q, const int total, int &k, int &ty, int &t0,
                                        int &t1) {
  k = 0;
#pragma unroll
  for (int step = 32; step > 0; step >>= 1)
    if (W.pref[k + step - 1] <= q) k += step;
  k = k < 63 ? k : 63;
  const SplatRec r = W.rec[k];
  ty = (int)(r.box0 >> 16) + q - (k ? W.pref[k - 1] : 0);
  t0 = t1 = 0;
  if (q < total) row_range(r, W.par[k], ty, t0, t1);
}

// ---- P1: entries per (wave, tile row) ----------------------------------------------------
__global__ __launch_bounds__(64) void rowcount_kernel(const Dims D, const int *__restrict__ order,
                                                      const SplatRec *__restrict__ recs, int *__restrict__ tableC) {
  __shared__ WaveItems W;
  extern __shared__ int rowcnt[];  // [tiles_y]
  const int lane = threadIdx.x, wave = blockIdx.x;
  const int total = load_wave(W, lane, wave * 64 + lane, D.n, order, recs);
  for (int r = lane; r < D.tiles_y; r += 64) rowcnt[r] = 0;
  __syncthreads();
  for (int q0 = 0; q0 < total; q0 += 64) {
    int k, ty, t0, t1;
    item_of(W, q0 + lane, total, k, ty, t0, t1);
    if (t1 > t0) atomicAdd(&rowcnt[ty], t1 - t0);
  }
  __syncthreads();
  int *out = tableC + (size_t)wave * D.tiles_y;
  for (int r = lane; r < D.tiles_y; r += 64) out[r] = rowcnt[r];
}

// ---- P2a: per (group of 1024 waves, block of 64 rows): exclusive prefix down the waves -----
// Lane l of every wave owns row r0 + l (coalesced 256-byte reads along a table row); the four
// waves of the workgroup take a quarter of the group's waves each.
__global__ __launch_bounds__(256) void rowscan1_kernel(const Dims D, const int *__restrict__ tableC,
                                                       int *__restrict__ tableA, int *__restrict__ gtot) {
  __shared__ int part[4][64];
  const int grp = blockIdx.x, row = blockIdx.y * 64 + (threadIdx.x & 63), q = threadIdx.x >> 6;
  const int w_beg = grp * kGroup + q * (kGroup / 4);
  const int w_end = min(w_beg + kGroup / 4, D.waves);
  const bool live = row < D.tiles_y;
  int sum = 0;
  if (live)
    for (int w = w_beg; w < w_end; ++w) sum += tableC[(size_t)w * D.tiles_y + row];
  part[q][threadIdx.x & 63] = sum;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < q; ++k) run += part[k][threadIdx.x & 63];
  if (live) {
    for (int w = w_beg; w < w_end; ++w) {
      const size_t at = (size_t)w * D.tiles_y + row;
      const int v = tableC[at];
      tableA[at] = run;
      run += v;
    }
    if (q == 3) gtot[(size_t)row * D.groups + grp] = run;  // run = the group's total for this row
  }
}

// ---- P2b: one workgroup: group bases per row, row starts, chunk starts, the total ---------
__global__ __launch_bounds__(1024) void rowscan2_kernel(const Dims D, const int capacity, int *__restrict__ gtot,
                                                        int *__restrict__ row_start, int *__restrict__ row_chunk_start,
                                                        int *__restrict__ count_out) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  // thread r: exclusive prefix over the groups of row r (in place) and the row's total
  int rt = 0;
  if (tid < D.tiles_y) {
    int *g = gtot + (size_t)tid * D.groups;
    for (int k = 0; k < D.groups; ++k) {
      const int v = g[k];
      g[k] = rt;
      rt += v;
    }
  }
  // exclusive scans over the rows: entries and 4096-entry chunks (cut at the capacity)
  auto block_excl = [&](int v, int &total) {
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = 0, all = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < w) base += wsum[k];
      all += wsum[k];
    }
    total = all;
    return base + incl - v;
  };
  int total = 0;
  const int start = block_excl(rt, total);
  if (tid <= D.tiles_y) row_start[tid] = tid < D.tiles_y ? start : total;
  if (tid == 0 && count_out) *count_out = total;
  const int s_cut = start < capacity ? start : capacity;
  const int e_cut = (start + rt) < capacity ? (start + rt) : capacity;
  const int chunks = tid < D.tiles_y ? (e_cut - s_cut + kChunk - 1) / kChunk : 0;
  int ctotal = 0;
  const int cstart = block_excl(chunks, ctotal);
  if (tid <= D.tiles_y) row_chunk_start[tid] = tid < D.tiles_y ? cstart : ctotal;
}

// ---- P3: emission, row-partitioned --------------------------------------------------------
// One wave; its (Gaussian, row) items are taken kItems at a time (in (Gaussian, row) order),
// sorted stably by row in LDS, and every row's entries of the batch are written as one run.
__global__ __launch_bounds__(64) void emit_kernel(const Dims D, const int capacity, const int *__restrict__ order,
                                                  const SplatRec *__restrict__ recs, const int *__restrict__ tableA,
                                                  const int *__restrict__ gbase, const int *__restrict__ row_start,
                                                  unsigned short *__restrict__ tx_out, int *__restrict__ gid_out) {
  __shared__ WaveItems W;
  __shared__ unsigned it_a[kItems];      // row | t0 << 16
  __shared__ unsigned it_b[kItems];      // count | owner lane << 16
  __shared__ unsigned short sorted[kItems];
  __shared__ int sout[kItems + 1];       // entry offset (within the batch) of sorted item s
  // start masks: bit e & 63 of word e >> 6 is set where a sorted item's first entry sits; the item
  // that owns entry e is (#starts at or before e) - 1: two popcounts instead of a binary search
  __shared__ unsigned long long smask[kMaskBlocks];
  __shared__ int blkfirst[kMaskBlocks];  // starts before the block
  extern __shared__ int rows_lds[];      // 4 arrays of tiles_y ints
  int *bcnt = rows_lds, *bcur = bcnt + D.tiles_y, *bent = bcur + D.tiles_y, *gcur = bent + D.tiles_y;
  const int lane = threadIdx.x, wave = blockIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  const int total = load_wave(W, lane, wave * 64 + lane, D.n, order, recs);
  const int grp = wave / kGroup;
  for (int r = lane; r < D.tiles_y; r += 64)
    gcur[r] = row_start[r] + gbase[(size_t)r * D.groups + grp] + tableA[(size_t)wave * D.tiles_y + r];
  __syncthreads();
  int row_bits = 1;
  while ((1 << row_bits) < D.tiles_y) ++row_bits;

  for (int qa = 0; qa < total; qa += kItems) {
    const int nb = total - qa < kItems ? total - qa : kItems;
    for (int r = lane; r < D.tiles_y; r += 64) bcnt[r] = 0, bent[r] = 0;
    __syncthreads();
    // 1. the batch's items, in item order
    for (int i0 = 0; i0 < nb; i0 += 64) {
      const int idx = i0 + lane;
      int k = 0, ty = 0, t0 = 0, t1 = 0;
      if (idx < nb) item_of(W, qa + idx, total, k, ty, t0, t1);
      const int cnt = t1 - t0;
      if (idx < nb) {
        it_a[idx] = (unsigned)ty | ((unsigned)t0 << 16);
        it_b[idx] = (unsigned)cnt | ((unsigned)k << 16);
        if (cnt > 0) {
          atomicAdd(&bcnt[ty], 1);
          atomicAdd(&bent[ty], cnt);
        }
      }
    }
    __syncthreads();
    // 2. exclusive scans over the rows: items (-> bcur) and entries (-> bent, kept exclusive)
    {
      int carry_i = 0, carry_e = 0;
      for (int r0 = 0; r0 < D.tiles_y; r0 += 64) {
        const int r = r0 + lane;
        const int ci = r < D.tiles_y ? bcnt[r] : 0, ce = r < D.tiles_y ? bent[r] : 0;
        int ii = ci, ie = ce;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int a = __shfl_up(ii, o), b = __shfl_up(ie, o);
          if (lane >= o) ii += a, ie += b;
        }
        if (r < D.tiles_y) {
          bcur[r] = carry_i + ii - ci;
          bent[r] = carry_e + ie - ce;  // exclusive: first entry of row r within the batch
        }
        carry_i += __shfl(ii, 63);
        carry_e += __shfl(ie, 63);
      }
    }
    __syncthreads();
    // 3. stable placement by row: slot = cursor[row]++ in item order; lanes of one step that
    //    share a row are ranked by lane (= item order) through wave-wide key matching
    for (int i0 = 0; i0 < nb; i0 += 64) {
      const int idx = i0 + lane;
      const bool live = idx < nb && (it_b[idx < nb ? idx : 0] & 0xffffu) != 0;
      const unsigned row = live ? (it_a[idx] & 0xffffu) : 0u;
      unsigned long long peers = __ballot(live);
      for (int bit = 0; bit < row_bits; ++bit) {
        const bool b = (row >> bit) & 1u;
        const unsigned long long set = __ballot(b);
        peers &= b ? set : ~set;
      }
      if (live) {
        const int rank = __popcll(peers & lt);
        sorted[bcur[row] + rank] = (unsigned short)idx;
      }
      __syncthreads();
      if (live && (peers & lt) == 0) bcur[row] += __popcll(peers);  // the group's first lane advances the cursor
      __syncthreads();
    }
    // 4. entry offsets of the sorted items (items with no entries were not placed)
    int nplaced = 0;
    {
      int carry = 0;
      // number of placed items = bcur of the last row after placement = sum of bcnt
      for (int r0 = 0; r0 < D.tiles_y; r0 += 64) {
        const int r = r0 + lane;
        int c = r < D.tiles_y ? bcnt[r] : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o);
        nplaced += c;
      }
      for (int s0 = 0; s0 < nplaced; s0 += 64) {
        const int s = s0 + lane;
        const int c = s < nplaced ? (int)(it_b[sorted[s]] & 0xffffu) : 0;
        int ic = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int a = __shfl_up(ic, o);
          if (lane >= o) ic += a;
        }
        if (s < nplaced) sout[s] = carry + ic - c;
        carry += __shfl(ic, 63);
      }
      if (lane == 0) sout[nplaced] = carry;
    }
    __syncthreads();
    const int E = sout[nplaced];
    // 5. output-driven, coalesced: entry j of the batch belongs to sorted item s(j)
    const int nblk = (E + 63) >> 6;
    const bool use_masks = nblk <= kMaskBlocks;
    if (use_masks) {
      for (int b = lane; b < nblk; b += 64) smask[b] = 0ull;
      __syncthreads();
      for (int s0 = 0; s0 < nplaced; s0 += 64) {
        const int s = s0 + lane;
        if (s < nplaced) atomicOr(&smask[sout[s] >> 6], 1ull << (sout[s] & 63));
      }
      __syncthreads();
      int carry = 0;
      for (int b0 = 0; b0 < nblk; b0 += 64) {
        const int b = b0 + lane;
        const int c = b < nblk ? __popcll(smask[b]) : 0;
        int ic = c;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
          const int a = __shfl_up(ic, o);
          if (lane >= o) ic += a;
        }
        if (b < nblk) blkfirst[b] = carry + ic - c;
        carry += __shfl(ic, 63);
      }
      __syncthreads();
    }
    for (int j0 = 0; j0 < E; j0 += 64) {
      const int j = j0 + lane;
      if (j < E) {
        int lo;
        if (use_masks) {
          const unsigned long long m = smask[j0 >> 6];
          lo = blkfirst[j0 >> 6] + __popcll(m & ((2ull << lane) - 1ull)) - 1;
        } else {
          lo = 0;
          int hi = nplaced;  // last s with sout[s] <= j
          while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (sout[mid] <= j) lo = mid;
            else hi = mid;
          }
        }
        const unsigned a = it_a[sorted[lo]], b = it_b[sorted[lo]];
        const int row = (int)(a & 0xffffu), t0 = (int)(a >> 16), k = (int)(b >> 16);
        const long long pos = (long long)gcur[row] + (j - bent[row]);
        if (pos < capacity) {
          tx_out[pos] = (unsigned short)(t0 + (j - sout[lo]));
          gid_out[pos] = W.gid[k];
        }
      }
    }
    __syncthreads();
    // 6. advance the rows' global cursors by what the batch wrote (bent is exclusive: the next
    //    row's start minus this row's start; recompute the counts from the items)
    for (int s0 = 0; s0 < nplaced; s0 += 64) {
      const int s = s0 + lane;
      if (s < nplaced) {
        const unsigned a = it_a[sorted[s]], b = it_b[sorted[s]];
        atomicAdd(&gcur[a & 0xffffu], (int)(b & 0xffffu));
      }
    }
    __syncthreads();
  }
}

// chunk c of the row-partitioned stream -> its row and its slice [beg, end)
__device__ __forceinline__ bool chunk_slice(const Dims &D, const int capacity, const int c,
                                            const int *__restrict__ row_start,
                                            const int *__restrict__ row_chunk_start, int &row, int &beg, int &end) {
  if (c >= row_chunk_start[D.tiles_y]) return false;
  int lo = 0, hi = D.tiles_y;  // last row with row_chunk_start[row] <= c
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (row_chunk_start[mid] <= c) lo = mid;
    else hi = mid;
  }
  row = lo;
  const int rs = min(row_start[row], capacity), re = min(row_start[row + 1], capacity);
  beg = rs + (c - row_chunk_start[row]) * kChunk;
  end = min(beg + kChunk, re);
  return beg < end;
}

// ---- P4: per chunk, histogram over the tile columns ---------------------------------------
__global__ __launch_bounds__(256) void colhist_kernel(const Dims D, const int capacity,
                                                      const int *__restrict__ row_start,
                                                      const int *__restrict__ row_chunk_start,
                                                      const unsigned short *__restrict__ tx_in,
                                                      int *__restrict__ tableB) {
  __shared__ int h[kMaxDim];
  int row, beg, end;
  if (!chunk_slice(D, capacity, blockIdx.x, row_start, row_chunk_start, row, beg, end)) return;
  for (int t = threadIdx.x; t < D.txp; t += 256) h[t] = 0;
  __syncthreads();
  for (int e = beg + threadIdx.x; e < end; e += 256) atomicAdd(&h[tx_in[e]], 1);
  __syncthreads();
  int *out = tableB + (size_t)blockIdx.x * D.txp;
  for (int t = threadIdx.x; t < D.txp; t += 256) out[t] = h[t];
}

// ---- P5: per (row, column): exclusive prefix down the row's chunks; tile totals -----------
__global__ __launch_bounds__(256) void colscan_kernel(const Dims D, const int *__restrict__ row_chunk_start,
                                                      int *__restrict__ tableB, unsigned *__restrict__ tile_cnt) {
  const int tx = blockIdx.x * 256 + threadIdx.x, row = blockIdx.y;
  if (tx >= D.tiles_x) return;
  const int c0 = row_chunk_start[row], c1 = row_chunk_start[row + 1];
  int run = 0;
  for (int c = c0; c < c1; ++c) {
    int *p = tableB + (size_t)c * D.txp + tx;
    const int v = *p;
    *p = run;
    run += v;
  }
  tile_cnt[(size_t)row * D.tiles_x + tx] = (unsigned)run;
}

// ---- P7: per chunk, rank by column in LDS (stream order kept), write runs -----------------
// Every wave ranks its own quarter of the chunk (16 rounds of 64 entries): the lanes of a
// round that share a column are found by wave-wide key matching and ranked by lane, a
// wave-private counter per column carries the rank across rounds (no atomics, no barrier
// inside the loop); one barrier later the four waves' counts are prefixed per column and
// every entry knows its slot.  CMAX: column capacity of the LDS tables (256 or 1024).
template <int CMAX>
__global__ __launch_bounds__(256) void colscatter_kernel(const Dims D, const int capacity,
                                                         const int *__restrict__ row_start,
                                                         const int *__restrict__ row_chunk_start,
                                                         const unsigned short *__restrict__ tx_in,
                                                         const int *__restrict__ gid_in,
                                                         const int *__restrict__ tableB,
                                                         const unsigned *__restrict__ tile_base,
                                                         int *__restrict__ ids_out) {
  constexpr int kRounds = kChunk / 256;     // 16 entries per lane
  __shared__ unsigned short s_stx[kChunk];  // columns in ranked order
  __shared__ int s_gid[kChunk];             // ids in ranked order
  __shared__ int loff[CMAX];                // first ranked slot of a column
  __shared__ int wcnt[4][CMAX];             // per-wave column counts, then per-wave bases
  __shared__ int wsum[4];
  int row, beg, end;
  if (!chunk_slice(D, capacity, blockIdx.x, row_start, row_chunk_start, row, beg, end)) return;
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, cnt = end - beg;
  const unsigned long long lt = (1ull << lane) - 1ull;
  for (int t = tid; t < 4 * CMAX; t += 256) (&wcnt[0][0])[t] = 0;
  __syncthreads();
  int key_bits = 1;
  while ((1 << key_bits) < D.tiles_x) ++key_bits;
  int key[kRounds], gid[kRounds], rank[kRounds];
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const int e = w * (kChunk / 4) + r * 64 + lane;
    const bool live = e < cnt;
    key[r] = live ? (int)tx_in[beg + e] : -1;
    gid[r] = live ? gid_in[beg + e] : 0;
  }
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    const bool live = key[r] >= 0;
    const unsigned k = live ? (unsigned)key[r] : 0u;
    unsigned long long peers = __ballot(live);
    for (int bit = 0; bit < key_bits; ++bit) {
      const bool b = (k >> bit) & 1u;
      const unsigned long long set = __ballot(b);
      peers &= b ? set : ~set;
    }
    const int below = __popcll(peers & lt);
    const int prev = live ? wcnt[w][k] : 0;  // every peer reads before the group's first lane writes
    rank[r] = prev + below;
    if (live && below == 0) wcnt[w][k] = prev + __popcll(peers);
  }
  __syncthreads();
  // per column: the chunk's count, its exclusive prefix (loff), and each wave's base
  {
    constexpr int kPer = CMAX / 256;
    int v[kPer], s = 0;
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int c = tid * kPer + j;
      v[j] = wcnt[0][c] + wcnt[1][c] + wcnt[2][c] + wcnt[3][c];
      s += v[j];
    }
    int incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int base = incl - s;
    for (int k = 0; k < w; ++k) base += wsum[k];
#pragma unroll
    for (int j = 0; j < kPer; ++j) {
      const int c = tid * kPer + j;
      loff[c] = base;
      int run = base;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int n = wcnt[q][c];
        wcnt[q][c] = run;
        run += n;
      }
      base += v[j];
    }
  }
  __syncthreads();
#pragma unroll
  for (int r = 0; r < kRounds; ++r) {
    if (key[r] >= 0) {
      const int pos = wcnt[w][key[r]] + rank[r];
      s_gid[pos] = gid[r];
      s_stx[pos] = (unsigned short)key[r];
    }
  }
  __syncthreads();
  const int *pre = tableB + (size_t)blockIdx.x * D.txp;
  const unsigned *tb = tile_base + (size_t)row * D.tiles_x;
  for (int j = tid; j < cnt; j += 256) {
    const int t = s_stx[j];
    ids_out[tb[t] + (unsigned)pre[t] + (unsigned)(j - loff[t])] = s_gid[j];
  }
}

inline size_t align_up(size_t v) { return (v + 255) & ~(size_t)255; }

struct Layout {
  size_t tableC, tableA, gtot, row_start, row_chunk_start, tx, gid, tableB, tile_cnt, total;
  int chunk_slots;
};
inline Layout make_layout(const Dims &D, int capacity) {
  Layout L{};
  size_t off = 0;
  auto take = [&](size_t bytes) {
    const size_t o = off;
    off += align_up(bytes);
    return o;
  };
  L.tableC = take(4 * (size_t)D.waves * D.tiles_y);
  L.tableA = take(4 * (size_t)D.waves * D.tiles_y);
  L.gtot = take(4 * (size_t)D.tiles_y * D.groups);
  L.row_start = take(4 * (size_t)(D.tiles_y + 1));
  L.row_chunk_start = take(4 * (size_t)(D.tiles_y + 1));
  L.tx = take(2 * (size_t)capacity);
  L.gid = take(4 * (size_t)capacity);
  L.chunk_slots = capacity / kChunk + D.tiles_y + 1;
  L.tableB = take(4 * (size_t)L.chunk_slots * D.txp);
  L.tile_cnt = take(4 * (size_t)D.tiles_x * D.tiles_y);
  L.total = off;
  return L;
}
inline Dims make_dims(int n, int tiles_x, int tiles_y) {
  Dims D;
  D.n = n;
  D.waves = (n + 63) / 64;
  D.groups = (D.waves + kGroup - 1) / kGroup;
  D.tiles_x = tiles_x;
  D.tiles_y = tiles_y;
  D.txp = (tiles_x + 3) & ~3;
  return D;
}

}  // namespace gsr_p2

bool gsr_tile_partition2_supported(int tiles_x, int tiles_y) {
  return tiles_x >= 1 && tiles_y >= 1 && tiles_x <= gsr_p2::kMaxDim && tiles_y < gsr_p2::kMaxDim;
}

size_t gsr_tile_partition2_workspace_bytes(int n, int capacity, int tiles_x, int tiles_y) {
  using namespace gsr_p2;
  if (n <= 0 || capacity <= 0) return 0;
  return make_layout(make_dims(n, tiles_x, tiles_y), capacity).total;
}

// order[n]: Gaussians by depth; recs[n]: the records of gsr_count_reach (index order).
// -> ids_sorted (cut at `capacity` entries), tile_bins[tiles][2], count_out (nullable,
// device-accessible): the uncut number of list entries.
int gsr_tile_partition2(int n, int capacity, const int *order, const void *recs, int tiles_x, int tiles_y,
                        int *ids_sorted, int *tile_bins, int *count_out, void *workspace, size_t workspace_bytes,
                        hipStream_t s) {
  using namespace gsr_p2;
  if (!gsr_tile_partition2_supported(tiles_x, tiles_y)) {
    gsr_set_error("tile_partition2: a %d x %d tile grid is not supported", tiles_x, tiles_y);
    return GSR_EINVAL;
  }
  const Dims D = make_dims(n, tiles_x, tiles_y);
  const Layout L = make_layout(D, capacity);
  if (workspace_bytes < L.total) {
    gsr_set_error("tile_partition2: workspace %zu < %zu bytes", workspace_bytes, L.total);
    return GSR_ENOMEM;
  }
  char *ws = static_cast<char *>(workspace);
  int *tableC = reinterpret_cast<int *>(ws + L.tableC), *tableA = reinterpret_cast<int *>(ws + L.tableA);
  int *gtot = reinterpret_cast<int *>(ws + L.gtot), *row_start = reinterpret_cast<int *>(ws + L.row_start);
  int *row_chunk_start = reinterpret_cast<int *>(ws + L.row_chunk_start);
  unsigned short *txs = reinterpret_cast<unsigned short *>(ws + L.tx);
  int *gids = reinterpret_cast<int *>(ws + L.gid), *tableB = reinterpret_cast<int *>(ws + L.tableB);
  unsigned *tile_cnt = reinterpret_cast<unsigned *>(ws + L.tile_cnt);
  const SplatRec *R = static_cast<const SplatRec *>(recs);
  hipLaunchKernelGGL(rowcount_kernel, dim3(D.waves), dim3(64), 4 * (size_t)tiles_y, s, D, order, R, tableC);
  hipLaunchKernelGGL(rowscan1_kernel, dim3(D.groups, gsr_cdiv(tiles_y, 64)), dim3(256), 0, s, D, (const int *)tableC,
                     tableA, gtot);
  hipLaunchKernelGGL(rowscan2_kernel, dim3(1), dim3(1024), 0, s, D, capacity, gtot, row_start, row_chunk_start,
                     count_out);
  hipLaunchKernelGGL(emit_kernel, dim3(D.waves), dim3(64), 16 * (size_t)tiles_y, s, D, capacity, order, R,
                     (const int *)tableA, (const int *)gtot, (const int *)row_start, txs, gids);
  hipLaunchKernelGGL(colhist_kernel, dim3(L.chunk_slots), dim3(256), 0, s, D, capacity, (const int *)row_start,
                     (const int *)row_chunk_start, (const unsigned short *)txs, tableB);
  hipLaunchKernelGGL(colscan_kernel, dim3(gsr_cdiv(tiles_x, 256), tiles_y), dim3(256), 0, s, D,
                     (const int *)row_chunk_start, tableB, tile_cnt);
  int rc = gsr_tile_bases(tiles_x * tiles_y, tile_cnt, tile_bins, s);
  if (rc != GSR_OK) return rc;
  if (tiles_x <= 256)
    hipLaunchKernelGGL(colscatter_kernel<256>, dim3(L.chunk_slots), dim3(256), 0, s, D, capacity,
                       (const int *)row_start, (const int *)row_chunk_start, (const unsigned short *)txs,
                       (const int *)gids, (const int *)tableB, (const unsigned *)tile_cnt, ids_sorted);
  else
    hipLaunchKernelGGL(colscatter_kernel<1024>, dim3(L.chunk_slots), dim3(256), 0, s, D, capacity,
                       (const int *)row_start, (const int *)row_chunk_start, (const unsigned short *)txs,
                       (const int *)gids, (const int *)tableB, (const unsigned *)tile_cnt, ids_sorted);
  GSR_CHECK_LAUNCH("tile_partition2");
  return GSR_OK;
}
