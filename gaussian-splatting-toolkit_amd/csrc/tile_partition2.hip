// tile_partition2.hip -- the per-tile depth-sorted lists by a two-level partition whose every
// global store is coalesced, gfx950.  The list builder of every list of at least 1 M entries and
// of every tile grid above 16384 tiles (binning_fast.hip: tile_sort_mode); it counts its entries
// itself, so its callers need neither per-Gaussian counts nor their scan.
//
// Why: the single-pass tile scatter (tile_scatter.hip) writes each list entry as one 4-byte
// store to its own cache line.  The L2 retires such partial-line writes at ~70-85 M per ms
// chip-wide whatever the band size: 1.39 ms for BASELINE config 5's 97.7 M
// entries, with 8x write amplification.  Here no entry is ever written alone:
//
//   level A, by tile ROW, fused into the emission (no intermediate depth-ordered stream):
//     P1 rowcount   one workgroup per SLAB of 256 Gaussians in depth order: entries per tile
//                   row -> tableC[row][slab]
//     P2 rowscan    one workgroup per tile row: exclusive prefix down the row's slabs + the row's
//                   total (one launch; a two-level scan in two launches until round 3)
//     P3 emit       every workgroup scans the row totals itself (tiles_y values) for the rows'
//                   starts -- workgroup 0 also leaves them, the rows' chunk starts and the total
//                   count for the kernels that follow -- then the same walk again; the workgroup sorts its (Gaussian, row) items by row
//                   in LDS (1024 at a time) and writes each row's entries -- (tile column,
//                   Gaussian id) -- as ONE contiguous run at row_start + prefix, threads
//                   owning consecutive addresses.  Within a row the stream stays in depth
//                   order.  (First version: one wave per 64 Gaussians, 256 items at a time --
//                   runs of ~9 entries, i.e. partial-line writes again, and 13 waves per CU.)
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

constexpr int kSlab = 256;     // Gaussians per workgroup of P1 / P3
constexpr int kGroup = 64;     // slabs per scan group
constexpr int kItems = 1024;   // (Gaussian, row) items sorted per batch in P3
constexpr int kMaskBlocks = 256;  // batches of up to 16384 entries use the start masks (larger: binary search)
constexpr int kChunk = 4096;   // entries per level-B chunk
constexpr int kMaxDim = 1024;  // tile rows / columns supported

struct Dims {
  int n, slabs, groups, tiles_x, tiles_y, txp;  // txp: padded tiles_x (row stride of tableB)
};

// the slab's 256 Gaussians (depth order) and the prefix of their box heights: shared by P1 / P3
struct SlabItems {
  int pref[kSlab];  // inclusive prefix of the box heights (rows) over the threads
  int gid[kSlab];
  SplatRec rec[kSlab];
  RowParams par[kSlab];
};

// exclusive prefix of v over the 256 threads of the workgroup (and the total); two barriers
__device__ __forceinline__ int block_excl(const int v, int *__restrict__ wsum, int &total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  int incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const int t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  int base = 0;
  for (int k = 0; k < w; ++k) base += wsum[k];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

__device__ __forceinline__ long long block_excl64(const long long v, long long *__restrict__ wsum, long long &total) {
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  long long incl = v;
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    const long long t = __shfl_up(incl, o);
    if (lane >= o) incl += t;
  }
  __syncthreads();
  if (lane == 63) wsum[w] = incl;
  __syncthreads();
  long long base = 0;
  for (int k = 0; k < w; ++k) base += wsum[k];
  total = wsum[0] + wsum[1] + wsum[2] + wsum[3];
  return base + incl - v;
}

__device__ __forceinline__ int load_slab(SlabItems &W, int *__restrict__ wsum, const int i, const int n,
                                         const int *__restrict__ order, const SplatRec *__restrict__ recs) {
  const int tid = threadIdx.x;
  SplatRec rec{0.f, 0.f, 1.f, 0.f, 1.f, -1.f, 0u, 0u};
  int g = 0;
  if (i < n) {
    g = order[i];
    rec = recs[g];
  }
  const int h = (int)(rec.box1 >> 16);
  int total;
  W.pref[tid] = block_excl(h, wsum, total) + h;
  W.gid[tid] = g;
  W.rec[tid] = rec;
  W.par[tid] = make_row_params(rec);
  __syncthreads();
  return total;
}

// item q of the slab -> owner thread k, tile row, tile range [t0, t1)
__device__ __forceinline__ void item_of(const SlabItems &W, const int q, const int total, int &k, int &ty, int &t0,
                                        int &t1) {
  k = 0;
#pragma unroll
  for (int step = kSlab / 2; step > 0; step >>= 1)
    if (W.pref[k + step - 1] <= q) k += step;
  k = k < kSlab - 1 ? k : kSlab - 1;
  const SplatRec r = W.rec[k];
  ty = (int)(r.box0 >> 16) + q - (k ? W.pref[k - 1] : 0);
  t0 = t1 = 0;
  if (q < total) row_range(r, W.par[k], ty, t0, t1);
}

// ---- P1: entries per (slab, tile row) ----------------------------------------------------
__global__ __launch_bounds__(kSlab) void rowcount_kernel(const Dims D, const int *__restrict__ order,
                                                         const SplatRec *__restrict__ recs, int *__restrict__ tableC) {
  __shared__ SlabItems W;
  __shared__ int wsum[4];
  extern __shared__ int rowcnt[];  // [tiles_y]
  const int tid = threadIdx.x, slab = blockIdx.x;
  for (int r = tid; r < D.tiles_y; r += kSlab) rowcnt[r] = 0;
  const int total = load_slab(W, wsum, slab * kSlab + tid, D.n, order, recs);
  for (int q = tid; q < total; q += kSlab) {
    int k, ty, t0, t1;
    item_of(W, q, total, k, ty, t0, t1);
    if (t1 > t0) atomicAdd(&rowcnt[ty], t1 - t0);
  }
  __syncthreads();
  // [row][slab]: the scan below runs along a row's slabs (adjacent slabs write adjacent words)
  for (int r = tid; r < D.tiles_y; r += kSlab) tableC[(size_t)r * D.slabs + slab] = rowcnt[r];
}

// ---- P2: per tile row, the exclusive prefix of its entries down the slabs + the row's total -----------
// (round 3: one workgroup per row over the transposed table, instead of a two-level scan in two launches; the
// scan over the rows themselves -- tiles_y values -- is redone by every workgroup of `emit` in its prologue)
__global__ __launch_bounds__(1024) void rowscan_kernel(const Dims D, const int *__restrict__ tableC,
                                                       int *__restrict__ tableA, int *__restrict__ rowtot) {
  __shared__ int wsum[16];
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, row = blockIdx.x;
  const int *src = tableC + (size_t)row * D.slabs;
  int *dst = tableA + (size_t)row * D.slabs;
  int carry = 0;
  for (int base = 0; base < D.slabs; base += 1024) {
    const int i = base + tid;
    const int v = i < D.slabs ? src[i] : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int before = carry, all = 0;
    for (int k = 0; k < 16; ++k) {
      if (k < w) before += wsum[k];
      all += wsum[k];
    }
    if (i < D.slabs) dst[i] = before + incl - v;
    carry += all;
    __syncthreads();
  }
  if (tid == 0) rowtot[row] = carry;
}

// Exclusive scan of f(r) over the tile rows by a workgroup of kSlab threads: out[r] (LDS or global, ny + 1 values
// when `with_total`), returns the total.  wsum: 4 ints of LDS.
template <class F>
__device__ __forceinline__ int scan_rows(const int ny, F f, int *out, const bool with_total, int *wsum) {
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
  int carry = 0;
  for (int base = 0; base < ny; base += kSlab) {
    const int r = base + tid;
    const int v = r < ny ? f(r) : 0;
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    __syncthreads();
    if (lane == 63) wsum[w] = incl;
    __syncthreads();
    int before = carry;
    for (int k = 0; k < w; ++k) before += wsum[k];
    if (r < ny) out[r] = before + incl - v;
    carry += wsum[0] + wsum[1] + wsum[2] + wsum[3];
  }
  __syncthreads();
  if (with_total && tid == 0) out[ny] = carry;
  return carry;
}

// One workgroup per slab; its (Gaussian, row) items are taken kItems at a time (in (Gaussian,
// row) order), sorted stably by row in LDS, and every row's entries of the batch are written as
// one run.  The sort ranks like P7 below: every wave ranks its own contiguous quarter of the
// batch with wave-private row counters, one prefix over (row, wave) later every item has its slot.
__global__ __launch_bounds__(kSlab) void emit_kernel(const Dims D, const int capacity, const int *__restrict__ order,
                                                     const SplatRec *__restrict__ recs, const int *__restrict__ tableA,
                                                     const int *__restrict__ rowtot, int *__restrict__ row_start,
                                                     int *__restrict__ row_chunk_start, int *__restrict__ count_out,
                                                     unsigned short *__restrict__ tx_out, int *__restrict__ gid_out) {
  constexpr int kR = kItems / kSlab;     // items per thread and batch
  __shared__ SlabItems W;
  __shared__ unsigned it_a[kItems];      // row | t0 << 16
  __shared__ unsigned it_b[kItems];      // count | owner thread << 16
  __shared__ unsigned short sorted[kItems];
  __shared__ int sout[kItems + 1];       // entry offset (within the batch) of sorted item s
  // start masks: bit e & 63 of word e >> 6 is set where a sorted item's first entry sits; the item
  // that owns entry e is (#starts at or before e) - 1: two popcounts instead of a binary search
  __shared__ unsigned long long smask[kMaskBlocks];
  __shared__ int blkfirst[kMaskBlocks];  // starts before the block
  __shared__ int wsum[4];
  __shared__ long long wsum64[4];
  extern __shared__ int rows_lds[];      // 7 arrays of tiles_y ints
  const int ny = D.tiles_y;
  int *wcnt = rows_lds;                  // [4][ny] per-wave item counts of a row, then the waves' first slots
  int *bent = wcnt + 4 * ny;             // first entry of the row within the batch
  int *bentc = bent + ny;                // entries of the row in the batch
  int *gcur = bentc + ny;                // the row's global cursor
  const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6, slab = blockIdx.x;
  const unsigned long long lt = (1ull << lane) - 1ull;
  // (this thread's row total and slab prefix are requested before the slab's two dependent gathers, order -> record,
  //  so that the row scan below does not add a round trip of its own; grids of more than kSlab tile rows load later)
  const bool few_rows = ny <= kSlab;
  int my_total = 0, my_prefix = 0;
  if (few_rows && tid < ny) my_total = rowtot[tid], my_prefix = tableA[(size_t)tid * D.slabs + slab];
  const int total = load_slab(W, wsum, slab * kSlab + tid, D.n, order, recs);
  // where every row's segment starts: the exclusive scan of the row totals, redone by every workgroup (tiles_y
  // values) instead of a launch of its own; workgroup 0 leaves it -- and the rows' first 4096-entry chunks, cut at the
  // capacity, and the number of entries -- for the kernels that follow
  const int entries = scan_rows(ny, [&](int r) { return few_rows ? my_total : rowtot[r]; }, gcur, false, wsum);
  if (slab == 0) {
    for (int r = tid; r < ny; r += kSlab) row_start[r] = gcur[r];
    if (tid == 0) {
      row_start[ny] = entries;
      if (count_out) *count_out = entries;
    }
    scan_rows(ny, [&](int r) {
      const int s0 = min(gcur[r], capacity), s1 = min(gcur[r] + rowtot[r], capacity);
      return (s1 - s0 + kChunk - 1) / kChunk;
    }, row_chunk_start, true, wsum);
  }
  for (int r = tid; r < ny; r += kSlab) gcur[r] += few_rows ? my_prefix : tableA[(size_t)r * D.slabs + slab];
  int row_bits = 1;
  while ((1 << row_bits) < ny) ++row_bits;
  const int rp = (ny + kSlab - 1) / kSlab;  // rows per thread in the row scans (contiguous)

  for (int qa = 0; qa < total; qa += kItems) {
    const int nb = total - qa < kItems ? total - qa : kItems;
    for (int r = tid; r < 4 * ny; r += kSlab) wcnt[r] = 0;
    for (int r = tid; r < ny; r += kSlab) bentc[r] = 0;
    __syncthreads();
    // 1. the batch's items; wave w ranks the w-th contiguous quarter of them (rounded up to whole
    //    rounds of 64: a short batch still keeps all four waves busy) by row, in item order
    const int per = ((nb + 255) >> 8) << 6;  // items per wave
    int rank[kR], myrow[kR];
#pragma unroll
    for (int r = 0; r < kR; ++r) {
      myrow[r] = -1;
      rank[r] = 0;
      if (r * 64 >= per) continue;  // (uniform)
      const int idx = w * per + r * 64 + lane;
      int k = 0, ty = 0, t0 = 0, t1 = 0;
      if (idx < nb) item_of(W, qa + idx, total, k, ty, t0, t1);
      const int cnt = t1 - t0;
      if (idx < nb) {
        it_a[idx] = (unsigned)ty | ((unsigned)t0 << 16);
        it_b[idx] = (unsigned)cnt | ((unsigned)k << 16);
      }
      const bool live = idx < nb && cnt > 0;
      unsigned long long peers = __ballot(live);
      for (int bit = 0; bit < row_bits; ++bit) {
        const bool b = (ty >> bit) & 1;
        const unsigned long long set = __ballot(b);
        peers &= b ? set : ~set;
      }
      const int below = __popcll(peers & lt);
      const int prev = live ? wcnt[w * ny + ty] : 0;  // every peer reads before the group's first lane writes
      rank[r] = prev + below;
      if (live && below == 0) wcnt[w * ny + ty] = prev + __popcll(peers);
      if (live) atomicAdd(&bentc[ty], cnt);
      myrow[r] = live ? ty : -1;
    }
    __syncthreads();
    // 2. prefix over (row, wave) of the item counts -> first slots; over the rows of the entries
    int nplaced, E;
    {
      int ci = 0, ce = 0;
      for (int j = 0; j < rp; ++j) {
        const int r = tid * rp + j;
        if (r < ny) {
          ci += wcnt[r] + wcnt[ny + r] + wcnt[2 * ny + r] + wcnt[3 * ny + r];
          ce += bentc[r];
        }
      }
      // one scan for both: items in the low half (<= kItems per batch: no carry), entries in the high half
      long long tot;
      const long long ex = block_excl64(((long long)ce << 32) | (long long)ci, wsum64, tot);
      int run_i = (int)(ex & 0xffffffffll), run_e = (int)(ex >> 32);
      nplaced = (int)(tot & 0xffffffffll);
      E = (int)(tot >> 32);
      for (int j = 0; j < rp; ++j) {
        const int r = tid * rp + j;
        if (r < ny) {
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const int c = wcnt[q * ny + r];
            wcnt[q * ny + r] = run_i;
            run_i += c;
          }
          bent[r] = run_e;
          run_e += bentc[r];
        }
      }
    }
    __syncthreads();
    // 3. placement
#pragma unroll
    for (int r = 0; r < kR; ++r)
      if (myrow[r] >= 0) sorted[wcnt[w * ny + myrow[r]] + rank[r]] = (unsigned short)(w * per + r * 64 + lane);
    __syncthreads();
    // 4. entry offsets of the sorted items (thread t: sorted items t kR .. t kR + kR - 1)
    {
      int c[kR], sum = 0;
#pragma unroll
      for (int j = 0; j < kR; ++j) {
        const int sidx = tid * kR + j;
        c[j] = sidx < nplaced ? (int)(it_b[sorted[sidx]] & 0xffffu) : 0;
        sum += c[j];
      }
      int tot;
      int run = block_excl(sum, wsum, tot);
#pragma unroll
      for (int j = 0; j < kR; ++j) {
        const int sidx = tid * kR + j;
        if (sidx < nplaced) sout[sidx] = run;
        run += c[j];
      }
      if (tid == 0) sout[nplaced] = E;
    }
    // 5. output-driven, coalesced: entry j of the batch belongs to sorted item s(j)
    const int nblk = (E + 63) >> 6;
    const bool use_masks = nblk <= kMaskBlocks;
    if (use_masks) {
      if (tid < nblk) smask[tid] = 0ull;
      __syncthreads();
      for (int sidx = tid; sidx < nplaced; sidx += kSlab) atomicOr(&smask[sout[sidx] >> 6], 1ull << (sout[sidx] & 63));
      __syncthreads();
      const int c = tid < nblk ? __popcll(smask[tid]) : 0;
      int tot;
      const int ex = block_excl(c, wsum, tot);
      if (tid < nblk) blkfirst[tid] = ex;
      __syncthreads();
      // four independent entries per thread and step: the LDS look-up chain (mask -> item -> row
      // cursors) is latency, not bandwidth
      for (int j0 = 0; j0 < E; j0 += 4 * kSlab) {
        int jj[4], lo[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          jj[u] = j0 + u * kSlab + tid;
          const int jc = jj[u] < E ? jj[u] : E - 1;
          lo[u] = blkfirst[jc >> 6] + __popcll(smask[jc >> 6] & ((2ull << (jc & 63)) - 1ull)) - 1;
        }
        unsigned a[4], b[4];
        int so[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int si = sorted[lo[u]];
          a[u] = it_a[si];
          b[u] = it_b[si];
          so[u] = sout[lo[u]];
        }
        long long pos[4];
        int gidv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
          const int row = (int)(a[u] & 0xffffu);
          pos[u] = (long long)gcur[row] + (jj[u] - bent[row]);
          gidv[u] = W.gid[b[u] >> 16];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u)
          if (jj[u] < E && pos[u] < capacity) {
            tx_out[pos[u]] = (unsigned short)((int)(a[u] >> 16) + (jj[u] - so[u]));
            gid_out[pos[u]] = gidv[u];
          }
      }
    } else {
      __syncthreads();
      for (int j = tid; j < E; j += kSlab) {
        int lo = 0, hi = nplaced;  // last s with sout[s] <= j
        while (hi - lo > 1) {
          const int mid = (lo + hi) >> 1;
          if (sout[mid] <= j) lo = mid;
          else hi = mid;
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
    // 6. advance the rows' global cursors by what the batch wrote
    for (int r = tid; r < ny; r += kSlab) gcur[r] += bentc[r];
    __syncthreads();
  }
}

// chunk c of the row-partitioned stream -> its row and its slice [beg, end).  (Staging the
// rows' chunk starts in LDS for the search was tried: the extra barrier cost more than the eight
// dependent L2 hits it saved.)
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
  // four entries (8 bytes) per load, from the aligned group that holds `beg` on
  const uint2 *in4 = reinterpret_cast<const uint2 *>(tx_in);
  for (int g = (beg >> 2) + threadIdx.x; g * 4 < end; g += 256) {
    const uint2 v = in4[g];
    const int e = g * 4;
    if (e >= beg && e < end) atomicAdd(&h[v.x & 0xffffu], 1);
    if (e + 1 >= beg && e + 1 < end) atomicAdd(&h[v.x >> 16], 1);
    if (e + 2 >= beg && e + 2 < end) atomicAdd(&h[v.y & 0xffffu], 1);
    if (e + 3 >= beg && e + 3 < end) atomicAdd(&h[v.y >> 16], 1);
  }
  __syncthreads();
  int *out = tableB + (size_t)blockIdx.x * D.txp;
  for (int t = threadIdx.x; t < D.txp; t += 256) out[t] = h[t];
}

// ---- P5: per (row, column): exclusive prefix down the row's chunks; tile totals -----------
// Lane l of every wave owns column tx0 + l (coalesced reads along a table row); the four waves
// of the workgroup take a quarter of the row's chunks each (sum, then prefix: the table is read
// twice, but a row's ~180 chunks at config 5 are walked by four waves and the grid has 4x the
// workgroups of one-thread-per-column).
__global__ __launch_bounds__(256) void colscan_kernel(const Dims D, const int *__restrict__ row_chunk_start,
                                                      int *__restrict__ tableB, unsigned *__restrict__ tile_cnt) {
  __shared__ int part[4][64];
  const int lane = threadIdx.x & 63, q = threadIdx.x >> 6;
  const int tx = blockIdx.x * 64 + lane, row = blockIdx.y;
  const int c0 = row_chunk_start[row], c1 = row_chunk_start[row + 1];
  const int per = (c1 - c0 + 3) >> 2;
  const int cb = min(c0 + q * per, c1), ce = min(cb + per, c1);
  const bool live = tx < D.tiles_x;
  int sum = 0;
  if (live)
    for (int c = cb; c < ce; ++c) sum += tableB[(size_t)c * D.txp + tx];
  part[q][lane] = sum;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < q; ++k) run += part[k][lane];
  if (!live) return;
  for (int c = cb; c < ce; ++c) {
    int *p = tableB + (size_t)c * D.txp + tx;
    const int v = *p;
    *p = run;
    run += v;
  }
  if (q == 3) tile_cnt[(size_t)row * D.tiles_x + tx] = (unsigned)run;
}

// ---- P5 + P6 in one launch for rows of up to 256 tiles: one workgroup of 16 waves per tile row
// (4 column groups of 64 x 4 quarters of the row's chunks); the row's tiles start at the row's
// segment start (the lists are row-major), so the prefix over the row's columns finishes
// tile_bins without the separate launch over all tiles.
__global__ __launch_bounds__(1024) void colscan_row_kernel(const Dims D, const int capacity,
                                                           const int *__restrict__ row_start,
                                                           const int *__restrict__ row_chunk_start,
                                                           int *__restrict__ tableB, unsigned *__restrict__ tile_base,
                                                           int *__restrict__ tile_bins) {
  __shared__ int part[4][256];
  __shared__ int s_tot[256];
  __shared__ int wsum[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6, cg = wv & 3, q = wv >> 2;
  const int tx = cg * 64 + lane, row = blockIdx.x;
  const int c0 = row_chunk_start[row], c1 = row_chunk_start[row + 1];
  const int per = (c1 - c0 + 3) >> 2;
  const int cb = min(c0 + q * per, c1), ce = min(cb + per, c1);
  const bool live = tx < D.tiles_x;
  int sum = 0;
  if (live)
    for (int c = cb; c < ce; ++c) sum += tableB[(size_t)c * D.txp + tx];
  part[q][tx] = sum;
  __syncthreads();
  int run = 0;
  for (int k = 0; k < q; ++k) run += part[k][tx];
  if (live)
    for (int c = cb; c < ce; ++c) {
      int *p = tableB + (size_t)c * D.txp + tx;
      const int v = *p;
      *p = run;
      run += v;
    }
  if (q == 3) s_tot[tx] = live ? run : 0;
  __syncthreads();
  if (tid < 256) {  // exclusive prefix over the row's columns
    const int v = s_tot[tid];
    int incl = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
      const int t = __shfl_up(incl, o);
      if (lane >= o) incl += t;
    }
    if (lane == 63) wsum[wv] = incl;
    s_tot[tid] = incl - v;
  }
  __syncthreads();
  if (tid < D.tiles_x) {
    int before = 0;
    for (int k = 0; k < (tid >> 6); ++k) before += wsum[k];
    const int v = part[0][tid] + part[1][tid] + part[2][tid] + part[3][tid];
    const int base = min(row_start[row], capacity) + before + s_tot[tid];
    const size_t tile = (size_t)row * D.tiles_x + tid;
    tile_base[tile] = (unsigned)base;
    tile_bins[2 * tile] = v ? base : 0;
    tile_bins[2 * tile + 1] = v ? base + v : 0;
  }
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
  // destination of ranked slot j of column t: tile base + the chunk's prefix + (j - loff[t]);
  // the first three are folded into loff (one coalesced load per table instead of two L2
  // look-ups per entry)
  const int *pre = tableB + (size_t)blockIdx.x * D.txp;
  const unsigned *tb = tile_base + (size_t)row * D.tiles_x;
  for (int t = tid; t < D.tiles_x; t += 256) loff[t] = (int)(tb[t] + (unsigned)pre[t] - (unsigned)loff[t]);
  __syncthreads();
  for (int j = tid; j < cnt; j += 256) ids_out[(unsigned)loff[s_stx[j]] + (unsigned)j] = s_gid[j];
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
  L.tableC = take(4 * (size_t)D.slabs * D.tiles_y);
  L.tableA = take(4 * (size_t)D.slabs * D.tiles_y);
  L.gtot = take(4 * (size_t)D.tiles_y * (D.groups > 1 ? D.groups : 1));  // (the row totals: tiles_y ints)
  L.row_start = take(4 * (size_t)(D.tiles_y + 1));
  L.row_chunk_start = take(4 * (size_t)(D.tiles_y + 1));
  L.tx = take(2 * (size_t)capacity + 8);  // P4 reads whole groups of four
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
  D.slabs = (n + kSlab - 1) / kSlab;
  D.groups = (D.slabs + kGroup - 1) / kGroup;
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
  hipLaunchKernelGGL(rowcount_kernel, dim3(D.slabs), dim3(kSlab), 4 * (size_t)tiles_y, s, D, order, R, tableC);
  hipLaunchKernelGGL(rowscan_kernel, dim3(tiles_y), dim3(1024), 0, s, D, (const int *)tableC, tableA, gtot);
  hipLaunchKernelGGL(emit_kernel, dim3(D.slabs), dim3(kSlab), 28 * (size_t)tiles_y, s, D, capacity, order, R,
                     (const int *)tableA, (const int *)gtot, row_start, row_chunk_start, count_out, txs, gids);
  hipLaunchKernelGGL(colhist_kernel, dim3(L.chunk_slots), dim3(256), 0, s, D, capacity, (const int *)row_start,
                     (const int *)row_chunk_start, (const unsigned short *)txs, tableB);
  if (tiles_x <= 256) {
    hipLaunchKernelGGL(colscan_row_kernel, dim3(tiles_y), dim3(1024), 0, s, D, capacity, (const int *)row_start,
                       (const int *)row_chunk_start, tableB, tile_cnt, tile_bins);
  } else {
    hipLaunchKernelGGL(colscan_kernel, dim3(gsr_cdiv(tiles_x, 64), tiles_y), dim3(256), 0, s, D,
                       (const int *)row_chunk_start, tableB, tile_cnt);
    int rc = gsr_tile_bases(tiles_x * tiles_y, tile_cnt, tile_bins, s);
    if (rc != GSR_OK) return rc;
  }
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
