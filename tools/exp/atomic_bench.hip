// atomic_bench.hip -- round 3 feasibility: can "atomic tile cursors + per-tile LDS sort" replace the
// depth sort (12 launches, 100 us) and the stable two-level partition (7 launches, 128 us)?
//   A: I' returnless atomic adds into T tile counters           (count pass)
//   B: I' returned atomic adds + one 8-byte store each          (emit pass)
//   C: one workgroup per tile radix-sorts its (key, id) segment in LDS and writes the ids
// hipcc --offload-arch=gfx950 -O3 -o atomic_bench atomic_bench.hip && ./atomic_bench [N] [tiles_x] [tiles_y] [box]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <random>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void count_kernel(int n, const int2 *centre, int box, int tx, int *counts) {
  int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  int2 c = centre[g];
  for (int y = 0; y < box; ++y)
    for (int x = 0; x < box; ++x) atomicAdd(&counts[(c.y + y) * tx + c.x + x], 1);
}

__global__ __launch_bounds__(256) void emit_kernel(int n, const int2 *centre, const unsigned *keys, int box, int tx,
                                                   int *cursor, uint2 *pairs) {
  int g = blockIdx.x * 256 + threadIdx.x;
  if (g >= n) return;
  int2 c = centre[g];
  unsigned k = keys[g];
  for (int y = 0; y < box; ++y)
    for (int x = 0; x < box; ++x) {
      int slot = atomicAdd(&cursor[(c.y + y) * tx + c.x + x], 1);
      pairs[slot] = make_uint2(k, (unsigned)g);
    }
}

// LSD radix sort of one tile's segment, 4-bit digits, per-thread contiguous blocks (stable)
constexpr int kCap = 2048;
__global__ __launch_bounds__(256) void tile_sort_kernel(const int2 *bins, const uint2 *pairs, int *ids_out) {
  __shared__ unsigned kA[kCap], vA[kCap], kB[kCap], vB[kCap];
  __shared__ unsigned cnt[16 * 256];
  __shared__ unsigned wsum[4];
  const int2 r = bins[blockIdx.x];
  const int L = r.y - r.x, t = threadIdx.x;
  if (L <= 0 || L > kCap) return;
  for (int i = t; i < L; i += 256) { uint2 p = pairs[r.x + i]; kA[i] = p.x; vA[i] = p.y; }
  const int ipt = (L + 255) / 256, i0 = min(t * ipt, L), i1 = min(i0 + ipt, L);
  unsigned *ks = kA, *vs = vA, *kd = kB, *vd = vB;
  __syncthreads();
  for (int shift = 0; shift < 32; shift += 4) {
    unsigned c[16];
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = 0;
    for (int i = i0; i < i1; ++i) {
      const unsigned d = (ks[i] >> shift) & 15u;
#pragma unroll
      for (int b = 0; b < 16; ++b) c[b] += (d == (unsigned)b);
    }
#pragma unroll
    for (int b = 0; b < 16; ++b) cnt[b * 256 + t] = c[b];
    __syncthreads();
    // exclusive scan over the 4096 counters in (bin, thread) order: thread t scans 16 consecutive ones
    unsigned loc[16], s = 0;
#pragma unroll
    for (int j = 0; j < 16; ++j) { loc[j] = s; s += cnt[t * 16 + j]; }
    unsigned incl = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned u = __shfl_up(incl, o); if ((t & 63) >= o) incl += u; }
    if ((t & 63) == 63) wsum[t >> 6] = incl;
    __syncthreads();
    unsigned base = incl - s;
    for (int w = 0; w < (t >> 6); ++w) base += wsum[w];
#pragma unroll
    for (int j = 0; j < 16; ++j) cnt[t * 16 + j] = base + loc[j];
    __syncthreads();
#pragma unroll
    for (int b = 0; b < 16; ++b) c[b] = cnt[b * 256 + t];
    for (int i = i0; i < i1; ++i) {
      const unsigned k = ks[i], d = (k >> shift) & 15u;
      unsigned dst = 0;
#pragma unroll
      for (int b = 0; b < 16; ++b) if (d == (unsigned)b) { dst = c[b]; c[b]++; }
      kd[dst] = k; vd[dst] = vs[i];
    }
    __syncthreads();
    unsigned *tk = ks; ks = kd; kd = tk;
    unsigned *tv = vs; vs = vd; vd = tv;
  }
  for (int i = t; i < L; i += 256) ids_out[r.x + i] = (int)vs[i];
}

int main(int argc, char **argv) {
  const int n = argc > 1 ? atoi(argv[1]) : 1000000, tx = argc > 2 ? atoi(argv[2]) : 120, ty = argc > 3 ? atoi(argv[3]) : 68;
  const int box = argc > 4 ? atoi(argv[4]) : 2;
  const int T = tx * ty;
  std::mt19937 rng(1);
  std::vector<int2> centre(n);
  std::vector<unsigned> keys(n);
  for (int i = 0; i < n; ++i) {
    centre[i] = make_int2(rng() % (tx - box + 1), rng() % (ty - box + 1));
    float d = 3.f + 5.f * (rng() / 4294967296.f);
    keys[i] = *reinterpret_cast<unsigned *>(&d);
  }
  const long long I = (long long)n * box * box;
  int2 *d_c; unsigned *d_k; int *d_cnt, *d_cur, *d_ids; uint2 *d_pairs; int2 *d_bins;
  CK(hipMalloc(&d_c, n * sizeof(int2))); CK(hipMalloc(&d_k, n * 4)); CK(hipMalloc(&d_cnt, T * 4)); CK(hipMalloc(&d_cur, T * 4));
  CK(hipMalloc(&d_ids, I * 4)); CK(hipMalloc(&d_pairs, I * 8)); CK(hipMalloc(&d_bins, T * sizeof(int2)));
  CK(hipMemcpy(d_c, centre.data(), n * sizeof(int2), hipMemcpyHostToDevice));
  CK(hipMemcpy(d_k, keys.data(), n * 4, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int reps = 20;
  float ms;
  std::vector<int> cnt(T), start(T + 1, 0);
  // A
  float tA = 0;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipMemsetAsync(d_cnt, 0, T * 4));
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(count_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_c, box, tx, d_cnt);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) tA += ms;
  }
  CK(hipMemcpy(cnt.data(), d_cnt, T * 4, hipMemcpyDeviceToHost));
  for (int i = 0; i < T; ++i) start[i + 1] = start[i] + cnt[i];
  std::vector<int2> bins(T);
  int maxL = 0;
  for (int i = 0; i < T; ++i) { bins[i] = make_int2(start[i], start[i + 1]); maxL = std::max(maxL, cnt[i]); }
  CK(hipMemcpy(d_bins, bins.data(), T * sizeof(int2), hipMemcpyHostToDevice));
  printf("N %d tiles %dx%d box %d -> I %lld, mean list %.1f, max %d\n", n, tx, ty, box, I, (double)I / T, maxL);
  printf("A count  (returnless atomics): %.1f us\n", 1e3 * tA / reps);
  // B
  float tB = 0;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipMemcpyAsync(d_cur, start.data(), T * 4, hipMemcpyHostToDevice));
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(emit_kernel, dim3((n + 255) / 256), dim3(256), 0, 0, n, d_c, d_k, box, tx, d_cur, d_pairs);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) tB += ms;
  }
  printf("B emit   (returned atomics + 8-byte store): %.1f us\n", 1e3 * tB / reps);
  // C
  float tC = 0;
  for (int r = 0; r < reps + 2; ++r) {
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(tile_sort_kernel, dim3(T), dim3(256), 0, 0, d_bins, d_pairs, d_ids);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    if (r >= 2) tC += ms;
  }
  printf("C sort   (LDS radix, 8 x 4-bit, cap %d): %.1f us\n", kCap, 1e3 * tC / reps);
  // check C
  std::vector<int> ids(I);
  std::vector<uint2> pairs(I);
  CK(hipMemcpy(ids.data(), d_ids, I * 4, hipMemcpyDeviceToHost));
  CK(hipMemcpy(pairs.data(), d_pairs, I * 8, hipMemcpyDeviceToHost));
  long long bad = 0, skipped = 0;
  for (int t = 0; t < T; ++t) {
    if (cnt[t] > kCap) { skipped++; continue; }
    std::vector<unsigned> ref;
    for (int i = start[t]; i < start[t + 1]; ++i) ref.push_back(pairs[i].x);
    std::sort(ref.begin(), ref.end());
    for (int i = start[t]; i < start[t + 1]; ++i) bad += keys[ids[i]] != ref[i - start[t]];
  }
  printf("sort check: %lld wrong, %lld tiles over capacity\n", bad, skipped);
  return 0;
}
