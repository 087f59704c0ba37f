// bs_sort.hpp — batched queue ordering: the permutation that sorts the pending pods the way the
// reference's queue does through ScheduleOperation.Compare (core.go:368-411, plugged in as Less, batchscheduler.go:214).
//
// Compare as a sort key (ascending):
//   1. priority, DESCENDING                                              (:379-381)
//   2. kind: 0 no PodGroup label  <  1 labelled, group known  <  2 labelled, lister error            (:384-399)
//      (a lister error makes Compare return false in BOTH directions against every pod of the same priority, so Compare
//       is not a strict weak order there; such pods are placed last within their priority — consistent with every
//       Compare(x, y) == true the reference can produce)
//   3. the group's order rank = dense rank of (CreationTimestamp ascending, group NAME descending)   (:400-406)
//      — equal (timestamp, name) pairs must share a rank: the reference compares names, not namespaces
//   4. the pod's queue timestamp, ascending                                                          (:384-386, :407-408)
// Ties keep their input order (stable).
//
// One workgroup, least-significant-digit radix sort over the 17 key bytes; a byte on which every pod agrees (most of
// them: priorities and the high bytes of the timestamps rarely differ) costs one histogram and no scatter.  The stable
// rank inside a 1024-pod tile comes from eight wave ballots (the peers of a lane = lanes with the same byte) plus
// per-wave counts in LDS — no atomics on the scatter path.  Not a hot path: one launch per scheduling cycle at most.
#pragma once

#include "bs_common.hpp"

namespace bs {

constexpr int kSortBlock = 1024;
constexpr int kSortDigits = 17;

struct SortIn {
  uint32_t p, g;
  const int32_t* prio;
  const int32_t* group;
  const int64_t* ts;
  const uint32_t* order_rank;     // [g]
};

__device__ __forceinline__ uint32_t sort_digit(const SortIn& in, uint32_t e, int d) {
  if (d < 8) return (uint32_t)((((unsigned long long)in.ts[e] ^ 0x8000000000000000ull) >> (8 * d)) & 0xFFull);
  if (d < 13) {
    const int32_t gi = in.group[e];
    const uint32_t kind = gi == BS_POD_NOT_GROUPED ? 0u : ((gi >= 0 && (uint32_t)gi < in.g) ? 1u : 2u);
    const unsigned long long w = ((unsigned long long)kind << 32) | (kind == 1u ? in.order_rank[gi] : 0u);
    return (uint32_t)((w >> (8 * (d - 8))) & 0xFFull);
  }
  const uint32_t w = ~((uint32_t)in.prio[e] ^ 0x80000000u);
  return (w >> (8 * (d - 13))) & 0xFFu;
}

__global__ __launch_bounds__(kSortBlock) void k_queue_sort(SortIn in, uint32_t* idx_a, uint32_t* idx_b, uint32_t* perm_out) {
  __shared__ uint32_t s_hist[256], s_base[256], s_wtot[4], s_uniform;
  __shared__ uint32_t s_wcnt[kSortBlock / 64][256];
  const uint32_t P = in.p, t = threadIdx.x;
  const int lane = lane_id(), wave = wave_id();
  for (uint32_t i = t; i < P; i += kSortBlock) idx_a[i] = i;
  uint32_t* src = idx_a;
  uint32_t* dst = idx_b;
  __syncthreads();
  for (int d = 0; d < kSortDigits; ++d) {
    if (t < 256) s_hist[t] = 0;
    if (t == 0) s_uniform = 0;
    __syncthreads();
    for (uint32_t i = t; i < P; i += kSortBlock) atomicAdd(&s_hist[sort_digit(in, src[i], d)], 1u);
    __syncthreads();
    if (t < 256 && s_hist[t] == P) s_uniform = 1;
    __syncthreads();
    const bool uniform = s_uniform != 0;           // read by every wave before anyone can reset it for the next digit
    __syncthreads();
    if (uniform) continue;                         // every pod has the same byte here: the order does not change
    if (t < 256) {                                 // exclusive scan of the 256 counts (4 waves)
      const uint32_t v = s_hist[t];
      uint32_t incl = wave_incl_scan_add<uint32_t>(v);
      s_base[t] = incl - v;
      if (lane == 63) s_wtot[wave] = incl;         // wave totals, in their own words: other waves may still be reading their counts
    }
    __syncthreads();
    if (t < 256) {
      uint32_t off = 0;
      for (int w = 0; w < wave; ++w) off += s_wtot[w];
      s_base[t] += off;
    }
    __syncthreads();
    for (uint32_t base = 0; base < P; base += kSortBlock) {
      for (uint32_t k = t; k < (kSortBlock / 64) * 256u; k += kSortBlock) (&s_wcnt[0][0])[k] = 0;
      const uint32_t i = base + t;
      const bool valid = i < P;
      const uint32_t e = valid ? src[i] : 0u;
      const uint32_t dg = valid ? sort_digit(in, e, d) : 0u;
      unsigned long long peers = __ballot(valid);
#pragma unroll
      for (int bit = 0; bit < 8; ++bit) {
        const unsigned long long b = __ballot((dg >> bit) & 1u);
        peers &= ((dg >> bit) & 1u) ? b : ~b;
      }
      const uint32_t rank = (uint32_t)__popcll(peers & ((1ull << lane) - 1ull));
      __syncthreads();                             // s_wcnt is zero
      if (valid && rank == 0) s_wcnt[wave][dg] = (uint32_t)__popcll(peers);
      __syncthreads();
      if (valid) {
        uint32_t before = 0;
        for (int w = 0; w < wave; ++w) before += s_wcnt[w][dg];
        dst[s_base[dg] + before + rank] = e;
      }
      __syncthreads();
      if (t < 256) {
        uint32_t tot = 0;
#pragma unroll
        for (int w = 0; w < kSortBlock / 64; ++w) tot += s_wcnt[w][t];
        s_base[t] += tot;
      }
      __syncthreads();
    }
    uint32_t* tmp = src; src = dst; dst = tmp;
    __threadfence_block();
    __syncthreads();
  }
  for (uint32_t i = t; i < P; i += kSortBlock) perm_out[i] = src[i];
}

}  // namespace bs
