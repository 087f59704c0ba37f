// bs_common.hpp — shared device helpers for the gfx950 gang-feasibility kernels.
//
// Wave = 64 lanes everywhere (CDNA4).  All resource arithmetic is int64 with two's-complement
// wrap (Go semantics), done in uint64 to stay defined in C++.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/bsched.h"

// Which translation unit emits the NON-template kernels of the shared headers (a template kernel is emitted where it is instantiated):
// bsched.hip (BS_TU_MAIN) everything but the two Filter kernels tu_fast.hip launches; tu_fast.hip (BS_TU_FAST) those two; tu_seq.hip
// (BS_TU_SEQ) none; a unity build (-DBS_UNITY: none of the three macros) all of them.
#if defined(BS_TU_FAST) || defined(BS_TU_SEQ)
#define BS_EMIT_MAIN 0
#else
#define BS_EMIT_MAIN 1
#endif
#if defined(BS_TU_MAIN) || defined(BS_TU_SEQ)
#define BS_EMIT_FAST 0
#else
#define BS_EMIT_FAST 1
#endif

#define BS_INF 0xFFFFFFFFu

namespace bs {

constexpr int kWave = 64;

// Probe build only (-DBS_PROBE, tools/stamp_probe.py; never the shipped library): thread 0 of the first kProbeBlocks blocks of a
// launch leaves constant-rate clock stamps (s_memrealtime, 100 MHz) at a few points, after draining what it has in flight —
// where the microseconds of a latency-bound launch go (dispatch spread, depth of the dependent-load chains, store drain).
#ifdef BS_PROBE
constexpr int kProbeKernels = 8, kProbeBlocks = 128, kProbeStamps = 8;
__device__ unsigned long long g_probe[kProbeKernels][kProbeBlocks][kProbeStamps];
__device__ __forceinline__ void probe_stamp(int k, int s, uint32_t blk) {
  if (threadIdx.x == 0 && blk < (uint32_t)kProbeBlocks) {
    unsigned long long t;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
    g_probe[k][blk][s] = t;
  }
}
// a COUNT in a stamp slot: shown by tools/stamp_probe.py as `v` microseconds behind the block's entry stamp (100 ticks = 1 us)
__device__ __forceinline__ void probe_count(int k, int s, uint32_t blk, uint32_t v) {
  if (threadIdx.x == 0 && blk < (uint32_t)kProbeBlocks) g_probe[k][blk][s] = g_probe[k][blk][0] + 100ull * v;
}
#define BS_STAMP(k, s) probe_stamp((k), (s), blockIdx.x)
#define BS_COUNT(k, s, v) probe_count((k), (s), blockIdx.x, (v))
#define BS_STAMP_AT(k, s, blk) probe_stamp((k), (s), (blk))
#define BS_COUNT_AT(k, s, blk, v) probe_count((k), (s), (blk), (v))
#else
#define BS_STAMP_AT(k, s, blk) ((void)0)
#define BS_COUNT_AT(k, s, blk, v) ((void)0)
#define BS_STAMP(k, s) ((void)0)
#define BS_COUNT(k, s, v) ((void)0)
#endif
constexpr int kScanBlock = 1024;   // single-block sequential-chunk scans

__device__ __forceinline__ int64_t wadd(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wsub(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int64_t wmul(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }
__device__ __forceinline__ int wave_id() { return (int)(threadIdx.x >> 6); }

// int64(float32(a) * pct) exactly as Go/amd64 computes it (core.go:656-659,667):
// i64 -> f32 round-to-nearest-even (CVTSQ2SS), one f32 multiply (MULSS, no contraction),
// truncation toward zero (CVTTSS2SQ; out of range / NaN -> 0x8000000000000000).
__device__ __forceinline__ int64_t scale_f32(int64_t a, float pct) {
  float f = __ll2float_rn((long long)a);
  float m = __fmul_rn(f, pct);
  if (!(m < 9223372036854775808.0f) || m < -9223372036854775808.0f) return INT64_MIN;
  return (int64_t)m;
}

// Put a wave-uniform value into one (wave-uniform) lane of a VGPR.  gfx9 v_writelane_b32 accepts only
// one SGPR source (constant-bus limit), so this is the select form: v_cmp_eq + v_cndmask.
__device__ __forceinline__ uint32_t writelane_u32(uint32_t value, uint32_t lane, uint32_t old) {
  return ((uint32_t)lane_id() == lane) ? value : old;
}

template <typename T>
__device__ __forceinline__ T wave_incl_scan_add(T v) {
  const int lane = lane_id();
#pragma unroll
  for (int o = 1; o < 64; o <<= 1) {
    T u = __shfl_up(v, o);
    if (lane >= o) v += u;
  }
  return v;
}

// Wave64 inclusive scan (sum) of 64-bit values on the DPP path: row_shr 1/2/4/8 inside the 16-lane rows, then
// row_bcast:15 into rows 1 and 3 and row_bcast:31 into rows 2 and 3 — VALU moves with a lane pattern, no LDS crossbar
// (__shfl_up is ds_bpermute: ~100 cycles per dependent step, and the table build / the scan's chunk prefix do L of them).
// Lanes without a source keep the old operand (0), so they add nothing.  Needs all 64 lanes active.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned long long dpp_shift_u64(unsigned long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)v, CTRL, ROWMASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(v >> 32), CTRL, ROWMASK, 0xF, false);
  return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ unsigned long long wave_incl_scan_add_u64(unsigned long long v) {
  v += dpp_shift_u64<0x111, 0xF>(v);     // row_shr:1
  v += dpp_shift_u64<0x112, 0xF>(v);     // row_shr:2
  v += dpp_shift_u64<0x114, 0xF>(v);     // row_shr:4
  v += dpp_shift_u64<0x118, 0xF>(v);     // row_shr:8
  v += dpp_shift_u64<0x142, 0xA>(v);     // row_bcast:15 -> rows 1, 3
  v += dpp_shift_u64<0x143, 0xC>(v);     // row_bcast:31 -> rows 2, 3
  return v;
}

// Wave64 maximum of signed 64-bit values on the same DPP path; the result is valid in LANE 63 only (the inclusive "scan"
// of max ends there).  Lanes without a source see INT64_MIN.  A __shfl_xor butterfly is 12 ds_bpermute per value — the
// table builders reduce L values per 64-row group in every block, and with every CU full of such blocks the LDS crossbar
// was what the launch waited for.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ long long dpp_shift_i64_min(long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)(uint32_t)(unsigned long long)v, CTRL, ROWMASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0x80000000u, (int)(uint32_t)((unsigned long long)v >> 32), CTRL, ROWMASK, 0xF, false);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_max_i64_lane63(long long v) {
  long long u;
  u = dpp_shift_i64_min<0x111, 0xF>(v); v = u > v ? u : v;     // row_shr:1
  u = dpp_shift_i64_min<0x112, 0xF>(v); v = u > v ? u : v;     // row_shr:2
  u = dpp_shift_i64_min<0x114, 0xF>(v); v = u > v ? u : v;     // row_shr:4
  u = dpp_shift_i64_min<0x118, 0xF>(v); v = u > v ? u : v;     // row_shr:8
  u = dpp_shift_i64_min<0x142, 0xA>(v); v = u > v ? u : v;     // row_bcast:15 -> rows 1, 3
  u = dpp_shift_i64_min<0x143, 0xC>(v); v = u > v ? u : v;     // row_bcast:31 -> rows 2, 3
  return v;
}

// ... and the minimum, broadcast to every lane (two v_readlane of lane 63): the node scan needs the tile's smallest
// request per resource lane before it can look at a single group bound.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ long long dpp_shift_i64_max(long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_update_dpp((int)0xFFFFFFFFu, (int)(uint32_t)(unsigned long long)v, CTRL, ROWMASK, 0xF, false);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_update_dpp((int)0x7FFFFFFFu, (int)(uint32_t)((unsigned long long)v >> 32), CTRL, ROWMASK, 0xF, false);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long readlane63_i64(long long v) {
  const uint32_t lo = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(unsigned long long)v, 63);
  const uint32_t hi = (uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)((unsigned long long)v >> 32), 63);
  return (long long)(((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ long long wave_min_i64_all(long long v) {
  long long u;
  u = dpp_shift_i64_max<0x111, 0xF>(v); v = u < v ? u : v;
  u = dpp_shift_i64_max<0x112, 0xF>(v); v = u < v ? u : v;
  u = dpp_shift_i64_max<0x114, 0xF>(v); v = u < v ? u : v;
  u = dpp_shift_i64_max<0x118, 0xF>(v); v = u < v ? u : v;
  u = dpp_shift_i64_max<0x142, 0xA>(v); v = u < v ? u : v;
  u = dpp_shift_i64_max<0x143, 0xC>(v); v = u < v ? u : v;
  return readlane63_i64(v);
}

__device__ __forceinline__ uint32_t wave_min_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = min(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, (uint32_t)__shfl_xor((int)v, o));
  return v;
}
__device__ __forceinline__ int wave_max_i32(int v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = max(v, __shfl_xor(v, o));
  return v;
}

// Block-wide reductions for up to 1024 threads; `lds` needs 16 uint32.  All threads get the result.
__device__ __forceinline__ uint32_t block_min_u32(uint32_t v, uint32_t* lds) {
  v = wave_min_u32(v);
  __syncthreads();
  if (lane_id() == 0) lds[wave_id()] = v;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  uint32_t r = BS_INF;
  for (int w = 0; w < nw; ++w) r = min(r, lds[w]);
  return r;
}
__device__ __forceinline__ uint32_t block_max_u32(uint32_t v, uint32_t* lds) {
  v = wave_max_u32(v);
  __syncthreads();
  if (lane_id() == 0) lds[wave_id()] = v;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6);
  uint32_t r = 0;
  for (int w = 0; w < nw; ++w) r = max(r, lds[w]);
  return r;
}

// Block inclusive scan (sum) for T in {uint32_t, unsigned long long}; `lds` needs 16 T.
// Returns the inclusive value; `total` = block total.
template <typename T>
__device__ __forceinline__ T block_incl_scan_add(T v, T* lds, T& total) {
  T s = wave_incl_scan_add<T>(v);
  __syncthreads();
  if (lane_id() == 63) lds[wave_id()] = s;
  __syncthreads();
  const int nw = (int)(blockDim.x >> 6), w = wave_id();
  T off = 0, tot = 0;
  for (int i = 0; i < nw; ++i) {
    T x = lds[i];
    if (i < w) off += x;
    tot += x;
  }
  total = tot;
  return s + off;
}

// One atomicAdd per distinct key per wave: lanes with `active` contribute 1 to counters[key];
// returns this lane's slot (old counter value + rank among same-key lanes).  The reduction uses
// readfirstlane + ballot + mbcnt — no LDS.
__device__ __forceinline__ uint32_t wave_aggregated_inc(uint32_t* counters, uint32_t key, bool active) {
  uint32_t slot = 0;
  unsigned long long todo = __ballot(active);
  while (todo) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
    const unsigned long long same = __ballot(active && key == k0) & todo;
    uint32_t base = 0;
    if (lane_id() == leader) base = atomicAdd(&counters[k0], (uint32_t)__popcll(same));
    base = (uint32_t)__shfl((int)base, leader);
    if (active && key == k0) {
      const unsigned long long below = same & ((1ull << lane_id()) - 1ull);
      slot = base + (uint32_t)__popcll(below);
    }
    todo &= ~same;
  }
  return slot;
}

// One lane per distinct key among the active lanes of the wave (the first one): for work every holder of a key would
// repeat identically (filling a shared slot).  A queue is mostly gang-sorted, so a wave of pods holds a handful of keys;
// the tail a resident queue grows (pods appended in arrival order) holds up to 64.  The loop costs one ballot + one
// v_readlane per distinct key and gives up after kElectRounds keys: the lanes still waiting then all count as elected —
// their keys are (nearly) all different anyway, and what they write is identical for equal keys.
constexpr int kElectRounds = 8;
__device__ __forceinline__ bool wave_elect_by_key(uint32_t key, bool active) {
  bool elected = false;
  unsigned long long todo = __ballot(active);
  for (int round = 0; todo && round < kElectRounds; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    if (lane_id() == leader) elected = true;
    todo &= ~__ballot(active && key == k0);
  }
  if (todo & (1ull << lane_id())) elected = true;
  return elected;
}

// Same aggregation when nobody needs the old counter value: the adds are fire-and-forget, so the loop over the distinct
// keys of a wave does not wait for an atomic round trip per key (it did: ~15 distinct groups per wave of pods).  After
// kElectRounds keys the remaining lanes add for themselves (a wave of all-different keys is 64 independent atomics either way).
__device__ __forceinline__ void wave_aggregated_add(uint32_t* counters, uint32_t key, bool active) {
  unsigned long long todo = __ballot(active);
  for (int round = 0; todo && round < kElectRounds; ++round) {
    const int leader = __ffsll((long long)todo) - 1;
    const uint32_t k0 = (uint32_t)__builtin_amdgcn_readlane((int)key, leader);
    const unsigned long long same = __ballot(active && key == k0) & todo;
    if (lane_id() == leader) (void)__hip_atomic_fetch_add(&counters[k0], (uint32_t)__popcll(same), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    todo &= ~same;
  }
  if (todo & (1ull << lane_id())) (void)__hip_atomic_fetch_add(&counters[key], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace bs
