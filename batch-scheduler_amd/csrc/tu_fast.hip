// tu_fast.hip — translation unit of the steady-state chain's second launch: the scan / Filter role kernels of bs_fast.hpp and
// bs_filter_t.hpp (13 scalar-lane instantiations each) and their launch wrappers.  A translation unit of its own so that the library
// builds in parallel and a change to one kernel family recompiles that family only (bs_launch.hpp is the interface; -DBS_UNITY, the
// probe builds, includes this file into bsched.hip instead).
#ifndef BS_UNITY
#define BS_TU_FAST
#endif
#include "bs_fast.hpp"
#include "bs_launch.hpp"

#include <algorithm>

namespace bs {

static inline uint32_t tu_cdiv(uint32_t a, uint32_t b) { return (a + b - 1) / b; }

template <int S>
void launch_fast_bc_s(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                             const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks, uint32_t filter_blocks) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_scan_filter_final<S>), grid, dim3(256), 0, c.stream, pd, gr, nd, b, bt, prm, c.M, nseg, scan_blocks, filter_blocks,
                     c.filter_waves, c.filter_slots_cap, tu_cdiv(c.P, kTblChunk));
}
template <int S>
void launch_fast_b_s(const FastLaunch& c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg,
                            uint32_t scan_blocks) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_scan_filter<S>), grid, dim3(256), 0, c.stream, pd, nd, bt, prm, c.M, nseg, scan_blocks, c.filter_waves,
                     c.filter_slots_cap);
}
void launch_fast_b(const FastLaunch& c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg,
                          uint32_t scan_blocks) {
  switch (c.S) {
    case 0: launch_fast_b_s<0>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 1: launch_fast_b_s<1>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 2: launch_fast_b_s<2>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 3: launch_fast_b_s<3>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 4: launch_fast_b_s<4>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 5: launch_fast_b_s<5>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 6: launch_fast_b_s<6>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 7: launch_fast_b_s<7>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 8: launch_fast_b_s<8>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 9: launch_fast_b_s<9>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 10: launch_fast_b_s<10>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    case 11: launch_fast_b_s<11>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
    default: launch_fast_b_s<12>(c, grid, pd, nd, bt, prm, nseg, scan_blocks); break;
  }
}
template <int S>
void launch_fast_scan_s(const FastLaunch& c, dim3 grid, const BatchDev& bt, const BatchParams& prm, uint32_t nseg) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_scan<S>), grid, dim3(256), 0, c.stream, bt, prm, c.M, nseg);
}
void launch_fast_scan(const FastLaunch& c, dim3 grid, const BatchDev& bt, const BatchParams& prm, uint32_t nseg) {
  switch (c.S) {
    case 0: launch_fast_scan_s<0>(c, grid, bt, prm, nseg); break;
    case 1: launch_fast_scan_s<1>(c, grid, bt, prm, nseg); break;
    case 2: launch_fast_scan_s<2>(c, grid, bt, prm, nseg); break;
    case 3: launch_fast_scan_s<3>(c, grid, bt, prm, nseg); break;
    case 4: launch_fast_scan_s<4>(c, grid, bt, prm, nseg); break;
    case 5: launch_fast_scan_s<5>(c, grid, bt, prm, nseg); break;
    case 6: launch_fast_scan_s<6>(c, grid, bt, prm, nseg); break;
    case 7: launch_fast_scan_s<7>(c, grid, bt, prm, nseg); break;
    case 8: launch_fast_scan_s<8>(c, grid, bt, prm, nseg); break;
    case 9: launch_fast_scan_s<9>(c, grid, bt, prm, nseg); break;
    case 10: launch_fast_scan_s<10>(c, grid, bt, prm, nseg); break;
    case 11: launch_fast_scan_s<11>(c, grid, bt, prm, nseg); break;
    default: launch_fast_scan_s<12>(c, grid, bt, prm, nseg); break;
  }
}
template <int S>
void launch_fast_bt_s(const FastLaunch& c, dim3 grid, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_scan_filter_t<S>), grid, dim3(256), 0, c.stream, nd, bt, prm, c.M, nseg, scan_blocks, c.filter_split ? c.filter_split : c.filter_waves,
                     c.filter_slots_cap, (c.tp_filter == 7u ? 1u : 0u) | (c.filter_split ? 2u : 0u));
}
void launch_fast_bt(const FastLaunch& c, dim3 grid, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks) {
  // S <= 4 only (run_fast sends wider contexts through k_fast_scan + k_fast_filter_t): beyond that the two roles in one kernel run out
  // of SGPRs and the instantiation reserves scratch memory (36 bytes at S = 5, tools/kernel_resources.py), which every launch pays for
  switch (c.S) {
    case 0: launch_fast_bt_s<0>(c, grid, nd, bt, prm, nseg, scan_blocks); break;
    case 1: launch_fast_bt_s<1>(c, grid, nd, bt, prm, nseg, scan_blocks); break;
    case 2: launch_fast_bt_s<2>(c, grid, nd, bt, prm, nseg, scan_blocks); break;
    case 3: launch_fast_bt_s<3>(c, grid, nd, bt, prm, nseg, scan_blocks); break;
    default: launch_fast_bt_s<4>(c, grid, nd, bt, prm, nseg, scan_blocks); break;
  }
}
void launch_fast_filter(const FastLaunch& c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm) {
  switch (c.tp_filter) {
    case 1: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_filter<4, true>), grid, dim3(256), 0, c.stream, pd, nd, bt, prm, c.filter_waves, c.filter_slots_cap); break;
    case 2: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_filter<2, true>), grid, dim3(256), 0, c.stream, pd, nd, bt, prm, c.filter_waves, c.filter_slots_cap); break;
    case 3: hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_filter<2, false>), grid, dim3(256), 0, c.stream, pd, nd, bt, prm, c.filter_waves, c.filter_slots_cap); break;
    case 4: hipLaunchKernelGGL(k_fast_filter_w7, grid, dim3(256), 0, c.stream, pd, nd, bt, prm, c.filter_waves, c.filter_slots_cap); break;
    default: hipLaunchKernelGGL(k_fast_filter_t, grid, dim3(256), 0, c.stream, nd, bt, prm, c.filter_split ? c.filter_split : c.filter_waves, c.filter_slots_cap, c.filter_split ? 1u : 0u); break;
  }
}
// How many blocks of the fused launch the chip holds at once (occupancy API, minus one block per CU: the API can be one high,
// MI355X_MICROARCH.md "Residency").  The fused launch is only taken when its whole grid fits: then no producer block can be
// waiting for a slot that a spinning final block occupies, whatever order the dispatcher hands blocks out in.
template <int S>
static int fused_residency_s(const FastLaunch& c) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fast_scan_filter_final<S>, 256, 0) != hipSuccess || per_cu <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) return 0;
  return std::max(0, per_cu - 1) * prop.multiProcessorCount;
}
int fused_residency_query(const FastLaunch& c) {
  int r = 0;
  switch (c.S) {
    case 0: r = fused_residency_s<0>(c); break;   case 1: r = fused_residency_s<1>(c); break;   case 2: r = fused_residency_s<2>(c); break;
    case 3: r = fused_residency_s<3>(c); break;   case 4: r = fused_residency_s<4>(c); break;   case 5: r = fused_residency_s<5>(c); break;
    case 6: r = fused_residency_s<6>(c); break;   case 7: r = fused_residency_s<7>(c); break;   case 8: r = fused_residency_s<8>(c); break;
    case 9: r = fused_residency_s<9>(c); break;   case 10: r = fused_residency_s<10>(c); break; case 11: r = fused_residency_s<11>(c); break;
    default: r = fused_residency_s<12>(c); break;
  }
  return r;
}
void launch_fast_bc(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                           const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks, uint32_t filter_blocks) {
  switch (c.S) {
    case 0: launch_fast_bc_s<0>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 1: launch_fast_bc_s<1>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 2: launch_fast_bc_s<2>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 3: launch_fast_bc_s<3>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 4: launch_fast_bc_s<4>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 5: launch_fast_bc_s<5>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 6: launch_fast_bc_s<6>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 7: launch_fast_bc_s<7>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 8: launch_fast_bc_s<8>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 9: launch_fast_bc_s<9>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 10: launch_fast_bc_s<10>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    case 11: launch_fast_bc_s<11>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
    default: launch_fast_bc_s<12>(c, grid, pd, gr, nd, b, bt, prm, nseg, scan_blocks, filter_blocks); break;
  }
}


template <int TS, bool WHOLE>
static void launch_fast_step_a_s(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                                 const BatchParams& prm, const TableDesc* forced, uint32_t nchunks, uint32_t query_blocks, uint32_t nshares, uint32_t filter_blocks,
                                 uint32_t tk_pods0, uint32_t tk_tab0, uint32_t param_blocks, const int64_t* ckeys, const uint32_t* cpres, uint32_t kcap,
                                 uint32_t tk_p1, uint32_t tk_done, uint32_t forced_cls) {
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_fast_step_a<TS, WHOLE>), grid, dim3(kTblChunk), 0, c.stream, pd, gr, nd, b, bt, prm, forced, nchunks, query_blocks, nshares, filter_blocks,
                     c.filter_waves, c.filter_slots_cap, tk_pods0, tk_tab0, param_blocks, ckeys, cpres, kcap, tk_p1, tk_done, forced_cls);
}
#define BS_STEP_A_ARGS c, grid, pd, gr, nd, b, bt, prm, forced, nchunks, query_blocks, nshares, filter_blocks, tk_pods0, tk_tab0, param_blocks, ckeys, cpres, kcap, tk_p1, tk_done, forced_cls
void launch_fast_step_a(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                        const BatchParams& prm, const TableDesc* forced, uint32_t nchunks, uint32_t query_blocks, uint32_t nshares, uint32_t filter_blocks,
                        uint32_t tk_pods0, uint32_t tk_tab0, uint32_t param_blocks, const int64_t* ckeys, const uint32_t* cpres, uint32_t kcap,
                        uint32_t whole, uint32_t tk_p1, uint32_t tk_done, uint32_t forced_cls) {
  switch (c.S * 2u + (whole ? 1u : 0u)) {
    case 0: launch_fast_step_a_s<0, false>(BS_STEP_A_ARGS); break;
    case 1: launch_fast_step_a_s<0, true>(BS_STEP_A_ARGS); break;
    case 2: launch_fast_step_a_s<1, false>(BS_STEP_A_ARGS); break;
    case 3: launch_fast_step_a_s<1, true>(BS_STEP_A_ARGS); break;
    case 4: launch_fast_step_a_s<2, false>(BS_STEP_A_ARGS); break;
    case 5: launch_fast_step_a_s<2, true>(BS_STEP_A_ARGS); break;
    case 6: launch_fast_step_a_s<3, false>(BS_STEP_A_ARGS); break;
    case 7: launch_fast_step_a_s<3, true>(BS_STEP_A_ARGS); break;
    case 8: launch_fast_step_a_s<4, false>(BS_STEP_A_ARGS); break;
    default: launch_fast_step_a_s<4, true>(BS_STEP_A_ARGS); break;
  }
}
#undef BS_STEP_A_ARGS
template <int TS, bool WHOLE>
static int step_a_residency_s(const FastLaunch& c) {
  int per_cu = 0;
  if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_fast_step_a<TS, WHOLE>, (int)kTblChunk, 0) != hipSuccess || per_cu <= 0) return 0;
  hipDeviceProp_t prop;
  if (hipGetDeviceProperties(&prop, c.device) != hipSuccess) return 0;
  return std::max(0, per_cu - 1) * prop.multiProcessorCount;      // (minus one block per CU: the API can be one high, see fused_residency_query)
}
int step_a_residency_query(const FastLaunch& c, bool whole) {
  switch (c.S * 2u + (whole ? 1u : 0u)) {
    case 0: return step_a_residency_s<0, false>(c);
    case 1: return step_a_residency_s<0, true>(c);
    case 2: return step_a_residency_s<1, false>(c);
    case 3: return step_a_residency_s<1, true>(c);
    case 4: return step_a_residency_s<2, false>(c);
    case 5: return step_a_residency_s<2, true>(c);
    case 6: return step_a_residency_s<3, false>(c);
    case 7: return step_a_residency_s<3, true>(c);
    case 8: return step_a_residency_s<4, false>(c);
    default: return step_a_residency_s<4, true>(c);
  }
}

}  // namespace bs
