// bs_launch.hpp — the launch wrappers that live in translation units of their own (tu_fast.hip, tu_seq.hip), so that the library builds
// in parallel and a change to one kernel family recompiles that family only.  Kernels in the shared headers are `inline __global__`:
// each translation unit emits exactly the kernels it launches.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "bs_kernels.hpp"

namespace bs {

// what the wrappers of the steady-state chain's second launch need from the context
struct FastLaunch {
  hipStream_t stream;
  uint32_t S, M, P, filter_waves, filter_slots_cap, tp_filter;
  int device;
  uint32_t filter_split = 0;   // waves the transposed Filter items are CUT for (0 = filter_waves, the grid): a rank of a sharded job cuts finer, see run_fast
};
void launch_fast_bc(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                    const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks, uint32_t filter_blocks);
void launch_fast_b(const FastLaunch& c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg,
                   uint32_t scan_blocks);
void launch_fast_scan(const FastLaunch& c, dim3 grid, const BatchDev& bt, const BatchParams& prm, uint32_t nseg);
void launch_fast_bt(const FastLaunch& c, dim3 grid, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm, uint32_t nseg, uint32_t scan_blocks);
void launch_fast_filter(const FastLaunch& c, dim3 grid, const PodsDev& pd, const NodesDev& nd, const BatchDev& bt, const BatchParams& prm);
int fused_residency_query(const FastLaunch& c);
// round 5: launch A + the scan / Filter roles in one launch (k_fast_step_a<S>, S <= 4 or the generic-lane instantiation)
void launch_fast_step_a(const FastLaunch& c, dim3 grid, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const BatchDev& b, const BatchDev& bt,
                        const BatchParams& prm, const TableDesc* forced, uint32_t nchunks, uint32_t query_blocks, uint32_t nshares, uint32_t filter_blocks,
                        uint32_t tk_pods0, uint32_t tk_tab0, uint32_t param_blocks, const int64_t* ckeys, const uint32_t* cpres, uint32_t kcap,
                        uint32_t whole, uint32_t tk_p1, uint32_t tk_done, uint32_t forced_cls);
int step_a_residency_query(const FastLaunch& c, bool whole);      // blocks of k_fast_scan_filter_final<S> the chip holds at once (0 = unknown)

struct SeqDev;
struct SeqParams;
void launch_seq(hipStream_t stream, uint32_t S, size_t lds, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const SeqDev& sq, const SeqParams& prm);

}  // namespace bs
