// tu_seq.hip — translation unit of the sequential pass (bs_seq.hpp: k_seq_pass, six instantiations) and its launch wrapper; see
// tu_fast.hip for why.
#ifndef BS_UNITY
#define BS_TU_SEQ
#endif
#include "bs_seq.hpp"
#include "bs_launch.hpp"

namespace bs {

template <int TS>
static void launch_seq_s(hipStream_t stream, size_t lds, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const SeqDev& sq, const SeqParams& prm) {
  // static LDS (first-fit bounds, reduction slots) + the key window can exceed the default 64 KB of dynamic LDS
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&k_seq_pass<TS>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k_seq_pass<TS>), dim3(1), dim3(kSeqBlock), lds, stream, pd, gr, nd, sq, prm);
}

void launch_seq(hipStream_t stream, uint32_t S, size_t lds, const PodsDev& pd, const GroupsDev& gr, const NodesDev& nd, const SeqDev& sq, const SeqParams& prm) {
  switch (S <= 4 ? (int)S : -1) {
    case 0: launch_seq_s<0>(stream, lds, pd, gr, nd, sq, prm); break;
    case 1: launch_seq_s<1>(stream, lds, pd, gr, nd, sq, prm); break;
    case 2: launch_seq_s<2>(stream, lds, pd, gr, nd, sq, prm); break;
    case 3: launch_seq_s<3>(stream, lds, pd, gr, nd, sq, prm); break;
    case 4: launch_seq_s<4>(stream, lds, pd, gr, nd, sq, prm); break;
    default: launch_seq_s<-1>(stream, lds, pd, gr, nd, sq, prm); break;
  }
}

}  // namespace bs
