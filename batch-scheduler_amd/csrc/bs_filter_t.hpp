// bs_filter_t.hpp — the TRANSPOSED Filter item (computeResourceSatisfied, core.go:514-564), for the throughput regime.
//
// filter_item (bs_kernels.hpp) is written for the latency regime: lanes are NODES while it compares, a slot's request is the
// wave-uniform operand (broadcast from LDS), the compare's EXEC mask is dropped into the slot's lane with two v_writelane, and
// what counts is the length of one item's load chain.  With thousands of distinct requests in a batch (bench.py
// scenarios.all_distinct_requests; 10^8 .. 10^9 pod x node pairs really evaluated) every resident wave has compares to do, and
// what counts is VALU instructions per pair and waves per SIMD.  Here lanes are REQUEST SLOTS for the whole item: a lane keeps
// its request R (4 x int64) in VGPRs, and the NODE is the wave-uniform operand — a node's `left` (getLeftResource, core.go:436-475)
// comes through the scalar cache (one s_load_dwordx8 = four nodes of one resource lane; the waves of a node run read the same
// few KB) and sits in SGPRs:
//     k x v_cmp_ge_i64 mask, left[j] (SGPR), R[j] (VGPR)  [masks of the lanes ANDed on the scalar unit] ; v_addc_co_u32 word, word, word, mask
// mask = the slots this node can hold (case 2, core.go:551-555); the add-with-carry shifts the lane's word and drops the lane's
// mask bit into it: (k + 1) VALU per node and 64 slots — filter_item: (k + 2) VALU + 4 wait states per slot and 64
// nodes —, no LDS, no v_writelane, no EXEC write, no transposition at the end (a lane already owns its slot's word), and a
// register footprint that lets eight waves share a SIMD.
// The pod-independent masks of a node block (in range, evaluable, case 3 for the tile's common leader: core.go:558-563) still
// want lanes = NODES: one vector load of the block + ballots, as in filter_item; a tile whose slots name different leaders (it
// straddles the two leader halves of the slot array) gets case 3 from a second pass of the same chain against the lane's own M.
// Scalar loads run up to the end of the last 64-node block: left4 carries 64 entries of padding behind its fourth lane
// (upload_nodes), a block's tail beyond n reads the next lane's row or the padding, and those nodes are masked by `okmask`.
// Results are bit-identical to filter_item's (tests/test_gpu_throughput.py runs every form against the oracle).
#pragma once

#include "bs_kernels.hpp"

namespace bs {

typedef const __attribute__((address_space(4))) int64_t* cnode_t;

// four consecutive nodes (a[u], bq[u], cq[u], d[u] = node u's left on resource lane 0..3) against the lanes' requests R; w = the word
// half the nodes belong to, shifted left by four with the nodes' verdicts in its low bits, FIRST node highest (the caller reverses a
// finished half: v_bfrev_b32).
// Per node: one v_cmp_ge_i64 per compared lane into an SGPR pair (the first straight into the node's mask, the others through VCC and
// s_and_b64 on the scalar unit), then ONE v_addc_co_u32 w, w, w, mask — w = 2 w + (this lane's mask bit): the shift and the insertion
// of the lane's verdict in one VALU instruction, no EXEC write anywhere.  Rounds 3-5 ran the compares as an EXEC chain
// (s_mov exec · k x v_cmpx · v_or under that EXEC · s_lshl): the same k + 1 VALU instructions, but every one of them waits for the
// EXEC the previous one wrote, and at the three waves per SIMD the one-launch form runs at (the scan role's 137 VGPRs) nothing hides
// that chain — 21 cycles per node and SIMD against the 8-10 the sequence needs (tools/ubench/node_loop).  Here the four nodes of a
// group are independent until their addc, and the masks are read three instructions after they were written (a VALU-written SGPR
// read as a carry needs two wait states on gfx950; the compiler puts s_nop 1 there, this sequence needs none).
// Lanes that are not to be evaluated get verdicts too; the callers drop them (c2pod / the (myff & 2) test).
template <int MASK>
__device__ __forceinline__ void filter_node4(const int64_t (&R)[4], const int64_t (&a)[4], const int64_t (&bq)[4], const int64_t (&cq)[4],
                                             const int64_t (&d)[4], uint32_t& w);
#define BS_FN4_NODE(g0, g1, g2, g3, h0, h1, h2, h3, u)                                                                   \
  BS_OPT(g0, "v_cmp_ge_i64_e64 %[m" #u "], %[a" #u "], %[R0]\n\t") BS_OPT(g1, "v_cmp_ge_i64_e64 %[m" #u "], %[b" #u "], %[R1]\n\t") \
  BS_OPT(g2, "v_cmp_ge_i64_e64 %[m" #u "], %[c" #u "], %[R2]\n\t") BS_OPT(g3, "v_cmp_ge_i64_e64 %[m" #u "], %[d" #u "], %[R3]\n\t") \
  BS_OPT(h1, "v_cmp_ge_i64_e32 vcc, %[b" #u "], %[R1]\n\ts_and_b64 %[m" #u "], %[m" #u "], vcc\n\t")                      \
  BS_OPT(h2, "v_cmp_ge_i64_e32 vcc, %[c" #u "], %[R2]\n\ts_and_b64 %[m" #u "], %[m" #u "], vcc\n\t")                      \
  BS_OPT(h3, "v_cmp_ge_i64_e32 vcc, %[d" #u "], %[R3]\n\ts_and_b64 %[m" #u "], %[m" #u "], vcc\n\t")
#define BS_FN4_ADDC(u) "v_addc_co_u32_e64 %[w], %[m" #u "], %[w], %[w], %[m" #u "]\n\t"
#define BS_DEF_FILTER_NODE4(MASK, g0, g1, g2, g3, h0, h1, h2, h3)                                                        \
  template <>                                                                                                            \
  __device__ __forceinline__ void filter_node4<MASK>(const int64_t (&R)[4], const int64_t (&a)[4], const int64_t (&bq)[4], \
                                                     const int64_t (&cq)[4], const int64_t (&d)[4], uint32_t& w) {         \
    unsigned long long m0, m1, m2, m3;                                                                                   \
    asm volatile(BS_FN4_NODE(g0, g1, g2, g3, h0, h1, h2, h3, 0) BS_FN4_NODE(g0, g1, g2, g3, h0, h1, h2, h3, 1)             \
                 BS_FN4_NODE(g0, g1, g2, g3, h0, h1, h2, h3, 2) BS_FN4_NODE(g0, g1, g2, g3, h0, h1, h2, h3, 3)             \
                 BS_FN4_ADDC(0) BS_FN4_ADDC(1) BS_FN4_ADDC(2) BS_FN4_ADDC(3)                                             \
                 : [w] "+v"(w), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)                           \
                 : [R0] "v"(R[0]), [R1] "v"(R[1]), [R2] "v"(R[2]), [R3] "v"(R[3]), [a0] "s"(a[0]),                       \
                   [a1] "s"(a[1]), [a2] "s"(a[2]), [a3] "s"(a[3]), [b0] "s"(bq[0]), [b1] "s"(bq[1]), [b2] "s"(bq[2]),    \
                   [b3] "s"(bq[3]), [c0] "s"(cq[0]), [c1] "s"(cq[1]), [c2] "s"(cq[2]), [c3] "s"(cq[3]), [d0] "s"(d[0]),  \
                   [d1] "s"(d[1]), [d2] "s"(d[2]), [d3] "s"(d[3])                                                        \
                 : "vcc", "scc");                                                                                        \
  }
BS_DEF_FILTER_NODE4(1, 1, 0, 0, 0, 0, 0, 0, 0)
BS_DEF_FILTER_NODE4(2, 0, 1, 0, 0, 0, 0, 0, 0)
BS_DEF_FILTER_NODE4(3, 1, 0, 0, 0, 0, 1, 0, 0)
BS_DEF_FILTER_NODE4(4, 0, 0, 1, 0, 0, 0, 0, 0)
BS_DEF_FILTER_NODE4(5, 1, 0, 0, 0, 0, 0, 1, 0)
BS_DEF_FILTER_NODE4(6, 0, 1, 0, 0, 0, 0, 1, 0)
BS_DEF_FILTER_NODE4(7, 1, 0, 0, 0, 0, 1, 1, 0)
BS_DEF_FILTER_NODE4(8, 0, 0, 0, 1, 0, 0, 0, 0)
BS_DEF_FILTER_NODE4(9, 1, 0, 0, 0, 0, 0, 0, 1)
BS_DEF_FILTER_NODE4(10, 0, 1, 0, 0, 0, 0, 0, 1)
BS_DEF_FILTER_NODE4(11, 1, 0, 0, 0, 0, 1, 0, 1)
BS_DEF_FILTER_NODE4(12, 0, 0, 1, 0, 0, 0, 0, 1)
BS_DEF_FILTER_NODE4(13, 1, 0, 0, 0, 0, 0, 1, 1)
BS_DEF_FILTER_NODE4(14, 0, 1, 0, 0, 0, 0, 1, 1)
BS_DEF_FILTER_NODE4(15, 1, 0, 0, 0, 0, 1, 1, 1)

// one 64-node block (first node n0, a multiple of 64) against the lanes in `em`: per lane the 64 bits "left >= R on every
// compared resource lane".  Bits of lanes outside `em` stay 0.
// The nodes travel in GROUPS (eight with one compared lane: one s_load_dwordx16; four otherwise: one s_load_dwordx8 per lane),
// double-buffered in SGPRs: the scalar loads of the NEXT group are issued before the compares of the current one (whose values are
// already there: `filter_sgprs_ready4` makes the compiler wait for them BEFORE it issues the next loads — scalar loads return out of
// order, the only wait there is waits for everything outstanding, so a wait behind the issue would wait for both).
// Group size, measured (cfg4 all-distinct, one lane binding).  Round 4, compares as an EXEC chain: 16 / 8 nodes per group 203 us per
// step against 192 with four (profiles/r04c_tp_sweep4_groups16.jsonl) — at 56 cycles per node a group of four covered the load.
// Round 5, v_cmp + v_addc (8 cycles per node): eight nodes 161.7 us against 169.2 with four, sixteen 172.3 (the SGPR file: 2 x 32 node
// registers + 8 of masks); cfg3 32.9 / 34.0 / 33.2 (profiles/r05_tp_groups.txt; digests equal).
__device__ __forceinline__ void filter_sgprs_ready4(const int64_t& a, const int64_t& b, const int64_t& c, const int64_t& d) {
  asm volatile("" ::"s"(a), "s"(b), "s"(c), "s"(d));
}
#ifndef BS_FT_G1
#define BS_FT_G1 8                 // nodes per group with ONE compared lane (s_load_dwordx16), BS_FT_G2 with two; four beyond that
#endif
#ifndef BS_FT_G2
#define BS_FT_G2 4
#endif
template <int MASK>
__device__ __forceinline__ void filter_block_t(cnode_t L4, uint32_t stride, uint32_t n0, const int64_t (&R)[4], uint32_t (&wd)[2]) {
  constexpr int K = ((MASK >> 0) & 1) + ((MASK >> 1) & 1) + ((MASK >> 2) & 1) + ((MASK >> 3) & 1);
  static_assert(K >= 1 && K <= 4, "MASK names the compared lanes");
  constexpr int GN = K == 1 ? BS_FT_G1 : (K == 2 ? BS_FT_G2 : 4);    // nodes per group
  constexpr int NG = 64 / GN;
  constexpr int J0 = (MASK & 1) ? 0 : ((MASK & 2) ? 1 : ((MASK & 4) ? 2 : 3));      // a compared lane: stands in for the lanes that are not
  auto load = [&](uint32_t nn, int64_t (&s)[4][GN]) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < GN; ++u)
        if ((MASK >> j) & 1) s[j][u] = L4[(size_t)j * stride + nn + (uint32_t)u];
  };
  int64_t cur[4][GN], nxt[4][GN];
  load(n0, cur);
  uint32_t w = 0;
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < GN; c += 4)
        if ((MASK >> j) & 1) filter_sgprs_ready4(cur[j][c], cur[j][c + 1], cur[j][c + 2], cur[j][c + 3]);
    if (g + 1 < NG) load(n0 + (uint32_t)(g + 1) * GN, nxt);
#pragma unroll
    for (int c = 0; c < GN; c += 4) {
      if (g * GN + c == 32) { wd[0] = __builtin_bitreverse32(w); w = 0; }
      int64_t t[4][4];
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int u = 0; u < 4; ++u) t[j][u] = cur[((MASK >> j) & 1) ? j : J0][c + u];
      filter_node4<MASK>(R, t[0], t[1], t[2], t[3], w);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < GN; ++u)
        if ((MASK >> j) & 1) cur[j][u] = nxt[j][u];
  }
  wd[1] = __builtin_bitreverse32(w);
}
__device__ __forceinline__ void filter_block_any(uint32_t lane_mask, cnode_t L4, uint32_t stride, uint32_t n0, const int64_t (&R)[4],
                                                 uint32_t (&wd)[2]) {
  switch (lane_mask) {
    case 1: filter_block_t<1>(L4, stride, n0, R, wd); break;
    case 2: filter_block_t<2>(L4, stride, n0, R, wd); break;
    case 3: filter_block_t<3>(L4, stride, n0, R, wd); break;
    case 4: filter_block_t<4>(L4, stride, n0, R, wd); break;
    case 5: filter_block_t<5>(L4, stride, n0, R, wd); break;
    case 6: filter_block_t<6>(L4, stride, n0, R, wd); break;
    case 7: filter_block_t<7>(L4, stride, n0, R, wd); break;
    case 8: filter_block_t<8>(L4, stride, n0, R, wd); break;
    case 9: filter_block_t<9>(L4, stride, n0, R, wd); break;
    case 10: filter_block_t<10>(L4, stride, n0, R, wd); break;
    case 11: filter_block_t<11>(L4, stride, n0, R, wd); break;
    case 12: filter_block_t<12>(L4, stride, n0, R, wd); break;
    case 13: filter_block_t<13>(L4, stride, n0, R, wd); break;
    case 14: filter_block_t<14>(L4, stride, n0, R, wd); break;
    default: filter_block_t<15>(L4, stride, n0, R, wd); break;
  }
}

// ------------------------------------------------------------------------------------------------
// Round 6: the scalar-lean block loop.
//
// What bounds the transposed item is not VALU issue (round 5's reading) but the CU's ONE scalar unit, which its four SIMDs share: the counters
// of the cfg4 all-distinct launch (profiles/r05_distinct4_pmc_summary.txt) have SQ_ACTIVE_INST_SCA = 23.7 M quad-cycles = 370 k cycles per CU
// — the whole 364 k-cycle launch —, 19.5 M SALU + 4.1 M SMEM instructions for 15.06 M (node, wave) pairs, and every further compared lane adds
// one s_and_b64 per pair and ~80 us per launch (profiles/r06_nodew_ab_v1.jsonl: k = 1 / 2 / 4 lanes 150 / 227 / 390 us) = 4 cycles of the CU's
// scalar unit per SALU instruction.  Removing VALU instructions (the node words alone) therefore changed nothing.  This loop spends scalar
// instructions only where there is no other way:
//   * per resource lane its OWN accumulator word: k x (v_cmp_ge_i64 -> SGPR mask, v_addc_co_u32 acc_j, acc_j, acc_j, mask) per node, the k words
//     ANDed once per 32 nodes on the VALU — no s_and_b64 per node and lane (2 k VALU per node instead of k + 1, 0 SALU instead of k - 1);
//   * the switch over the compared lanes OUTSIDE the block loop (one loop per lane mask), EXEC set once for the run (the evaluated slots);
//   * the node-only words of a block as ONE s_load_dwordx4 from the batch's node words, complement already taken;
//   * nothing else per block than the loop counter and the row pointer (VALU).
// Bits of lanes outside EXEC are never written (the callers' rows of unevaluated slots are written by a loop of their own).
// ------------------------------------------------------------------------------------------------
typedef const __attribute__((address_space(4))) unsigned long long* cword_t;

// four consecutive nodes of ONE resource lane against the lanes' request r: acc = (acc << 4) | verdicts, first node highest
__device__ __forceinline__ void cmp4_acc(const int64_t& r, const int64_t& a0, const int64_t& a1, const int64_t& a2, const int64_t& a3, uint32_t& acc) {
  unsigned long long m0, m1, m2, m3;
  asm volatile("v_cmp_ge_i64_e64 %[m0], %[a0], %[r]\n\tv_cmp_ge_i64_e64 %[m1], %[a1], %[r]\n\tv_cmp_ge_i64_e64 %[m2], %[a2], %[r]\n\t"
               "v_cmp_ge_i64_e64 %[m3], %[a3], %[r]\n\t"
               "v_addc_co_u32_e64 %[w], %[m0], %[w], %[w], %[m0]\n\tv_addc_co_u32_e64 %[w], %[m1], %[w], %[w], %[m1]\n\t"
               "v_addc_co_u32_e64 %[w], %[m2], %[w], %[w], %[m2]\n\tv_addc_co_u32_e64 %[w], %[m3], %[w], %[w], %[m3]\n\t"
               : [w] "+v"(acc), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
               : [r] "v"(r), [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3));
}
#ifndef BS_FL_G1
#define BS_FL_G1 16                // nodes per group in the lean loop with ONE compared lane (2 x s_load_dwordx16; cur + nxt = 64 SGPRs)
#endif
#ifndef BS_FL_G2
#define BS_FL_G2 8                 // ... with two compared lanes (2 x s_load_dwordx16, cur + nxt = 64 SGPRs); four nodes per group beyond that
#endif
template <int MASK>
struct LeanGroup {
  static constexpr int K = ((MASK >> 0) & 1) + ((MASK >> 1) & 1) + ((MASK >> 2) & 1) + ((MASK >> 3) & 1);
  static constexpr int GN = MASK <= 0 ? 4 : (K == 1 ? BS_FL_G1 : (K == 2 ? BS_FL_G2 : 4));    // nodes per group
};
template <int MASK, int GN>
__device__ __forceinline__ void lean_load(cnode_t L4, uint32_t stride, uint32_t nn, int64_t (&s)[4][GN]) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int u = 0; u < GN; ++u)
      if ((MASK >> j) & 1) s[j][u] = L4[(size_t)j * stride + nn + (uint32_t)u];
}
// one 64-node block against the lanes in EXEC: wd = the 64 bits "left >= R on every compared resource lane" (node n0 + i = bit i).
// cur: on entry the block's FIRST group, issued by the caller (or by the block before) and not yet waited for; on exit the first group of the
// block behind this one — the group loads run ahead across the block boundary (a run's last block reads one group past its end: the next
// lane's row or left4's padding of 64 entries, never used).  What a wave keeps in flight is what covers the scalar loads' latency under
// load (~1 000 cycles and more with every SIMD full of such waves, profiles/r06_lean_v2_roles_trace.txt: one s_load_dwordx16 per wave in
// flight left the SIMDs 40 % idle at k = 1), hence sixteen nodes per group where the SGPRs allow it.
template <int MASK>
__device__ __forceinline__ void filter_block_lanes(cnode_t L4, uint32_t stride, uint32_t n0, const int64_t (&R)[4], int64_t (&cur)[4][LeanGroup<MASK>::GN],
                                                   uint32_t (&wd)[2]) {
  constexpr int K = LeanGroup<MASK>::K, GN = LeanGroup<MASK>::GN;
  static_assert(K >= 1 && K <= 4, "MASK names the compared lanes");
  constexpr int NG = 64 / GN;
  auto fold = [&](uint32_t (&acc)[4]) -> uint32_t {
    uint32_t v = ~0u;
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if ((MASK >> j) & 1) { v &= acc[j]; acc[j] = 0; }
    return __builtin_bitreverse32(v);
  };
  int64_t nxt[4][GN];
  uint32_t acc[4] = {0u, 0u, 0u, 0u};
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < GN; c += 4)
        if ((MASK >> j) & 1) filter_sgprs_ready4(cur[j][c], cur[j][c + 1], cur[j][c + 2], cur[j][c + 3]);
    lean_load<MASK, GN>(L4, stride, n0 + (uint32_t)(g + 1) * GN, nxt);          // (g == NG - 1: the next block's first group)
#pragma unroll
    for (int c = 0; c < GN; c += 4) {
      if (g * GN + c == 32) wd[0] = fold(acc);
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((MASK >> j) & 1) cmp4_acc(R[j], cur[j][c], cur[j][c + 1], cur[j][c + 2], cur[j][c + 3], acc[j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < GN; ++u)
        if ((MASK >> j) & 1) cur[j][u] = nxt[j][u];
  }
  wd[1] = fold(acc);
}
// node blocks [w0, w1) for the lanes in EXEC (evaluated slots).  MASK 1..15: the compared lanes; 0: every lane is free (case 2 holds wherever the
// node can be evaluated); -1: case 2 holds nowhere (no slot of the tile can pass it).  c2m: all ones in the lanes case 2 can hold for at all.
// NW: the (evaluable, cannot-hold-a-leader-member) word pairs of the tile's leader.  Returns the lane's count of passing nodes.
template <int MASK>
__device__ __forceinline__ uint32_t filter_run_lean(cnode_t L4, uint32_t nstride, cword_t NW, uint32_t w0, uint32_t w1, const int64_t (&R)[4], uint32_t c2m,
                                                    uint64_t* out, uint32_t ustride) {
  uint32_t cnt = 0;
  int64_t cur[4][LeanGroup<MASK>::GN];
  if constexpr (MASK > 0) lean_load<MASK, LeanGroup<MASK>::GN>(L4, nstride, w0 * 64u, cur);
  for (uint32_t w = w0; w < w1; ++w, out += ustride) {
    const unsigned long long ok = NW[2u * w], nohold = NW[2u * w + 1u];
    uint32_t c2w[2] = {~0u, ~0u};
    if constexpr (MASK > 0) filter_block_lanes<MASK>(L4, nstride, w * 64u, R, cur, c2w);
    else if constexpr (MASK < 0) { c2w[0] = 0u; c2w[1] = 0u; }
    const uint32_t lo = ((c2w[0] & c2m) | (uint32_t)nohold) & (uint32_t)ok;          // core.go:551-563: case 2, else case 3
    const uint32_t hi = ((c2w[1] & c2m) | (uint32_t)(nohold >> 32)) & (uint32_t)(ok >> 32);
    cnt += (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
    *out = ((uint64_t)hi << 32) | lo;
  }
  return cnt;
}

// ------------------------------------------------------------------------------------------------
// T tiles per wave (round 6).  One scalar-loaded node serves T x 64 request slots: the wave keeps T request sets in its VGPRs (2 k VGPRs per
// tile) and T x k accumulators, and everything that is per NODE — the scalar loads, their waits, the node words, the loop — is paid once for T
// tiles.  What bounds the one-tile loop is the latency of the scalar loads a wave can keep in flight (<= 128 bytes of node data in 64 SGPRs
// against ~1 000 cycles under load: tools/ubench/lane_loop has the VALU sequence at 8.4 cycles per node and lane, the launch ran at 22 with one
// lane, profiles/r06_lean_v3_ab.jsonl); T tiles multiply the compares a load feeds without one more SGPR.
// ------------------------------------------------------------------------------------------------
#ifndef BS_FL_T
#define BS_FL_T 2
#endif
template <int MASK, int T>
__device__ __forceinline__ void filter_block_lanes_t(cnode_t L4, uint32_t stride, uint32_t n0, const int64_t (&R)[T][4], int64_t (&cur)[4][LeanGroup<MASK>::GN],
                                                     uint32_t (&wd)[T][2]) {
  constexpr int K = LeanGroup<MASK>::K, GN = LeanGroup<MASK>::GN;
  static_assert(K >= 1 && K <= 4, "MASK names the compared lanes");
  constexpr int NG = 64 / GN;
  int64_t nxt[4][GN];
  uint32_t acc[T][4];
#pragma unroll
  for (int t = 0; t < T; ++t)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[t][j] = 0u;
  auto fold = [&](int half) {
#pragma unroll
    for (int t = 0; t < T; ++t) {
      uint32_t v = ~0u;
#pragma unroll
      for (int j = 0; j < 4; ++j)
        if ((MASK >> j) & 1) { v &= acc[t][j]; acc[t][j] = 0u; }
      wd[t][half] = __builtin_bitreverse32(v);
    }
  };
#pragma unroll
  for (int g = 0; g < NG; ++g) {
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < GN; c += 4)
        if ((MASK >> j) & 1) filter_sgprs_ready4(cur[j][c], cur[j][c + 1], cur[j][c + 2], cur[j][c + 3]);
    lean_load<MASK, GN>(L4, stride, n0 + (uint32_t)(g + 1) * GN, nxt);          // (g == NG - 1: the next block's first group)
#pragma unroll
    for (int c = 0; c < GN; c += 4) {
      if (g * GN + c == 32) fold(0);
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < T; ++t)
          if ((MASK >> j) & 1) cmp4_acc(R[t][j], cur[j][c], cur[j][c + 1], cur[j][c + 2], cur[j][c + 3], acc[t][j]);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int u = 0; u < GN; ++u)
        if ((MASK >> j) & 1) cur[j][u] = nxt[j][u];
  }
  fold(1);
}
// node blocks [w0, w1) for T tiles, EVERY lane active: c2m / evm are the per-lane masks (all ones or zero) "case 2 can hold for this slot" /
// "the slot is evaluated"; a lane outside its tile's slots carries zeros and stores into the spare columns behind the slot rows (reserve_slots
// keeps 64 of them); live[t]: the tile has evaluated slots (wave-uniform).  MASK: the union of the tiles' compared lanes — a lane that is free
// for one of them compares true on every node that can be evaluated, which is what "free" means —, 0: no lane to compare.
template <int MASK, int T>
__device__ __forceinline__ void filter_run_lean_t(cnode_t L4, uint32_t nstride, cword_t NW, uint32_t w0, uint32_t w1, const int64_t (&R)[T][4],
                                                  const uint32_t (&c2m)[T], const uint32_t (&evm)[T], const bool (&live)[T], uint64_t* (&out)[T], uint32_t ustride,
                                                  uint32_t (&cnt)[T]) {
  int64_t cur[4][LeanGroup<MASK>::GN];
  if constexpr (MASK > 0) lean_load<MASK, LeanGroup<MASK>::GN>(L4, nstride, w0 * 64u, cur);
  for (uint32_t w = w0; w < w1; ++w) {
    const unsigned long long ok = NW[2u * w], nohold = NW[2u * w + 1u];
    uint32_t c2w[T][2];
#pragma unroll
    for (int t = 0; t < T; ++t) { c2w[t][0] = ~0u; c2w[t][1] = ~0u; }
    if constexpr (MASK > 0) filter_block_lanes_t<MASK, T>(L4, nstride, w * 64u, R, cur, c2w);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint32_t lo = ((c2w[t][0] & c2m[t]) | (uint32_t)nohold) & (uint32_t)ok & evm[t];          // core.go:551-563: case 2, else case 3
      const uint32_t hi = ((c2w[t][1] & c2m[t]) | (uint32_t)(nohold >> 32)) & (uint32_t)(ok >> 32) & evm[t];
      cnt[t] += (uint32_t)__popc(lo) + (uint32_t)__popc(hi);
      if (live[t]) *out[t] = ((uint64_t)hi << 32) | lo;
      out[t] += ustride;
    }
  }
}
// what the lean loop needs of one tile (the prologue of filter_item_t, without what only the fallback path uses):
//   R          the lane's request (zeros outside the evaluated slots)
//   c2m, evm   all ones / zero: case 2 can hold for this slot at all; the slot is evaluated
//   lane_mask  the tile's compared lanes (0 when case 2 is decided for the whole tile)
//   sel        node-word table of the tile's leader; -1: the tile needs filter_item_t (mixed leaders, no node words); -2: no slot in use
//   fix, other a slot of the tile that is NOT evaluated (its rows: every node in range when `other`, else nothing)
__device__ __forceinline__ void lean_tile_head(const NodesDev& nd, const BatchDev& b, uint32_t U, uint32_t ptile, uint32_t stamp, uint32_t ff, const int64_t (&gl)[8],
                                               int64_t (&R)[4], uint32_t& c2m_o, uint32_t& evm_o, uint32_t& lane_mask_o, int& sel_o, bool& fix_o, bool& other_o) {
  const int lane = lane_id();
  const uint32_t p0 = ptile * 64u;
  const uint32_t np = p0 < U ? min(64u, U - p0) : 0u;
  const bool mine = (uint32_t)lane < np;
  uint32_t myff = mine ? ff : ((uint32_t)BS_FL_NOT_RUN << 8);
  if (stamp) myff = (myff >> 16) == stamp ? (myff & 0xFFFFu) : ((uint32_t)BS_FL_NOT_RUN << 8);
  const uint32_t myfl = myff >> 8;
  const bool ev = myfl == BS_FL_EVALUATED;
  const unsigned long long evmask = __ballot(ev);
  sel_o = -2;
  c2m_o = evm_o = lane_mask_o = 0u;
  fix_o = other_o = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) R[j] = 0;
  if (!evmask) return;
  int64_t M[4] = {0, 0, 0, 0};
  if (ev) {
    const int64_t* rs = b.uparams + (size_t)(p0 + (uint32_t)lane) * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { R[j] = rs[j]; M[j] = rs[4 + j]; }
  }
  const bool c2pod = ev && !(myff & 1u);
  const unsigned long long c2mask = __ballot(c2pod);
  uint32_t lane_mask = 0;
  bool tile_allfail = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (__ballot(c2pod && !(gl[j] >= R[j]))) lane_mask |= 1u << j;
    if (c2mask && !__ballot(c2pod && gl[4 + j] >= R[j])) tile_allfail = true;
  }
  const bool c2run = c2mask && !tile_allfail;
  int64_t M0[4];
  bool same = true;
  const int first = __ffsll((long long)evmask) - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    M0[j] = __shfl(M[j], first);
    same = same && (M[j] == M0[j]);
  }
  const uint32_t lb0 = (uint32_t)__shfl((int)(myff & 2u), first);
  same = same && ((myff & 2u) == lb0);
  const bool uniformM = __ballot(ev && !same) == 0;
  int sel = -1;
  if (b.nodew && !b.h_rows && uniformM) {
    if (lb0) sel = 2;
    else {
      const cnode_t ref = (cnode_t)(uintptr_t)(b.nodew + (size_t)6 * b.nodew_stride);
#pragma unroll
      for (int s2 = 1; s2 >= 0; --s2) {
        const bool eq = (ref[8 + s2] & 3) == 1 && ref[4 * s2] == M0[0] && ref[4 * s2 + 1] == M0[1] && ref[4 * s2 + 2] == M0[2] && ref[4 * s2 + 3] == M0[3];
        if (__ballot(eq) != 0) sel = s2;
      }
    }
  }
  sel_o = __builtin_amdgcn_readfirstlane(sel);
  c2m_o = (c2pod && c2run) ? ~0u : 0u;
  evm_o = ev ? ~0u : 0u;
  lane_mask_o = c2run ? lane_mask : 0u;
  fix_o = mine && !ev;
  other_o = myfl < 16u;
}
__device__ __forceinline__ void filter_item_t(const NodesDev& nd, const BatchDev& b, uint32_t U, uint32_t ustride, uint32_t ptile, uint32_t w0, uint32_t w1,
                                              uint32_t stamp, uint32_t ff, uint32_t collect_stats);
// One item = (T neighbouring tiles of 64 request slots, node blocks [w0, w1)).  When every tile that has slots in use can take the lean loop on
// the same node-word table, they take it together; otherwise (the tile across the two leader halves of the slot array, latency-mode batches with
// host rows, contexts without node words) tile by tile through filter_item_t.
template <int T>
__device__ __forceinline__ void filter_item_multi(const NodesDev& nd, const BatchDev& b, uint32_t U, uint32_t ustride, uint32_t ptile0, uint32_t w0, uint32_t w1,
                                                  uint32_t stamp, const uint32_t (&ff)[T], uint32_t collect_stats) {
  const int lane = lane_id();
  int64_t gl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gl[j] = nd.lglob[j];
  int64_t R[T][4];
  uint32_t c2m[T], evm[T], cnt[T];
  bool live[T], fix[T], other[T];
  int tsel[T];
  uint64_t* out[T];
  int sel = -2;
  bool together = true;
  uint32_t mask = 0;
#pragma unroll
  for (int t = 0; t < T; ++t) {
    uint32_t lm;
    lean_tile_head(nd, b, U, ptile0 + (uint32_t)t, stamp, ff[t], gl, R[t], c2m[t], evm[t], lm, tsel[t], fix[t], other[t]);
    if (tsel[t] == -1) together = false;
    else if (tsel[t] >= 0) {
      if (sel >= 0 && sel != tsel[t]) together = false;
      sel = tsel[t];
      mask |= lm;
    }
  }
  if (sel == -2 && together) return;                  // no slot in use in any of the tiles
  if (!together) {
#pragma unroll
    for (int t = 0; t < T; ++t)
      if (tsel[t] != -2) filter_item_t(nd, b, U, ustride, ptile0 + (uint32_t)t, w0, w1, stamp, ff[t], collect_stats);
    return;
  }
  const cnode_t L4 = (cnode_t)(uintptr_t)nd.left4;
  const cword_t NW = (cword_t)(uintptr_t)(b.nodew + (size_t)sel * b.nodew_stride * 2);
#pragma unroll
  for (int t = 0; t < T; ++t) {
    cnt[t] = 0u; live[t] = tsel[t] >= 0;
    out[t] = b.fu_bitmap + (size_t)w0 * ustride + (size_t)(ptile0 + (uint32_t)t) * 64u + lane;
  }
  if (collect_stats && lane == 0) {                  // the tiles of a pair are compared on the UNION of their lanes: that is what the launch executes
    uint32_t nlive = 0;
#pragma unroll
    for (int t = 0; t < T; ++t) nlive += live[t] ? 1u : 0u;
    atomicAdd((unsigned long long*)&b.stats[5], (unsigned long long)__popc(mask) * nlive * (w1 - w0));
    atomicAdd((unsigned long long*)&b.stats[6], (unsigned long long)nlive * (w1 - w0));
  }
  switch (mask) {
    case 0: filter_run_lean_t<0, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 1: filter_run_lean_t<1, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 2: filter_run_lean_t<2, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 3: filter_run_lean_t<3, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 4: filter_run_lean_t<4, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 5: filter_run_lean_t<5, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 6: filter_run_lean_t<6, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 7: filter_run_lean_t<7, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 8: filter_run_lean_t<8, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 9: filter_run_lean_t<9, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 10: filter_run_lean_t<10, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 11: filter_run_lean_t<11, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 12: filter_run_lean_t<12, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 13: filter_run_lean_t<13, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    case 14: filter_run_lean_t<14, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
    default: filter_run_lean_t<15, T>(L4, nd.stride, NW, w0, w1, R, c2m, evm, live, out, ustride, cnt); break;
  }
#pragma unroll
  for (int t = 0; t < T; ++t) {
    const uint32_t slot = (ptile0 + (uint32_t)t) * 64u + (uint32_t)lane;
    if (cnt[t]) atomicAdd(&b.fu_feas[slot], cnt[t]);                                 // (only evaluated slots count anything)
    if (live[t] && __ballot(fix[t])) {             // slots of the tile that are not evaluated (rare): nil before any node lookup / error -> every node in range
      if (fix[t]) {
        uint64_t* o = b.fu_bitmap + (size_t)w0 * ustride + slot;
        uint32_t c = 0;
        for (uint32_t w = w0; w < w1; ++w, o += ustride) {
          const uint32_t left = nd.n - w * 64u;
          const unsigned long long word = other[t] ? (left >= 64u ? ~0ull : ((1ull << left) - 1ull)) : 0ull;
          c += (uint32_t)__popcll(word);
          *o = word;
        }
        if (c) atomicAdd(&b.fu_feas[slot], c);
      }
    }
  }
}

// One item = (tile of 64 request slots, node blocks [w0, w1)); same contract and same outputs as filter_item.
// ff = the lane's slot flags word (b.uflags[p0 + lane], fetched by filter_loop_t one item ahead): a tile without a slot in use — the
// carried-leader half of the slot array in most batches, 7 tiles of 8 on a rank of 8 (class ids follow the queue, k_pod_class_ids) —
// returns before it has issued another load (as written up to round 4 it had asked for its 4 KB of requests and 2.5 KB of nodes by
// then: 118 895 items, 772 MB of dead fetches, when a rank of 8 cut its items as fine as its share of live tiles called for).
__device__ __forceinline__ void filter_item_t(const NodesDev& nd, const BatchDev& b, uint32_t U, uint32_t ustride, uint32_t ptile, uint32_t w0,
                                              uint32_t w1, uint32_t stamp, uint32_t ff, uint32_t collect_stats) {
  const int lane = lane_id();
  const uint32_t p0 = ptile * 64u;
  const uint32_t np = min(64u, U - p0);
  const bool mine = (uint32_t)lane < np;
  const uint32_t src = p0 + (uint32_t)lane;
  uint32_t myff = mine ? ff : ((uint32_t)BS_FL_NOT_RUN << 8);
  if (stamp) myff = (myff >> 16) == stamp ? (myff & 0xFFFFu) : ((uint32_t)BS_FL_NOT_RUN << 8);
  const uint32_t myfl = myff >> 8;
  const bool ev = myfl == BS_FL_EVALUATED;
  const unsigned long long evmask = __ballot(ev);
  if (!evmask) return;                              // no slot of this tile is in use
  // one round trip: the slots' requests R and leader requests M, the cluster-wide bounds, the first node block
  int64_t M[4] = {0, 0, 0, 0}, R[4] = {0, 0, 0, 0};
  if (mine) {
    const int64_t* rs = b.uparams + (size_t)src * 8;
#pragma unroll
    for (int j = 0; j < 4; ++j) { R[j] = rs[j]; M[j] = rs[4 + j]; }
  }
  int64_t gl[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) gl[j] = nd.lglob[j];
  int64_t l[4], ln[4];                              // lanes = nodes: the block's left values, for the pod-independent masks only
  uint8_t nfl, nfln = 0xFF;
  auto load_block = [&](uint32_t w, int64_t (&dst)[4], uint8_t& fl) {
    const uint32_t n = w * 64u + (uint32_t)lane;
    fl = 0xFF;                                      // invalid
#pragma unroll
    for (int j = 0; j < 4; ++j) dst[j] = INT64_MIN;
    if (w < w1 && n < nd.n) {
#pragma unroll
      for (int j = 0; j < 4; ++j) dst[j] = nd.left4[(size_t)j * nd.stride + n];
      fl = nd.flags[n];
    }
  };
  if (!ev) {
#pragma unroll
    for (int j = 0; j < 4; ++j) M[j] = 0;
  }
  // which resource lanes can decide anything for this tile (nd.lglob: cluster-wide min[4] / max[4] of left over the nodes Filter
  // can evaluate): a lane is free when even the smallest left covers every request of the tile; the tile fails everywhere when
  // the largest left of some lane is below every request
  const bool c2pod = ev && !(myff & 1u);
  const unsigned long long c2mask = __ballot(c2pod);
  uint32_t lane_mask = 0;
  bool tile_allfail = false;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    if (__ballot(c2pod && !(gl[j] >= R[j]))) lane_mask |= 1u << j;
    if (c2mask && !__ballot(c2pod && gl[4 + j] >= R[j])) tile_allfail = true;
  }
  if (collect_stats && lane == 0) {                  // compared lanes x node blocks | node blocks of this tile run (bs_batch_stats: filter_lane_blocks / filter_tile_blocks)
    atomicAdd((unsigned long long*)&b.stats[5], (unsigned long long)((c2mask && !tile_allfail) ? __popc(lane_mask) : 0) * (w1 - w0));
    atomicAdd((unsigned long long*)&b.stats[6], (unsigned long long)(w1 - w0));
  }
  // is the leader's single-member request M the same for every evaluated slot of the tile?
  int64_t M0[4];
  bool same = true;
  const int first = __ffsll((long long)evmask) - 1;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    M0[j] = __shfl(M[j], first);
    same = same && (M[j] == M0[j]);
  }
  const uint32_t lb0 = (uint32_t)__shfl((int)(myff & 2u), first);
  same = same && ((myff & 2u) == lb0);
  const bool uniformM = __ballot(ev && !same) == 0;
  const unsigned long long lfmask = __ballot(ev && !(myff & 2u));     // lanes whose case 3 depends on the node (own M)

  const cnode_t L4 = (cnode_t)(uintptr_t)nd.left4;
  uint32_t cnt = 0;
  // Round 6: the pod-independent words of a node block come from the batch's node words (launch A, node_words_block) when the tile's slots
  // all name one of the batch's two leaders — every tile but the one that straddles the two halves of the slot array — and the item runs the
  // scalar-lean block loop (filter_run_lean, above).  sel = which table (2: the leader's MinResources names a scalar: no node holds a member).
  int sel = -1;
  if (b.nodew && !b.h_rows && uniformM) {
    if (lb0) sel = 2;
    else {
      const cnode_t ref = (cnode_t)(uintptr_t)(b.nodew + (size_t)6 * b.nodew_stride);
#pragma unroll
      for (int s2 = 1; s2 >= 0; --s2) {
        const bool eq = (ref[8 + s2] & 3) == 1 && ref[4 * s2] == M0[0] && ref[4 * s2 + 1] == M0[1] && ref[4 * s2 + 2] == M0[2] && ref[4 * s2 + 3] == M0[3];
        if (__ballot(eq) != 0) sel = s2;              // (M0 is the same in every lane)
      }
    }
  }
  sel = __builtin_amdgcn_readfirstlane(sel);          // (uniform by construction; the compiler sees lb0's __shfl)
  if (sel >= 0) {
    const cword_t NW = (cword_t)(uintptr_t)(b.nodew + (size_t)sel * b.nodew_stride * 2);
    uint64_t* out = b.fu_bitmap + (size_t)w0 * ustride + p0 + lane;
    if (ev) {                                         // EXEC = the evaluated slots, once for the whole run of blocks
      const uint32_t c2m = c2pod ? ~0u : 0u;
      uint32_t cnt = 0;
      switch ((c2mask && !tile_allfail) ? (int)lane_mask : -1) {
        case -1: cnt = filter_run_lean<-1>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 0: cnt = filter_run_lean<0>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 1: cnt = filter_run_lean<1>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 2: cnt = filter_run_lean<2>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 3: cnt = filter_run_lean<3>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 4: cnt = filter_run_lean<4>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 5: cnt = filter_run_lean<5>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 6: cnt = filter_run_lean<6>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 7: cnt = filter_run_lean<7>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 8: cnt = filter_run_lean<8>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 9: cnt = filter_run_lean<9>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 10: cnt = filter_run_lean<10>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 11: cnt = filter_run_lean<11>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 12: cnt = filter_run_lean<12>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 13: cnt = filter_run_lean<13>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        case 14: cnt = filter_run_lean<14>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
        default: cnt = filter_run_lean<15>(L4, nd.stride, NW, w0, w1, R, c2m, out, ustride); break;
      }
      if (cnt) atomicAdd(&b.fu_feas[p0 + lane], cnt);
    }
    if (__ballot(mine && !ev)) {                      // slots of the tile that are not evaluated (rare): nil before any node lookup / error -> every node in range
      if (mine && !ev) {
        uint32_t cnt = 0;
        for (uint32_t w = w0; w < w1; ++w, out += ustride) {
          const uint32_t left = nd.n - w * 64u;
          const unsigned long long word = myfl < 16u ? (left >= 64u ? ~0ull : ((1ull << left) - 1ull)) : 0ull;
          cnt += (uint32_t)__popcll(word);
          *out = word;
        }
        if (cnt) atomicAdd(&b.fu_feas[p0 + lane], cnt);
      }
    }
    return;
  }
  load_block(w0, l, nfl);
  for (uint32_t w = w0; w < w1; ++w) {
    load_block(w + 1u, ln, nfln);                   // (in flight during this block's compare loop; invalid behind w1)
    const bool nvalid = nfl != 0xFF;
    const bool node_ok = nvalid && !(nfl & (BS_NODE_NIL | BS_NODE_NO_NODE));     // core.go:442-449
    const unsigned long long in_range = __ballot(nvalid), okmask = __ballot(node_ok);
    // case 2 per (slot, node)
    uint32_t c2w[2] = {0u, 0u};
    if (c2mask && !tile_allfail) {
      if (lane_mask == 0u) { c2w[0] = (uint32_t)okmask; c2w[1] = (uint32_t)(okmask >> 32); }     // every lane is free
      else filter_block_any(lane_mask, L4, nd.stride, w * 64u, R, c2w);
    }
    unsigned long long c2 = (((unsigned long long)c2w[1] << 32) | c2w[0]) & okmask;
    if (!c2pod) c2 = 0;
    // case 3: nodes that cannot hold one leader member pass
    unsigned long long lf;
    if (uniformM) {
      lf = 0;
      if (!lb0) lf = __ballot(l[0] >= M0[0]) & __ballot(l[1] >= M0[1]) & __ballot(l[2] >= M0[2]) & __ballot(l[3] >= M0[3]);
    } else {
      uint32_t lfw[2] = {0u, 0u};
      if (lfmask) filter_block_t<15>(L4, nd.stride, w * 64u, M, lfw);
      lf = ((unsigned long long)lfw[1] << 32) | lfw[0];
      if (myff & 2u) lf = 0;
    }
    unsigned long long word;
    if (ev) word = okmask & (c2 | ~lf);
    else word = myfl < 16u ? in_range : 0ull;       // nil before any node lookup / error
    if (mine) {
      cnt += (uint32_t)__popcll(word);
      b.fu_bitmap[(size_t)w * ustride + p0 + lane] = word;
      if (b.h_rows && p0 + (uint32_t)lane < b.hstride) b.h_rows[(size_t)w * b.hstride + p0 + lane] = word;   // latency mode: the row goes home as well
    }
    nfl = nfln;
#pragma unroll
    for (int j = 0; j < 4; ++j) l[j] = ln[j];
  }
  if (mine && cnt) atomicAdd(&b.fu_feas[p0 + lane], cnt);
}

// filter_loop's split of the work (tiles x node runs), items taken by filter_item_t.  AHEAD: the flags of the wave's next item are
// fetched while the current one runs (one VGPR held across the item).
// Item order.  by_tile == 0: chunk-major (item = chunk * tiles + tile) — with at most one item per wave the order only decides who
// shares a node run.  by_tile != 0 (a rank of a sharded job: more items than waves, most of them idle): quads of four neighbouring
// tiles, chunk by chunk (item = ((tile / 4) * nchunk + chunk) * 4 + tile % 4: the four waves of a block still share their node run).
// A rank's live tiles are neighbours (class ids follow the queue), so its live items are CONSECUTIVE and the waves' strided walk deals
// them out evenly; chunk-major, a wave's items fall on tiles (w + k * waves) mod tiles, live by chance — rank 0 of 8 on cfg4, 60 200
// items on 16 384 waves: some waves drew two or three live items of 16 us each and the launch lasted as long as those, while the
// stamped live blocks were done after 16 us (tools/stamp_probe.py, profiles/r05_shard_scaling.md: 72 -> 60 us per step).
template <bool AHEAD, int T>
__device__ __forceinline__ void filter_loop_tiles(const NodesDev& nd, const BatchDev& b, uint32_t target_waves, uint32_t ustride, uint32_t collect_stats,
                                              uint32_t bx, uint32_t nblocks, uint32_t stamp, uint32_t slots, uint32_t by_tile) {
  const uint32_t U = slots ? slots : 2u * __builtin_amdgcn_readfirstlane(*b.kclass);
  const uint32_t W = (nd.n + 63u) / 64u;
  if (!U || !W) return;
  const uint32_t tiles1 = (U + 63u) / 64u;
  const uint32_t tiles = (tiles1 + (uint32_t)T - 1u) / (uint32_t)T;                 // items take T neighbouring tiles together (filter_item_multi)
  uint32_t nsplit = max(1u, target_waves / tiles);
  nsplit = min(nsplit, max((W + 1u) / 2u, 1u));
  const uint32_t bpw = max(2u, (((W + nsplit - 1u) / nsplit + 1u) / 2u) * 2u);
  const uint32_t nchunk = (W + bpw - 1u) / bpw;
  const uint32_t items = by_tile ? ((tiles + 3u) / 4u) * nchunk * 4u : tiles * nchunk;
  auto decode = [&](uint32_t it, uint32_t& tile, uint32_t& chunk) {
    if (by_tile) {
      const uint32_t q = it / (nchunk * 4u), rem = it - q * (nchunk * 4u);
      chunk = rem >> 2;
      tile = q * 4u + (rem & 3u);                                                   // (may lie behind the last tile: no slots, idle)
    } else {
      chunk = it / tiles;
      tile = it - chunk * tiles;                                                    // neighbours share the node run
    }
  };
  auto slot_flags = [&](uint32_t it, uint32_t (&ff)[T]) {                           // the item's tiles: their lanes' flags words
    uint32_t tile, chunk;
    decode(it, tile, chunk);
#pragma unroll
    for (int t = 0; t < T; ++t) {
      const uint32_t sl = (tile * (uint32_t)T + (uint32_t)t) * 64u + (uint32_t)lane_id();
      ff[t] = (it < items && sl < U) ? b.uflags[sl] : ((uint32_t)BS_FL_NOT_RUN << 8);
    }
  };
  uint32_t it = __builtin_amdgcn_readfirstlane(bx * 4u + (uint32_t)wave_id());
  uint32_t ff[T], ffn[T];
#pragma unroll
  for (int t = 0; t < T; ++t) ff[t] = ffn[t] = 0u;
  if (AHEAD) slot_flags(it, ff);
  while (it < items) {
    if (!AHEAD) slot_flags(it, ff);
    const uint32_t nx = it + nblocks * 4u;
    if (AHEAD) slot_flags(nx, ffn);                                                 // (in flight while this item runs)
    uint32_t tile, chunk;
    decode(it, tile, chunk);
    if (collect_stats && chunk == 0) {
#pragma unroll
      for (int t = 0; t < T; ++t) {
        const unsigned long long evs = __ballot(((ff[t] >> 8) & 0xFFu) == BS_FL_EVALUATED && (!stamp || (ff[t] >> 16) == stamp));
        if (lane_id() == 0 && evs) atomicAdd((unsigned long long*)&b.stats[3], (unsigned long long)__popcll(evs));
      }
    }
    if (tile < tiles) {
      if constexpr (T == 1) filter_item_t(nd, b, U, ustride, tile, chunk * bpw, min(W, chunk * bpw + bpw), stamp, ff[0], collect_stats);
      else filter_item_multi<T>(nd, b, U, ustride, tile * (uint32_t)T, chunk * bpw, min(W, chunk * bpw + bpw), stamp, ff, collect_stats);
    }
    it = nx;
#pragma unroll
    for (int t = 0; t < T; ++t) ff[t] = ffn[t];
  }
}
// Tiles per item.  Two where the batch has tiles enough to fill the chip twice over with pairs of them and the node words are there for the
// lean loop (cfg4 all-distinct, 1 506 tiles: 174 -> 151 us per step at one compared lane, 407 -> 324 at four, profiles/r06_lean_v4_T2_ab.jsonl);
// one below that (cfg3 all-distinct, 303 tiles: pairs halve the items a launch of ~20 us has: 34.4 -> 36.8 us) and without node words
// (BS_NO_NODEW=1: round 5's item, tile by tile).  Four tiles per item measured WORSE than two at cfg4 (195 / 279 / 344 us at one / two / four
// lanes against 151 / 200 / 324: a tile's prologue is then spread over a quarter of the node blocks).
// The threshold travels with the batch (BatchDev::tiles2_min = BS_TP_TMIN, default 768, times the ranks of a sharded context: a rank's live
// tiles are its share of them).
template <bool AHEAD>
__device__ __forceinline__ void filter_loop_t(const NodesDev& nd, const BatchDev& b, uint32_t target_waves, uint32_t ustride, uint32_t collect_stats,
                                              uint32_t bx, uint32_t nblocks, uint32_t stamp, uint32_t slots, uint32_t by_tile) {
  const uint32_t U = slots ? slots : 2u * __builtin_amdgcn_readfirstlane(*b.kclass);
  if (BS_FL_T > 1 && b.nodew && !b.h_rows && (U + 63u) / 64u >= b.tiles2_min)
    filter_loop_tiles<AHEAD, BS_FL_T>(nd, b, target_waves, ustride, collect_stats, bx, nblocks, stamp, U, by_tile);
  else
    filter_loop_tiles<AHEAD, 1>(nd, b, target_waves, ustride, collect_stats, bx, nblocks, stamp, U, by_tile);
}

}  // namespace bs
