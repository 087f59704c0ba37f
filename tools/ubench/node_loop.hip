// Micro-benchmark (not part of the product): issue cost of the per-NODE sequence of the transposed Filter item
// (csrc/bs_filter_t.hpp) and of the alternatives DESIGN.md section 9 lists, per compared-lane count k = 1 and k = 4, at 1 .. 8 waves per SIMD.
// The product loop measured ~2x its own issue estimate (k = 1: 2 VALU + 2 SALU per node and wave); this tells which part of the
// sequence pays.  Build + run on the GPU box:
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/node_loop tools/ubench/node_loop.hip && tools/ubench/node_loop
// Output: cycles per NODE per SIMD (at 2.4 GHz) for
//   A  the product sequence:      s_mov exec, em ; k x v_cmpx_ge_i64 (SGPR left, VGPR R) ; v_or word, sb, word ; s_lshl sb
//   B  literal bits:              the same without the s_lshl (bit as an inline constant / literal)
//   C  no EXEC reset:             k x v_cmpx + v_or only (not a valid program: the bound the EXEC set-up is measured against)
//   D  32-bit compares:           v_cmpx_ge_i32 on the low dwords (valid when left and R fit 31 bits: cpu-milli, pod counts)
//   E  v_cmp into SGPRs + s_and:  k x v_cmp_ge_i64 s[m], .. ; s_and exec ; v_or   (no EXEC chain through the VALU)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define NODES 64
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

#define N1(body) body
#define N4(body) body body body body
#define N16(body) N4(body) N4(body) N4(body) N4(body)
#define N64(body) N16(body) N16(body) N16(body) N16(body)

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tab, int64_t* out, int iters) {
  const int64_t R0 = out[threadIdx.x & 63], R1 = R0 + 1, R2 = R0 + 2, R3 = R0 + 3;
  const int64_t a0 = tab[0], a1 = tab[1], a2 = tab[2], a3 = tab[3];     // uniform -> SGPR pairs (a node's left)
  const unsigned long long em = 0xFFFFFFFF0000FFFFull | (unsigned long long)tab[4];
  uint32_t w = 0, sb = 1;
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
      if (K == 1)
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_or_b32 %[w], %[sb], %[w]\n\ts_lshl_b32 %[sb], %[sb], 1\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w), [sb] "+s"(sb) : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0) : "vcc", "scc");
      else
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[R1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[R2]\n\t"
                         "v_cmpx_ge_i64 vcc, %[a3], %[R3]\n\tv_or_b32 %[w], %[sb], %[w]\n\ts_lshl_b32 %[sb], %[sb], 1\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w), [sb] "+s"(sb) : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0), [a1] "s"(a1), [R1] "v"(R1), [a2] "s"(a2), [R2] "v"(R2), [a3] "s"(a3), [R3] "v"(R3)
                     : "vcc", "scc");
    } else if (MODE == 1) {
      if (K == 1)
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w) : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0) : "vcc");
      else
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[R1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[R2]\n\t"
                         "v_cmpx_ge_i64 vcc, %[a3], %[R3]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w) : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0), [a1] "s"(a1), [R1] "v"(R1), [a2] "s"(a2), [R2] "v"(R2), [a3] "s"(a3), [R3] "v"(R3) : "vcc");
    } else if (MODE == 2) {
      if (K == 1)
        asm volatile(N64("v_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1" : [w] "+v"(w) : [a0] "s"(a0), [R0] "v"(R0) : "vcc");
      else
        asm volatile(N64("v_cmpx_ge_i64 vcc, %[a0], %[R0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[R1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[R2]\n\tv_cmpx_ge_i64 vcc, %[a3], %[R3]\n\t"
                         "v_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w) : [a0] "s"(a0), [R0] "v"(R0), [a1] "s"(a1), [R1] "v"(R1), [a2] "s"(a2), [R2] "v"(R2), [a3] "s"(a3), [R3] "v"(R3) : "vcc");
    } else if (MODE == 3) {
      const int32_t b0 = (int32_t)a0, b1 = (int32_t)a1, b2 = (int32_t)a2, b3 = (int32_t)a3, Q0 = (int32_t)R0, Q1 = (int32_t)R1, Q2 = (int32_t)R2, Q3 = (int32_t)R3;
      if (K == 1)
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i32 vcc, %[a0], %[R0]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w) : [em] "s"(em), [a0] "s"(b0), [R0] "v"(Q0) : "vcc");
      else
        asm volatile(N64("s_mov_b64 exec, %[em]\n\tv_cmpx_ge_i32 vcc, %[a0], %[R0]\n\tv_cmpx_ge_i32 vcc, %[a1], %[R1]\n\tv_cmpx_ge_i32 vcc, %[a2], %[R2]\n\t"
                         "v_cmpx_ge_i32 vcc, %[a3], %[R3]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w) : [em] "s"(em), [a0] "s"(b0), [R0] "v"(Q0), [a1] "s"(b1), [R1] "v"(Q1), [a2] "s"(b2), [R2] "v"(Q2), [a3] "s"(b3), [R3] "v"(Q3) : "vcc");
    } else {
      unsigned long long m0, m1, m2, m3;
      if (K == 1)
        asm volatile(N64("v_cmp_ge_i64 %[m0], %[a0], %[R0]\n\ts_and_b64 exec, %[m0], %[em]\n\tv_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w), [m0] "=&s"(m0) : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0) : "vcc", "scc");
      else
        asm volatile(N64("v_cmp_ge_i64 %[m0], %[a0], %[R0]\n\tv_cmp_ge_i64 %[m1], %[a1], %[R1]\n\tv_cmp_ge_i64 %[m2], %[a2], %[R2]\n\tv_cmp_ge_i64 %[m3], %[a3], %[R3]\n\t"
                         "s_and_b64 %[m0], %[m0], %[m1]\n\ts_and_b64 %[m2], %[m2], %[m3]\n\ts_and_b64 %[m0], %[m0], %[m2]\n\ts_and_b64 exec, %[m0], %[em]\n\t"
                         "v_or_b32 %[w], 16, %[w]\n\t") "s_mov_b64 exec, -1"
                     : [w] "+v"(w), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3)
                     : [em] "s"(em), [a0] "s"(a0), [R0] "v"(R0), [a1] "s"(a1), [R1] "v"(R1), [a2] "s"(a2), [R2] "v"(R2), [a3] "s"(a3), [R3] "v"(R3) : "vcc", "scc");
    }
  }
  if (w == 0x12345u) out[threadIdx.x] = (int64_t)w + sb;
}

template <int MODE, int K>
int run(const char* name, const int64_t* d_tab, int64_t* d_out, int waves_per_simd) {
  const int iters = 400, blocks = 256 * waves_per_simd;   // n blocks x 4 waves per CU = n waves per SIMD
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE, K>), dim3(blocks), dim3(256), 0, 0, d_tab, d_out, 2);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE, K>), dim3(blocks), dim3(256), 0, 0, d_tab, d_out, iters);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double nodes_per_simd = waves_per_simd * (double)iters * NODES;
  printf("[%d waves/SIMD] k=%d %-44s %8.3f ms -> %6.2f cycles per node per SIMD (at 2.4 GHz)\n", waves_per_simd, K, name, ms, ms * 1e-3 * 2.4e9 / nodes_per_simd);
  return 0;
}

int main() {
  int64_t *d_tab, *d_out;
  CHK(hipMalloc(&d_tab, 4096)); CHK(hipMalloc(&d_out, 4096));
  CHK(hipMemset(d_tab, 1, 4096)); CHK(hipMemset(d_out, 2, 4096));
  for (int w : {1, 2, 4, 8}) {
    run<0, 1>("A product: s_mov exec, cmpx, v_or sb, s_lshl", d_tab, d_out, w);
    run<1, 1>("B literal bit (no s_lshl)", d_tab, d_out, w);
    run<2, 1>("C no EXEC reset (bound)", d_tab, d_out, w);
    run<3, 1>("D 32-bit compare", d_tab, d_out, w);
    run<4, 1>("E v_cmp -> SGPR, s_and exec", d_tab, d_out, w);
    run<0, 4>("A product", d_tab, d_out, w);
    run<1, 4>("B literal bit", d_tab, d_out, w);
    run<2, 4>("C no EXEC reset (bound)", d_tab, d_out, w);
    run<3, 4>("D 32-bit compares", d_tab, d_out, w);
    run<4, 4>("E v_cmp -> SGPRs, s_and tree", d_tab, d_out, w);
  }
  return 0;
}
