// Micro-benchmark (not part of the product): VALU issue cost of the lean Filter loop's per-(node, resource lane) sequence (csrc/bs_filter_t.hpp, round 6) and of
// alternatives, without any memory traffic: what is the floor once the scalar loads are covered?
//   hipcc --offload-arch=gfx950 -O3 -o tools/ubench/lane_loop tools/ubench/lane_loop.hip && tools/ubench/lane_loop
// Output: cycles per (node, lane) per SIMD at 2.4 GHz, for K accumulators (= compared lanes) at 1 / 3 / 8 waves per SIMD
//   F  product:  4 x v_cmp_ge_i64_e64 -> SGPR pairs, 4 x v_addc_co_u32_e64 acc, acc, acc, mask            (2 VALU per node and lane)
//   G  through VCC:  v_cmp_ge_i64_e32 vcc ; v_addc_co_u32_e32 acc, vcc, acc, acc, vcc                      (same count, 32-bit encodings, serial on VCC)
//   I  32-bit compare:  v_cmp_ge_u32_e64 -> SGPR ; v_addc                                                 (rank-compressed operands)
//   P  packed 16-bit pairs:  v_pk_min_u16 t, a, R ; v_cmp_eq_u32_e64 -> SGPR (t == R  <=>  both halves a >= R) ; v_addc   (3 VALU per node and TWO lanes)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
#define N4(body) body body body body
#define N16(body) N4(body) N4(body) N4(body) N4(body)

template <int MODE, int K>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tab, int64_t* out, int iters) {
  int64_t R[4];
  for (int j = 0; j < 4; ++j) R[j] = out[(threadIdx.x & 63) + j];
  const int64_t a0 = tab[0], a1 = tab[1], a2 = tab[2], a3 = tab[3];     // uniform -> SGPR pairs (four nodes of one lane)
  uint32_t acc[4] = {0, 0, 0, 0};
  unsigned long long m0, m1, m2, m3;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < K; ++j) {
      if (MODE == 0)
        asm volatile(N16("v_cmp_ge_i64_e64 %[m0], %[a0], %[r]\n\tv_cmp_ge_i64_e64 %[m1], %[a1], %[r]\n\tv_cmp_ge_i64_e64 %[m2], %[a2], %[r]\n\tv_cmp_ge_i64_e64 %[m3], %[a3], %[r]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m0], %[w], %[w], %[m0]\n\tv_addc_co_u32_e64 %[w], %[m1], %[w], %[w], %[m1]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m2], %[w], %[w], %[m2]\n\tv_addc_co_u32_e64 %[w], %[m3], %[w], %[w], %[m3]\n\t")
                     : [w] "+v"(acc[j]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [r] "v"(R[j]), [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3));
      else if (MODE == 1)
        asm volatile(N16("v_cmp_ge_i64_e32 vcc, %[a0], %[r]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\tv_cmp_ge_i64_e32 vcc, %[a1], %[r]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\t"
                         "v_cmp_ge_i64_e32 vcc, %[a2], %[r]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\tv_cmp_ge_i64_e32 vcc, %[a3], %[r]\n\tv_addc_co_u32_e32 %[w], vcc, %[w], %[w], vcc\n\t")
                     : [w] "+v"(acc[j]) : [r] "v"(R[j]), [a0] "s"(a0), [a1] "s"(a1), [a2] "s"(a2), [a3] "s"(a3) : "vcc");
      else if (MODE == 2) {
        const uint32_t q = (uint32_t)R[j], b0 = (uint32_t)a0, b1 = (uint32_t)a1, b2 = (uint32_t)a2, b3 = (uint32_t)a3;
        asm volatile(N16("v_cmp_ge_u32_e64 %[m0], %[a0], %[r]\n\tv_cmp_ge_u32_e64 %[m1], %[a1], %[r]\n\tv_cmp_ge_u32_e64 %[m2], %[a2], %[r]\n\tv_cmp_ge_u32_e64 %[m3], %[a3], %[r]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m0], %[w], %[w], %[m0]\n\tv_addc_co_u32_e64 %[w], %[m1], %[w], %[w], %[m1]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m2], %[w], %[w], %[m2]\n\tv_addc_co_u32_e64 %[w], %[m3], %[w], %[w], %[m3]\n\t")
                     : [w] "+v"(acc[j]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3) : [r] "v"(q), [a0] "s"(b0), [a1] "s"(b1), [a2] "s"(b2), [a3] "s"(b3));
      } else if (MODE == 3) {
        const uint32_t q = (uint32_t)R[j], b0 = (uint32_t)a0, b1 = (uint32_t)a1, b2 = (uint32_t)a2, b3 = (uint32_t)a3;
        uint32_t t0, t1, t2, t3;
        asm volatile(N16("v_pk_min_u16 %[t0], %[a0], %[r]\n\tv_pk_min_u16 %[t1], %[a1], %[r]\n\tv_pk_min_u16 %[t2], %[a2], %[r]\n\tv_pk_min_u16 %[t3], %[a3], %[r]\n\t"
                         "v_cmp_eq_u32_e64 %[m0], %[t0], %[r]\n\tv_cmp_eq_u32_e64 %[m1], %[t1], %[r]\n\tv_cmp_eq_u32_e64 %[m2], %[t2], %[r]\n\tv_cmp_eq_u32_e64 %[m3], %[t3], %[r]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m0], %[w], %[w], %[m0]\n\tv_addc_co_u32_e64 %[w], %[m1], %[w], %[w], %[m1]\n\t"
                         "v_addc_co_u32_e64 %[w], %[m2], %[w], %[w], %[m2]\n\tv_addc_co_u32_e64 %[w], %[m3], %[w], %[w], %[m3]\n\t")
                     : [w] "+v"(acc[j]), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [t0] "=&v"(t0), [t1] "=&v"(t1), [t2] "=&v"(t2), [t3] "=&v"(t3)
                     : [r] "v"(q), [a0] "s"(b0), [a1] "s"(b1), [a2] "s"(b2), [a3] "s"(b3));
      }
    }
  }
  if ((acc[0] ^ acc[1] ^ acc[2] ^ acc[3]) == 0x12345u) out[threadIdx.x] = (int64_t)acc[0];
}

template <int MODE, int K>
int run(const char* name, const int64_t* d_tab, int64_t* d_out, int waves_per_simd, int lanes_per_acc) {
  const int iters = 100, blocks = 256 * waves_per_simd;   // n blocks x 4 waves per CU = n waves per SIMD
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
  const double node_lanes_per_simd = waves_per_simd * (double)iters * 64 * K * lanes_per_acc;
  printf("[%d waves/SIMD] K=%d %-50s %8.3f ms -> %6.2f cycles per (node, resource lane) per SIMD\n", waves_per_simd, K, name, ms, ms * 1e-3 * 2.4e9 / node_lanes_per_simd);
  return 0;
}

int main() {
  int64_t *d_tab, *d_out;
  CHK(hipMalloc(&d_tab, 4096)); CHK(hipMalloc(&d_out, 4096));
  CHK(hipMemset(d_tab, 1, 4096)); CHK(hipMemset(d_out, 2, 4096));
  for (int w : {1, 3, 8}) {
    run<0, 1>("F v_cmp_i64 e64 -> SGPR, v_addc e64", d_tab, d_out, w, 1);
    run<0, 4>("F", d_tab, d_out, w, 1);
    run<1, 1>("G v_cmp_i64 e32 -> vcc, v_addc e32", d_tab, d_out, w, 1);
    run<1, 4>("G", d_tab, d_out, w, 1);
    run<2, 1>("I v_cmp_u32 e64 -> SGPR, v_addc", d_tab, d_out, w, 1);
    run<2, 4>("I", d_tab, d_out, w, 1);
    run<3, 1>("P v_pk_min_u16, v_cmp_eq_u32, v_addc (2 lanes)", d_tab, d_out, w, 2);
    run<3, 2>("P (K = 2 accumulators = 4 lanes)", d_tab, d_out, w, 2);
  }
  return 0;
}
