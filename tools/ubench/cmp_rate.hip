// Micro-benchmark (not part of the product): issue rate of the compare forms the scan kernel could use.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#define REP 64
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ __launch_bounds__(256) void k(const int64_t* __restrict__ tab, int64_t* out, int iters) {
  const int64_t r0 = out[threadIdx.x & 63], r1 = r0 + 1, r2 = r0 + 2, r3 = r0 + 3, r4 = r0 + 4;
  unsigned long long acc = 0;
  const int64_t a0 = tab[0], a1 = tab[1], a2 = tab[2], a3 = tab[3], a4 = tab[4];   // uniform -> SGPR
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int u = 0; u < REP; ++u) {
      if (MODE == 0) {        // 5 x v_cmpx_ge_i64 chain (what k_scan does)
        unsigned long long nf = ~0ull;
        uint32_t myk = 0;
        asm volatile("s_mov_b64 exec, %[nf]\n\tv_cmpx_ge_i64 vcc, %[a0], %[r0]\n\tv_cmpx_ge_i64 vcc, %[a1], %[r1]\n\tv_cmpx_ge_i64 vcc, %[a2], %[r2]\n\t"
                     "v_cmpx_ge_i64 vcc, %[a3], %[r3]\n\tv_cmpx_ge_i64 vcc, %[a4], %[r4]\n\tv_mov_b32 %[myk], %[k]\n\ts_andn2_b64 %[nf], %[nf], exec\n\ts_mov_b64 exec, -1"
                     : [nf] "+s"(nf), [myk] "+v"(myk) : [k] "s"(it), [a0] "s"(a0), [r0] "v"(r0), [a1] "s"(a1), [r1] "v"(r1), [a2] "s"(a2), [r2] "v"(r2),
                       [a3] "s"(a3), [r3] "v"(r3), [a4] "s"(a4), [r4] "v"(r4) : "vcc");
        acc += nf + myk;
      } else if (MODE == 1) { // 5 x v_cmp_ge_i64 (no exec), results discarded into vcc
        asm volatile("v_cmp_ge_i64 vcc, %[a0], %[r0]\n\tv_cmp_ge_i64 vcc, %[a1], %[r1]\n\tv_cmp_ge_i64 vcc, %[a2], %[r2]\n\tv_cmp_ge_i64 vcc, %[a3], %[r3]\n\tv_cmp_ge_i64 vcc, %[a4], %[r4]"
                     :: [a0] "s"(a0), [r0] "v"(r0), [a1] "s"(a1), [r1] "v"(r1), [a2] "s"(a2), [r2] "v"(r2), [a3] "s"(a3), [r3] "v"(r3), [a4] "s"(a4), [r4] "v"(r4) : "vcc");
      } else if (MODE == 2) { // 5 x v_cmp_ge_u32
        asm volatile("v_cmp_ge_u32 vcc, %[a0], %[r0]\n\tv_cmp_ge_u32 vcc, %[a1], %[r1]\n\tv_cmp_ge_u32 vcc, %[a2], %[r2]\n\tv_cmp_ge_u32 vcc, %[a3], %[r3]\n\tv_cmp_ge_u32 vcc, %[a4], %[r4]"
                     :: [a0] "s"((uint32_t)a0), [r0] "v"((uint32_t)r0), [a1] "s"((uint32_t)a1), [r1] "v"((uint32_t)r1), [a2] "s"((uint32_t)a2), [r2] "v"((uint32_t)r2),
                        [a3] "s"((uint32_t)a3), [r3] "v"((uint32_t)r3), [a4] "s"((uint32_t)a4), [r4] "v"((uint32_t)r4) : "vcc");
      } else if (MODE == 3) { // 5 x v_cmp_ge_u64
        asm volatile("v_cmp_ge_u64 vcc, %[a0], %[r0]\n\tv_cmp_ge_u64 vcc, %[a1], %[r1]\n\tv_cmp_ge_u64 vcc, %[a2], %[r2]\n\tv_cmp_ge_u64 vcc, %[a3], %[r3]\n\tv_cmp_ge_u64 vcc, %[a4], %[r4]"
                     :: [a0] "s"(a0), [r0] "v"(r0), [a1] "s"(a1), [r1] "v"(r1), [a2] "s"(a2), [r2] "v"(r2), [a3] "s"(a3), [r3] "v"(r3), [a4] "s"(a4), [r4] "v"(r4) : "vcc");
      } else if (MODE == 4) { // 5 x v_cmp_ge_f64 (DP compare)
        asm volatile("v_cmp_ge_f64 vcc, %[a0], %[r0]\n\tv_cmp_ge_f64 vcc, %[a1], %[r1]\n\tv_cmp_ge_f64 vcc, %[a2], %[r2]\n\tv_cmp_ge_f64 vcc, %[a3], %[r3]\n\tv_cmp_ge_f64 vcc, %[a4], %[r4]"
                     :: [a0] "s"(a0), [r0] "v"(r0), [a1] "s"(a1), [r1] "v"(r1), [a2] "s"(a2), [r2] "v"(r2), [a3] "s"(a3), [r3] "v"(r3), [a4] "s"(a4), [r4] "v"(r4) : "vcc");
      } else if (MODE == 5) { // 5 x (v_sub_co_u32 + v_subb_co_u32): 64-bit unsigned compare from 32-bit ops
        uint32_t t0, t1;
        asm volatile("v_sub_co_u32 %[t0], vcc, %[a0], %[r0]\n\tv_subb_co_u32 %[t1], vcc, %[a1], %[r1], vcc\n\t"
                     "v_sub_co_u32 %[t0], vcc, %[a0], %[r2]\n\tv_subb_co_u32 %[t1], vcc, %[a1], %[r3], vcc\n\t"
                     "v_sub_co_u32 %[t0], vcc, %[a0], %[r0]\n\tv_subb_co_u32 %[t1], vcc, %[a1], %[r1], vcc\n\t"
                     "v_sub_co_u32 %[t0], vcc, %[a0], %[r2]\n\tv_subb_co_u32 %[t1], vcc, %[a1], %[r3], vcc\n\t"
                     "v_sub_co_u32 %[t0], vcc, %[a0], %[r0]\n\tv_subb_co_u32 %[t1], vcc, %[a1], %[r1], vcc"
                     : [t0] "=&v"(t0), [t1] "=&v"(t1) : [a0] "v"((uint32_t)a0), [a1] "v"((uint32_t)a1), [r0] "v"((uint32_t)r0), [r1] "v"((uint32_t)r1), [r2] "v"((uint32_t)r2), [r3] "v"((uint32_t)r3) : "vcc");
        acc += t0 + t1;
      } else if (MODE == 7) { // independent compares (no EXEC chain) + scalar ANDs + one exec write for the v_mov
        unsigned long long nf = ~0ull, m0, m1, m2, m3, m4;
        uint32_t myk = 0;
        asm volatile("v_cmp_ge_i64 %[m0], %[a0], %[r0]\n\tv_cmp_ge_i64 %[m1], %[a1], %[r1]\n\tv_cmp_ge_i64 %[m2], %[a2], %[r2]\n\t"
                     "v_cmp_ge_i64 %[m3], %[a3], %[r3]\n\tv_cmp_ge_i64 %[m4], %[a4], %[r4]\n\t"
                     "s_and_b64 %[m0], %[m0], %[m1]\n\ts_and_b64 %[m2], %[m2], %[m3]\n\ts_and_b64 %[m0], %[m0], %[m4]\n\ts_and_b64 %[m0], %[m0], %[m2]\n\t"
                     "s_and_b64 exec, %[m0], %[nf]\n\tv_mov_b32 %[myk], %[k]\n\ts_andn2_b64 %[nf], %[nf], exec\n\ts_mov_b64 exec, -1"
                     : [nf] "+s"(nf), [myk] "+v"(myk), [m0] "=&s"(m0), [m1] "=&s"(m1), [m2] "=&s"(m2), [m3] "=&s"(m3), [m4] "=&s"(m4)
                     : [k] "s"(it), [a0] "s"(a0), [r0] "v"(r0), [a1] "s"(a1), [r1] "v"(r1), [a2] "s"(a2), [r2] "v"(r2),
                       [a3] "s"(a3), [r3] "v"(r3), [a4] "s"(a4), [r4] "v"(r4) : "vcc", "scc");
        acc += nf + myk;
      } else if (MODE == 6) { // SALU only: the 3 scalar ops of a block
        unsigned long long nf = ~0ull;
        asm volatile("s_mov_b64 %[nf], exec\n\ts_andn2_b64 %[nf], %[nf], exec\n\ts_mov_b64 exec, -1" : [nf] "+s"(nf));
        acc += nf;
      }
    }
  }
  if (acc == 0x123456789ull) out[threadIdx.x] = (int64_t)acc;
}

template <int MODE>
int run(const char* name, const int64_t* d_tab, int64_t* d_out, int ops_per_rep, int waves_per_simd = 8) {
  const int iters = 200, blocks = 256 * waves_per_simd;   // n blocks x 4 waves per CU = n waves per SIMD
  hipEvent_t a, b;
  CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE>), dim3(blocks), dim3(256), 0, 0, d_tab, d_out, 2);
  CHK(hipDeviceSynchronize());
  CHK(hipEventRecord(a));
  hipLaunchKernelGGL(HIP_KERNEL_NAME(k<MODE>), dim3(blocks), dim3(256), 0, 0, d_tab, d_out, iters);
  CHK(hipEventRecord(b));
  CHK(hipEventSynchronize(b));
  float ms = 0;
  CHK(hipEventElapsedTime(&ms, a, b));
  const double wps = blocks * 4.0 / 1024.0;
  const double wave_instr_per_simd = wps * iters * (double)REP * ops_per_rep;
  const double cycles = ms * 1e-3 * 2.4e9;
  printf("[%d waves/SIMD] %-34s %8.3f ms  -> %6.2f cycles per wave-instruction per SIMD (at 2.4 GHz)\n", waves_per_simd, name, ms, cycles / wave_instr_per_simd);
  return 0;
}

int main() {
  int64_t *d_tab, *d_out;
  CHK(hipMalloc(&d_tab, 4096)); CHK(hipMalloc(&d_out, 4096));
  CHK(hipMemset(d_tab, 1, 4096)); CHK(hipMemset(d_out, 2, 4096));
  run<0>("v_cmpx_ge_i64 x5 + mov + 3 salu", d_tab, d_out, 6);
  run<1>("v_cmp_ge_i64 x5", d_tab, d_out, 5);
  run<2>("v_cmp_ge_u32 x5", d_tab, d_out, 5);
  run<3>("v_cmp_ge_u64 x5", d_tab, d_out, 5);
  run<4>("v_cmp_ge_f64 x5", d_tab, d_out, 5);
  run<5>("v_sub_co/v_subb_co x5 pairs", d_tab, d_out, 10);
  run<6>("3 salu", d_tab, d_out, 3);
  for (int w : {1, 2, 4, 8}) {
    run<0>("cmpx chain block (6 valu)", d_tab, d_out, 6, w);
    run<7>("independent cmp block (6 valu)", d_tab, d_out, 6, w);
    run<1>("v_cmp_ge_i64 x5 plain", d_tab, d_out, 5, w);
  }
  return 0;
}
