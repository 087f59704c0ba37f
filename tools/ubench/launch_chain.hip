// Micro-benchmark (not part of the product): what a launch boundary, a dependent load and an in-kernel flag hand-over cost
// on this part.  The batch is a chain of small, latency-bound launches; these numbers are the floor it is measured against.
//   hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip && ./launch_chain
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_empty(uint32_t* sink) { if (threadIdx.x == 9999) *sink = 1; }

// a chain of `depth` dependent loads per thread (pointer chase through a shuffled table that lives in HBM / MALL)
__global__ void k_chase(const uint32_t* __restrict__ next, uint32_t* out, int depth, uint32_t n) {
  uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) * 977u % n;
  for (int d = 0; d < depth; ++d) i = next[i];
  if (i == 0xFFFFFFFFu) out[0] = i;
}

// producer / consumer inside ONE launch: blocks [0, np) write 16 KB each (write-through stores) and then bump a counter;
// the other blocks wait for the counter, then read what the producers wrote (plain loads: the lines were never cached here).
__global__ __launch_bounds__(256) void k_handover(unsigned long long* data, uint32_t* counter, uint32_t np, uint32_t seq, unsigned long long* out) {
  if (blockIdx.x < np) {
    unsigned long long* mine = data + (size_t)blockIdx.x * 2048;
    for (int j = 0; j < 8; ++j) __hip_atomic_store(&mine[j * 256 + threadIdx.x], (unsigned long long)seq * 1000003ull + j * 256 + threadIdx.x, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return;
  }
  if (threadIdx.x == 0) {
    while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < seq * np) __builtin_amdgcn_s_sleep(2);
  }
  __syncthreads();
  const uint32_t b = (blockIdx.x - np) % np;
  unsigned long long acc = 0, bad = 0;
  for (int j = 0; j < 8; ++j) {
    const unsigned long long v = data[(size_t)b * 2048 + j * 256 + threadIdx.x];
    acc += v;
    bad += v != (unsigned long long)seq * 1000003ull + j * 256 + threadIdx.x;
  }
  if (bad) atomicAdd(&out[1], bad);
  if (acc == 1) out[0] = acc;
}
// the same work as two launches
__global__ __launch_bounds__(256) void k_produce(unsigned long long* data, uint32_t seq) {
  unsigned long long* mine = data + (size_t)blockIdx.x * 2048;
  for (int j = 0; j < 8; ++j) mine[j * 256 + threadIdx.x] = (unsigned long long)seq * 1000003ull + j * 256 + threadIdx.x;
}
__global__ __launch_bounds__(256) void k_consume(const unsigned long long* data, uint32_t np, uint32_t seq, unsigned long long* out) {
  const uint32_t b = blockIdx.x % np;
  unsigned long long acc = 0, bad = 0;
  for (int j = 0; j < 8; ++j) {
    const unsigned long long v = data[(size_t)b * 2048 + j * 256 + threadIdx.x];
    acc += v;
    bad += v != (unsigned long long)seq * 1000003ull + j * 256 + threadIdx.x;
  }
  if (bad) atomicAdd(&out[1], bad);
  if (acc == 1) out[0] = acc;
}

template <class F> double per_iter_us(hipStream_t st, int iters, F f) {
  for (int i = 0; i < 20; ++i) f(i);
  (void)hipStreamSynchronize(st);
  const auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < iters; ++i) f(20 + i);
  (void)hipStreamSynchronize(st);
  return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / iters;
}

int main() {
  hipStream_t st;
  CHK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  uint32_t* sink; CHK(hipMalloc(&sink, 64));
  const uint32_t n = 1u << 22;                                  // 16 MB of links
  std::vector<uint32_t> h(n);
  uint64_t s = 88172645463325252ull;
  for (uint32_t i = 0; i < n; ++i) { s ^= s << 13; s ^= s >> 7; s ^= s << 17; h[i] = (uint32_t)(s % n); }
  uint32_t* next; CHK(hipMalloc(&next, (size_t)n * 4));
  CHK(hipMemcpy(next, h.data(), (size_t)n * 4, hipMemcpyHostToDevice));
  printf("empty kernel, back to back on one stream:        %.2f us per launch (1 block), %.2f us (1024 blocks)\n",
         per_iter_us(st, 2000, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(256), 0, st, sink); }),
         per_iter_us(st, 2000, [&](int) { hipLaunchKernelGGL(k_empty, dim3(1024), dim3(256), 0, st, sink); }));
  for (int depth : {1, 2, 3, 4, 6, 8}) {
    printf("40 blocks x 256 threads, %d dependent loads:       %.2f us per launch\n", depth,
           per_iter_us(st, 1000, [&](int) { hipLaunchKernelGGL(k_chase, dim3(40), dim3(256), 0, st, next, sink, depth, n); }));
  }
  const uint32_t np = 20, nc = 300;
  unsigned long long* data; CHK(hipMalloc(&data, (size_t)np * 2048 * 8));
  unsigned long long* out; CHK(hipMalloc(&out, 16)); CHK(hipMemset(out, 0, 16));
  uint32_t* counter; CHK(hipMalloc(&counter, 64)); CHK(hipMemset(counter, 0, 64));
  uint32_t seq = 0;
  const double one = per_iter_us(st, 1000, [&](int) { ++seq; hipLaunchKernelGGL(k_handover, dim3(np + nc), dim3(256), 0, st, data, counter, np, seq, out); });
  unsigned long long res[2];
  CHK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
  printf("20 producers -> counter -> 300 consumers, ONE launch: %.2f us (stale reads: %llu)\n", one, res[1]);
  CHK(hipMemset(out, 0, 16));
  uint32_t seq2 = 0;
  const double two = per_iter_us(st, 1000, [&](int) {
    ++seq2;
    hipLaunchKernelGGL(k_produce, dim3(np), dim3(256), 0, st, data, seq2);
    hipLaunchKernelGGL(k_consume, dim3(nc), dim3(256), 0, st, data, np, seq2, out);
  });
  CHK(hipMemcpy(res, out, 16, hipMemcpyDeviceToHost));
  printf("the same as two launches:                          %.2f us (stale reads: %llu)\n", two, res[1]);
  return 0;
}
