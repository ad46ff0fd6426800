// Micro-benchmark: LDS-pipe time per wave64 LDS ATOMIC instruction on gfx950, at conflict-free (lane-linear) and at
// pseudo-random addresses of a table (the access pattern of a hash table in LDS: shz_table.hip, vt_fold_kernel).
//   hipcc -O3 --offload-arch=gfx950 -o lds_atomics.bin lds_atomics.hip && ./lds_atomics.bin   (result: lds_atomics_mi355x.txt)
#include <hip/hip_runtime.h>
#include <cstdio>

enum { ADD32, ADD32_RTN, CAS32_RTN, ADD64, MAX64, RD32, WR32 };

template <int OP, bool RANDOM>
__global__ __launch_bounds__(1024) void k(unsigned* out, int iters) {
  __shared__ unsigned long long lds[8192];  // 64 KB
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = 0;
  __syncthreads();
  typedef __attribute__((address_space(3))) unsigned long long lds_t;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_t*)lds;
  const int lane = threadIdx.x & 63;
  constexpr int W = (OP == ADD64 || OP == MAX64) ? 8 : 4;
  uint32_t a[16];
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const uint32_t h = ((uint32_t)(threadIdx.x * 16 + r) * 2654435761u) >> 19;   // 13 bits
    a[r] = base + (RANDOM ? h * W : (uint32_t)(lane * W + r * 64 * W));
  }
  unsigned acc = 0;
  unsigned long long one = 1, big = threadIdx.x;
  unsigned cmp = 0xFFFFFFFFu, val = threadIdx.x;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (OP == ADD32) asm volatile("ds_add_u32 %0, %1" :: "v"(a[r]), "v"(val));
      if (OP == ADD32_RTN) { unsigned v; asm volatile("ds_add_rtn_u32 %0, %1, %2" : "=v"(v) : "v"(a[r]), "v"(val)); asm volatile("" :: "v"(v)); }
      if (OP == CAS32_RTN) { unsigned v; asm volatile("ds_cmpst_rtn_b32 %0, %1, %2, %3" : "=v"(v) : "v"(a[r]), "v"(cmp), "v"(val)); asm volatile("" :: "v"(v)); }
      if (OP == ADD64) asm volatile("ds_add_u64 %0, %1" :: "v"(a[r]), "v"(one));
      if (OP == MAX64) asm volatile("ds_max_u64 %0, %1" :: "v"(a[r]), "v"(big));
      if (OP == RD32) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a[r])); asm volatile("" :: "v"(v)); }
      if (OP == WR32) asm volatile("ds_write_b32 %0, %1" :: "v"(a[r]), "v"(val));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + (unsigned)lds[threadIdx.x];
}

template <int OP, bool RANDOM>
static void run(const char* name, int waves) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount, threads = waves * 64, iters = 5000;
  unsigned* out;
  hipMalloc(&out, (size_t)grid * threads * 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP, RANDOM>), dim3(grid), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP, RANDOM>), dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)waves * iters * 16;   // LDS instructions per CU
  const double ns = ms * 1e6 / n;
  printf("%-20s %-8s waves/CU %2d: %7.2f ns per wave instruction per CU = %6.1f clk at 2.1 GHz\n", name, RANDOM ? "random" : "linear", waves, ns, ns * 2.1);
  hipFree(out);
}

int main() {
  for (int w : {4, 16}) {
    run<RD32, false>("ds_read_b32", w);       run<RD32, true>("ds_read_b32", w);
    run<WR32, false>("ds_write_b32", w);      run<WR32, true>("ds_write_b32", w);
    run<ADD32, false>("ds_add_u32", w);       run<ADD32, true>("ds_add_u32", w);
    run<ADD32_RTN, false>("ds_add_rtn_u32", w); run<ADD32_RTN, true>("ds_add_rtn_u32", w);
    run<CAS32_RTN, false>("ds_cmpst_rtn_b32", w); run<CAS32_RTN, true>("ds_cmpst_rtn_b32", w);
    run<ADD64, false>("ds_add_u64", w);       run<ADD64, true>("ds_add_u64", w);
    run<MAX64, false>("ds_max_u64", w);       run<MAX64, true>("ds_max_u64", w);
    printf("\n");
  }
  return 0;
}
