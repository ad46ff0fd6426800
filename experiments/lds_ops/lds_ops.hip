// Micro-benchmark: LDS-pipe time per wave64 LDS instruction on gfx950, by instruction width, conflict-free addressing.
// One workgroup of WAVES x 64 threads per CU issues `iters` x 16 LDS instructions per wave back to back; time per
// instruction per CU = kernel time / (WAVES x iters x 16).  With 4+ waves per CU the LDS pipe, not the issue latency of
// one wave, is what is measured.
//   hipcc -O3 --offload-arch=gfx950 -o lds_ops.bin lds_ops.hip && ./lds_ops.bin      (result: lds_ops_mi355x.txt)
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double v2d __attribute__((ext_vector_type(2)));
enum { RD64, RD128, RD2_64, WR64, WR128, WR2_64, RD64_S16, WR64_S16, RD32, WR32 };

template <int OP>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
  __shared__ double lds[8192];  // 64 KB
  for (int i = threadIdx.x; i < 8192; i += blockDim.x) lds[i] = i;
  __syncthreads();
  typedef __attribute__((address_space(3))) double lds_d;
  const uint32_t base = (uint32_t)(uintptr_t)(lds_d*)lds;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // each wave works in its own 4 KB window; lane stride = access width (conflict-free) unless *_S16
  uint32_t a = base + wave * 4096;
  if (OP == RD64 || OP == WR64) a += lane * 8;
  if (OP == RD128 || OP == WR128 || OP == RD2_64 || OP == WR2_64) a += lane * 16;
  if (OP == RD64_S16 || OP == WR64_S16) a += lane * 16;   // 8-byte access at 16-byte stride (AoS real parts)
  if (OP == RD32 || OP == WR32) a += lane * 4;
  double x0 = lane, x1 = lane + 1;
  double acc = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (OP == RD64 || OP == RD64_S16) { double v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(a)); acc += 0; asm volatile("" :: "v"(v)); }
      if (OP == RD32) { float v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
      if (OP == RD128) { v2d v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
      if (OP == RD2_64) { v2d v; asm volatile("ds_read2_b64 %0, %1 offset1:1" : "=v"(v) : "v"(a)); asm volatile("" :: "v"(v)); }
      if (OP == WR64 || OP == WR64_S16) asm volatile("ds_write_b64 %0, %1" :: "v"(a), "v"(x0));
      if (OP == WR32) { float f = (float)x0; asm volatile("ds_write_b32 %0, %1" :: "v"(a), "v"(f)); }
      if (OP == WR128) { v2d v = {x0, x1}; asm volatile("ds_write_b128 %0, %1" :: "v"(a), "v"(v)); }
      if (OP == WR2_64) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" :: "v"(a), "v"(x0), "v"(x1));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  out[blockIdx.x * blockDim.x + threadIdx.x] = acc + lds[threadIdx.x];
}

template <int OP>
static void run(const char* name, int waves, int bytes_per_lane) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount, threads = waves * 64, iters = 20000;
  double* out;
  hipMalloc(&out, (size_t)grid * threads * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double n = (double)waves * iters * 16;   // LDS instructions per CU
  const double ns = ms * 1e6 / n;
  printf("%-28s waves/CU %2d: %6.2f ns per instruction per CU = %5.1f clk at 2.1 GHz, %6.1f B/clk/CU\n", name, waves, ns, ns * 2.1,
         64.0 * bytes_per_lane / (ns * 2.1));
  hipFree(out);
}

int main() {
  for (int w : {4, 8, 16}) {
    run<RD32>("ds_read_b32", w, 4);
    run<RD64>("ds_read_b64", w, 8);
    run<RD64_S16>("ds_read_b64 stride 16 B", w, 8);
    run<RD128>("ds_read_b128", w, 16);
    run<RD2_64>("ds_read2_b64 (adjacent)", w, 16);
    run<WR32>("ds_write_b32", w, 4);
    run<WR64>("ds_write_b64", w, 8);
    run<WR64_S16>("ds_write_b64 stride 16 B", w, 8);
    run<WR128>("ds_write_b128", w, 16);
    run<WR2_64>("ds_write2_b64 (adjacent)", w, 16);
    printf("\n");
  }
  return 0;
}
