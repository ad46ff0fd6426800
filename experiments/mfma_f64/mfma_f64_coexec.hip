// Micro-benchmark: does the fp64 matrix pipe of gfx950 run beside fp64 vector work of the same waves, and at what rate?
// (stft_psd is bound by fp64 VALU issue; its small DFTs could be matrix products on an otherwise idle pipe -- DESIGN.md 9.)
// Three kernels, `iters` x 16 instructions of each kind per wave, 4 / 8 waves per SIMD-set (256 / 512 threads per CU x 4):
//   valu   16 independent v_fma_f64 per iteration
//   mfma   16 independent v_mfma_f64_16x16x4_f64 per iteration (4 accumulators of 4 doubles)
//   both   the two interleaved one to one
//   hipcc -O3 --offload-arch=gfx950 -o mfma_f64_coexec.bin mfma_f64_coexec.hip && ./mfma_f64_coexec.bin
#include <hip/hip_runtime.h>
#include <cstdio>

typedef double d4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(1024) void k(double* out, int iters) {
  double a = threadIdx.x * 1e-3 + 1.0, b = 0.999;
  double v[8];
  d4 c[4];
#pragma unroll
  for (int i = 0; i < 8; ++i) v[i] = i + a;
#pragma unroll
  for (int i = 0; i < 4; ++i) c[i] = d4{a, a, a, a};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (MODE == 0 || MODE == 2) v[r & 7] = __builtin_fma(v[r & 7], b, a);
      if (MODE == 1 || MODE == 2) c[r & 3] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c[r & 3], 0, 0, 0);
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) s += v[i];
#pragma unroll
  for (int i = 0; i < 4; ++i) s += c[i][0] + c[i][1] + c[i][2] + c[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
static double run(int threads, int iters) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount;
  double* out;
  hipMalloc(&out, (size_t)grid * threads * 8);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), 0, 0, out, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<MODE>), dim3(grid), dim3(threads), 0, 0, out, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  hipFree(out);
  return ms;
}

int main() {
  const int iters = 20000;
  for (int threads : {256, 512, 1024}) {
    const double waves_per_simd = threads / 64 / 4.0;
    const double tv = run<0>(threads, iters), tm = run<1>(threads, iters), tb = run<2>(threads, iters);
    const double n = (double)iters * 16 * waves_per_simd;   // instructions of one kind per SIMD
    printf("%4d threads/CU (%.0f waves/SIMD): valu %.3f ms = %.1f clk/instr/SIMD | mfma %.3f ms = %.1f clk/instr/SIMD | both %.3f ms "
           "(sum %.3f, max %.3f)\n", threads, waves_per_simd, tv, tv * 2.4e6 / n, tm, tm * 2.4e6 / n, tb, tv + tm, tv > tm ? tv : tm);
  }
  return 0;
}
