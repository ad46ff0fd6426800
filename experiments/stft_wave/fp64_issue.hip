// Micro-benchmark: cycles per wave64 fp64 VALU instruction on gfx950 as a function of the waves resident
// per SIMD and of the independent chains per wave (ILP).  Decides whether one wave per SIMD can keep the
// DP pipe full (it sizes the workgroup/wave organisation of stft_psd).
//   hipcc -O3 --offload-arch=gfx950 -o fp64_issue.bin fp64_issue.hip && ./fp64_issue.bin   (result: fp64_issue_mi355x.txt)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int ILP, int OP>
__global__ void k(double* out, double a, double b, int iters, long long* cyc) {
  double x[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) x[i] = a + threadIdx.x + i;
  long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (OP == 0) x[i] = __builtin_fma(x[i], a, b);
        else if (OP == 1) x[i] = x[i] + b;
        else x[i] = x[i] * a;
      }
    }
  }
  long long t1 = __builtin_readcyclecounter();
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += x[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int ILP, int OP>
static void run(int waves_per_simd, const char* name) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int threads = 256 * waves_per_simd > 1024 ? 1024 : 256 * waves_per_simd;
  const int blocks_per_cu = (256 * waves_per_simd) / threads;
  const int grid = p.multiProcessorCount * blocks_per_cu;
  double* out;
  long long* cyc;
  hipMalloc(&out, (size_t)grid * threads * 8);
  hipMalloc(&cyc, 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<ILP, OP>), dim3(grid), dim3(threads), 0, 0, out, 1.0000001, 1e-9, 10, cyc);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<ILP, OP>), dim3(grid), dim3(threads), 0, 0, out, 1.0000001, 1e-9, iters, cyc);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  long long c;
  hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
  const double instr_per_wave = (double)iters * 8 * ILP;
  // shader clock from time: instructions per SIMD = waves_per_simd * instr_per_wave
  const double ns_per_instr = ms * 1e6 / (instr_per_wave * waves_per_simd);
  printf("%-4s ILP=%2d waves/SIMD=%d  %.3f ms  %.2f ns per wave-instr per SIMD (= %.2f clk @2.4GHz)  s_memtime ticks/instr=%.2f\n",
         name, ILP, waves_per_simd, ms, ns_per_instr, ns_per_instr * 2.4, (double)c / instr_per_wave);
  hipFree(out);
  hipFree(cyc);
}

int main() {
  for (int w : {1, 2, 3, 4}) {
    run<1, 0>(w, "fma");
    run<2, 0>(w, "fma");
    run<4, 0>(w, "fma");
    run<8, 0>(w, "fma");
    run<16, 0>(w, "fma");
    run<8, 1>(w, "add");
    run<8, 2>(w, "mul");
  }
  return 0;
}
