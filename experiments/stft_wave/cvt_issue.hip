// Micro-benchmark: what the conversions of stft_psd cost on gfx950 -- v_cvt_f64_i32 (16 a frame and thread: the PCM), the
// magic-number alternative (xor into the mantissa of 2^52 + 2^31, one fp64 subtraction), v_cvt_f32_f64 (8: the staged row),
// beside v_add_f64.   hipcc -O3 --offload-arch=gfx950 -o cvt_issue.bin cvt_issue.hip && ./cvt_issue.bin   (result: cvt_issue_mi355x.txt)
#include <hip/hip_runtime.h>
#include <cstdio>

template <int OP>
__global__ void k(double* out, int seed, int iters) {
  constexpr int ILP = 8;
  int xi[ILP];
  double acc[ILP];
#pragma unroll
  for (int i = 0; i < ILP; ++i) { xi[i] = seed + threadIdx.x * 3 + i; acc[i] = 0.0; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int r = 0; r < 8; ++r) {
#pragma unroll
      for (int i = 0; i < ILP; ++i) {
        if (OP == 0) {            // cvt_f64_i32 + add (the add keeps the result alive; counted separately by OP 3)
          acc[i] += (double)xi[i];
        } else if (OP == 1) {     // magic number: bits(2^52 + 2^31) ^ x as the low word, then - (2^52 + 2^31)
          const double m = __hiloint2double(0x43300000, xi[i] ^ (int)0x80000000);
          acc[i] += m - 4503601774854144.0;
        } else if (OP == 2) {     // cvt_f32_f64 (+ back, to keep a chain)
          acc[i] += (double)(float)acc[i];
        } else {
          acc[i] += 1.0;
        }
        xi[i] += 7;
      }
    }
  }
  double s = 0;
#pragma unroll
  for (int i = 0; i < ILP; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int OP>
static void run(const char* name) {
  hipDeviceProp_t p;
  hipGetDeviceProperties(&p, 0);
  const int grid = p.multiProcessorCount * 3, threads = 256;   // 3 waves per SIMD, as stft_psd
  double* out;
  hipMalloc(&out, (size_t)grid * threads * 8);
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(threads), 0, 0, out, 5, 10);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<OP>), dim3(grid), dim3(threads), 0, 0, out, 5, iters);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double groups = (double)iters * 8 * 8 * 3;   // (op group) per SIMD
  printf("%-28s %.3f ms  %.2f ns per group per SIMD (= %.2f clk @2.2GHz)\n", name, ms, ms * 1e6 / groups, ms * 1e6 / groups * 2.2);
  hipFree(out);
}

int main() {
  run<3>("add_f64 + int add");
  run<0>("cvt_f64_i32 + add + int add");
  run<1>("xor + sub_f64 + add + int add");
  run<2>("cvt_f32_f64 + cvt back + add");
  return 0;
}
