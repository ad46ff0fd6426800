// What does a large hipMalloc cost on this box, and does it stall kernels of another host thread?
// (round 3: the 1M-song table build spent ~3 s in hipMalloc of fresh memory, 2.2 s of it in the first 32 GB call)
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <time.h>
#include <atomic>
#include <thread>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
__global__ void spin(unsigned* p, int n) { unsigned x = threadIdx.x; for (int i = 0; i < n; ++i) x = x * 1664525u + 1013904223u; if (x == 7u) *p = x; }
int main(int argc, char** argv) {
  size_t fr, tot; hipMemGetInfo(&fr, &tot);
  printf("free %.1f GB of %.1f GB\n", fr / 1e9, tot / 1e9);
  const size_t sizes[] = {8, 32, 64, 128, 64, 32, 128};
  for (size_t gb : sizes) {
    void* p = nullptr;
    double t0 = now();
    hipError_t e = hipMalloc(&p, gb << 30);
    double t1 = now();
    if (e != hipSuccess) { printf("%zu GB: alloc failed\n", gb); (void)hipGetLastError(); continue; }
    hipMemset(p, 1, 1 << 20); hipDeviceSynchronize();
    double t2 = now();
    hipFree(p);
    double t3 = now();
    printf("%3zu GB: alloc %.4f s, first touch %.4f s, free %.4f s\n", gb, t1 - t0, t2 - t1, t3 - t2);
  }
  // two allocations held at once (fresh VA each), then a third
  {
    void *a = nullptr, *b = nullptr;
    double t0 = now(); hipMalloc(&a, 100ull << 30); double t1 = now(); hipMalloc(&b, 100ull << 30); double t2 = now();
    printf("100 GB + 100 GB held: %.4f s, %.4f s\n", t1 - t0, t2 - t1);
    hipFree(a); hipFree(b);
  }
  // does a big allocation on thread B stall launches + syncs of thread A?
  unsigned* d; hipMalloc(&d, 4);
  hipStream_t s; hipStreamCreate(&s);
  std::atomic<int> stop{0};
  double worst = 0, sum = 0; long iters = 0;
  std::thread A([&] {
    hipSetDevice(0);
    while (!stop.load()) {
      double t0 = now();
      hipLaunchKernelGGL(spin, dim3(256), dim3(256), 0, s, d, 2000);
      hipStreamSynchronize(s);
      double dt = now() - t0;
      worst = dt > worst ? dt : worst; sum += dt; ++iters;
    }
  });
  std::this_thread::sleep_for(std::chrono::milliseconds(200));
  double base_avg = sum / (iters ? iters : 1), base_worst = worst;
  worst = 0;
  void* big = nullptr;
  double t0 = now();
  hipError_t e = hipMalloc(&big, 160ull << 30);
  double t1 = now();
  std::this_thread::sleep_for(std::chrono::milliseconds(50));
  stop = 1; A.join();
  printf("concurrent: 160 GB alloc %s in %.4f s; kernel+sync of the other thread: avg %.1f us before (worst %.1f us), worst during %.1f us\n",
         e == hipSuccess ? "ok" : "FAILED", t1 - t0, base_avg * 1e6, base_worst * 1e6, worst * 1e6);
  if (big) hipFree(big);
  return 0;
}
