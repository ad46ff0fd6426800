// One allocation strategy per process (fresh driver state each time): seconds to get 64 GB of device memory usable.
//   mode 0: one hipMalloc          mode 1: 32 x 2 GB hipMalloc      mode 2: hipMallocAsync (default pool)
//   mode 3: hipMemCreate + hipMemMap (VMM), one 64 GB handle          mode 4: VMM, 32 handles of 2 GB into one range
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <time.h>
#include <vector>
static double now() { timespec t; clock_gettime(CLOCK_MONOTONIC, &t); return t.tv_sec + 1e-9 * t.tv_nsec; }
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s failed: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void touch(unsigned* p, size_t n) { for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i; }
int main(int argc, char** argv) {
  const int mode = argc > 1 ? atoi(argv[1]) : 0;
  const size_t GB = 1ull << 30, total = 64 * GB;
  CK(hipSetDevice(0));
  CK(hipFree(0));
  double t0 = now();
  void* base = nullptr;
  std::vector<void*> parts;
  if (mode == 0) { CK(hipMalloc(&base, total)); }
  else if (mode == 1) { for (int i = 0; i < 32; ++i) { void* p; CK(hipMalloc(&p, 2 * GB)); parts.push_back(p); } base = parts[0]; }
  else if (mode == 2) { hipStream_t s; CK(hipStreamCreate(&s)); CK(hipMallocAsync(&base, total, s)); CK(hipStreamSynchronize(s)); }
  else {
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = 0;
    size_t gran = 0;
    CK(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    printf("granularity %zu\n", gran);
    CK(hipMemAddressReserve(&base, total, gran, nullptr, 0));
    const int n = mode == 3 ? 1 : 32;
    const size_t each = total / n;
    hipMemAccessDesc acc = {};
    acc.location = prop.location;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    for (int i = 0; i < n; ++i) {
      hipMemGenericAllocationHandle_t h;
      CK(hipMemCreate(&h, each, &prop, 0));
      CK(hipMemMap((char*)base + i * each, each, 0, h, 0));
      CK(hipMemSetAccess((char*)base + i * each, each, &acc, 1));
    }
  }
  double t1 = now();
  if (mode == 1) { for (void* p : parts) hipLaunchKernelGGL(touch, dim3(4096), dim3(256), 0, 0, (unsigned*)p, 2 * GB / 4); }
  else hipLaunchKernelGGL(touch, dim3(8192), dim3(256), 0, 0, (unsigned*)base, total / 4);
  CK(hipDeviceSynchronize());
  double t2 = now();
  printf("mode %d: allocate %.3f s, touch all 64 GB %.3f s\n", mode, t1 - t0, t2 - t1);
  return 0;
}
