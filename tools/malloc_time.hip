// tools/malloc_time.hip -- what does hipMalloc / hipFree of tens of GB cost on this box?
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
int main() {
  hipFree(0);
  for (size_t gb : {1, 4, 16, 34, 34}) {
    void* p = nullptr;
    auto t0 = std::chrono::steady_clock::now();
    hipError_t e = hipMalloc(&p, gb << 30);
    auto t1 = std::chrono::steady_clock::now();
    hipMemsetAsync(p, 0, 64, 0);
    hipDeviceSynchronize();
    auto t2 = std::chrono::steady_clock::now();
    hipFree(p);
    auto t3 = std::chrono::steady_clock::now();
    printf("%2zu GB: hipMalloc %7.1f ms (%s), first touch %6.1f ms, hipFree %7.1f ms\n", gb,
           std::chrono::duration<double, std::milli>(t1 - t0).count(), hipGetErrorString(e),
           std::chrono::duration<double, std::milli>(t2 - t1).count(), std::chrono::duration<double, std::milli>(t3 - t2).count());
  }
  return 0;
}
