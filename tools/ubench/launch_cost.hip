// Host cost of one kernel launch on this box: empty kernel, small and large (by-value struct) kernarg blocks,
// shallow (sync every 36) and deep (sync at the end) queues.   hipcc --offload-arch=gfx950 -O2 -o build/launch_cost
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
struct Big { long v[60]; };
__global__ void k_small(int* p) { if (p && threadIdx.x == 9999) *p = 1; }
__global__ void k_big(Big b) { if (b.v[0] == 12345 && threadIdx.x == 9999) *(int*)b.v[1] = 1; }
__global__ __launch_bounds__(512) void k_lds(Big b) {
  extern __shared__ char sm[];
  if (b.v[0] == 12345 && threadIdx.x == 9999) sm[0] = 1;
}
template <class F> static double run(F f, int n, int sync_every) {
  (void)hipDeviceSynchronize();
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < n; ++i) { f(); if (sync_every && (i + 1) % sync_every == 0) (void)hipStreamSynchronize(0); }
  auto t1 = std::chrono::steady_clock::now();
  (void)hipDeviceSynchronize();
  return std::chrono::duration<double, std::micro>(t1 - t0).count() / n;
}
int main() {
  Big b{}; hipStream_t s; (void)hipStreamCreate(&s);
  (void)hipFuncSetAttribute((const void*)k_lds, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int rep = 0; rep < 2; ++rep) {
    printf("small deep   %.2f us\n", run([&] { k_small<<<1, 64, 0, s>>>(nullptr); }, 20000, 0));
    printf("big   deep   %.2f us\n", run([&] { k_big<<<1, 64, 0, s>>>(b); }, 20000, 0));
    printf("big+err deep %.2f us\n", run([&] { k_big<<<1, 64, 0, s>>>(b); (void)hipGetLastError(); }, 20000, 0));
    printf("lds150k 256wg deep %.2f us\n", run([&] { k_lds<<<256, 512, 150 * 1024, s>>>(b); }, 20000, 0));
    printf("big shallow(36) %.2f us (includes the sync)\n", run([&] { k_big<<<1, 64, 0, s>>>(b); }, 3600, 36));
  }
  return 0;
}
