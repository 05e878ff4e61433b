// What does a dependency ACROSS two streams cost (hipEventRecord on one, hipStreamWaitEvent on the other) compared with the boundary between
// two kernels of ONE stream?  Chain: kernel on A -> kernel on B -> kernel on A -> ... (each kernel spins `us` microseconds), against the
// same kernels on one stream.  The closed loop's side stream (csrc/vo_pipeline.hip) pays two such hops per frame.
//   hipcc --offload-arch=gfx950 -O2 -o stream_hop_probe tools/stream_hop_probe.hip && ./stream_hop_probe
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void k_spin(unsigned long long ticks, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();                  // constant 100 MHz counter
  while (wall_clock64() - t0 < ticks) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t A, B;
  CK(hipStreamCreateWithFlags(&A, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&B, hipStreamNonBlocking));
  unsigned long long* sink; CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1, ab, ba;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 300;
  struct { const char* name; unsigned flags; } kinds[] = {
      {"hipEventDisableTiming", hipEventDisableTiming},
      {"hipEventDisableTiming | hipEventReleaseToDevice", hipEventDisableTiming | hipEventReleaseToDevice},
      {"hipEventDisableTiming | hipEventDisableSystemFence", hipEventDisableTiming | hipEventDisableSystemFence},
      {"hipEventDefault (timing)", hipEventDefault}};
  for (auto& kind : kinds) {
  printf("events created with %s\n", kind.name);
  CK(hipEventCreateWithFlags(&ab, kind.flags)); CK(hipEventCreateWithFlags(&ba, kind.flags));
  printf("%-10s %-8s %22s %22s %12s\n", "kernel us", "blocks", "one stream, us/pair", "two streams, us/pair", "per hop us");
  for (double us : {5.0, 20.0})
    for (int blocks : {1, 256}) {
      const unsigned long long ticks = (unsigned long long)(us * 100.0);
      auto one = [&]() { for (int i = 0; i < N; i++) { hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, A, ticks, sink); hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, A, ticks, sink); } };
      auto two = [&]() {
        for (int i = 0; i < N; i++) {
          hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, A, ticks, sink);
          (void)hipEventRecord(ab, A); (void)hipStreamWaitEvent(B, ab, 0);
          hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, B, ticks, sink);
          (void)hipEventRecord(ba, B); (void)hipStreamWaitEvent(A, ba, 0);
        }
      };
      float ms1, ms2;
      one(); CK(hipStreamSynchronize(A));
      CK(hipEventRecord(e0, A)); one(); CK(hipEventRecord(e1, A)); CK(hipStreamSynchronize(A)); CK(hipEventElapsedTime(&ms1, e0, e1));
      two(); CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B));
      CK(hipEventRecord(e0, A)); two(); CK(hipEventRecord(e1, A)); CK(hipStreamSynchronize(A)); CK(hipStreamSynchronize(B)); CK(hipEventElapsedTime(&ms2, e0, e1));
      printf("%-10.0f %-8d %22.2f %22.2f %12.2f\n", us, blocks, ms1 * 1e3 / N, ms2 * 1e3 / N, (ms2 - ms1) * 1e3 / N / 2);
    }
  CK(hipEventDestroy(ab)); CK(hipEventDestroy(ba));
  }
  return 0;
}
