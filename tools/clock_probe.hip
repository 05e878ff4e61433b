// Diagnostic: effective shader clock seen by short, sparsely launched kernels vs a long-running one.
// Build: hipcc --offload-arch=gfx950 -O3 tools/clock_probe.hip -o /tmp/clock_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <unistd.h>
#include <chrono>
__global__ void spin(unsigned long long* out, int iters) {
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  float a = threadIdx.x * 1e-3f, b = 1.0001f;
  for (int i = 0; i < iters; i++) { a = a * b + 0.5f; a = a * b - 0.5f; a = a * b + 0.25f; a = a * b - 0.25f; }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; out[2] = (unsigned long long)a; }
}
int main() {
  unsigned long long* d; unsigned long long h[3];
  hipMalloc(&d, 64);
  hipStream_t s; hipStreamCreate(&s);
  int configs[][3] = {{2000, 1, 0}, {2000, 1, 2000}, {2000, 256, 0}, {200000, 256, 0}, {2000000, 1024, 0}};
  for (auto& c : configs) {
    double mhz_sum = 0; double us_sum = 0; int reps = 20;
    for (int r = 0; r < reps; r++) {
      if (c[2]) usleep(c[2]);
      hipLaunchKernelGGL(spin, dim3(c[1]), dim3(256), 0, s, d, c[0]);
      hipStreamSynchronize(s);
      hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
      mhz_sum += (double)h[0] / ((double)h[1] / 100.0);   // memrealtime = 100 MHz
      us_sum += (double)h[1] / 100.0;
    }
    printf("iters %8d blocks %5d gap_us %5d : kernel %.1f us, shader clock %.0f MHz\n", c[0], c[1], c[2], us_sum / reps, mhz_sum / reps);
  }
  // back-to-back short kernels
  for (int r = 0; r < 2000; r++) hipLaunchKernelGGL(spin, dim3(64), dim3(256), 0, s, d, 2000);
  hipStreamSynchronize(s); hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
  printf("after 2000 back-to-back short kernels: %.1f us, %.0f MHz\n", (double)h[1] / 100.0, (double)h[0] / ((double)h[1] / 100.0));
  return 0;
}
