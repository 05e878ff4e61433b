// Why does hipGraph replay of the per-frame step lose to plain launches on this ROCm (7.2)?  (round-2 review item 6.)
// A chain of N dependent small kernels -- the shape of a single-sequence frame: ~45 launches of 3...60 us -- issued (a) as plain
// launches on one stream, the host running ahead, (b) as ONE hipGraphLaunch of the captured chain, (c) plain launches with the host
// synchronising every chain (no run-ahead).  Reports the GPU-side time per chain (hipEvents around K chains) and the host time to
// issue a chain.   hipcc --offload-arch=gfx950 -O2 -o graph_probe tools/graph_probe.hip && ./graph_probe
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void k_spin(unsigned long long cycles, unsigned long long* sink) {
  const unsigned long long t0 = wall_clock64();                  // constant 100 MHz counter
  while (wall_clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 0 && blockIdx.x == 0) *sink = t0;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  unsigned long long* sink;
  CK(hipMalloc(&sink, 8));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 45, K = 200;
  // wall_clock64 ticks at 100 MHz: 100 ticks = 1 us
  for (double us : {0.0, 5.0}) {
    const unsigned long long cyc = (unsigned long long)(us * 100.0);
    for (int blocks : {1, 256}) {
      auto chain = [&]() { for (int i = 0; i < N; i++) hipLaunchKernelGGL(k_spin, dim3(blocks), dim3(64), 0, st, cyc, sink); };
      // warm
      chain(); CK(hipStreamSynchronize(st));
      // (a) plain, host runs ahead
      auto h0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < K; k++) chain();
      CK(hipEventRecord(e1, st));
      auto h1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(st));
      float ms_plain; CK(hipEventElapsedTime(&ms_plain, e0, e1));
      const double host_plain = std::chrono::duration<double, std::micro>(h1 - h0).count() / K;
      // (b) graph
      hipGraph_t g; hipGraphExec_t ge;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      chain();
      CK(hipStreamEndCapture(st, &g));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
      h0 = std::chrono::steady_clock::now();
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < K; k++) CK(hipGraphLaunch(ge, st));
      CK(hipEventRecord(e1, st));
      h1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(st));
      float ms_graph; CK(hipEventElapsedTime(&ms_graph, e0, e1));
      const double host_graph = std::chrono::duration<double, std::micro>(h1 - h0).count() / K;
      // (c) plain, one sync per chain
      h0 = std::chrono::steady_clock::now();
      for (int k = 0; k < K; k++) { chain(); CK(hipStreamSynchronize(st)); }
      h1 = std::chrono::steady_clock::now();
      const double wall_sync = std::chrono::duration<double, std::micro>(h1 - h0).count() / K;
      // (d) graph = the chain + ONE device-to-host copy node at its end (what a captured frame step carries: its result copy);
      // (e) the kernel-only graph followed by the same copy as a plain hipMemcpyAsync
      static unsigned char* dbuf = nullptr; static unsigned char* hbuf = nullptr;
      const size_t nbytes = 64 * 1024;
      if (!dbuf) { CK(hipMalloc(&dbuf, nbytes)); CK(hipHostMalloc(&hbuf, nbytes, hipHostMallocDefault)); }
      hipGraph_t g2; hipGraphExec_t ge2;
      CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
      chain();
      CK(hipMemcpyAsync(hbuf, dbuf, nbytes, hipMemcpyDeviceToHost, st));
      CK(hipStreamEndCapture(st, &g2));
      CK(hipGraphInstantiate(&ge2, g2, nullptr, nullptr, 0));
      CK(hipGraphLaunch(ge2, st)); CK(hipStreamSynchronize(st));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < K; k++) CK(hipGraphLaunch(ge2, st));
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms_gcopy; CK(hipEventElapsedTime(&ms_gcopy, e0, e1));
      CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < K; k++) { CK(hipGraphLaunch(ge, st)); CK(hipMemcpyAsync(hbuf, dbuf, nbytes, hipMemcpyDeviceToHost, st)); }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms_gplain; CK(hipEventElapsedTime(&ms_gplain, e0, e1));
      CK(hipEventRecord(e0, st));
      for (int k = 0; k < K; k++) { chain(); CK(hipMemcpyAsync(hbuf, dbuf, nbytes, hipMemcpyDeviceToHost, st)); }
      CK(hipEventRecord(e1, st));
      CK(hipStreamSynchronize(st));
      float ms_pcopy; CK(hipEventElapsedTime(&ms_pcopy, e0, e1));
      printf("   + a 64 KB device-to-host copy per chain: plain launches + async copy %7.1f us | graph WITH a copy node %7.1f us | kernel-only graph + async copy %7.1f us\n",
             ms_pcopy * 1e3 / K, ms_gcopy * 1e3 / K, ms_gplain * 1e3 / K);
      CK(hipGraphExecDestroy(ge2)); CK(hipGraphDestroy(g2));
      printf("kernel %5.1f us x %3d workgroups, chain of %d: plain %7.1f us per chain on the GPU (%5.2f per launch; host issues a chain in %6.1f us) | "
             "graph %7.1f us (%5.2f per node; host %5.1f us) | plain + sync per chain %7.1f us wall\n",
             us, blocks, N, ms_plain * 1e3 / K, ms_plain * 1e3 / K / N, host_plain, ms_graph * 1e3 / K, ms_graph * 1e3 / K / N, host_graph, wall_sync);
      CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    }
  }
  return 0;
}
