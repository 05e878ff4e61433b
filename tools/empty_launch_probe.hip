// What does an EARLY-EXITING launch cost on the stream?  The bundle adjustment enqueues its LM iterations blind (4 launches each); the
// iterations behind the one that converged load the problem's state and return.  This probe times chains of such launches as a function of
// the grid (1 ... 1024 workgroups), the workgroup size, dynamic LDS, and the size of the argument block, and the same chains when the flag
// they read was written by the previous kernel (as k_ba_build copies the state forward).
//   hipcc --offload-arch=gfx950 -O2 -o empty_launch_probe tools/empty_launch_probe.hip && ./empty_launch_probe
#include <hip/hip_runtime.h>
#include <cstdio>

struct big_args { const int* flag; int* out; double pad[36]; };   // ~300 bytes, like ba_ptrs

__global__ void k_exit_small(const int* flag, int* out) {
  if (flag[0]) return;
  out[blockIdx.x * blockDim.x + threadIdx.x] = 1;
}
__global__ void k_exit_big(big_args a) {
  extern __shared__ double dyn[];
  if (a.flag[0]) return;
  dyn[threadIdx.x] = a.pad[threadIdx.x & 31];
  a.out[blockIdx.x * blockDim.x + threadIdx.x] = (int)dyn[threadIdx.x ^ 1];
}
// copies the flag forward (parity slots) like k_ba_build does with the problem's state, then exits
__global__ void k_exit_forward(int* flags, int it, int* out) {
  if (flags[(it - 1) & 1]) { if (blockIdx.x == 0 && threadIdx.x == 0) flags[it & 1] = flags[(it - 1) & 1]; return; }
  out[blockIdx.x * blockDim.x + threadIdx.x] = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  hipStream_t st;
  CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  int *flag, *out;
  CK(hipMalloc(&flag, 64)); CK(hipMalloc(&out, 4 * 1024 * 1024));
  const int one[2] = {1, 1};
  CK(hipMemcpy(flag, one, 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int N = 400;
  printf("%-44s %8s %8s\n", "chain of 400 early-exiting launches, us each", "plain", "graph");
  // every chain twice: plain launches (the host issues 2.5 ... 4.5 us per launch: with nothing to run the chain is HOST-bound) and the same
  // chain captured once and replayed as a graph (no host work per launch: what the GPU itself needs per early-exiting kernel)
  auto run = [&](const char* name, auto launch) -> int {
    for (int i = 0; i < 20; i++) launch(i);
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < N; i++) launch(i + 1);
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(st, hipStreamCaptureModeThreadLocal));
    for (int i = 0; i < N; i++) launch(i + 1);
    CK(hipStreamEndCapture(st, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, st)); CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int k = 0; k < 5; k++) CK(hipGraphLaunch(ge, st));
    CK(hipEventRecord(e1, st));
    CK(hipStreamSynchronize(st));
    float msg; CK(hipEventElapsedTime(&msg, e0, e1));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
    printf("%-44s %8.2f %8.2f\n", name, ms * 1e3 / N, msg * 1e3 / (5 * N));
    return 0;
  };
  char name[128];
  for (int tpb : {256, 1024})
    for (int grid : {1, 2, 4, 8, 16, 45, 125, 1024}) {
      if (tpb == 1024 && grid > 125) continue;
      snprintf(name, sizeof name, "small args, grid %4d x %4d threads", grid, tpb);
      if (run(name, [&](int) { hipLaunchKernelGGL(k_exit_small, dim3(grid), dim3(tpb), 0, st, flag, out); })) return 1;
    }
  for (int grid : {1, 8, 125, 1024}) {
    big_args a{}; a.flag = flag; a.out = out;
    snprintf(name, sizeof name, "300-byte args + 32 KB LDS, grid %4d x 256", grid);
    if (run(name, [&](int) { hipLaunchKernelGGL(k_exit_big, dim3(grid), dim3(256), 32768, st, a); })) return 1;
    snprintf(name, sizeof name, "300-byte args, no LDS, grid %4d x 256", grid);
    if (run(name, [&](int) { hipLaunchKernelGGL(k_exit_big, dim3(grid), dim3(256), 0, st, a); })) return 1;
  }
  for (int grid : {1, 8, 125, 1024}) {
    snprintf(name, sizeof name, "flag copied forward, grid %4d x 256", grid);
    if (run(name, [&](int it) { hipLaunchKernelGGL(k_exit_forward, dim3(grid), dim3(256), 0, st, flag, it, out); })) return 1;
  }
  // a group as the BA enqueues it: 125 x 256 (build), 45 x 256 (reduce), 1 x 1024 (solve), 125 x 256 (update)
  if (run("BA-shaped group of 4 (per group)", [&](int it) {
        hipLaunchKernelGGL(k_exit_forward, dim3(125), dim3(256), 0, st, flag, it, out);
        hipLaunchKernelGGL(k_exit_small, dim3(45), dim3(256), 0, st, flag, out);
        hipLaunchKernelGGL(k_exit_small, dim3(1), dim3(1024), 0, st, flag, out);
        hipLaunchKernelGGL(k_exit_small, dim3(125), dim3(256), 0, st, flag, out); })) return 1;
  if (run("the same group as 4 single-workgroup launches", [&](int it) {
        hipLaunchKernelGGL(k_exit_forward, dim3(1), dim3(256), 0, st, flag, it, out);
        hipLaunchKernelGGL(k_exit_small, dim3(1), dim3(256), 0, st, flag, out);
        hipLaunchKernelGGL(k_exit_small, dim3(1), dim3(1024), 0, st, flag, out);
        hipLaunchKernelGGL(k_exit_small, dim3(1), dim3(256), 0, st, flag, out); })) return 1;
  return 0;
}
