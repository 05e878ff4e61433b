// Diagnostic: cost of __syncthreads() and of an LDS compare-exchange pass for different workgroup sizes / LDS sizes.
#include <hip/hip_runtime.h>
#include <stdio.h>
template <int MODE>
__global__ void probe(unsigned long long* out, int iters, int n2) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(smem);
  const int tid = threadIdx.x;
  for (int i = tid; i < n2; i += blockDim.x) keys[i] = (unsigned long long)(i * 2654435761u);
  __syncthreads();
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (MODE == 0) {
    for (int it = 0; it < iters; it++) __syncthreads();
  } else if (MODE == 1) {
    for (int it = 0; it < iters; it++) {
      const int j = 1 << (it % 10);
      for (int i = tid; i < n2; i += blockDim.x) {
        const int l = i ^ j;
        if (l > i) { unsigned long long a = keys[i], b = keys[l]; if (a < b) { keys[i] = b; keys[l] = a; } }
      }
      __syncthreads();
    }
  } else {
    unsigned long long acc = 0;
    int q = tid;
    for (int it = 0; it < iters; it++) { q = (int)(keys[q & (n2 - 1)] >> 7) + tid; acc += q; }   // dependent LDS reads
    keys[tid] = acc;
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (tid == 0) out[0] = t1 - t0;
}
int main() {
  unsigned long long* d; unsigned long long h;
  hipMalloc(&d, 8);
  int tpbs[] = {256, 512, 1024};
  size_t ldss[] = {16 * 1024, 64 * 1024, 148 * 1024};
  hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  hipFuncSetAttribute((const void*)probe<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
  for (int tpb : tpbs) for (size_t lds : ldss) {
    const int iters = 200;
    hipLaunchKernelGGL(probe<0>, dim3(1), dim3(tpb), lds, 0, d, iters, 2048); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    double c0 = (double)h / iters;
    hipLaunchKernelGGL(probe<1>, dim3(1), dim3(tpb), lds, 0, d, iters, 2048); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    double c1 = (double)h / iters;
    hipLaunchKernelGGL(probe<2>, dim3(1), dim3(tpb), lds, 0, d, iters, 2048); hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost);
    double c2 = (double)h / iters;
    printf("tpb %4d lds %6zu : barrier %.0f cyc, cmpxchg pass (2048 keys) %.0f cyc, dependent LDS read %.0f cyc\n", tpb, lds, c0, c1, c2);
  }
  return 0;
}
