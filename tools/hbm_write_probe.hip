// What a streaming WRITE reaches on this chip (k_scharr_pyrdown / k_pad_level0 are write-dominated: 0.67 GB written, 0.15 GB read per 256-frame
// level-0 launch, 2.7-3.1 TB/s measured).  Variants: plain 16-byte stores, non-temporal stores, a 4:1 write:read mix like the Scharr pass.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/hbm_write_probe tools/hbm_write_probe.hip && tools/bin/hbm_write_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ void __launch_bounds__(256) k_write(uint4* __restrict__ dst, const uint32_t* __restrict__ src, size_t n16, int per_thread) {
  const size_t base = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
  for (int i = 0; i < per_thread; i++) {
    const size_t k = base + (size_t)i * 256;
    if (k >= n16) return;
    uint32_t s = (uint32_t)k;
    if (MODE == 2) s = src[k];                      // 4 bytes read per 16 written
    uint4 v; v.x = s; v.y = s + 1; v.z = s + 2; v.w = s + 3;
    if (MODE == 1) {
      __builtin_nontemporal_store(v.x, &dst[k].x); __builtin_nontemporal_store(v.y, &dst[k].y);
      __builtin_nontemporal_store(v.z, &dst[k].z); __builtin_nontemporal_store(v.w, &dst[k].w);
    } else dst[k] = v;
  }
}

__global__ void __launch_bounds__(256) k_read(const uint4* __restrict__ src, uint32_t* __restrict__ out, size_t n16, int per_thread) {
  const size_t base = ((size_t)blockIdx.x * per_thread) * 256 + threadIdx.x;
  uint32_t a = 0;
  for (int i = 0; i < per_thread; i++) {
    const size_t k = base + (size_t)i * 256;
    if (k < n16) { const uint4 v = src[k]; a += v.x ^ v.y ^ v.z ^ v.w; }
  }
  if (a == 0x12345678u) out[0] = a;
}

int main() {
  const size_t bytes = (size_t)1 << 30, n16 = bytes / 16;
  uint4* d; uint32_t* s;
  hipMalloc(&d, bytes); hipMalloc(&s, bytes / 4);
  hipMemset(d, 1, bytes); hipMemset(s, 1, bytes / 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int per_thread : {1, 4, 16}) {
    const unsigned grid = (unsigned)((n16 + (size_t)256 * per_thread - 1) / ((size_t)256 * per_thread));
    for (int mode = 0; mode < 4; mode++) {
      float best = 1e9f;
      for (int rep = 0; rep < 6; rep++) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_write<0>, dim3(grid), dim3(256), 0, 0, d, s, n16, per_thread);
        else if (mode == 1) hipLaunchKernelGGL(k_write<1>, dim3(grid), dim3(256), 0, 0, d, s, n16, per_thread);
        else if (mode == 2) hipLaunchKernelGGL(k_write<2>, dim3(grid), dim3(256), 0, 0, d, s, n16, per_thread);
        else hipLaunchKernelGGL(k_read, dim3(grid), dim3(256), 0, 0, d, s, n16, per_thread);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 0 && ms < best) best = ms;
      }
      const double moved = mode == 2 ? bytes * 1.25 : (double)bytes;
      printf("%-28s 16-byte accesses per thread %2d: %7.1f us  %6.2f TB/s\n",
             mode == 0 ? "write (plain stores)" : mode == 1 ? "write (non-temporal)" : mode == 2 ? "write 16 B + read 4 B" : "read", per_thread, best * 1e3, moved / best * 1e-9);
    }
  }
  return 0;
}
