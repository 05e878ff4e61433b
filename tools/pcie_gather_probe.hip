// How fast a kernel reads page-locked HOST memory over PCIe (what k_gather_frames of vo_step.hip does: 256 images of 1241 x 376 bytes per step),
// by workgroup count and loads in flight per lane, against one hipMemcpyAsync of the same bytes and against 256 copies of one image each.
//   hipcc --offload-arch=gfx950 -O3 -o tools/bin/pcie_gather_probe tools/pcie_gather_probe.hip && tools/bin/pcie_gather_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int U>
__global__ void __launch_bounds__(256) k_flat(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t n16) {
  const size_t T = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < n16; i0 += (size_t)U * T) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * T < n16) v[u] = src[i0 + u * T];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * T < n16) dst[i0 + u * T] = v[u];
  }
}
// a workgroup column per image (the first form of k_gather_frames): blockIdx.y = image, gridDim.x workgroups walk its chunks
template <int U>
__global__ void __launch_bounds__(256) k_per_image(const uint4* __restrict__ src, uint4* __restrict__ dst, size_t per16) {
  const uint4* s = src + (size_t)blockIdx.y * per16;
  uint4* d = dst + (size_t)blockIdx.y * per16;
  const size_t T = (size_t)gridDim.x * 256;
  for (size_t i0 = (size_t)blockIdx.x * 256 + threadIdx.x; i0 < per16; i0 += (size_t)U * T) {
    uint4 v[U];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * T < per16) v[u] = s[i0 + u * T];
#pragma unroll
    for (int u = 0; u < U; u++) if (i0 + u * T < per16) d[i0 + u * T] = v[u];
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
  const size_t per = (size_t)1241 * 376 / 16 * 16, n_img = 256, bytes = per * n_img, n16 = bytes / 16;
  uint4 *h = nullptr, *d = nullptr;
  CK(hipHostMalloc((void**)&h, bytes, hipHostMallocDefault));
  CK(hipMalloc((void**)&d, bytes));
  for (size_t i = 0; i < n16; i++) h[i] = make_uint4((uint32_t)i, 1, 2, 3);
  hipStream_t st; CK(hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  auto time = [&](const char* name, auto fn) {
    fn(); (void)hipStreamSynchronize(st);
    float best = 1e9f;
    for (int r = 0; r < 5; r++) {
      (void)hipEventRecord(e0, st); fn(); (void)hipEventRecord(e1, st); (void)hipStreamSynchronize(st);
      float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-56s %8.3f ms  %6.1f GB/s\n", name, best, bytes / best / 1e6);
  };
  printf("%zu images of %zu bytes = %.1f MB of page-locked host memory -> device\n", n_img, per, bytes / 1e6);
  time("hipMemcpyAsync, ONE copy", [&] { (void)hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, st); });
  time("hipMemcpyAsync, one copy per image (256)", [&] { for (size_t b = 0; b < n_img; b++) (void)hipMemcpyAsync((char*)d + b * per, (char*)h + b * per, per, hipMemcpyHostToDevice, st); });
  {
    std::vector<void*> dsts(n_img), srcs(n_img); std::vector<size_t> sizes(n_img, per);
    for (size_t b = 0; b < n_img; b++) { dsts[b] = (char*)d + b * per; srcs[b] = (char*)h + b * per; }
    hipMemcpyAttributes at = {}; at.srcAccessOrder = hipMemcpySrcAccessOrderStream; size_t idx0 = 0, fail = 0;
    hipError_t probe = hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), n_img, &at, &idx0, 1, &fail, st);
    (void)hipStreamSynchronize(st);
    if (probe == hipSuccess) time("hipMemcpyBatchAsync, 256 entries", [&] { (void)hipMemcpyBatchAsync(dsts.data(), srcs.data(), sizes.data(), n_img, &at, &idx0, 1, &fail, st); });
    else { printf("hipMemcpyBatchAsync -> %s\n", hipGetErrorString(probe)); (void)hipGetLastError(); }
    hipStream_t s4[4]; for (auto& x : s4) (void)hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    hipEvent_t ev[4]; for (auto& x : ev) (void)hipEventCreateWithFlags(&x, hipEventDisableTiming);
    time("hipMemcpyAsync, 256 copies dealt to 4 streams", [&] {
      for (size_t b = 0; b < n_img; b++) (void)hipMemcpyAsync((char*)d + b * per, (char*)h + b * per, per, hipMemcpyHostToDevice, s4[b & 3]);
      for (int k = 0; k < 4; k++) { (void)hipEventRecord(ev[k], s4[k]); (void)hipStreamWaitEvent(st, ev[k], 0); }
    });
  }
  char name[128];
  for (int g : {16, 32, 64, 128, 256, 512, 1024, 2048}) {
    snprintf(name, sizeof(name), "kernel, flat index, %4d workgroups x 256, 4 in flight", g);
    time(name, [&] { hipLaunchKernelGGL(k_flat<4>, dim3(g), dim3(256), 0, st, h, d, n16); });
    snprintf(name, sizeof(name), "kernel, flat index, %4d workgroups x 256, 8 in flight", g);
    time(name, [&] { hipLaunchKernelGGL(k_flat<8>, dim3(g), dim3(256), 0, st, h, d, n16); });
    snprintf(name, sizeof(name), "kernel, flat index, %4d workgroups x 256, 16 in flight", g);
    time(name, [&] { hipLaunchKernelGGL(k_flat<16>, dim3(g), dim3(256), 0, st, h, d, n16); });
  }
  for (int gx : {1, 2, 4}) {
    snprintf(name, sizeof(name), "kernel, %d workgroup(s) per image x 256 images, 4 in flight", gx);
    time(name, [&] { hipLaunchKernelGGL(k_per_image<4>, dim3(gx, n_img), dim3(256), 0, st, h, d, per / 16); });
  }
  return 0;
}
