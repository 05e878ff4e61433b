// Diagnostic: cost of ONE vector-memory load instruction on a CU's texture path (TA / TCP / TD) as a function of the lane ->
// address shape and of the bytes per lane, for the shapes k_klt_track uses or could use.  Answers "is the memory pipe priced per
// instruction, per quad of lanes, per cache line or per byte" (profiles/r02_pmc_klt_*.txt show TD busy 90 % in that kernel).
// Build: hipcc --offload-arch=gfx950 -O3 tools/vmem_probe.hip -o /tmp/vmem_probe ; run: /tmp/vmem_probe
// A body is 8 independent loads (one per window row step) + s_waitcnt inside a long loop; W waves per SIMD on all 256 CUs; every
// wave reads its own window of an L2-resident image (spread = 1) or all waves the same window (spread = 0: L1 hits only).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

#define PITCH 1312

template <int BYTES>
__device__ __forceinline__ unsigned ld(const unsigned char* p) {
  unsigned r;
  if (BYTES == 1) { asm volatile("global_load_ubyte %0, %1, off" : "=v"(r) : "v"(p) : "memory"); return r; }
  if (BYTES == 4) { asm volatile("global_load_dword %0, %1, off" : "=v"(r) : "v"(p) : "memory"); return r; }
  typedef unsigned u2 __attribute__((ext_vector_type(2)));
  typedef unsigned u3 __attribute__((ext_vector_type(3)));
  typedef unsigned u4 __attribute__((ext_vector_type(4)));
  if (BYTES == 8) { u2 v; asm volatile("global_load_dwordx2 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v.x ^ v.y; }
  if (BYTES == 12) { u3 v; asm volatile("global_load_dwordx3 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v.x ^ v.y ^ v.z; }
  u4 v; asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v) : "v"(p) : "memory"); return v.x ^ v.y ^ v.z ^ v.w;
}

// SHAPE: 0 lane*BYTES (coalesced), 1 KLT now: row 8r + s, 2-pixel stride (16 lanes per row), 2: rows 4r + s, 4-pixel stride
// (8 lanes per row), 3: rows (lane >> 3) + 8 s, 4-pixel stride, 4: rows 2r + s, 8-pixel stride (4 lanes per row)
// PX = bytes per pixel (1 image, 4 derivative pairs)
template <int SHAPE, int BYTES, int PX>
__global__ void __launch_bounds__(256) probe(const unsigned char* img, unsigned long long* out, int iters, int spread) {
  const int lane = threadIdx.x & 63, wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
  unsigned wx = 0, wy = 0;
  if (spread) { const unsigned h = wave * 2654435761u; wx = (h >> 8) % 900u; wy = (h >> 20) % 300u; }
  const unsigned char* base = img + ((size_t)(wy + 3) * PITCH + wx + 5) * PX;
  size_t off[8];
#pragma unroll
  for (int s = 0; s < 8; s++) {
    int row, col;
    if (SHAPE == 0) { row = s; col = lane * (BYTES / PX > 0 ? BYTES / PX : 1); }
    else if (SHAPE == 1) { row = 8 * (lane >> 4) + s; col = 2 * (lane & 15); }
    else if (SHAPE == 2) { row = 4 * (lane >> 3) + (s & 3); col = 4 * (lane & 7) + 40 * (s >> 2); }
    else if (SHAPE == 3) { row = (lane >> 3) + 8 * (s & 3); col = 4 * (lane & 7) + 40 * (s >> 2); }
    else { row = 2 * (lane >> 2) + (s & 1); col = 8 * (lane & 3) + 40 * (s >> 1); }
    off[s] = ((size_t)row * PITCH + col) * PX;
  }
  unsigned acc = 0;
  for (int i = 0; i < iters; i++) {
    unsigned v[8];
#pragma unroll
    for (int s = 0; s < 8; s++) v[s] = ld<BYTES>(base + off[s]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int s = 0; s < 8; s++) acc ^= v[s];
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
  if (acc == 0x12345u) out[7] = acc;
  if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = t1 - t0; out[1] = r1 - r0; }
}

struct test { const char* name; void (*k)(const unsigned char*, unsigned long long*, int, int); };

int main() {
  const size_t bytes = (size_t)PITCH * 400 * 4;
  unsigned char* img; unsigned long long* out;
  hipMalloc((void**)&img, bytes); hipMemset(img, 7, bytes);
  hipMalloc((void**)&out, 64); hipMemset(out, 0, 64);
  const test T[] = {
      {"dword   coalesced lane*4                     ", probe<0, 4, 1>},
      {"dwordx4 coalesced lane*16                    ", probe<0, 16, 1>},
      {"ubyte   img  16 lanes/row, 2 px stride (KLT) ", probe<1, 1, 1>},
      {"dword   img  16 lanes/row, 2 px stride (KLT) ", probe<1, 4, 1>},
      {"dwordx2 img   8 lanes/row, 4 px, rows 4r+s   ", probe<2, 8, 1>},
      {"dwordx2 img   8 lanes/row, 4 px, rows r+8s   ", probe<3, 8, 1>},
      {"dwordx3 img   4 lanes/row, 8 px, rows 2r+s   ", probe<4, 12, 1>},
      {"dwordx2 der  16 lanes/row, 2 px stride       ", probe<1, 8, 4>},
      {"dwordx3 der  16 lanes/row, 2 px stride (KLT) ", probe<1, 12, 4>},
      {"dwordx4 der   8 lanes/row, 4 px, rows 4r+s   ", probe<2, 16, 4>},
      {"dwordx4 der   8 lanes/row, 4 px, rows r+8s   ", probe<3, 16, 4>},
  };
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  printf("cycles per load instruction per CU (256 CUs), by waves per SIMD and window placement\n");
  printf("%-48s %10s %10s %10s %10s\n", "shape", "W=2 same", "W=5 same", "W=2 spread", "W=5 spread");
  for (const test& t : T) {
    printf("%s", t.name);
    for (int spread = 0; spread < 2; spread++)
      for (int W : {2, 5}) {
        const int iters = 2000;
        // W waves per SIMD = 4W waves per CU = W workgroups of 256 lanes per CU
        hipLaunchKernelGGL(t.k, dim3(256 * W), dim3(256), 0, 0, img, out, 50, spread);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        hipLaunchKernelGGL(t.k, dim3(256 * W), dim3(256), 0, 0, img, out, iters, spread);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2]; hipMemcpy(h, out, 16, hipMemcpyDeviceToHost);
        const double clk = (double)h[0] / ((double)h[1] / 100e6);          // shader Hz (s_memrealtime ticks at 100 MHz)
        const double cyc = ms * 1e-3 * clk / ((double)iters * 8 * 4 * W);  // per instruction per CU
        printf(" %10.1f", cyc);
      }
    printf("\n");
  }
  return 0;
}
