// Does a float64 MFMA of one wave overlap with float64 (or float32) VALU work of ANOTHER wave on the same SIMD, or of the same wave?
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_overlap_probe.hip -o /tmp/mfma_overlap_probe
// One workgroup of 512 lanes per CU: waves 0..3 and 4..7 pair up on the four SIMDs.  role(wave) by mode:
//   0: all waves MFMA          1: all waves f64 FMA        2: waves 0..3 MFMA, 4..7 f64 FMA       3: waves 0..3 MFMA, 4..7 f32 FMA
//   4: every wave alternates 1 MFMA : 14 f64 FMA (same totals as mode 2 per SIMD)       5: all waves f32 FMA
//   6: all waves MFMA f32 16x16x4     7: waves 0..3 MFMA f32, 4..7 f64 FMA     8: every wave alternates 1 MFMA f32 : 7 f64 FMA
// Prints the wall time per launch and the SIMD cycles per (MFMA, 14 FMA) unit.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float fv4 __attribute__((ext_vector_type(4)));
#define NM 2000          // MFMAs per MFMA wave
#define VPM 14           // VALU instructions that take as long as one MFMA (64 cycles / 4.5)
__global__ void __launch_bounds__(512) k(int mode, double* out, unsigned long long* span) {
  const int wave = threadIdx.x >> 6;
  d4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
  double x = threadIdx.x * 1e-3, y = 1.0 + 1e-9 * threadIdx.x;
  double f0 = x, f1 = x + 1, f2 = x + 2, f3 = x + 3, f4 = x + 4, f5 = x + 5, f6 = x + 6;
  float g0 = (float)x, g1 = g0 + 1, g2 = g0 + 2, g3 = g0 + 3, g4 = g0 + 4, g5 = g0 + 5, g6 = g0 + 6, gy = (float)y;
  const bool mf = mode == 0 || ((mode == 2 || mode == 3) && wave < 4);
  const bool v64 = mode == 1 || (mode == 2 && wave >= 4);
  const bool v32 = mode == 5 || (mode == 3 && wave >= 4);
  fv4 b0 = {0, 0, 0, 0}, b1 = b0, b2 = b0, b3 = b0;
  const bool mf32 = mode == 6 || (mode == 7 && wave < 4);
  const bool v64b = mode == 7 && wave >= 4;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (mf) {
    for (int i = 0; i < NM; i += 4) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      a2 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a2, 0, 0, 0);
      a3 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a3, 0, 0, 0);
    }
  } else if (v64) {
    for (int i = 0; i < NM * VPM; i += 7) {
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
    }
  } else if (v32) {
    for (int i = 0; i < NM * VPM; i += 7) {
      g0 = fmaf(g0, gy, g1); g1 = fmaf(g1, gy, g2); g2 = fmaf(g2, gy, g3); g3 = fmaf(g3, gy, g4); g4 = fmaf(g4, gy, g5); g5 = fmaf(g5, gy, g6); g6 = fmaf(g6, gy, g0);
    }
  } else if (mf32) {
    for (int i = 0; i < NM; i += 4) {
      b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b0, 0, 0, 0);
      b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b1, 0, 0, 0);
      b2 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b2, 0, 0, 0);
      b3 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b3, 0, 0, 0);
    }
  } else if (v64b) {
    for (int i = 0; i < NM * VPM / 2; i += 7) {
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
    }
  } else if (mode == 8) {
    for (int i = 0; i < NM / 2; i += 2) {
      b0 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b0, 0, 0, 0);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
      b1 = __builtin_amdgcn_mfma_f32_16x16x4f32(gy, g0, b1, 0, 0, 0);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
    }
  } else if (mode == 4) {
    for (int i = 0; i < NM / 2; i += 2) {
      a0 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a0, 0, 0, 0);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
      a1 = __builtin_amdgcn_mfma_f64_16x16x4f64(x, y, a1, 0, 0, 0);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
      f0 = fma(f0, y, x); f1 = fma(f1, y, x); f2 = fma(f2, y, x); f3 = fma(f3, y, x); f4 = fma(f4, y, x); f5 = fma(f5, y, x); f6 = fma(f6, y, x);
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const double r = a0[0] + a1[1] + a2[2] + a3[3] + f0 + f1 + f2 + f3 + f4 + f5 + f6 + (double)(g0 + g1 + g2 + g3 + g4 + g5 + g6) + (double)(b0[0] + b1[1] + b2[2] + b3[3]);
  if (r == 12345.678) out[0] = r;
  if ((threadIdx.x & 63) == 0) span[blockIdx.x * 8 + wave] = t1 - t0;
}
int main() {
  double* out; unsigned long long* span;
  hipMalloc(&out, 8); hipMalloc(&span, 256 * 8 * 8);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[9] = {"all MFMA f64", "all FMA f64", "MFMA f64 | FMA f64 (wave pairs)", "MFMA f64 | FMA f32 (wave pairs)", "1 MFMA : 14 FMA f64 inside every wave", "all FMA f32", "all MFMA f32 16x16x4", "MFMA f32 | FMA f64 (wave pairs; 7 FMA per MFMA)", "1 MFMA f32 : 7 FMA f64 inside every wave"};
  for (int mode = 0; mode < 9; mode++) {
    hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, out, span);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int r = 0; r < 5; r++) hipLaunchKernelGGL(k, dim3(256), dim3(512), 0, 0, mode, out, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[8]; hipMemcpy(h, span, sizeof(h), hipMemcpyDeviceToHost);
    printf("mode %d %-40s %8.1f us per launch; wave spans (memtime ticks) %llu %llu | %llu %llu\n", mode, names[mode], ms * 1000 / 5, h[0], h[1], h[4], h[5]);
  }
  printf("modes 6-8 per SIMD: 6: %d MFMA f32; 7: %d MFMA f32 + %d FMA f64; 8: %d MFMA f32 + %d FMA f64\n", 2 * NM, NM, NM * VPM / 2, NM, NM * 7);
  printf("units per SIMD: mode 0: %d MFMA; mode 1: %d FMA; modes 2, 3: %d MFMA + %d FMA; mode 4: %d MFMA + %d FMA\n", 2 * NM, 2 * NM * VPM, NM, NM * VPM, NM, NM * VPM);
  return 0;
}
