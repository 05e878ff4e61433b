// Brute-force 2-nearest-neighbour descriptor matching (SURVEY.md 8f "next" row 4, the matching part of the bootstrap).
//
// Replaces  self._matcher.knnMatch(desc_1, desc_2, k=2)  with cv2.BFMatcher() (NORM_L2, no cross check) in
// Extractor.match, /root/reference/src/extractor/extractor.py:134-145 (the ratio test on the two distances stays in the
// Python adapter, as in the reference); called once per sequence through match_lists by Pipeline._get_init_state,
// pipeline.py:52.  Distance = sqrt of the sum of squared float32 differences (accumulated in float64, rounded to
// float32 once: independent of the summation order; OpenCV accumulates in float32 SIMD lanes -- unpinned at that
// level), neighbours ordered by (distance, train index), as OpenCV's batchDistance keeps the first of equals.
//
// GPU mapping: a wave per query descriptor; lane l scans train rows l, l + 64, ... with 16-byte loads (the query is
// wave-uniform -> scalar loads), keeps its two best, the 64 lane pairs are merged by shuffles.  HBM/L2 traffic is
// n1 x n2 x dim x 4 B (0.5 GB at 1000 x 1000 x 128): a one-off per sequence, not tiled further.
#include "vo_internal.h"

#include <math.h>

struct vo_match_ws {
  size_t cap1 = 0, cap2 = 0;       // floats per sequence
  int cap_q = 0;
  float* d_a = nullptr; float* d_b = nullptr;
  int32_t* d_idx = nullptr; float* d_dist = nullptr;
};

struct knn2 { float d0, d1; int i0, i1; };

__device__ __forceinline__ bool knn_less(float da, int ia, float db, int ib) { return da < db || (da == db && ia < ib); }

__device__ __forceinline__ void knn_insert(knn2& s, float d, int i) {
  if (knn_less(d, i, s.d0, s.i0)) { s.d1 = s.d0; s.i1 = s.i0; s.d0 = d; s.i0 = i; }
  else if (knn_less(d, i, s.d1, s.i1)) { s.d1 = d; s.i1 = i; }
}

// grid (ceil(n1 / 4), batch), 256 threads = 4 queries
__global__ void __launch_bounds__(256) k_match_knn2(const float* __restrict__ A, const float* __restrict__ Bm, int n1, int n2, int dim,
                                                    size_t seq_a, size_t seq_b, int32_t* __restrict__ idx, float* __restrict__ dist, int cap_q) {
  const int b = blockIdx.y, lane = threadIdx.x & 63;
  const int qi = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (qi >= n1) return;
  const float* q = A + (size_t)b * seq_a + (size_t)qi * dim;
  const float* T = Bm + (size_t)b * seq_b;
  knn2 s;
  s.d0 = s.d1 = __builtin_inff(); s.i0 = s.i1 = 0x7fffffff;
  for (int j = lane; j < n2; j += 64) {
    const float* r = T + (size_t)j * dim;
    double acc = 0.0;
    int k = 0;
    if ((dim & 3) == 0) {
      for (; k < dim; k += 4) {
        const float4 x = *reinterpret_cast<const float4*>(r + k);
        const float4 y = *reinterpret_cast<const float4*>(q + k);
        const double e0 = (double)y.x - (double)x.x, e1 = (double)y.y - (double)x.y, e2 = (double)y.z - (double)x.z, e3 = (double)y.w - (double)x.w;
        acc += e0 * e0; acc += e1 * e1; acc += e2 * e2; acc += e3 * e3;
      }
    }
    for (; k < dim; k++) { const double e = (double)q[k] - (double)r[k]; acc += e * e; }
    knn_insert(s, sqrtf((float)acc), j);
  }
  for (int o = 32; o > 0; o >>= 1) {
    const float od0 = __shfl_xor(s.d0, o), od1 = __shfl_xor(s.d1, o);
    const int oi0 = __shfl_xor(s.i0, o), oi1 = __shfl_xor(s.i1, o);
    knn_insert(s, od0, oi0);
    knn_insert(s, od1, oi1);
  }
  if (lane == 0) {
    int32_t* io = idx + ((size_t)b * cap_q + qi) * 2;
    float* dd = dist + ((size_t)b * cap_q + qi) * 2;
    io[0] = (s.i0 == 0x7fffffff) ? -1 : s.i0; io[1] = (s.i1 == 0x7fffffff) ? -1 : s.i1;
    dd[0] = s.d0; dd[1] = s.d1;
  }
}

void vo_match_destroy(vo_ctx* c) {
  if (!c->match) return;
  vo_match_ws* w = c->match;
  void* bufs[] = {w->d_a, w->d_b, w->d_idx, w->d_dist};
  for (void* p : bufs) if (p) (void)hipFree(p);
  delete w;
  c->match = nullptr;
}

// desc1 [batch][n1][dim], desc2 [batch][n2][dim] f32 -> idx [batch][n1][2] (train index of the nearest / second nearest,
// -1 if n2 < 2 leaves the slot empty), dist [batch][n1][2] f32 (inf for an empty slot).  A NaN anywhere in a train
// descriptor makes its distance NaN, which never beats a finite one.
extern "C" int32_t vo_match_knn2(vo_ctx* c, const float* desc1, int32_t n1, const float* desc2, int32_t n2, int32_t dim,
                                 int32_t* idx, float* dist) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, desc1 && desc2 && idx && dist, VO_E_INVALID, "null buffer");
  VO_CHECK(c, n1 >= 1 && n2 >= 1 && dim >= 1, VO_E_INVALID, "empty descriptor set");
  VO_HIP(c, hipSetDevice(c->device));
  const size_t B = c->batch;
  const size_t na = (size_t)n1 * dim, nb = (size_t)n2 * dim;
  if (c->match && (c->match->cap1 < na || c->match->cap2 < nb || c->match->cap_q < n1)) vo_match_destroy(c);
  if (!c->match) {
    vo_match_ws* w = new vo_match_ws();
    c->match = w;
    w->cap1 = na; w->cap2 = nb; w->cap_q = n1;
    VO_HIP(c, hipMalloc((void**)&w->d_a, sizeof(float) * na * B));
    VO_HIP(c, hipMalloc((void**)&w->d_b, sizeof(float) * nb * B));
    VO_HIP(c, hipMalloc((void**)&w->d_idx, sizeof(int32_t) * 2 * (size_t)n1 * B));
    VO_HIP(c, hipMalloc((void**)&w->d_dist, sizeof(float) * 2 * (size_t)n1 * B));
  }
  vo_match_ws* w = c->match;
  VO_HIP(c, hipMemcpy2DAsync(w->d_a, sizeof(float) * w->cap1, desc1, sizeof(float) * na, sizeof(float) * na, B, hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(w->d_b, sizeof(float) * w->cap2, desc2, sizeof(float) * nb, sizeof(float) * nb, B, hipMemcpyHostToDevice, c->stream));
  hipLaunchKernelGGL(k_match_knn2, dim3(vo_div_up(n1, 4), (unsigned)B), dim3(256), 0, c->stream, w->d_a, w->d_b, n1, n2, dim, w->cap1, w->cap2,
                     w->d_idx, w->d_dist, w->cap_q);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipMemcpy2DAsync(idx, sizeof(int32_t) * 2 * n1, w->d_idx, sizeof(int32_t) * 2 * w->cap_q, sizeof(int32_t) * 2 * n1, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipMemcpy2DAsync(dist, sizeof(float) * 2 * n1, w->d_dist, sizeof(float) * 2 * w->cap_q, sizeof(float) * 2 * n1, B, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}
