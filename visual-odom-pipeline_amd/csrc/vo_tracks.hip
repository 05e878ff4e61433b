// Device-resident track table (SURVEY.md 8f "next" row 3): the bookkeeping the reference does in Python lists of
// Keypoint objects, kept as a structure of arrays in HBM so that a frame's tracking, pruning, history append,
// re-detection and the bundle adjuster's observation table never leave the GPU.
//
// One table per sequence of the context, an ORDERED dense list like the reference's Python lists (survivors keep
// their relative order, new detections are appended):
//   uv        the resident point set of the context (result slab, ping-pong)          Keypoint.uv
//   uv_first, t_first, t_total                                                         Keypoint.uv_first / t_first / t_total
//   tag       stable identity (what a host-side object list would be keyed by)
//   hist      ring of the last VO_TRK_HIST positions by ABSOLUTE frame index            Keypoint.uv_history (its tail)
// Semantics follow /root/reference/src/extractor/extractor.py:
//   vo_tracks_track    = extend_tracks / extend_landmarks (:38-88): KLT prev -> cur of every live track, keep iff
//                        0 <= x <= W and 0 <= y <= H (both ends inclusive; KLT status is ignored, the bidirectional
//                        test is disabled by max_bidir_error = inf: pipeline.py:98-100), survivors get uv, t_total + 1
//                        and a history entry, the others are reported dead;
//   vo_tracks_detect   = extract(..., 'shi-tomasi') (:90-132): exclusion discs at the live tracks, goodFeaturesToTrack,
//                        every corner becomes Keypoint(t_first = t, t_total = 1, uv_first = uv = corner, history = [corner]);
//   vo_tracks_obs      = the observation lookup of BundleAdjuster.adjust (bundle_adjuster.py:150-158, :56-59) for live
//                        tracks (t_latest = t_now): slot s of the window <-> frame t_now - s, present iff
//                        t_now - s >= t_first.
// The counts differ per sequence of a batch: they live on the device (c->d_pt_counts) and the KLT / disc kernels skip
// the points beyond them; the host only knows an upper bound.
#include "vo_internal.h"

#define VO_TRK_HIST 32          // ring depth (>= the largest BA window, 20)

struct vo_trk_ws {
  int cap = 0;                  // = max_pts
  int n_hi = 0;                 // host-side upper bound of the per-sequence counts
  int32_t* d_n = nullptr;       // [B] live tracks
  int32_t* d_next_tag = nullptr;// [B]
  int32_t* d_ndead = nullptr;   // [B] tracks that died in the last vo_tracks_track
  int32_t* d_dead_tag = nullptr;// [B][cap]
  float* d_first = nullptr;     // [B][cap][2]
  int32_t* d_tf = nullptr;      // [B][cap]
  int32_t* d_tt = nullptr;      // [B][cap]
  int32_t* d_tag = nullptr;     // [B][cap]
  float* d_hist = nullptr;      // [B][VO_TRK_HIST][cap][2]
  double* d_obs = nullptr;      // staging of vo_tracks_obs [B][window][cap][2]
  size_t obs_cap = 0;
};

void vo_trk_destroy(vo_ctx* c) {
  if (!c->trk) return;
  vo_trk_ws* t = c->trk;
  void* bufs[] = {t->d_n, t->d_next_tag, t->d_ndead, t->d_dead_tag, t->d_first, t->d_tf, t->d_tt, t->d_tag, t->d_hist, t->d_obs};
  for (void* p : bufs) if (p) (void)hipFree(p);
  delete t;
  c->trk = nullptr;
  c->d_pt_counts = nullptr;
}

static int32_t trk_init(vo_ctx* c) {
  if (c->trk) return VO_OK;
  vo_trk_ws* t = new vo_trk_ws();
  c->trk = t;
  t->cap = c->max_pts;
  const size_t B = c->batch, cap = t->cap;
  VO_HIP(c, hipMalloc((void**)&t->d_n, sizeof(int32_t) * B));
  VO_HIP(c, hipMalloc((void**)&t->d_next_tag, sizeof(int32_t) * B));
  VO_HIP(c, hipMalloc((void**)&t->d_ndead, sizeof(int32_t) * B));
  VO_HIP(c, hipMalloc((void**)&t->d_dead_tag, sizeof(int32_t) * B * cap));
  VO_HIP(c, hipMalloc((void**)&t->d_first, sizeof(float) * 2 * B * cap));
  VO_HIP(c, hipMalloc((void**)&t->d_tf, sizeof(int32_t) * B * cap));
  VO_HIP(c, hipMalloc((void**)&t->d_tt, sizeof(int32_t) * B * cap));
  VO_HIP(c, hipMalloc((void**)&t->d_tag, sizeof(int32_t) * B * cap));
  VO_HIP(c, hipMalloc((void**)&t->d_hist, sizeof(float) * 2 * B * cap * VO_TRK_HIST));
  VO_HIP(c, hipMemsetAsync(t->d_n, 0, sizeof(int32_t) * B, c->stream));
  VO_HIP(c, hipMemsetAsync(t->d_next_tag, 0, sizeof(int32_t) * B, c->stream));
  VO_HIP(c, hipMemsetAsync(t->d_ndead, 0, sizeof(int32_t) * B, c->stream));
  return VO_OK;
}

struct trk_ptrs {
  int32_t* n; int32_t* next_tag; int32_t* ndead; int32_t* dead_tag;
  float* first; int32_t* tf; int32_t* tt; int32_t* tag; float* hist;
  int cap;
};
static trk_ptrs trk_make(vo_trk_ws* t) {
  trk_ptrs P;
  P.n = t->d_n; P.next_tag = t->d_next_tag; P.ndead = t->d_ndead; P.dead_tag = t->d_dead_tag;
  P.first = t->d_first; P.tf = t->d_tf; P.tt = t->d_tt; P.tag = t->d_tag; P.hist = t->d_hist; P.cap = t->cap;
  return P;
}
__device__ __forceinline__ float2* trk_hist_row(const trk_ptrs& P, int b, int tframe) {
  return reinterpret_cast<float2*>(P.hist) + ((size_t)b * VO_TRK_HIST + (size_t)(tframe & (VO_TRK_HIST - 1))) * P.cap;
}

// ---- seed: n tracks per sequence from the resident points ----
__global__ void __launch_bounds__(256) k_trk_seed(trk_ptrs P, const float* __restrict__ pts, size_t slab_seq, int n, int t) {
  const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i == 0) { P.n[b] = n; P.next_tag[b] = n; P.ndead[b] = 0; }
  if (i >= n) return;
  const float2 p = reinterpret_cast<const float2*>(vo_seq(pts, slab_seq, b))[i];
  const size_t o = (size_t)b * P.cap + i;
  reinterpret_cast<float2*>(P.first)[o] = p;
  P.tf[o] = t; P.tt[o] = 1; P.tag[o] = i;
  const float2 nan2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
  for (int h = 0; h < VO_TRK_HIST; h++) trk_hist_row(P, b, h)[i] = (h == (t & (VO_TRK_HIST - 1))) ? p : nan2;
}

// block-wide exclusive scan of a 0/1 flag (1024 threads), returns the position and the total
__device__ __forceinline__ int trk_scan(int flag, int* s_w /* 16 */, int& total) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const unsigned long long bal = __ballot(flag);
  const int within = __popcll(bal & ((1ull << lane) - 1ull));
  __syncthreads();                       // s_w may still be read from the previous call
  if (lane == 0) s_w[wave] = __popcll(bal);
  __syncthreads();
  int off = 0, tot = 0;
#pragma unroll
  for (int w = 0; w < 16; w++) { const int v = s_w[w]; if (w < wave) off += v; tot += v; }
  total = tot;
  return off + within;
}

// ---- extend: keep rule, ordered in-place compaction of every column of the table, counters, history, dead list ----
// One workgroup per sequence walks the list in chunks of 1024: a chunk is read completely into registers before any of
// its survivors is written, and survivors only move towards the front, so nothing unread is overwritten.
__global__ void __launch_bounds__(1024) k_trk_extend(trk_ptrs P, float* __restrict__ pts, size_t slab_seq, int t, int W, int H) {
  __shared__ int s_w[16];
  const int b = blockIdx.x, tid = threadIdx.x;
  float2* p = reinterpret_cast<float2*>(vo_seq(pts, slab_seq, b));
  float2* first = reinterpret_cast<float2*>(P.first) + (size_t)b * P.cap;
  int32_t* tf = P.tf + (size_t)b * P.cap; int32_t* tt = P.tt + (size_t)b * P.cap; int32_t* tag = P.tag + (size_t)b * P.cap;
  int32_t* dead = P.dead_tag + (size_t)b * P.cap;
  const int n = P.n[b];
  int out_n = 0, out_dead = 0;
  for (int base = 0; base < n; base += 1024) {
    const int i = base + tid;
    const bool valid = i < n;
    float2 pi = make_float2(0.f, 0.f), fi = pi;
    int tfi = 0, tti = 0, tagi = 0;
    float2 hrow[VO_TRK_HIST];
    if (valid) {
      pi = p[i]; fi = first[i]; tfi = tf[i]; tti = tt[i]; tagi = tag[i];
#pragma unroll
      for (int h = 0; h < VO_TRK_HIST; h++) hrow[h] = trk_hist_row(P, b, h)[i];
    }
    // 0 <= x <= W and 0 <= y <= H, ends included; NaN fails like in Python
    const int keep = (valid && pi.x >= 0.f && pi.x <= (float)W && pi.y >= 0.f && pi.y <= (float)H) ? 1 : 0;
    int kept, died;
    const int pos = trk_scan(keep, s_w, kept);
    const int dpos = trk_scan(valid && !keep, s_w, died);
    __syncthreads();                     // the whole chunk is in registers
    if (keep) {
      const int o = out_n + pos;
      p[o] = pi; first[o] = fi; tf[o] = tfi; tt[o] = tti + 1; tag[o] = tagi;
#pragma unroll
      for (int h = 0; h < VO_TRK_HIST; h++) trk_hist_row(P, b, h)[o] = (h == (t & (VO_TRK_HIST - 1))) ? pi : hrow[h];
    } else if (valid) {
      dead[out_dead + dpos] = tagi;
    }
    out_n += kept; out_dead += died;
    __syncthreads();
  }
  if (tid == 0) { P.n[b] = out_n; P.ndead[b] = out_dead; }
}

// ---- spawn: append the corners of the last re-detection as new tracks ----
__global__ void __launch_bounds__(256) k_trk_spawn(trk_ptrs P, float* __restrict__ pts, size_t slab_seq, const uint32_t* __restrict__ st_scalars,
                                                   const float* __restrict__ st_out, int t, int max_new) {
  const int b = blockIdx.x, tid = threadIdx.x;
  float2* p = reinterpret_cast<float2*>(vo_seq(pts, slab_seq, b));
  const uint32_t* sc = vo_seq(st_scalars, slab_seq, b);
  const float2* corners = reinterpret_cast<const float2*>(vo_seq(st_out, slab_seq, b));
  const int n = P.n[b];
  const uint32_t nd = sc[2];
  int m = (nd == 0xFFFFFFFFu) ? 0 : (int)nd;
  if (m > max_new) m = max_new;
  if (m > P.cap - n) m = P.cap - n;
  const int tag0 = P.next_tag[b];
  const float2 nan2 = make_float2(__builtin_nanf(""), __builtin_nanf(""));
  for (int k = tid; k < m; k += 256) {
    const float2 q = corners[k];
    const size_t o = (size_t)b * P.cap + n + k;
    p[n + k] = q;
    reinterpret_cast<float2*>(P.first)[o] = q;
    P.tf[o] = t; P.tt[o] = 1; P.tag[o] = tag0 + k;
    for (int h = 0; h < VO_TRK_HIST; h++) trk_hist_row(P, b, h)[n + k] = (h == (t & (VO_TRK_HIST - 1))) ? q : nan2;
  }
  __syncthreads();
  if (tid == 0) { P.n[b] = n + m; P.next_tag[b] = tag0 + m; }
}

// ---- BA observation table of the live tracks: obs [window][cap][2] f64, NaN = not observed ----
__global__ void __launch_bounds__(256) k_trk_obs(trk_ptrs P, int t_now, int window, double* __restrict__ obs) {
  const int b = blockIdx.z, s = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
  if (i >= P.cap) return;
  double2 v = make_double2(__builtin_nan(""), __builtin_nan(""));
  if (i < P.n[b] && s < VO_TRK_HIST) {
    const int tau = t_now - s;
    if (tau >= P.tf[(size_t)b * P.cap + i]) {
      const float2 q = trk_hist_row(P, b, tau)[i];
      v = make_double2((double)q.x, (double)q.y);
    }
  }
  reinterpret_cast<double2*>(obs)[((size_t)b * window + s) * P.cap + i] = v;
}

// ================================================================================================
// host
// ================================================================================================
// pts [batch][n][2]: the initial tracks (e.g. the bootstrap's keypoints) at frame index t
extern "C" int32_t vo_tracks_seed(vo_ctx* c, const float* pts, int32_t n, int32_t t) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, n >= 0 && n <= c->max_pts && (n == 0 || pts), VO_E_CAPACITY, "bad point set");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  int32_t r = trk_init(c);
  if (r != VO_OK) return r;
  r = vo_points_upload(c, pts, n);
  if (r != VO_OK) return r;
  c->n_resident = c->max_pts;            // the list may grow up to the capacity
  vo_trk_ws* tw = c->trk;
  hipLaunchKernelGGL(k_trk_seed, dim3(vo_div_up(n > 0 ? n : 1, 256), c->batch), dim3(256), 0, c->stream, trk_make(tw),
                     vo_slab<const float>(c, vo_off_p(c)), c->slab_seq, n, t);
  VO_HIP(c, hipGetLastError());
  tw->n_hi = n;
  c->d_pt_counts = tw->d_n;
  return VO_OK;
}

// KLT prev -> cur of every live track (the frame store must hold both frames), then the reference's pruning and bookkeeping
extern "C" int32_t vo_tracks_track(vo_ctx* c, int32_t t, const vo_klt_params* prm) {
  if (!c) return VO_E_INVALID;
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  VO_CHECK(c, c->trk && c->d_pt_counts, VO_E_STATE, "vo_tracks_seed first");
  vo_trk_ws* tw = c->trk;
  if (tw->n_hi > 0) {
    const int32_t r = vo_klt_track_resident(c, tw->n_hi, prm);
    if (r != VO_OK) return r;
  }
  hipLaunchKernelGGL(k_trk_extend, dim3(c->batch), dim3(1024), 0, c->stream, trk_make(tw), vo_slab<float>(c, vo_off_p(c)),
                     c->slab_seq, t, c->width, c->height);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}

// Shi-Tomasi re-detection on the current frame with exclusion discs at the live tracks; up to max_new corners per
// sequence are appended as new tracks born at frame t
extern "C" int32_t vo_tracks_detect(vo_ctx* c, int32_t t, int32_t mask_radius, const vo_st_params* st, int32_t max_new) {
  if (!c) return VO_E_INVALID;
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  VO_CHECK(c, c->trk && c->d_pt_counts, VO_E_STATE, "vo_tracks_seed first");
  vo_trk_ws* tw = c->trk;
  vo_st_params def;
  if (!st) { vo_st_default_params(&def); st = &def; }
  const int32_t r = vo_shi_tomasi_resident(c, tw->n_hi, mask_radius, st);
  if (r != VO_OK) return r;
  if (max_new < 0) max_new = 0;
  hipLaunchKernelGGL(k_trk_spawn, dim3(c->batch), dim3(256), 0, c->stream, trk_make(tw), vo_slab<float>(c, vo_off_p(c)), c->slab_seq,
                     vo_slab<const uint32_t>(c, c->off_st_scalars), vo_slab<const float>(c, c->off_st_out), t, max_new);
  VO_HIP(c, hipGetLastError());
  int add = st->max_corners > 0 ? st->max_corners : 4096;
  if (add > max_new) add = max_new;
  tw->n_hi += add;
  if (tw->n_hi > tw->cap) tw->n_hi = tw->cap;
  return VO_OK;
}

// Synchronous read-back.  n, n_dead: [batch]; every other array [batch][max_pts] (uv / uv_first: x 2), valid up to n / n_dead.
// Any pointer may be NULL.
extern "C" int32_t vo_tracks_read(vo_ctx* c, int32_t* n, float* uv, float* uv_first, int32_t* t_first, int32_t* t_total,
                                  int32_t* tag, int32_t* n_dead, int32_t* dead_tag) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->trk, VO_E_STATE, "vo_tracks_seed first");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_trk_ws* tw = c->trk;
  const size_t B = c->batch, cap = tw->cap;
  VO_HIP(c, hipStreamSynchronize(c->stream));
  if (n) VO_HIP(c, hipMemcpy(n, tw->d_n, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  if (n_dead) VO_HIP(c, hipMemcpy(n_dead, tw->d_ndead, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  if (uv) VO_HIP(c, hipMemcpy2D(uv, sizeof(float) * 2 * cap, c->d_slab + vo_off_p(c), c->slab_seq, sizeof(float) * 2 * cap, B, hipMemcpyDeviceToHost));
  if (uv_first) VO_HIP(c, hipMemcpy(uv_first, tw->d_first, sizeof(float) * 2 * B * cap, hipMemcpyDeviceToHost));
  if (t_first) VO_HIP(c, hipMemcpy(t_first, tw->d_tf, sizeof(int32_t) * B * cap, hipMemcpyDeviceToHost));
  if (t_total) VO_HIP(c, hipMemcpy(t_total, tw->d_tt, sizeof(int32_t) * B * cap, hipMemcpyDeviceToHost));
  if (tag) VO_HIP(c, hipMemcpy(tag, tw->d_tag, sizeof(int32_t) * B * cap, hipMemcpyDeviceToHost));
  if (dead_tag) VO_HIP(c, hipMemcpy(dead_tag, tw->d_dead_tag, sizeof(int32_t) * B * cap, hipMemcpyDeviceToHost));
  // tighten the host's upper bound while we are synchronised anyway
  std::vector<int32_t> cnt(B);
  VO_HIP(c, hipMemcpy(cnt.data(), tw->d_n, sizeof(int32_t) * B, hipMemcpyDeviceToHost));
  int hi = 0;
  for (size_t b = 0; b < B; b++) hi = cnt[b] > hi ? cnt[b] : hi;
  tw->n_hi = hi;
  return VO_OK;
}

// obs [batch][window][max_pts][2] float64 (NaN = not observed): slot s <-> frame t_now - s, the layout vo_ba_upload takes
// (with N = max_pts; tracks beyond a sequence's count are all-NaN = unobserved landmarks)
extern "C" int32_t vo_tracks_obs(vo_ctx* c, int32_t t_now, int32_t window, double* obs) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->trk, VO_E_STATE, "vo_tracks_seed first");
  VO_CHECK(c, obs && window >= 1 && window <= VO_TRK_HIST, VO_E_INVALID, "window must be 1..32");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  vo_trk_ws* tw = c->trk;
  const size_t total = (size_t)c->batch * window * tw->cap * 2;
  if (tw->obs_cap < total) {
    if (tw->d_obs) (void)hipFree(tw->d_obs);
    tw->d_obs = nullptr; tw->obs_cap = 0;
    VO_HIP(c, hipMalloc((void**)&tw->d_obs, sizeof(double) * total));
    tw->obs_cap = total;
  }
  hipLaunchKernelGGL(k_trk_obs, dim3(vo_div_up(tw->cap, 256), window, c->batch), dim3(256), 0, c->stream, trk_make(tw), t_now, window, tw->d_obs);
  VO_HIP(c, hipGetLastError());
  VO_HIP(c, hipMemcpyAsync(obs, tw->d_obs, sizeof(double) * total, hipMemcpyDeviceToHost, c->stream));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  return VO_OK;
}

// The same table written straight into the resident bundle-adjustment problem (vo_ba_upload with N = max_pts landmarks
// = the track slots and W window poses): the observations of the next vo_ba_solve_resident never leave the GPU.  (async)
extern "C" int32_t vo_ba_obs_from_tracks(vo_ctx* c, int32_t t_now) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->trk, VO_E_STATE, "vo_tracks_seed first");
  int W = 0, N = 0;
  double* d_obs = vo_ba_obs_device(c, &W, &N);
  VO_CHECK(c, d_obs, VO_E_STATE, "vo_ba_upload first");
  VO_CHECK(c, N == c->trk->cap && W <= VO_TRK_HIST, VO_E_INVALID, "the resident BA problem must have max_pts landmarks and <= 32 slots");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  hipLaunchKernelGGL(k_trk_obs, dim3(vo_div_up(N, 256), W, c->batch), dim3(256), 0, c->stream, trk_make(c->trk), t_now, W, d_obs);
  VO_HIP(c, hipGetLastError());
  return VO_OK;
}
