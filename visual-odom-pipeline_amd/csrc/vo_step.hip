// Fused per-frame step for resident sequences: ONE host call enqueues the whole hot path of a frame
// (pyramid + Scharr, KLT, DLT, bundle adjustment, Shi-Tomasi re-detection, result copies), and after the first
// frame of each buffer parity the launch sequence is replayed from a captured hipGraph -- the ~40 launches of a
// frame are launch-latency bound, so the host cost per frame drops from ~40 launches to one graph launch.
//
// Mirrors the order of Pipeline.step, /root/reference/src/pipeline/pipeline.py:92-167 (track -> triangulate ->
// bundle-adjust -> re-detect), with the Python object bookkeeping left to the caller.
#include "vo_internal.h"

#include <string.h>

struct step_cfg {
  int n_pts, do_dlt, do_ba, do_st, mask_radius;
  vo_klt_params klt;
  vo_st_params st;
  vo_ba_params ba;
};

// enqueue everything of one frame on the ctx stream (also used under stream capture)
static int32_t step_enqueue(vo_ctx* c, const step_cfg& s, const int32_t* d_frame_idx, int frame_idx, int half) {
  int32_t r;
  const size_t fr = (size_t)c->width * c->height;
  if (d_frame_idx) r = vo_build_pyramid(c, c->d_seq, fr * c->seq_n, d_frame_idx);
  else r = vo_build_pyramid(c, c->d_seq + (size_t)frame_idx * fr, fr * c->seq_n, nullptr);
  if (r != VO_OK) return r;
  r = vo_klt_track_resident(c, s.n_pts, &s.klt);
  if (r != VO_OK) return r;
  // Re-detection needs only the new frame and the tracked points, DLT + BA only the tracked points: the two branches
  // run side by side (the BA iterations are chains of narrow latency-bound launches, Shi-Tomasi is wide streaming
  // kernels).  Under graph capture everything stays on the one captured stream.
  const bool fork = s.do_st && (s.do_dlt || s.do_ba) && !d_frame_idx && c->side_stream;
  const bool dlt_side = fork && s.do_dlt && s.do_ba;
  if (fork) {
    VO_HIP(c, hipEventRecord(c->ev_fork, c->stream));
    VO_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    hipStream_t main_stream = c->stream;
    c->stream = c->stream2;
    r = vo_shi_tomasi_resident(c, s.n_pts, s.mask_radius, &s.st);
    // the triangulation does not feed this frame's bundle adjustment (the BA problem is resident): beside a BA it goes to
    // the side branch too, off the KLT -> BA critical path (48 us of latency-bound work per step)
    if (r == VO_OK && dlt_side) r = vo_dlt_resident(c);
    c->stream = main_stream;
    if (r != VO_OK) return r;
    VO_HIP(c, hipEventRecord(c->ev_join, c->stream2));
  }
  if (s.do_dlt && !dlt_side) { r = vo_dlt_resident(c); if (r != VO_OK) return r; }
  if (s.do_ba) { r = vo_ba_solve_resident(c, &s.ba); if (r != VO_OK) return r; }
  if (fork) VO_HIP(c, hipStreamWaitEvent(c->stream, c->ev_join, 0));
  else if (s.do_st) { r = vo_shi_tomasi_resident(c, s.n_pts, s.mask_radius, &s.st); if (r != VO_OK) return r; }
  VO_HIP(c, hipMemcpyAsync(c->h_slab + (size_t)half * c->slab_bytes, c->d_slab, c->slab_bytes, hipMemcpyDeviceToHost, c->stream));
  if (s.do_ba) { r = vo_ba_enqueue_pub_copy(c, half); if (r != VO_OK) return r; }
  c->step_off_p[half] = vo_off_p(c);
  return VO_OK;
}

static void step_signature(const vo_ctx* c, const step_cfg& s, int sig[8]) {
  sig[0] = s.n_pts; sig[1] = s.do_dlt | (s.do_ba << 1) | (s.do_st << 2) | (c->bil_maxk << 8) | (c->ba_sharded << 16); sig[2] = s.mask_radius;
  sig[3] = s.klt.win | (s.klt.max_level << 8) | (s.klt.max_count << 16);
  sig[4] = s.ba.max_iters; sig[5] = s.st.max_corners | (s.st.block_size << 16);
  sig[6] = c->p_parity;   // point ping-pong parity
  sig[7] = c->dlt_n;
}

extern "C" int32_t vo_frame_step_resident(vo_ctx* c, int32_t frame_idx, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                                          int32_t do_st, int32_t mask_radius, const vo_klt_params* klt,
                                          const vo_st_params* st, const vo_ba_params* ba) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->d_seq && frame_idx >= 0 && frame_idx < c->seq_n, VO_E_STATE, "no resident sequence / bad index");
  VO_CHECK(c, c->n_pushed >= 1, VO_E_STATE, "push one frame before stepping");
  VO_HIP(c, hipSetDevice(c->device));
  step_cfg s;
  s.n_pts = n_pts; s.do_dlt = do_dlt ? 1 : 0; s.do_ba = do_ba ? 1 : 0; s.do_st = do_st ? 1 : 0; s.mask_radius = mask_radius;
  if (klt) s.klt = *klt; else vo_klt_default_params(&s.klt);
  if (st) s.st = *st; else vo_st_default_params(&s.st);
  if (ba) s.ba = *ba; else vo_ba_default_params(&s.ba);
  if (s.do_dlt) VO_CHECK(c, c->dlt_n > 0, VO_E_STATE, "vo_dlt_upload first");
  if (s.do_ba) VO_CHECK(c, vo_ba_ready(c), VO_E_STATE, "vo_ba_upload first");
  if (s.do_st) { int32_t r = vo_st_prepare(c, &s.st); if (r != VO_OK) return r; }    // allocations happen outside any capture

  // up to two steps may be in flight: step t + 1 is enqueued while the host still reads step t's (pinned) results.
  // A captured graph has its host destination baked in, so graph mode keeps one step in flight and one mirror half.
  VO_CHECK(c, c->steps_enq - c->steps_fetched < (c->use_graph ? 1 : 2), VO_E_STATE, "vo_frame_fetch the previous step(s) first");
  const int half = c->use_graph ? 0 : (int)(c->steps_enq & 1);
  const bool graph_ok = c->use_graph && c->prof.mask == 0 && c->n_pushed >= 2;
  if (!graph_ok) {
    const int32_t r = step_enqueue(c, s, nullptr, frame_idx, half);
    if (r != VO_OK) return r;
    VO_HIP(c, hipEventRecord(c->ev_step[half], c->stream));
    c->steps_enq++;
    return VO_OK;
  }

  const int parity = c->cur;                 // frame-store parity BEFORE this step
  int sig[8];
  step_signature(c, s, sig);
  if (!c->step_graph[parity] || memcmp(sig, c->step_sig[parity], sizeof(sig)) != 0) {
    // (re)capture: the enqueue functions advance the host-side state exactly as a direct call would
    if (c->step_graph[parity]) { (void)hipGraphExecDestroy(c->step_graph[parity]); c->step_graph[parity] = nullptr; }
    hipGraph_t g = nullptr;
    VO_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    const int32_t r = step_enqueue(c, s, c->d_frame_idx, frame_idx, 0);
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (r != VO_OK) { if (g) (void)hipGraphDestroy(g); return r; }
    VO_HIP(c, e);
    VO_HIP(c, hipGraphInstantiate(&c->step_graph[parity], g, nullptr, nullptr, 0));
    VO_HIP(c, hipGraphDestroy(g));
    memcpy(c->step_sig[parity], sig, sizeof(sig));
  } else {
    // replay: redo the host-side state changes the enqueue functions would have made
    c->cur ^= 1; c->n_pushed++;
    c->p_parity ^= 1;
  }
  c->frame_ring = (c->frame_ring + 1) & 63;
  c->h_frame_idx[c->frame_ring] = frame_idx;
  VO_HIP(c, hipMemcpyAsync(c->d_frame_idx, &c->h_frame_idx[c->frame_ring], sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipGraphLaunch(c->step_graph[parity], c->stream));
  c->step_off_p[0] = vo_off_p(c);
  VO_HIP(c, hipEventRecord(c->ev_step[0], c->stream));
  c->steps_enq++;
  return VO_OK;
}

// Waits for the step and unpacks the pinned result mirrors.  Any pointer may be NULL.  All arrays carry the leading
// batch dimension: p [batch][n_pts][2], status/err [batch][n_pts]; X4 [batch][4][dlt_n], depth1/reproj [batch][dlt_n];
// poses [batch][W][6], points [batch][N][3], stats [batch]; corners [batch][max_corners][2], n_corners [batch].
extern "C" int32_t vo_frame_fetch(vo_ctx* c, int32_t n_pts, float* p, uint8_t* status, float* err, float* X4,
                                  double* depth1, double* reproj, double* poses, double* points, vo_ba_stats* stats,
                                  float* corners, int32_t* n_corners) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, n_pts >= 0 && n_pts <= c->max_pts, VO_E_INVALID, "bad n_pts");
  VO_HIP(c, hipSetDevice(c->device));
  // the OLDEST step not fetched yet; with nothing in flight: the last one again
  int half;
  if (c->steps_fetched < c->steps_enq) {
    half = c->use_graph ? 0 : (int)(c->steps_fetched & 1);
    VO_HIP(c, hipEventSynchronize(c->ev_step[half]));
    c->steps_fetched++;
  } else {
    VO_CHECK(c, c->steps_enq > 0, VO_E_STATE, "no step to fetch");
    half = c->use_graph ? 0 : (int)((c->steps_enq - 1) & 1);
    VO_HIP(c, hipStreamSynchronize(c->stream));
  }
  const size_t off_p = c->step_off_p[half];
  const int mc = vo_st_last_max_corners(c) > 0 ? vo_st_last_max_corners(c) : 4096;
  int32_t rc = VO_OK;
  for (int b = 0; b < c->batch; b++) {
    const uint8_t* h = c->h_slab + (size_t)half * c->slab_bytes + (size_t)b * c->slab_seq;
    if (p) memcpy(p + (size_t)b * 2 * n_pts, h + off_p, sizeof(float) * 2 * n_pts);
    if (status) memcpy(status + (size_t)b * n_pts, h + c->off_status, n_pts);
    if (err) memcpy(err + (size_t)b * n_pts, h + c->off_err, sizeof(float) * n_pts);
    if (c->dlt_n > 0) {
      const size_t n = (size_t)c->dlt_n;
      if (X4) memcpy(X4 + b * 4 * n, h + c->off_X4, sizeof(float) * 4 * n);
      if (depth1) memcpy(depth1 + b * n, h + c->off_depth, sizeof(double) * n);
      if (reproj) memcpy(reproj + b * n, h + c->off_reproj, sizeof(double) * n);
    }
    if (n_corners) {
      const uint32_t* sc = reinterpret_cast<const uint32_t*>(h + c->off_st_scalars);
      if (sc[2] == 0xFFFFFFFFu) { n_corners[b] = 0; rc = vo_fail(c, VO_E_CAPACITY, "shi_tomasi: the 16384 strongest of the NMS candidates did not yield max_corners corners (or more than 262144 candidates)"); continue; }
      n_corners[b] = (int32_t)sc[2];
      if (corners && n_corners[b] > 0) memcpy(corners + (size_t)b * 2 * mc, h + c->off_st_out, sizeof(float) * 2 * (size_t)n_corners[b]);
    }
  }
  if ((poses || points || stats) && vo_ba_ready(c)) vo_ba_unpack_pub(c, half, poses, points, stats);
  return rc;
}

extern "C" int32_t vo_set_graph_mode(vo_ctx* c, int32_t on) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->steps_enq == c->steps_fetched, VO_E_STATE, "fetch the steps in flight before switching the launch mode");
  c->use_graph = on ? 1 : 0;
  return VO_OK;
}
