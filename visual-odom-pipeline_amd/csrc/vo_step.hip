// Fused per-frame step for resident sequences: ONE host call enqueues the whole hot path of a frame
// (pyramid + Scharr, KLT, DLT, bundle adjustment, Shi-Tomasi re-detection, result copies), and after the first
// frame of each buffer parity the launch sequence is replayed from a captured hipGraph -- the ~40 launches of a
// frame are launch-latency bound, so the host cost per frame drops from ~40 launches to one graph launch.
//
// Mirrors the order of Pipeline.step, /root/reference/src/pipeline/pipeline.py:92-167 (track -> triangulate ->
// bundle-adjust -> re-detect), with the Python object bookkeeping left to the caller.
#include "vo_internal.h"

#include <stdlib.h>
#include <string.h>

#include <mutex>
#include <vector>
#include <stdio.h>
// VO_STEP_TRACE=1 (debug): in-stream event stamps of the pipelined layout -- (start, end) of every bundle adjustment on stream C,
// (start, end) and (after the pyramid, after the KLT) of the front end on stream A.  One trace per process, filled under a mutex
// (several host threads step their own contexts); vo_debug_step_trace_dump(first, count) prints a window of it and frees the events.
static std::mutex g_trace_mu;
static std::vector<hipEvent_t> g_tr, g_ta, g_tb;
static void trace_push(std::vector<hipEvent_t>& v, hipStream_t st, size_t cap) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  if (v.size() >= cap) return;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess) return;
  (void)hipEventRecord(e, st);
  v.push_back(e);
}
extern "C" void vo_debug_step_trace_dump(int first_step, int n_steps) {
  std::lock_guard<std::mutex> lk(g_trace_mu);
  for (size_t i = 2 * (size_t)(first_step < 0 ? 0 : first_step); i + 3 < g_tr.size() && i + 3 < g_ta.size() && i + 3 < g_tb.size() && i < 2 * (size_t)(first_step + n_steps); i += 2) {
    float d = 0, gap = 0, a0 = 0, a1 = 0;
    (void)hipEventElapsedTime(&d, g_tr[i], g_tr[i + 1]);
    (void)hipEventElapsedTime(&gap, g_tr[i + 1], g_tr[i + 2]);
    (void)hipEventElapsedTime(&a0, g_tr[i], g_ta[i + 2]);          // start of the NEXT step's front end relative to this BA's start
    (void)hipEventElapsedTime(&a1, g_tr[i], g_ta[i + 3]);
    float b0 = 0, b1 = 0;
    (void)hipEventElapsedTime(&b0, g_tr[i], g_tb[i + 2]);
    (void)hipEventElapsedTime(&b1, g_tr[i], g_tb[i + 3]);
    fprintf(stderr, "step %zu: BA %.1f us, idle until the next BA starts %.1f us; next front end: start %.1f, pyramid done %.1f, KLT done %.1f, copy done %.1f us after this BA started\n",
            i / 2, d * 1e3, gap * 1e3, a0 * 1e3, b0 * 1e3, b1 * 1e3, a1 * 1e3);
  }
  for (std::vector<hipEvent_t>* v : {&g_tr, &g_ta, &g_tb}) { for (hipEvent_t e : *v) (void)hipEventDestroy(e); v->clear(); }
}

int32_t vo_quiesce_side(vo_ctx* c) {
  if (!c->in_step) c->main_dirty = true;                    // (vo_pipe_step clears it: see there)
  if (c->side_stream != 2 || c->in_step) return VO_OK;      // layouts 0 / 1 join their side stream inside the step
  if (c->stream2) VO_HIP(c, hipStreamSynchronize(c->stream2));
  if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
  return VO_OK;
}

static int32_t step_layout_apply(vo_ctx* c);

struct step_cfg {
  int n_pts, do_dlt, do_ba, do_st, mask_radius;
  vo_klt_params klt;
  vo_st_params st;
  vo_ba_params ba;
};

// enqueue everything of one frame on the ctx stream (also used under stream capture); *recorded: ev_step[half] has been recorded
// frame_idx < 0: the frames of this step arrive from the host in c->d_host_raw[half] (vo_frame_step_host: the upload runs on the copy stream,
// ev_h2d[half] says when it is there)
static bool step_trace_on() { static const bool on = getenv("VO_STEP_TRACE") != nullptr; return on; }

static int32_t step_enqueue(vo_ctx* c, const step_cfg& s, const int32_t* d_frame_idx, int frame_idx, int half, bool* recorded) {
  int32_t r;
  *recorded = false;
  const bool trace_a = step_trace_on();
  if (trace_a && c->side_stream == 2) trace_push(g_ta, c->stream, 4000);
  const size_t fr = (size_t)c->width * c->height;
  if (frame_idx < 0) {
    VO_HIP(c, hipStreamWaitEvent(c->stream, c->ev_h2d[half], 0));
    r = vo_build_pyramid(c, c->d_host_raw[half], fr, nullptr);
    if (r == VO_OK) { VO_HIP(c, hipEventRecord(c->ev_raw_free[half], c->stream)); c->raw_free_recorded[half] = true; }
  } else if (d_frame_idx) r = vo_build_pyramid(c, c->d_seq, fr * c->seq_n, d_frame_idx);
  else r = vo_build_pyramid(c, c->d_seq + (size_t)frame_idx * fr, fr * c->seq_n, nullptr);
  if (r != VO_OK) return r;
  if (trace_a && c->side_stream == 2) trace_push(g_tb, c->stream, 6000);
  // pipelined layout: the tracker's launch takes every free wave slot for its whole duration, the previous frame's LM chain would stand still
  // beside it (EXPERIMENTS.md round 5 item 9).  Its wide groups -- all problems running -- go first and get the whole chip; the tracker follows and
  // overlaps the chain's narrow tail groups, which run on the compute units stream A's CU mask leaves free (vo_set_side_stream)
  if (c->side_stream == 2 && s.do_ba && !d_frame_idx && c->ba_wide_groups > 0 && c->ba_wide_recorded)
    VO_HIP(c, hipStreamWaitEvent(c->stream, c->ev_ba_wide[half ^ 1], 0));
  r = vo_klt_track_resident(c, s.n_pts, &s.klt);
  if (r != VO_OK) return r;
  if (trace_a && c->side_stream == 2) trace_push(g_tb, c->stream, 6000);
  uint8_t* const h_dst = c->h_slab + (size_t)half * c->slab_bytes;
  if (c->side_stream == 2 && s.do_ba && !d_frame_idx) {
    // ---- pipelined layout: three in-order streams.  A (c->stream): pyramid, KLT and the copy of the KLT results -- then it is free
    // for the NEXT frame's front end; B (stream2): re-detection, triangulation and the copy of their results; C (stream3): the
    // bundle adjustment and the copy of its solution.  B waits for C at the end, ev_step[half] is recorded on B.  What makes it
    // safe with two steps in flight: the KLT results of step t leave the device on A before KLT(t + 1) can overwrite status / err;
    // B's and C's outputs are copied on B / C before step t + 1's kernels on the same stream; step t + 2 (which overwrites the
    // frame buffer and the point buffer B reads) is not enqueued before step t has been fetched. ----
    const size_t part1 = c->off_X4;
    // the fork comes BEFORE A's result copy: device-to-host copies of all streams pass through the copy engine in submission
    // order, and the copies of the previous step (submitted earlier, waiting for ITS bundle adjustment) would hold this one -- and
    // with it the start of this step's bundle adjustment -- until that BA has finished (measured: 185 us of a 250 us frame)
    VO_HIP(c, hipEventRecord(c->ev_fork, c->stream));
    VO_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    VO_HIP(c, hipStreamWaitEvent(c->stream3, c->ev_fork, 0));
    if (c->batch == 1) VO_HIP(c, hipMemcpyAsync(h_dst, c->d_slab, part1, hipMemcpyDeviceToHost, c->stream));
    else VO_HIP(c, hipMemcpy2DAsync(h_dst, c->slab_seq, c->d_slab, c->slab_seq, part1, c->batch, hipMemcpyDeviceToHost, c->stream));
    VO_HIP(c, hipEventRecord(c->ev_copy1[half], c->stream));
    if (trace_a) trace_push(g_ta, c->stream, 4000);
    hipStream_t main_stream = c->stream;
    c->stream = c->stream3;
    const bool trace = trace_a;                                        // debug: GPU-side timeline of stream C, printed by vo_debug_step_trace_dump
    if (trace) trace_push(g_tr, c->stream3, 4000);
    // C carries nothing but the LM iterations and k_ba_finalize: the copy of the solution goes to B (behind ev_ba), so that the next
    // frame's iterations follow this frame's directly; k_ba_finalize of the next step waits for that copy (ev_pub) -- long done by then
    c->ba_wait_before_publish = c->pub_copy_pending ? c->ev_pub[half ^ 1] : nullptr;
    c->ba_wide_event = c->ba_wide_groups > 0 ? c->ev_ba_wide[half] : nullptr;
    r = vo_ba_solve_resident(c, &s.ba);
    c->ba_wait_before_publish = nullptr;
    if (c->ba_wide_event && r == VO_OK) c->ba_wide_recorded = true;
    c->ba_wide_event = nullptr;
    hipError_t e = hipEventRecord(c->ev_ba[half], c->stream3);
    if (trace) trace_push(g_tr, c->stream3, 4000);
    c->stream = c->stream2;
    if (r == VO_OK && s.do_st) r = vo_shi_tomasi_resident(c, s.n_pts, s.mask_radius, &s.st);
    if (r == VO_OK && s.do_dlt) r = vo_dlt_resident(c);
    if (e == hipSuccess) e = (c->batch == 1) ? hipMemcpyAsync(h_dst + part1, c->d_slab + part1, c->slab_seq - part1, hipMemcpyDeviceToHost, c->stream2)
                                             : hipMemcpy2DAsync(h_dst + part1, c->slab_seq, c->d_slab + part1, c->slab_seq, c->slab_seq - part1, c->batch,
                                                                hipMemcpyDeviceToHost, c->stream2);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream2, c->ev_copy1[half], 0);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream2, c->ev_ba[half], 0);
    if (r == VO_OK && e == hipSuccess) r = vo_ba_enqueue_pub_copy(c, half);      // on B (c->stream is stream2 here)
    if (e == hipSuccess) e = hipEventRecord(c->ev_pub[half], c->stream2);
    c->pub_copy_pending = true;
    c->stream = main_stream;
    if (e == hipSuccess) e = hipEventRecord(c->ev_step[half], c->stream2);
    if (r != VO_OK) return r;
    VO_HIP(c, e);
    *recorded = true;
    c->step_off_p[half] = vo_off_p(c);
    return VO_OK;
  }
  // Re-detection needs only the new frame and the tracked points, DLT + BA only the tracked points: the two branches
  // run side by side (the BA iterations are chains of narrow latency-bound launches, Shi-Tomasi is wide streaming
  // kernels).  Under graph capture everything stays on the one captured stream.
  const bool fork = s.do_st && (s.do_dlt || s.do_ba) && !d_frame_idx && c->side_stream;
  const bool dlt_side = fork && s.do_dlt && s.do_ba;
  bool forked = false;
  // every exit after the fork joins the side stream again: work queued there must stay ordered before whatever the caller
  // enqueues (or frees) next on c->stream, also when a later enqueue fails
  auto join = [&]() -> hipError_t {
    if (!forked) return hipSuccess;
    forked = false;
    hipError_t e = hipEventRecord(c->ev_join, c->stream2);
    if (e == hipSuccess) e = hipStreamWaitEvent(c->stream, c->ev_join, 0);
    return e;
  };
  if (fork) {
    VO_HIP(c, hipEventRecord(c->ev_fork, c->stream));
    VO_HIP(c, hipStreamWaitEvent(c->stream2, c->ev_fork, 0));
    forked = true;
    hipStream_t main_stream = c->stream;
    c->stream = c->stream2;
    r = vo_shi_tomasi_resident(c, s.n_pts, s.mask_radius, &s.st);
    // the triangulation does not feed this frame's bundle adjustment (the BA problem is resident): beside a BA it goes to
    // the side branch too, off the KLT -> BA critical path (48 us of latency-bound work per step)
    if (r == VO_OK && dlt_side) r = vo_dlt_resident(c);
    c->stream = main_stream;
    if (r != VO_OK) { (void)join(); return r; }
  }
  if (s.do_dlt && !dlt_side) { r = vo_dlt_resident(c); if (r != VO_OK) { (void)join(); return r; } }
  if (s.do_ba) { r = vo_ba_solve_resident(c, &s.ba); if (r != VO_OK) { (void)join(); return r; } }
  if (fork) VO_HIP(c, join());
  else if (s.do_st) { r = vo_shi_tomasi_resident(c, s.n_pts, s.mask_radius, &s.st); if (r != VO_OK) return r; }
  VO_HIP(c, hipMemcpyAsync(h_dst, c->d_slab, c->slab_bytes, hipMemcpyDeviceToHost, c->stream));
  if (s.do_ba) { r = vo_ba_enqueue_pub_copy(c, half); if (r != VO_OK) return r; }
  c->step_off_p[half] = vo_off_p(c);
  return VO_OK;
}

// A captured step bakes in: every kernel argument passed by value (all of the three parameter structs), the device
// pointers of the resident buffers (the sequence store is reallocated by vo_seq_upload, the BA workspace by vo_ba_upload
// when the window or the landmark count outgrows it, the Shi-Tomasi workspace by vo_st_prepare) and the problem shapes the
// strides derive from.  The signature hashes all of it (FNV-1a); any difference re-captures.
static uint64_t step_signature(vo_ctx* c, const step_cfg& s) {
  uint64_t h = 1469598103934665603ull;
  auto mix = [&](const void* p, size_t n) { const unsigned char* b = static_cast<const unsigned char*>(p); for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 1099511628211ull; } };
  auto mix_i = [&](long long v) { mix(&v, sizeof(v)); };
  mix(&s, sizeof(s));                       // the caller zero-fills the struct, so padding bytes are defined
  mix_i(c->p_parity); mix_i(c->dlt_n); mix_i(c->dlt_stats); mix_i(c->bil_maxk); mix_i(c->ba_sharded); mix_i(c->side_stream);
  mix_i((long long)(uintptr_t)c->d_seq); mix_i(c->seq_n);
  mix_i((long long)(uintptr_t)c->d_uv0); mix_i((long long)(uintptr_t)c->d_dlt_cam);
  mix_i((long long)(uintptr_t)c->st); mix_i((long long)(uintptr_t)c->ba); mix_i((long long)(uintptr_t)c->d_pt_counts);
  int W = 0, N = 0;
  mix_i((long long)(uintptr_t)vo_ba_obs_device(c, &W, &N)); mix_i(W); mix_i(N);
  mix_i((long long)(uintptr_t)c->comm);
  mix_i(vo_st_launch_state(c));             // a clean exclusion mask drops k_st_mask_init from the launch list
  return h;
}

// ---- frames from the host -------------------------------------------------------------------------------------------------------------
// Page-locked host arrays are read by the GPU itself over PCIe: ONE launch per step gathers the `batch` images (wherever each of them lies,
// whatever their row stride) into d_host_raw[half] -- a copy-engine transfer per image cost ~32 us each (256 images: 8 ms for 119 MB).
// Workgroups take (image, part) units in turn, U chunks in flight per lane (tools/pcie_gather_probe.hip: 32 workgroups x 256 lanes x 4 chunks
// reach the bus's 56 GB/s; more buy nothing for the bus).  How many are launched is decided by who else is on the chip: vo_host_frames_upload.
template <int U>
__global__ __launch_bounds__(256) void k_gather_frames(const uint8_t* const* __restrict__ tab, int stride, int w, int h, uint8_t* __restrict__ dst, size_t fr, int n_img,
                                                        int parts) {
  // a workgroup takes (image, part) units in turn -- a part = a contiguous 1 / parts of an image's 16-byte chunks (rows) --, its lanes walk the
  // unit with a fixed stride: no division per chunk (beside the tracker a wave of this kernel gets one issue slot in seven; index arithmetic
  // is what it cannot afford)
  for (int unit = (int)blockIdx.x; unit < n_img * parts; unit += (int)gridDim.x) {
    const int img = unit / parts, part = unit - img * parts;
    const uint8_t* __restrict__ src = tab[img];
    uint8_t* __restrict__ d = dst + (size_t)img * fr;
    if (h == 1) {                                                          // rows contiguous: one run of w bytes
      const int n16 = w >> 4, per = (n16 + parts - 1) / parts, lo = part * per, hi = min(n16, lo + per);
      for (int i0 = lo + (int)threadIdx.x; i0 < hi; i0 += U * 256) {
        uint4 v[U];
#pragma unroll
        for (int u = 0; u < U; u++) if (i0 + u * 256 < hi) __builtin_memcpy(&v[u], src + (size_t)(i0 + u * 256) * 16, 16);   // (any alignment)
#pragma unroll
        for (int u = 0; u < U; u++) if (i0 + u * 256 < hi) __builtin_memcpy(d + (size_t)(i0 + u * 256) * 16, &v[u], 16);
      }
      if (part == parts - 1) for (int k = (n16 << 4) + (int)threadIdx.x; k < w; k += 256) d[k] = src[k];
    } else {
      const int n16 = w >> 4, per = (h + parts - 1) / parts, lo = part * per, hi = min(h, lo + per);
      for (int row = lo + ((int)threadIdx.x >> 6); row < hi; row += 4) {   // a wave per row
        const uint8_t* s = src + (size_t)row * stride;
        uint8_t* o = d + (size_t)row * w;
        for (int c = (int)threadIdx.x & 63; c < n16; c += 64) { uint4 v; __builtin_memcpy(&v, s + c * 16, 16); __builtin_memcpy(o + c * 16, &v, 16); }
        for (int k = (n16 << 4) + ((int)threadIdx.x & 63); k < w; k += 64) o[k] = s[k];
      }
    }
  }
}

// the upload of a host-frame step on the copy stream into d_host_raw[half].  Page-locked images: the gather kernel; pageable ones: a copy per
// image (runs of images that follow each other in host memory -- a [batch][h][w] array -- go as ONE copy), which the runtime stages
int32_t vo_host_frames_upload(vo_ctx* c, const uint8_t* const* frames, int32_t stride, int half) {
  const size_t fr = (size_t)c->width * c->height;
  if (!c->stream_h2d) {
    VO_HIP(c, hipStreamCreateWithFlags(&c->stream_h2d, hipStreamNonBlocking));
    VO_HIP(c, hipHostMalloc((void**)&c->h_ptr_tab, 2 * sizeof(void*) * (size_t)c->batch, hipHostMallocDefault));
    for (int k = 0; k < 2; k++) {
      VO_HIP(c, hipMalloc((void**)&c->d_host_raw[k], fr * c->batch));
      VO_HIP(c, hipEventCreateWithFlags(&c->ev_h2d[k], hipEventDisableTiming));
      VO_HIP(c, hipEventCreateWithFlags(&c->ev_raw_free[k], hipEventDisableTiming));
    }
  }
  // the pyramid of the step that used this half two steps ago has read it (that step has been fetched, so this wait never blocks in practice)
  if (c->raw_free_recorded[half]) VO_HIP(c, hipStreamWaitEvent(c->stream_h2d, c->ev_raw_free[half], 0));
  uint8_t* const dst = c->d_host_raw[half];
  // device-visible addresses of the images, if every one of them is page-locked
  const uint8_t** tab = c->h_ptr_tab + (size_t)half * c->batch;
  bool pinned = true;
  for (int b = 0; b < c->batch && pinned; b++) {
    hipPointerAttribute_t at;
    if (hipPointerGetAttributes(&at, frames[b]) != hipSuccess) { (void)hipGetLastError(); pinned = false; break; }
    if (at.type != hipMemoryTypeHost || !at.devicePointer) pinned = false;
    else tab[b] = static_cast<const uint8_t*>(at.devicePointer);
  }
  if (pinned) {
    const bool flat = stride == c->width;
    const int w = flat ? (int)fr : c->width, h = flat ? 1 : c->height;
    // how many workgroups: the bus is saturated from 32 x 256 lanes on (tools/pcie_gather_probe.hip), so the count is about who else is on the chip.
    // Gated layout of a batch (compute units set aside beside the tracker): many short units, they find the free compute units (512 workgroups:
    // host frames = resident frames; 32: 0.90 x).  Otherwise (closed loop, one stream layouts): few -- every resident workgroup of this kernel
    // holds wave slots for the 2.1 ms the bus needs, and the one-workgroup-per-problem kernels of the LM chain wait for compute units with
    // sixteen free ones (closed loop, 256 sequences: 47 300 frames/s with 16-32 workgroups, 41 700 with 128, 35 600 with 512)
    int parts = 1, gx = 32;
    if (c->side_stream == 2 && c->stream_reserve > 0) { parts = 2; gx = 2 * c->batch; }
    if (c->tune.gather_workgroups > 0) { gx = c->tune.gather_workgroups; parts = gx > c->batch ? (gx + c->batch - 1) / c->batch : 1; }
    if (gx > parts * c->batch) gx = parts * c->batch;
    hipLaunchKernelGGL(k_gather_frames<4>, dim3(gx), dim3(256), 0, c->stream_h2d, tab, flat ? (int)fr : stride, w, h, dst, fr, c->batch, parts);
    VO_HIP(c, hipGetLastError());
    VO_HIP(c, hipEventRecord(c->ev_h2d[half], c->stream_h2d));
    return VO_OK;
  }
  for (int b = 0; b < c->batch;) {
    int e = b + 1;
    if (stride == c->width) while (e < c->batch && frames[e] == frames[e - 1] + fr) e++;
    if (stride == c->width) VO_HIP(c, hipMemcpyAsync(dst + (size_t)b * fr, frames[b], fr * (size_t)(e - b), hipMemcpyHostToDevice, c->stream_h2d));
    else VO_HIP(c, hipMemcpy2DAsync(dst + (size_t)b * fr, c->width, frames[b], stride, c->width, c->height, hipMemcpyHostToDevice, c->stream_h2d));
    b = e;
  }
  VO_HIP(c, hipEventRecord(c->ev_h2d[half], c->stream_h2d));
  return VO_OK;
}

static int32_t frame_step(vo_ctx* c, int32_t frame_idx, const uint8_t* const* host_frames, int32_t stride, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                          int32_t do_st, int32_t mask_radius, const vo_klt_params* klt, const vo_st_params* st, const vo_ba_params* ba);

extern "C" int32_t vo_frame_step_resident(vo_ctx* c, int32_t frame_idx, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                                          int32_t do_st, int32_t mask_radius, const vo_klt_params* klt,
                                          const vo_st_params* st, const vo_ba_params* ba) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->d_seq && frame_idx >= 0 && frame_idx < c->seq_n, VO_E_STATE, "no resident sequence / bad index");
  return frame_step(c, frame_idx, nullptr, 0, n_pts, do_dlt, do_ba, do_st, mask_radius, klt, st, ba);
}

extern "C" int32_t vo_frame_step_host(vo_ctx* c, const uint8_t* const* frames, int32_t stride, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                                      int32_t do_st, int32_t mask_radius, const vo_klt_params* klt,
                                      const vo_st_params* st, const vo_ba_params* ba) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, frames != nullptr && stride >= c->width, VO_E_INVALID, "bad frame pointers / stride");
  for (int b = 0; b < c->batch; b++) VO_CHECK(c, frames[b] != nullptr, VO_E_INVALID, "null frame pointer");
  return frame_step(c, -1, frames, stride, n_pts, do_dlt, do_ba, do_st, mask_radius, klt, st, ba);
}

extern "C" int32_t vo_host_alloc(uint64_t bytes, void** out) {
  if (!out || bytes == 0) return VO_E_INVALID;
  *out = nullptr;
  return hipHostMalloc(out, (size_t)bytes, hipHostMallocDefault) == hipSuccess ? VO_OK : VO_E_NOMEM;
}

extern "C" int32_t vo_host_free(void* p) {
  if (!p) return VO_OK;
  return hipHostFree(p) == hipSuccess ? VO_OK : VO_E_HIP;
}

static int32_t frame_step(vo_ctx* c, int32_t frame_idx, const uint8_t* const* host_frames, int32_t stride, int32_t n_pts, int32_t do_dlt, int32_t do_ba,
                          int32_t do_st, int32_t mask_radius, const vo_klt_params* klt, const vo_st_params* st, const vo_ba_params* ba) {
  VO_CHECK(c, c->n_pushed >= 1, VO_E_STATE, "push one frame before stepping");
  VO_HIP(c, hipSetDevice(c->device));
  if (c->layout_suspended) {
    // a closed-loop step (vo_pipe_step) took the gate and the CU mask off; this is the first frame step since: put the layout back
    VO_CHECK(c, !vo_pipe_busy(c), VO_E_STATE, "vo_pipe_fetch the closed-loop steps in flight first");
    if (c->steps_enq == c->steps_fetched) { const int32_t rl = step_layout_apply(c); if (rl != VO_OK) return rl; }
  }
  step_cfg s;
  memset(&s, 0, sizeof(s));
  s.n_pts = n_pts; s.do_dlt = do_dlt ? 1 : 0; s.do_ba = do_ba ? 1 : 0; s.do_st = do_st ? 1 : 0; s.mask_radius = mask_radius;
  if (klt) s.klt = *klt; else vo_klt_default_params(&s.klt);
  if (st) s.st = *st; else vo_st_default_params(&s.st);
  if (ba) s.ba = *ba; else vo_ba_default_params(&s.ba);
  if (s.do_dlt) VO_CHECK(c, c->dlt_n > 0, VO_E_STATE, "vo_dlt_upload first");
  if (s.do_ba) VO_CHECK(c, vo_ba_ready(c), VO_E_STATE, "vo_ba_upload first");
  if (s.do_st) { int32_t r = vo_st_prepare(c, &s.st); if (r != VO_OK) return r; }    // allocations happen outside any capture

  // up to two steps may be in flight: step t + 1 is enqueued while the host still reads step t's (pinned) results.
  // A captured graph has its host destination baked in: one graph per (frame parity, mirror half), so graph mode keeps two steps
  // in flight as well (round 2 kept ONE graph per parity and one step in flight -- that, not the replay itself, is what made graph
  // mode 0.2 ms per frame slower: tools/graph_probe.hip, EXPERIMENTS.md, design section 8)
  VO_CHECK(c, c->steps_enq - c->steps_fetched < 2, VO_E_STATE, "vo_frame_fetch the previous step(s) first");
  c->main_dirty = true;
  const int half = (int)(c->steps_enq & 1);
  if (host_frames) { const int32_t ru = vo_host_frames_upload(c, host_frames, stride, half); if (ru != VO_OK) return ru; }
  const bool graph_ok = c->use_graph && c->prof.mask == 0 && c->n_pushed >= 2 && !host_frames;     // (host frames: plain launches)
  if (!graph_ok) {
    bool recorded = false;
    c->in_step = true;
    const int32_t r = step_enqueue(c, s, nullptr, frame_idx, half, &recorded);
    c->in_step = false;
    if (r != VO_OK) return r;
    if (!recorded) VO_HIP(c, hipEventRecord(c->ev_step[half], c->stream));
    c->steps_enq++;
    return VO_OK;
  }

  uint64_t sig = step_signature(c, s);
  sig = (sig ^ (uint64_t)(2 * c->cur + half + 1)) * 1099511628211ull;       // frame-store parity BEFORE this step x the pinned mirror half
  hipGraphExec_t exec = nullptr;
  for (auto& g : c->step_graphs) if (g.first == sig) { exec = g.second; break; }
  if (!exec) {
    // capture: the enqueue functions advance the host-side state exactly as a direct call would
    if (c->step_graphs.size() >= 64) {             // (a caller that keeps changing parameters: start over rather than grow without bound)
      for (auto& g : c->step_graphs) (void)hipGraphExecDestroy(g.second);
      c->step_graphs.clear();
    }
    hipGraph_t g = nullptr;
    const int cur0 = c->cur, pushed0 = c->n_pushed, parity0 = c->p_parity, st0 = vo_st_flags_save(c);
    VO_HIP(c, hipStreamBeginCapture(c->stream, hipStreamCaptureModeThreadLocal));
    bool recorded = false;
    c->in_step = true;
    const int32_t r = step_enqueue(c, s, c->d_frame_idx, frame_idx, half, &recorded);
    c->in_step = false;
    const hipError_t e = hipStreamEndCapture(c->stream, &g);
    if (r != VO_OK || e != hipSuccess) {
      // nothing was launched: undo what the enqueue functions did to the host-side frame / point parities and launch flags
      c->cur = cur0; c->n_pushed = pushed0; c->p_parity = parity0; vo_st_flags_restore(c, st0);
      if (g) (void)hipGraphDestroy(g);
      if (r != VO_OK) return r;
    }
    VO_HIP(c, e);
    const hipError_t ei = hipGraphInstantiate(&exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    if (ei != hipSuccess) { c->cur = cur0; c->n_pushed = pushed0; c->p_parity = parity0; vo_st_flags_restore(c, st0); }
    VO_HIP(c, ei);
    c->step_graphs.emplace_back(sig, exec);
  } else {
    // replay: redo the host-side state changes the enqueue functions would have made
    c->cur ^= 1; c->n_pushed++;
    c->p_parity ^= 1;
  }
  c->frame_ring = (c->frame_ring + 1) & 63;
  c->h_frame_idx[c->frame_ring] = frame_idx;
  VO_HIP(c, hipMemcpyAsync(c->d_frame_idx, &c->h_frame_idx[c->frame_ring], sizeof(int32_t), hipMemcpyHostToDevice, c->stream));
  VO_HIP(c, hipGraphLaunch(exec, c->stream));
  c->step_off_p[half] = vo_off_p(c);
  VO_HIP(c, hipEventRecord(c->ev_step[half], c->stream));
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
    half = (int)(c->steps_fetched & 1);
    VO_HIP(c, hipEventSynchronize(c->ev_step[half]));
    c->steps_fetched++;
  } else {
    VO_CHECK(c, c->steps_enq > 0, VO_E_STATE, "no step to fetch");
    half = (int)((c->steps_enq - 1) & 1);
    VO_HIP(c, hipStreamSynchronize(c->stream));
    if (c->stream2) VO_HIP(c, hipStreamSynchronize(c->stream2));
    if (c->stream3) VO_HIP(c, hipStreamSynchronize(c->stream3));
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

// the ctx stream as a queue that leaves `reserve` compute units free (0: all of them): a fresh stream, after the old one has drained
int32_t vo_main_stream_reserve(vo_ctx* c, int reserve) {
  if (reserve == c->stream_reserve) return VO_OK;
  VO_HIP(c, hipSetDevice(c->device));
  VO_HIP(c, hipStreamSynchronize(c->stream));
  hipStream_t fresh = nullptr;
  hipError_t e = vo_stream_create(&fresh, reserve);
  if (e != hipSuccess && reserve > 0) {          // no CU masks on this runtime: then no gate either -- the tail groups would have no compute units of their own
    (void)hipGetLastError(); reserve = 0; c->ba_wide_groups = 0;
    e = vo_stream_create(&fresh, 0);
  }
  VO_HIP(c, e);
  VO_HIP(c, hipStreamDestroy(c->stream));
  c->stream = fresh;
  c->stream_reserve = reserve;
  return VO_OK;
}

// Pipelined layout of a batch: the tracker launch of frame t + 1 follows the first `ba_wide_groups` LM groups of frame t (step_enqueue), and
// stream A leaves 32 compute units to the chain's tail groups that then run beside it -- one CU of each shader engine of each XCD: the mask
// bits are dealt round-robin over XCDs and engines, a count that is no multiple of 32 unbalances the engines (240, 232 and 248 of 256 CUs
// measured SLOWER than 224; tools/gate_ab.sh, EXPERIMENTS.md round 5 item 15).  One sequence (a 34 us tracker launch that starves nobody)
// keeps the plain layout, and so does graph replay (a captured step stays on one stream, which must then have the whole chip).
// vo_tuning.gate_groups / reserve_cus override (-1 = off).
static int32_t step_layout_apply(vo_ctx* c) {
  int groups = 0, reserve = 0;
  if (c->side_stream == 2 && !c->use_graph) {
    if (c->batch >= 8) { groups = c->batch >= 256 ? 4 : 5; reserve = 32; }
    if (c->tune.gate_groups) groups = c->tune.gate_groups > 0 ? c->tune.gate_groups : 0;
    if (c->tune.reserve_cus) reserve = c->tune.reserve_cus > 0 ? c->tune.reserve_cus : 0;
  }
  c->ba_wide_groups = groups;
  c->ba_wide_recorded = false;
  c->layout_suspended = false;
  return vo_main_stream_reserve(c, reserve);
}

extern "C" int32_t vo_set_side_stream(vo_ctx* c, int32_t on) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->steps_enq == c->steps_fetched && !vo_pipe_busy(c), VO_E_STATE, "fetch the steps in flight before switching the stream layout");
  if (on == 2 && !c->stream3) {
    // created on demand: streams share the hardware queues, and a third (idle) stream per context re-deals which of them share one --
    // three batched contexts lost 10 % of their throughput to it
    VO_HIP(c, hipSetDevice(c->device));
    VO_HIP(c, hipStreamCreateWithFlags(&c->stream3, hipStreamNonBlocking));     // (a highest-priority stream for the BA chain: no difference)
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_ba_wide[0], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_ba_wide[1], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_ba[0], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_ba[1], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_pub[0], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_pub[1], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_copy1[0], hipEventDisableTiming));
    VO_HIP(c, hipEventCreateWithFlags(&c->ev_copy1[1], hipEventDisableTiming));
  }
  c->side_stream = (on == 2) ? 2 : (on ? 1 : 0);      // 2: pipelined (BA of frame t beside the front end of frame t + 1)
  return step_layout_apply(c);
}

extern "C" int32_t vo_get_tuning(vo_ctx* c, vo_tuning* out) {
  if (!c || !out) return VO_E_INVALID;
  *out = c->tune;
  return VO_OK;
}

extern "C" int32_t vo_set_tuning(vo_ctx* c, const vo_tuning* t) {
  if (!c || !t) return VO_E_INVALID;
  VO_CHECK(c, c->steps_enq == c->steps_fetched && !vo_pipe_busy(c), VO_E_STATE, "fetch the steps in flight before changing the tuning");
  VO_CHECK(c, (t->ba_kernels >= 0 && t->ba_kernels <= 2) && (t->ba_lanes == 0 || t->ba_lanes == 8 || t->ba_lanes == 16) &&
              (t->ba_threads == 0 || t->ba_threads == 256 || t->ba_threads == 512 || t->ba_threads == 1024) && t->ba_pitch_pad >= 0 && t->ba_pitch_pad <= 64 &&
              t->ba_chunks >= 0 && t->ba_workgroups >= 0 && t->ba_workgroup_cap >= 0 && t->ba_fold >= 0 && t->ba_fold <= 2 &&
              (t->klt_waves == 0 || (t->klt_waves >= 4 && t->klt_waves <= 6)) && t->st_band_rows >= 0 && t->gate_groups >= -1 && t->reserve_cus >= -1 &&
              t->reserve_cus < 256, VO_E_INVALID, "field out of range");
#ifndef VO_EXPERIMENTS
  VO_CHECK(c, t->klt_pair == 0, VO_E_INVALID, "klt_pair needs a build with -DVO_EXPERIMENTS");
#else
  VO_CHECK(c, t->klt_pair == 0 || (t->klt_pair >= 3 && t->klt_pair <= 5), VO_E_INVALID, "klt_pair: 3, 4 or 5");
#endif
  VO_CHECK(c, t->gather_workgroups >= 0 && t->gather_workgroups <= 65536, VO_E_INVALID, "gather_workgroups out of range");
  for (int k = 0; k < 13; k++) VO_CHECK(c, t->reserved[k] == 0, VO_E_INVALID, "reserved fields must be 0");
  VO_HIP(c, hipSetDevice(c->device));
  { const int32_t rq = vo_quiesce_side(c); if (rq != VO_OK) return rq; }
  c->tune = *t;
  for (auto& g : c->step_graphs) if (g.second) (void)hipGraphExecDestroy(g.second);      // a captured step bakes the kernel choices in
  c->step_graphs.clear();
  return step_layout_apply(c);
}

extern "C" int32_t vo_step_layout(vo_ctx* c, int32_t* layout, int32_t* gate_groups, int32_t* reserved_cus) {
  if (!c) return VO_E_INVALID;
  if (layout) *layout = c->side_stream;
  if (gate_groups) *gate_groups = c->ba_wide_groups;
  if (reserved_cus) *reserved_cus = c->stream_reserve;
  return VO_OK;
}

extern "C" int32_t vo_set_graph_mode(vo_ctx* c, int32_t on) {
  if (!c) return VO_E_INVALID;
  VO_CHECK(c, c->steps_enq == c->steps_fetched && !vo_pipe_busy(c), VO_E_STATE, "fetch the steps in flight before switching the launch mode");
  c->use_graph = on ? 1 : 0;
  { const int32_t rr = step_layout_apply(c); if (rr != VO_OK) return rr; }
  return VO_OK;
}
