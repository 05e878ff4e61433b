// Internal definitions shared by the HIP translation units of libvo_mi355x.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string>
#include <type_traits>

#include "vo_mi355x.h"

#define VO_PAD 32          // border (pixels) around every pyramid level, >= win + 1
#define VO_MAX_LEVELS 8
#define VO_MAX_WIN 31

struct vo_level {
  int w, h;        // interior size
  int pitch;       // padded row pitch in PIXELS (multiple of 64)
  int ph;          // padded rows = h + 2*VO_PAD
};

struct vo_frame {
  uint8_t* img[VO_MAX_LEVELS];   // padded, origin of the interior at (VO_PAD, VO_PAD)
  int16_t* der[VO_MAX_LEVELS];   // padded, interleaved (Ix, Iy), same pixel pitch; border = 0
};

#include <vector>
struct vo_prof {
  int mask = 0;                                                  // bit r = region r is timed
  std::vector<hipEvent_t> pool;                                  // all events ever created (reused)
  size_t used = 0;
  std::vector<std::pair<hipEvent_t, hipEvent_t>> pairs[VO_PROF_COUNT];
};

struct vo_st_ws;   // Shi-Tomasi workspace (vo_shi_tomasi.hip)
struct vo_ba_ws;   // bundle-adjustment workspace (vo_ba.hip)

// per-sequence camera data of the two-view triangulation (vo_dlt.hip; written on the device by the closed-loop pipeline)
struct vo_dlt_cam {
  float P0[12], P1[12];    // K @ H[:3,:] rounded to float32, as the reference hands them to cv2.triangulatePoints (extractor.py:268-269)
  double M0[12], M1[12];   // the same products in float64 (filter statistics)
  double H1z[4];           // third row of H1
};

// A context carries `batch` independent sequences in lockstep: every device buffer has a leading sequence
// dimension with a uniform stride and every kernel a grid dimension over sequences, so that one launch serves
// the whole batch (the path is launch / latency bound per sequence; batching is what fills the 256 CUs).
struct vo_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream2 = nullptr;     // side stream of the fused frame step: Shi-Tomasi runs beside DLT + BA (fork / join by events)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  hipStream_t stream3 = nullptr;     // pipelined frame step (vo_set_side_stream(c, 2)): the bundle adjustment of frame t runs here beside the front end of frame t + 1
  hipEvent_t ev_ba[2] = {nullptr, nullptr};     // the BA of the step using half k has published its solution (d_pub)
  hipEvent_t ev_copy1[2] = {nullptr, nullptr};  // the KLT results of the step using half k have left the device (copy on stream A)
  hipEvent_t ev_pub[2] = {nullptr, nullptr};    // ... and the copy of it (on stream2) has left the device: the next k_ba_finalize may overwrite d_pub
  hipEvent_t ba_wait_before_publish = nullptr;  // set by the pipelined step around vo_ba_solve_resident
  hipEvent_t ev_ba_wide[2] = {nullptr, nullptr}; // pipelined step: the first `ba_wide_groups` LM groups of the step using half k have run (recorded on stream C);
                                                // the NEXT step's tracker launch waits for it -- the wide groups get the whole chip, the tail groups run beside the tracker
  hipEvent_t ba_wide_event = nullptr;           // set by the pipelined step around vo_ba_solve_resident: record after `ba_wide_groups` groups
  int ba_wide_groups = 0;                       // 0: no gating
  bool ba_wide_recorded = false;
  int stream_reserve = 0;                        // compute units `stream` leaves free (CU mask of its queue; vo_set_side_stream)
  bool layout_suspended = false;                 // vo_pipe_step took the gate and the CU mask off (its chain needs the whole chip): the next frame step re-applies them
  bool pub_copy_pending = false;                // a pipelined step has recorded ev_pub at least once
  // loader pre-filter (vo_set_prefilter): cv2.bilateralFilter taps applied while a frame enters the frame store
  int bil_maxk = 0;                  // 0 = off
  signed char bil_dx[49], bil_dy[49];
  float bil_sw[49];
  float* d_bil_cw = nullptr;         // [256] colour weights
  int side_stream = 1;               // vo_set_side_stream
  vo_tuning tune = {};               // forced forms (vo_set_tuning); all zero = the library's rules
  bool in_step = false;              // inside vo_frame_step_resident: its stage calls must not wait for the side streams (vo_quiesce_side)
  bool main_dirty = true;            // an entry point other than vo_pipe_step may have enqueued work on `stream` since the last pipe step (set by
                                     // vo_quiesce_side, which every such entry point calls): the next pipe step orders its side streams behind it
  int batch = 1;
  int width = 0, height = 0, max_pts = 0, max_level = 0, win = 0;
  int top = 0;                       // highest pyramid level index built
  vo_level lv[VO_MAX_LEVELS];
  size_t lvl_px[VO_MAX_LEVELS];      // pixels per sequence per level (= pitch * ph): the per-sequence stride
  vo_frame fr[2];                    // sequence 0 of each buffer; sequence b at + b * lvl_px[l] pixels
  int cur = 0;                       // index of the current frame in fr[]
  int n_pushed = 0;
  uint8_t* d_raw = nullptr;          // staging of one raw frame per sequence
  uint8_t* h_raw = nullptr;          // pinned host staging of vo_frame_push (allocated on first use)
  hipEvent_t ev_raw = nullptr;       // the upload out of h_raw is done
  // vo_frame_step_host: the frames of a step arrive from the host on a copy stream of their own, double-buffered like the steps in flight
  hipStream_t stream_h2d = nullptr;
  uint8_t* d_host_raw[2] = {nullptr, nullptr};      // [batch][h][w], by step parity
  hipEvent_t ev_h2d[2] = {nullptr, nullptr};        // the upload into d_host_raw[k] is complete (recorded on stream_h2d, awaited by the ctx stream)
  hipEvent_t ev_raw_free[2] = {nullptr, nullptr};   // the pyramid has read d_host_raw[k] (recorded on the ctx stream, awaited by stream_h2d)
  bool raw_free_recorded[2] = {false, false};
  int pipe_host_slot = 0;                           // vo_pipe_step_host alternates the two buffers on its own count
  const uint8_t** h_ptr_tab = nullptr;              // page-locked [2][batch]: device-visible addresses of a step's images (read by k_gather_frames)
  uint8_t* d_seq = nullptr;          // preloaded sequences [batch][seq_n][h][w] (vo_seq_upload)
  int seq_n = 0;
  // tracked point sets live in the result slab (off_pa / off_pb, ping-pong selected by p_parity)
  int p_parity = 0;                  // 0: current points at off_pa
  int32_t* d_iters = nullptr;        // [batch][n x (max_level + 1)]
  int n_resident = 0;
  int iters_stride = 0;              // of the last KLT launch
  // DLT inputs
  float* d_uv0 = nullptr; float* d_uv1 = nullptr;    // [batch][max_pts][2]
  vo_dlt_cam* d_dlt_cam = nullptr;   // [batch]
  int dlt_n = 0, dlt_stats = 0;
  vo_st_ws* st = nullptr;
  struct vo_pnp_ws* pnp = nullptr;   // PnP-RANSAC workspace (vo_pnp.hip)
  struct vo_sift_ws* sift = nullptr;     // SIFT scale space and lists (vo_sift.hip)
  struct vo_match_ws* match = nullptr;   // descriptor matcher buffers (vo_match.hip)
  struct vo_ess_ws* ess = nullptr;   // essential-matrix RANSAC workspace (vo_essential.hip)
  struct vo_trk_ws* trk = nullptr;   // device-resident track table (vo_tracks.hip)
  struct vo_pipe_ws* pipe = nullptr; // closed-loop Pipeline.step on the device (vo_pipeline.hip)
  const int32_t* d_pt_counts = nullptr;   // vo_tracks_* only (set by vo_tracks_seed, cleared by vo_trk_destroy): per-sequence number of live tracks the
                                          // PUBLIC resident entry points (vo_klt_track_resident, vo_shi_tomasi_resident) then use; null = uniform n.
                                          // The closed-loop pipeline passes its own counters explicitly (vo_*_resident_counts) and never touches this.
  vo_ba_ws* ba = nullptr;
  vo_prof prof;
  unsigned long long* d_dbg = nullptr;   // 4 x 8 phase stamps (vo_debug_cycles)
  // result slab: every per-frame output of the front end lives in ONE device allocation (mirrored in pinned host
  // memory) so that a frame's results come back with a single D2H copy instead of ten.
  uint8_t* d_slab = nullptr;             // [batch][slab_seq]
  uint8_t* h_slab = nullptr;             // pinned, 2 x slab_bytes: steps alternate between the halves so that step t + 1 can
                                         // be enqueued while the host still reads step t (vo_frame_step_resident / vo_frame_fetch)
  hipEvent_t ev_step[2] = {nullptr, nullptr};   // recorded after the result copies of the step using half k
  size_t step_off_p[2] = {0, 0};         // slab offset of the tracked points of that step
  long steps_enq = 0, steps_fetched = 0;
  size_t slab_seq = 0;                   // bytes per sequence
  size_t slab_bytes = 0;                 // batch * slab_seq
  size_t off_pa = 0, off_pb = 0, off_status = 0, off_err = 0, off_X4 = 0, off_depth = 0, off_reproj = 0,
         off_st_scalars = 0, off_st_out = 0;
  // per-frame step (vo_frame_step_resident) captured as hipGraphs, one per frame parity
  // captured steps, keyed by the hash of everything a capture bakes in: parameters by value, device pointers (the selected problem of a
  // BA bank among them), problem shapes, frame-store parity and the pinned mirror half the results go to
  std::vector<std::pair<uint64_t, hipGraphExec_t>> step_graphs;
  int32_t* d_frame_idx = nullptr;        // frame index consumed by the captured k_pad_level0
  int32_t* h_frame_idx = nullptr;        // pinned ring of frame indices (H2D source must outlive the copy)
  int frame_ring = 0;
  int use_graph = 0;                     // hipGraph replay is opt-in (vo_set_graph_mode): on ROCm 7.2 it is slower than plain launches
  // landmark-sharded bundle adjustment (vo_comm.hip): RCCL communicator of this context, one process per GPU
  void* comm = nullptr;                  // ncclComm_t
  int comm_rank = 0, comm_ranks = 1;
  int ba_sharded = 0;                    // batch entries (and ranks) are landmark shards of ONE problem
  std::string err;
};

inline int32_t vo_fail(vo_ctx* c, int32_t code, const std::string& msg) {
  if (c) c->err = msg;
  return code;
}

#define VO_HIP(c, expr)                                                                   \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      char _b[512];                                                                       \
      snprintf(_b, sizeof(_b), "%s:%d: %s -> %s", __FILE__, __LINE__, #expr,              \
               hipGetErrorString(_e));                                                    \
      return vo_fail((c), VO_E_HIP, _b);                                                  \
    }                                                                                     \
  } while (0)

#define VO_CHECK(c, cond, code, msg)                                                      \
  do {                                                                                    \
    if (!(cond)) return vo_fail((c), (code), std::string(__func__) + ": " + (msg));       \
  } while (0)

// slab accessors (sequence 0; sequence b at + b * slab_seq bytes)
template <class T> inline T* vo_slab(const vo_ctx* c, size_t off) { return reinterpret_cast<T*>(c->d_slab + off); }
inline size_t vo_off_p(const vo_ctx* c) { return c->p_parity ? c->off_pb : c->off_pa; }       // current points
inline size_t vo_off_p_next(const vo_ctx* c) { return c->p_parity ? c->off_pa : c->off_pb; }  // KLT output
// device side: pointer of sequence b given the sequence-0 pointer and the byte stride
template <class T> __device__ __forceinline__ T* vo_seq(T* base, size_t stride_bytes, int b) {
  return reinterpret_cast<T*>(reinterpret_cast<unsigned char*>(const_cast<typename std::remove_const<T>::type*>(base)) + (size_t)b * stride_bytes);
}

// XCD-aware (block, sequence) assignment for grids of `nblk` blocks per sequence x `batch` sequences, batch % 8 == 0:
// workgroups are dealt to the 8 XCDs round-robin in dispatch order and every XCD has its own L2, so with the plain mapping the
// neighbouring blocks of one image land on eight different L2s and each fetches the rows they share.  Remapped, XCD k works on
// sequences k, k + 8, ... one after the other: a sequence's rows are fetched by ONE L2, and what one kernel of the frame chain
// wrote there (plain stores keep the line) the next kernel finds there.  Placement is an observation, not a contract: the
// mapping is a bijection of the grid, so results never depend on it.  id = linear block index in dispatch order.
__device__ __forceinline__ void vo_xcd_assign(unsigned id, unsigned nblk, int remap, int& blk, int& bseq) {
  if (remap) {
    const unsigned q = id >> 3;
    bseq = (int)(id & 7u) + 8 * (int)(q / nblk);
    blk = (int)(q % nblk);
  } else {
    bseq = (int)(id / nblk);
    blk = (int)(id - (unsigned)bseq * nblk);
  }
}

// RAII bracket: records an event pair on the ctx stream around a region when profiling is on
struct vo_prof_scope {
  vo_ctx* c; int region; hipEvent_t e0 = nullptr, e1 = nullptr;
  vo_prof_scope(vo_ctx* c_, int region_);
  ~vo_prof_scope();
};

// phase stamp helper for the diagnostic cycle counters (thread 0 of a workgroup)
#define VO_STAMP(buf, idx) do { if ((buf) && threadIdx.x == 0) (buf)[idx] = __builtin_amdgcn_s_memtime(); } while (0)

static inline int vo_div_up(int a, int b) { return (a + b - 1) / b; }

// cross-unit internals used by the fused frame step (vo_step.hip)
int32_t vo_build_pyramid(vo_ctx* c, const uint8_t* d_raw_img, size_t raw_seq_stride, const int32_t* d_frame_idx);
// a step's `batch` images from the host into d_host_raw[slot] on the copy stream (vo_step.hip); ev_h2d[slot] is recorded behind it
int32_t vo_host_frames_upload(vo_ctx* c, const uint8_t* const* frames, int32_t stride, int slot);
int32_t vo_ba_enqueue_pub_copy(vo_ctx* c, int half);     // half: which pinned mirror (0 / 1)
void vo_ba_unpack_pub(vo_ctx* c, int half, double* poses_out, double* points_out, vo_ba_stats* stats);   // arrays over the batch
bool vo_ba_ready(const vo_ctx* c);
double* vo_ba_obs_device(vo_ctx* c, int* n_slots, int* n_pts);   // resident observation table [batch][W][N][2] of the uploaded problem
bool vo_st_ready(const vo_ctx* c);
int vo_st_last_max_corners(const vo_ctx* c);
int vo_st_launch_state(const vo_ctx* c);
int vo_st_flags_save(const vo_ctx* c);
void vo_st_flags_restore(vo_ctx* c, int saved);
// the pipelined stream layout (vo_set_side_stream 2) leaves work on streams B and C after a step: every entry point outside the
// step / fetch pair that touches the result slab, the frame store, the BA / Shi-Tomasi / DLT / PnP workspaces waits for them first
int32_t vo_quiesce_side(vo_ctx* c);
bool vo_blocking_sync();             // environment VO_BLOCKING_SYNC=1: events a host thread waits on sleep in the driver instead of spinning
int32_t vo_main_stream_reserve(vo_ctx* c, int reserve);               // the ctx stream re-created with / without a CU mask (vo_set_side_stream(c, 2) of a batch)
hipError_t vo_stream_create(hipStream_t* st, int reserve_cus);      // reserve_cus > 0: the queue never uses the last `reserve_cus` bits of the CU mask
int32_t vo_st_prepare(vo_ctx* c, const vo_st_params* prm);
// the resident entry points with the per-sequence counters named by the caller (device arrays [batch], null = uniform): d_counts = live
// points of each sequence (KLT input / exclusion discs), d_limit = cap on the corners each sequence's re-detection needs
int32_t vo_klt_track_resident_counts(vo_ctx* c, int32_t n, const vo_klt_params* prm, const int32_t* d_counts);
int32_t vo_shi_tomasi_resident_counts(vo_ctx* c, int32_t n_cur, int32_t mask_radius, const vo_st_params* prm, const int32_t* d_counts,
                                      const int32_t* d_limit);

// collectives on the ctx stream (vo_comm.hip); identity / device copy without a communicator
int32_t vo_comm_allreduce_f64(vo_ctx* c, double* buf, size_t count);
int32_t vo_comm_allgather_f64(vo_ctx* c, const double* send, double* recv, size_t count);

void vo_trk_destroy(vo_ctx* c);
void vo_pipe_destroy(vo_ctx* c);
bool vo_pipe_busy(const vo_ctx* c);            // closed-loop steps enqueued and not fetched yet

// ---- hooks of the closed-loop pipeline (vo_pipeline.hip) into the stage units: device-resident inputs, per-sequence counts ----
// bundle adjustment: workspace for W slots x N landmark slots with K uploaded, problem written on the device into x0 / obs
// LM state of one bundle-adjustment problem; also the header of the publish buffer [ba_state, padded to VO_BA_PUB_HEADER bytes | x] that
// k_ba_finalize writes and vo_ba_fetch / k_pipe_writeback read
struct ba_state {
  double lambda, nu, cost, cost0;
  int cur, iter, accepted, status, done, n_obs;
};
#define VO_BA_PUB_HEADER 64
static_assert(sizeof(ba_state) <= VO_BA_PUB_HEADER, "the publish buffer's header holds one ba_state");
struct vo_ba_view { double* x0; double* obs; const uint8_t* pub; size_t pub_bytes; size_t x_stride; size_t obs_stride; int W, N; };
int32_t vo_ba_reserve(vo_ctx* c, const double* K_host, int W, int N);
int32_t vo_ba_get_view(vo_ctx* c, vo_ba_view* v);
void vo_ba_set_live(vo_ctx* c, const int32_t* d_counts, int stride);   // landmark slots in use per problem (device counters), or null
int32_t vo_ba_enqueue_budget(vo_ctx* c, const vo_ba_params* prm, int it0, int n_it);   // iterations it0 .. it0 + n_it - 1, then publish
// 3D-2D pose: correspondences written on the device, counts[b] of them per sequence
struct vo_pnp_view { float* X; float* uv; const uint8_t* mask; const double* out; const int32_t* ctrl; size_t ctrl_stride; int cap; };
int32_t vo_pnp_reserve(vo_ctx* c, const double* K_host);
int32_t vo_pnp_get_view(vo_ctx* c, vo_pnp_view* v);
int32_t vo_pnp_enqueue_counts(vo_ctx* c, const vo_pnp_params* prm, int blind_batches, const int32_t* d_counts);
// triangulation of up to n_hi pairs per sequence (c->d_uv0 / d_uv1), counts[b] valid; point i of sequence b uses cams[b][cam_sel[b][i]]
int32_t vo_dlt_enqueue_counts(vo_ctx* c, int n_hi, const int32_t* d_counts, const vo_dlt_cam* d_cams, const int32_t* d_cam_sel, int cams_per_seq);
void vo_pnp_destroy(vo_ctx* c);
void vo_ess_destroy(vo_ctx* c);
void vo_match_destroy(vo_ctx* c);
void vo_sift_destroy(vo_ctx* c);
// sub-workspace lifetime hooks
void vo_st_destroy(vo_ctx* c);
void vo_ba_destroy(vo_ctx* c);
